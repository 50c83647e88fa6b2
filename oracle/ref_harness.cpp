/* oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin extern "C" shim around the UNMODIFIED reference sources, compiled where they lie under
 * /root/reference by oracle/Makefile into oracle/_ref/libgnumap_ref.so.  It exists to pin the
 * from-scratch CPU restatement (oracle/gm_oracle.c) against the real reference functions:
 *
 *   bwa_index                       src/bwtindex.c:187     (index build, .gnumap.{pac,ann,amb,bwt,sa})
 *   bwt_restore_bwt/bwt_restore_sa  src/bwt.c:443,421
 *   bns_restore / pac load          src/bntseq.c:169 ; loader shape of GenomeBwt.cpp:112-140
 *   bwt_occ / bwt_match_exact       src/bwt.c:107,222      (via nst_nt4_table like GenomeBwt.cpp:438-474)
 *   bwt_sa                          src/bwt.c:86
 *   bns_intv2rid / bns_get_seq      src/bntseq.c:365,398   (window fetch, shape of GenomeBwt::GetString :384-415)
 *   setup_alignment_matrices        inc/a_matrices.c:25
 *   bin_seq::get_align_score(…)     src/bin_seq.cpp:739,761
 *   bin_seq::get_align_score_w_traceback  src/bin_seq.cpp:445
 *   SeqReader (FASTQ -> PWM)        src/SeqReader.cpp:1023-1292
 *   bin_seq::pairHMM                src/bin_seq.cpp:60-244 (the --snp deposit of SNPScoredSeq::score, src/SNPScoredSeq.cpp:25-109)
 *
 * The driver-level code (Driver.cpp, align_seq2_raw.cpp, ScoredSeq.h, GenomeBwt.cpp) cannot be
 * compiled here: it includes gsl/gsl_cdf.h (inc/Genome.h:45), an external library the image lacks.
 * Nothing in this file restates reference logic beyond the few glue lines noted above.
 */
#include "const_define.h"           /* defines the reference's globals (gALIGN_SCORES, gGAP, ...) */
const char* pos_matrix = NULL;      /* Driver.cpp:73 */
#include "a_matrices.c"             /* setup_alignment_matrices() */
#include "bin_seq.h"
#include "SeqReader.h"
#include "SequenceOperations.h"
#include "bwt.h"
#include "bntseq.h"
#include <unistd.h>

extern "C" {
int bwa_index(int argc, char* argv[]);
}

struct ref_index {
    bwt_t* bwt;
    bntseq_t* bns;
    uint8_t* pac;
};

static bool g_setup_done = false;

extern "C" {

/* mode: 0 normal, 1 bisulfite (-b, up strand), 2 bisulfite2 / down strand, 3 A->G (-d), 4 A->G down */
int ref_setup(int mode) {
    if (g_setup_done) return -1;    /* setup_alignment_matrices() scales globals in place: once only */
    setup_alignment_matrices();
    /* Driver.cpp:1260-1313: the table edits touch lowercase rows only */
    if (mode == 1) gALIGN_SCORES[(int)'c'][3] = gMATCH;
    if (mode == 2) gALIGN_SCORES[(int)'g'][0] = gMATCH;
    if (mode == 3) gALIGN_SCORES[(unsigned int)'a'][2] = gMATCH;
    if (mode == 4) gALIGN_SCORES[(unsigned int)'t'][1] = gMATCH;
    g_setup_done = true;
    return 0;
}

void ref_get_scores(float* table /*256*4*/, float* gap, int* max_gap) {
    memcpy(table, gALIGN_SCORES, sizeof(float) * 256 * 4);
    *gap = gGAP;
    *max_gap = gMAX_GAP;
}

int ref_index_build(const char* fasta) {
    char* args[2] = { strdup("index"), strdup(fasta) };
    optind = 1;
    int r = bwa_index(2, args);
    free(args[0]); free(args[1]);
    return r;
}

ref_index* ref_index_load(const char* fasta) {
    /* shape of GenomeBwt::bwa_idx_load_from_disk, GenomeBwt.cpp:112-140 */
    char fn[4096];
    ref_index* ix = (ref_index*)calloc(1, sizeof(ref_index));
    snprintf(fn, sizeof fn, "%s.gnumap.bwt", fasta);
    ix->bwt = bwt_restore_bwt(fn);
    snprintf(fn, sizeof fn, "%s.gnumap.sa", fasta);
    bwt_restore_sa(fn, ix->bwt);
    ix->bns = bns_restore(fasta);
    ix->pac = (uint8_t*)calloc(ix->bns->l_pac / 4 + 1, 1);
    size_t got = fread(ix->pac, 1, ix->bns->l_pac / 4 + 1, ix->bns->fp_pac);
    (void)got;
    fclose(ix->bns->fp_pac);
    ix->bns->fp_pac = 0;
    return ix;
}

void ref_index_free(ref_index* ix) {
    if (!ix) return;
    bwt_destroy(ix->bwt);
    bns_destroy(ix->bns);
    free(ix->pac);
    free(ix);
}

uint64_t ref_seq_len(ref_index* ix) { return ix->bwt->seq_len; }
uint64_t ref_primary(ref_index* ix) { return ix->bwt->primary; }
int ref_n_seqs(ref_index* ix) { return ix->bns->n_seqs; }
const char* ref_seq_name(ref_index* ix, int i) { return ix->bns->anns[i].name; }
int64_t ref_seq_offset(ref_index* ix, int i) { return ix->bns->anns[i].offset; }
int ref_seq_length(ref_index* ix, int i) { return ix->bns->anns[i].len; }

uint64_t ref_occ(ref_index* ix, uint64_t k, int c) { return bwt_occ(ix->bwt, k, (ubyte_t)c); }

/* GenomeBwt::get_sa_int, GenomeBwt.cpp:438-474 (glue only: nt4 conversion + (0,0) on no match) */
void ref_sa_interval(ref_index* ix, const char* kmer, int len, uint64_t* start, uint64_t* end) {
    unsigned char* c_seq = new unsigned char[len + 1];
    for (int i = 0; i < len; ++i)
        c_seq[i] = kmer[i] < 4 ? kmer[i] : nst_nt4_table[(int)kmer[i]];
    bwtint_t s, e;
    int r = bwt_match_exact(ix->bwt, len, c_seq, &s, &e);
    delete[] c_seq;
    if (r > 0) { *start = s; *end = e; } else { *start = 0; *end = 0; }
}

uint64_t ref_sa_coord(ref_index* ix, uint64_t k) { return bwt_sa(ix->bwt, k); }

/* GenomeBwt::GetString, GenomeBwt.cpp:384-415: returns length written (0 = boundary) */
int ref_window(ref_index* ix, uint64_t begin, unsigned size, char* out) {
    int rid = bns_intv2rid(ix->bns, begin, begin + size);
    if (rid < 0) { out[0] = 0; return 0; }
    int64_t rlen;
    uint8_t* rseq = bns_get_seq(ix->bns->l_pac, ix->pac, begin, begin + size, &rlen);
    for (int i = 0; i < rlen; ++i) out[i] = "acgtn"[(int)rseq[i]];
    out[rlen] = 0;
    free(rseq);
    return (int)rlen;
}

int ref_pos2rid(ref_index* ix, int64_t pos) { return bns_pos2rid(ix->bns, pos); }

static Read make_read(const float* pwm, int L, float**& rows) {
    rows = new float*[L];
    for (int i = 0; i < L; ++i) {
        rows[i] = new float[4];
        for (int j = 0; j < 4; ++j) rows[i][j] = pwm[i * 4 + j];
    }
    return Read(rows, L);
}
static void free_rows(float** rows, int L) {
    for (int i = 0; i < L; ++i) delete[] rows[i];
    delete[] rows;
}

/* bin_seq::get_align_score(const Read&, const string&), bin_seq.cpp:761 */
float ref_nw_score(const float* pwm, int L, const char* window) {
    float** rows;
    Read r = make_read(pwm, L, rows);
    bin_seq bs;
    float s = bs.get_align_score(r, std::string(window));
    free_rows(rows, L);
    return s;
}

/* scores n windows with ONE bin_seq instance (scratch reuse as in align_sequence, align_seq2_raw.cpp:184) */
void ref_nw_score_many(const float* pwm, int L, const char* windows /* n*(L+1) */, int n, float* out) {
    float** rows;
    Read r = make_read(pwm, L, rows);
    bin_seq bs;
    for (int i = 0; i < n; ++i) out[i] = bs.get_align_score(r, std::string(windows + (size_t)i * (L + 1)));
    free_rows(rows, L);
}

/* set_top_matches self score, Driver.cpp:466: get_align_score(read, consensus, 0, L-1) */
float ref_self_score(const float* pwm, int L, const char* cons) {
    float** rows;
    Read r = make_read(pwm, L, rows);
    bin_seq bs;
    float s = bs.get_align_score(r, std::string(cons), 0u, (unsigned)(L - 1));
    free_rows(rows, L);
    return s;
}

float ref_align_score_be(const float* pwm, int L, const char* gen, unsigned begin, unsigned end) {
    float** rows;
    Read r = make_read(pwm, L, rows);
    bin_seq bs;
    float s = bs.get_align_score(r, std::string(gen), begin, end);
    free_rows(rows, L);
    return s;
}

/* bin_seq::get_align_score_w_traceback, bin_seq.cpp:445; outputs must hold 2L+2 / 1024 bytes */
void ref_traceback(const float* pwm, int L, const char* cons, const char* window, char* aligned, int* aligned_len, char* cigar) {
    float** rows;
    Read r = make_read(pwm, L, rows);
    bin_seq bs;
    std::pair<std::string, std::string> res = bs.get_align_score_w_traceback(r, std::string(cons), std::string(window));
    *aligned_len = (int)res.first.size();
    memcpy(aligned, res.first.data(), res.first.size());   /* may contain '\0' (the consense[i] quirk) */
    aligned[res.first.size()] = 0;
    strcpy(cigar, res.second.c_str());
    free_rows(rows, L);
}

/* bin_seq::pairHMM, bin_seq.cpp:60: out = window length x NUM_SNP_VALS (5: this build has no _INDEL) floats */
void ref_pair_hmm(const float* pwm, int L, const char* cons, const char* window, float* out) {
    static bool init = false;
    if (!init) { InitProg(); init = true; }      /* the program's start-up (Driver.cpp main): fills g_gen_CONVERSION, which pairHMM indexes */
    float** rows;
    Read r = make_read(pwm, L, rows);
    bin_seq bs;
    const std::string g(window);
    float** res = bs.pairHMM(r, std::string(cons), g);
    for (size_t i = 0; i < g.size(); ++i) { for (int k = 0; k < NUM_SNP_VALS; ++k) out[i * NUM_SNP_VALS + k] = res[i][k]; delete[] res[i]; }
    delete[] res;
    free_rows(rows, L);
}

/* FASTQ -> Read via the reference's own parser.  Returns number of reads parsed (<= cap).
 * pwm: cap * max_len * 4 floats; seq/fq: cap * (max_len+1) chars; name: cap * 256 */
int ref_read_fastq(const char* fn, int illumina, int cap, int max_len, float* pwm, int* lens, char* seq, char* fq, char* names) {
    gILLUMINA = illumina != 0;
    SeqReader sr;
    sr.use(fn);      /* Init + first ReadBatch, SeqReader.cpp:251-256 */
    int n = 0;
    while (n < cap) {
        Read* r = sr.GetNextSequence();
        if (!r) break;
        if ((int)r->length > max_len) { delete_read(r); return -2; }
        lens[n] = r->length;
        for (unsigned i = 0; i < r->length; ++i)
            for (int j = 0; j < 4; ++j) pwm[((size_t)n * max_len + i) * 4 + j] = r->pwm[i][j];
        strncpy(seq + (size_t)n * (max_len + 1), r->seq.c_str(), max_len + 1);
        strncpy(fq + (size_t)n * (max_len + 1), r->fq.c_str(), max_len);
        fq[(size_t)n * (max_len + 1) + max_len] = 0;
        strncpy(names + (size_t)n * 256, r->name, 255);
        names[(size_t)n * 256 + 255] = 0;
        delete_read(r);
        ++n;
    }
    return n;
}

/* SequenceOperations.h helpers used on the output side */
void ref_reverse_comp(const char* s, char* out) {
    std::string t(s);
    std::string r = reverse_comp(t);
    strcpy(out, r.c_str());
}
void ref_reverse_cigar(const char* s, char* out) {
    char buf[1024];
    strncpy(buf, s, 1023); buf[1023] = 0;
    std::string r = reverse_CIGAR(buf);
    strcpy(out, r.c_str());
}
void ref_fix_cigar(const char* s, char* out) {
    std::string t(s);
    if (t.size()) fix_CIGAR_for_deletions(t);
    strcpy(out, t.c_str());
}

} /* extern "C" */
