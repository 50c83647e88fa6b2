/* oracle/gm_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of GNUMAP's per-read seed-and-extend hot path, written from the
 * reference's behaviour (file:line cited at every function in gm_oracle.c).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * (gnumap_amd/, include/) never links, imports or executes anything under oracle/.
 *
 * Parity pin (both levels pinned to the UNMODIFIED reference compiled by oracle/Makefile):
 *  - function level: FM-index query (occ / SA interval / locate), window fetch, FASTQ->PWM, self score, banded NW score and
 *    traceback/CIGAR, reverse_comp / reverse_CIGAR / fix_CIGAR_for_deletions against oracle/_ref/libgnumap_ref.so (vectors in
 *    tests/golden/ref_vectors.npz made by tests/golden/make_fixtures.py, checked by tests/test_oracle_golden.py) and the
 *    known answers of bin_seq::Test (src/bin_seq.cpp:1095-1127);
 *  - driver level (adaptive k-mer walk, voting, accept test, unique map, denominator, winner, MAPQ, SAM text, .sgr, .gmp:
 *    inc/align_seq2_raw.cpp:22-328, src/Driver.cpp:432-753,2146-2217, inc/ScoredSeq.h:293-404, src/*ScoredSeq.cpp,
 *    src/GenomeBwt.cpp:483-490,1092-1273): gmo_run writes byte-identical SAM (record order included), .sgr and .gmp to the
 *    reference PROGRAM oracle/_ref/gnumap_ref (src/Driver.cpp etc. compiled in place) in the 30 modes of
 *    tests/golden/ref_runs/manifest.json (made by tests/golden/make_driver_fixtures.py, checked by tests/test_driver_golden.py).
 */
#ifndef GM_ORACLE_H
#define GM_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    char* name;
    uint64_t offset;
    uint32_t len;
} gmo_contig;

typedef struct {
    uint64_t primary, L2[5], seq_len, bwt_size;
    uint32_t* bwt;              /* occ-interleaved BWT exactly as in <fa>.gnumap.bwt */
    uint64_t sa_intv, n_sa;
    uint64_t* sa;               /* rank-sampled SA, sa[0] = (u64)-1 */
    uint64_t l_pac;
    uint8_t* pac;               /* 2 bit/base, 4 per byte MSB first */
    int n_seqs;
    gmo_contig* contigs;
} gmo_index;

enum { GMO_MODE_NORMAL = 0, GMO_MODE_BS = 1, GMO_MODE_BS2 = 2, GMO_MODE_ATOG = 3, GMO_MODE_ATOG2 = 4, GMO_MODE_SNP = 5 /* --snp: SNPScoredSeq */ };

typedef struct {
    int mer, jump, min_seed_hits;
    uint32_t max_kmer_hits, max_matches;
    int max_gap, nw, fast, unique_only, pos_strand, neg_strand, mode;
    float align_score;          /* -a */
    int align_is_fraction;      /* perc */
    float cutoff;               /* -q */
    float adjust, match, transition, transversion, gap;   /* after scaling by adjust */
    float S[256][4];
    int bin_size, print_all_sam, illumina;
} gmo_params;

typedef struct { uint64_t pos; int strand; } gmo_pos;

typedef struct {
    char* key;                  /* unique-map key: window in read orientation */
    char* seq;                  /* genomic window of the FIRST hit (ScoredSeq::sequence) */
    double score;               /* align score of the FIRST hit */
    int first_strand;
    gmo_pos* pos; int n_pos, cap_pos;   /* ordered set of (pos,strand) */
} gmo_hit;

typedef struct {
    uint64_t kmers, occ_calls, occ_blocks, locates, lf_steps, nw, tracebacks;
} gmo_counters;

enum { GMO_OK = 0, GMO_TOO_MANY = 1, GMO_NONE = 2, GMO_TOO_SHORT = -2, GMO_TOO_POOR = -3 };

typedef struct {
    int status;
    float self_score;
    double min_score, top_score, denominator;
    gmo_hit* hits; int n_hits, cap_hits;       /* ascending key order (std::map order) */
    gmo_counters ctr;
} gmo_result;

typedef struct {
    uint64_t pos; int strand; int contig; uint64_t chr_pos /*1-based*/;
    int mapq; char cigar[1024];
    float a_score, post_prob; int sim_matches;
} gmo_sam;

typedef struct { uint64_t pos; uint32_t span; float w; uint8_t* codes; /* -b/-d: g_gen_CONVERSION of the gapped read string per base, else NULL */
                 float* hmm; /* --snp: span x 5 pair-HMM posteriors (a, c, g, t, n) in the orientation of this position, else NULL */ } gmo_deposit;

/* ---- index ---- */
gmo_index* gmo_index_load(const char* fasta_prefix);
void gmo_index_free(gmo_index*);
uint64_t gmo_occ(const gmo_index*, uint64_t k, int c, gmo_counters*);
int gmo_sa_interval(const gmo_index*, const char* kmer, int m, uint64_t* start, uint64_t* end, gmo_counters*);
uint64_t gmo_locate(const gmo_index*, uint64_t k, gmo_counters*);
int gmo_window(const gmo_index*, uint64_t begin, uint32_t L, char* out);
/* bin_seq::pairHMM src/bin_seq.cpp:60-244: out[m][5] = per window position the posterior weight of a, c, g, t, n of the read (n x 4 PWM, argmax consensus) */
void gmo_pair_hmm(const float* pwm, int n, const char* cons, const char* genome, int m, float* out);
int gmo_pos2rid(const gmo_index*, int64_t pos);

/* ---- params / scoring ---- */
void gmo_params_default(gmo_params*);
void gmo_params_finalize(gmo_params*);      /* builds S, scales by adjust, derives jump */
int gmo_pwm_from_fastq(const char* seq, const char* qual, int L, int* illumina, float* pwm);
void gmo_revcomp_pwm(const float* pwm, int L, float* out);
void gmo_revcomp_str(const char* s, int L, char* out);
float gmo_self_score(const gmo_params*, const float* pwm, const char* cons, int L);
float gmo_nw_score(const gmo_params*, const float* pwm, int L, const char* window);
float gmo_align_score_be(const gmo_params*, const float* pwm, int L, const char* gen, unsigned begin, unsigned end);
int gmo_traceback(const gmo_params*, const float* pwm, int L, const char* cons, const char* window,
                  char* aligned /*2L+2*/, int* aligned_len, char* cigar /*1024*/);
void gmo_fix_cigar(char* cigar);
void gmo_reverse_cigar(const char* in, char* out);

/* ---- per read ---- */
void gmo_map_read(const gmo_index*, const gmo_params*, const float* pwm, const char* cons, int L, gmo_result* out);
void gmo_result_free(gmo_result*);
/* SAM records + coverage deposits of one mapped read; returns #records (caller frees *recs, *deps) */
int gmo_read_output(const gmo_index*, const gmo_params*, const gmo_result*, const float* pwm, const char* cons, int L,
                    gmo_sam** recs, gmo_deposit** deps, int* n_deps, gmo_counters*);
int gmo_format_sam(const gmo_index*, const gmo_params*, const gmo_sam*, const char* name, const char* cons, const char* qual,
                   char* out, size_t cap);

/* ---- whole run: FASTQ -> <out>.sam + <out>.sgr (deterministic, = reference -c 1 order) ---- */
typedef struct {
    uint64_t n_reads, n_matched, n_records;
    double map_seconds;             /* wall time of the mapping loop only (no I/O, no index load) */
    gmo_counters ctr;
} gmo_run_stats;
int gmo_run(const gmo_index*, const gmo_params*, const char* fastq, const char* out_prefix, int threads,
            uint64_t max_reads, const char* cmdline, gmo_run_stats*);

#ifdef __cplusplus
}
#endif
#endif
