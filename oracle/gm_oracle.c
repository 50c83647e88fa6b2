/* oracle/gm_oracle.c — TEST INFRASTRUCTURE ONLY (see gm_oracle.h for the rules and the parity pin).
 *
 * From-scratch CPU restatement of GNUMAP's seed-and-extend hot path.  Every function cites the
 * reference file:line (relative to /root/reference) whose behaviour it follows.  Compiled with
 * -ffp-contract=off: the reference (g++ -m64 -O3, no -march) never fuses a*b+c.
 */
#define _GNU_SOURCE
#include "gm_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <ctype.h>
#include <pthread.h>
#include <sys/time.h>

#define NEG_INF (-100000.0f)        /* inc/bin_seq.h:38 */
#define SAME_DIFF 0.00001           /* inc/const_include.h:187 */
#define MAX_NAME_SZ 1024            /* inc/const_include.h:46 */
#define MIN_PRINT 0.001             /* GenomeBwt.cpp:928,1261 */

static double now_s(void) {
    struct timeval tp;
    gettimeofday(&tp, NULL);
    return (double)tp.tv_sec + (double)tp.tv_usec * 1e-6;
}

/* ============================================================================================
 * Index files  (bwt_restore_bwt bwt.c:443, bwt_restore_sa bwt.c:421, bns_restore_core bntseq.c:98,
 *               pac read GenomeBwt.cpp:131-134)
 * ==========================================================================================*/
static void* slurp(const char* fn, size_t* size) {
    FILE* f = fopen(fn, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* buf = malloc((size_t)n + 16);
    if (!buf) { fclose(f); return NULL; }
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) { free(buf); fclose(f); return NULL; }
    fclose(f);
    *size = (size_t)n;
    return buf;
}

gmo_index* gmo_index_load(const char* prefix) {
    char fn[4096];
    size_t sz;
    gmo_index* ix = (gmo_index*)calloc(1, sizeof *ix);
    /* .bwt: primary u64, L2[1..4] u64, then bwt_size u32 words (bwt.c:385-393, 443-461) */
    snprintf(fn, sizeof fn, "%s.gnumap.bwt", prefix);
    uint8_t* raw = (uint8_t*)slurp(fn, &sz);
    if (!raw || sz < 40) { free(raw); free(ix); return NULL; }
    memcpy(&ix->primary, raw, 8);
    ix->L2[0] = 0;
    memcpy(&ix->L2[1], raw + 8, 32);
    ix->bwt_size = (sz - 40) >> 2;
    ix->bwt = (uint32_t*)malloc(ix->bwt_size * 4 + 64);
    memcpy(ix->bwt, raw + 40, ix->bwt_size * 4);
    free(raw);
    ix->seq_len = ix->L2[4];
    /* .sa: primary, 4 skipped u64, sa_intv, seq_len, then n_sa-1 u64 (bwt.c:395-441) */
    snprintf(fn, sizeof fn, "%s.gnumap.sa", prefix);
    raw = (uint8_t*)slurp(fn, &sz);
    if (!raw || sz < 56) { free(raw); gmo_index_free(ix); return NULL; }
    uint64_t primary, seq_len;
    memcpy(&primary, raw, 8);
    memcpy(&ix->sa_intv, raw + 40, 8);
    memcpy(&seq_len, raw + 48, 8);
    if (primary != ix->primary || seq_len != ix->seq_len) { free(raw); gmo_index_free(ix); return NULL; }
    ix->n_sa = (ix->seq_len + ix->sa_intv) / ix->sa_intv;
    ix->sa = (uint64_t*)calloc(ix->n_sa, 8);
    ix->sa[0] = (uint64_t)-1;
    if (sz - 56 < (ix->n_sa - 1) * 8) { free(raw); gmo_index_free(ix); return NULL; }
    memcpy(ix->sa + 1, raw + 56, (ix->n_sa - 1) * 8);
    free(raw);
    /* .ann (bntseq.c:74-81,107-136): "l_pac n_seqs seed" then per contig "gi name[ comment]" / "offset len n_ambs" */
    snprintf(fn, sizeof fn, "%s.gnumap.ann", prefix);
    FILE* f = fopen(fn, "r");
    if (!f) { gmo_index_free(ix); return NULL; }
    long long xx; unsigned seed;
    if (fscanf(f, "%lld%d%u", &xx, &ix->n_seqs, &seed) != 3) { fclose(f); gmo_index_free(ix); return NULL; }
    ix->l_pac = (uint64_t)xx;
    ix->contigs = (gmo_contig*)calloc((size_t)ix->n_seqs, sizeof(gmo_contig));
    for (int i = 0; i < ix->n_seqs; ++i) {
        unsigned gi; char str[8192]; int c, len, nambs;
        if (fscanf(f, "%u%8191s", &gi, str) != 2) { fclose(f); gmo_index_free(ix); return NULL; }
        ix->contigs[i].name = strdup(str);
        while ((c = fgetc(f)) != '\n' && c != EOF) {}
        if (fscanf(f, "%lld%d%d", &xx, &len, &nambs) != 3) { fclose(f); gmo_index_free(ix); return NULL; }
        ix->contigs[i].offset = (uint64_t)xx;
        ix->contigs[i].len = (uint32_t)len;
    }
    fclose(f);
    /* .pac: l_pac/4+1 bytes are read (GenomeBwt.cpp:131-133) */
    snprintf(fn, sizeof fn, "%s.gnumap.pac", prefix);
    raw = (uint8_t*)slurp(fn, &sz);
    if (!raw || sz < ix->l_pac / 4 + 1) { free(raw); gmo_index_free(ix); return NULL; }
    ix->pac = raw;
    return ix;
}

void gmo_index_free(gmo_index* ix) {
    if (!ix) return;
    free(ix->bwt); free(ix->sa); free(ix->pac);
    if (ix->contigs) for (int i = 0; i < ix->n_seqs; ++i) free(ix->contigs[i].name);
    free(ix->contigs);
    free(ix);
}

/* ============================================================================================
 * FM-index query
 * ==========================================================================================*/
/* number of bases == c among the first `upto` (0..32) bases of a 64-bit word holding 32 bases MSB first.
 * (what __occ_aux + the tail mask compute, bwt.c:98-105,124-126, restated with a popcount) */
static inline unsigned count_base(uint64_t w, int c, unsigned upto) {
    static const uint64_t pat[4] = { 0x0000000000000000ull, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull, 0xFFFFFFFFFFFFFFFFull };
    uint64_t x = w ^ pat[c];                    /* 00 where the base equals c */
    uint64_t m = ~(x | (x >> 1)) & 0x5555555555555555ull;
    if (upto == 0) return 0;
    if (upto < 32) m &= ~0ull << (2 * (32 - upto));
    return (unsigned)__builtin_popcountll(m);
}

/* bwt_occ, bwt.c:107-129: occurrences of c in BWT[0..k] */
uint64_t gmo_occ(const gmo_index* ix, uint64_t k, int c, gmo_counters* ctr) {
    if (ctr) ctr->occ_calls++;
    if (k == ix->seq_len) return ix->L2[c + 1] - ix->L2[c];
    if (k == (uint64_t)-1) return 0;
    k -= (k >= ix->primary);                    /* $ is not stored */
    const uint32_t* blk = ix->bwt + ((k >> 7) << 4);     /* 64-byte block per 128 bases (bwt.h:73) */
    if (ctr) ctr->occ_blocks++;
    uint64_t n;
    memcpy(&n, blk + 2 * c, 8);                 /* cumulative count before the block */
    const uint32_t* p = blk + 8;
    unsigned within = (unsigned)(k & 127) + 1;  /* bases [block start .. k] inclusive */
    for (unsigned w = 0; within > 0; ++w) {
        unsigned take = within > 32 ? 32 : within;
        uint64_t word = (uint64_t)p[2 * w] << 32 | p[2 * w + 1];
        n += count_base(word, c, take);
        within -= take;
    }
    return n;
}

/* nst_nt4_table, bntseq.c:47-64: ACGT/acgt -> 0..3, everything else 4 */
static inline int nt4(unsigned char ch) {
    switch (ch) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return ch < 4 ? ch : 4;        /* get_sa_int passes codes < 4 through, GenomeBwt.cpp:446 */
    }
}

/* bwt_match_exact bwt.c:222-239 through GenomeBwt::get_sa_int GenomeBwt.cpp:438-474.
 * Returns number of hits; (0,0) in *start,*end on no match.  *fail_step (optional) = number of
 * characters consumed when the search died (diagnostic only). */
int gmo_sa_interval(const gmo_index* ix, const char* kmer, int m, uint64_t* start, uint64_t* end, gmo_counters* ctr) {
    uint64_t k = 0, l = ix->seq_len;
    if (ctr) ctr->kmers++;
    *start = 0; *end = 0;
    for (int i = m - 1; i >= 0; --i) {
        int c = nt4((unsigned char)kmer[i]);
        if (c > 3) return 0;
        uint64_t ok = gmo_occ(ix, k - 1, c, ctr);      /* bwt_2occ(k-1,l,c) == two bwt_occ calls, bwt.c:132-163 */
        uint64_t ol = gmo_occ(ix, l, c, ctr);
        k = ix->L2[c] + ok + 1;
        l = ix->L2[c] + ol;
        if (k > l) return 0;
    }
    *start = k; *end = l;
    return (int)(l - k + 1);
}

/* bwt_invPsi bwt.c:53-59 */
static inline uint64_t inv_psi(const gmo_index* ix, uint64_t k, gmo_counters* ctr) {
    uint64_t x = k - (k > ix->primary);
    uint32_t word = ix->bwt[((x >> 7) << 4) + 8 + ((x & 0x7f) >> 4)];       /* bwt_bwt, bwt.h:72 */
    int c = (int)(word >> ((~x & 0xf) << 1) & 3);                           /* bwt_B0, bwt.h:78 */
    uint64_t r = ix->L2[c] + gmo_occ(ix, k, c, ctr);
    return k == ix->primary ? 0 : r;
}

/* bwt_sa bwt.c:86-96 (GenomeBwt::get_sa_coord GenomeBwt.cpp:431) */
uint64_t gmo_locate(const gmo_index* ix, uint64_t k, gmo_counters* ctr) {
    uint64_t sa = 0, mask = ix->sa_intv - 1;
    if (ctr) ctr->locates++;
    while (k & mask) {
        ++sa;
        k = inv_psi(ix, k, ctr);
        if (ctr) ctr->lf_steps++;
    }
    return sa + ix->sa[k / ix->sa_intv];
}

/* bns_pos2rid bntseq.c:349-363 */
int gmo_pos2rid(const gmo_index* ix, int64_t pos) {
    if (pos < 0 || (uint64_t)pos >= ix->l_pac) return -1;
    int lo = 0, hi = ix->n_seqs - 1;
    while (lo < hi) {                            /* last contig whose offset <= pos */
        int mid = (lo + hi + 1) >> 1;
        if ((uint64_t)pos >= ix->contigs[mid].offset) lo = mid; else hi = mid - 1;
    }
    return lo;
}

/* GenomeBwt::GetString GenomeBwt.cpp:384-415 -> bns_intv2rid bntseq.c:365-373, bns_get_seq :398-419.
 * Writes L lowercase bases + NUL; returns L, or 0 when the window is not inside one contig. */
int gmo_window(const gmo_index* ix, uint64_t begin, uint32_t L, char* out) {
    uint64_t end = begin + L;
    out[0] = 0;
    if (begin < ix->l_pac && end > ix->l_pac) return 0;
    if (begin >= ix->l_pac) return 0;            /* never produced by locate; reference would look at the reverse strand */
    if (L == 0) return 0;
    if (gmo_pos2rid(ix, (int64_t)begin) != gmo_pos2rid(ix, (int64_t)(end - 1))) return 0;
    for (uint32_t t = 0; t < L; ++t) {
        uint64_t p = begin + t;
        out[t] = "acgt"[ix->pac[p >> 2] >> ((~p & 3) << 1) & 3];          /* _get_pac bntseq.c:225 */
    }
    out[L] = 0;
    return (int)L;
}

/* ============================================================================================
 * Parameters and scoring tables  (const_define.h:61-90, a_matrices.c:55-83, Driver.cpp:1206,1260-1313)
 * ==========================================================================================*/
void gmo_params_default(gmo_params* p) {
    memset(p, 0, sizeof *p);
    p->mer = 10;                /* DEF_MER_SIZE const_define.h:46 */
    p->jump = 0;                /* -> mer/2, Driver.cpp:1206 */
    p->min_seed_hits = 2;       /* gMIN_JUMP_MATCHES */
    p->max_kmer_hits = 0;       /* gMAX_KMER_SIZE (-h), 0 = unlimited */
    p->max_matches = 1000;      /* gMAX_MATCHES (-T) */
    p->max_gap = 3;
    p->nw = 1; p->fast = 0; p->unique_only = 0; p->pos_strand = 1; p->neg_strand = 1;
    p->mode = GMO_MODE_NORMAL;
    p->align_score = 0.9f; p->align_is_fraction = 1; p->cutoff = 0.0f;
    p->adjust = 0.25f; p->match = 3; p->transition = -2; p->transversion = -3; p->gap = -4;
    p->bin_size = 8; p->print_all_sam = 0; p->illumina = 0;
}

void gmo_params_finalize(gmo_params* p) {
    if (p->jump <= 0) p->jump = p->mer / 2;
    /* a_matrices.c:55-58 */
    p->match *= p->adjust; p->transition *= p->adjust; p->transversion *= p->adjust; p->gap *= p->adjust;
    for (int i = 0; i < 256; ++i) for (int j = 0; j < 4; ++j) p->S[i][j] = p->transversion;
    const char lo[4] = { 'a', 'c', 'g', 't' }, up[4] = { 'A', 'C', 'G', 'T' };
    for (int g = 0; g < 4; ++g)
        for (int b = 0; b < 4; ++b) {
            float v = (g == b) ? p->match : ((g ^ b) == 2 ? p->transition : p->transversion);   /* a<->g, c<->t */
            p->S[(int)lo[g]][b] = v;
            p->S[(int)up[g]][b] = v;
        }
    /* lowercase rows only, Driver.cpp:1266,1276,1301,1311 */
    if (p->mode == GMO_MODE_BS)    p->S['c'][3] = p->match;
    if (p->mode == GMO_MODE_BS2)   p->S['g'][0] = p->match;
    if (p->mode == GMO_MODE_ATOG)  p->S['a'][2] = p->match;
    if (p->mode == GMO_MODE_ATOG2) p->S['t'][1] = p->match;
    if (p->mode != GMO_MODE_NORMAL) p->bin_size = 1;        /* Driver.cpp:2805-2816 */
}

/* SeqReader::get_more_fastq SeqReader.cpp:1152-1243, Q2Prb_std :623-627, Q2Prb_ill :618-622.
 * pwm is L x 4 floats.  Returns 0, or -1 for a negative probability (reference throws). */
int gmo_pwm_from_fastq(const char* seq, const char* qual, int L, int* illumina, float* pwm) {
    for (int i = 0; i < L; ++i) {
        int Q = (int)qual[i];
        double p;
        if (*illumina) {
            Q -= 64;
            p = 1.0 - 1.0 / pow(10.0, ((double)Q / 10.0));
        } else {
            Q -= 33;
            p = 1 - exp((-(double)Q / 10.0) * log(10.0));
        }
        if (p > 1.0) p = 1.0;
        double other = (1 - p) / 3;
        if (p < 0) {
            if (*illumina) { *illumina = 0; i = -1; continue; }     /* SeqReader.cpp:1171-1180: switch off, redo this read */
            return -1;
        }
        float* row = pwm + 4 * i;
        row[0] = row[1] = row[2] = row[3] = (float)other;
        switch (tolower((unsigned char)seq[i])) {
            case 'a': row[0] = (float)p; break;
            case 'c': row[1] = (float)p; break;
            case 'g': row[2] = (float)p; break;
            case 't': row[3] = (float)p; break;
            default: break;
        }
    }
    return 0;
}

/* reverse_comp_cpy SequenceOperations.h:149-161 */
void gmo_revcomp_pwm(const float* pwm, int L, float* out) {
    for (int i = 0; i < L; ++i) {
        float* d = out + 4 * (L - 1 - i);
        const float* s = pwm + 4 * i;
        d[0] = s[3]; d[1] = s[2]; d[2] = s[1]; d[3] = s[0];
    }
}

/* reverse_comp SequenceOperations.h:56-96 */
void gmo_revcomp_str(const char* s, int L, char* out) {
    for (int i = 0; i < L; ++i) {
        char c = s[L - 1 - i], r;
        switch (c) {
            case 'a': r = 't'; break; case 'A': r = 'T'; break;
            case 't': r = 'a'; break; case 'T': r = 'A'; break;
            case 'c': r = 'g'; break; case 'C': r = 'G'; break;
            case 'g': r = 'c'; break; case 'G': r = 'C'; break;
            case '-': r = '-'; break;
            default: r = 'n'; break;
        }
        out[i] = r;
    }
    out[L] = 0;
}

/* bin_seq::get_val bin_seq.cpp:975-987: ((r0*s0 + r1*s1) + r2*s2) + r3*s3, fp32, unfused */
static inline float get_val(const gmo_params* p, const float* row, unsigned char g) {
    const float* s = p->S[g];
    float t0 = row[0] * s[0], t1 = row[1] * s[1], t2 = row[2] * s[2], t3 = row[3] * s[3];
    float a = t0 + t1;
    a = a + t2;
    a = a + t3;
    return a;
}

/* bin_seq::max_flt bin_seq.cpp:1013-1026 */
static inline float max3(float a, float b, float c) {
    if (a >= b) return a >= c ? a : c;
    return b >= c ? b : c;
}

/* get_align_score_mid bin_seq.cpp:860-893 called as get_align_score(read,cons,0,L-1) bin_seq.cpp:739-759 */
float gmo_self_score(const gmo_params* p, const float* pwm, const char* cons, int L) {
    float score = 0;
    for (int i = 0; i < L; ++i) score += get_val(p, pwm + 4 * i, (unsigned char)cons[i]);
    float value = 0;
    value += 0.0f; value += score; value += 0.0f;      /* begin(0)=0, mid, end(L-1)=0 */
    return value;
}

/* get_align_score_begin(read, gen, end) bin_seq.cpp:781-850.  Band storage: only |i-j|<=G+1 is ever
 * read, so the (end+1)^2 scratch matrix of the reference is kept as rows of 2G+3 cells. */
static float nw_begin(const gmo_params* p, const float* pwm, const char* gen, int end) {
    if (end == 0) return 0;
    const int G = p->max_gap, W = 2 * G + 3;    /* offsets d = j-i in [-(G+1), G+1] */
    float* nm = (float*)malloc(sizeof(float) * (size_t)(end + 1) * (size_t)W);
#define NM(i, j) nm[(size_t)(i) * W + ((j) - (i) + G + 1)]
    for (int i = 0; i <= end; ++i)
        for (int j = i - G - 1; j <= i + G + 1 && j <= end; ++j)
            if (j >= 0) NM(i, j) = NEG_INF;
    for (int i = end; i > end - G - 2 && i >= 0; --i) NM(i, end) = p->gap * (float)(unsigned)(end - i);
    for (int j = end; j > end - G - 2 && j >= 0; --j) NM(end, j) = p->gap * (float)(unsigned)(end - j);
    for (int i = end - 1; i >= 0; --i)
        for (int j = i + G; j >= i - G; --j) {
            if (j >= end) continue;
            if (j < 0) break;
            float mm = NM(i + 1, j + 1) + get_val(p, pwm + 4 * i, (unsigned char)gen[j]);
            float g1 = NM(i + 1, j) + p->gap;
            float g2 = NM(i, j + 1) + p->gap;
            NM(i, j) = max3(mm, g1, g2);
        }
    float r = NM(0, 0);
#undef NM
    free(nm);
    return r;
}

/* get_align_score_end(read, gen, start) bin_seq.cpp:908-972 */
static float nw_end(const gmo_params* p, const float* pwm, int L, const char* gen, int start) {
    if (start == L - 1) return 0;
    const int G = p->max_gap, W = 2 * G + 3;
    int length = L - start;
    float* nm = (float*)malloc(sizeof(float) * (size_t)length * (size_t)W);
#define NM(i, j) nm[(size_t)(i) * W + ((j) - (i) + G + 1)]
    for (int i = 0; i < length; ++i)
        for (int j = i - G - 1; j <= i + G + 1 && j < length; ++j)
            if (j >= 0) NM(i, j) = NEG_INF;
    for (int i = 0; i <= G + 1 && i < length; ++i) NM(i, 0) = p->gap * (float)(unsigned)i;
    for (int j = 0; j <= G + 1 && j < length; ++j) NM(0, j) = p->gap * (float)(unsigned)j;
    for (int i = 1; i < length; ++i)
        for (int j = i - G; j <= i + G && j < length; ++j) {
            if (j <= 0) continue;
            float mm = NM(i - 1, j - 1) + get_val(p, pwm + 4 * (i + start), (unsigned char)gen[j + start]);
            float g1 = NM(i, j - 1) + p->gap;
            float g2 = NM(i - 1, j) + p->gap;
            NM(i, j) = max3(mm, g1, g2);
        }
    float r = NM(length - 1, length - 1);
#undef NM
    free(nm);
    return r;
}

/* bin_seq::get_align_score(read, gen) bin_seq.cpp:761-767 */
float gmo_nw_score(const gmo_params* p, const float* pwm, int L, const char* window) {
    return nw_begin(p, pwm, window, L);
}

/* bin_seq::get_align_score(read, gen, begin, end) bin_seq.cpp:739-759 */
float gmo_align_score_be(const gmo_params* p, const float* pwm, int L, const char* gen, unsigned begin, unsigned end) {
    float value = 0;
    value += nw_begin(p, pwm, gen, (int)begin);
    float mid = 0;
    for (unsigned i = begin; i <= end; ++i) mid += get_val(p, pwm + 4 * i, (unsigned char)gen[i]);
    value += mid;
    value += nw_end(p, pwm, L, gen, (int)end);
    return value;
}

/* prepend "<n><op>" to a CIGAR being built back to front (bin_seq.cpp:590-596 and siblings) */
static void cigar_prepend(char* cigar, int n, char op) {
    char tmp[1024];
    strcpy(tmp, cigar);
    snprintf(cigar, 1024, "%d%c%s", n, op, tmp);
}

/* bin_seq::get_align_score_w_traceback bin_seq.cpp:445-718, max_flt(char&,...) :989-1011.
 * window length == L (GetString always returns L bases).  aligned may contain NUL (the
 * consense[i] quirk at :607,:660); its length is returned separately. */
int gmo_traceback(const gmo_params* p, const float* pwm, int L, const char* cons, const char* window,
                  char* aligned, int* aligned_len, char* cigar) {
    const int G = p->max_gap, W = 2 * G + 3, N = L;     /* N = gen.size() */
    float* nm = (float*)malloc(sizeof(float) * (size_t)(L + 1) * (size_t)W);
    char* mv = (char*)malloc((size_t)(L + 1) * (size_t)W);
#define IDX(i, j) ((size_t)(i) * W + ((j) - (i) + G + 1))
    for (int i = 0; i <= L; ++i)
        for (int j = i - G - 1; j <= i + G + 1; ++j) {
            if (j < 0 || j > N) continue;
            nm[IDX(i, j)] = NEG_INF; mv[IDX(i, j)] = 'D';
        }
    /* first column / first row up to G+2; index G+2 lies outside the band and is never read */
    for (int i = 0; i <= G + 1 && i <= L; ++i) { nm[IDX(i, 0)] = p->gap * (float)i; mv[IDX(i, 0)] = 'U'; }
    for (int j = 0; j <= G + 1 && j <= N; ++j) { nm[IDX(0, j)] = p->gap * (float)j; mv[IDX(0, j)] = 'L'; }
    mv[IDX(0, 0)] = 'D';
    for (int i = 1; i <= L; ++i)
        for (int j = i - G; j <= i + G; ++j) {
            if (j <= 0) continue;
            if (j > N) break;
            float d = nm[IDX(i - 1, j - 1)] + get_val(p, pwm + 4 * (i - 1), (unsigned char)window[j - 1]);
            float u = nm[IDX(i - 1, j)] + p->gap;
            float l = nm[IDX(i, j - 1)] + p->gap;
            char m; float best;
            if (d >= u) { if (d >= l) { m = 'D'; best = d; } else { m = 'L'; best = l; } }
            else        { if (u >= l) { m = 'U'; best = u; } else { m = 'L'; best = l; } }
            nm[IDX(i, j)] = best; mv[IDX(i, j)] = m;
        }
    int i = L, j = N, n = 0, ctype = 0 /*0 M,1 I,2 D*/, ccount = 0;
    static const char OPS[3] = { 'M', 'I', 'D' };
    cigar[0] = 0;
    while (i != 0 && j != 0) {
        char m = mv[IDX(i, j)];
        if (m == 'D') {
            aligned[n++] = cons[i - 1];
            if (ctype == 0) ccount++; else { if (ccount) cigar_prepend(cigar, ccount, OPS[ctype]); ctype = 0; ccount = 1; }
            i--; j--;
        } else if (m == 'U') {
            aligned[n++] = (i < L) ? cons[i] : '\0';        /* consense[i], sic (:607) */
            if (ctype == 1) ccount++; else { if (ccount) cigar_prepend(cigar, ccount, OPS[ctype]); ctype = 1; ccount = 1; }
            i--;
        } else {
            aligned[n++] = '-';
            if (ctype == 2) ccount++; else { if (ccount) cigar_prepend(cigar, ccount, OPS[ctype]); ctype = 2; ccount = 1; }
            j--;
        }
    }
    while (i > 0) {
        aligned[n++] = (i < L) ? cons[i] : '\0';
        if (ctype == 1) ccount++; else { cigar_prepend(cigar, ccount, OPS[ctype]); ctype = 1; ccount = 1; }
        i--;
    }
    while (j > 0) {
        aligned[n++] = '-';
        if (ctype == 2) ccount++; else { cigar_prepend(cigar, ccount, OPS[ctype]); ctype = 2; ccount = 1; }
        j--;
    }
    if (ccount > 0) cigar_prepend(cigar, ccount, OPS[ctype]);
    for (int a = 0, b = n - 1; a < b; ++a, --b) { char t = aligned[a]; aligned[a] = aligned[b]; aligned[b] = t; }
    aligned[n] = 0;
    *aligned_len = n;
#undef IDX
    free(nm); free(mv);
    return 0;
}

/* fix_CIGAR_for_deletions SequenceOperations.h:32-42 */
void gmo_fix_cigar(char* cigar) {
    int n = (int)strlen(cigar);
    if (n == 0 || cigar[n - 1] != 'D') return;
    int i;
    for (i = n - 2; i >= 0; --i) if (!isdigit((unsigned char)cigar[i])) break;
    cigar[i + 1] = 0;
}

/* reverse_CIGAR SequenceOperations.h:109-123 */
void gmo_reverse_cigar(const char* in, char* out) {
    char acc[1024]; acc[0] = 0;
    char num[32]; int nn = 0;
    for (size_t i = 0; i < strlen(in); ++i) {
        if (in[i] >= 48 && in[i] <= 58) { if (nn < 30) num[nn++] = in[i]; }
        else {
            char tmp[1024];
            num[nn] = 0;
            snprintf(tmp, sizeof tmp, "%s%c%s", num, in[i], acc);
            strcpy(acc, tmp);
            nn = 0;
        }
    }
    strcpy(out, acc);
}

/* ============================================================================================
 * Per read: vote map, candidate processing, unique map
 * ==========================================================================================*/
typedef struct { uint64_t key; int val; } loc_t;
typedef struct { loc_t* a; int n, cap; } locmap_t;     /* std::map<unsigned long,int>, kept sorted by key */

static int locmap_find(const locmap_t* m, uint64_t key, int* found) {
    int lo = 0, hi = m->n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (m->a[mid].key < key) lo = mid + 1; else hi = mid; }
    *found = (lo < m->n && m->a[lo].key == key);
    return lo;
}
static int* locmap_get(locmap_t* m, uint64_t key) {     /* operator[]: inserts 0 when absent */
    int found, at = locmap_find(m, key, &found);
    if (!found) {
        if (m->n == m->cap) { m->cap = m->cap ? m->cap * 2 : 64; m->a = (loc_t*)realloc(m->a, sizeof(loc_t) * (size_t)m->cap); }
        memmove(m->a + at + 1, m->a + at, sizeof(loc_t) * (size_t)(m->n - at));
        m->a[at].key = key; m->a[at].val = 0; m->n++;
    }
    return &m->a[at].val;
}

static gmo_hit* result_find(gmo_result* r, const char* key, int* at) {
    int lo = 0, hi = r->n_hits;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (strcmp(r->hits[mid].key, key) < 0) lo = mid + 1; else hi = mid; }
    *at = lo;
    return (lo < r->n_hits && strcmp(r->hits[lo].key, key) == 0) ? &r->hits[lo] : NULL;
}

/* set<pair<unsigned long,int>>::insert, ScoredSeq.h:238-252 */
static int hit_add_spot(gmo_hit* h, uint64_t pos, int strand) {
    int lo = 0, hi = h->n_pos;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        gmo_pos* q = &h->pos[mid];
        if (q->pos < pos || (q->pos == pos && q->strand < strand)) lo = mid + 1; else hi = mid;
    }
    if (lo < h->n_pos && h->pos[lo].pos == pos && h->pos[lo].strand == strand) return 0;
    if (h->n_pos == h->cap_pos) { h->cap_pos = h->cap_pos ? h->cap_pos * 2 : 2; h->pos = (gmo_pos*)realloc(h->pos, sizeof(gmo_pos) * (size_t)h->cap_pos); }
    memmove(h->pos + lo + 1, h->pos + lo, sizeof(gmo_pos) * (size_t)(h->n_pos - lo));
    h->pos[lo].pos = pos; h->pos[lo].strand = strand; h->n_pos++;
    return 1;
}

/* process_hits align_seq2_raw.cpp:22-178.  Returns 0 for the -u early exit. */
static int process_hits(const gmo_index* ix, const gmo_params* p, gmo_result* r, locmap_t* locs,
                        const float* pwm, int L, int strand) {
    char* w = (char*)malloc((size_t)L + 1);
    char* key = (char*)malloc((size_t)L + 1);
    int ok = 1;
    for (int t = 0; t < locs->n; ++t) {
        int cnt = locs->a[t].val;
        if (cnt < p->min_seed_hits) continue;
        if (cnt == -1) continue;
        uint64_t b = locs->a[t].key;
        if (!gmo_window(ix, b, (uint32_t)L, w)) continue;           /* contig boundary */
        double sc;
        if (p->nw) { sc = (double)gmo_nw_score(p, pwm, L, w); r->ctr.nw++; }
        else sc = (double)cnt;
        locs->a[t].val = -1;
        if (sc > r->top_score) r->top_score = sc;
        if (sc >= r->min_score) {
            if (strand == 1) gmo_revcomp_str(w, L, key); else strcpy(key, w);
            int at;
            gmo_hit* h = result_find(r, key, &at);
            if (!h) {
                if (r->n_hits == r->cap_hits) { r->cap_hits = r->cap_hits ? r->cap_hits * 2 : 4; r->hits = (gmo_hit*)realloc(r->hits, sizeof(gmo_hit) * (size_t)r->cap_hits); }
                memmove(r->hits + at + 1, r->hits + at, sizeof(gmo_hit) * (size_t)(r->n_hits - at));
                h = &r->hits[at];
                memset(h, 0, sizeof *h);
                h->key = strdup(key); h->seq = strdup(w); h->score = sc; h->first_strand = strand;
                hit_add_spot(h, b, strand);
                r->n_hits++;
                r->denominator += exp(sc);
            } else {
                if (p->unique_only) { ok = 0; break; }
                if (hit_add_spot(h, b, strand)) r->denominator += exp(sc);
            }
        }
    }
    free(w); free(key);
    return ok;
}

/* align_sequence align_seq2_raw.cpp:180-328 */
static int align_strand(const gmo_index* ix, const gmo_params* p, gmo_result* r,
                        const float* pwm, const char* cons, int L, int strand) {
    locmap_t locs = { 0, 0, 0 };
    unsigned last = (unsigned)(L - p->mer);
    int ok = 1;
    for (unsigned i = 0; i < last; i += (unsigned)p->jump) {
        uint64_t start = 0, end = 0;
        unsigned j;
        for (j = 0; j + i < last; j++) {
            gmo_sa_interval(ix, cons + i + j, p->mer, &start, &end, &r->ctr);
            if (end == 0 && start == 0) continue;
            else if (p->max_kmer_hits > 0 && end - start + 1 > p->max_kmer_hits) continue;
            else break;
        }
        i += j;
        if (end == 0 && start == 0) break;
        if (p->max_kmer_hits > 0 && end - start + 1 > p->max_kmer_hits) break;
        for (uint64_t v = start; v <= end; v++) {       /* unsigned int vit in the reference: fine below 2^32 ranks */
            uint64_t c = gmo_locate(ix, v, &r->ctr);
            uint64_t b = (c <= i) ? 0 : c - i;
            int* cnt = locmap_get(&locs, b);
            if (*cnt != -1) (*cnt)++;
        }
        if (!p->nw) continue;
        if (!process_hits(ix, p, r, &locs, pwm, L, strand)) { ok = 0; break; }
        if ((unsigned)r->n_hits > p->max_matches) { ok = 0; break; }
        if (p->fast) break;
    }
    if (ok && !p->nw) process_hits(ix, p, r, &locs, pwm, L, strand);   /* return value ignored, :317-325 */
    free(locs.a);
    return ok;
}

void gmo_result_free(gmo_result* r) {
    for (int i = 0; i < r->n_hits; ++i) { free(r->hits[i].key); free(r->hits[i].seq); free(r->hits[i].pos); }
    free(r->hits);
    r->hits = NULL; r->n_hits = r->cap_hits = 0;
}

/* set_top_matches Driver.cpp:432-612 */
void gmo_map_read(const gmo_index* ix, const gmo_params* p, const float* pwm, const char* cons, int L, gmo_result* r) {
    memset(r, 0, sizeof *r);
    if ((unsigned)L < (unsigned)p->mer) { r->status = GMO_TOO_SHORT; r->top_score = -2; return; }
    if (p->nw) {
        r->self_score = gmo_self_score(p, pwm, cons, L);
        double max_align = (double)r->self_score;
        if (max_align < p->cutoff) { r->status = GMO_TOO_POOR; r->top_score = -3; return; }
        r->min_score = p->align_is_fraction ? p->align_score * max_align : (double)p->align_score;
    } else {
        r->min_score = p->min_seed_hits;
    }
    if (p->pos_strand) {
        if (!align_strand(ix, p, r, pwm, cons, L, 0)) { gmo_result_free(r); r->status = GMO_TOO_MANY; r->denominator = 0; r->top_score = 999999; return; }
    }
    if (p->neg_strand) {
        float* rpwm = (float*)malloc(sizeof(float) * 4 * (size_t)L);
        char* rcons = (char*)malloc((size_t)L + 1);
        gmo_revcomp_pwm(pwm, L, rpwm);
        gmo_revcomp_str(cons, L, rcons);
        int ok = align_strand(ix, p, r, rpwm, rcons, L, 1);
        free(rpwm); free(rcons);
        if (!ok) { gmo_result_free(r); r->status = GMO_TOO_MANY; r->denominator = 0; r->top_score = 999999; return; }
    }
    if (r->n_hits == 0) { r->status = GMO_NONE; r->denominator = 0; r->top_score = 0; return; }
    r->status = GMO_OK;
}

/* ---- --snp: the pair HMM of SNPScoredSeq::score (src/SNPScoredSeq.cpp:25-109) ----------------------------------------------------------
 * bin_seq::pairHMM src/bin_seq.cpp:60-244 restated with the reference's types and operation order: the transition constants are FLOATs
 * (inc/bin_seq.h:49-69; products such as PHMM_q*PHMM_Tmg are float products), the three (n+1) x (m+1) matrices doubles, p_seq (:41-57) a
 * float sum of four float products times 3, the result rows floats that take `+= double` one read position at a time. */
static const float PH_q = 0.25f, PH_t = 0.05f, PH_d = 0.0025f, PH_e = 0.5f;
static float ph_score(char g, int x) {                         /* gPHMM_ALIGN_SCORES[genome char][read base], inc/a_matrices.c:92-120 */
    const float match = 0.98f, syn = 0.01f, nsyn = 0.005f;     /* inc/const_define.h:79-81 */
    int gi = g == 'a' || g == 'A' ? 0 : g == 'c' || g == 'C' ? 1 : g == 'g' || g == 'G' ? 2 : g == 't' || g == 'T' ? 3 : 4;
    if (gi == 4) return nsyn;
    if (gi == x) return match;
    return (gi ^ x) == 2 ? syn : nsyn;                          /* a <-> g, c <-> t are the transitions */
}
static float ph_pseq(const float* x, char y) {
    float sum = 0;
    sum += x[0] * ph_score(y, 0); sum += x[1] * ph_score(y, 1); sum += x[2] * ph_score(y, 2); sum += x[3] * ph_score(y, 3);
    return 3 * sum;
}
void gmo_pair_hmm(const float* pwm, int n, const char* cons, const char* genome, int m, float* out) {
    const float Tmm = 1 - 2 * PH_d - PH_t, Tgm = 1 - PH_d - PH_t, Tmg = PH_d, Tgg = PH_e;
    const int cs = (m + 1) * 3, pcs = m * 3;
    const size_t sz = (size_t)(n + 1) * (size_t)(m + 1) * 3;
    double* f = (double*)calloc(sz, sizeof(double)); double* b = (double*)calloc(sz, sizeof(double)); double* pp = (double*)calloc(sz, sizeof(double));
    for (int i = 0; i < m * 5; ++i) out[i] = 0;
    f[0] = 1;
    for (int i = 1; i < n + 1; ++i)
        for (int j = 1; j < m + 1; ++j) {
            f[i * cs + 3 * j] = ph_pseq(pwm + 4 * (i - 1), genome[j - 1]) * (Tmm * f[(i - 1) * cs + 3 * (j - 1)] + Tgm * f[(i - 1) * cs + 3 * (j - 1) + 1] + Tgm * f[(i - 1) * cs + 3 * (j - 1) + 2]);
            f[i * cs + 3 * j + 1] = PH_q * (Tmg * f[(i - 1) * cs + 3 * j] + Tgg * f[(i - 1) * cs + 3 * j + 1]);
            f[i * cs + 3 * j + 2] = PH_q * (Tmg * f[i * cs + 3 * (j - 1)] + Tgg * f[i * cs + 3 * (j - 1) + 2]);
        }
    const double fE = PH_t * (f[n * cs + 3 * m] + f[n * cs + 3 * m + 1] + f[n * cs + 3 * m + 2]);
    b[(n - 1) * cs + 3 * (m - 1)] = b[(n - 1) * cs + 3 * (m - 1) + 1] = b[(n - 1) * cs + 3 * (m - 1) + 2] = PH_t;
    for (int i = n - 1; i >= 0; --i)
        for (int j = m - 1; j >= 0; --j) {
            if (j == m - 1 && i == n - 1) continue;
            if (j == m - 1) {
                b[i * cs + 3 * j] = PH_q * Tmg * b[(i + 1) * cs + 3 * j + 1];
                b[i * cs + 3 * j + 1] = PH_q * Tgg * b[(i + 1) * cs + 3 * j + 1];
                b[i * cs + 3 * j + 2] = 0;
                continue;
            }
            if (i == n - 1) {
                b[i * cs + 3 * j] = PH_q * Tmg * b[i * cs + 3 * (j + 1) + 2];
                b[i * cs + 3 * j + 2] = PH_q * Tgg * b[i * cs + 3 * (j + 1) + 2];
                b[i * cs + 3 * j + 1] = 0;
                continue;
            }
            const float ps = ph_pseq(pwm + 4 * (i + 1), genome[j + 1]);
            b[i * cs + 3 * j] = ps * Tmm * b[(i + 1) * cs + 3 * (j + 1)] + PH_q * Tmg * b[(i + 1) * cs + 3 * j + 1] + PH_q * Tmg * b[i * cs + 3 * (j + 1) + 2];
            b[i * cs + 3 * j + 1] = ps * Tgm * b[(i + 1) * cs + 3 * (j + 1)] + PH_q * Tgg * b[(i + 1) * cs + 3 * j + 1];
            b[i * cs + 3 * j + 2] = ps * Tgm * b[(i + 1) * cs + 3 * (j + 1)] + PH_q * Tgg * b[i * cs + 3 * (j + 1) + 2];
        }
    for (int i = 1; i < n + 1; ++i)
        for (int j = 1; j < m + 1; ++j) {
            pp[(i - 1) * pcs + 3 * (j - 1)] = f[i * cs + 3 * j] * b[(i - 1) * cs + 3 * (j - 1)] / fE;
            pp[(i - 1) * pcs + 3 * (j - 1) + 1] = f[i * cs + 3 * j + 1] * b[(i - 1) * cs + 3 * (j - 1) + 1] / fE;
            pp[(i - 1) * pcs + 3 * (j - 1) + 2] = f[i * cs + 3 * j + 2] * b[(i - 1) * cs + 3 * (j - 1) + 2] / fE;
        }
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {                           /* g_gen_CONVERSION[consensus[j]] (no _INDEL build: Makefile:60): M + Y of the cell */
            const char c = cons[j];
            const int idx = c == 'a' || c == 'A' ? 0 : c == 'c' || c == 'C' ? 1 : c == 'g' || c == 'G' ? 2 : c == 't' || c == 'T' ? 3 : 4;
            out[i * 5 + idx] += pp[j * pcs + i * 3 + 2] + pp[j * pcs + i * 3];
        }
    free(f); free(b); free(pp);
}

/* ScoredSeq::max_char ScoredSeq.h:72-103 (argmax consensus, 'n' when all four equal) */
static char max_char(const float* c) {
    if (c[0] == c[1] && c[0] == c[2] && c[0] == c[3]) return 'n';
    if (c[0] >= c[1]) {
        if (c[0] >= c[2]) return c[0] >= c[3] ? 'a' : 't';
        return c[2] >= c[3] ? 'g' : 't';
    }
    if (c[1] >= c[2]) return c[1] >= c[3] ? 'c' : 't';
    return c[2] >= c[3] ? 'g' : 't';
}

/* ScoredSeq::get_SAM ScoredSeq.h:293-404: one record per (pos,strand) of `hit` */
static void emit_sam(const gmo_index* ix, const gmo_params* p, const gmo_result* r, const gmo_hit* hit,
                     const float* pwm, const float* rpwm, const char* cons, const char* rcons, int L, char* aligned,
                     gmo_sam** recs, int* n_recs, int* cap_recs, gmo_counters* ctr) {
    double total = exp(hit->score) / r->denominator;
    int mapq;
    if (total == 1) mapq = 30; else mapq = (int)round(-10 * log(1 - total) / log(10));
    if (mapq > 30) mapq = 30;
    char cig[1024]; int alen;
    if (p->nw) {
        if (hit->first_strand == 1) gmo_traceback(p, rpwm, L, rcons, hit->seq, aligned, &alen, cig);
        else gmo_traceback(p, pwm, L, cons, hit->seq, aligned, &alen, cig);
        if (ctr) ctr->tracebacks++;
        if (cig[0] == 0) strcpy(cig, "*"); else gmo_fix_cigar(cig);
    } else {
        snprintf(cig, sizeof cig, "%dM", L);                    /* to_string(consensus.size()) + "M" */
    }
    for (int q = 0; q < hit->n_pos; ++q) {
        if (*n_recs == *cap_recs) { *cap_recs = *cap_recs ? *cap_recs * 2 : 2; *recs = (gmo_sam*)realloc(*recs, sizeof(gmo_sam) * (size_t)*cap_recs); }
        gmo_sam* s = &(*recs)[(*n_recs)++];
        memset(s, 0, sizeof *s);
        s->pos = hit->pos[q].pos; s->strand = hit->pos[q].strand;
        s->contig = gmo_pos2rid(ix, (int64_t)s->pos);
        int base = (int)(s->pos - ix->contigs[s->contig].offset);         /* int chr_base_pos, GenomeBwt.cpp:632 */
        s->chr_pos = (uint64_t)(unsigned long)base + 1;
        s->mapq = mapq;
        strncpy(s->cigar, cig, sizeof s->cigar - 1);
        s->a_score = (float)hit->score; s->post_prob = (float)total; s->sim_matches = hit->n_pos;
    }
}

/* create_match_output Driver.cpp:614-753 with NormalScoredSeq::score NormalScoredSeq.cpp:24-76 */
int gmo_read_output(const gmo_index* ix, const gmo_params* p, const gmo_result* r, const float* pwm, const char* cons, int L,
                    gmo_sam** recs_out, gmo_deposit** deps_out, int* n_deps_out, gmo_counters* ctr) {
    *recs_out = NULL; *deps_out = NULL; *n_deps_out = 0;
    if (r->n_hits == 0) return 0;
    int n_recs = 0, cap_recs = 0, n_deps = 0, cap_deps = 0;
    gmo_sam* recs = NULL; gmo_deposit* deps = NULL;
    float* rpwm = (float*)malloc(sizeof(float) * 4 * (size_t)L);
    char* rcons = (char*)malloc((size_t)L + 1);     /* argmax consensus of the rc PWM (GetConsensus(rc), ScoredSeq.h:57-65) */
    char* fcons = (char*)malloc((size_t)L + 1);     /* argmax consensus of the forward PWM: score() uses it, not read.seq */
    char* aligned = (char*)malloc(2 * (size_t)L + 4);
    gmo_revcomp_pwm(pwm, L, rpwm);
    for (int i = 0; i < L; ++i) { rcons[i] = max_char(rpwm + 4 * i); fcons[i] = max_char(pwm + 4 * i); }
    rcons[L] = fcons[L] = 0;

    const gmo_hit* best = NULL;
    double best_log = exp(-1.0);                    /* empty NormalScoredSeq: align_score -1, ScoredSeq.h:117-120 */
    for (int h = 0; h < r->n_hits; ++h) {           /* std::map order = ascending key */
        const gmo_hit* hit = &r->hits[h];
        /* score(): span = aligned.size() of a traceback, weight = (float)(exp(score)/denom), every (pos,strand) */
        int alen = L; char cig[1024];
        float* hmm = NULL; float* hmm_rev = NULL;
        if (p->mode == GMO_MODE_SNP) {
            /* SNPScoredSeq::score SNPScoredSeq.cpp:25-109: no traceback - the pair HMM of (first strand's PWM, its argmax consensus) against the
             * window; span = the window's length; the other strand takes reverse_comp_cpy_phmm (SequenceOperations.h:164-181) */
            hmm = (float*)malloc(sizeof(float) * 5 * (size_t)L); hmm_rev = (float*)malloc(sizeof(float) * 5 * (size_t)L);
            if (hit->first_strand == 1) gmo_pair_hmm(rpwm, L, rcons, hit->seq, L, hmm); else gmo_pair_hmm(pwm, L, fcons, hit->seq, L, hmm);
            for (int i = 0; i < L; ++i) { float* d = hmm_rev + 5 * (L - 1 - i); const float* q5 = hmm + 5 * i; d[0] = q5[3]; d[1] = q5[2]; d[2] = q5[1]; d[3] = q5[0]; d[4] = q5[4]; }
        } else {
        if (hit->first_strand == 1) gmo_traceback(p, rpwm, L, rcons, hit->seq, aligned, &alen, cig);
        else gmo_traceback(p, pwm, L, fcons, hit->seq, aligned, &alen, cig);
        if (ctr) ctr->tracebacks++;
        }
        double total = exp(hit->score) / r->denominator;
        for (int q = 0; q < hit->n_pos; ++q) {
            if (n_deps == cap_deps) { cap_deps = cap_deps ? cap_deps * 2 : 4; deps = (gmo_deposit*)realloc(deps, sizeof(gmo_deposit) * (size_t)cap_deps); }
            deps[n_deps].pos = hit->pos[q].pos; deps[n_deps].span = (uint32_t)alen; deps[n_deps].w = (float)total; deps[n_deps].codes = NULL; deps[n_deps].hmm = NULL;
            if (p->mode == GMO_MODE_SNP) {
                deps[n_deps].hmm = (float*)malloc(sizeof(float) * 5 * (size_t)L);
                memcpy(deps[n_deps].hmm, hit->pos[q].strand == hit->first_strand ? hmm : hmm_rev, sizeof(float) * 5 * (size_t)L);
            } else if (p->mode != GMO_MODE_NORMAL) {
                /* BSScoredSeq::score BSScoredSeq.cpp:24-88: the gapped read string for positions on the first strand, its
                 * reverse_comp for the others; AddSeqScore(pos+i, w, g_gen_CONVERSION[char]) */
                uint8_t* codes = (uint8_t*)malloc((size_t)alen + 1);
                for (int t = 0; t < alen; ++t) {
                    char ch;
                    if (hit->pos[q].strand == hit->first_strand) ch = aligned[t];
                    else {
                        char c0 = aligned[alen - 1 - t];               /* reverse_comp SequenceOperations.h:56-96 */
                        switch (c0) { case 'a': ch = 't'; break; case 'A': ch = 'T'; break; case 't': ch = 'a'; break; case 'T': ch = 'A'; break;
                                      case 'c': ch = 'g'; break; case 'C': ch = 'G'; break; case 'g': ch = 'c'; break; case 'G': ch = 'C'; break;
                                      case '-': ch = '-'; break; default: ch = 'n'; break; }
                    }
                    uint8_t cv;                                        /* g_gen_CONVERSION Driver.cpp:912-925 */
                    switch (ch) { case 'a': case 'A': cv = 0; break; case 'c': case 'C': cv = 1; break; case 'g': case 'G': cv = 2; break;
                                  case 't': case 'T': cv = 3; break; case 'n': case 'N': cv = 4; break; case '\n': case '\r': case 11: case 12: cv = 5; break;
                                  case '\0': cv = 6; break; case '>': cv = 7; break; default: cv = 4; break; }
                    codes[t] = cv;
                }
                deps[n_deps].codes = codes;
            }
            n_deps++;
        }
        free(hmm); free(hmm_rev);
        if (p->print_all_sam) emit_sam(ix, p, r, hit, pwm, rpwm, cons, rcons, L, aligned, &recs, &n_recs, &cap_recs, ctr);
        if (exp(hit->score) > best_log) { best = hit; best_log = exp(hit->score); }     /* is_greater: strict, first wins */
    }
    if (!p->print_all_sam && best && best->score > r->top_score - SAME_DIFF)             /* Driver.cpp:695 */
        emit_sam(ix, p, r, best, pwm, rpwm, cons, rcons, L, aligned, &recs, &n_recs, &cap_recs, ctr);
    free(rpwm); free(rcons); free(fcons); free(aligned);
    *recs_out = recs; *deps_out = deps; *n_deps_out = n_deps;
    return n_recs;
}

/* single_write_cond_wait Driver.cpp:2146-2217 (valid-match branch) */
int gmo_format_sam(const gmo_index* ix, const gmo_params* p, const gmo_sam* s, const char* name, const char* cons, const char* qual,
                   char* out, size_t cap) {
    size_t Lc = strlen(cons), Lq = strlen(qual);
    char* seq = (char*)malloc(Lc + 1); char* q = (char*)malloc(Lq + 1); char cig[1024];
    if (s->strand == 0) { strcpy(seq, cons); strcpy(q, qual); strcpy(cig, s->cigar); }
    else {
        gmo_revcomp_str(cons, (int)Lc, seq);
        for (size_t i = 0; i < Lq; ++i) q[i] = qual[Lq - 1 - i];
        q[Lq] = 0;
        gmo_reverse_cigar(s->cigar, cig);
    }
    char nm[MAX_NAME_SZ]; strncpy(nm, name, MAX_NAME_SZ - 1); nm[MAX_NAME_SZ - 1] = 0;
    int n = snprintf(out, cap, "%s\t%d\t%s\t%lu\t%d\t%s\t*\t0\t0\t%s\t%s\tXA:f:%g\tXP:f:%g\tX0:i:%d\n",
                     nm, s->strand == 0 ? 0 : 16, ix->contigs[s->contig].name, (unsigned long)s->chr_pos, s->mapq, cig, seq, q,
                     (double)(float)s->a_score * (1.0 / p->adjust), (double)(float)s->post_prob, s->sim_matches);
    free(seq); free(q);
    return n;
}

/* ============================================================================================
 * Whole run (parallel_thread_run Driver.cpp:2303-2407; PrintFinalSGR GenomeBwt.cpp:1212-1273)
 * ==========================================================================================*/
typedef struct { char* name; char* seq; char* qual; int L; int ill; } fq_rec;
typedef struct { char* sam; size_t sam_len; gmo_deposit* deps; int n_deps; int matched; gmo_counters ctr; } read_out;

typedef struct {
    const gmo_index* ix; const gmo_params* p; fq_rec* recs; read_out* outs; uint64_t n;
    volatile uint64_t next; pthread_mutex_t mu; int illumina;
} work_t;

static void ctr_add(gmo_counters* a, const gmo_counters* b) {
    a->kmers += b->kmers; a->occ_calls += b->occ_calls; a->occ_blocks += b->occ_blocks; a->locates += b->locates;
    a->lf_steps += b->lf_steps; a->nw += b->nw; a->tracebacks += b->tracebacks;
}

static void map_one(work_t* w, uint64_t i) {
    fq_rec* fr = &w->recs[i]; read_out* ro = &w->outs[i];
    memset(ro, 0, sizeof *ro);
    int L = fr->L;
    float* pwm = (float*)malloc(sizeof(float) * 4 * (size_t)(L > 0 ? L : 1));
    int ill = fr->ill;
    if (gmo_pwm_from_fastq(fr->seq, fr->qual, L, &ill, pwm) != 0) { free(pwm); return; }
    gmo_result r;
    gmo_map_read(w->ix, w->p, pwm, fr->seq, L, &r);
    ro->ctr = r.ctr;
    if (r.status == GMO_OK || r.status == GMO_TOO_MANY) ro->matched = 1;
    if (r.status == GMO_OK) {
        gmo_sam* recs; int n = gmo_read_output(w->ix, w->p, &r, pwm, fr->seq, L, &recs, &ro->deps, &ro->n_deps, &ro->ctr);
        size_t cap = (size_t)n * (2 * (size_t)L + strlen(fr->qual) + 2400) + 16, len = 0;
        ro->sam = (char*)malloc(cap);
        ro->sam[0] = 0;
        for (int k = 0; k < n; ++k) len += (size_t)gmo_format_sam(w->ix, w->p, &recs[k], fr->name, fr->seq, fr->qual, ro->sam + len, cap - len);
        ro->sam_len = len;
        free(recs);
    }
    gmo_result_free(&r);
    free(pwm);
}

static void* worker(void* arg) {
    work_t* w = (work_t*)arg;
    for (;;) {
        pthread_mutex_lock(&w->mu);
        uint64_t b = w->next; w->next += 64;
        pthread_mutex_unlock(&w->mu);
        if (b >= w->n) break;
        uint64_t e = b + 64 < w->n ? b + 64 : w->n;
        for (uint64_t i = b; i < e; ++i) map_one(w, i);
    }
    return NULL;
}

/* std::getline on an ifstream, as SeqReader::get_more_fastq uses it (SeqReader.cpp:1073-1141): a stream that is no longer good
 * fails WITHOUT touching the string; at the end of input the string is emptied and eofbit + failbit are set; a last line without a
 * newline is delivered and sets eofbit.  in.eof() is eofbit. */
typedef struct { FILE* f; int eofbit, failbit; } fq_in;
static void fq_getline(fq_in* in, char** s) {
    if (in->eofbit || in->failbit) { in->failbit = 1; return; }
    char* line = NULL; size_t cap = 0;
    ssize_t n = getline(&line, &cap, in->f);
    free(*s);
    if (n < 0) { free(line); *s = strdup(""); in->eofbit = 1; in->failbit = 1; return; }
    if (n > 0 && line[n - 1] == '\n') line[n - 1] = 0; else in->eofbit = 1;
    *s = line;
}

int gmo_run(const gmo_index* ix, const gmo_params* p, const char* fastq, const char* out_prefix, int threads,
            uint64_t max_reads, const char* cmdline, gmo_run_stats* st) {
    memset(st, 0, sizeof *st);
    FILE* f = fopen(fastq, "r");
    if (!f) return -1;
    fq_rec* recs = NULL; uint64_t n = 0, cap = 0;
    fq_in in = { f, 0, 0 };
    char* name = strdup(""); char* seq = strdup(""); char* plus = strdup(""); char* qual = strdup("");
    for (;;) {                                        /* get_more_fastq SeqReader.cpp:1060-1150, incl. its recovery from malformed records */
        if (max_reads && n >= max_reads) break;
        if (in.eofbit) break;                         /* :1061 */
        fq_getline(&in, &name);
        while (name[0] == 0 && !in.eofbit) fq_getline(&in, &name);      /* blank lines before a name :1076-1079 */
        if (in.eofbit) break;                         /* :1082 */
        fq_getline(&in, &seq); fq_getline(&in, &plus); fq_getline(&in, &qual);
        int ended = 0;
        while (name[0] != '@' || plus[0] != '+' || strlen(seq) > strlen(qual)) {       /* :1095 */
            if (name[0] != '@' || plus[0] != '+') {
                while ((name[0] != '@' || plus[0] != '+') && !in.eofbit) {             /* shift the four lines by one :1101-1115 */
                    char* t = name; name = seq; seq = plus; plus = qual; qual = t;     /* the old name is dropped: the stream is good here, getline replaces it */
                    fq_getline(&in, &qual);
                    if (in.eofbit) { ended = 1; break; }
                }
                if (ended || in.eofbit) { ended = 1; break; }
            }
            if (strlen(seq) > strlen(qual)) {         /* a quality line shorter than its sequence: take the next four lines :1128-1142 */
                fq_getline(&in, &name); fq_getline(&in, &seq); fq_getline(&in, &plus); fq_getline(&in, &qual);
                if (in.eofbit) { ended = 1; break; }
            }
        }
        if (ended) break;
        if (n == cap) { cap = cap ? cap * 2 : 1024; recs = (fq_rec*)realloc(recs, sizeof(fq_rec) * cap); }
        recs[n].name = strdup(name + 1); recs[n].seq = strdup(seq); recs[n].qual = strdup(qual); recs[n].L = (int)strlen(seq);
        n++;
    }
    free(name); free(seq); free(plus); free(qual);
    fclose(f);
    /* --illumina falls back to Phred+33 at the first negative quality and STAYS there for every later read of the run
     * (gILLUMINA is a global cleared at SeqReader.cpp:1174; reads are parsed in file order under read_lock) */
    {
        int ill = p->illumina;
        for (uint64_t i = 0; i < n; ++i) {
            if (ill) for (const char* q = recs[i].qual; *q && q < recs[i].qual + recs[i].L; ++q) if ((int)*q - 64 < 0) { ill = 0; break; }
            recs[i].ill = ill;
        }
    }
    work_t w; memset(&w, 0, sizeof w);
    w.ix = ix; w.p = p; w.recs = recs; w.n = n; w.illumina = p->illumina;
    w.outs = (read_out*)calloc(n ? n : 1, sizeof(read_out));
    pthread_mutex_init(&w.mu, NULL);
    if (threads < 1) threads = 1;
    double t0 = now_s();
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, worker, &w);
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    st->map_seconds = now_s() - t0;
    free(th);

    char fn[4096];
    FILE* sam = NULL;
    if (out_prefix) {
        snprintf(fn, sizeof fn, "%s.sam", out_prefix);
        sam = fopen(fn, "w");
        if (!sam) return -2;
        for (int i = 0; i < ix->n_seqs; ++i) {            /* Driver.cpp:2322-2327 */
            uint64_t next = (i + 1 < ix->n_seqs) ? ix->contigs[i + 1].offset : ix->l_pac;
            fprintf(sam, "@SQ\tSN:%s\tLN:%lu\n", ix->contigs[i].name, (unsigned long)(next - ix->contigs[i].offset));
        }
        fprintf(sam, "@PG\tID:gnumap\tPN:gnumap\tVN:4.0.0 BETA\tCL:%s\n", cmdline ? cmdline : "");
    }
    size_t nbins = ix->l_pac / (uint64_t)p->bin_size + 64;
    float* cov = (float*)calloc(nbins, sizeof(float));
    float* nuc[5] = { 0, 0, 0, 0, 0 };                       /* reads[A,C,G,T,N][loc], make_extra_arrays GenomeBwt.cpp:198-280 */
    if (p->mode != GMO_MODE_NORMAL) for (int c = 0; c < 5; ++c) nuc[c] = (float*)calloc(nbins, sizeof(float));
    for (uint64_t i = 0; i < n; ++i) {
        read_out* ro = &w.outs[i];
        st->n_matched += (uint64_t)ro->matched;
        ctr_add(&st->ctr, &ro->ctr);
        if (ro->sam) {
            if (sam) fwrite(ro->sam, 1, ro->sam_len, sam);
            for (size_t k = 0; k < ro->sam_len; ++k) st->n_records += ro->sam[k] == '\n';
        }
        for (int d = 0; d < ro->n_deps; ++d)              /* AddScore GenomeBwt.cpp:483-490 */
            for (uint32_t t = 0; t < ro->deps[d].span; ++t) {
                uint64_t bin = (ro->deps[d].pos + t) / (uint64_t)p->bin_size;
                if (bin < nbins) {
                    cov[bin] += ro->deps[d].w;
                    /* AddSeqScore(pos, amt, which) GenomeBwt.cpp:556-603; which >= 5 ('\0' of the consense[i] quirk) indexes past
                     * reads[] in the reference (undefined behaviour): not deposited here */
                    if (ro->deps[d].codes && ro->deps[d].codes[t] < 5) nuc[ro->deps[d].codes[t]][bin] += ro->deps[d].w;
                    /* AddSeqScore(pos, amt[], scale) GenomeBwt.cpp:496-551 (plain build): reads[c][loc] += amt[c] * scale, float product, float add */
                    if (ro->deps[d].hmm) for (int c = 0; c < 5; ++c) nuc[c][bin] += ro->deps[d].hmm[5 * t + c] * ro->deps[d].w;
                }
            }
        for (int d = 0; d < ro->n_deps; ++d) { free(ro->deps[d].codes); free(ro->deps[d].hmm); }
        free(ro->sam); free(ro->deps);
    }
    st->n_reads = n;
    if (sam) fclose(sam);
    if (out_prefix && p->mode == GMO_MODE_SNP) {
        /* PrintFinalSNP GenomeBwt.cpp:930-1090 as far as it can be pinned here: chr, position, total, the five per-nucleotide sums of every
         * position above MIN_PRINT.  The reference then appends PrintSNPCall's likelihood-ratio columns, which need gsl_cdf_chisq_P - the one
         * symbol the reference build of oracle/Makefile leaves unresolved: not restated, the line ends after the eighth column. */
        snprintf(fn, sizeof fn, "%s.gmp", out_prefix);
        FILE* gm = fopen(fn, "w");
        if (!gm) return -3;
        uint64_t count = 0;
        for (int i = 0; i < ix->n_seqs; ++i) {
            uint64_t next = (i + 1 < ix->n_seqs) ? ix->contigs[i + 1].offset : ix->l_pac;
            for (; count < next; count += (uint64_t)p->bin_size) {
                uint64_t locus = count / (uint64_t)p->bin_size;
                if (cov[locus] > MIN_PRINT) {
                    fprintf(gm, "%s\t%ld\t%.5f", ix->contigs[i].name, (long)(count - ix->contigs[i].offset) + 1, cov[locus]);
                    for (int c = 0; c < 5; ++c) fprintf(gm, "\t%.5f", nuc[c][locus]);
                    fprintf(gm, "\n");
                }
            }
        }
        fclose(gm);
    } else if (out_prefix && p->mode != GMO_MODE_NORMAL) {       /* PrintFinalBisulfite GenomeBwt.cpp:1092-1210 (PrintFinal :915-926: .gmp INSTEAD of .sgr) */
        snprintf(fn, sizeof fn, "%s.gmp", out_prefix);
        FILE* gm = fopen(fn, "w");
        if (!gm) return -3;
        char want = p->mode == GMO_MODE_BS ? 'c' : p->mode == GMO_MODE_BS2 ? 'g' : p->mode == GMO_MODE_ATOG ? 'a' : 't';
        uint64_t count = 0;
        for (int i = 0; i < ix->n_seqs; ++i) {
            uint64_t next = (i + 1 < ix->n_seqs) ? ix->contigs[i + 1].offset : ix->l_pac;
            for (; count < next; count += (uint64_t)p->bin_size) {
                char at = "acgt"[ix->pac[count >> 2] >> ((~count & 3) << 1) & 3];
                if (at != want) continue;
                uint64_t locus = count / (uint64_t)p->bin_size;
                if (cov[locus] > 0.0f) {
                    fprintf(gm, "%s\t%ld\t%f", ix->contigs[i].name, (long)(count - ix->contigs[i].offset) + 1, cov[locus]);
                    for (int c = 0; c < 5; ++c) fprintf(gm, "\t%.5f", nuc[c][locus]);
                    fprintf(gm, "\n");
                }
            }
        }
        fclose(gm);
    } else if (out_prefix) {                              /* PrintFinalSGR GenomeBwt.cpp:1212-1273 */
        snprintf(fn, sizeof fn, "%s.sgr", out_prefix);
        FILE* sg = fopen(fn, "w");
        if (!sg) return -3;
        uint64_t count = 0;
        for (int i = 0; i < ix->n_seqs; ++i) {
            uint64_t next = (i + 1 < ix->n_seqs) ? ix->contigs[i + 1].offset : ix->l_pac;
            for (; count < next; count += (uint64_t)p->bin_size)
                if (cov[count / (uint64_t)p->bin_size] > MIN_PRINT)
                    fprintf(sg, "%s\t%ld\t%.5f\n", ix->contigs[i].name, (long)(count - ix->contigs[i].offset) + 1, cov[count / (uint64_t)p->bin_size]);
        }
        fclose(sg);
    }
    free(cov);
    for (int c = 0; c < 5; ++c) free(nuc[c]);
    for (uint64_t i = 0; i < n; ++i) { free(recs[i].name); free(recs[i].seq); free(recs[i].qual); }
    free(recs); free(w.outs);
    pthread_mutex_destroy(&w.mu);
    return 0;
}
