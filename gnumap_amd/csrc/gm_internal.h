// gm_internal.h — structures shared by the HIP kernels (gm_kernels.hip) and the host side of libgnumap_hip.
// Nothing here is part of the public ABI (include/gnumap_hip.h).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

// ---- HBM-resident index, passed to kernels by value -------------------------------------------------
// bwt      : the occ-interleaved BWT exactly as <fa>.gnumap.bwt stores it: one 64-byte block per 128 bases
//            = 4 x u64 cumulative counts (A,C,G,T before the block) + 8 x u32 of 16 bases, MSB first
//            (reference: bwt_bwtupdate_core src/bwtindex.c:128-150, bwt_occ_intv inc/bwt.h:73)
// sa_samples: rank-sampled suffix array (every 32nd rank), narrowed to u32; [0] = 0xFFFFFFFF (= (u64)-1 mod 2^32)
// full_sa  : optional full suffix array SA[0..seq_len], u32, expanded on device from the samples
// pac      : 2 bit/base, 4 per byte MSB first (reference: _get_pac src/bntseq.c:225)
// occ_planes: the MI355X-native rank structure derived from bwt once per index: for each base c a bit plane over the
//            $-removed BWT, cut into 16-byte granules {u32 count of c before the granule, 96 membership bits}, so that
//            one rank query (bwt_occ) is ONE 16-byte load + three popcounts.  planes[c * occ_nblk + x / 96].
struct GmDevIndex {
    const uint32_t* bwt;
    const uint4* occ_planes;
    uint32_t occ_nblk;
    const uint32_t* sa_samples;
    const uint32_t* full_sa;
    const uint8_t* pac;
    const uint32_t* contig_off;     // n_seqs + 1 offsets, last = l_pac
    uint32_t seq_len, primary, l_pac, n_seqs;
    uint32_t sa_mask, sa_shift;     // sa_intv - 1, log2(sa_intv)
    uint32_t L2[5];
};

struct GmDevParams {
    int mer, jump, kmin, nw, fast, pos_strand, neg_strand, align_is_fraction;
    int max_gap;                    // -M: band half-width of the DP (3 = the register-band kernels, anything else = gm_band.hip)
    int fused;                      // 1 = the one-wave vote kernels look their seeds up themselves (no k_seed launch, no seed rows in HBM)
    uint32_t heavy_min;             // read x strands with more SA hits than this go to the sorted-key path (gm_heavy.hip)
    uint32_t hcap;
    int dbg;                        // GM_DBG bits for kernel timing experiments (0 in production)
    float gap, align_score, cutoff;
    const float* S256;              // gALIGN_SCORES, 256 x 4, in HBM: the self score indexes it with the FASTQ characters,
                                    // the DP with rows 'a','c','g','t' (genome windows are lowercase acgt)
    const uint2* kmer_tab;          // SA interval of every kmer_T-mer (suffix of the seed k-mer), or null: {k,l}; empty = {0xFFFFFFFF, depth}
    int kmer_T;
    const uint4* kmer_ctab;         // compact form of kmer_tab, 16 B per 8 consecutive codes: {first SA rank, 8 x u8 hit counts (2 words), escape flag}; null = not built
    const float2* lut;              // [0..255] Phred+33, [256..511] Phred+64: (p, (1-p)/3) as fp32; p = NaN when negative
    const uint4* bucket;            // direct-addressed k-mer -> positions table (gm_bucket.hip): 128 bytes per mer-mer code, or null
    uint32_t bucket_ecap, bucket_ovcap;   // k_vote_bucket votes itself on a strand with at most this many SA hits / seeds beyond 28 hits; more -> list kernel
    int bucket_T, bucket_ctx;       // the table's k-mer length; 1 = context records: seeds of bucket_T + 1 .. bucket_T + 5 characters (k_build_bucket_ctx)
};

struct GmSeed { uint32_t k, l, pos; };

#define GM_FIXED_C 4
// own candidate slots of read x strand rs, in blocks of 64 read x strands: the 64 slots 0 side by side (what k_cand_gather reads of
// EVERY read x strand: one 1 KB stretch per wavefront instead of 16 bytes out of every 64), slots 1 .. 3 behind them
#define GM_FIXED_AT(b, rs, idx) (((size_t)(rs) >> 6) * (64 * GM_FIXED_C) + ((idx) == 0 ? ((size_t)(rs) & 63) : 64 + ((size_t)(rs) & 63) * (GM_FIXED_C - 1) + (size_t)(idx) - 1))
#define GM_GROUP_BIG 64          // accepted hits per read above which grouping takes the hash-set + sort path
#define GM_NSHARD 1024
#define GM_SHARD_STRIDE 32

// flags of GmCand
enum { GMC_VALID = 1, GMC_ACCEPT = 2 };
struct GmCand {
    uint32_t rs;                    // read * 2 + strand
    uint32_t b;                     // candidate window start on the concatenated reference
    uint16_t step;                  // seed index at which the vote count reached kmin (NW order of the reference)
    uint8_t flags, pad;
    float score;                    // NW score, or the vote count with --no_nw
};

struct GmRawHit { uint32_t read; uint32_t pos; float score; uint16_t step; uint8_t strand; uint8_t pad; };

// device-side counters, one u64 each (see gm_counters in the public header)
enum {
    GMK_KMERS = 0, GMK_OCC, GMK_SEEDS, GMK_SA_HITS, GMK_LF_STEPS, GMK_CANDS, GMK_NW_CELLS, GMK_ACCEPTED,
    GMK_OVERFLOW_RS, GMK_BAD_QUAL, GMK_HEAVY_SLOTS, GMK_OCC_BLOCKS, GMK_TAB_LOOKUPS,
    GMK_HIGH_QUAL,                  // k_prep: reads with a quality character above 127 (k_nw_rows' value table stops there)
    GMK_DBG0, GMK_DBG1, GMK_DBG2, GMK_DBG3, GMK_DBG4, GMK_DBG5, GMK_DBG6, GMK_DBG7, GMK_DBG8,      // GM_DBG & 64: sampled phase clocks of the vote kernel
    GMK_N
};

struct GmDevBatch {
    uint32_t n, stride, max_seeds, illumina_until;
    uint32_t read_base;             // index of read 0 of this (sub-)batch in the caller's batch (raw hits carry absolute indices)
    const uint8_t* bases;
    const uint8_t* quals;
    const uint16_t* len;
    int8_t* status;                 // n
    float* self_score;              // n
    double* min_score;              // n
    float* top_score;               // n
    GmSeed* seeds;                  // 2n x max_seeds
    uint16_t* n_seeds;              // 2n
    uint32_t* n_entries;            // 2n   number of SA hits of the seeds used
    uint64_t* entry_off;            // 2n+1 exclusive scan of n_entries (sampled-SA mode only)
    uint32_t* coords;               // located coordinates, entry_off order (sampled-SA mode only)
    uint8_t* rs_overflow;           // 2n   1 = LDS vote table overflowed, handled by the global-table kernel
    uint32_t* retry_list;           // list of overflowed rs
    uint64_t* retry_off;            // table offsets for the retry list
    uint32_t* gtab_keys;            // global vote tables (retry path)
    uint32_t* gtab_vals;
    // candidates are appended through GM_NSHARD bump counters (one 128-byte line each) instead of one global counter:
    // shard s owns cands[s * cand_region, (s+1) * cand_region)
    GmCand* cands;  uint32_t cand_cap, cand_region;
    uint32_t* shard_cnt;            // GM_NSHARD counters, GM_SHARD_STRIDE words apart
    // the one-wave vote kernels (k_vote_tiny*) do not wait for a bump counter: a read x strand with at most GM_FIXED_C candidates
    // stores them into its own slots (plain stores, the wave retires) and k_cand_gather moves them into the shards afterwards with
    // ONE atomic per 64 read x strands; null = not in use
    GmCand* fixed_cands;            // 2n x GM_FIXED_C
    uint8_t* fixed_cnt;             // 2n, zeroed before the vote kernel
    uint32_t fixed_epoch;           // != 0: no fixed_cnt - slot 0 of a read x strand says how many of its slots are in use (pad) and for which launch (score bits = this value)
    uint32_t* hit_count;            // n
    uint64_t* hit_begin;            // n+1
    uint32_t* hit_cursor;           // n
    GmRawHit* raw_hits;  uint64_t raw_cap;
    unsigned long long* counters;   // GMK_N
    uint32_t* n_retry;              // device counter
    uint32_t* n_big;                // device counter: read x strands handed from k_vote_sparse to k_vote_fast_list
    uint32_t* big_list;             // 2n
    // 2-bit forms of the reads for the fused seed lookup (written by k_prep when non-null), pack_words words per read:
    //   [0]                  length | (some base is not ACGT) << 16 | (status != 0) << 17
    //   F  [1, w2+2)         base p of the read at bit 2 (16 w2 - 1 - p): 2 mer bits from bit 2 (16 w2 - i - mer) are the table code of
    //                        the k-mer [i, i+mer) (last character lowest)
    //   F' [w2+2, 2 w2+3)    the same for the reverse complement of the read (its base p' at bit 2 (16 w2 - 1 - p'))
    // so that a k-mer's words sit at an address that does not depend on the read's length: one round trip for both.
    uint32_t* pack; uint32_t pack_w2, pack_words;
    unsigned long long* band_moves; // -M other than 3: move words of k_traceback_band, [row][item]; band_moves_words of them
    uint64_t band_moves_words;
};

// device mirrors of the public records (include/gnumap_hip.h: gm_match, gm_pos, gm_sam_rec; layouts asserted in gm_api.cpp), so that
// the kernels of gm_output.hip write them in HBM in their final form
struct GmDevMatch { uint32_t read; float score; unsigned long long first_pos; uint8_t first_strand; uint8_t pad[3]; uint32_t pos_begin, pos_end; uint32_t tail; };
struct GmDevPos { unsigned long long pos; uint8_t strand; uint8_t pad[7]; };
struct GmDevSamRec { uint32_t read, pad0; unsigned long long pos; uint32_t contig, pad1; unsigned long long chr_pos; uint8_t strand; uint8_t pad2[3];
                     int32_t mapq; float a_score, post_prob; int32_t sim_matches; uint32_t cigar_off; };

// workspace of the grouping kernels (process_hits' unique map on the device); per-hit arrays share the CSR of hit_begin
struct GmDevGroup {
    GmRawHit* sorted;               // accepted hits of a read in the reference's processing order
    float* ord_score;               // their scores in that order (-inf: dropped by -u with --no_nw): the host sums exp() over these
    uint32_t* lead;                 // index (within the read) of the first hit with the same key; 0xFFFFFFFF = dropped
    uint32_t* krank;                // leaders: rank of the key among the read's keys (std::map<string> order)
    unsigned long long* khash;
    uint32_t* n_match;              // n: matches of a read (0 unless its status stays OK)
    uint64_t* match_begin;          // n+1: exclusive scan of n_match
    uint32_t* multi_list;           // reads with 2 .. GM_GROUP_BIG accepted hits (and the ones the big path hands back)
    uint32_t* n_multi;
    uint32_t* big_list;             // reads with more accepted hits: hash-set + sort path (k_group_big), linear in the hits
    uint32_t* n_big;
    uint8_t* big_done;              // per big_list entry: 1 = grouped by the big path (k_group_write_big finishes it)
    unsigned long long* sk0; unsigned long long* sk1; uint32_t* si0; uint32_t* si1;     // sort scratch, per hit
    GmDevMatch* matches;
    uint32_t* match_hit;            // per match: index (in the hit CSR, processing order) of the hit that gave the match its score
    GmDevPos* positions;            // same CSR as the hits (a read's positions live in [hit_begin[r], hit_begin[r+1]))
};

// Launchers (gm_kernels.hip, gm_output.hip).  All asynchronous on `stream`; they return a hipError_t cast to int.
#ifdef __cplusplus
extern "C++" {
// run-time switches (GM_*): the value set with gm_set_option(), else the environment variable, else null.  Read on EVERY call:
// a long-lived host (and one test process) can change a choice between two batches.  Defined in gm_api.cpp.
const char* gm_opt(const char* name);
inline long long gm_opt_ll(const char* name, long long dflt) { const char* e = gm_opt(name); return e && *e ? atoll(e) : dflt; }
inline bool gm_opt_is(const char* name, const char* value) { const char* e = gm_opt(name); return e && !strcmp(e, value); }
int gmk_expand_full_sa(const GmDevIndex& ix, uint32_t* full_sa, void* stream);
int gmk_build_occ_planes(const GmDevIndex& ix, uint4* planes, uint32_t nblk, void* stream);
int gmk_build_kmer_table(const GmDevIndex& ix, uint2* tab, int T, void* stream);
int gmk_build_kmer_compact(const uint2* tab, uint4* ctab, int T, void* stream);
int gmk_extend_kmer_table(const GmDevIndex& ix, const uint2* prev, uint2* next, int T, void* stream);
int gmk_prep(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, void* stream);
int gmk_prep_rows(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, void* stream);      // gm_prep.hip (stride <= 152)
int gmk_seed(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, void* stream);
int gmk_scan_entries(const GmDevBatch& b, void* stream);
int gmk_locate_sampled(const GmDevIndex& ix, const GmDevBatch& b, unsigned long long n_entries /* SA hits of the block = entries of coords[] */, void* stream);
int gmk_vote(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, int use_full_sa, int dense, int slots_hint, void* stream);
int gmk_cand_gather(const GmDevBatch& b, void* stream);
int gmk_shard_stats(const GmDevBatch& b, uint32_t* out /* {total, max} in device memory */, void* stream);
// gm_bucket.hip: the bucket table and the one-wave-per-read vote kernel that looks its seeds up in it
int gmk_build_bucket(const uint2* tab, const uint32_t* full_sa, const uint8_t* pac, uint4* bucket, int T, int ctx, void* stream);
// rlist / n_rlist (device memory) != null: only the reads of that list
int gmk_vote_bucket(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, uint32_t max_reg, const uint32_t* rlist, const uint32_t* n_rlist, void* stream);
// gm_pair.hip: two reads per wavefront for the common case; the reads it flags (fallback[r] = 1) are listed for gmk_vote_bucket
int gmk_vote_pair(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, uint32_t max_reg, uint8_t* fallback, uint32_t* list, uint32_t* n_list, void* stream);
int gmk_vote_list(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, int use_full_sa, void* stream);     // k_vote_fast_list over b.big_list
// gm_heavy.hip: read x strands with more than heavy_min SA hits (sorted-key vote path)
int gmk_heavy_collect(const GmDevBatch& b, uint32_t heavy_min, uint32_t* n_heavy, uint32_t* heavy_list /* {rs, n_seeds, SA hits} triples */, int sum_counters, void* stream);
inline uint32_t gm_pack_w2(uint32_t stride) { return (stride + 15u) / 16u; }
inline uint32_t gm_pack_words(uint32_t stride) { const uint32_t w2 = gm_pack_w2(stride); return (1u + 2u * (w2 + 1u) + 3u) & ~3u; }
size_t gmk_heavy_sort_temp_bytes(size_t n_keys);
int gmk_heavy_chunk(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, int use_full_sa, const uint32_t* heavy_list, uint32_t j0, uint32_t nj,
                    const unsigned long long* key_off, unsigned long long* keys0, unsigned long long* keys1, unsigned long long n_keys, void* tmp,
                    size_t tmp_bytes, unsigned item_bits, void* stream);
int gmk_vote_retry(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, int use_full_sa, uint32_t j0, uint32_t n_retry, void* stream);
// rows_len != 0: every read of the block has this length and no quality character is above 127 -> k_nw_rows (gm_nw.hip) may take it
int gmk_nw(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, uint32_t n_cands, uint32_t rows_len, void* stream);
const char* gmk_nw_form(const GmDevParams& p, const GmDevBatch& b, uint32_t n_cands, uint32_t rows_len);      // which DP kernel gmk_nw launches for these arguments
int gmk_nw_rows(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, uint32_t n_cands, uint32_t L, void* stream);
// gm_band.hip: the DP kernels for a band half-width other than 3 (-M)
int gmk_nw_band(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, uint32_t n_cands, int G, void* stream);
int gmk_traceback_band(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, const GmCand* items, uint32_t n, unsigned long long* ops,
                       uint32_t ops_words, uint16_t* ops_len, const uint8_t* emit, uint32_t* cig_cnt, uint32_t* max_span, int G, void* stream);
inline uint64_t gm_band_moves_words(uint32_t n, uint32_t stride) {          // at most 512 MB, at least one workgroup's worth of items
    const uint64_t rows = (uint64_t)stride + 1, cap = (512ull << 20) / 8, all = (uint64_t)n * rows;
    return all < cap ? all : (cap > 128 * rows ? cap : 128 * rows);
}
// gm_snp.hip: --snp (SNPScoredSeq): pair HMM per kept sequence + the deposit of its posteriors
size_t gmk_pair_hmm_cells(uint32_t Lmax);            // doubles of scratch per wavefront (64 kept sequences)
int gmk_pair_hmm(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, const GmCand* items, uint32_t n, double* scratch, uint32_t Lmax, float* hmm, void* stream);
int gmk_snp_deposit(float* cov, float* nuc, uint64_t bins, uint32_t bin_size, const GmDevBatch& b, const GmDevMatch* matches, const GmDevPos* positions,
                    uint32_t m0, uint32_t count, const float* post, const float* hmm, uint32_t Lmax, void* stream);
int gmk_compact(const GmDevBatch& b, void* stream);
int gmk_scan_hits(const GmDevBatch& b, void* stream);
int gmk_scatter(const GmDevBatch& b, uint32_t grid, void* stream);
int gmk_sa_interval(const GmDevIndex& ix, const uint8_t* kmers, uint32_t n, uint32_t m, uint32_t* start, uint32_t* end, void* stream);
int gmk_locate(const GmDevIndex& ix, const uint32_t* ranks, uint32_t n, int use_full_sa, uint32_t* out, void* stream);
// traceback operations: 2 bits each (0 M, 1 I, 2 D), operation k in 64-bit word k / 32 at bits 2 (k % 32); ops_words words per item
inline uint32_t gm_ops_words(uint32_t stride) { return (2u * ((stride + 7u) & ~7u) + 8u + 31u) / 32u; }
int gmk_traceback(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, const GmCand* items, uint32_t n,
                  unsigned long long* ops, uint32_t ops_words, uint16_t* ops_len, const uint8_t* emit, uint32_t* cig_cnt, uint32_t* max_span, void* stream);
int gmk_scan_u32(const uint32_t* in, uint64_t n, uint64_t* out, unsigned long long* tmp, void* stream);
int gmk_group_count(const GmDevIndex& ix, const GmDevBatch& b, const GmDevGroup& g, int nw, int unique_only, uint32_t max_matches, void* stream);
int gmk_group_write(const GmDevBatch& b, const GmDevGroup& g, void* stream);
int gmk_out_items(const GmDevMatch* matches, uint32_t n_m, uint32_t read_base, GmCand* items, uint32_t* pos_match, void* stream);
int gmk_out_sizes(const GmDevMatch* matches, uint32_t n_m, const uint8_t* emit, const uint32_t* cig_all, uint32_t* rec_cnt, uint32_t* cig_cnt, void* stream);
int gmk_out_write(const GmDevIndex& ix, const GmDevBatch& b, const GmDevMatch* matches, const GmDevPos* positions, uint32_t n_m, const uint8_t* emit,
                  const int32_t* mapq, const float* post, const unsigned long long* ops, uint32_t ops_words, const uint16_t* ops_len, int nw,
                  const uint64_t* rec_off, const uint64_t* cig_off, GmDevSamRec* recs, char* pool, void* stream);
int gmk_out_codes(const GmDevBatch& b, const GmDevParams& p, const GmDevMatch* matches, uint32_t n_m, const unsigned long long* ops, uint32_t ops_words,
                  const uint16_t* ops_len, uint8_t* codes, uint32_t codes_stride, void* stream);
int gmk_out_deposit(float* cov, uint64_t bins, uint32_t bin_size, const GmDevMatch* matches, const GmDevPos* positions, const uint32_t* pos_match,
                    uint64_t n_p, const uint16_t* ops_len, const float* post, uint32_t max_span, float* nuc, const uint8_t* codes, uint32_t codes_stride,
                    void* stream);
int gmk_coverage_add(float* cov, uint64_t bins, uint32_t bin_size, const uint64_t* pos, const uint32_t* span, const float* w,
                     uint32_t n, uint32_t max_span, float* nuc, const uint8_t* codes, const uint64_t* code_off, void* stream);
}
#endif
