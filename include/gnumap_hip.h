/* gnumap_hip.h — C ABI of libgnumap_hip.so: the MI355X (gfx950) implementation of GNUMAP's per-read
 * seed-and-extend hot path.  Plain C types only; no exceptions cross the boundary; every function
 * returns GM_OK (0) or a negative gm_status.
 *
 * What each entry point replaces in the reference (paths relative to the reference tree):
 *
 *   gm_index_open / gm_index_close   GenomeBwt::LoadGenome            src/GenomeBwt.cpp:282-328
 *                                    bwa_idx_load_from_disk           src/GenomeBwt.cpp:112-140
 *                                    (bwt_restore_bwt/_sa src/bwt.c:421-461, bns_restore src/bntseq.c:169)
 *   gm_index_build                   GenomeBwt::index_and_store       src/GenomeBwt.cpp:330-339 -> bwa_index src/bwtindex.c:187
 *   gm_params_default/_finalize      globals inc/const_define.h:29-131, setup_alignment_matrices inc/a_matrices.c:25,
 *                                    -b/-d table edits src/Driver.cpp:1260-1313, jump default src/Driver.cpp:1206
 *   gm_map_batch                     the per-thread block loop over set_top_matches   src/Driver.cpp:2344-2356
 *                                    = set_top_matches src/Driver.cpp:432-612, align_sequence/process_hits
 *                                    inc/align_seq2_raw.cpp:22-328, Genome::get_sa_int/get_sa_coord/GetString
 *                                    inc/Genome.h:88,98,99, bin_seq::get_align_score inc/bin_seq.h:133
 *   gm_output_batch                  the block loop over create_match_output          src/Driver.cpp:2360-2373
 *                                    = create_match_output src/Driver.cpp:614-753, NormalScoredSeq::score
 *                                    src/NormalScoredSeq.cpp:24-76, ScoredSeq::get_SAM inc/ScoredSeq.h:293-404,
 *                                    bin_seq::get_align_score_w_traceback inc/bin_seq.h:175, GenomeBwt::AddScore
 *                                    src/GenomeBwt.cpp:483-490
 *   gm_coverage_*                    amount_genome, PrintFinalSGR src/GenomeBwt.cpp:1212-1273; the MPI
 *                                    Allreduce of the track src/Driver.cpp:1660-1672 (-> RCCL)
 *
 * Threading: one gm_index per device; a gm_batch is used by one host thread at a time; different
 * batches of the same index may be driven concurrently from different host threads / HIP streams
 * (the index is read-only; coverage deposits are float atomics, like the reference's mutex-ordered adds).
 */
#ifndef GNUMAP_HIP_H
#define GNUMAP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GM_OK = 0,
    GM_E_ARG = -1,          /* bad argument */
    GM_E_IO = -2,           /* index / file problem (reference: throw new Exception / err_fatal) */
    GM_E_NO_DEVICE = -3,    /* no usable HIP device or the gfx950 code object cannot be loaded */
    GM_E_HIP = -4,          /* a HIP call failed: gm_last_error() has the text */
    GM_E_CAPACITY = -5,     /* caller-provided output buffers too small: required sizes are written back */
    GM_E_UNSUPPORTED = -6,  /* e.g. reference longer than 2^32-2 bases (the reference's own `unsigned int vit`
                               loop, inc/align_seq2_raw.cpp:262, has the same limit) */
    GM_E_NOMEM = -7,
    GM_E_BAD_QUAL = -8,     /* negative base probability: reference throws "Invalid Fastq Character", SeqReader.cpp:1181-1189 */
    GM_E_BATCH_TOO_LARGE = -9   /* the block's intermediate lists (candidates, SA hits of one read) exceed what one launch addresses:
                                   map the block in smaller pieces (a repeat-rich reference without -h can yield > 2^31 candidates) */
} gm_status;

/* per-read status, the reference's sentinel values (inc/const_include.h:183-186) */
enum { GM_READ_OK = 0, GM_READ_TOO_MANY = 1, GM_READ_NONE = 2, GM_READ_TOO_SHORT = -2, GM_READ_TOO_POOR = -3 };
enum { GM_MODE_NORMAL = 0, GM_MODE_BS = 1, GM_MODE_BS2 = 2, GM_MODE_ATOG = 3, GM_MODE_ATOG2 = 4,
       GM_MODE_SNP = 5 /* --snp: SNPScoredSeq (src/SNPScoredSeq.cpp:25-109) - mapping as GM_MODE_NORMAL; gm_output_batch deposits the pair-HMM
                          posteriors of every kept sequence into the five per-nucleotide tracks (bin size 1, gm_coverage_enable_nuc first) */ };
enum { GM_POS_STRAND = 0, GM_NEG_STRAND = 1 };

/* gm_index_open flags */
enum {
    GM_INDEX_FULL_SA = 1,       /* expand the rank-sampled SA to a full 32-bit SA in HBM once (locate = one 4-byte read) */
    GM_INDEX_BUILD = 2,         /* build <fa>.gnumap.* when missing, like GenomeBwt::LoadGenome */
    GM_INDEX_HOST_ONLY = 4      /* no device: header/contig queries and gm_index_build only (CPU-side tests) */
};

typedef struct gm_index gm_index;
typedef struct gm_batch gm_batch;

typedef struct {
    uint64_t l_pac, seq_len, primary, bwt_words, n_sa;
    uint32_t sa_intv, n_seqs;
    int device_id, full_sa;
    uint64_t hbm_bytes;         /* device bytes held by the index */
} gm_index_info;

/* mirrors the globals of inc/const_define.h; fill with gm_params_default, edit, then gm_params_finalize */
typedef struct {
    int mer;                    /* -m  gMER_SIZE (10) */
    int jump;                   /* -j  gJUMP_SIZE (0 -> mer/2) */
    int min_seed_hits;          /* -k  gMIN_JUMP_MATCHES (2) */
    uint32_t max_kmer_hits;     /* -h  gMAX_KMER_SIZE (0 = unlimited) */
    uint32_t max_matches;       /* -T  gMAX_MATCHES (1000) */
    int max_gap;                /* -M  gMAX_GAP (3); 1 .. 7 (3 runs the register-band kernels, the rest the LDS-band kernels of gm_band.hip) */
    int nw;                     /* !--no_nw */
    int fast;                   /* --fast */
    int unique_only;            /* -u */
    int pos_strand, neg_strand; /* --up_strand / --down_strand */
    int mode;                   /* GM_MODE_* (-b, --b2, -d, --snp) */
    float align_score;          /* -a  gALIGN_SCORE (0.9) */
    int align_is_fraction;      /* perc (1); 0 with -r */
    float cutoff;               /* -q  gCUTOFF_SCORE (0) */
    float adjust, match, transition, transversion, gap;     /* gADJUST .25, 3, -2, -3, -4 (scaled by finalize) */
    float S[256][4];            /* gALIGN_SCORES, built by finalize */
    int bin_size;               /* --bin_size gGEN_SIZE (8; forced to 1 by -b/-d) */
    int print_all_sam;          /* --print_all_sam */
    int illumina;               /* --illumina (Phred+64 with automatic fallback, SeqReader.cpp:1171-1180) */
    int finalized;
} gm_params;

/* a block of reads exactly as they stand in the FASTQ file; caller-owned host memory */
typedef struct {
    uint32_t n;                 /* number of reads */
    uint32_t stride;            /* bytes between consecutive reads in bases[] / quals[] (>= longest read, multiple of 8) */
    const uint8_t* bases;       /* n x stride, FASTQ sequence characters verbatim (any case, N allowed) */
    const uint8_t* quals;       /* n x stride, raw FASTQ quality characters */
    const uint16_t* len;        /* n read lengths */
} gm_reads;

typedef struct { uint64_t pos; uint8_t strand; } gm_pos;

/* one ScoredSeq (inc/ScoredSeq.h:33): a unique genomic sequence the read matched, with all its places */
typedef struct {
    uint32_t read;              /* index of the read in the batch */
    float score;                /* align_score of the FIRST hit (fp32 NW score, or the vote count with --no_nw) */
    uint64_t first_pos;         /* position whose window is ScoredSeq::sequence */
    uint8_t first_strand;       /* ScoredSeq::firstStrand */
    uint32_t pos_begin, pos_end;/* [pos_begin,pos_end) in positions[]: the ordered set<(pos,strand)> */
} gm_match;

/* result of gm_map_batch: what set_top_matches leaves in gReadLocs / gReadDenominator / gTopReadScore.
 * All arrays are caller-owned; on GM_E_CAPACITY matches_cap / positions_cap hold the required sizes. */
typedef struct {
    uint32_t n;
    int8_t* status;             /* n: GM_READ_* */
    float* self_score;          /* n: max_align_score (Driver.cpp:466) */
    double* top_score;          /* n: gTopReadScore */
    double* denominator;        /* n: gReadDenominator */
    uint64_t* match_begin;      /* n+1: CSR into matches[], matches of a read in std::map (key) order */
    gm_match* matches;  uint64_t matches_cap;
    gm_pos* positions;  uint64_t positions_cap;
    uint64_t stamp;             /* set by gm_map_batch: names the result that is still resident in the batch's HBM.  gm_output_batch given the
                                   same stamp uses the resident matches / positions as they are (no upload, no re-validation); a caller
                                   that EDITS matches / positions / match_begin sets stamp = 0 (status, denominator and top_score are always
                                   taken from the host arrays).  A gm_hits the caller builds by hand must be ZERO-INITIALISED (stamp = 0 =
                                   "nothing resident"); gm_map_batch sets it to 0 first thing on every call (ABI 0.2) */
} gm_hits;

/* one SAM record (TopReadOutput, inc/const_include.h:193-206) */
typedef struct {
    uint32_t read;
    uint64_t pos;               /* 0-based position on the concatenated reference */
    uint32_t contig;            /* index into the contig table */
    uint64_t chr_pos;           /* 1-based position on the contig */
    uint8_t strand;
    int32_t mapq;
    float a_score;              /* XA before the 1/gADJUST rescale */
    float post_prob;            /* XP */
    int32_t sim_matches;        /* X0 */
    uint32_t cigar_off;         /* offset of the NUL-terminated CIGAR (forward orientation) in cigar_pool */
} gm_sam_rec;

typedef struct {
    gm_sam_rec* recs;  uint64_t recs_cap, n_recs;        /* caller-owned; in read order, then ScoredSeq position order */
    char* cigar_pool;  uint64_t cigar_cap, cigar_len;    /* caller-owned */
} gm_sam_out;

/* kernel-side work counters of the last gm_map_batch on a batch (algorithmic-bytes accounting) */
typedef struct {
    uint64_t reads, kmers_searched, occ_calls, occ_blocks, seeds_used, sa_hits, lf_steps, candidates, nw_cells, accepted, vote_retries,
        table_lookups;
} gm_counters;

/* kernels of gm_map_batch_device, for the HIP-event timing of gm_batch_kernel_times.  GM_K_SEED has no launches when the seed
 * lookup runs inside the vote kernel (full SA, k-mer table as long as the seed, k-mers expected >= 4 times in the reference, no -h;
 * env GM_SEED_FUSED=0|1 overrides): its time is then part of GM_K_VOTE; the work counters are the same either way */
enum { GM_K_PREP = 0, GM_K_SEED, GM_K_LOCATE, GM_K_VOTE, GM_K_VOTE_RETRY, GM_K_NW, GM_K_COMPACT, GM_K_COUNT };

const char* gm_last_error(void);
const char* gm_version(void);
/* run-time switches of the library (the GM_* names documented in DESIGN.md: kernel choices, test switches).  Every switch is read
 * on each call that uses it - from this table first, then from the environment - so a long-lived host can change a choice
 * between two batches.  value = NULL removes the override (back to the environment); "" hides an environment variable. */
int gm_set_option(const char* name, const char* value);

/* host-only self test of the library's slice pool (the fp64 passes of the two batch calls are cut over it): `callers` threads run
 * `iters` passes over [0, n) each, at once; returns the number of passes in which an item was not visited exactly once (0 = fine) */
int gm_selftest_pass_parallel(uint32_t n, uint32_t grain, uint32_t callers, uint32_t iters);

/* ---- index ---- */
int gm_index_build(const char* fasta_path);                      /* writes <fa>.gnumap.{pac,ann,amb,bwt,sa}; = gm_index_build_on(fa, GM_BUILD_AUTO, 0) */
/* where the suffix-array stage of the build runs (is_sa/is_bwt src/is.c:53-223): the files are byte-identical either way */
enum { GM_BUILD_AUTO = 0 /* device if there is one, else host; env GM_INDEX_BUILD=host|device */, GM_BUILD_HOST = 1 /* SA-IS */,
       GM_BUILD_DEVICE = 2 /* prefix doubling in HBM, gm_sa_build.hip */ };
int gm_index_build_on(const char* fasta_path, int where, int device_id);
int gm_index_open(const char* fasta_path, int device_id, int flags, gm_index** out);
void gm_index_close(gm_index*);
/* stage what the mapping kernels derive from the index for these parameters (k-mer tables, k-mer -> positions records: tens of GB of HBM
 * at human scale, built on the device in well under a second) now instead of inside the first gm_map_batch; optional */
int gm_index_prepare(gm_index*, const gm_params*);
int gm_index_get_info(const gm_index*, gm_index_info* out);
const char* gm_index_contig_name(const gm_index*, uint32_t i);
uint64_t gm_index_contig_offset(const gm_index*, uint32_t i);   /* i == n_seqs gives l_pac */
/* window fetch on the host copy of the packed reference (GenomeBwt::GetString); returns L or 0 at a boundary */
int gm_index_window(const gm_index*, uint64_t begin, uint32_t L, char* out);

/* ---- parameters ---- */
void gm_params_default(gm_params*);
int gm_params_finalize(gm_params*);
/* -S / --subst_file (readPWM, src/Driver.cpp:768-859; call AFTER gm_params_finalize): 5 rows (genome a, c, g, t, n) x 4 read bases,
 * with or without the label line / row labels; overwrites the LOWERCASE rows of S unscaled, sets adjust = 1 (XA is then printed
 * unscaled) and leaves the gap penalty as finalize scaled it - exactly what the reference does. */
int gm_params_load_subst(gm_params*, const char* path);

/* ---- batches: device-resident reads + workspace + raw results ---- */
int gm_batch_create(gm_index*, uint32_t max_reads /* <= 16 000 000 */, uint32_t max_len /* <= 2048 */, gm_batch** out);
void gm_batch_destroy(gm_batch*);
int gm_batch_upload(gm_batch*, const gm_params*, const gm_reads*, void* hip_stream);       /* host -> HBM */
/* the hot path proper, everything resident in HBM: prep -> seed -> locate+vote -> NW -> hit compaction.
 * Asynchronous on hip_stream except for one 16-byte size read-back used to size the workspace. */
int gm_map_batch_device(gm_index*, const gm_params*, gm_batch*, void* hip_stream);
int gm_batch_counters(gm_batch*, gm_counters* out);
/* per-kernel device time: HIP events recorded on the launch stream around every kernel of gm_map_batch_device while
 * profiling is on; gm_batch_kernel_times adds up what has completed since the last call (ms and launch counts) */
int gm_batch_set_profiling(gm_batch*, int on);
int gm_batch_kernel_times(gm_batch*, double* ms /* GM_K_COUNT */, uint64_t* launches /* GM_K_COUNT */);
const char* gm_kernel_name(int which);
/* which seed lookup / vote kernel / locate form the last gm_map_batch_device on this batch chose, e.g.
 * "seeds=bucket-table (in the vote kernel) vote=k_vote_bucket<4> locate=full-SA" (valid until the next call on the batch) */
const char* gm_batch_path(gm_batch*);
/* raw device results (accepted candidates), for tests and for callers that post-process themselves */
typedef struct { uint32_t read; uint32_t pos; float score; uint16_t step; uint8_t strand; uint8_t pad; } gm_raw_hit;
int gm_batch_raw_hits(gm_batch*, gm_raw_hit* out, uint64_t cap, uint64_t* n_out,
                      int8_t* status, float* self_score, float* top_score);

/* ---- the drop-in pair for the two block loops of parallel_thread_run ----
 * gm_map_batch   = upload + gm_map_batch_device + the unique-sequence map of process_hits as device kernels (processing order, key
 *                  grouping on the 2-bit reference, std::map order, -T / -u exits); the host does one flat fp64 pass
 *                  (denominator += exp(score) in the reference's order) on the calling thread.
 * gm_output_batch= one flat fp64 pass on the calling thread (posterior, winner, MAPQ), then traceback, run-length CIGAR text,
 *                  SAM rows and the coverage deposit as device kernels.  `reads` must be the block the batch was mapped with (it
 *                  is still resident in HBM); `hits` may have been edited by the caller (it is uploaded again).
 * Both are synchronous on hip_stream and use only the calling thread: a driver overlaps blocks by calling them from two or more
 * threads with one gm_batch and one stream each.  Host buffers from gm_host_alloc are page-locked (DMA at link rate). */
int gm_stream_create(gm_index*, void** hip_stream_out);          /* a non-blocking HIP stream on the index's device (one per driver thread) */
void gm_stream_destroy(gm_index*, void* hip_stream);
void* gm_host_alloc(size_t bytes);                               /* page-locked host memory for gm_reads / gm_hits / gm_sam_out buffers */
void gm_host_free(void*);
int gm_map_batch(gm_index*, const gm_params*, gm_batch*, const gm_reads*, gm_hits* out, void* hip_stream);
int gm_output_batch(gm_index*, const gm_params*, gm_batch*, const gm_reads*, const gm_hits*, gm_sam_out* out, void* hip_stream);

/* enqueue / wait forms: the call is queued on the batch's own service thread and runs there exactly as the synchronous form would;
 * the caller goes on (with another batch).  Calls queued on one batch run in order - gm_output_batch_enqueue may be queued right
 * behind the gm_map_batch_enqueue whose gm_hits it reads - and after a failing call the rest of the batch's queue is skipped.
 * gm_batch_wait returns when the batch's queue is empty: GM_OK, or the status of the first call that failed (gm_last_error() has its
 * text).  gm_params / gm_reads are copied at enqueue time; the arrays behind gm_reads, gm_hits and gm_sam_out stay the caller's until
 * the wait.  One HIP stream per batch, as with the synchronous forms. */
int gm_map_batch_enqueue(gm_index*, const gm_params*, gm_batch*, const gm_reads*, gm_hits* out, void* hip_stream);
int gm_output_batch_enqueue(gm_index*, const gm_params*, gm_batch*, const gm_reads*, const gm_hits*, gm_sam_out* out, void* hip_stream);
int gm_batch_wait(gm_batch*);

/* ---- unit-level device entry points (parity tests) ---- */
int gm_dev_sa_interval(gm_index*, const char* kmers, uint32_t n, uint32_t m, uint64_t* start, uint64_t* end);
int gm_dev_locate(gm_index*, const uint64_t* ranks, uint32_t n, int use_full_sa, uint64_t* out);
int gm_dev_nw_score(gm_index*, const gm_params*, const gm_reads*, const uint32_t* read_idx, const uint8_t* strand,
                    const uint64_t* pos, uint32_t n, float* score, uint8_t* valid);
int gm_dev_traceback(gm_index*, const gm_params*, const gm_reads*, const uint32_t* read_idx, const uint8_t* strand,
                     const uint64_t* pos, uint32_t n, char* ops /* n x ops_stride, 'M','I','D', NUL padded */,
                     uint32_t ops_stride, uint16_t* ops_len);

/* bin_seq::pairHMM (src/bin_seq.cpp:60-244) of read read_idx[k] in the orientation of strand[k] against the window at pos[k]:
 * out[k][max_len][5] floats (a, c, g, t, n per window position), max_len = gm_reads.stride */
int gm_dev_pair_hmm(gm_index*, const gm_params*, const gm_reads*, const uint32_t* read_idx, const uint8_t* strand, const uint64_t* pos, uint32_t n, float* out);

/* ---- coverage track (amount_genome) ---- */
int gm_coverage_reset(gm_index*, uint32_t bin_size);
uint64_t gm_coverage_bins(const gm_index*);
void* gm_coverage_device_ptr(gm_index*);                         /* float[bins] in HBM, for RCCL all-reduce by the caller */
int gm_coverage_add(gm_index*, const uint64_t* pos, const uint32_t* span, const float* w, uint32_t n, void* hip_stream);
int gm_coverage_download(gm_index*, float* host /* bins */);
int gm_coverage_allreduce(gm_index** per_gpu, int n_gpu);        /* single-process multi-GPU: ncclAllReduce(sum) over xGMI */
int gm_coverage_write_sgr(gm_index*, const float* host_bins, const char* path, int append);
/* -b / -d (bisulfite, A->G): the per-nucleotide track reads[A,C,G,T,N][loc] of BSScoredSeq::score (src/BSScoredSeq.cpp:24-88,
 * GenomeBwt::AddSeqScore src/GenomeBwt.cpp:556-603), 5 x bins floats in HBM, filled by gm_output_batch once enabled, and the
 * .gmp writer (GenomeBwt::PrintFinalBisulfite src/GenomeBwt.cpp:1092-1210; these modes write <out>.gmp INSTEAD of <out>.sgr) */
int gm_coverage_enable_nuc(gm_index*);
void* gm_coverage_nuc_device_ptr(gm_index*);
int gm_coverage_download_nuc(gm_index*, float* host /* 5 x bins */);
int gm_coverage_write_gmp(gm_index*, const gm_params*, const float* host_bins, const float* host_nuc, const char* path, int append);
/* (GM_MODE_SNP: PrintFinalSNP src/GenomeBwt.cpp:930-1090 - every position above 0.001: contig, position, %.5f total, five %.5f sums.  The
 * likelihood-ratio columns PrintSNPCall appends are NOT written: they need GSL's gsl_cdf_chisq_P, outside this hot path.) */

#ifdef __cplusplus
}
#endif
#endif
