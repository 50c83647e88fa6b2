// gm_api.cpp — the C ABI of libgnumap_hip.so (include/gnumap_hip.h): index residency in HBM, batch staging,
// the kernel pipeline, and the host-side mirror of the reference's per-read bookkeeping
// (unique-sequence map, denominator, posterior, MAPQ: src/Driver.cpp:432-753, inc/align_seq2_raw.cpp:102-165,
// inc/ScoredSeq.h:293-404).  There is NO CPU fallback for the device work: without a usable gfx950 device every
// compute entry point fails with GM_E_NO_DEVICE.
#include "gm_host.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <set>
#include <thread>
#include <fcntl.h>
#include <unistd.h>

static thread_local std::string g_err;
// ---- run-time switches: gm_set_option() overrides, else the environment; never latched (see gm_internal.h) ----
namespace { std::mutex g_opt_mu; std::map<std::string, const char*> g_opts; }      // values are interned (never freed): a reader may still hold one
const char* gm_opt(const char* name) {
    const char* v = nullptr; bool have = false;
    {
        std::lock_guard<std::mutex> lk(g_opt_mu);
        auto it = g_opts.find(name);
        if (it != g_opts.end()) { v = it->second; have = true; }
    }
    if (!have) v = getenv(name);
    return v && *v ? v : nullptr;                              // an empty value counts as unset
}
extern "C" int gm_set_option(const char* name, const char* value) {
    if (!name || strncmp(name, "GM_", 3) != 0) return GM_E_ARG;
    std::lock_guard<std::mutex> lk(g_opt_mu);
    if (value) g_opts[name] = strdup(value); else g_opts.erase(name);       // null: back to the environment
    return GM_OK;
}
static bool gm_trace_on() { return gm_opt_ll("GM_TRACE", 0) != 0; }
static double gm_trace_ms() { static const auto t0 = std::chrono::steady_clock::now(); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
#define GM_TRACE(...) do { if (gm_trace_on()) { fprintf(stderr, "[gm_trace %9.1f ms] ", gm_trace_ms()); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)
void gm_set_error(const std::string& s) { g_err = s; }
extern "C" const char* gm_last_error(void) { return g_err.c_str(); }
extern "C" const char* gm_version(void) { return "gnumap-mi355x 0.2 (gfx950)"; }

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            gm_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                                  \
            return GM_E_HIP;                                                                                  \
        }                                                                                                     \
    } while (0)
#define KCHK(expr)                                                                                            \
    do {                                                                                                      \
        int e_ = (expr);                                                                                      \
        if (e_ != 0) {                                                                                        \
            gm_set_error(std::string(#expr) + ": " + hipGetErrorString((hipError_t)e_));                      \
            return GM_E_HIP;                                                                                  \
        }                                                                                                     \
    } while (0)

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return GM_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { gm_set_error(std::string("hipMalloc: ") + hipGetErrorString(e)); p = nullptr; return GM_E_NOMEM; }
        cap = want;
        return GM_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct PinBuf {                              // page-locked host staging: device <-> host copies at link rate, no zero fill
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return GM_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) { gm_set_error(std::string("hipHostMalloc: ") + hipGetErrorString(e)); p = nullptr; return GM_E_NOMEM; }
        cap = want;
        return GM_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

struct gm_index {
    GmHostIndex h;
    int device = -1;
    bool host_only = false;
    bool full_sa = false;
    DevBuf d_bwt, d_sa, d_full, d_pac, d_contig, d_cov, d_ptab, d_planes, d_nuc;
    bool nuc_on = false;
    GmDevIndex dev{};
    uint64_t cov_bins = 0;
    uint32_t cov_bin_size = 0;
    // parameter tables resident in HBM: S256 (256x4 floats) + lut (512 float2)
    std::map<std::vector<float>, DevBuf> ptabs;   // by content
    std::map<int, DevBuf> kmer_tabs;        // memoised backward search of the last T characters of a seed, per T
    std::map<int, DevBuf> kmer_ctabs;       // its compact form (16 B per 8 codes), per T
    std::map<int, DevBuf> buckets;          // k-mer -> positions records (128 B per code; gm_bucket.hip), per T; empty DevBuf = tried, no room
    std::mutex mu;
    uint64_t hbm_bytes = 0;
};

struct gm_batch {
    gm_index* ix = nullptr;
    uint32_t max_reads = 0, max_len = 0;
    uint32_t len_max = 0;                 // longest read of the uploaded block
    uint32_t len_min = 0;                 // and the shortest
    uint32_t n = 0, stride = 0, max_seeds = 0, illumina_until = 0;
    DevBuf bases, quals, len, status, self_score, min_score, top_score, seeds, n_seeds, n_entries, entry_off, coords,
        rs_overflow, retry_list, retry_off, gtab_keys, gtab_vals, cands, fixed_cands, fixed_cnt, heavy_list, heavy_off, heavy_k0, heavy_k1, heavy_tmp, hit_count, hit_begin, hit_cursor, raw_hits, counters, small, shards, big_list,
        tb_items, tb_ops, tb_len, band_moves, pack,
        // grouping (process_hits' unique map) and output stage, gm_output.hip
        g_sorted, g_ord, g_lead, g_krank, g_khash, g_nmatch, g_mbegin, g_multi, g_big, g_bigdone, g_sk0, g_sk1, g_si0, g_si1, g_matches, g_mhit, g_positions, scan_tmp,
        o_small, o_posmatch, o_post, o_mapq, o_emit, o_reccnt, o_cigcnt, o_cigall, o_recoff, o_cigoff, o_recs, o_pool, o_codes,
        pair_fb, pair_list,             // k_vote_pair: the reads it leaves to k_vote_bucket (one byte each), their list + its counter
        snp_scratch, snp_hmm;           // --snp: forward matrices of a chunk of kept sequences, their 5 floats per window position
    PinBuf h_top, h_hbegin, h_ord, h_post, h_mapq, h_emit, h_mhit, h_stat;      // h_stat: the small status words a phase reads back (page-locked: one short DMA)
    std::vector<double> h_exp;          // exp(score) of every accepted hit of the last gm_map_batch (reused by gm_output_batch)
    uint64_t cache_hits = 0, cache_matches = 0;
    uint32_t fixed_epoch = 0;           // launch stamp of the own-slot candidates of k_vote_bucket (GmDevBatch::fixed_epoch); 0 = the count array form
    uint32_t epoch_ctr = 0;
    uint64_t stamp = 0;                 // names the gm_map_batch result resident in g_matches / g_positions (gm_hits::stamp); 0 = none
    std::string path;                   // which kernels the last gm_map_batch_device chose (gm_batch_path)
    // gm_*_enqueue / gm_batch_wait: the batch's own service thread runs the queued calls in order, the caller goes on
    struct Service {
        std::thread th; std::mutex mu; std::condition_variable cv;
        std::deque<std::function<int()>> q;
        bool busy = false, quit = false;
        int rc = GM_OK; std::string err;
    } svc;
    bool use_pack = false;              // the fused seed lookup is on for this pass (k_prep writes the 2-bit read forms)
    const void* resume_ptr = nullptr;   // set when gm_map_batch returned GM_E_CAPACITY: the next call with the same reads resumes at the copies
    uint32_t cand_cap = 0;
    bool use_fixed = false;             // k_vote_tiny* leave their candidates in per read x strand slots (gathered by k_cand_gather)
    uint64_t raw_cap = 0;
    uint32_t n_cands = 0;
    uint64_t n_raw = 0;
    bool mapped = false;
    unsigned long long counters_host[GMK_N] = { 0 };
    GmDevBatch dev{};
    std::vector<uint16_t> len_host;
    // sub-batch pipeline (gm_map_batch_device on large batches): streams + per-stream retry tables + host counter sums
    static const int NS = 3;
    hipStream_t sub_streams[NS] = { nullptr, nullptr, nullptr };
    hipEvent_t sub_ready = nullptr;
    DevBuf sub_gk[NS], sub_gv[NS], sub_counters, sub_small, sub_shards;
    bool counters_on_host = false;
    // HIP-event profiling of the kernels of gm_map_batch_device
    bool profiling = false;
    struct Ev { int which; hipEvent_t a, b; };
    std::vector<Ev> pending;
    std::vector<hipEvent_t> pool;
    double k_ms[GM_K_COUNT] = { 0 };
    uint64_t k_launches[GM_K_COUNT] = { 0 };
};

namespace {
struct KTimer {     // brackets one kernel (or one small group of launches) with events on its stream
    gm_batch* b; int which; hipStream_t st; hipEvent_t a = nullptr, e = nullptr; bool on;
    KTimer(gm_batch* b_, int w, hipStream_t s) : b(b_), which(w), st(s), on(b_->profiling) {
        if (!on) return;
        auto get = [&]() { hipEvent_t ev = nullptr; if (!b->pool.empty()) { ev = b->pool.back(); b->pool.pop_back(); } else if (hipEventCreate(&ev) != hipSuccess) ev = nullptr; return ev; };
        a = get(); e = get();
        if (!a || !e) { on = false; return; }
        (void)hipEventRecord(a, st);
    }
    ~KTimer() {
        if (!on) return;
        (void)hipEventRecord(e, st);
        b->pending.push_back({ which, a, e });
    }
};
}  // namespace

// ------------------------------------------------------------------------------------------------
// parameters (const_define.h:29-131, a_matrices.c:55-83, Driver.cpp:1206,1260-1313,2805-2816)
// ------------------------------------------------------------------------------------------------
extern "C" void gm_params_default(gm_params* p) {
    memset(p, 0, sizeof *p);
    p->mer = 10; p->jump = 0; p->min_seed_hits = 2; p->max_kmer_hits = 0; p->max_matches = 1000; p->max_gap = 3;
    p->nw = 1; p->fast = 0; p->unique_only = 0; p->pos_strand = 1; p->neg_strand = 1; p->mode = GM_MODE_NORMAL;
    p->align_score = 0.9f; p->align_is_fraction = 1; p->cutoff = 0.0f;
    p->adjust = 0.25f; p->match = 3; p->transition = -2; p->transversion = -3; p->gap = -4;
    p->bin_size = 8; p->print_all_sam = 0; p->illumina = 0; p->finalized = 0;
}

extern "C" int gm_params_finalize(gm_params* p) {
    if (!p) return GM_E_ARG;
    if (p->finalized) return GM_OK;
    if (p->mer < 1 || p->mer > 32) { gm_set_error("-m/--mer_size: must be 1..32 (MAX_MER_SIZE)"); return GM_E_ARG; }
    if (p->min_seed_hits < 1) { gm_set_error("-k/--num_seed: Invalid matching seed number"); return GM_E_ARG; }
    if (p->max_gap < 1 || p->max_gap > 7) { gm_set_error("-M/--max_gap: 1 .. 7 (band rows of up to 17 cells)"); return GM_E_UNSUPPORTED; }
    if (p->jump <= 0) p->jump = p->mer / 2;
    if (p->jump < 1) p->jump = 1;
    p->match *= p->adjust; p->transition *= p->adjust; p->transversion *= p->adjust; p->gap *= p->adjust;
    for (int i = 0; i < 256; ++i) for (int j = 0; j < 4; ++j) p->S[i][j] = p->transversion;
    static const char lo[4] = { 'a', 'c', 'g', 't' }, up[4] = { 'A', 'C', 'G', 'T' };
    for (int g = 0; g < 4; ++g)
        for (int b = 0; b < 4; ++b) {
            float v = g == b ? p->match : ((g ^ b) == 2 ? p->transition : p->transversion);
            p->S[(int)lo[g]][b] = v;
            p->S[(int)up[g]][b] = v;
        }
    if (p->mode == GM_MODE_BS) p->S['c'][3] = p->match;
    if (p->mode == GM_MODE_BS2) p->S['g'][0] = p->match;
    if (p->mode == GM_MODE_ATOG) p->S['a'][2] = p->match;
    if (p->mode == GM_MODE_ATOG2) p->S['t'][1] = p->match;
    if (p->mode != GM_MODE_NORMAL) p->bin_size = 1;
    if (p->bin_size < 1) { gm_set_error("--bin_size: must be >= 1"); return GM_E_ARG; }
    p->finalized = 1;
    return GM_OK;
}

extern "C" int gm_params_load_subst(gm_params* p, const char* path) {
    if (!p || !path || !p->finalized) { gm_set_error("gm_params_load_subst: finalize the parameters first"); return GM_E_ARG; }
    FILE* f = fopen(path, "r");
    if (!f) { gm_set_error(std::string("cannot open substitution file ") + path); return GM_E_IO; }
    float t[5][4];
    char line[100], l0[100], l1[100], l2[100], l3[100];
    int count = 0; bool labels = false;
    auto next_line = [&]() -> bool {                           // ifstream::getline(buf, 100): at most 99 characters of a line
        if (!fgets(line, sizeof line, f)) return false;
        size_t n = strlen(line);
        if (n && line[n - 1] == '\n') line[n - 1] = 0;
        return true;
    };
    while (count < 5 && next_line()) {
        if (sscanf(line, "%99s %99s %99s %99s", l0, l1, l2, l3) == 4 && tolower((unsigned char)l0[0]) == 'a' && tolower((unsigned char)l1[0]) == 'c') {
            if (!next_line()) break;                           // the label line: rows carry a label from here on
            labels = true;
        }
        float a, c, g, tt;
        const int got = labels ? sscanf(line, "%99s %f %f %f %f", l0, &a, &c, &g, &tt) - 1 : sscanf(line, "%f %f %f %f", &a, &c, &g, &tt);
        if (got != 4) { fclose(f); gm_set_error(std::string("Error in Score File: ") + line); return GM_E_IO; }
        t[count][0] = a; t[count][1] = c; t[count][2] = g; t[count][3] = tt;
        ++count;
    }
    fclose(f);
    if (count < 5) { gm_set_error("Error in Score File:  Not enough lines"); return GM_E_IO; }
    static const char rows[5] = { 'a', 'c', 'g', 't', 'n' };
    for (int r = 0; r < 5; ++r) for (int b = 0; b < 4; ++b) p->S[(int)rows[r]][b] = t[r][b];
    p->adjust = 1.0f;                                          // gADJUST = 1: "don't adjust the scores" (XA:f: prints score * 1 / gADJUST)
    return GM_OK;
}

// Q -> (p, (1-p)/3) in fp32 exactly as SeqReader::get_more_fastq computes them (fp64 libm, then one cast):
// Q2Prb_std src/SeqReader.cpp:623-627, Q2Prb_ill :618-622.  Negative p is stored as NaN.
static void build_lut(float* lut /* 512 x 2 */) {
    for (int which = 0; which < 2; ++which)
        for (int ch = 0; ch < 256; ++ch) {
            double p;
            const int sc = ch >= 128 ? ch - 256 : ch;                // `int Q = (int)fastq[i]` (SeqReader.cpp:1155): the character is a signed char,
            if (which == 0) { int Q = sc - 33; p = 1 - exp((-(double)Q / 10.0) * log(10.0)); }       // a byte above 127 a negative quality -> "Invalid Fastq Character"
            else { int Q = sc - 64; p = 1.0 - 1.0 / (pow(10.0, ((double)Q / 10.0))); }
            if (p > 1.0) p = 1.0;
            double other = (1 - p) / 3;
            float* o = lut + ((size_t)which * 256 + ch) * 2;
            if (p < 0) { o[0] = NAN; o[1] = NAN; } else { o[0] = (float)p; o[1] = (float)other; }
        }
}

// want_bucket: the caller may use the bucket table of gm_bucket.hip (gm_map_batch_device): built on first use when it applies
static int sync_params(gm_index* ix, const gm_params* p, GmDevParams& dp, hipStream_t st, bool want_bucket = false) {
    std::vector<float> tab(256 * 4 + 512 * 2);
    memcpy(tab.data(), p->S, sizeof(float) * 1024);
    build_lut(tab.data() + 1024);
    // parameter tables are kept per CONTENT and never overwritten: batches of one index may run concurrently with different
    // gm_params (e.g. --illumina switched off from some block on) without one call rewriting what another call's kernels read
    const float* d_tab = nullptr;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        auto it = ix->ptabs.find(tab);
        if (it == ix->ptabs.end()) {
            if (ix->ptabs.size() >= 64) { gm_set_error("more than 64 distinct parameter tables on one index"); return GM_E_NOMEM; }
            DevBuf nb;
            int rc = nb.ensure(tab.size() * sizeof(float));
            if (rc) return rc;
            HIPCHK(hipMemcpyAsync(nb.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
            it = ix->ptabs.emplace(tab, nb).first;
        }
        d_tab = it->second.as<float>();
    }
    dp.mer = p->mer; dp.jump = p->jump; dp.kmin = p->min_seed_hits; dp.nw = p->nw; dp.fast = p->fast;
    dp.pos_strand = p->pos_strand; dp.neg_strand = p->neg_strand; dp.align_is_fraction = p->align_is_fraction; dp.max_gap = p->max_gap;
    dp.fused = 0; dp.heavy_min = 0xFFFFFFFFu;
    dp.dbg = (int)gm_opt_ll("GM_DBG", 0);
    // k-mer interval table: the last T characters of every seed are one lookup (GM_KMER_TABLE=0 keeps the pure occ walk;
    // GM_KMER_TABLE=<T> picks another suffix length, at most 16)
    dp.kmer_tab = nullptr; dp.kmer_T = 0; dp.kmer_ctab = nullptr; dp.bucket = nullptr; dp.bucket_T = 0; dp.bucket_ctx = 0;
    dp.bucket_ecap = (uint32_t)std::min<long long>(384, gm_opt_ll("GM_BUCKET_ECAP", 384)); dp.bucket_ovcap = (uint32_t)std::min<long long>(16, gm_opt_ll("GM_BUCKET_OVCAP", 16));
    {
        // up to 12 characters by default; more for longer seeds on references where the extra occ steps are HBM misses anyway
        // (>= 50 Mbp): 14 characters = 2 GB + 0.5 GB compact,
        // 15 / 16 characters (8.6 + 2.1 GB / 34 + 8.6 GB: what 288 GB of HBM is for) when the seeds are that long: a 16-mer is then
        // ONE random 16-byte probe instead of a probe + two search steps
        int T = std::min(p->mer, ix->h.seq_len >= 50000000ull ? 16 : 12);
        if (const char* e = gm_opt("GM_KMER_TABLE")) { if (*e) T = std::min(std::min(atoi(e), p->mer), 16); }
        const long long bucket_opt = gm_opt_ll("GM_SEED_BUCKET", -1);
        if (T >= 4) {
            std::lock_guard<std::mutex> lk(ix->mu);
            // the 15- / 16-character tables are bought with free HBM: step down while table + previous level + compact form (+ 8 GB
            // for the batches) do not fit what is free right now
            while (T > 14 && !ix->kmer_tabs.count(T)) {
                size_t fr = 0, tot = 0;
                const size_t need = (((size_t)1 << (2 * T)) + ((size_t)1 << (2 * T - 2))) * 8 + ((size_t)1 << (2 * T - 3)) * 16 + ((size_t)8 << 30);
                if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr >= need) break;
                --T;
            }
            DevBuf& tb = ix->kmer_tabs[T];
            if (!tb.p) {
                if (tb.ensure(((size_t)1 << (2 * T)) * 8)) return GM_E_NOMEM;
                if (T <= 12) {
                    KCHK(gmk_build_kmer_table(ix->dev, tb.as<uint2>(), T, st));
                } else {                             // 12 characters directly, then one character (one search step per entry) at a time
                    DevBuf cur, nxt;
                    if (cur.ensure(((size_t)1 << 24) * 8)) return GM_E_NOMEM;
                    int e2 = gmk_build_kmer_table(ix->dev, cur.as<uint2>(), 12, st);
                    for (int tt = 13; tt <= T && !e2; ++tt) {
                        uint2* dst = tb.as<uint2>();
                        if (tt < T) { if (nxt.ensure(((size_t)1 << (2 * tt)) * 8)) { cur.release(); return GM_E_NOMEM; } dst = nxt.as<uint2>(); }
                        e2 = gmk_extend_kmer_table(ix->dev, cur.as<uint2>(), dst, tt, st);
                        if (hipStreamSynchronize(st) != hipSuccess) e2 = 1;
                        if (tt < T) { cur.release(); cur = nxt; nxt = DevBuf(); }
                    }
                    cur.release(); nxt.release();
                    if (e2) { gm_set_error("k-mer table construction failed"); return GM_E_HIP; }
                }
                HIPCHK(hipStreamSynchronize(st));
                ix->hbm_bytes += tb.cap;
            }
            dp.kmer_tab = tb.as<uint2>(); dp.kmer_T = T;
            const bool compact = !gm_opt_is("GM_KMER_COMPACT", "0");
            if (compact) {
                DevBuf& cb = ix->kmer_ctabs[T];
                if (!cb.p) {
                    if (cb.ensure(((size_t)1 << (2 * T - 3)) * 16)) return GM_E_NOMEM;
                    KCHK(gmk_build_kmer_compact(tb.as<uint2>(), cb.as<uint4>(), T, st));
                    HIPCHK(hipStreamSynchronize(st));
                    ix->hbm_bytes += cb.cap;
                }
                dp.kmer_ctab = cb.as<uint4>();
            }
            // k-mer -> positions records (gm_bucket.hip): full SA, the table covering the whole seed, a k-mer expected between a
            // fraction of a time and ~20 times in the reference (a record holds 28 positions; more go through the suffix array),
            // and 128 bytes per code have to fit beside everything else: -m 14 = 34 GB of the 288.  GM_SEED_BUCKET=0 / 1: never /
            // whenever the table can be built.
            const double occ = (double)ix->h.seq_len / pow(4.0, (double)std::min(p->mer, 31));
            if (want_bucket && bucket_opt != 0 && ix->full_sa && T <= 15 && ix->h.seq_len < 0xFFFFE000ull &&
                T == p->mer && (bucket_opt > 0 || (occ >= 1.0 && occ <= 20.0))) {
                auto it = ix->buckets.find(T);
                if (it == ix->buckets.end()) {
                    DevBuf bb;
                    const size_t need = (((size_t)1 << (2 * T)) + 1) * 128;            // + the all-zero record behind the last code
                    size_t fr = 0, tot = 0;
                    if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr >= need + ((size_t)24 << 30) && bb.ensure(need) == GM_OK) {
                        const auto t0 = std::chrono::steady_clock::now();
                        // a failed build: the memory goes back and the empty entry is recorded, so that it is neither leaked nor tried again on every call
                        if (gmk_build_bucket(tb.as<uint2>(), ix->d_full.as<uint32_t>(), ix->dev.pac, bb.as<uint4>(), T, 0, st) != 0 || hipStreamSynchronize(st) != hipSuccess) {
                            bb.release();
                            ix->buckets.emplace(T, DevBuf());
                            gm_set_error("k-mer -> positions table construction failed");
                            return GM_E_HIP;
                        }
                        ix->hbm_bytes += bb.cap;
                        GM_TRACE("bucket table: %d-mers%s, %.1f GB, built in %.0f ms", T, "", need / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                    }
                    it = ix->buckets.emplace(T, bb).first;     // an empty entry = does not fit: not tried again
                }
                dp.bucket = it->second.as<uint4>(); dp.bucket_T = T; dp.bucket_ctx = 0;
            }
        }
    }
    dp.hcap = p->max_kmer_hits; dp.gap = p->gap; dp.align_score = p->align_score; dp.cutoff = p->cutoff;
    dp.S256 = d_tab;
    dp.lut = reinterpret_cast<const float2*>(d_tab + 1024);
    return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// index
// ------------------------------------------------------------------------------------------------
extern "C" int gm_index_build_on(const char* fasta_path, int where, int device_id) {
    if (!fasta_path || where < GM_BUILD_AUTO || where > GM_BUILD_DEVICE || device_id < 0) return GM_E_ARG;
    std::string err;
    int rc = gm_host_index_build(fasta_path, where, device_id, err);
    if (rc) gm_set_error(err);
    return rc;
}

extern "C" int gm_index_build(const char* fasta_path) { return gm_index_build_on(fasta_path, GM_BUILD_AUTO, 0); }

extern "C" int gm_index_open(const char* fasta_path, int device_id, int flags, gm_index** out) {
    if (!fasta_path || !out) return GM_E_ARG;
    *out = nullptr;
    std::string fa = fasta_path, err;
    if (!gm_host_index_files_exist(fa)) {
        if (!(flags & GM_INDEX_BUILD)) { gm_set_error("fail to locate the index files for " + fa); return GM_E_IO; }
        int rc = gm_host_index_build(fa, GM_BUILD_AUTO, device_id < 0 ? 0 : device_id, err);   // GenomeBwt::LoadGenome: "Could not find reference genome index! Building now."
        if (rc) { gm_set_error(err); return rc; }
    }
    gm_index* ix = new gm_index();
    int rc = gm_host_index_load(fa, ix->h, err);
    if (rc) { gm_set_error(err); delete ix; return rc; }
    if (ix->h.seq_len >= 0xFFFFFFFEull) {
        gm_set_error("reference too long for 32-bit ranks");
        delete ix;
        return GM_E_UNSUPPORTED;
    }
    ix->host_only = (flags & GM_INDEX_HOST_ONLY) != 0;
    if (ix->host_only) { *out = ix; return GM_OK; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) {
        gm_set_error("no usable HIP device (libgnumap_hip needs an MI355X / gfx950; there is no CPU fallback)");
        delete ix;
        return GM_E_NO_DEVICE;
    }
    ix->device = device_id;
    auto fail = [&](int code) { gm_index_close(ix); return code; };
    if (hipSetDevice(device_id) != hipSuccess) { gm_set_error("hipSetDevice failed"); return fail(GM_E_NO_DEVICE); }
    const GmHostIndex& h = ix->h;
    // stage the index into HBM once
    if (ix->d_bwt.ensure(h.bwt.size() * 4)) return fail(GM_E_NOMEM);
    if (hipMemcpy(ix->d_bwt.p, h.bwt.data(), h.bwt.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { gm_set_error("bwt upload failed"); return fail(GM_E_HIP); }
    std::vector<uint32_t> sa32(h.n_sa);
    for (uint64_t i = 0; i < h.n_sa; ++i) sa32[i] = (uint32_t)h.sa[i];      // sa[0] = (u64)-1 -> 0xFFFFFFFF
    if (ix->d_sa.ensure(sa32.size() * 4)) return fail(GM_E_NOMEM);
    if (hipMemcpy(ix->d_sa.p, sa32.data(), sa32.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { gm_set_error("sa upload failed"); return fail(GM_E_HIP); }
    if (ix->d_pac.ensure(h.pac.size())) return fail(GM_E_NOMEM);
    if (hipMemcpy(ix->d_pac.p, h.pac.data(), h.pac.size(), hipMemcpyHostToDevice) != hipSuccess) { gm_set_error("pac upload failed"); return fail(GM_E_HIP); }
    std::vector<uint32_t> coff(h.contigs.size() + 1);
    for (size_t i = 0; i < h.contigs.size(); ++i) coff[i] = (uint32_t)h.contigs[i].offset;
    coff[h.contigs.size()] = (uint32_t)h.l_pac;
    if (ix->d_contig.ensure(coff.size() * 4)) return fail(GM_E_NOMEM);
    if (hipMemcpy(ix->d_contig.p, coff.data(), coff.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { gm_set_error("contig upload failed"); return fail(GM_E_HIP); }
    GmDevIndex& d = ix->dev;
    d.bwt = ix->d_bwt.as<uint32_t>(); d.sa_samples = ix->d_sa.as<uint32_t>(); d.full_sa = nullptr;
    d.pac = ix->d_pac.as<uint8_t>(); d.contig_off = ix->d_contig.as<uint32_t>();
    d.seq_len = (uint32_t)h.seq_len; d.primary = (uint32_t)h.primary; d.l_pac = (uint32_t)h.l_pac; d.n_seqs = (uint32_t)h.contigs.size();
    d.sa_mask = h.sa_intv - 1; d.sa_shift = 0;
    while ((1u << d.sa_shift) < h.sa_intv) ++d.sa_shift;
    for (int i = 0; i < 5; ++i) d.L2[i] = (uint32_t)h.L2[i];
    {   // derive the bit-plane rank structure (one 16-byte load per rank query) from the reference-format BWT, once
        d.occ_nblk = (uint32_t)((h.seq_len + 95) / 96) + 1;
        if (ix->d_planes.ensure((size_t)4 * d.occ_nblk * 16)) return fail(GM_E_NOMEM);
        if (hipMemset(ix->d_planes.p, 0, (size_t)4 * d.occ_nblk * 16) != hipSuccess) { gm_set_error("plane memset failed"); return fail(GM_E_HIP); }
        int e = gmk_build_occ_planes(d, ix->d_planes.as<uint4>(), d.occ_nblk, nullptr);
        if (e || hipDeviceSynchronize() != hipSuccess) { gm_set_error("occ plane construction failed"); return fail(GM_E_HIP); }
        d.occ_planes = ix->d_planes.as<uint4>();
    }
    if (flags & GM_INDEX_FULL_SA) {
        if (ix->d_full.ensure(((size_t)h.seq_len + 1) * 4)) return fail(GM_E_NOMEM);
        int e = gmk_expand_full_sa(d, ix->d_full.as<uint32_t>(), nullptr);
        if (e || hipDeviceSynchronize() != hipSuccess) { gm_set_error("full SA expansion failed"); return fail(GM_E_HIP); }
        d.full_sa = ix->d_full.as<uint32_t>();
        ix->full_sa = true;
    }
    ix->hbm_bytes = ix->d_bwt.cap + ix->d_sa.cap + ix->d_pac.cap + ix->d_contig.cap + ix->d_full.cap + ix->d_planes.cap;
    *out = ix;
    return GM_OK;
}

// the tables the mapping kernels need for these parameters (memoised backward search of the seed, its compact form, the k-mer ->
// positions records) are built on the first gm_map_batch_device otherwise: a driver that wants its first block at full rate stages
// them with the index
extern "C" int gm_index_prepare(gm_index* ix, const gm_params* p) {
    if (!ix || !p || !p->finalized) return GM_E_ARG;
    if (ix->host_only) return GM_E_NO_DEVICE;
    HIPCHK(hipSetDevice(ix->device));
    GmDevParams dp;
    return sync_params(ix, p, dp, nullptr, true);
}

extern "C" void gm_index_close(gm_index* ix) {
    if (!ix) return;
    if (!ix->host_only && ix->device >= 0) {
        (void)hipSetDevice(ix->device);
        ix->d_bwt.release(); ix->d_sa.release(); ix->d_full.release(); ix->d_pac.release(); ix->d_contig.release();
        ix->d_cov.release(); ix->d_ptab.release(); ix->d_planes.release(); ix->d_nuc.release();
        for (auto& kv : ix->ptabs) kv.second.release();
        for (auto& kv : ix->kmer_tabs) kv.second.release();
        for (auto& kv : ix->kmer_ctabs) kv.second.release();
        for (auto& kv : ix->buckets) kv.second.release();
    }
    delete ix;
}

extern "C" int gm_index_get_info(const gm_index* ix, gm_index_info* o) {
    if (!ix || !o) return GM_E_ARG;
    o->l_pac = ix->h.l_pac; o->seq_len = ix->h.seq_len; o->primary = ix->h.primary; o->bwt_words = ix->h.bwt_words; o->n_sa = ix->h.n_sa;
    o->sa_intv = ix->h.sa_intv; o->n_seqs = (uint32_t)ix->h.contigs.size(); o->device_id = ix->device; o->full_sa = ix->full_sa;
    o->hbm_bytes = ix->hbm_bytes;
    return GM_OK;
}

extern "C" const char* gm_index_contig_name(const gm_index* ix, uint32_t i) {
    if (!ix || i >= ix->h.contigs.size()) return nullptr;
    return ix->h.contigs[i].name.c_str();
}

extern "C" uint64_t gm_index_contig_offset(const gm_index* ix, uint32_t i) {
    if (!ix) return 0;
    if (i >= ix->h.contigs.size()) return ix->h.l_pac;
    return ix->h.contigs[i].offset;
}

static uint32_t host_pos2rid(const GmHostIndex& h, uint64_t pos) {      // bns_pos2rid src/bntseq.c:349-363
    uint32_t lo = 0, hi = (uint32_t)h.contigs.size() - 1;
    while (lo < hi) {
        uint32_t mid = (lo + hi + 1) >> 1;
        if (pos >= h.contigs[mid].offset) lo = mid; else hi = mid - 1;
    }
    return lo;
}

static bool host_window(const GmHostIndex& h, uint64_t begin, uint32_t L, char* out) {   // GenomeBwt::GetString src/GenomeBwt.cpp:384-415
    if (L == 0 || begin >= h.l_pac || begin + L > h.l_pac) return false;
    if (host_pos2rid(h, begin) != host_pos2rid(h, begin + L - 1)) return false;
    for (uint32_t t = 0; t < L; ++t) {
        uint64_t p = begin + t;
        out[t] = "acgt"[(h.pac[p >> 2] >> ((~p & 3) << 1)) & 3];
    }
    return true;
}

extern "C" int gm_index_window(const gm_index* ix, uint64_t begin, uint32_t L, char* out) {
    if (!ix || !out) return 0;
    out[0] = 0;
    if (!host_window(ix->h, begin, L, out)) return 0;
    out[L] = 0;
    return (int)L;
}

// ------------------------------------------------------------------------------------------------
// batches
// ------------------------------------------------------------------------------------------------
extern "C" int gm_batch_create(gm_index* ix, uint32_t max_reads, uint32_t max_len, gm_batch** out) {
    if (!ix || !out || max_reads == 0 || max_len == 0) return GM_E_ARG;
    if (max_len > 2048) { gm_set_error("gm_batch_create: reads of at most 2048 bases (LDS budget of the seed and DP kernels)"); return GM_E_ARG; }
    // the vote kernels run one workgroup of 128 threads per read x strand and HIP refuses launches of 2^32 threads or more
    if (max_reads > 16000000u) { gm_set_error("gm_batch_create: at most 16 000 000 reads per batch (2 x reads x 128 threads per launch must stay below 2^32); use several batches"); return GM_E_ARG; }
    if (ix->host_only) { gm_set_error("index opened host-only"); return GM_E_NO_DEVICE; }
    HIPCHK(hipSetDevice(ix->device));
    gm_batch* b = new gm_batch();
    b->ix = ix; b->max_reads = max_reads; b->max_len = max_len;
    *out = b;
    return GM_OK;
}

extern "C" void gm_batch_destroy(gm_batch* b) {
    if (!b) return;
    if (b->svc.th.joinable()) {
        { std::lock_guard<std::mutex> lk(b->svc.mu); b->svc.quit = true; }
        b->svc.cv.notify_all();
        b->svc.th.join();
    }
    (void)hipSetDevice(b->ix->device);
    DevBuf* all[] = { &b->bases, &b->quals, &b->len, &b->status, &b->self_score, &b->min_score, &b->top_score, &b->seeds, &b->n_seeds,
                      &b->n_entries, &b->entry_off, &b->coords, &b->rs_overflow, &b->retry_list, &b->retry_off, &b->gtab_keys, &b->gtab_vals,
                      &b->cands, &b->fixed_cands, &b->fixed_cnt, &b->heavy_list, &b->heavy_off, &b->heavy_k0, &b->heavy_k1, &b->heavy_tmp, &b->hit_count, &b->hit_begin, &b->hit_cursor, &b->raw_hits, &b->counters, &b->small, &b->shards, &b->big_list, &b->tb_items, &b->tb_ops,
                      &b->tb_len, &b->band_moves, &b->pack,
                      &b->g_sorted, &b->g_ord, &b->g_lead, &b->g_krank, &b->g_khash, &b->g_nmatch, &b->g_mbegin, &b->g_multi, &b->g_big, &b->g_bigdone, &b->g_sk0, &b->g_sk1, &b->g_si0, &b->g_si1, &b->g_matches, &b->g_mhit, &b->g_positions,
                      &b->scan_tmp, &b->o_small, &b->o_posmatch, &b->o_post, &b->o_mapq, &b->o_emit, &b->o_reccnt, &b->o_cigcnt, &b->o_cigall, &b->o_recoff, &b->o_cigoff,
                      &b->o_recs, &b->o_pool, &b->o_codes, &b->snp_scratch, &b->snp_hmm, &b->pair_fb, &b->pair_list };
    for (DevBuf* d : all) d->release();
    PinBuf* pins[] = { &b->h_top, &b->h_hbegin, &b->h_ord, &b->h_post, &b->h_mapq, &b->h_emit, &b->h_mhit, &b->h_stat };
    for (PinBuf* d : pins) d->release();
    for (int i = 0; i < gm_batch::NS; ++i) { if (b->sub_streams[i]) (void)hipStreamDestroy(b->sub_streams[i]); b->sub_gk[i].release(); b->sub_gv[i].release(); }
    if (b->sub_ready) (void)hipEventDestroy(b->sub_ready);
    b->sub_counters.release(); b->sub_small.release(); b->sub_shards.release();
    for (auto& ev : b->pending) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto ev : b->pool) (void)hipEventDestroy(ev);
    delete b;
}

static int ensure_batch_buffers(gm_batch* b, const gm_params* p) {
    const size_t n = b->n, n2 = 2 * (size_t)b->n;
    uint32_t last = b->stride > (uint32_t)p->mer ? b->stride - (uint32_t)p->mer : 0;
    b->max_seeds = last / (uint32_t)p->jump + 2;
    int rc = 0;
    rc |= b->status.ensure(n); rc |= b->self_score.ensure(n * 4); rc |= b->min_score.ensure(n * 8); rc |= b->top_score.ensure(n * 4);
    rc |= b->seeds.ensure(n2 * b->max_seeds * sizeof(GmSeed)); rc |= b->n_seeds.ensure(n2 * 2); rc |= b->n_entries.ensure(n2 * 4);
    rc |= b->entry_off.ensure((n2 + 1) * 8); rc |= b->rs_overflow.ensure(n2); rc |= b->retry_list.ensure(n2 * 4);
    rc |= b->retry_off.ensure((n2 + 1024) * 8); rc |= b->big_list.ensure(n2 * 4 + 64);
    rc |= b->hit_count.ensure(n * 4); rc |= b->hit_begin.ensure((n + 1) * 8); rc |= b->hit_cursor.ensure(n * 4);
    rc |= b->counters.ensure(GMK_N * 8); rc |= b->small.ensure(64); rc |= b->shards.ensure((size_t)GM_NSHARD * GM_SHARD_STRIDE * 4);
    if (b->cand_cap < 16 * n + 64 * GM_NSHARD) { b->cand_cap = (uint32_t)std::min<size_t>(16 * n + 64 * GM_NSHARD, 0x7FFFFFFF); }
    rc |= b->cands.ensure((size_t)b->cand_cap * sizeof(GmCand));
    return rc ? GM_E_NOMEM : GM_OK;
}

static void fill_dev_batch(gm_batch* b) {
    GmDevBatch& d = b->dev;
    d.n = b->n; d.stride = b->stride; d.max_seeds = b->max_seeds; d.illumina_until = b->illumina_until; d.read_base = 0;
    d.bases = b->bases.as<uint8_t>(); d.quals = b->quals.as<uint8_t>(); d.len = b->len.as<uint16_t>();
    d.status = b->status.as<int8_t>(); d.self_score = b->self_score.as<float>(); d.min_score = b->min_score.as<double>();
    d.top_score = b->top_score.as<float>(); d.seeds = b->seeds.as<GmSeed>(); d.n_seeds = b->n_seeds.as<uint16_t>();
    d.n_entries = b->n_entries.as<uint32_t>(); d.entry_off = b->entry_off.as<uint64_t>(); d.coords = b->coords.as<uint32_t>();
    d.rs_overflow = b->rs_overflow.as<uint8_t>(); d.retry_list = b->retry_list.as<uint32_t>(); d.retry_off = b->retry_off.as<uint64_t>();
    d.gtab_keys = b->gtab_keys.as<uint32_t>(); d.gtab_vals = b->gtab_vals.as<uint32_t>();
    d.cands = b->cands.as<GmCand>(); d.cand_cap = b->cand_cap; d.cand_region = b->cand_cap / GM_NSHARD;
    d.shard_cnt = b->shards.as<uint32_t>();
    d.fixed_cands = b->use_fixed ? b->fixed_cands.as<GmCand>() : nullptr; d.fixed_cnt = b->use_fixed ? b->fixed_cnt.as<uint8_t>() : nullptr; d.fixed_epoch = b->use_fixed ? b->fixed_epoch : 0u;
    d.hit_count = b->hit_count.as<uint32_t>(); d.hit_begin = b->hit_begin.as<uint64_t>(); d.hit_cursor = b->hit_cursor.as<uint32_t>();
    d.raw_hits = b->raw_hits.as<GmRawHit>(); d.raw_cap = b->raw_cap;
    d.counters = b->counters.as<unsigned long long>();
    d.n_retry = b->small.as<uint32_t>() + 1; d.n_big = b->small.as<uint32_t>() + 2; d.big_list = b->big_list.as<uint32_t>();
    d.band_moves = b->band_moves.as<unsigned long long>(); d.band_moves_words = b->band_moves.cap / 8;
    d.pack = b->use_pack ? b->pack.as<uint32_t>() : nullptr; d.pack_w2 = gm_pack_w2(b->stride); d.pack_words = gm_pack_words(b->stride);
}

extern "C" int gm_batch_upload(gm_batch* b, const gm_params* p, const gm_reads* r, void* stream) {
    if (!b || !p || !r || !p->finalized) return GM_E_ARG;
    if (r->n > b->max_reads || r->stride > 2048) { gm_set_error("batch larger than gm_batch_create allowed, or reads longer than 2048 bases"); return GM_E_ARG; }
    if (r->stride % 8 != 0) { gm_set_error("gm_reads.stride must be a multiple of 8"); return GM_E_ARG; }
    HIPCHK(hipSetDevice(b->ix->device));
    hipStream_t st = S_(stream);
    b->n = r->n; b->stride = r->stride; b->mapped = false; b->cache_hits = b->cache_matches = 0; b->resume_ptr = nullptr; b->stamp = 0;
    size_t bytes = (size_t)r->n * r->stride;
    if (b->bases.ensure(bytes + 16) || b->quals.ensure(bytes + 16) || b->len.ensure((size_t)r->n * 2 + 16)) return GM_E_NOMEM;
    b->len_host.assign(r->len, r->len + r->n);
    b->len_max = 0; b->len_min = r->n ? 0xFFFFFFFFu : 0u;
    for (uint32_t i = 0; i < r->n; ++i) {
        if (r->len[i] > r->stride) { gm_set_error("read longer than stride"); return GM_E_ARG; }
        b->len_max = std::max<uint32_t>(b->len_max, r->len[i]);
        b->len_min = std::min<uint32_t>(b->len_min, r->len[i]);
    }
    // --illumina with automatic fallback (SeqReader.cpp:1171-1180): reads before the first one that shows a
    // quality below '@' keep Phred+64, that read and all later ones use Phred+33
    b->illumina_until = 0;
    if (p->illumina) {
        uint32_t until = r->n;
        for (uint32_t i = 0; i < r->n && until == r->n; ++i) {
            const uint8_t* q = r->quals + (size_t)i * r->stride;
            for (uint32_t t = 0; t < r->len[i]; ++t) if ((int8_t)q[t] < 64) { until = i; break; }      // `int Q = (int)fastq[i]` (SeqReader.cpp:1155): a signed char
        }
        b->illumina_until = until;
    }
    if (bytes) {
        HIPCHK(hipMemcpyAsync(b->bases.p, r->bases, bytes, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(b->quals.p, r->quals, bytes, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(b->len.p, r->len, (size_t)r->n * 2, hipMemcpyHostToDevice, st));
    }
    int rc = ensure_batch_buffers(b, p);
    if (rc) return rc;
    fill_dev_batch(b);
    return GM_OK;
}

// Large batches in full-SA mode: the batch is cut into sub-batches that go through prep -> seed -> vote -> NW on NS
// streams, so that the LDS-bound vote kernel of one sub-batch runs beside the seed / NW kernels of its neighbours.
// Returns 1 when a candidate shard overflowed (the caller then takes the single-pass path, which can grow the list).
static int map_pipelined(gm_index* ix, const gm_params* p, const GmDevParams& dp, gm_batch* b, hipStream_t st, int dense, int slots_hint, uint32_t sub_n) {
    const uint32_t n = b->n;
    const uint32_t nsb = (n + sub_n - 1) / sub_n;
    for (int i = 0; i < gm_batch::NS; ++i)
        if (!b->sub_streams[i]) HIPCHK(hipStreamCreateWithFlags(&b->sub_streams[i], hipStreamNonBlocking));
    if (!b->sub_ready) HIPCHK(hipEventCreateWithFlags(&b->sub_ready, hipEventDisableTiming));
    const size_t shard_bytes = (size_t)GM_NSHARD * GM_SHARD_STRIDE * 4;
    if (b->sub_counters.ensure((size_t)nsb * GMK_N * 8) || b->sub_small.ensure((size_t)nsb * 64) || b->sub_shards.ensure((size_t)nsb * shard_bytes)) return GM_E_NOMEM;
    fill_dev_batch(b);
    const uint32_t region = (b->cand_cap / nsb) / GM_NSHARD;          // per sub-batch, per shard
    if (region < 16) return 1;
    HIPCHK(hipEventRecord(b->sub_ready, st));                          // the upload / earlier work on the caller's stream
    std::vector<GmDevBatch> view(nsb);
    std::vector<uint32_t> ncand(nsb, 0);
    std::vector<uint32_t> shard_host((size_t)GM_NSHARD * GM_SHARD_STRIDE);
    bool overflow = false;
    auto make_view = [&](uint32_t i) {
        GmDevBatch v = b->dev;
        const uint32_t lo = i * sub_n, cnt = std::min(sub_n, n - lo);
        v.n = cnt; v.read_base = lo; v.fixed_cands = nullptr;
        v.illumina_until = b->illumina_until > lo ? std::min(b->illumina_until - lo, cnt) : 0;
        v.bases += (size_t)lo * b->stride; v.quals += (size_t)lo * b->stride; v.len += lo;
        v.status += lo; v.self_score += lo; v.min_score += lo; v.top_score += lo;
        v.seeds += (size_t)2 * lo * b->max_seeds; v.n_seeds += 2 * (size_t)lo; v.n_entries += 2 * (size_t)lo;
        v.rs_overflow += 2 * (size_t)lo; v.retry_list += 2 * (size_t)lo; v.retry_off += 2 * (size_t)lo;
        v.cands += (size_t)i * region * GM_NSHARD; v.cand_cap = region * GM_NSHARD; v.cand_region = region;
        v.shard_cnt = b->sub_shards.as<uint32_t>() + (size_t)i * GM_NSHARD * GM_SHARD_STRIDE;
        v.hit_count += lo; v.hit_begin += lo; v.hit_cursor += lo;
        v.counters = b->sub_counters.as<unsigned long long>() + (size_t)i * GMK_N;
        v.n_retry = b->sub_small.as<uint32_t>() + (size_t)i * 16 + 1; v.n_big = b->sub_small.as<uint32_t>() + (size_t)i * 16 + 2;
        v.big_list += 2 * (size_t)lo;
        return v;
    };
    auto finish = [&](uint32_t i) -> int {                            // vote(i) is done: size check, retry, then NW
        const int s = (int)(i % gm_batch::NS);
        hipStream_t ss = b->sub_streams[s];
        uint32_t small[2]; unsigned long long ctr[GMK_N];
        HIPCHK(hipMemcpyAsync(small, b->sub_small.as<uint32_t>() + (size_t)i * 16, 8, hipMemcpyDeviceToHost, ss));
        HIPCHK(hipMemcpyAsync(ctr, view[i].counters, sizeof ctr, hipMemcpyDeviceToHost, ss));
        HIPCHK(hipMemcpyAsync(shard_host.data(), view[i].shard_cnt, shard_bytes, hipMemcpyDeviceToHost, ss));
        HIPCHK(hipStreamSynchronize(ss));
        if (ctr[GMK_BAD_QUAL]) { gm_set_error("Invalid Fastq Character? (negative base probability)"); return GM_E_BAD_QUAL; }
        auto tally = [&](uint64_t& total, uint32_t& mx) { total = 0; mx = 0; for (int q = 0; q < GM_NSHARD; ++q) { uint32_t c = shard_host[(size_t)q * GM_SHARD_STRIDE]; total += c; mx = std::max(mx, c); } };
        uint64_t total; uint32_t mx;
        tally(total, mx);
        if (small[1] && mx <= region) {
            size_t slots = (size_t)ctr[GMK_HEAVY_SLOTS];
            if (b->sub_gk[s].ensure(slots * 4) || b->sub_gv[s].ensure(slots * 4)) return GM_E_NOMEM;
            view[i].gtab_keys = b->sub_gk[s].as<uint32_t>(); view[i].gtab_vals = b->sub_gv[s].as<uint32_t>();
            HIPCHK(hipMemsetAsync(b->sub_gk[s].p, 0xFF, slots * 4, ss));
            HIPCHK(hipMemsetAsync(b->sub_gv[s].p, 0, slots * 4, ss));
            { KTimer t(b, GM_K_VOTE_RETRY, ss); KCHK(gmk_vote_retry(ix->dev, dp, view[i], 1, 0, small[1], ss)); }
            HIPCHK(hipMemcpyAsync(shard_host.data(), view[i].shard_cnt, shard_bytes, hipMemcpyDeviceToHost, ss));
            HIPCHK(hipStreamSynchronize(ss));
            tally(total, mx);
        }
        if (mx > region) { overflow = true; return GM_OK; }
        ncand[i] = (uint32_t)total;
        { KTimer t(b, GM_K_NW, ss); KCHK(gmk_nw(ix->dev, dp, view[i], ncand[i], (b->len_min == b->len_max && !ctr[GMK_HIGH_QUAL]) ? b->len_max : 0u, ss)); }
        return GM_OK;
    };
    uint32_t next_finish = 0;
    for (uint32_t i = 0; i < nsb && !overflow; ++i) {
        view[i] = make_view(i);
        hipStream_t ss = b->sub_streams[i % gm_batch::NS];
        if (i < (uint32_t)gm_batch::NS) HIPCHK(hipStreamWaitEvent(ss, b->sub_ready, 0));
        HIPCHK(hipMemsetAsync(view[i].counters, 0, GMK_N * 8, ss));
        HIPCHK(hipMemsetAsync(b->sub_small.as<uint32_t>() + (size_t)i * 16, 0, 64, ss));
        HIPCHK(hipMemsetAsync(view[i].shard_cnt, 0, shard_bytes, ss));
        HIPCHK(hipMemsetAsync(view[i].rs_overflow, 0, 2 * (size_t)view[i].n, ss));
        { KTimer t(b, GM_K_PREP, ss); KCHK(gmk_prep(ix->dev, dp, view[i], ss)); }
        { KTimer t(b, GM_K_SEED, ss); KCHK(gmk_seed(ix->dev, dp, view[i], ss)); }
        { KTimer t(b, GM_K_VOTE, ss); KCHK(gmk_vote(ix->dev, dp, view[i], 1, dense, slots_hint, ss)); }
        if (i + 1 >= (uint32_t)gm_batch::NS) { int rc = finish(next_finish++); if (rc) return rc; }
    }
    while (next_finish < nsb && !overflow) { int rc = finish(next_finish++); if (rc) return rc; }
    for (int s = 0; s < gm_batch::NS; ++s) HIPCHK(hipStreamSynchronize(b->sub_streams[s]));
    if (overflow) return 1;
    // counters of all sub-batches, summed on the host
    std::vector<unsigned long long> all((size_t)nsb * GMK_N);
    HIPCHK(hipMemcpy(all.data(), b->sub_counters.p, all.size() * 8, hipMemcpyDeviceToHost));
    memset(b->counters_host, 0, sizeof b->counters_host);
    for (uint32_t i = 0; i < nsb; ++i) for (int q = 0; q < GMK_N; ++q) b->counters_host[q] += all[(size_t)i * GMK_N + q];
    b->counters_on_host = true;
    uint64_t total_c = 0;
    for (uint32_t i = 0; i < nsb; ++i) total_c += ncand[i];
    b->n_cands = (uint32_t)std::min<uint64_t>(total_c, 0xFFFFFFFFull);
    if (b->raw_cap < total_c + 16ull) b->raw_cap = total_c + 16ull;
    if (b->raw_hits.ensure(b->raw_cap * sizeof(GmRawHit))) return GM_E_NOMEM;
    fill_dev_batch(b);
    { KTimer t(b, GM_K_COMPACT, st); KCHK(gmk_scan_hits(b->dev, st)); }
    for (uint32_t i = 0; i < nsb; ++i) {
        view[i].raw_hits = b->dev.raw_hits; view[i].raw_cap = b->dev.raw_cap;
        uint32_t grid = std::max<uint32_t>(64, std::min<uint32_t>(2048, (ncand[i] + 255) / 256));
        KTimer t(b, GM_K_COMPACT, st);
        KCHK(gmk_scatter(view[i], grid, st));
    }
    b->mapped = true;
    return GM_OK;
}

extern "C" int gm_map_batch_device(gm_index* ix, const gm_params* p, gm_batch* b, void* stream) {
    if (!ix || !p || !b || !p->finalized || b->ix != ix) return GM_E_ARG;
    HIPCHK(hipSetDevice(ix->device));
    hipStream_t st = S_(stream);
    GmDevParams dp;
    int rc = sync_params(ix, p, dp, st, true);
    if (rc) return rc;
    b->mapped = false;
    if (b->n == 0) { b->n_cands = 0; b->n_raw = 0; b->mapped = true; memset(b->counters_host, 0, sizeof b->counters_host); return GM_OK; }
    const int use_full = ix->full_sa ? 1 : 0;
    // which vote kernel: expected SA hits per seed ~ reference length / 4^mer (capped by -h), expected seeds per strand from the
    // longest read.  0 = sparse (<= 16 hits per read x strand fit a 16-lane group), 1 = k_vote_slots (its hits fit 40 slots of
    // 64 lanes), 2 = its 64-slot form, 3 = k_vote_block (more: rounds of 2048 hits).  GM_VOTE=wave|block|big|rounds forces 0..3, GM_VOTE_KERNEL the dense form.
    int dense = 0, slots_hint = 40;                  // slots_hint: expected 64-lane slots per read x strand (picks the k_vote_slots form)
    {
        double per_seed = (double)ix->h.seq_len / pow(4.0, (double)std::min(p->mer, 31));
        if (p->max_kmer_hits > 0) per_seed = std::min(per_seed, (double)p->max_kmer_hits);
        const double L = (double)b->stride;
        double ns = L > p->mer ? floor((L - p->mer - 1) / std::max(1, p->jump)) + 1 : 1;
        if (p->nw && p->fast) ns = 1;
        ns = std::min(ns, (double)b->max_seeds);
        const double e_exp = ns * (1.0 + per_seed);                                      // the true locus + chance hits
        const double slots_exp = ns * ceil((per_seed + 3.0 * sqrt(per_seed) + 1.0) / 64.0);
        dense = e_exp > 14.0 ? (slots_exp <= 38.0 ? 1 : slots_exp <= 60.0 ? 2 : 3) : 0;
        slots_hint = (int)std::min(1000.0, slots_exp);
        // few hits in many short runs (long seeds on a large reference): one wave per read x strand, 16-rank groups (k_vote_tiny)
        const double groups_exp = ns * ceil((per_seed + 3.0 * sqrt(per_seed) + 1.0) / 16.0);
        if (dense == 1 && p->min_seed_hits >= 2 && e_exp + 4.0 * sqrt(e_exp) <= 230.0 && groups_exp <= 28.0) slots_hint = 0;
        else if (dense == 1 && p->min_seed_hits >= 2 && e_exp + 4.0 * sqrt(e_exp) <= 350.0 && groups_exp <= 56.0) slots_hint = -1;      // k_vote_tiny2
        if (const char* ev = gm_opt("GM_VOTE_SLOTS")) { if (*ev) slots_hint = atoi(ev); }
        if (const char* ev = gm_opt("GM_VOTE")) dense = !strcmp(ev, "block") ? 1 : !strcmp(ev, "big") ? 2 : !strcmp(ev, "rounds") ? 3 : !strcmp(ev, "wave") ? 0 : dense;
    }
    b->counters_on_host = false;
    {   // optional sub-batch pipeline over several streams for large full-SA batches (GM_PIPELINE=<sub-batch size>).  Measured
        // +4 % at configs[1]: three vote kernels end up sharing the LDS rather than hiding seed / NW work, and per-kernel
        // timings stop being attributable, so it is off by default.
        uint32_t sub_n = 0;
        sub_n = (uint32_t)gm_opt_ll("GM_PIPELINE", 0);
        if (use_full && sub_n >= 4096 && b->n >= 3 * (uint64_t)sub_n) {
            int prc = map_pipelined(ix, p, dp, b, st, dense, slots_hint, sub_n);
            if (prc <= 0) return prc;                    // done, or a real error
        }
    }
    // read x strands with very many SA hits (repeat seeds without -h) leave the ordinary vote kernels before they start: sorted-key
    // path of gm_heavy.hip, routed by the seed search's own hit count.  GM_HEAVY_MIN / GM_HEAVY_BUDGET (keys per chunk) are test switches.
    const uint32_t heavy_min = (uint32_t)gm_opt_ll("GM_HEAVY_MIN", 16384);
    const uint64_t heavy_budget = (uint64_t)gm_opt_ll("GM_HEAVY_BUDGET", 1ll << 27);
    dp.heavy_min = heavy_min;
    bool use_bucket = false, use_pair = false; uint32_t bucket_reg = 0;
    // seed lookup inside the vote kernels that take one read x strand per wave / workgroup (full SA, the k-mer table covering the
    // whole seed): no k_seed launch, no seed rows through HBM.  A k-mer that does not occur changes the positions of all later ones;
    // the wave then walks again round by round (gm_seed_rewalk_ool), one probe round trip per failing k-mer.  That stays rare while
    // even a k-mer with a sequencing error still occurs somewhere by chance - every k-mer expected >= 4 times in the reference
    // (measured on 100 Mbp: -m 12, 6 per k-mer: 21.0 ms fused against 27.7; -m 14, 0.4 per k-mer: an erroneous k-mer dies one or two
    // characters before its end, the walk creeps past the error in ~9 rounds, 207 against 29.5 ms - k_seed amortises those serial
    // steps over 64 lanes).  GM_SEED_FUSED=0 / 1: never / whenever possible.
    {
        const int fused_env = (int)gm_opt_ll("GM_SEED_FUSED", -1);
        const double occ = (double)ix->h.seq_len / pow(4.0, (double)std::min(p->mer, 31));
        // -h: on a real reference the k-mers of repeats exceed any cap, and each of them makes the walk slide base by base (:213-217) in
        // one lane - not measurable on the synthetic references of bench.py, so a capped run keeps k_seed unless forced
        const bool pays = occ >= 4.0 && p->max_kmer_hits == 0;
        dp.fused = (fused_env < 0 ? pays : fused_env != 0) && use_full && (dense == 1 || dense == 2) && !gm_opt("GM_VOTE_KERNEL") && b->max_seeds <= 64 && dp.kmer_tab &&
                   dp.kmer_ctab && dp.kmer_T == p->mer && p->mer <= 16 && p->jump >= 1 && !(dp.dbg & 128);
        // ... or in the bucket table (one read per wave, both strands; handles -h and k-mers that do not occur by walking again):
        // every seed is ONE random line
        const uint32_t lastmax = b->len_max > (uint32_t)p->mer ? b->len_max - (uint32_t)p->mer : 0;      // (the longest read of the block, not the row stride)
        const uint32_t max_reg = (lastmax + (uint32_t)p->jump - 1) / (uint32_t)p->jump;
        use_bucket = dp.bucket && use_full && dp.kmer_tab && dp.kmer_T == dp.bucket_T && dp.bucket_T == p->mer && p->min_seed_hits >= 2 && max_reg <= 32 && b->max_seeds <= 34 && !gm_opt("GM_VOTE_KERNEL") &&
                     !gm_opt("GM_VOTE") && gm_opt_ll("GM_PIPELINE", 0) == 0 && !(dp.dbg & 128) && fused_env != 0;
        if (use_bucket) { dp.fused = 1; bucket_reg = max_reg; } else dp.bucket = nullptr;
        // ... two reads per wavefront (gm_pair.hip) where a strand has at most 16 seeds; GM_VOTE_PAIR=0: one read per wavefront always
        use_pair = use_bucket && !dp.bucket_ctx && max_reg <= 16 && !(p->nw && p->fast) && !gm_opt_is("GM_VOTE_PAIR", "0") && !(dp.dbg & (64 | 256 | 512 | 1024 | 2048));       // (GM_DBG 4096 .. 32768: timing experiments of k_vote_pair)
        if (use_pair && (b->pair_fb.ensure((size_t)b->n + 64) || b->pair_list.ensure(((size_t)b->n + 16) * 4))) return GM_E_NOMEM;
        b->use_pack = dp.fused != 0;
        if (b->use_pack && b->pack.ensure((size_t)b->n * gm_pack_words(b->stride) * 4 + 64)) return GM_E_NOMEM;
    }
    // the one-wave vote kernels leave their candidates in per read x strand slots (no bump counter on the wave's critical path)
    {
        const bool fixed_ok = !gm_opt_is("GM_VOTE_FIXED", "0");
        b->use_fixed = fixed_ok && (use_bucket || ((dense == 1 || dense == 2) && !gm_opt("GM_VOTE_KERNEL")));      // k_vote_bucket, k_vote_tiny*, k_vote_slots (all forms)
        if (b->use_fixed) {
            const size_t cap_before = b->fixed_cands.cap;       // by CAPACITY: hipFree + a larger hipMalloc may hand back the same base address
            if (b->fixed_cands.ensure(((2 * (size_t)b->n + 63) / 64) * 64 * GM_FIXED_C * sizeof(GmCand)) || b->fixed_cnt.ensure(2 * (size_t)b->n + 64)) return GM_E_NOMEM;
            if (b->fixed_cands.cap != cap_before) { HIPCHK(hipMemsetAsync(b->fixed_cands.p, 0, b->fixed_cands.cap, st)); b->epoch_ctr = 0; }      // new memory: no stale launch stamps
        }
        b->fixed_epoch = 0;
    }
    {
        char buf[224];
        const int pp_env = (int)gm_opt_ll("GM_SLOTS_PIPE", -1);                                               // (gmk_vote's condition for the pipelined form of k_vote_slots)
        const bool slots_pp = use_full && !gm_opt("GM_VOTE_KERNEL") && (pp_env < 0 ? dense == 2 : pp_env != 0) && !(dp.dbg & 64);
        snprintf(buf, sizeof buf, "seeds=%s vote=%s locate=%s", use_bucket ? "bucket-table (in the vote kernel)" : dp.fused ? "k-mer table (in the vote kernel)" : "k_seed",
                 use_pair ? (bucket_reg <= 8 ? "k_vote_pair<4> + k_vote_bucket<2>" : bucket_reg <= 14 ? "k_vote_pair<7> + k_vote_bucket<4>" : "k_vote_pair<8> + k_vote_bucket<4>")
                 : use_bucket ? (bucket_reg <= 8 ? "k_vote_bucket<2>" : bucket_reg <= 16 ? "k_vote_bucket<4>" : bucket_reg <= 24 ? "k_vote_bucket<6>" : "k_vote_bucket<8>")
                            : dense == 0 ? "sparse" : dense == 3 ? "k_vote_block" : dense == 2 ? (slots_pp ? "k_vote_slots_pp<64>" : "k_vote_slots<64>") : slots_hint == 0 ? "k_vote_tiny" : slots_hint < 0 ? "k_vote_tiny2" : slots_pp ? "k_vote_slots_pp" : "k_vote_slots",
                 use_full ? "full-SA" : "sampled-SA");
        b->path = buf;
        GM_TRACE("path: %s", buf);
    }
    fill_dev_batch(b);
    HIPCHK(hipMemsetAsync(b->counters.p, 0, GMK_N * 8, st));
    { KTimer t(b, GM_K_PREP, st); KCHK(gmk_prep(ix->dev, dp, b->dev, st)); }
    if (!dp.fused) { KTimer t(b, GM_K_SEED, st); KCHK(gmk_seed(ix->dev, dp, b->dev, st)); }
    unsigned long long ctr[GMK_N];
    if (!use_full) {
        // faithful mode: locate every SA hit by LF walks into coords[] first (exact size from the seed kernel)
        HIPCHK(hipMemcpyAsync(ctr, b->counters.p, sizeof ctr, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (b->coords.ensure((size_t)ctr[GMK_SA_HITS] * 4 + 64)) return GM_E_NOMEM;
        fill_dev_batch(b);
        KCHK(gmk_scan_entries(b->dev, st));
        { KTimer t(b, GM_K_LOCATE, st); KCHK(gmk_locate_sampled(ix->dev, b->dev, ctr[GMK_SA_HITS], st)); }
    }
    std::vector<uint32_t> heavy;                     // {rs, n_seeds, SA hits} triples
    if (b->heavy_list.ensure(3 * 2 * (size_t)b->n * 4 + 64)) return GM_E_NOMEM;
    // the list of the heavy path: right after k_seed, or (fused form) after the first vote launch, which left them alone
    // status words of a phase come back in ONE short DMA into page-locked memory: [0..15] = b->small (n_retry, n_big, n_heavy, shard total,
    // shard maximum), [16..] = the work counters
    if (b->h_stat.ensure(64 + GMK_N * 8 + 64)) return GM_E_NOMEM;
    uint32_t* const hs_small = b->h_stat.as<uint32_t>();
    unsigned long long* const hs_ctr = reinterpret_cast<unsigned long long*>(b->h_stat.as<uint8_t>() + 64);
    auto collect_heavy_enqueue = [&]() -> int {      // (the count is read back with the phase's status words)
        HIPCHK(hipMemsetAsync(b->small.as<uint32_t>() + 3, 0, 4, st));
        KCHK(gmk_heavy_collect(b->dev, heavy_min, b->small.as<uint32_t>() + 3, b->heavy_list.as<uint32_t>(), dp.fused, st));
        return GM_OK;
    };
    auto fetch_heavy = [&](uint32_t nh) -> int {
        GM_TRACE("map_device: %u reads, seeds done, %u read x strands on the heavy path (> %u SA hits)", b->n, nh, heavy_min);
        heavy.clear();
        if (nh) {
            heavy.resize(3 * (size_t)nh);
            HIPCHK(hipMemcpy(heavy.data(), b->heavy_list.p, heavy.size() * 4, hipMemcpyDeviceToHost));
        }
        return GM_OK;
    };
    auto collect_heavy = [&]() -> int {              // the two-kernel form: right after k_seed
        int rc2 = collect_heavy_enqueue();
        if (rc2) return rc2;
        HIPCHK(hipMemcpyAsync(hs_small, b->small.p, 64, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return fetch_heavy(hs_small[3]);
    };
    if (!dp.fused) { rc = collect_heavy(); if (rc) return rc; }
    auto run_heavy = [&]() -> int {                  // expand + sort + run-length vote, chunk by chunk under the key budget
        const uint32_t nh = (uint32_t)(heavy.size() / 3);
        std::vector<unsigned long long> off;
        for (uint32_t j = 0; j < nh;) {
            const uint32_t j0 = j;
            unsigned long long acc = 0;
            off.clear();
            while (j < nh && j - j0 < 65536u) {
                const unsigned long long e = heavy[3 * (size_t)j + 2];
                if (e == 0xFFFFFFFFull || e > heavy_budget) { gm_set_error("a read x strand with more SA hits than one heavy-path chunk holds; use -h"); return GM_E_BATCH_TOO_LARGE; }
                if (acc + e > heavy_budget) break;
                off.push_back(acc); acc += e; ++j;
            }
            const uint32_t nj = j - j0;
            unsigned item_bits = 1; while ((1u << item_bits) < nj) ++item_bits;
            const size_t tmp_bytes = gmk_heavy_sort_temp_bytes((size_t)acc);
            if (b->heavy_k0.ensure((size_t)acc * 8 + 64) || b->heavy_k1.ensure((size_t)acc * 8 + 64) || b->heavy_tmp.ensure(tmp_bytes + 64) ||
                b->heavy_off.ensure((size_t)nj * 8 + 64)) return GM_E_NOMEM;
            HIPCHK(hipMemcpyAsync(b->heavy_off.p, off.data(), (size_t)nj * 8, hipMemcpyHostToDevice, st));
            KCHK(gmk_heavy_chunk(ix->dev, dp, b->dev, use_full, b->heavy_list.as<uint32_t>(), j0, nj, b->heavy_off.as<unsigned long long>(),
                                 b->heavy_k0.as<unsigned long long>(), b->heavy_k1.as<unsigned long long>(), acc, b->heavy_tmp.p, tmp_bytes, item_bits, st));
            HIPCHK(hipStreamSynchronize(st));        // off[] and the key buffers are reused by the next chunk
            GM_TRACE("heavy chunk: items %u..%u of %u, %llu keys sorted", j0, j, nh, acc);
        }
        return GM_OK;
    };
    // candidates per shard: total and maximum are reduced on the device (gmk_shard_stats -> small[4], small[5]) and come back with the status words
    auto read_status = [&](uint64_t& total, uint32_t& mx) -> int {
        KCHK(gmk_shard_stats(b->dev, b->small.as<uint32_t>() + 4, st));
        HIPCHK(hipMemcpyAsync(hs_small, b->small.p, 64, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(hs_ctr, b->counters.p, GMK_N * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        total = hs_small[4]; mx = hs_small[5];
        return GM_OK;
    };
    auto read_shards = read_status;
    for (int attempt = 0;; ++attempt) {
        HIPCHK(hipMemsetAsync(b->small.p, 0, 64, st));
        HIPCHK(hipMemsetAsync(b->shards.p, 0, (size_t)GM_NSHARD * GM_SHARD_STRIDE * 4, st));
        HIPCHK(hipMemsetAsync(b->rs_overflow.p, 0, 2 * (size_t)b->n, st));
        HIPCHK(hipMemsetAsync(b->counters.as<unsigned long long>() + GMK_HEAVY_SLOTS, 0, 8, st));
        HIPCHK(hipMemsetAsync(b->counters.as<unsigned long long>() + GMK_OVERFLOW_RS, 0, 8, st));
        if (b->use_fixed && use_bucket) {           // own-slot candidates carry the launch's stamp: nothing to zero but the flags "goes to the list kernel"
            HIPCHK(hipMemsetAsync(b->fixed_cnt.p, 0, 2 * (size_t)b->n + 32, st));
            if (++b->epoch_ctr == 0) { HIPCHK(hipMemsetAsync(b->fixed_cands.p, 0, b->fixed_cands.cap, st)); b->epoch_ctr = 1; }
            b->fixed_epoch = b->epoch_ctr;
            fill_dev_batch(b);
        } else if (b->use_fixed) HIPCHK(hipMemsetAsync(b->fixed_cnt.p, 0, 2 * (size_t)b->n, st));
        if (dp.fused) {                              // the seed search runs inside the vote launch: its work counters start over with it
            HIPCHK(hipMemsetAsync(b->counters.as<unsigned long long>() + GMK_KMERS, 0, 4 * 8, st));            // KMERS, OCC, SEEDS, SA_HITS
            HIPCHK(hipMemsetAsync(b->counters.as<unsigned long long>() + GMK_OCC_BLOCKS, 0, 2 * 8, st));       // OCC_BLOCKS, TAB_LOOKUPS
        }
        {
            KTimer t(b, GM_K_VOTE, st);
            if (use_bucket && use_pair) {
                // two reads per wavefront for the reads without complications; the flagged ones go through k_vote_bucket
                HIPCHK(hipMemsetAsync(b->pair_fb.p, 0, (size_t)b->n + 16, st));
                HIPCHK(hipMemsetAsync(b->pair_list.p, 0, 16, st));
                KCHK(gmk_vote_pair(ix->dev, dp, b->dev, bucket_reg, b->pair_fb.as<uint8_t>(), b->pair_list.as<uint32_t>() + 4, b->pair_list.as<uint32_t>(), st));
                KCHK(gmk_vote_bucket(ix->dev, dp, b->dev, bucket_reg, b->pair_list.as<uint32_t>() + 4, b->pair_list.as<uint32_t>(), st));
                KCHK(gmk_vote_list(ix->dev, dp, b->dev, use_full, st));
            } else if (use_bucket) { KCHK(gmk_vote_bucket(ix->dev, dp, b->dev, bucket_reg, nullptr, nullptr, st)); KCHK(gmk_vote_list(ix->dev, dp, b->dev, use_full, st)); }
            else KCHK(gmk_vote(ix->dev, dp, b->dev, use_full, dense, slots_hint, st));
            KCHK(gmk_cand_gather(b->dev, st));
        }
        if (dp.fused) { rc = collect_heavy_enqueue(); if (rc) return rc; }
        uint64_t total = 0; uint32_t mx = 0;
        rc = read_status(total, mx);                 // the one wait of the vote phase
        if (rc) return rc;
        memcpy(ctr, hs_ctr, sizeof ctr);
        const uint32_t small[2] = { hs_small[0], hs_small[1] };
        if (dp.fused) { rc = fetch_heavy(hs_small[3]); if (rc) return rc; }
        if (dp.dbg & 64) {                           // sampled phase clocks of the vote kernel (cycles of one lane per sampled workgroup)
            const double ns = (double)std::max<unsigned long long>(1, ctr[GMK_DBG8]);
            if (use_bucket) fprintf(stderr, "[gm_dbg] k_vote_bucket phases (mean clock ticks of lane 0 per sampled read, %llu samples; reads that leave early do not reach the later marks): forms + codes %.0f, "
                            "records %.0f, first walk %.0f, routing %.0f, filter pass %.0f, sweeps %.0f, store %.0f\n", ctr[GMK_DBG8], ctr[GMK_DBG0] / ns, ctr[GMK_DBG1] / ns, ctr[GMK_DBG2] / ns,
                            ctr[GMK_DBG3] / ns, ctr[GMK_DBG4] / ns, ctr[GMK_DBG5] / ns, ctr[GMK_DBG6] / ns);
            else
            fprintf(stderr, "[gm_dbg] vote phases (mean cycles per sampled read x strand, %llu samples): desc %.0f, loads+window %.0f, pass1 %.0f, pass2a %.0f, scan %.0f, "
                            "filter2 %.0f, table %.0f, emit %.0f\n", ctr[GMK_DBG8], ctr[GMK_DBG0] / ns, ctr[GMK_DBG1] / ns, ctr[GMK_DBG2] / ns, ctr[GMK_DBG3] / ns, ctr[GMK_DBG4] / ns,
                    ctr[GMK_DBG5] / ns, ctr[GMK_DBG6] / ns, ctr[GMK_DBG7] / ns);
        }
        if (ctr[GMK_BAD_QUAL]) { gm_set_error("Invalid Fastq Character? (negative base probability)"); return GM_E_BAD_QUAL; }
        uint32_t n_retry = small[1];
        if (n_retry && mx <= b->dev.cand_region) {
            size_t slots = (size_t)ctr[GMK_HEAVY_SLOTS];
            const size_t budget = (size_t)gm_opt_ll("GM_RETRY_BUDGET", 1ll << 28);   // table slots per launch (2 GB of keys + counts; the switch is for tests)
            if (slots <= budget) {
                if (b->gtab_keys.ensure(slots * 4) || b->gtab_vals.ensure(slots * 4)) return GM_E_NOMEM;
                fill_dev_batch(b);
                HIPCHK(hipMemsetAsync(b->gtab_keys.p, 0xFF, slots * 4, st));
                HIPCHK(hipMemsetAsync(b->gtab_vals.p, 0, slots * 4, st));
                { KTimer t(b, GM_K_VOTE_RETRY, st); KCHK(gmk_vote_retry(ix->dev, dp, b->dev, use_full, 0, n_retry, st)); }
            } else {
                // many read x strands at once (a kernel choice that did not fit the data): the tables are handed out again per
                // launch instead of all at once
                std::vector<uint32_t> list(n_retry), nent((size_t)2 * b->n);
                HIPCHK(hipMemcpyAsync(list.data(), b->retry_list.p, (size_t)n_retry * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(hipMemcpyAsync(nent.data(), b->n_entries.p, nent.size() * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                if (b->gtab_keys.ensure(budget * 4) || b->gtab_vals.ensure(budget * 4)) return GM_E_NOMEM;
                fill_dev_batch(b);
                std::vector<unsigned long long> off(n_retry);
                KTimer t(b, GM_K_VOTE_RETRY, st);
                for (uint32_t j = 0; j < n_retry;) {
                    const uint32_t j0 = j;
                    size_t acc = 0;
                    while (j < n_retry) {
                        size_t need = 2 * (size_t)nent[list[j]], sz = 1024;
                        while (sz < need && sz < 0x80000000u) sz <<= 1;
                        if (sz > budget) { gm_set_error("a read x strand with more SA hits than the retry table budget; use -h"); return GM_E_BATCH_TOO_LARGE; }
                        if (acc + sz > budget) break;
                        off[j] = acc; acc += sz; ++j;
                    }
                    HIPCHK(hipMemcpyAsync(b->retry_off.as<unsigned long long>() + j0, off.data() + j0, (size_t)(j - j0) * 8, hipMemcpyHostToDevice, st));
                    HIPCHK(hipMemsetAsync(b->gtab_keys.p, 0xFF, acc * 4, st));
                    HIPCHK(hipMemsetAsync(b->gtab_vals.p, 0, acc * 4, st));
                    KCHK(gmk_vote_retry(ix->dev, dp, b->dev, use_full, j0, j - j0, st));
                    HIPCHK(hipStreamSynchronize(st));        // off[] is reused by the next group
                }
            }
            rc = read_shards(total, mx);
            if (rc) return rc;
        }
        if (!heavy.empty() && mx <= b->dev.cand_region) {
            KTimer t(b, GM_K_VOTE_RETRY, st);
            rc = run_heavy();
            if (rc) return rc;
            rc = read_shards(total, mx);
            if (rc) return rc;
        }
        if (mx > b->dev.cand_region) {                  // a candidate shard overflowed: grow the list and vote again
            if (attempt > 8) { gm_set_error("candidate list keeps overflowing"); return GM_E_NOMEM; }
            size_t want = ((size_t)mx + mx / 4 + 64) * GM_NSHARD;
            if (want > 0x7FFFFFFFu) { gm_set_error("more than 2^31 candidates in one batch; map the block in smaller pieces (or use -h)"); return GM_E_BATCH_TOO_LARGE; }
            b->cand_cap = (uint32_t)want;
            if (b->cands.ensure((size_t)b->cand_cap * sizeof(GmCand))) return GM_E_NOMEM;
            fill_dev_batch(b);
            continue;
        }
        b->n_cands = (uint32_t)total;
        GM_TRACE("vote done: %llu candidates (attempt %d)", (unsigned long long)total, attempt);
        break;
    }
    if (b->raw_cap < b->n_cands + 16ull) b->raw_cap = b->n_cands + 16ull;
    if (b->raw_hits.ensure(b->raw_cap * sizeof(GmRawHit))) return GM_E_NOMEM;
    fill_dev_batch(b);
    // one read length in the block and no quality character above 127 (k_prep counted them): the DP kernel with the rows in DP order
    const uint32_t nw_rows_len = (b->len_min == b->len_max && ctr[GMK_HIGH_QUAL] == 0) ? b->len_max : 0u;
    b->path += std::string(" nw=") + gmk_nw_form(dp, b->dev, b->n_cands, nw_rows_len);
    { KTimer t(b, GM_K_NW, st); KCHK(gmk_nw(ix->dev, dp, b->dev, b->n_cands, nw_rows_len, st)); }
    { KTimer t(b, GM_K_COMPACT, st); KCHK(gmk_compact(b->dev, st)); }
    if (gm_trace_on()) { HIPCHK(hipStreamSynchronize(st)); GM_TRACE("NW + compaction done"); }
    b->mapped = true;
    return GM_OK;
}

extern "C" int gm_batch_counters(gm_batch* b, gm_counters* o) {
    if (!b || !o) return GM_E_ARG;
    HIPCHK(hipSetDevice(b->ix->device));
    unsigned long long c[GMK_N];
    if (b->counters_on_host) memcpy(c, b->counters_host, sizeof c);
    else HIPCHK(hipMemcpy(c, b->counters.p, sizeof c, hipMemcpyDeviceToHost));
    memset(o, 0, sizeof *o);
    o->reads = b->n; o->kmers_searched = c[GMK_KMERS]; o->occ_calls = c[GMK_OCC]; o->occ_blocks = c[GMK_OCC_BLOCKS];
    o->seeds_used = c[GMK_SEEDS]; o->sa_hits = c[GMK_SA_HITS];
    o->lf_steps = c[GMK_LF_STEPS]; o->candidates = b->n_cands; o->nw_cells = c[GMK_NW_CELLS]; o->accepted = c[GMK_ACCEPTED];
    o->vote_retries = c[GMK_OVERFLOW_RS]; o->table_lookups = c[GMK_TAB_LOOKUPS];
    return GM_OK;
}

extern "C" const char* gm_batch_path(gm_batch* b) { return b ? b->path.c_str() : ""; }

extern "C" const char* gm_kernel_name(int which) {
    static const char* names[GM_K_COUNT] = { "k_prep", "k_seed", "k_locate_sampled", "k_vote", "k_vote_retry", "k_nw", "k_compact(scan+scatter)" };
    return which >= 0 && which < GM_K_COUNT ? names[which] : "?";
}

extern "C" int gm_batch_set_profiling(gm_batch* b, int on) {
    if (!b) return GM_E_ARG;
    b->profiling = on != 0;
    return GM_OK;
}

extern "C" int gm_batch_kernel_times(gm_batch* b, double* ms, uint64_t* launches) {
    if (!b || !ms || !launches) return GM_E_ARG;
    HIPCHK(hipSetDevice(b->ix->device));
    for (auto& ev : b->pending) {
        HIPCHK(hipEventSynchronize(ev.b));
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, ev.a, ev.b));
        b->k_ms[ev.which] += t; b->k_launches[ev.which] += 1;
        b->pool.push_back(ev.a); b->pool.push_back(ev.b);
    }
    b->pending.clear();
    for (int i = 0; i < GM_K_COUNT; ++i) { ms[i] = b->k_ms[i]; launches[i] = b->k_launches[i]; b->k_ms[i] = 0; b->k_launches[i] = 0; }
    return GM_OK;
}

static bool raw_less(const GmRawHit& a, const GmRawHit& b) {
    if (a.strand != b.strand) return a.strand < b.strand;       // POS pass first (Driver.cpp:506-556)
    if (a.step != b.step) return a.step < b.step;               // seed order (align_sequence loop)
    return a.pos < b.pos;                                       // std::map<unsigned long,int> order (process_hits)
}

struct RawDownload {
    std::vector<GmRawHit> hits;
    std::vector<uint64_t> begin;
    std::vector<int8_t> status;
    std::vector<float> self_score, top;
};

static int download_raw(gm_batch* b, RawDownload& r, hipStream_t st, bool no_nw) {
    if (!b->mapped) { gm_set_error("batch has not been mapped"); return GM_E_ARG; }
    const uint32_t n = b->n;
    r.begin.assign(n + 1, 0); r.status.assign(n, 0); r.self_score.assign(n, 0); r.top.assign(n, 0);
    if (n == 0) return GM_OK;
    HIPCHK(hipMemcpyAsync(r.begin.data(), b->hit_begin.p, (n + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r.status.data(), b->status.p, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r.self_score.data(), b->self_score.p, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r.top.data(), b->top_score.p, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    uint64_t total = r.begin[n];
    if (total > b->raw_cap) { gm_set_error("internal: raw hit buffer too small"); return GM_E_CAPACITY; }
    r.hits.resize(total);
    if (total) HIPCHK(hipMemcpy(r.hits.data(), b->raw_hits.p, total * sizeof(GmRawHit), hipMemcpyDeviceToHost));
    b->n_raw = total;
    for (uint32_t i = 0; i < n; ++i) {
        auto first = r.hits.begin() + (ptrdiff_t)r.begin[i], last = r.hits.begin() + (ptrdiff_t)r.begin[i + 1];
        if (no_nw) std::sort(first, last, [](const GmRawHit& a, const GmRawHit& c) { return a.strand != c.strand ? a.strand < c.strand : a.pos < c.pos; });
        else std::sort(first, last, raw_less);
    }
    return GM_OK;
}

extern "C" int gm_batch_raw_hits(gm_batch* b, gm_raw_hit* out, uint64_t cap, uint64_t* n_out, int8_t* status, float* self_score, float* top_score) {
    if (!b || !n_out) return GM_E_ARG;
    HIPCHK(hipSetDevice(b->ix->device));
    RawDownload r;
    int rc = download_raw(b, r, nullptr, false);
    if (rc) return rc;
    *n_out = r.hits.size();
    if (status) memcpy(status, r.status.data(), b->n);
    if (self_score) memcpy(self_score, r.self_score.data(), (size_t)b->n * 4);
    if (top_score) memcpy(top_score, r.top.data(), (size_t)b->n * 4);
    if (r.hits.size() > cap) return GM_E_CAPACITY;
    static_assert(sizeof(gm_raw_hit) == sizeof(GmRawHit), "layout");
    if (out && !r.hits.empty()) memcpy(out, r.hits.data(), r.hits.size() * sizeof(GmRawHit));
    return GM_OK;
}

// host-side bookkeeping that is independent per item: cut into contiguous chunks, one host thread each (GM_HOST_THREADS; used by
// the track writers: threads made per call) ...
static unsigned host_threads() {
    static const unsigned n = [] {
        const char* e = getenv("GM_HOST_THREADS");               // sizes thread pools once per process
        unsigned v = e ? (unsigned)atoi(e) : std::min(16u, std::thread::hardware_concurrency());
        return std::max(1u, v);
    }();
    return n;
}
template <class F> static unsigned parallel_chunks(uint32_t n, uint32_t grain, unsigned want, F&& fn) {   // fn(chunk, lo, hi); returns #chunks
    unsigned T = (unsigned)std::min<uint64_t>(want, std::max<uint32_t>(1, n / std::max<uint32_t>(1, grain)));
    if (T <= 1) { fn(0u, 0u, n); return 1; }
    std::vector<std::thread> th;
    const uint64_t per = (n + T - 1) / T;
    for (unsigned c = 1; c < T; ++c)
        th.emplace_back([&, c] { uint32_t lo = (uint32_t)std::min<uint64_t>(n, c * per), hi = (uint32_t)std::min<uint64_t>(n, (c + 1) * per); fn(c, lo, hi); });
    fn(0u, 0u, (uint32_t)std::min<uint64_t>(n, per));
    for (auto& x : th) x.join();
    return T;
}

// ... and the two fp64 passes of the batch calls (denominator += exp(score); posterior / winner / MAPQ): every read's sum is its own,
// in its own order, so the READS of a block are cut into slices that helper threads of a process-wide pool take beside the calling
// thread (GM_PASS_THREADS helpers, default 3 per process; 0 = the calling thread alone).  The pool is shared by every batch of the
// process: a slice is a queue entry, callers of different batches interleave.
namespace {
struct PassPool {
    std::mutex mu; std::condition_variable cv;
    std::deque<std::function<void()>> q;
    std::vector<std::thread> th;
    bool stop = false;
    unsigned n = 0;
    void start(unsigned helpers) {
        n = helpers;
        for (unsigned i = 0; i < helpers; ++i)
            th.emplace_back([this] {
                for (;;) {
                    std::function<void()> job;
                    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; job = std::move(q.front()); q.pop_front(); }
                    job();
                }
            });
    }
    ~PassPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
};
PassPool& pass_pool() {
    static PassPool* pool = [] { auto* p = new PassPool(); const char* e = getenv("GM_PASS_THREADS"); p->start(e ? (unsigned)std::max(0, atoi(e)) : 3u); return p; }();
    return *pool;
}
// fn(lo, hi) over [0, n) in slices of at least `grain` items: the helpers take slices from the queue, the caller takes the rest
template <class F> void pass_parallel(uint32_t n, uint32_t grain, F&& fn) {
    PassPool& pool = pass_pool();
    const unsigned parts = (unsigned)std::min<uint64_t>(pool.n + 1, std::max<uint32_t>(1, n / std::max<uint32_t>(1, grain)));
    if (parts <= 1) { fn(0u, n); return; }
    const uint64_t per = (n + parts - 1) / parts;
    // completion state lives on the heap and is shared with every job: the last helper touches the mutex / condition variable AFTER the
    // count reaches zero, so they must outlive the caller's frame (the caller may already have seen zero and returned)
    struct Done { std::mutex mu; std::condition_variable cv; unsigned left; };
    auto done = std::make_shared<Done>();
    done->left = parts - 1;
    {
        std::lock_guard<std::mutex> lk(pool.mu);
        for (unsigned c = 1; c < parts; ++c)
            pool.q.emplace_back([&fn, done, n, per, c] {
                fn((uint32_t)std::min<uint64_t>(n, c * per), (uint32_t)std::min<uint64_t>(n, (c + 1) * per));
                std::lock_guard<std::mutex> lk2(done->mu);          // decrement UNDER the mutex: the waiter cannot leave before the notify
                if (--done->left == 0) done->cv.notify_one();
            });
    }
    pool.cv.notify_all();
    fn(0u, (uint32_t)std::min<uint64_t>(n, per));
    // a slice nobody has taken yet is run here rather than waited for
    for (;;) {
        std::function<void()> job;
        { std::lock_guard<std::mutex> lk(pool.mu); if (pool.q.empty()) break; job = std::move(pool.q.front()); pool.q.pop_front(); }
        job();
    }
    std::unique_lock<std::mutex> lk(done->mu);
    done->cv.wait(lk, [&] { return done->left == 0; });
}
}  // namespace

// host-only self test of the slice pool (no device): `callers` threads each run `iters` passes over [0, n) at once and check that
// every item was visited exactly once per pass.  Returns 0, or the number of passes that came out wrong.
extern "C" int gm_selftest_pass_parallel(uint32_t n, uint32_t grain, uint32_t callers, uint32_t iters) {
    std::atomic<int> bad{ 0 };
    auto run = [&] {
        std::vector<uint8_t> seen(n);
        for (uint32_t it = 0; it < iters; ++it) {
            std::fill(seen.begin(), seen.end(), (uint8_t)0);
            std::atomic<uint64_t> sum{ 0 };
            pass_parallel(n, grain, [&](uint32_t lo, uint32_t hi) { uint64_t s2 = 0; for (uint32_t i = lo; i < hi; ++i) { ++seen[i]; s2 += i; } sum += s2; });
            bool ok = sum.load() == (uint64_t)n * (n ? n - 1 : 0) / 2;
            for (uint32_t i = 0; i < n && ok; ++i) ok = seen[i] == 1;
            if (!ok) ++bad;
        }
    };
    std::vector<std::thread> th;
    for (uint32_t c = 1; c < callers; ++c) th.emplace_back(run);
    run();
    for (auto& t : th) t.join();
    return bad.load();
}

struct PhaseClock {                         // GM_TIMING=1: host-side phase times of the two batch calls on stderr
    bool on; const char* what; std::chrono::steady_clock::time_point t0; std::string line;
    explicit PhaseClock(const char* w) : on(gm_opt_ll("GM_TIMING", 0) != 0), what(w), t0(std::chrono::steady_clock::now()) {}
    void lap(const char* name) {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        char buf[64]; snprintf(buf, sizeof buf, " %s %.2f ms", name, std::chrono::duration<double, std::milli>(t1 - t0).count());
        line += buf; t0 = t1;
    }
    ~PhaseClock() { if (on) fprintf(stderr, "[gm_timing] %s:%s\n", what, line.c_str()); }
};

extern "C" int gm_stream_create(gm_index* ix, void** out) {
    if (!ix || !out || ix->host_only) return GM_E_ARG;
    HIPCHK(hipSetDevice(ix->device));
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = s;
    return GM_OK;
}
extern "C" void gm_stream_destroy(gm_index* ix, void* s) {
    if (!ix || !s) return;
    (void)hipSetDevice(ix->device);
    (void)hipStreamDestroy(S_(s));
}

extern "C" void* gm_host_alloc(size_t bytes) {
    void* q = nullptr;
    if (hipHostMalloc(&q, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) { gm_set_error("hipHostMalloc failed"); return nullptr; }
    return q;
}
extern "C" void gm_host_free(void* q) { if (q) (void)hipHostFree(q); }

static_assert(sizeof(gm_match) == sizeof(GmDevMatch) && offsetof(gm_match, first_pos) == offsetof(GmDevMatch, first_pos) &&
              offsetof(gm_match, first_strand) == offsetof(GmDevMatch, first_strand) && offsetof(gm_match, pos_begin) == offsetof(GmDevMatch, pos_begin) &&
              offsetof(gm_match, pos_end) == offsetof(GmDevMatch, pos_end), "gm_match layout");
static_assert(sizeof(gm_pos) == sizeof(GmDevPos) && offsetof(gm_pos, strand) == offsetof(GmDevPos, strand), "gm_pos layout");
static_assert(sizeof(gm_sam_rec) == sizeof(GmDevSamRec) && offsetof(gm_sam_rec, pos) == offsetof(GmDevSamRec, pos) &&
              offsetof(gm_sam_rec, contig) == offsetof(GmDevSamRec, contig) && offsetof(gm_sam_rec, chr_pos) == offsetof(GmDevSamRec, chr_pos) &&
              offsetof(gm_sam_rec, strand) == offsetof(GmDevSamRec, strand) && offsetof(gm_sam_rec, mapq) == offsetof(GmDevSamRec, mapq) &&
              offsetof(gm_sam_rec, a_score) == offsetof(GmDevSamRec, a_score) && offsetof(gm_sam_rec, post_prob) == offsetof(GmDevSamRec, post_prob) &&
              offsetof(gm_sam_rec, sim_matches) == offsetof(GmDevSamRec, sim_matches) && offsetof(gm_sam_rec, cigar_off) == offsetof(GmDevSamRec, cigar_off),
              "gm_sam_rec layout");

// ------------------------------------------------------------------------------------------------
// gm_map_batch = the block loop over set_top_matches (src/Driver.cpp:2344-2356).
// Device: prep -> seed -> locate+vote -> NW -> compaction (gm_map_batch_device), then process_hits' unique map as kernels
// (gm_output.hip: processing order, key grouping on the 2-bit reference, std::map order, -T / -u exits) writing gm_match /
// gm_pos records in HBM.  Host: ONE flat fp64 pass, denominator += exp(score) in the reference's order (glibc exp, the
// order-dependent part that has to stay bit-identical to the CPU program).
// ------------------------------------------------------------------------------------------------
extern "C" int gm_map_batch(gm_index* ix, const gm_params* p, gm_batch* b, const gm_reads* reads, gm_hits* out, void* stream) {
    if (!ix || !p || !b || !reads || !out) return GM_E_ARG;
    out->stamp = 0;                                            // every return path that does not leave a resident result leaves "none"
    PhaseClock pc("gm_map_batch");
    hipStream_t st = S_(stream);
    int rc;
    {   // test switch: behave as if blocks above this size outgrew a launch (exercises the callers' halving)
        const uint32_t test_max = (uint32_t)gm_opt_ll("GM_TEST_MAX_BLOCK", 0);
        if (test_max && reads->n > test_max) { gm_set_error("GM_TEST_MAX_BLOCK: block treated as too large"); return GM_E_BATCH_TOO_LARGE; }
    }
    // a call repeated with larger output buffers after GM_E_CAPACITY picks up where the first one stopped: the block is still mapped
    // and grouped in HBM (same reads pointer, same count), only the copies to the host are left to do
    const bool resume = b->resume_ptr != nullptr && b->resume_ptr == (const void*)reads->bases && b->n == reads->n;
    b->resume_ptr = nullptr;
    if (!resume) {
        rc = gm_batch_upload(b, p, reads, stream);
        if (rc) return rc;
        pc.lap("upload");
        rc = gm_map_batch_device(ix, p, b, stream);
        if (rc) return rc;
    } else HIPCHK(hipSetDevice(ix->device));
    const uint32_t n = b->n;
    out->n = n;
    if (n == 0) { out->match_begin[0] = 0; return GM_OK; }
    // the number of accepted hits stays on the device until the grouping kernels are done (no wait here): the workspace is sized by
    // its upper bound, the candidates of the block (resume: the count the first call read)
    const uint64_t hits_bound = resume ? b->n_raw : (uint64_t)b->n_cands;
    if (hits_bound > 0xFFFFFFF0ull) { gm_set_error("more than 2^32 accepted hits in one batch; map the block in smaller pieces"); return GM_E_BATCH_TOO_LARGE; }
    const size_t nh = (size_t)hits_bound + 16;
    if (b->g_sorted.ensure(nh * sizeof(GmRawHit)) || b->g_ord.ensure(nh * 4) || b->g_lead.ensure(nh * 4) || b->g_krank.ensure(nh * 4) ||
        b->g_khash.ensure(nh * 8) || b->g_positions.ensure(nh * sizeof(GmDevPos)) || b->g_matches.ensure(nh * sizeof(GmDevMatch)) || b->g_mhit.ensure(nh * 4) ||
        b->g_nmatch.ensure((size_t)n * 4) || b->g_mbegin.ensure(((size_t)n + 1) * 8) || b->g_multi.ensure((size_t)n * 4 + 64) ||
        b->g_big.ensure((size_t)n * 4 + 64) || b->g_bigdone.ensure((size_t)n + 64) || b->g_sk0.ensure(nh * 8) || b->g_sk1.ensure(nh * 8) || b->g_si0.ensure(nh * 4) || b->g_si1.ensure(nh * 4) ||
        b->scan_tmp.ensure(((size_t)n / 1024 + 8) * 8) || b->o_small.ensure(64)) return GM_E_NOMEM;
    GmDevGroup g;
    g.sorted = b->g_sorted.as<GmRawHit>(); g.ord_score = b->g_ord.as<float>(); g.lead = b->g_lead.as<uint32_t>(); g.krank = b->g_krank.as<uint32_t>();
    g.khash = b->g_khash.as<unsigned long long>(); g.n_match = b->g_nmatch.as<uint32_t>(); g.match_begin = b->g_mbegin.as<uint64_t>();
    g.multi_list = b->g_multi.as<uint32_t>(); g.n_multi = b->o_small.as<uint32_t>();
    g.big_list = b->g_big.as<uint32_t>(); g.n_big = b->o_small.as<uint32_t>() + 1; g.big_done = b->g_bigdone.as<uint8_t>();
    g.sk0 = b->g_sk0.as<unsigned long long>(); g.sk1 = b->g_sk1.as<unsigned long long>(); g.si0 = b->g_si0.as<uint32_t>(); g.si1 = b->g_si1.as<uint32_t>();
    g.matches = b->g_matches.as<GmDevMatch>(); g.match_hit = b->g_mhit.as<uint32_t>(); g.positions = b->g_positions.as<GmDevPos>();
    b->cache_hits = b->cache_matches = 0; b->stamp = 0; out->stamp = 0;
    if (!resume) {
        HIPCHK(hipMemsetAsync(b->o_small.p, 0, 64, st));
        if (p->unique_only && !p->nw) HIPCHK(hipMemsetAsync(b->g_positions.p, 0, nh * sizeof(GmDevPos), st));      // dropped hits leave holes
        KCHK(gmk_group_count(ix->dev, b->dev, g, p->nw, p->unique_only, p->max_matches, st));
        KCHK(gmk_scan_u32(g.n_match, n, g.match_begin, b->scan_tmp.as<unsigned long long>(), st));
        KCHK(gmk_group_write(b->dev, g, st));
    }
    // what the host pass needs: per-read status / top score, the CSR of the hits and their scores in processing order
    if (b->h_top.ensure((size_t)n * 4) || b->h_hbegin.ensure(((size_t)n + 1) * 8) || b->h_ord.ensure(nh * 4)) return GM_E_NOMEM;
    if (b->h_stat.ensure(64 + GMK_N * 8 + 64)) return GM_E_NOMEM;
    uint64_t* const hs64 = b->h_stat.as<uint64_t>();        // [0] matches, [1] accepted hits: page-locked, with the rest of the phase's copies
    HIPCHK(hipMemcpyAsync(&hs64[0], g.match_begin + n, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&hs64[1], b->hit_begin.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out->status, b->status.p, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out->self_score, b->self_score.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(b->h_top.p, b->top_score.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(b->h_hbegin.p, b->hit_begin.p, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, st));
    if (hits_bound) HIPCHK(hipMemcpyAsync(b->h_ord.p, g.ord_score, (size_t)hits_bound * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out->match_begin, g.match_begin, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint64_t n_m = hs64[0], n_hits = hs64[1];
    if (n_hits > hits_bound) { gm_set_error("internal: more accepted hits than candidates"); return GM_E_HIP; }
    b->n_raw = n_hits;
    pc.lap("group");
    if (gm_trace_on()) {
        uint32_t cnt[16] = { 0 };
        HIPCHK(hipMemcpy(cnt, b->o_small.p, sizeof cnt, hipMemcpyDeviceToHost));
        GM_TRACE("grouping done: %llu accepted hits -> %llu matches; reads: %u all-pairs list (incl. handed back), %u big path; handed back: %u (-u --no_nw), %u (> set limit), %u (hash collision)",
                 (unsigned long long)n_hits, (unsigned long long)n_m, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4]);
    }
    if (n_m > out->matches_cap || n_hits > out->positions_cap) {
        out->matches_cap = n_m; out->positions_cap = n_hits;
        gm_set_error("output buffers too small");
        b->resume_ptr = (const void*)reads->bases;               // the repeated call only copies
        return GM_E_CAPACITY;
    }
    if (n_m) HIPCHK(hipMemcpyAsync(out->matches, g.matches, (size_t)n_m * sizeof(gm_match), hipMemcpyDeviceToHost, st));
    if (n_hits) HIPCHK(hipMemcpyAsync(out->positions, g.positions, (size_t)n_hits * sizeof(gm_pos), hipMemcpyDeviceToHost, st));
    if (b->h_mhit.ensure((size_t)n_m * 4 + 16)) return GM_E_NOMEM;
    if (n_m) HIPCHK(hipMemcpyAsync(b->h_mhit.p, g.match_hit, (size_t)n_m * 4, hipMemcpyDeviceToHost, st));
    // the fp64 pass (process_hits :134-165: denominator += exp(align_score) for every new key and every new place of an old key, in
    // processing order) runs while the records are still in flight
    const float* top = b->h_top.as<float>(); const uint64_t* hb = b->h_hbegin.as<uint64_t>(); const float* ord = b->h_ord.as<float>();
    b->h_exp.resize((size_t)n_hits + 1);
    double* hexp = b->h_exp.data();
    pass_parallel(n, 16384, [&](uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; ++i) {
            const int8_t s = out->status[i];
            double den = 0.0, tp = 0.0;
            if (s == GM_READ_OK) {
                for (uint64_t h = hb[i]; h < hb[i + 1]; ++h) {
                    const float sc = ord[h];
                    const double e = sc != -INFINITY ? exp((double)sc) : 0.0;
                    hexp[h] = e;
                    if (sc != -INFINITY) den += e;
                }
                tp = (double)top[i];
            } else if (s == GM_READ_TOO_SHORT) tp = -2.0;
            else if (s == GM_READ_TOO_POOR) tp = -3.0;
            else if (s == GM_READ_TOO_MANY) tp = 999999.0;
            out->denominator[i] = den; out->top_score[i] = tp;
        }
    });
    pc.lap("exp");
    HIPCHK(hipStreamSynchronize(st));
    b->cache_hits = n_hits; b->cache_matches = n_m;          // h_exp / h_ord / h_mhit describe this result
    {
        // a stamp a caller's uninitialised memory does not hit by accident: per-process random base (clock + address entropy through
        // the 64-bit mixer) + counter, with the batch's address mixed in; never 0
        static const uint64_t stamp_base = [] {
            uint64_t x = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((uint64_t)(uintptr_t)&gm_set_option << 17) ^ ((uint64_t)getpid() << 48);
            x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x | (1ull << 63);
        }();
        static std::atomic<uint64_t> next_stamp{ 1 };
        b->stamp = (stamp_base + next_stamp.fetch_add(1) * 0x9e3779b97f4a7c15ull) ^ ((uint64_t)(uintptr_t)b >> 4);
        if (b->stamp == 0) b->stamp = stamp_base;
        out->stamp = b->stamp;
    }
    pc.lap("records");
    return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// gm_output_batch = the block loop over create_match_output (src/Driver.cpp:2360-2373).
// Host: ONE flat fp64 pass over the matches (posterior = exp(score) / denominator, winner = first strict maximum in key order,
// MAPQ = round(-10 log10(1 - p)): glibc exp / log / round).  Device: traceback of every kept sequence, run-length CIGAR text, SAM
// rows, coverage deposit (+ the per-nucleotide track of -b / -d).  Nothing per read is done on the host beyond that pass.
// ------------------------------------------------------------------------------------------------
extern "C" int gm_output_batch(gm_index* ix, const gm_params* p, gm_batch* b, const gm_reads* reads, const gm_hits* hits, gm_sam_out* out, void* stream) {
    if (!ix || !p || !b || !reads || !hits || !out) return GM_E_ARG;
    if (hits->n != b->n || reads->n != b->n) { gm_set_error("hits / reads do not belong to the batch"); return GM_E_ARG; }
    PhaseClock pc("gm_output_batch");
    HIPCHK(hipSetDevice(ix->device));
    hipStream_t st = S_(stream);
    GmDevParams dp;
    int rc = sync_params(ix, p, dp, st);
    if (rc) return rc;
    const uint32_t n = hits->n;
    const uint64_t n_m64 = n ? hits->match_begin[n] : 0;
    out->n_recs = 0; out->cigar_len = 0;
    if (n_m64 == 0) return GM_OK;
    if (n_m64 > 0x7FFFFFFFull) { gm_set_error("too many matches in one batch; map the block in smaller pieces"); return GM_E_BATCH_TOO_LARGE; }
    const uint32_t n_m = (uint32_t)n_m64;
    // ---- device, part 1: everything that does not depend on the posteriors is enqueued BEFORE the host pass and runs under it ----
    // hits->stamp names this batch's resident result: its matches / positions are used where they are.  Otherwise `hits` may have been
    // edited by the caller: everything the kernels index with is checked HERE, before the first enqueue (a match of another read
    // would make k_traceback read rows that do not exist, a position range beyond the buffer would make k_out_items write past
    // o_posmatch), and the records are uploaded again.
    const bool resident = hits->stamp != 0 && hits->stamp == b->stamp && b->cache_matches == n_m64 && b->cache_hits <= hits->positions_cap;
    uint64_t n_p = 0;
    if (resident) n_p = b->cache_hits;
    else {
        if (hits->match_begin[0] != 0) { gm_set_error("gm_hits: match_begin[0] must be 0"); return GM_E_ARG; }
        bool bad = false;
        for (uint32_t i = 0; i < n && !bad; ++i) {
            const uint64_t m0 = hits->match_begin[i], m1 = hits->match_begin[i + 1];
            if (m1 < m0 || m1 > n_m64) { bad = true; break; }
            for (uint64_t m = m0; m < m1; ++m) {
                const gm_match& mm = hits->matches[m];
                if (mm.read != i || mm.pos_end < mm.pos_begin || mm.pos_end > hits->positions_cap || mm.first_strand > 1 || mm.first_pos >= ix->h.l_pac) { bad = true; break; }
                for (uint32_t q = mm.pos_begin; q < mm.pos_end; ++q) if (hits->positions[q].strand > 1 || hits->positions[q].pos >= ix->h.l_pac) bad = true;   // the places of THIS match (slots no match points at are never read)
                n_p = std::max<uint64_t>(n_p, mm.pos_end);
            }
        }
        if (bad) { gm_set_error("gm_hits: a match does not belong to its read, or its positions / first position are out of range"); return GM_E_ARG; }
        if (b->cache_matches == n_m64 && b->cache_hits <= hits->positions_cap && b->cache_hits >= n_p) n_p = b->cache_hits;     // positions share the hit CSR
    }
    const uint32_t ops_words = gm_ops_words(b->stride), codes_stride = 32u * ops_words;
    const bool snp = p->mode == GM_MODE_SNP;
    if (snp && !ix->nuc_on) { gm_set_error("GM_MODE_SNP deposits into the per-nucleotide tracks: call gm_coverage_enable_nuc first"); return GM_E_ARG; }
    const bool nuc = p->mode != GM_MODE_NORMAL && !snp && ix->nuc_on;
    if (b->g_matches.ensure((size_t)n_m * sizeof(GmDevMatch)) || b->g_positions.ensure((size_t)(n_p + 1) * sizeof(GmDevPos)) ||
        b->o_posmatch.ensure((size_t)(n_p + 1) * 4) || b->o_post.ensure((size_t)n_m * 4) || b->o_mapq.ensure((size_t)n_m * 4) || b->o_emit.ensure(n_m) ||
        b->tb_items.ensure((size_t)n_m * sizeof(GmCand)) || b->tb_ops.ensure((size_t)n_m * ops_words * 8) || b->tb_len.ensure((size_t)n_m * 2) ||
        b->o_reccnt.ensure((size_t)n_m * 4) || b->o_cigcnt.ensure((size_t)n_m * 4) || b->o_cigall.ensure((size_t)n_m * 4) || b->o_recoff.ensure(((size_t)n_m + 1) * 8) ||
        b->o_cigoff.ensure(((size_t)n_m + 1) * 8) || b->scan_tmp.ensure(((size_t)std::max<uint32_t>(n_m, n) / 1024 + 8) * 8) || b->o_small.ensure(64) ||
        (nuc && b->o_codes.ensure((size_t)n_m * codes_stride))) return GM_E_NOMEM;
    const GmDevMatch* d_m = b->g_matches.as<GmDevMatch>(); const GmDevPos* d_p = b->g_positions.as<GmDevPos>();
    if (!resident) {
        HIPCHK(hipMemcpyAsync(b->g_matches.p, hits->matches, (size_t)n_m * sizeof(gm_match), hipMemcpyHostToDevice, st));
        if (n_p) HIPCHK(hipMemcpyAsync(b->g_positions.p, hits->positions, (size_t)n_p * sizeof(gm_pos), hipMemcpyHostToDevice, st));
        b->stamp = 0;                                          // what is resident now is the caller's version
    }
    HIPCHK(hipMemsetAsync(b->o_posmatch.p, 0xFF, (size_t)(n_p + 1) * 4, st));
    HIPCHK(hipMemsetAsync(b->o_small.p, 0, 64, st));
    if (p->max_gap != 3 && b->band_moves.ensure(gm_band_moves_words(n_m, b->stride) * 8)) return GM_E_NOMEM;
    fill_dev_batch(b);
    KCHK(gmk_out_items(d_m, n_m, 0, b->tb_items.as<GmCand>(), b->o_posmatch.as<uint32_t>(), st));
    // one traceback per ScoredSeq, oriented by its first strand (NormalScoredSeq::score, ScoredSeq::get_SAM); the kernel also leaves the
    // length of every sequence's CIGAR text and finds the longest aligned length
    KCHK(gmk_traceback(ix->dev, dp, b->dev, b->tb_items.as<GmCand>(), n_m, b->tb_ops.as<unsigned long long>(), ops_words, b->tb_len.as<uint16_t>(),
                       nullptr, b->o_cigall.as<uint32_t>(), b->o_small.as<uint32_t>(), st));
    pc.lap("enqueue");
    // ---- host pass: ScoredSeq::get_SAM :300-309, is_greater :223-228, Driver.cpp:672-701 ----
    if (b->h_post.ensure((size_t)n_m * 4) || b->h_mapq.ensure((size_t)n_m * 4) || b->h_emit.ensure(n_m)) return GM_E_NOMEM;
    float* post = b->h_post.as<float>(); int32_t* mapq = b->h_mapq.as<int32_t>(); uint8_t* emit = b->h_emit.as<uint8_t>();
    const double log10v = log(10), e_m1 = exp(-1.0);                   // e_m1: the empty NormalScoredSeq a winner has to beat, ScoredSeq.h:117-120
    auto mapq_of = [&](double total) {
        int q;
        if (total == 1) q = 30; else q = (int)round(-10 * log(1 - total) / log10v);
        return q > 30 ? 30 : q;
    };
    const int all = p->print_all_sam;
    const uint64_t cached_m = b->cache_matches, cached_h = b->cache_hits;
    const uint32_t* mhit = b->h_mhit.as<uint32_t>(); const float* ord = b->h_ord.as<float>(); const double* hexp = b->h_exp.data();
    pass_parallel(n, 16384, [&](uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; ++i) {
            const uint64_t m0 = hits->match_begin[i], m1 = hits->match_begin[i + 1];
            if (m0 == m1) continue;
            if (hits->status[i] != GM_READ_OK) { for (uint64_t m = m0; m < m1; ++m) { post[m] = 0; mapq[m] = 0; emit[m] = 0; } continue; }
            const double den = hits->denominator[i], top = hits->top_score[i];
            int64_t best = -1;
            double best_log = e_m1, best_total = 0;
            for (uint64_t m = m0; m < m1; ++m) {
                const gm_match& mm = hits->matches[m];
                // exp(align_score): the value gm_map_batch already computed for the hit that gave this match its score, when the caller has
                // left the match as it was (exp is a function of the score alone, so equal score bits are all that has to hold)
                double lg;
                if (m < cached_m && mhit[m] < cached_h && memcmp(&ord[mhit[m]], &mm.score, 4) == 0) lg = hexp[mhit[m]];
                else lg = exp((double)mm.score);
                const double total = lg / den;                             // ScoredSeq.h:300
                post[m] = (float)total;                                    // AddScore(const float& amt), NormalScoredSeq.cpp:70
                emit[m] = (uint8_t)all;
                mapq[m] = all ? mapq_of(total) : 0;
                if (lg > best_log) { best = (int64_t)m; best_log = lg; best_total = total; }   // is_greater: strict, first in key order wins
            }
            if (!all && best >= 0 && (double)hits->matches[best].score > top - 0.00001) {      // Driver.cpp:695
                emit[best] = 1;
                mapq[best] = mapq_of(best_total);
            }
        }
    });
    pc.lap("fp64");
    // ---- device, part 2 ----
    HIPCHK(hipMemcpyAsync(b->o_post.p, post, (size_t)n_m * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(b->o_mapq.p, mapq, (size_t)n_m * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(b->o_emit.p, emit, n_m, hipMemcpyHostToDevice, st));
    KCHK(gmk_out_sizes(d_m, n_m, b->o_emit.as<uint8_t>(), b->o_cigall.as<uint32_t>(), b->o_reccnt.as<uint32_t>(), b->o_cigcnt.as<uint32_t>(), st));
    KCHK(gmk_scan_u32(b->o_reccnt.as<uint32_t>(), n_m, b->o_recoff.as<uint64_t>(), b->scan_tmp.as<unsigned long long>(), st));
    KCHK(gmk_scan_u32(b->o_cigcnt.as<uint32_t>(), n_m, b->o_cigoff.as<uint64_t>(), b->scan_tmp.as<unsigned long long>(), st));
    uint64_t n_recs = 0, cig_len = 0; uint32_t max_span = 0;
    HIPCHK(hipMemcpyAsync(&n_recs, b->o_recoff.as<uint64_t>() + n_m, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&cig_len, b->o_cigoff.as<uint64_t>() + n_m, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&max_span, b->o_small.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    pc.lap("traceback+count");
    // capacity first: a call that is going to be repeated with larger buffers must not deposit coverage twice
    out->n_recs = n_recs; out->cigar_len = cig_len;
    if (n_recs > out->recs_cap || cig_len > out->cigar_cap) {
        out->recs_cap = n_recs; out->cigar_cap = cig_len;
        gm_set_error("output buffers too small");
        return GM_E_CAPACITY;
    }
    if (cig_len > 0xFFFFFFFFull) { gm_set_error("CIGAR pool beyond 4 GB in one batch; map the block in smaller pieces"); return GM_E_BATCH_TOO_LARGE; }
    if (b->o_recs.ensure((size_t)(n_recs + 1) * sizeof(GmDevSamRec)) || b->o_pool.ensure((size_t)cig_len + 16)) return GM_E_NOMEM;
    if (n_recs) {
        KCHK(gmk_out_write(ix->dev, b->dev, d_m, d_p, n_m, b->o_emit.as<uint8_t>(), b->o_mapq.as<int32_t>(), b->o_post.as<float>(), b->tb_ops.as<unsigned long long>(),
                           ops_words, b->tb_len.as<uint16_t>(), p->nw, b->o_recoff.as<uint64_t>(), b->o_cigoff.as<uint64_t>(),
                           b->o_recs.as<GmDevSamRec>(), b->o_pool.as<char>(), st));
        HIPCHK(hipMemcpyAsync(out->recs, b->o_recs.p, (size_t)n_recs * sizeof(gm_sam_rec), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(out->cigar_pool, b->o_pool.p, (size_t)cig_len, hipMemcpyDeviceToHost, st));
    }
    if (snp && ix->cov_bins && n_p) {
        // SNPScoredSeq::score: no traceback in the deposit - the pair HMM of every kept sequence against its window, chunk by chunk (a
        // lane's forward matrix is (L+1)^2 x 3 doubles of scratch), then coverage + the five tracks at every place of the sequence
        const uint32_t Lmax = b->stride;
        const uint32_t chunk = (uint32_t)std::min<long long>(std::max<long long>(64, gm_opt_ll("GM_SNP_CHUNK", 16384)), n_m) / 64u * 64u + 64u;
        const size_t cells = gmk_pair_hmm_cells(Lmax);
        if (b->snp_scratch.ensure((size_t)((chunk + 63) / 64) * cells * 8) || b->snp_hmm.ensure((size_t)chunk * Lmax * 5 * 4)) return GM_E_NOMEM;
        for (uint32_t m0 = 0; m0 < n_m; m0 += chunk) {
            const uint32_t cnt = std::min<uint32_t>(chunk, n_m - m0);
            KCHK(gmk_pair_hmm(ix->dev, dp, b->dev, b->tb_items.as<GmCand>() + m0, cnt, b->snp_scratch.as<double>(), Lmax, b->snp_hmm.as<float>(), st));
            KCHK(gmk_snp_deposit(ix->d_cov.as<float>(), ix->d_nuc.as<float>(), ix->cov_bins, ix->cov_bin_size, b->dev, d_m, d_p, m0, cnt, b->o_post.as<float>(),
                                 b->snp_hmm.as<float>(), Lmax, st));
        }
    } else if (ix->cov_bins && n_p && max_span) {
        if (nuc) KCHK(gmk_out_codes(b->dev, dp, d_m, n_m, b->tb_ops.as<unsigned long long>(), ops_words, b->tb_len.as<uint16_t>(), b->o_codes.as<uint8_t>(), codes_stride, st));
        KCHK(gmk_out_deposit(ix->d_cov.as<float>(), ix->cov_bins, ix->cov_bin_size, d_m, d_p, b->o_posmatch.as<uint32_t>(), n_p, b->tb_len.as<uint16_t>(),
                             b->o_post.as<float>(), max_span, nuc ? ix->d_nuc.as<float>() : nullptr, nuc ? b->o_codes.as<uint8_t>() : nullptr, codes_stride, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    pc.lap("records+coverage");
    return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// enqueue / wait forms of the two batch calls: a caller thread hands a block to the batch's service thread and goes on with the next
// block on another batch; uploads, device work and the fp64 passes of different batches overlap without one caller thread per batch.
// Calls queued on one batch run in the order they were queued; after a failing call the rest of that batch's queue is skipped.
// ------------------------------------------------------------------------------------------------
static void svc_post(gm_batch* b, std::function<int()> job) {
    gm_batch::Service& sv = b->svc;
    std::lock_guard<std::mutex> lk(sv.mu);
    if (!sv.th.joinable())
        sv.th = std::thread([b] {
            gm_batch::Service& v = b->svc;
            for (;;) {
                std::function<int()> j;
                {
                    std::unique_lock<std::mutex> lk2(v.mu);
                    v.cv.wait(lk2, [&] { return v.quit || !v.q.empty(); });
                    if (v.q.empty()) return;
                    j = std::move(v.q.front()); v.q.pop_front();
                    v.busy = true;
                }
                int rc = GM_OK; std::string err;
                bool skip;
                { std::lock_guard<std::mutex> lk2(v.mu); skip = v.rc != GM_OK; }
                if (!skip) { rc = j(); if (rc != GM_OK) err = gm_last_error(); }
                {
                    std::lock_guard<std::mutex> lk2(v.mu);
                    if (rc != GM_OK && v.rc == GM_OK) { v.rc = rc; v.err = err; }
                    v.busy = false;
                }
                v.cv.notify_all();
            }
        });
    sv.q.push_back(std::move(job));
    sv.cv.notify_all();
}

extern "C" int gm_map_batch_enqueue(gm_index* ix, const gm_params* p, gm_batch* b, const gm_reads* reads, gm_hits* out, void* stream) {
    if (!ix || !p || !b || !reads || !out) return GM_E_ARG;
    const gm_params pc = *p; const gm_reads rc = *reads;           // the structs are copied; the arrays they point to stay the caller's until gm_batch_wait
    svc_post(b, [=]() { return gm_map_batch(ix, &pc, b, &rc, out, stream); });
    return GM_OK;
}

extern "C" int gm_output_batch_enqueue(gm_index* ix, const gm_params* p, gm_batch* b, const gm_reads* reads, const gm_hits* hits, gm_sam_out* out, void* stream) {
    if (!ix || !p || !b || !reads || !hits || !out) return GM_E_ARG;
    const gm_params pc = *p; const gm_reads rc = *reads;
    svc_post(b, [=]() { return gm_output_batch(ix, &pc, b, &rc, hits, out, stream); });     // `hits` is read when the call runs: the gm_hits a queued gm_map_batch fills
    return GM_OK;
}

extern "C" int gm_batch_wait(gm_batch* b) {
    if (!b) return GM_E_ARG;
    gm_batch::Service& sv = b->svc;
    std::unique_lock<std::mutex> lk(sv.mu);
    sv.cv.wait(lk, [&] { return sv.q.empty() && !sv.busy; });
    const int rc = sv.rc;
    if (rc != GM_OK) gm_set_error(sv.err);
    sv.rc = GM_OK; sv.err.clear();
    return rc;
}

// ------------------------------------------------------------------------------------------------
// unit-level device entry points
// ------------------------------------------------------------------------------------------------
extern "C" int gm_dev_sa_interval(gm_index* ix, const char* kmers, uint32_t n, uint32_t m, uint64_t* start, uint64_t* end) {
    if (!ix || !kmers || !start || !end || m == 0) return GM_E_ARG;
    if (ix->host_only) return GM_E_NO_DEVICE;
    HIPCHK(hipSetDevice(ix->device));
    DevBuf dk, ds, de;
    if (dk.ensure((size_t)n * m) || ds.ensure((size_t)n * 4) || de.ensure((size_t)n * 4)) return GM_E_NOMEM;
    int rc = GM_OK;
    std::vector<uint32_t> s(n), e(n);
    do {
        if (hipMemcpy(dk.p, kmers, (size_t)n * m, hipMemcpyHostToDevice) != hipSuccess) { rc = GM_E_HIP; break; }
        if (gmk_sa_interval(ix->dev, dk.as<uint8_t>(), n, m, ds.as<uint32_t>(), de.as<uint32_t>(), nullptr)) { rc = GM_E_HIP; break; }
        if (hipMemcpy(s.data(), ds.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = GM_E_HIP; break; }
        if (hipMemcpy(e.data(), de.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = GM_E_HIP; break; }
    } while (0);
    dk.release(); ds.release(); de.release();
    if (rc) { gm_set_error("gm_dev_sa_interval: HIP failure"); return rc; }
    for (uint32_t i = 0; i < n; ++i) { start[i] = s[i]; end[i] = e[i]; }
    return GM_OK;
}

extern "C" int gm_dev_locate(gm_index* ix, const uint64_t* ranks, uint32_t n, int use_full_sa, uint64_t* out) {
    if (!ix || !ranks || !out) return GM_E_ARG;
    if (ix->host_only) return GM_E_NO_DEVICE;
    if (use_full_sa && !ix->full_sa) { gm_set_error("index was opened without GM_INDEX_FULL_SA"); return GM_E_ARG; }
    HIPCHK(hipSetDevice(ix->device));
    std::vector<uint32_t> r32(n), o32(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (ranks[i] == 0 || ranks[i] > ix->h.seq_len) { gm_set_error("rank out of range"); return GM_E_ARG; }
        r32[i] = (uint32_t)ranks[i];
    }
    DevBuf dr, dout;
    if (dr.ensure((size_t)n * 4) || dout.ensure((size_t)n * 4)) return GM_E_NOMEM;
    int rc = GM_OK;
    do {
        if (hipMemcpy(dr.p, r32.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = GM_E_HIP; break; }
        if (gmk_locate(ix->dev, dr.as<uint32_t>(), n, use_full_sa, dout.as<uint32_t>(), nullptr)) { rc = GM_E_HIP; break; }
        if (hipMemcpy(o32.data(), dout.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = GM_E_HIP; break; }
    } while (0);
    dr.release(); dout.release();
    if (rc) { gm_set_error("gm_dev_locate: HIP failure"); return rc; }
    for (uint32_t i = 0; i < n; ++i) out[i] = o32[i];
    return GM_OK;
}

static int unit_batch(gm_index* ix, const gm_params* p, const gm_reads* reads, gm_batch** bo, GmDevParams& dp) {
    gm_batch* b = nullptr;
    int rc = gm_batch_create(ix, reads->n ? reads->n : 1, reads->stride ? reads->stride : 1, &b);
    if (rc) return rc;
    rc = gm_batch_upload(b, p, reads, nullptr);
    if (!rc) rc = sync_params(ix, p, dp, nullptr);
    if (!rc) {
        if (hipMemsetAsync(b->counters.p, 0, GMK_N * 8, nullptr) != hipSuccess || hipMemsetAsync(b->small.p, 0, 64, nullptr) != hipSuccess ||
            hipMemsetAsync(b->rs_overflow.p, 0, 2 * (size_t)b->n, nullptr) != hipSuccess || gmk_prep(ix->dev, dp, b->dev, nullptr)) rc = GM_E_HIP;
    }
    if (rc) { gm_batch_destroy(b); return rc; }
    *bo = b;
    return GM_OK;
}

extern "C" int gm_dev_nw_score(gm_index* ix, const gm_params* p, const gm_reads* reads, const uint32_t* read_idx, const uint8_t* strand,
                               const uint64_t* pos, uint32_t n, float* score, uint8_t* valid) {
    if (!ix || !p || !reads || !read_idx || !strand || !pos || !score || !p->finalized) return GM_E_ARG;
    if (ix->host_only) return GM_E_NO_DEVICE;
    HIPCHK(hipSetDevice(ix->device));
    gm_batch* b; GmDevParams dp;
    int rc = unit_batch(ix, p, reads, &b, dp);
    if (rc) return rc;
    std::vector<GmCand> c(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (read_idx[i] >= reads->n || pos[i] > 0xFFFFFFFFull) { gm_batch_destroy(b); return GM_E_ARG; }
        c[i].rs = read_idx[i] * 2 + (strand[i] ? 1 : 0); c[i].b = (uint32_t)pos[i]; c[i].step = 0; c[i].flags = 0; c[i].pad = 0; c[i].score = 0;
    }
    do {
        b->cand_cap = std::max<uint32_t>(b->cand_cap, n + 16);
        if (b->cands.ensure((size_t)b->cand_cap * sizeof(GmCand))) { rc = GM_E_NOMEM; break; }
        fill_dev_batch(b);
        b->dev.cand_region = b->cand_cap;               // everything sits in shard 0
        if (n && hipMemcpy(b->cands.p, c.data(), (size_t)n * sizeof(GmCand), hipMemcpyHostToDevice) != hipSuccess) { rc = GM_E_HIP; break; }
        if (hipMemset(b->shards.p, 0, (size_t)GM_NSHARD * GM_SHARD_STRIDE * 4) != hipSuccess) { rc = GM_E_HIP; break; }
        if (hipMemcpy(b->shards.p, &n, 4, hipMemcpyHostToDevice) != hipSuccess) { rc = GM_E_HIP; break; }
        uint32_t rows_len = (b->len_min == b->len_max) ? b->len_max : 0u;          // (no k_prep here: the quality characters are looked at on the host)
        for (size_t q = 0; q < (size_t)reads->n * reads->stride && rows_len; ++q) if (reads->quals[q] >= 128) rows_len = 0;
        if (gmk_nw(ix->dev, dp, b->dev, n, rows_len, nullptr)) { rc = GM_E_HIP; break; }
        if (n && hipMemcpy(c.data(), b->cands.p, (size_t)n * sizeof(GmCand), hipMemcpyDeviceToHost) != hipSuccess) { rc = GM_E_HIP; break; }
    } while (0);
    gm_batch_destroy(b);
    if (rc) { gm_set_error("gm_dev_nw_score: HIP failure"); return rc; }
    for (uint32_t i = 0; i < n; ++i) { score[i] = c[i].score; if (valid) valid[i] = (c[i].flags & GMC_VALID) ? 1 : 0; }
    return GM_OK;
}

extern "C" int gm_dev_traceback(gm_index* ix, const gm_params* p, const gm_reads* reads, const uint32_t* read_idx, const uint8_t* strand,
                                const uint64_t* pos, uint32_t n, char* ops, uint32_t ops_stride, uint16_t* ops_len) {
    if (!ix || !p || !reads || !read_idx || !strand || !pos || !ops || !ops_len || !p->finalized) return GM_E_ARG;
    if (ix->host_only) return GM_E_NO_DEVICE;
    HIPCHK(hipSetDevice(ix->device));
    gm_batch* b; GmDevParams dp;
    int rc = unit_batch(ix, p, reads, &b, dp);
    if (rc) return rc;
    std::vector<GmCand> c(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (read_idx[i] >= reads->n || pos[i] > 0xFFFFFFFFull) { gm_batch_destroy(b); return GM_E_ARG; }
        c[i].rs = read_idx[i] * 2 + (strand[i] ? 1 : 0); c[i].b = (uint32_t)pos[i]; c[i].step = 0; c[i].flags = 0; c[i].pad = 0; c[i].score = 0;
    }
    const uint32_t ow = gm_ops_words(b->stride);
    std::vector<unsigned long long> packed((size_t)n * ow);
    do {
        if (b->tb_items.ensure((size_t)n * sizeof(GmCand) + 16) || b->tb_ops.ensure((size_t)n * ow * 8 + 16) || b->tb_len.ensure((size_t)n * 2 + 16)) { rc = GM_E_NOMEM; break; }
        if (n && hipMemcpy(b->tb_items.p, c.data(), (size_t)n * sizeof(GmCand), hipMemcpyHostToDevice) != hipSuccess) { rc = GM_E_HIP; break; }
        if (n && hipMemset(b->tb_ops.p, 0, (size_t)n * ow * 8) != hipSuccess) { rc = GM_E_HIP; break; }
        if (p->max_gap != 3) { if (b->band_moves.ensure(gm_band_moves_words(n, b->stride) * 8)) { rc = GM_E_NOMEM; break; } fill_dev_batch(b); }
        if (gmk_traceback(ix->dev, dp, b->dev, b->tb_items.as<GmCand>(), n, b->tb_ops.as<unsigned long long>(), ow, b->tb_len.as<uint16_t>(), nullptr, nullptr, nullptr, nullptr)) { rc = GM_E_HIP; break; }
        if (n && hipMemcpy(packed.data(), b->tb_ops.p, (size_t)n * ow * 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = GM_E_HIP; break; }
        if (n && hipMemcpy(ops_len, b->tb_len.p, (size_t)n * 2, hipMemcpyDeviceToHost) != hipSuccess) { rc = GM_E_HIP; break; }
        for (uint32_t i = 0; i < n; ++i) {                                // the public form: one character per operation
            char* o = ops + (size_t)i * ops_stride;
            memset(o, 0, ops_stride);
            for (uint32_t k = 0; k < ops_len[i] && k < ops_stride; ++k) o[k] = "MID?"[(packed[(size_t)i * ow + (k >> 5)] >> (2 * (k & 31))) & 3];
        }
    } while (0);
    gm_batch_destroy(b);
    if (rc) { gm_set_error("gm_dev_traceback: HIP failure"); return rc; }
    return GM_OK;
}

extern "C" int gm_dev_pair_hmm(gm_index* ix, const gm_params* p, const gm_reads* reads, const uint32_t* read_idx, const uint8_t* strand, const uint64_t* pos,
                               uint32_t n, float* out) {
    if (!ix || !p || !reads || !read_idx || !strand || !pos || !out || !p->finalized) return GM_E_ARG;
    if (ix->host_only) return GM_E_NO_DEVICE;
    HIPCHK(hipSetDevice(ix->device));
    gm_batch* b; GmDevParams dp;
    int rc = unit_batch(ix, p, reads, &b, dp);
    if (rc) return rc;
    std::vector<GmCand> c(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (read_idx[i] >= reads->n || pos[i] + reads->len[read_idx[i]] > ix->h.l_pac) { gm_batch_destroy(b); return GM_E_ARG; }
        c[i].rs = read_idx[i] * 2 + (strand[i] ? 1 : 0); c[i].b = (uint32_t)pos[i]; c[i].step = 0; c[i].flags = 0; c[i].pad = 0; c[i].score = 0;
    }
    const uint32_t Lmax = b->stride;
    do {
        if (b->tb_items.ensure((size_t)n * sizeof(GmCand) + 16) || b->snp_scratch.ensure((size_t)((n + 63) / 64) * gmk_pair_hmm_cells(Lmax) * 8 + 16) ||
            b->snp_hmm.ensure((size_t)n * Lmax * 5 * 4 + 16)) { rc = GM_E_NOMEM; break; }
        if (n && hipMemcpy(b->tb_items.p, c.data(), (size_t)n * sizeof(GmCand), hipMemcpyHostToDevice) != hipSuccess) { rc = GM_E_HIP; break; }
        if (n && hipMemset(b->snp_hmm.p, 0, (size_t)n * Lmax * 5 * 4) != hipSuccess) { rc = GM_E_HIP; break; }
        if (gmk_pair_hmm(ix->dev, dp, b->dev, b->tb_items.as<GmCand>(), n, b->snp_scratch.as<double>(), Lmax, b->snp_hmm.as<float>(), nullptr)) { rc = GM_E_HIP; break; }
        if (n && hipMemcpy(out, b->snp_hmm.p, (size_t)n * Lmax * 5 * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = GM_E_HIP; break; }
    } while (0);
    gm_batch_destroy(b);
    if (rc) { gm_set_error("gm_dev_pair_hmm: HIP failure"); return rc; }
    return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// coverage track
// ------------------------------------------------------------------------------------------------
extern "C" int gm_coverage_reset(gm_index* ix, uint32_t bin_size) {
    if (!ix || bin_size == 0) return GM_E_ARG;
    if (ix->host_only) {                                // no track in HBM: only the geometry the text writers need (CPU-side tests of gm_coverage_write_*)
        ix->cov_bins = ix->h.l_pac / bin_size + 64; ix->cov_bin_size = bin_size;
        return GM_OK;
    }
    HIPCHK(hipSetDevice(ix->device));
    uint64_t bins = ix->h.l_pac / bin_size + 64;        // the reference allocates l_pac/gGEN_SIZE floats and writes a little past it
    if (ix->d_cov.ensure(bins * 4)) return GM_E_NOMEM;
    HIPCHK(hipMemset(ix->d_cov.p, 0, bins * 4));
    ix->cov_bins = bins; ix->cov_bin_size = bin_size;
    return GM_OK;
}

extern "C" uint64_t gm_coverage_bins(const gm_index* ix) { return ix ? ix->cov_bins : 0; }
extern "C" void* gm_coverage_device_ptr(gm_index* ix) { return ix ? ix->d_cov.p : nullptr; }

extern "C" int gm_coverage_add(gm_index* ix, const uint64_t* pos, const uint32_t* span, const float* w, uint32_t n, void* stream) {
    if (!ix || !pos || !span || !w) return GM_E_ARG;
    if (!ix->cov_bins) { gm_set_error("coverage track not initialised (gm_coverage_reset)"); return GM_E_ARG; }
    HIPCHK(hipSetDevice(ix->device));
    if (n == 0) return GM_OK;
    DevBuf dp_, ds_, dw_;
    if (dp_.ensure((size_t)n * 8) || ds_.ensure((size_t)n * 4) || dw_.ensure((size_t)n * 4)) return GM_E_NOMEM;
    uint32_t max_span = 0;
    for (uint32_t i = 0; i < n; ++i) max_span = std::max(max_span, span[i]);
    int rc = GM_OK;
    hipStream_t st = S_(stream);
    do {
        if (hipMemcpyAsync(dp_.p, pos, (size_t)n * 8, hipMemcpyHostToDevice, st) != hipSuccess) { rc = GM_E_HIP; break; }
        if (hipMemcpyAsync(ds_.p, span, (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess) { rc = GM_E_HIP; break; }
        if (hipMemcpyAsync(dw_.p, w, (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess) { rc = GM_E_HIP; break; }
        if (gmk_coverage_add(ix->d_cov.as<float>(), ix->cov_bins, ix->cov_bin_size, dp_.as<uint64_t>(), ds_.as<uint32_t>(), dw_.as<float>(), n, max_span, nullptr, nullptr, nullptr, st)) { rc = GM_E_HIP; break; }
        if (hipStreamSynchronize(st) != hipSuccess) { rc = GM_E_HIP; break; }
    } while (0);
    dp_.release(); ds_.release(); dw_.release();
    if (rc) gm_set_error("gm_coverage_add: HIP failure");
    return rc;
}

extern "C" int gm_coverage_download(gm_index* ix, float* host) {
    if (!ix || !host || !ix->cov_bins) return GM_E_ARG;
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipMemcpy(host, ix->d_cov.p, ix->cov_bins * 4, hipMemcpyDeviceToHost));
    return GM_OK;
}

enum { MAX_SGR_LINE = 1200 };

// ---- track text at memory speed ------------------------------------------------------------------------------------------------------
// printf("%.Nf") of a float bin, N = 5 or 6, without printf: a float has 24 significant bits, so value x 10^N is exact in a double for
// every value below 2^53 / 10^N x 2^-17 (far beyond any coverage), and rint() of an exact number in the default rounding mode is the
// correctly rounded decimal glibc's printf prints (ties to even included).  Larger or non-finite values take snprintf.
static inline char* put_fixed(char* w, float v, int decimals) {
    const double scale = decimals == 5 ? 100000.0 : 1000000.0;
    if (!(v >= 0.0f) || !(v < 1.0e9f)) return w + snprintf(w, 64, decimals == 5 ? "%.5f" : "%f", v);
    const uint64_t q = (uint64_t)rint((double)v * scale);
    const uint64_t ip = q / (uint64_t)scale; uint32_t fp = (uint32_t)(q % (uint64_t)scale);
    char tmp[24]; int k = 0;
    uint64_t t = ip;
    do { tmp[k++] = (char)('0' + t % 10); t /= 10; } while (t);
    while (k) *w++ = tmp[--k];
    *w++ = '.';
    for (int d = decimals - 1; d >= 0; --d) { w[d] = (char)('0' + fp % 10); fp /= 10; }
    return w + decimals;
}
static inline char* put_long(char* w, long v) {
    if (v < 0) { *w++ = '-'; v = -v; }
    char tmp[24]; int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) *w++ = tmp[--k];
    return w;
}

// bins [0, nb) in slabs: host_threads() threads format one slice of a slab each (emit(k, w) appends the line of bin k, if it has
// one), the pieces are written with pwrite() at their offsets by the same threads
template <class Emit> static int write_track_text(const char* path, int append, uint64_t nb, size_t max_line, Emit&& emit) {
    // no O_APPEND: on Linux pwrite() on an O_APPEND descriptor ignores its offset, and the slices below are written concurrently
    const int fd = ::open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0644);
    if (fd < 0) { gm_set_error(std::string("cannot write ") + path); return GM_E_IO; }
    uint64_t file_off = append ? (uint64_t)lseek(fd, 0, SEEK_END) : 0;
    const unsigned T = host_threads();
    const uint64_t per = (uint64_t)std::max<long long>(1, gm_opt_ll("GM_TRACK_SLICE", 1ll << 20));      // bins per slice (the option: tests with several slices on a small index)
    std::vector<std::vector<char>> buf(T);
    std::vector<size_t> used(T);
    std::atomic<int> bad{ 0 };
    for (uint64_t s0 = 0; s0 < nb && !bad; s0 += per * T) {
        const unsigned parts = (unsigned)std::min<uint64_t>(T, (nb - s0 + per - 1) / per);
        auto format = [&](unsigned c) {
            const uint64_t lo = s0 + c * per, hi = std::min<uint64_t>(nb, lo + per);
            std::vector<char>& o = buf[c];
            if (o.size() < (size_t)(hi - lo) * max_line) o.resize((size_t)(hi - lo) * max_line);
            char* w = o.data();
            for (uint64_t k = lo; k < hi; ++k) w = emit(k, w, c, k == lo);
            used[c] = (size_t)(w - o.data());
        };
        {
            std::vector<std::thread> th;
            for (unsigned c = 1; c < parts; ++c) th.emplace_back(format, c);
            format(0);
            for (auto& x : th) x.join();
        }
        std::vector<uint64_t> off(parts);
        for (unsigned c = 0; c < parts; ++c) { off[c] = file_off; file_off += used[c]; }
        auto put = [&](unsigned c) {
            const char* q = buf[c].data(); size_t n = used[c]; uint64_t at = off[c];
            while (n) { const ssize_t k = ::pwrite(fd, q, n, (off_t)at); if (k <= 0) { bad = 1; return; } q += k; n -= (size_t)k; at += (uint64_t)k; }
        };
        {
            std::vector<std::thread> th;
            for (unsigned c = 1; c < parts; ++c) th.emplace_back(put, c);
            put(0);
            for (auto& x : th) x.join();
        }
    }
    ::close(fd);
    if (bad) { gm_set_error(std::string("write failed: ") + path); return GM_E_IO; }
    return GM_OK;
}

extern "C" int gm_coverage_write_sgr(gm_index* ix, const float* bins, const char* path, int append) {
    // GenomeBwt::PrintFinalSGR src/GenomeBwt.cpp:1212-1273: bins run over the CONCATENATED coordinate
    if (!ix || !bins || !path || !ix->cov_bin_size) return GM_E_ARG;
    const GmHostIndex& h = ix->h;
    const uint64_t bs = ix->cov_bin_size;
    // the reference walks `count` over the concatenated coordinate in steps of bin_size without resetting it per contig,
    // so bin k is printed under the contig that holds k * bin_size
    const uint64_t nb = (h.l_pac + bs - 1) / bs;
    size_t max_name = 0;
    for (const auto& c : h.contigs) max_name = std::max(max_name, c.name.size());
    std::vector<int> cur(host_threads(), 0);
    return write_track_text(path, append, nb, max_name + 48, [&](uint64_t k, char* w, unsigned c, bool first) -> char* {
        int& i = cur[c];
        const uint64_t count = k * bs;
        if (first) i = (int)host_pos2rid(h, count);
        while ((size_t)i + 1 < h.contigs.size() && count >= h.contigs[(size_t)i + 1].offset) ++i;
        if (!((double)bins[k] > 0.001)) return w;       // MIN_PRINT, GenomeBwt.cpp:928
        const GmContig& cg = h.contigs[(size_t)i];
        memcpy(w, cg.name.data(), cg.name.size()); w += cg.name.size();
        *w++ = '\t'; w = put_long(w, (long)(count - cg.offset) + 1); *w++ = '\t';
        w = put_fixed(w, bins[k], 5); *w++ = '\n';
        return w;
    });
}

extern "C" int gm_coverage_enable_nuc(gm_index* ix) {
    if (!ix || !ix->cov_bins) { gm_set_error("gm_coverage_reset first"); return GM_E_ARG; }
    HIPCHK(hipSetDevice(ix->device));
    if (ix->d_nuc.ensure(5 * ix->cov_bins * 4)) return GM_E_NOMEM;
    HIPCHK(hipMemset(ix->d_nuc.p, 0, 5 * ix->cov_bins * 4));
    ix->nuc_on = true;
    return GM_OK;
}

extern "C" void* gm_coverage_nuc_device_ptr(gm_index* ix) { return ix && ix->nuc_on ? ix->d_nuc.p : nullptr; }

extern "C" int gm_coverage_download_nuc(gm_index* ix, float* host) {
    if (!ix || !host || !ix->nuc_on) return GM_E_ARG;
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipMemcpy(host, ix->d_nuc.p, 5 * ix->cov_bins * 4, hipMemcpyDeviceToHost));
    return GM_OK;
}

extern "C" int gm_coverage_write_gmp(gm_index* ix, const gm_params* p, const float* bins, const float* nuc, const char* path, int append) {
    // GenomeBwt::PrintFinalBisulfite src/GenomeBwt.cpp:1092-1210
    if (!ix || !p || !bins || !nuc || !path || !ix->cov_bin_size || p->mode == GM_MODE_NORMAL) return GM_E_ARG;
    const GmHostIndex& h = ix->h;
    const uint64_t bs = ix->cov_bin_size, nb = ix->cov_bins;
    if (p->mode == GM_MODE_SNP) {
        // GenomeBwt::PrintFinalSNP src/GenomeBwt.cpp:930-1090 up to the per-nucleotide columns: every position whose total is above MIN_PRINT,
        // "%.5f" for all six numbers.  PrintSNPCall's likelihood-ratio columns need gsl_cdf_chisq_P (GSL): not written, the line ends here.
        const uint64_t nbk = (h.l_pac + bs - 1) / bs;
        size_t max_name = 0;
        for (const auto& c : h.contigs) max_name = std::max(max_name, c.name.size());
        std::vector<int> cur(host_threads(), 0);
        return write_track_text(path, append, nbk, max_name + 160, [&](uint64_t k, char* w, unsigned c, bool first) -> char* {
            int& i = cur[c];
            const uint64_t count = k * bs;
            if (first) i = (int)host_pos2rid(h, count);
            while ((size_t)i + 1 < h.contigs.size() && count >= h.contigs[(size_t)i + 1].offset) ++i;
            if (!(bins[k] > 0.001f)) return w;
            const GmContig& cg = h.contigs[(size_t)i];
            memcpy(w, cg.name.data(), cg.name.size()); w += cg.name.size();
            *w++ = '\t'; w = put_long(w, (long)(count - cg.offset) + 1); *w++ = '\t';
            w = put_fixed(w, bins[k], 5);
            for (int q = 0; q < 5; ++q) { *w++ = '\t'; w = put_fixed(w, nuc[(uint64_t)q * nb + k], 5); }
            *w++ = '\n';
            return w;
        });
    }
    const char want = p->mode == GM_MODE_BS ? 'c' : p->mode == GM_MODE_BS2 ? 'g' : p->mode == GM_MODE_ATOG ? 'a' : 't';
    const uint64_t nbk = (h.l_pac + bs - 1) / bs;
    size_t max_name = 0;
    for (const auto& c : h.contigs) max_name = std::max(max_name, c.name.size());
    std::vector<int> cur(host_threads(), 0);
    return write_track_text(path, append, nbk, max_name + 160, [&](uint64_t k, char* w, unsigned c, bool first) -> char* {
        int& i = cur[c];
        const uint64_t count = k * bs;
        if (first) i = (int)host_pos2rid(h, count);
        while ((size_t)i + 1 < h.contigs.size() && count >= h.contigs[(size_t)i + 1].offset) ++i;
        const char at = "acgt"[(h.pac[count >> 2] >> ((~count & 3) << 1)) & 3];
        if (at != want || !(bins[k] > 0.0f)) return w;
        const GmContig& cg = h.contigs[(size_t)i];
        memcpy(w, cg.name.data(), cg.name.size()); w += cg.name.size();
        *w++ = '\t'; w = put_long(w, (long)(count - cg.offset) + 1); *w++ = '\t';
        w = put_fixed(w, bins[k], 6);                                         // "%f"
        for (int q = 0; q < 5; ++q) { *w++ = '\t'; w = put_fixed(w, nuc[(uint64_t)q * nb + k], 5); }
        *w++ = '\n';
        return w;
    });
}
