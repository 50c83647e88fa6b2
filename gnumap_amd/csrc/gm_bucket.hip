// gm_bucket.hip — seed lookup + locate + vote of one READ (both strands) per wavefront through a direct-addressed k-mer -> positions
// table ("bucket table"), for seeds the k-mer table covers whole (-m <= 16) on references where a k-mer occurs a few times to a few
// tens of times (3.1 Gbp, -m 14: 11.6 on average).
//
// What it replaces (reference: the per-seed body of align_sequence, inc/align_seq2_raw.cpp:200-274: bwt_match_exact of the k-mer,
// bwt_sa of every SA hit, locs[b]++): in k_vote_tiny a seed costs a probe of the compact k-mer table (one random 128-byte line for 16
// bytes) and, dependent on it, its run of the suffix array (~48 bytes in 1.37 lines).  The measured price of a random fetch on MI355X
// is the LINE, whatever part of it is used (tools/ubench/gather.hip: 47-48 G records/s for 16-, 32-, 64- and 128-byte records alike,
// = 6 TB/s only when the whole line is wanted).  So the table holds, per mer-mer CODE, one 128-byte record = the hit count + up to 31
// TEXT POSITIONS (the suffix-array values of its interval): one line and ONE round trip per seed, 26 lines per 100-bp read instead
// of ~65.
//
// Record of code c, 32 words.  Logical word w sits at physical word 4 (w % 8) + w / 8, so that lane q of the 8 lanes that fetch a
// record with one 16-byte load each holds logical words q, 8 + q, 16 + q, 24 + q: register j of ALL lanes covers words 8j .. 8j + 7,
// and a register whose words are beyond every count of the wave is skipped with one scalar branch (counts >= 24: almost never).
//   word 0         header: 1 .. 31 = hit count, positions in words 1 .. count, 0xFFFFFFFF in the rest
//                          0x40000000 | d = the k-mer does not occur, its backward search died after d characters (what the adaptive
//                                           walk needs to slide on, :213-226); all other words 0xFFFFFFFF
//                          0x80000000     = more than 31 hits: word 8 = first SA rank, word 16 = hit count (both in lane 0's load),
//                                           all other words 0xFFFFFFFF; the hits are read from the suffix array as before
// Positions of one k-mer may be taken in any order: a vote is locs[c - i]++ per position c (:262-274).
//
// One wavefront = one read: lanes 0-31 the + strand, lanes 32-63 the - strand (two independent vote problems side by side: every
// vector instruction works for both), step st of a half fetches the records of seeds 4 st .. 4 st + 3 (8 lanes each).
// While every k-mer at i = 0, jump, 2 jump .. occurs (and stays within -h) that is the adaptive walk; otherwise the half walks again
// (gm_bucket_rewalk: a regular round finds the first failing k-mer, a round over CONSECUTIVE positions the next one that does not
// fail - two round trips per failing seed however far the walk has to slide).  Votes: 4096 x 2-bit "seen / seen again" filter per
// strand, the hits whose slot was seen again compacted into a list, exact table, candidates - as in k_vote_tiny.  Read x strands that
// do not fit (too many hits, too many overflow seeds, a non-ACGT base) get their seed rows written and go to the list / retry /
// heavy kernels exactly as from k_vote_tiny.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "gm_internal.h"
#include "gm_device.h"

#define GMB_C 31                         // positions per record
#define GMB_TSZ 128                      // exact-table slots per strand
#define GMB_LCAP 128                     // "seen again" hits per strand that go through the table
#define GMB_ECAP 448                     // SA hits per strand voted on here; more -> the list kernel
#define GMB_OV 16                        // seeds with more than GMB_C hits per strand handled here
#define GMB_MAXS 32                      // seeds per strand (one step tag bit each)

// ---- table construction: 8 lanes per code, one 16-byte store each -----------------------------------------------------------------
__global__ void __launch_bounds__(256) k_build_bucket(const uint2* __restrict__ tab, const uint32_t* __restrict__ full_sa, uint4* __restrict__ bucket,
                                                      unsigned long long n_codes) {
    for (unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x; t < n_codes * 8ull; t += (unsigned long long)gridDim.x * 256) {
        const unsigned long long code = t >> 3;
        const uint32_t q = (uint32_t)t & 7u;
        const uint2 iv = tab[code];
        const bool empty = iv.x == 0xFFFFFFFFu;
        const uint32_t cnt = empty ? 0u : iv.y - iv.x + 1u;
        uint32_t w[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t lw = 8u * j + q;
            uint32_t v = 0xFFFFFFFFu;
            if (lw == 0u) v = empty ? (0x40000000u | iv.y) : cnt <= GMB_C ? cnt : 0x80000000u;
            else if (!empty && cnt <= GMB_C) { if (lw <= cnt) v = full_sa[iv.x + lw - 1u]; }
            else if (!empty) { if (lw == 8u) v = iv.x; else if (lw == 16u) v = cnt; }
            w[j] = v;
        }
        bucket[t] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---- per-wave LDS -------------------------------------------------------------------------------------------------------------------
struct GmBucketLds {
    uint4 zero[(2 * 256 + 3 * 2 * GMB_TSZ + 2 * 32) / 4];      // filter | keys | vals | step masks | b = 0 counts: zeroed by every wave
    uint32_t lb[2][GMB_LCAP];
    uint8_t lt[2][GMB_LCAP];
    uint32_t s_code[2][GMB_MAXS];                                // seeds of a half that walked again
    uint16_t s_pos[2][GMB_MAXS];
    uint32_t ov_k[2][GMB_OV], ov_n[2][GMB_OV], ov_ot[2][GMB_OV]; // seeds with more than GMB_C hits: first SA rank, count, read offset | tag << 16
};

__device__ __forceinline__ uint32_t gmb_half_bits(unsigned long long m, uint32_t h) { return h ? (uint32_t)(m >> 32) : (uint32_t)m; }
// number of set bits of the lane's own half of m below the lane
__device__ __forceinline__ uint32_t gmb_half_prefix(unsigned long long m, uint32_t h) {
    const uint32_t all = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    return all - (h ? (uint32_t)__popc((uint32_t)m) : 0u);
}
__device__ __forceinline__ uint32_t gmb_pick(uint32_t h, uint32_t a0, uint32_t a1) { return h ? a1 : a0; }

// 2 mer bits of the read's 2-bit form at read offset i (k_prep: GmDevBatch::pack)
__device__ __forceinline__ uint32_t gmb_code_at(const uint32_t* form, uint32_t w2, uint32_t m, uint32_t i, uint32_t cmask) {
    const uint32_t o = 2u * (16u * w2 - i - m);
    return (uint32_t)((((unsigned long long)form[(o >> 5) + 1u] << 32) | form[o >> 5]) >> (o & 31u)) & cmask;
}

// The walk of the halves (strands) whose regular positions do not all succeed; all 64 lanes call it, `need` = this lane's half walks.
// Seeds p_0 = first position >= 0 whose k-mer occurs (and stays within -h), p_{n+1} = first such position >= p_n + jump
// (inc/align_seq2_raw.cpp:200-231).  Round A: lane j probes pos0 + j * jump - the seeds before the first failing k-mer are final;
// round B: lane j probes pos0 + j, consecutive positions behind the failing k-mer (behind its dead suffix when it does not occur at
// all: every k-mer in between contains it) - the first that does not fail is the next seed.  Codes and offsets of the seeds go to
// s_code / s_pos; returns the half's seed count (0 for a half that did not walk).
static __device__ __attribute__((noinline)) uint32_t gm_bucket_rewalk(const GmKArgs* a, const uint32_t r, const int lane, const bool need, uint32_t* s_code /* [2][GMB_MAXS] */,
                                                                      uint16_t* s_pos /* [2][GMB_MAXS] */) {
    const GmDevParams& p = a->p;
    const GmDevBatch& b = a->b;
    const uint32_t m = (uint32_t)p.mer, jump = (uint32_t)p.jump, w2 = b.pack_w2;
    const uint32_t h = (uint32_t)lane >> 5, jj = (uint32_t)lane & 31u;
    const uint32_t* const row = b.pack + (size_t)r * b.pack_words;
    const uint32_t* const form = row + (h ? w2 + 2u : 1u);
    const uint32_t L = row[0] & 0xFFFFu, last = L - m;
    const uint32_t cmask = m >= 16u ? 0xFFFFFFFFu : ((1u << (2u * m)) - 1u);
    const uint4* const bucket = reinterpret_cast<const uint4*>(p.bucket);
    uint32_t pos0 = 0, ns = 0, mode = 0;
    bool walking = need;
    uint32_t extra = 0;                               // failing k-mers looked at (work counter)
    while (__builtin_amdgcn_ballot_w64(walking) != 0ull) {
        const uint32_t i = mode == 0u ? pos0 + jj * jump : pos0 + jj;
        const bool act = walking && i < last;
        uint32_t code = 0, hd = 0x40000000u, big_cnt = 0;
        if (act) {
            code = gmb_code_at(form, w2, m, i, cmask);
            const uint4 rec = bucket[(size_t)code * 8u];                         // header, word 8, word 16, word 24
            hd = rec.x; big_cnt = rec.z;
        }
        const bool empty = (hd & 0x40000000u) != 0u;
        const uint32_t cnt = (hd & 0x80000000u) ? big_cnt : hd;
        const bool capped = !empty && p.hcap > 0 && cnt > p.hcap;
        const bool ok = act && !empty && !capped;
        const uint32_t bh = gmb_half_bits(__builtin_amdgcn_ballot_w64(act && !ok), h), ah = gmb_half_bits(__builtin_amdgcn_ballot_w64(act), h),
                       oh = gmb_half_bits(__builtin_amdgcn_ballot_w64(ok), h);
        // round A: seeds before the first failing lane; what that lane found
        const uint32_t f = bh ? (uint32_t)(__ffs((int)bh) - 1) : (uint32_t)__popc(ah);
        const uint32_t adv_self = capped ? 1u : (m - (hd & 0xFFu)) + 1u;         // how far the walk moves on behind this k-mer
        const uint32_t adv_f = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane & 32u) + (f & 31u)) << 2, (int)adv_self);
        if (walking) {
            if (mode == 0u) {
                if (jj < f && ns + jj < GMB_MAXS) { s_code[h * GMB_MAXS + ns + jj] = code; s_pos[h * GMB_MAXS + ns + jj] = (uint16_t)i; }
                ns += f;
                if (!bh) walking = false;
                else { pos0 += f * jump + adv_f; mode = 1u; extra += 1u; }
            } else {
                if (oh) { const uint32_t s = (uint32_t)(__ffs((int)oh) - 1); pos0 += s; extra += s; mode = 0u; }
                else if (pos0 + 32u < last) { pos0 += 32u; extra += 32u; }
                else { extra += (uint32_t)__popc(ah); walking = false; }
            }
        }
    }
    if (jj == 0u && extra) { atomicAdd(&b.counters[GMK_KMERS], (unsigned long long)extra); atomicAdd(&b.counters[GMK_TAB_LOOKUPS], (unsigned long long)extra); }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    return ns < GMB_MAXS ? ns : GMB_MAXS;
}

// inclusive prefix sum within each half (32 lanes) of the wave: gm_wave_scan_incl without its last step
__device__ __forceinline__ uint32_t gmb_half_scan_incl(uint32_t x) {
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xE, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xC, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);        // row_bcast:15 -> rows 1, 3
    return v;
}

template <int STEPS>
__global__ void __launch_bounds__(64, 4) k_vote_bucket(GmDevIndex ix, GmDevParams p, GmDevBatch b) {
    __shared__ GmBucketLds S;
    uint32_t* const s_filt = reinterpret_cast<uint32_t*>(S.zero);          // [2][256]
    uint32_t* const s_keys = s_filt + 2 * 256;                               // [2][GMB_TSZ]
    uint32_t* const s_vals = s_keys + 2 * GMB_TSZ;
    uint32_t* const s_mlo = s_vals + 2 * GMB_TSZ;
    uint32_t* const s_cnt0 = s_mlo + 2 * GMB_TSZ;                            // [2][32]
    const uint32_t r = blockIdx.x;                     // grid = n reads
    const int lane = threadIdx.x;
    const uint32_t h = (uint32_t)lane >> 5, jj = (uint32_t)lane & 31u, g = jj >> 3, q = (uint32_t)lane & 7u;
    const uint32_t rs = 2u * r + h;
    const uint32_t m = (uint32_t)p.mer, jump = (uint32_t)p.jump, w2 = b.pack_w2;
    const uint32_t cmask = m >= 16u ? 0xFFFFFFFFu : ((1u << (2u * m)) - 1u);
    const uint4* const bucket = reinterpret_cast<const uint4*>(p.bucket);

    // ---- the read's header and the two words that hold lane jj's regular k-mer: one round trip ----
    const uint32_t* const row = b.pack + (size_t)r * b.pack_words;
    const uint32_t* const form = row + (h ? w2 + 2u : 1u);
    const uint32_t i_reg = jj * jump;
    const bool inrow = i_reg + m <= 16u * w2;
    const uint32_t o = inrow ? 2u * (16u * w2 - i_reg - m) : 0u;
    const uint32_t hdr = row[0], f0 = form[o >> 5], f1 = form[(o >> 5) + 1u];
    asm volatile("" :: "v"(f0), "v"(f1));
    // the LDS structures are zeroed under that trip
    {
        constexpr int NZ = (int)(sizeof(S.zero) / 16);
#pragma unroll
        for (int k = 0; k < (NZ + 63) / 64; ++k) if (lane + 64 * k < NZ) S.zero[lane + 64 * k] = make_uint4(0u, 0u, 0u, 0u);
    }
    const uint32_t L = hdr & 0xFFFFu;
    const bool strand_on = h ? (p.neg_strand != 0) : (p.pos_strand != 0);
    if ((hdr >> 17) & 1u) {                           // status != 0 (too short / too poor): nothing to look up, on either strand
        if (lane == 0) { reinterpret_cast<uint32_t*>(b.n_seeds)[r] = 0u; reinterpret_cast<unsigned long long*>(b.n_entries)[r] = 0ull; }
        return;
    }
    const uint32_t last = L - m;
    uint32_t ns_h;                                    // seeds of this lane's half
    bool listed = false;                              // the half's seed row is in HBM (serial walk of a read with a non-ACGT base)
    uint32_t cd[STEPS], of[STEPS], cnt[STEPS];
    uint4 rc[STEPS];
    bool sv[STEPS];
    if ((hdr >> 16) & 1u) {
        // a base that is not ACGT: the 2-bit forms cannot say where.  Lane 0 walks each strand like k_seed does, into the read x
        // strand's row in HBM, and the list kernel votes.
        uint32_t n0 = 0, n1 = 0;
        if (lane == 0) {
            if (p.pos_strand) n0 = gm_seed_walk_ool(gm_kargs(), 2u * r, nullptr, 1);
            if (p.neg_strand) n1 = gm_seed_walk_ool(gm_kargs(), 2u * r + 1u, nullptr, 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        n0 = (uint32_t)__builtin_amdgcn_readlane((int)n0, 0); n1 = (uint32_t)__builtin_amdgcn_readlane((int)n1, 0);
        ns_h = h ? n1 : n0;
        if (ns_h > b.max_seeds) ns_h = b.max_seeds;
        listed = true;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) { sv[st] = false; cd[st] = 0; of[st] = 0; cnt[st] = 0; rc[st] = make_uint4(~0u, ~0u, ~0u, ~0u); }
    } else {
        const bool act = strand_on && i_reg < last;
        const uint32_t code = act ? (uint32_t)((((unsigned long long)f1 << 32) | f0) >> (o & 31u)) & cmask : 0u;
        ns_h = strand_on ? (last + jump - 1u) / jump : 0u;                       // regular positions 0, jump, .. < last
        if (ns_h > 4u * STEPS) ns_h = 4u * STEPS;                                // (the host launches a form with enough slots)
        bool walked = false;
        for (int attempt = 0; attempt < 2; ++attempt) {
            // records of the half's seeds: step st, group g -> seed 4 st + g, its code from the lane that computed it (or from the walk)
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const uint32_t slot = 4u * st + g;
                if (attempt == 0) {
                    cd[st] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane & 32u) + slot) << 2, (int)code);
                    of[st] = slot * jump;
                    sv[st] = slot < ns_h;
                } else if (walked) {
                    sv[st] = slot < ns_h;
                    cd[st] = sv[st] ? S.s_code[h][slot] : 0u;
                    of[st] = sv[st] ? S.s_pos[h][slot] : 0u;
                }
            }
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                if (attempt == 0 || walked) {
                    if (sv[st]) rc[st] = bucket[(size_t)cd[st] * 8u + q];
                    else rc[st] = make_uint4(~0u, ~0u, ~0u, ~0u);
                }
            }
            // headers (lane q = 0 of each group): hit count, or what makes the walk slide
            bool fail = false;
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const uint32_t hd = rc[st].x;
                const bool head = q == 0u && sv[st];
                const bool big = (hd & 0x80000000u) != 0u, empty = (hd & 0x40000000u) != 0u;
                cnt[st] = head ? (big ? rc[st].z : empty ? 0u : hd) : 0u;
                fail |= head && (empty || (p.hcap > 0 && cnt[st] > p.hcap));
            }
            const unsigned long long fm = __builtin_amdgcn_ballot_w64(fail);
            if (attempt == 1 || fm == 0ull) break;
            walked = gmb_half_bits(fm, h) != 0u;
            const uint32_t nw = gm_bucket_rewalk(gm_kargs(), r, lane, walked, &S.s_code[0][0], &S.s_pos[0][0]);
            if (walked) ns_h = nw;
        }
    }
    // ---- seeds and SA hits of the two read x strands (k_heavy_collect sums them into the work counters and routes the heavy ones) ----
    uint32_t c_lane = 0;
    bool huge = false;
    GmSeed sd_list; sd_list.k = 0; sd_list.l = 0; sd_list.pos = 0;
    if (listed) {
        if (jj < ns_h) sd_list = b.seeds[(size_t)rs * b.max_seeds + jj];
        c_lane = jj < ns_h ? sd_list.l - sd_list.k + 1u : 0u;
        huge = c_lane > (1u << 24);
    } else {
#pragma unroll
        for (int st = 0; st < STEPS; ++st) { c_lane += cnt[st]; huge |= cnt[st] > (1u << 24); }
    }
    uint32_t E0, E1;
    if (__builtin_amdgcn_ballot_w64(huge) != 0ull) {  // the 32-bit sums could overflow: saturate like k_seed
        unsigned long long e64 = c_lane;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) e64 += __shfl_xor(e64, off);
        const unsigned long long a0 = __shfl(e64, 0), a1 = __shfl(e64, 32);
        E0 = a0 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)a0; E1 = a1 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)a1;
        if (lane == 0) { const unsigned long long over = (a0 - E0) + (a1 - E1); if (over) atomicAdd(&b.counters[GMK_SA_HITS], over); }
    } else {
        const uint32_t incl = gmb_half_scan_incl(c_lane);
        E0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31); E1 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    const uint32_t ns0 = (uint32_t)__builtin_amdgcn_readlane((int)ns_h, 0), ns1 = (uint32_t)__builtin_amdgcn_readlane((int)ns_h, 32);
    if (lane == 0) {
        reinterpret_cast<uint32_t*>(b.n_seeds)[r] = ns0 | (ns1 << 16);
        reinterpret_cast<unsigned long long*>(b.n_entries)[r] = (unsigned long long)E0 | ((unsigned long long)E1 << 32);
    }
    if (ns0 + ns1 == 0u) return;
    // ---- routing of each half by what its seeds actually hold ----
    const uint32_t E_h = gmb_pick(h, E0, E1);
    const bool heavy_h = ns_h != 0u && E_h > p.heavy_min;                         // sorted-key path (gm_heavy.hip), routed by k_heavy_collect
    const bool cut = p.nw && p.fast;                                             // --fast: only the first seed is looked at (:309-312)
    uint32_t n_ov_h = 0;
    uint32_t Ev_h = E_h;                                                          // hits this half would vote on
    if (!listed) {
        if (cut) {
#pragma unroll
            for (int st = 0; st < STEPS; ++st) { if (st > 0 || g != 0u) { sv[st] = false; cnt[st] = 0u; rc[st] = make_uint4(~0u, ~0u, ~0u, ~0u); } }
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cnt[0], 0), c1 = (uint32_t)__builtin_amdgcn_readlane((int)cnt[0], 32);
            Ev_h = gmb_pick(h, c0, c1);
        }
        // seeds with more than GMB_C hits: descriptors in LDS, their hits come from the suffix array
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const bool ov = q == 0u && sv[st] && (rc[st].x & 0x80000000u) != 0u;
            const unsigned long long om = __builtin_amdgcn_ballot_w64(ov);
            if (om != 0ull) {                         // wave-uniform
                const uint32_t at = n_ov_h + gmb_half_prefix(om, h);
                if (ov && at < GMB_OV) { S.ov_k[h][at] = rc[st].y; S.ov_n[h][at] = cnt[st]; S.ov_ot[h][at] = of[st] | ((4u * st + g) << 16); }
                n_ov_h += (uint32_t)__popc(gmb_half_bits(om, h));
            }
            if (q == 0u) { rc[st].x = ~0u; if (ov) { rc[st].y = ~0u; rc[st].z = ~0u; } }      // header words are not positions
        }
    }
    const bool big_h = ns_h != 0u && !heavy_h && (listed || Ev_h > GMB_ECAP || n_ov_h > GMB_OV);
    bool vote_h = ns_h != 0u && !heavy_h && !big_h;
    // halves that leave: their seed rows {first SA rank, last SA rank, read offset} for the kernel they go to
    auto write_rows = [&](const bool which) {
        if (listed) return;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const uint32_t slot = 4u * st + g;
            if (which && q == 0u && slot < ns_h && slot < b.max_seeds) {
                const uint2 iv = p.kmer_tab[cd[st]];
                GmSeed sd; sd.k = iv.x; sd.l = iv.y; sd.pos = of[st];
                b.seeds[(size_t)rs * b.max_seeds + slot] = sd;
            }
        }
    };
    if (__builtin_amdgcn_ballot_w64(heavy_h || big_h) != 0ull) {
        write_rows(heavy_h || big_h);
        if (jj == 0u && big_h) { const uint32_t at = atomicAdd(b.n_big, 1u); b.big_list[at] = rs; }
    }
    if (__builtin_amdgcn_ballot_w64(vote_h) == 0ull) return;
    const uint32_t no0 = (uint32_t)__builtin_amdgcn_readlane((int)(vote_h ? n_ov_h : 0u), 0), no1 = (uint32_t)__builtin_amdgcn_readlane((int)(vote_h ? n_ov_h : 0u), 32);
    const uint32_t no_max = no0 > no1 ? no0 : no1;
    __syncthreads();                                  // zeroed structures + descriptors (one wave: a wait, not a rendezvous)

    uint32_t* const filt = s_filt + h * 256;
    uint32_t* const cnt0 = s_cnt0 + h * 32;
    const uint32_t thr = (uint32_t)(p.kmin < 1 ? 1 : p.kmin);
    const uint32_t need = thr >= 2u ? 2u : 1u;
    // ---- pass 1: every hit sets "seen", or "seen again" when it finds "seen" set; b = 0 votes are counted per seed ----
    auto pass1 = [&](const uint32_t v, const uint32_t off, const uint32_t tag, const bool en) {
        const bool valid = en && v != 0xFFFFFFFFu;
        const uint32_t bp = __builtin_elementwise_sub_sat(v, off);                // :267
        if (valid && bp == 0u) atomicAdd(&cnt0[tag], 1u);
        if (valid && bp != 0u) {
            const uint32_t h2 = __umul24(bp, 0x9E3779u) >> 12, sh = (h2 >> 8 & 15u) << 1;
            const uint32_t old = atomicOr(&filt[h2 & 255u], 1u << sh);
            if ((old >> sh) & 1u) atomicOr(&filt[h2 & 255u], 2u << sh);
        }
    };
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        const uint32_t tag = 4u * st + g;
        pass1(rc[st].x, of[st], tag, vote_h);
        pass1(rc[st].y, of[st], tag, vote_h);
        if (__builtin_amdgcn_ballot_w64(vote_h && rc[st].z != 0xFFFFFFFFu) != 0ull) pass1(rc[st].z, of[st], tag, vote_h);      // counts >= 16
        if (__builtin_amdgcn_ballot_w64(vote_h && rc[st].w != 0xFFFFFFFFu) != 0ull) pass1(rc[st].w, of[st], tag, vote_h);      // counts >= 24
    }
    for (uint32_t d = 0; d < no_max; ++d) {           // seeds with more than GMB_C hits: 32 ranks of the suffix array per half and round
        const bool en = vote_h && d < n_ov_h;
        const uint32_t k = en ? S.ov_k[h][d] : 0u, n = en ? S.ov_n[h][d] : 0u, ot = en ? S.ov_ot[h][d] : 0u;
        for (uint32_t c = 0; __builtin_amdgcn_ballot_w64(32u * c < n) != 0ull; ++c) {
            const uint32_t idx = 32u * c + jj;
            const uint32_t v = idx < n ? ix.full_sa[k + idx] : 0xFFFFFFFFu;
            pass1(v, ot & 0xFFFFu, ot >> 16, en);
        }
    }
    __syncthreads();
    // ---- pass 2: the hits whose slot holds "seen again" (or "seen" with -k 1), compacted into the half's list ----
    uint32_t lc_h = 0;
    auto pass2 = [&](const uint32_t v, const uint32_t off, const uint32_t tag, const bool en) {
        const bool valid = en && v != 0xFFFFFFFFu;
        const uint32_t bp = __builtin_elementwise_sub_sat(v, off);
        const uint32_t h2 = __umul24(bp, 0x9E3779u) >> 12, sh = (h2 >> 8 & 15u) << 1;
        const bool reached = valid && bp != 0u && ((filt[h2 & 255u] >> sh) & need) != 0u;
        const unsigned long long rm = __builtin_amdgcn_ballot_w64(reached);
        if (rm != 0ull) {                             // wave-uniform
            const uint32_t at = lc_h + gmb_half_prefix(rm, h);
            if (reached && at < GMB_LCAP) { S.lb[h][at] = bp; S.lt[h][at] = (uint8_t)tag; }
            lc_h += (uint32_t)__popc(gmb_half_bits(rm, h));
        }
    };
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        const uint32_t tag = 4u * st + g;
        pass2(rc[st].x, of[st], tag, vote_h);
        pass2(rc[st].y, of[st], tag, vote_h);
        if (__builtin_amdgcn_ballot_w64(vote_h && rc[st].z != 0xFFFFFFFFu) != 0ull) pass2(rc[st].z, of[st], tag, vote_h);
        if (__builtin_amdgcn_ballot_w64(vote_h && rc[st].w != 0xFFFFFFFFu) != 0ull) pass2(rc[st].w, of[st], tag, vote_h);
    }
    for (uint32_t d = 0; d < no_max; ++d) {
        const bool en = vote_h && d < n_ov_h;
        const uint32_t k = en ? S.ov_k[h][d] : 0u, n = en ? S.ov_n[h][d] : 0u, ot = en ? S.ov_ot[h][d] : 0u;
        for (uint32_t c = 0; __builtin_amdgcn_ballot_w64(32u * c < n) != 0ull; ++c) {
            const uint32_t idx = 32u * c + jj;
            const uint32_t v = idx < n ? ix.full_sa[k + idx] : 0xFFFFFFFFu;
            pass2(v, ot & 0xFFFFu, ot >> 16, en);
        }
    }
    __syncthreads();
    // ---- the list through the exact table of the half ----
    uint32_t* const keys = s_keys + h * GMB_TSZ; uint32_t* const vals = s_vals + h * GMB_TSZ; uint32_t* const mlo = s_mlo + h * GMB_TSZ;
    bool full = lc_h > GMB_LCAP;
    uint32_t nkeys_h = 0;
    const uint32_t lc0 = (uint32_t)__builtin_amdgcn_readlane((int)lc_h, 0), lc1 = (uint32_t)__builtin_amdgcn_readlane((int)lc_h, 32);
    const uint32_t lc_max = (lc0 > lc1 ? lc0 : lc1) > GMB_LCAP ? GMB_LCAP : (lc0 > lc1 ? lc0 : lc1);
    for (uint32_t c = 0; 32u * c < lc_max; ++c) {
        const uint32_t idx = 32u * c + jj;
        bool fresh = false;
        if (vote_h && idx < lc_h && idx < GMB_LCAP) {
            const uint32_t bp = S.lb[h][idx], t = S.lt[h][idx];
            uint32_t slot = (bp * 0x9E3779B1u) >> 25;
            uint32_t old;
            int probes = 0;
            while ((old = atomicCAS(&keys[slot], 0u, bp)) != 0u && old != bp && ++probes < GMB_TSZ) slot = (slot + 1) & (GMB_TSZ - 1);
            fresh = old == 0u;
            if (!(old == 0u || old == bp)) full = true;
            else { atomicAdd(&vals[slot], 1u); atomicOr(&mlo[slot], 1u << t); }
        }
        nkeys_h += (uint32_t)__popc(gmb_half_bits(__builtin_amdgcn_ballot_w64(fresh), h));
    }
    __syncthreads();
    {   // a half whose list or table filled up goes to the global-table kernel (its seed row first)
        const unsigned long long fmask = __builtin_amdgcn_ballot_w64(vote_h && (full || nkeys_h > (uint32_t)(GMB_TSZ * 3 / 4)));
        if (fmask != 0ull) {
            const bool over_h = gmb_half_bits(fmask, h) != 0u;
            write_rows(over_h);
            if (over_h && jj == 0u) {
                b.rs_overflow[rs] = 1;
                const uint32_t j = atomicAdd(b.n_retry, 1u);
                const uint32_t want = 2 * Ev_h; uint32_t sz = 1024; while (sz < want && sz < 0x80000000u) sz <<= 1;
                const unsigned long long off = atomicAdd(&b.counters[GMK_HEAVY_SLOTS], (unsigned long long)sz);
                b.retry_list[j] = rs;
                b.retry_off[j] = off;
                atomicAdd(&b.counters[GMK_OVERFLOW_RS], 1ull);
            }
            if (over_h) vote_h = false;
            if (__builtin_amdgcn_ballot_w64(vote_h) == 0ull) return;
        }
    }
    // ---- candidates: four table slots per lane; a half with at most GM_FIXED_C of them stores into its own slots ----
    {
        bool em[4]; uint32_t ky[4], stp[4];
        uint32_t n_h = 0, before[4];
        unsigned long long ms[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t slot = (uint32_t)(u * 32) + jj;
            const uint32_t key = keys[slot], v = vals[slot];
            em[u] = vote_h && key != 0u && v >= (uint32_t)p.kmin;
            ky[u] = key; stp[u] = 0;
            if (em[u]) {
                if (p.nw) {
                    uint32_t mm = mlo[slot];
                    for (int rr = 1; rr < p.kmin; ++rr) mm &= mm - 1;
                    stp[u] = mm ? (uint32_t)(__ffs((int)mm) - 1) : 0u;
                } else stp[u] = v > 65535u ? 65535u : v;
            }
            ms[u] = __builtin_amdgcn_ballot_w64(em[u]);
            before[u] = n_h + gmb_half_prefix(ms[u], h);
            n_h += (uint32_t)__popc(gmb_half_bits(ms[u], h));
        }
        const bool fixed_h = b.fixed_cands != nullptr && n_h <= GM_FIXED_C;
        const unsigned long long shm = __builtin_amdgcn_ballot_w64(n_h != 0u && !fixed_h);
        uint32_t base = 0;
        const uint32_t shard = (2u * blockIdx.x + h) & (GM_NSHARD - 1);
        if (shm != 0ull) {                            // wave-uniform: some half needs the shared list
            if (jj == 0u && n_h != 0u && !fixed_h) base = atomicAdd(&b.shard_cnt[shard * GM_SHARD_STRIDE], n_h);
            base = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((uint32_t)lane & 32u) << 2, (int)base);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (em[u]) {
                GmCand c;
                c.rs = rs; c.b = ky[u]; c.step = (uint16_t)stp[u]; c.flags = 4; c.pad = 0; c.score = 0.0f;
                if (fixed_h) b.fixed_cands[(size_t)rs * GM_FIXED_C + before[u]] = c;
                else if (base + before[u] < b.cand_region) b.cands[(size_t)shard * b.cand_region + base + before[u]] = c;
            }
        if (jj == 0u && fixed_h && n_h != 0u) b.fixed_cnt[rs] = (uint8_t)n_h;
    }
    {   // b = 0: cumulative per-seed counts (a position at the very start of the reference)
        const uint32_t c0 = vote_h ? cnt0[jj] : 0u;
        if (__builtin_amdgcn_ballot_w64(c0 != 0u) != 0ull) {
            const uint32_t run = gmb_half_scan_incl(c0);
            const uint32_t total = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane & 32u) + 31u) << 2, (int)run);
            const uint32_t reached = gmb_half_bits(__builtin_amdgcn_ballot_w64(run >= (uint32_t)p.kmin), h);
            const bool emit = jj == 0u && total >= (uint32_t)p.kmin;
            const uint32_t step = p.nw ? (reached ? (uint32_t)(__ffs((int)reached) - 1) : 0u) : (total > 65535u ? 65535u : total);
            gm_emit<GmLdsTable>(b, emit, rs, 0u, step, 4);
        }
    }
}

// ---- launchers ------------------------------------------------------------------------------------------------------------------------
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

int gmk_build_bucket(const uint2* tab, const uint32_t* full_sa, uint4* bucket, int T, void* stream) {
    const unsigned long long n = 1ull << (2 * T);
    hipLaunchKernelGGL(k_build_bucket, dim3((uint32_t)std::min<unsigned long long>((n * 8 + 255) / 256, 256ull * 64)), dim3(256), 0, S_(stream), tab, full_sa, bucket, n);
    return (int)hipGetLastError();
}

int gmk_bucket_max_seeds(void) { return 32; }

// seeds per strand the launch has to hold: max_reg = ceil((longest read - mer) / jump)
int gmk_vote_bucket(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, uint32_t max_reg, void* stream) {
    if (b.n == 0) return 0;
    if (max_reg <= 16) hipLaunchKernelGGL((k_vote_bucket<4>), dim3(b.n), dim3(64), 0, S_(stream), ix, p, b);
    else if (max_reg <= 24) hipLaunchKernelGGL((k_vote_bucket<6>), dim3(b.n), dim3(64), 0, S_(stream), ix, p, b);
    else if (max_reg <= 32) hipLaunchKernelGGL((k_vote_bucket<8>), dim3(b.n), dim3(64), 0, S_(stream), ix, p, b);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}
