// gm_bucket.hip — seed lookup + locate + vote of one READ (both strands) per wavefront through a direct-addressed k-mer -> positions
// table ("bucket table"), for seeds the k-mer table covers whole (-m <= 16) on references where a k-mer occurs a few times to a few
// tens of times (3.1 Gbp, -m 14: 11.6 on average).
//
// What it replaces (reference: the per-seed body of align_sequence, inc/align_seq2_raw.cpp:200-274: bwt_match_exact of the k-mer,
// bwt_sa of every SA hit, locs[b]++): in k_vote_tiny a seed costs a probe of the compact k-mer table (one random 128-byte line for 16
// bytes) and, dependent on it, its run of the suffix array (~48 bytes in 1.37 lines).  The measured price of a random fetch on MI355X
// is the LINE, whatever part of it is used (tools/ubench/gather.hip: 47-48 G records/s for 16-, 32-, 64- and 128-byte records alike,
// = 6 TB/s only when the whole line is wanted).  So the table holds, per mer-mer CODE, one 128-byte record = the hit count + up to 28
// TEXT POSITIONS (the suffix-array values of its interval): one line and ONE round trip per seed, 26 lines per 100-bp read instead
// of ~65.
//
// One wavefront = one read: lanes 0-31 the + strand, lanes 32-63 the - strand (two independent vote problems side by side: every
// vector instruction works for both), step st of a half fetches the records of seeds 4 st .. 4 st + 3 (8 lanes each; record layout
// at k_build_bucket).  While every k-mer at i = 0, jump, 2 jump .. occurs (and stays within -h) that is the adaptive walk; otherwise
// the half walks again (gm_bucket_rewalk: it asks about EVERY position of the read at once and walks over the answers - one more round
// trip however many k-mers are capped or absent).  Votes: see k_vote_bucket.  Read x
// strands that do not fit (too many hits, a record with an early position, a non-ACGT base) get their seed rows written and go to
// the list / heavy kernels exactly as from k_vote_tiny.
//
// (Round 3's "context records" - the table of the seed's last 15 characters with the 5 reference characters in front of every position,
// for -m 16 .. 20 - were correct and 3.5 x slower than k_seed on reads with 1 % errors; removed in round 4, DESIGN.md has the numbers.)
#include <hip/hip_runtime.h>
#include <algorithm>
#include "gm_internal.h"
#include "gm_device.h"

#define GMB_C 28                         // positions per record
#define GMB_EARLY 4096u                  // a record with a position below this may vote for the clamped window start b = 0 (:267): list kernel
#define GMB_LCAP 32                      // second-or-later arrivals in a filter slot per strand (the keys of the sweeps)
#define GMB_ECAP 384                     // SA hits per strand voted on here; more -> the list kernel
#define GMB_OV 16                        // seeds with more than GMB_C hits per strand handled here
#define GMB_MAXS 32                      // seeds per strand (one step tag bit each)
#define GMB_FWORDS 512                   // filter words per strand: 16384 one-bit slots

// ---- table construction: 8 lanes per code, one 16-byte store each -----------------------------------------------------------------
// Record of code c, 32 words, as the 8 lanes that fetch it with one 16-byte load each see it:
//   lane 0       { header, first SA rank, hit count, 0 }   (rank and count only when the hits do not fit)
//   lane q = 1..7, register j = 0..3: text position p = 7 j + q - 1 of the k-mer's SA interval, stored as position + 1; 0 = none
// header: bits 0-15 hit count (1 .. 28) when the positions are in the record; 0x80000000 = more hits than that (lane 0 has rank and
// count, all position words 0); 0x40000000 | d = the k-mer does not occur, its backward search died after d characters;
// 0x20000000 = one of its positions is below GMB_EARLY (a vote for the clamped b = 0 is possible: the list kernel counts those).
// Register j of ALL lanes covers positions 7j .. 7j + 6: a register beyond every count of the wave is skipped with one scalar branch.
__global__ void __launch_bounds__(256) k_build_bucket(const uint2* __restrict__ tab, const uint32_t* __restrict__ full_sa, uint4* __restrict__ bucket,
                                                      unsigned long long n_codes) {
    // (one more record behind the last code, all zero: where the lanes of a seed slot that is not in use load from)
    for (unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x; t < (n_codes + 1ull) * 8ull; t += (unsigned long long)gridDim.x * 256) {
        const unsigned long long code = t >> 3;
        const uint32_t q = (uint32_t)t & 7u;
        if (code == n_codes) { bucket[t] = make_uint4(0u, 0u, 0u, 0u); continue; }
        const uint2 iv = tab[code];
        const bool empty = iv.x == 0xFFFFFFFFu;
        const uint32_t cnt = empty ? 0u : iv.y - iv.x + 1u;
        const bool inl = !empty && cnt <= GMB_C;
        uint32_t w[4] = { 0u, 0u, 0u, 0u };
        bool early = false;
        if (inl) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t pi = 7u * j + q - 1u;                             // lane 0: wraps, never < cnt
                if (q != 0u && pi < cnt) w[j] = full_sa[iv.x + pi] + 1u;
            }
        }
        // the early flag: any position of the interval below GMB_EARLY (lane q looks at positions q, q + 8, .. - every lane the same answer)
        if (!empty) {
            if (cnt <= 4096u) { for (uint32_t i = 0; i < cnt; ++i) early |= full_sa[iv.x + i] < GMB_EARLY; }
            else early = true;                                                   // not worth a scan: read x strands with such seeds leave this kernel anyway
        }
        if (q == 0u) {
            w[0] = empty ? (0x40000000u | iv.y) : ((inl ? cnt : 0x80000000u) | (early ? 0x20000000u : 0u));
            if (!empty && !inl) { w[1] = iv.x; w[2] = cnt; }
        }
        bucket[t] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---- per-wave LDS -------------------------------------------------------------------------------------------------------------------
struct GmBucketLds {
    uint4 filt[2 * GMB_FWORDS / 4];                              // filter [2][512 words]: 4 x 16 bytes per lane, zeroed by every wave
    uint4 small[(2 * GMB_LCAP + 8) / 4];                         // dup keys [2][32] | tag masks [2] + pad: zeroed too
    union {                                                      // (the second is filled when the first is no longer needed)
        uint2 s_seed[2][GMB_MAXS];                               // the seeds of a half that walked: { table code, read offset | (context records) the characters in front of the table's k-mer << 16 }
        struct { uint32_t ov_k[2][GMB_OV], ov_n[2][GMB_OV], ov_ot[2][GMB_OV]; };      // seeds with more than GMB_C hits: first SA rank, count, (read offset + 1) | tag << 16
    };
};

__device__ __forceinline__ uint32_t gmb_half_bits(unsigned long long m, uint32_t h) { return h ? (uint32_t)(m >> 32) : (uint32_t)m; }
// number of set bits of the lane's own half of m below the lane
__device__ __forceinline__ uint32_t gmb_half_prefix(unsigned long long m, uint32_t h) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo(h ? 0u : (uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t gmb_pick(uint32_t h, uint32_t a0, uint32_t a1) { return h ? a1 : a0; }

// 2 mer bits of the read's 2-bit form at read offset i (k_prep: GmDevBatch::pack)
__device__ __forceinline__ uint32_t gmb_code_at(const uint32_t* form, uint32_t w2, uint32_t m, uint32_t i, uint32_t cmask) {
    const uint32_t o = 2u * (16u * w2 - i - m);
    return (uint32_t)((((unsigned long long)form[(o >> 5) + 1u] << 32) | form[o >> 5]) >> (o & 31u)) & cmask;
}

// The walk of the halves (strands) whose seeds do not all succeed; all 64 lanes call it, `need` = this lane's half walks.
// Seeds p_0 = first position >= 0 whose k-mer occurs (and stays within -h), p_{n+1} = first such position >= p_n + jump
// (inc/align_seq2_raw.cpp:200-231).  Which positions a seed lands on depends on every k-mer before it, but WHETHER a position can be a
// seed does not: the half keeps two bit sets over the positions of the read, "asked" and "can be a seed" (in scr; they live from the
// kernel's first call to its last), adds what the kernel has just learnt (ginfo: the positions of the records it fetched), walks over
// them by bit arithmetic until it needs a position nobody asked about, asks about a SET of positions in one round trip, and walks on.
// Which set: every position from there to the end of the read not asked yet (a strand fails here because it lies in a repeat - capped
// k-mers, position after position: one round trip, however many).
// Codes and offsets of the seeds go to s_seed; returns the half's seed count (0 for a half that did not
// walk).  scr: 4 KB of LDS.
#define GMB_PL 96                        // positions asked about per half and round
#define GMB_SCR_FAILED 930               // word of the scratch that holds the half's dropped k-mers
static __device__ __attribute__((noinline)) uint32_t gm_bucket_rewalk(const GmKArgs* a, const uint32_t r, const int lane, const bool need, const uint32_t ginfo,
                                                                      const bool first, const bool final, uint2* s_seed /* [2][GMB_MAXS] */, uint32_t* scr /* [1024] */) {
    const GmDevParams& p = a->p;
    const GmDevBatch& b = a->b;
    const uint32_t m = (uint32_t)p.mer, jump = (uint32_t)p.jump, w2 = b.pack_w2;
    const uint32_t TB = m;
    const uint32_t h = (uint32_t)lane >> 5, jj = (uint32_t)lane & 31u;
    const uint32_t* const row = b.pack + (size_t)r * b.pack_words;
    const uint32_t* const form = row + (h ? w2 + 2u : 1u);
    const uint32_t L = row[0] & 0xFFFFu, last = L - m;
    const uint32_t cmask = TB >= 16u ? 0xFFFFFFFFu : ((1u << (2u * TB)) - 1u);
    const uint32_t zero_code = cmask + 1u;
    const uint4* const bucket = reinterpret_cast<const uint4*>(p.bucket);
    uint32_t* const s_ok = scr + h * 64u;              // [2][64]: one bit per position
    uint32_t* const s_seen = scr + 128u + h * 64u;     // [2][64]
    uint16_t* const s_list = reinterpret_cast<uint16_t*>(scr + 256u) + h * GMB_PL;     // [2][GMB_PL] positions asked about in this round
    uint32_t* const s_failed = scr + GMB_SCR_FAILED + h;                                  // [2] k-mers tried and dropped on the way (for the work counters)
    static_assert(GMB_SCR_FAILED + 2 <= 1024 && 2 * GMB_PL * 2 <= 96 * 4, "rewalk scratch fits the 4 KB it is given");
    if (first) {
        for (uint32_t w = jj; w < 64u; w += 32u) { s_ok[w] = 0u; s_seen[w] = 0u; }
        if (jj == 0u) *s_failed = 0u;
        __syncthreads();
    }
    auto answer = [&](const uint32_t pos, const bool ok) {
        atomicOr(&s_seen[pos >> 5], 1u << (pos & 31u));
        if (ok) atomicOr(&s_ok[pos >> 5], 1u << (pos & 31u));
    };
    if (ginfo & 1u) answer(ginfo >> 2, (ginfo & 2u) != 0u);
    __syncthreads();
    uint32_t cur = 0, ns = 0, failed = 0, mypos = 0;
    for (;;) {
        // the walk over what is known (the same in every lane of a half); lane n keeps seed n.  One step = one seed: the k-mers the
        // reference tries and drops in front of it (asked, cannot be seeds) are jumped over 32 positions at a time
        while (need && cur < last && ns < GMB_MAXS) {
            const uint32_t w = cur >> 5, sh = cur & 31u;
            const uint32_t sn = (uint32_t)(((((unsigned long long)(w < 63u ? s_seen[w + 1u] : 0u)) << 32) | s_seen[w]) >> sh);      // positions cur .. cur + 31
            const uint32_t ok = (uint32_t)(((((unsigned long long)(w < 63u ? s_ok[w + 1u] : 0u)) << 32) | s_ok[w]) >> sh);
            const uint32_t stop = ok | ~sn;            // can be a seed, or nobody asked yet
            uint32_t d = stop ? (uint32_t)__builtin_ctz(stop) : 32u;
            if (cur + d > last) d = last - cur;
            failed += d; cur += d;
            if (stop == 0u || cur >= last) continue;
            if (!((sn >> d) & 1u)) break;                                         // not asked yet
            if (jj == ns) mypos = cur;
            ++ns; cur += jump;
        }
        const bool more = need && cur < last && ns < GMB_MAXS;
        if (__builtin_amdgcn_ballot_w64(more) == 0ull) break;
        // the positions to ask about: cur, cur + stride, .. below hi, those not asked yet, at most GMB_PL
        const uint32_t stride = 1u, hi = last;
        uint32_t nl = 0;
        for (uint32_t t = 0; __builtin_amdgcn_ballot_w64(more && cur + 32u * t * stride < hi && nl < GMB_PL) != 0ull; ++t) {
            const uint32_t pos = cur + (32u * t + jj) * stride;
            const bool want = more && pos < hi && nl < GMB_PL && !((s_seen[(pos >> 5) & 63u] >> (pos & 31u)) & 1u);
            const unsigned long long wm = __builtin_amdgcn_ballot_w64(want);
            const uint32_t at = nl + gmb_half_prefix(wm, h);
            if (want && at < GMB_PL) s_list[at] = (uint16_t)pos;
            nl += (uint32_t)__popc(gmb_half_bits(wm, h));
        }
        if (nl > GMB_PL) nl = GMB_PL;
        __syncthreads();
        for (uint32_t t0 = 0; __builtin_amdgcn_ballot_w64(32u * t0 < nl) != 0ull; t0 += 4u) {      // one lane per position: lane 0's part of the record says it all
            uint4 rec[4];
            uint32_t pos[4];
            bool act[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const uint32_t idx = 32u * (t0 + u) + jj;
                act[u] = idx < nl;
                pos[u] = act[u] ? s_list[idx] : 0u;
                rec[u] = bucket[(size_t)(act[u] ? gmb_code_at(form, w2, m, pos[u], cmask) : zero_code) * 8u];
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const uint32_t cnt = (rec[u].x & 0x80000000u) ? rec[u].z : (rec[u].x & 0xFFFFu);
                if (act[u]) answer(pos[u], !(rec[u].x & 0x40000000u) && !(p.hcap > 0 && cnt > p.hcap));
            }
        }
        __syncthreads();
    }
    // the seeds' codes
    if (need && jj < ns) {
        s_seed[h * GMB_MAXS + jj] = make_uint2(gmb_code_at(form, w2, m, mypos, cmask), mypos);
    }
    if (need && jj == 0u) *s_failed = failed;
    __syncthreads();
    return need ? ns : 0u;
}

// The lane's number, computed again: a value the compiler cannot tie to threadIdx.x, so that nothing derived from the lane number has to
// live across the (rare, out-of-line) walk - the hot path of k_vote_bucket then needs no scratch memory.
__device__ __forceinline__ int gmb_lane_again() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// four zero registers made HERE (an asm the compiler cannot hoist): inside the persistent loop of k_vote_bucket a plain zero vector is
// lifted out of the loop as a 4-register tuple, which cannot be rematerialised, is spilled, and comes back through scratch memory behind
// a wait for EVERY outstanding load - the prefetched words of the next read included
__device__ __forceinline__ uint4 gmb_zero4() {
    uint4 z;
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z.x), "=v"(z.y), "=v"(z.z), "=v"(z.w));
    return z;
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

// Votes (the part of the kernel that is bound by vector issue: written for few instructions per hit).
//   pass 1   every hit sets the bit of its window start in a 16384-bit filter of its strand with ONE returning atomic; a hit that finds
//            the bit set (a second or later arrival: the true locus' hits but for the first, and chance collisions) leaves its window
//            start in the strand's key list.  Nothing else per hit.
//   sweeps   per distinct key of the list: every hit of the strand is compared with the key (subtract + compare per hit; the lane
//            masks of a step are OR-ed in scalar registers), the lanes that match OR their seed's tag bit into the key's tag mask
//            in LDS.  A window start's votes ARE the seeds that hold it (one hit per seed can: a seed's positions are distinct, and
//            the clamped b = 0 never occurs here - records with an early position go to the list kernel), so
//            votes = popcount(tag mask), NW step = its -k-th lowest bit (inc/align_seq2_raw.cpp:262-274, process_hits :28-40).
// A strand without a second arrival has no candidate with -k >= 2: the wrong strand of a read ends after pass 1.
template <int STEPS>
__global__ void __launch_bounds__(64, STEPS <= 4 ? 8 : STEPS <= 6 ? 5 : 4) k_vote_bucket(GmDevIndex ix, GmDevParams p, GmDevBatch b, const uint32_t* rlist, const uint32_t* n_rlist) {
    __shared__ GmBucketLds S;
    uint32_t* const s_filt = reinterpret_cast<uint32_t*>(S.filt);          // [2][GMB_FWORDS]
    uint32_t* const s_keys = reinterpret_cast<uint32_t*>(S.small);         // [2][GMB_LCAP]
    uint32_t* const s_tagm = s_keys + 2 * GMB_LCAP;                          // [2]
    const uint32_t m = (uint32_t)p.mer, jump = (uint32_t)p.jump, w2 = b.pack_w2;
    const uint32_t TB = m;
    const uint32_t cmask = TB >= 16u ? 0xFFFFFFFFu : ((1u << (2u * TB)) - 1u);
    const uint32_t zero_code = cmask + 1u;             // the all-zero record behind the table (TB <= 15)
    const uint4* const bucket = reinterpret_cast<const uint4*>(p.bucket);
    // PERSISTENT waves: a wave takes reads blockIdx.x, blockIdx.x + gridDim.x, ..  What that buys: the header + form words of the NEXT
    // read are requested at the top of an iteration and have a whole read's work to arrive (the first of the read's two dependent HBM
    // trips disappears from the wave's lifetime), and the candidate stores of a read are not waited for - the wave goes on with the
    // next read instead of holding its slot until they are acknowledged.
    uint32_t n_hdr = 0, n_f0 = 0, n_f1 = 0;
    auto request_forms = [&](const uint32_t rr) {      // (the lane's word offsets are computed again each time: nothing per-lane lives across an iteration but these four words)
        const int ln = gmb_lane_again();
        const uint32_t hh = (uint32_t)ln >> 5, ireg = ((uint32_t)ln & 31u) * jump;
        const uint32_t oo = ireg + m <= 16u * w2 ? 2u * (16u * w2 - ireg - m) : 0u;
        const uint32_t* const rw = b.pack + (size_t)rr * b.pack_words;
        const uint32_t* const fm_ = rw + (hh ? w2 + 2u : 1u);
        n_hdr = rw[0]; n_f0 = fm_[oo >> 5]; n_f1 = fm_[(oo >> 5) + 1u];
    };
    // rlist: only the reads of this list (the ones k_vote_pair, gm_pair.hip, flagged and left alone): their words are requested when their turn comes
    const uint32_t n_items = rlist ? *n_rlist : b.n;
    if (!rlist && blockIdx.x < n_items) request_forms(blockIdx.x);
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    const uint32_t r = rlist ? rlist[item] : item;
    if (rlist) request_forms(r);
    int lane = gmb_lane_again();
    uint32_t h = (uint32_t)lane >> 5, jj = (uint32_t)lane & 31u, g = jj >> 3, q = (uint32_t)lane & 7u;
    uint32_t rs = 2u * r + h;

    // GM_DBG 64: phase clocks of one read in 256 (cycles of lane 0: forms, first records, first walk, its records, second walk + records, rest)
    const bool prof = (p.dbg & 64) && (r & 255u) == 0u && lane == 0;
    long long tck = prof ? clock64() : 0;
    auto tick = [&](int slot) { if (prof) { const long long now = clock64(); atomicAdd(&b.counters[GMK_DBG0 + slot], (unsigned long long)(now - tck)); tck = now; } };
    if (prof) atomicAdd(&b.counters[GMK_DBG8], 1ull);
    // ---- the read's header and the words that hold lane jj's regular k-mer (requested one read ago), and the next read's ----
    const uint32_t* const row = b.pack + (size_t)r * b.pack_words;
    const uint32_t* const form = row + (h ? w2 + 2u : 1u);
    const uint32_t i_reg = jj * jump;
    const bool inrow = i_reg + m <= 16u * w2;
    const uint32_t o = inrow ? 2u * (16u * w2 - i_reg - m) : 0u;
    const uint32_t hdr_v = n_hdr, f0 = n_f0, f1 = n_f1;
    if (!rlist && item + gridDim.x < n_items) request_forms(item + gridDim.x);
    (void)form;
    // the LDS structures are zeroed
    {
        static_assert(sizeof(S.filt) == 4 * 64 * 16 && sizeof(S.small) <= 64 * 16, "one wave zeroes the filter with four stores per lane");
        const uint4 z = gmb_zero4();
#pragma unroll
        for (int k = 0; k < 4; ++k) S.filt[lane + 64 * k] = z;
        if (lane < (int)(sizeof(S.small) / 16)) S.small[lane] = z;
    }
    const uint32_t hdr = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr_v);           // the same word in every lane: scalar from here on
    const uint32_t L = hdr & 0xFFFFu;
    const bool strand_on = h ? (p.neg_strand != 0) : (p.pos_strand != 0);
    if ((hdr >> 17) & 1u) {                           // status != 0 (too short / too poor): nothing to look up, on either strand
        if (lane == 0) { reinterpret_cast<uint32_t*>(b.n_seeds)[r] = 0u; reinterpret_cast<unsigned long long*>(b.n_entries)[r] = 0ull; }
        continue;
    }
    const uint32_t last = L - m;
    uint32_t ns_h;                                    // seeds of this lane's half
    bool listed = false;                              // the half's seed row is in HBM (serial walk of a read with a non-ACGT base)
    uint32_t cd[STEPS], of1[STEPS];                   // code and read offset + 1 (of the table's k-mer) of the seed whose record this lane's group fetches in step st
    uint4 rc[STEPS];
    uint4 hd4 = make_uint4(0u, 0u, 0u, 0u);           // lanes q < STEPS: lane 0's part of the record of seed 4 q + g {header, first SA rank, count, -}
    uint32_t hcode = 0, hof1 = 0;                     // ... its code and read offset + 1
    if ((hdr >> 16) & 1u) {
        // a base that is not ACGT: the 2-bit forms cannot say where.  Lane 0 walks each strand like k_seed does, into the read x
        // strand's row in HBM, and the list kernel votes.
        const int lane = gmb_lane_again();
        const uint32_t h = (uint32_t)lane >> 5;
        uint32_t n0 = 0, n1 = 0;
        if (lane == 0) {
            if (p.pos_strand) n0 = gm_seed_walk_ool(gm_kargs(), 2u * r, nullptr, 1);
            if (p.neg_strand) n1 = gm_seed_walk_ool(gm_kargs(), 2u * r + 1u, nullptr, 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        n0 = (uint32_t)__builtin_amdgcn_readlane((int)n0, 0); n1 = (uint32_t)__builtin_amdgcn_readlane((int)n1, 0);
        ns_h = h ? n1 : n0;
        if (ns_h > b.max_seeds) ns_h = b.max_seeds;
        if (ns_h > 32u) ns_h = 32u;
        listed = true;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) { cd[st] = 0; of1[st] = 0; rc[st] = make_uint4(0u, 0u, 0u, 0u); }
    } else {
        const bool act = strand_on && i_reg < last;
        unsigned long long win = (((unsigned long long)f1 << 32) | f0) >> (o & 31u);
        const uint32_t code = act ? (uint32_t)win & cmask : 0u;
        ns_h = strand_on ? (last + jump - 1u) / jump : 0u;                       // regular positions 0, jump, .. < last
        if (ns_h > 4u * STEPS) ns_h = 4u * STEPS;                                // (the host launches a form with enough slots)
        // The records of the half's seeds: step st, group g -> seed 4 st + g; lanes q < STEPS also fetch lane 0's part of the record of
        // seed 4 q + g: all headers of a half are looked at ONCE.  Returns the lanes whose seed cannot be one (does not occur, beyond -h).
        bool fail = false;
        auto fetch = [&](const uint32_t h, const uint32_t g, const uint32_t q) -> unsigned long long {
            const uint32_t hslot = 4u * q + g;
            // every lane loads: a slot that is not in use (and lane 0 of a group) reads the zero record
#pragma unroll
            for (int st = 0; st < STEPS; ++st) rc[st] = bucket[(size_t)((4u * st + g < ns_h && q != 0u) ? cd[st] : zero_code) * 8u + q];
            hd4 = bucket[(size_t)((q < (uint32_t)STEPS && hslot < ns_h) ? hcode : zero_code) * 8u];
            const bool empty = (hd4.x & 0x40000000u) != 0u;
            const uint32_t hc = (hd4.x & 0x80000000u) ? hd4.z : (hd4.x & 0xFFFFu);
            fail = empty || (p.hcap > 0 && hc > p.hcap);
            return __builtin_amdgcn_ballot_w64(fail);
        };
        // first the k-mers at the regular positions: the code of seed 4 st + g from the lane that computed it
        {
            const int bp_base = (int)((((uint32_t)lane & 32u) + g) << 2);
            const uint32_t of_g = g * jump + 1u, of_step = 4u * jump;
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                cd[st] = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_base + 16 * st, (int)code);
                of1[st] = of_g + (uint32_t)st * of_step;
            }
            hcode = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_base + (int)(q << 4), (int)code);
            hof1 = (4u * q + g) * jump + 1u;
        }
        tick(0);
        unsigned long long fm = fetch(h, g, q);
        if (prof && hd4.x == 0xFFFFFFFFu) tck = 0;      // (the clock is read after the records have arrived)
        tick(1);
        if (fm != 0ull) {
            // A k-mer that cannot be a seed: the halves that have one walk again (gm_bucket_rewalk).  From here on the seeds of a half go
            // through LDS (lane jj leaves the one at its regular position; a walk replaces those of its half), every attempt loads both
            // halves' records again, and nothing but the seed counts lives across a walk: the hot path above keeps its registers.
            // Context records: the first walk hands back seeds it only assumes good behind the slide; if one of them fails, the second
            // walk settles everything.
            if (jj < ns_h) S.s_seed[h][jj] = make_uint2(code, i_reg);
#pragma nounroll
            for (int attempt = 1; attempt < 3; ++attempt) {
                const int lane = gmb_lane_again();
                const uint32_t h = (uint32_t)lane >> 5, g = ((uint32_t)lane & 31u) >> 3, q = (uint32_t)lane & 7u;
                const uint32_t hslot = 4u * q + g;
                const bool walked = gmb_half_bits(fm, h) != 0u;
                // what the half knows now: the answers at the positions of its records.  The filter's LDS is the walk's scratch.
                const uint32_t ginfo = (q < (uint32_t)STEPS && hslot < ns_h) ? (((hof1 - 1u) << 2) | (fail ? 0u : 2u) | 1u) : 0u;
                const uint32_t nw = gm_bucket_rewalk(gm_kargs(), r, lane, walked, ginfo, attempt == 1, true, &S.s_seed[0][0], s_filt);
                if (prof && nw == 0xFFFFFFFFu) tck = 0;
                if (attempt == 1) tick(2);
                if (walked) ns_h = nw;
#pragma unroll
                for (int st = 0; st < STEPS; ++st) {
                    const uint32_t slot = 4u * st + g;
                    const uint2 sd = slot < ns_h ? S.s_seed[h][slot] : make_uint2(0u, 0u);
                    cd[st] = sd.x;
                    of1[st] = (sd.y & 0xFFFFu) + 1u;
                }
                {
                    const uint2 sd = (q < (uint32_t)STEPS && hslot < ns_h) ? S.s_seed[h][hslot] : make_uint2(0u, 0u);
                    hcode = sd.x; hof1 = (sd.y & 0xFFFFu) + 1u;
                }
                fm = fetch(h, g, q);
                if (fm == 0ull) break;             // (the seeds of a last walk are all good)
            }
            // the k-mers tried and dropped on the way (one counter per shard, summed by k_shard_stats: a single address takes a few hundred
            // million atomics per second at best, and with long seeds most reads come through here), and the filter's LDS zeroed again
            const int lane = gmb_lane_again();
            const uint32_t fl0 = s_filt[GMB_SCR_FAILED], fl1 = s_filt[GMB_SCR_FAILED + 1u];
            if (lane == 0 && fl0 + fl1 != 0u) atomicAdd(&b.shard_cnt[(size_t)(r & (GM_NSHARD - 1)) * GM_SHARD_STRIDE + 1u], fl0 + fl1);
            const uint4 z = gmb_zero4();
#pragma unroll
            for (int k = 0; k < 4; ++k) S.filt[lane + 64 * k] = z;
        }
    }
    lane = gmb_lane_again(); h = (uint32_t)lane >> 5; jj = (uint32_t)lane & 31u; g = jj >> 3; q = (uint32_t)lane & 7u; rs = 2u * r + h;
    if (p.dbg & 256) {                                // timing experiment (GM_DBG): stop when the records have arrived
        uint32_t acc = hd4.x;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) acc ^= rc[st].x ^ rc[st].y ^ rc[st].z ^ rc[st].w;
        if (acc == 0x12345u) b.counters[GMK_DBG0] = acc;
        continue;
    }
    // ---- seeds and SA hits of the two read x strands (k_heavy_collect sums them into the work counters and routes the heavy ones) ----
    uint32_t c_lane;
    if (listed) {
        GmSeed sd; sd.k = 1; sd.l = 0;
        if (jj < ns_h) sd = b.seeds[(size_t)rs * b.max_seeds + jj];
        c_lane = sd.l - sd.k + 1u;
    } else c_lane = (hd4.x & 0x40000000u) ? 0u : (hd4.x & 0x80000000u) ? hd4.z : (hd4.x & 0xFFFFu);
    uint32_t E0, E1;
    if (__builtin_amdgcn_ballot_w64(c_lane > (1u << 24)) != 0ull) {               // the 32-bit sums could overflow: saturate like k_seed
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
    if (ns0 + ns1 == 0u) continue;
    // ---- routing of each half by what its seeds actually hold ----
    const uint32_t E_h = gmb_pick(h, E0, E1);
    const bool heavy_h = ns_h != 0u && E_h > p.heavy_min;                         // sorted-key path (gm_heavy.hip), routed by k_heavy_collect
    const bool cut = p.nw && p.fast;                                             // --fast: only the first seed is looked at (:309-312)
    bool hvalid = !listed && q < (uint32_t)STEPS && 4u * q + g < ns_h;            // this lane holds a seed's header
    uint32_t Ev_h = E_h, n_ov_h = 0;
    bool early_h = false;
    if (!listed) {
        if (cut) {                                    // seed 0 of each half only: step 0, group 0
            hvalid = hvalid && q == 0u && g == 0u;
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c_lane, 0), c1 = (uint32_t)__builtin_amdgcn_readlane((int)c_lane, 32);
            Ev_h = gmb_pick(h, c0, c1);
#pragma unroll
            for (int st = 0; st < STEPS; ++st) if (st > 0 || g != 0u) rc[st] = make_uint4(0u, 0u, 0u, 0u);
        }
        early_h = gmb_half_bits(__builtin_amdgcn_ballot_w64(hvalid && (hd4.x & 0x20000000u) != 0u), h) != 0u;
        // seeds with more than GMB_C hits: descriptors in LDS, their hits come from the suffix array
        const bool ov = hvalid && (hd4.x & 0x80000000u) != 0u;
        const unsigned long long om = __builtin_amdgcn_ballot_w64(ov);
        if (om != 0ull) {                             // wave-uniform
            const uint32_t at = gmb_half_prefix(om, h);
            if (ov && at < GMB_OV) { S.ov_k[h][at] = hd4.y; S.ov_n[h][at] = hd4.z; S.ov_ot[h][at] = hof1 | ((4u * q + g) << 16); }
            n_ov_h = (uint32_t)__popc(gmb_half_bits(om, h));
        }
    }
    const bool big_h = ns_h != 0u && !heavy_h && (listed || early_h || Ev_h > p.bucket_ecap || n_ov_h > p.bucket_ovcap);
    bool vote_h = ns_h != 0u && !heavy_h && !big_h;
    // halves that leave: their seed rows {first SA rank, last SA rank, read offset} for the kernel they go to
    auto write_rows = [&](const bool which) {
        if (which && !listed && q < (uint32_t)STEPS && 4u * q + g < ns_h && 4u * q + g < b.max_seeds) {
            uint2 iv;
            iv = p.kmer_tab[hcode];
            GmSeed sd; sd.k = iv.x; sd.l = iv.y; sd.pos = hof1 - 1u;
            b.seeds[(size_t)rs * b.max_seeds + 4u * q + g] = sd;
        }
    };
    // ... and their entry in the list kernel's work list: a flag per read x strand, collected by k_big_collect (no counter that every
    // wave of the launch would queue up at)
    auto hand_over = [&]() {
        if (b.fixed_cnt != nullptr) b.fixed_cnt[rs] = 1;
        else { const uint32_t at = atomicAdd(b.n_big, 1u); b.big_list[at] = rs; }
    };
    if (__builtin_amdgcn_ballot_w64(heavy_h || big_h) != 0ull) {
        write_rows(heavy_h || big_h);
        if (jj == 0u && big_h) hand_over();
    }
    const unsigned long long vmask = __builtin_amdgcn_ballot_w64(vote_h);
    if (vmask == 0ull || (p.dbg & 512)) continue;      // (GM_DBG 512: timing experiment, stop before the votes)
    const uint32_t no0 = (uint32_t)__builtin_amdgcn_readlane((int)(vote_h ? n_ov_h : 0u), 0), no1 = (uint32_t)__builtin_amdgcn_readlane((int)(vote_h ? n_ov_h : 0u), 32);
    const uint32_t no_max = no0 > no1 ? no0 : no1;
    tick(3);
    __syncthreads();                                  // zeroed structures + descriptors (one wave: a wait, not a rendezvous)

    uint32_t* const keys = s_keys + h * GMB_LCAP;
    // ---- pass 1 ----
    uint32_t lc_h = 0;                                // keys in this half's list so far
    uint32_t known = 0;                               // the half's most recent key: the true locus' hits arrive one after the other, only the first is listed
    // the filter atomic of a hit: slot = the low 14 bits of its window start (chance hits are uniform, equal window starts meet).
    // Written for the instruction count (the kernel runs at the vector issue rate): the compares are taken as lane masks directly
    // (__builtin_amdgcn_uicmp: one v_cmp each, combined in scalar registers), the word's address is one shift + one and-or
    const uint32_t filt_off = (uint32_t)(size_t)(reinterpret_cast<unsigned char*>(s_filt) - reinterpret_cast<unsigned char*>(&S)) + h * (GMB_FWORDS * 4u);      // (S sits at LDS address 0: asserted by the launcher's static LDS size)
    auto arrive = [&](const uint32_t v, const uint32_t bp, uint32_t& bit) -> uint32_t {
        uint32_t old = 0;                              // lanes without a hit: no bit of theirs was set before
        bit = 1u << (bp & 31u);
        if (vote_h && v != 0u) {
            uint32_t addr;                             // byte address of the filter word: ((bp >> 3) & 0x7FC) | the half's base, as ONE and-or
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(addr) : "v"(bp >> 3), "s"(0x7FCu), "v"(filt_off));      // (no 32-bit literals in VOP3 on gfx9: the mask sits in a scalar register)
            old = atomicOr(reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(&S) + addr), bit);
        }
        return old;
    };
    // what its answer means: the bit was set before -> a second or later arrival; its window start goes to the key list unless it
    // is the half's most recent key (the true locus' hits arrive one after the other: listed once)
    auto settle = [&](const uint32_t bp, const uint32_t old, const uint32_t bit) {
        const unsigned long long dm = __builtin_amdgcn_uicmp(old & bit, 0u, 33 /* != */) & __builtin_amdgcn_uicmp(bp, known, 33 /* != */);
        if (dm != 0ull) {                             // wave-uniform; rare: once per distinct window start that is met again
            const bool dup = (old & bit) != 0u && bp != known;
            const uint32_t at = lc_h + gmb_half_prefix(dm, h);
            if (dup && at < GMB_LCAP) keys[at] = bp;
            lc_h += (uint32_t)__popc(gmb_half_bits(dm, h));
            const uint32_t d0 = (uint32_t)dm, d1 = (uint32_t)(dm >> 32);
            const uint32_t n0 = d0 ? (uint32_t)__builtin_amdgcn_readlane((int)bp, __builtin_ctz(d0)) : 0u;
            const uint32_t n1 = d1 ? (uint32_t)__builtin_amdgcn_readlane((int)bp, 32 + __builtin_ctz(d1)) : 0u;
            const uint32_t nk = gmb_pick(h, n0, n1);
            if (nk != 0u) known = nk;
        }
    };
    auto pass1 = [&](const uint32_t v, const uint32_t off1) {
        const uint32_t bp = v - off1; uint32_t bit;
        const uint32_t old = arrive(v, bp, bit);
        settle(bp, old, bit);
    };
    {   // registers 0 and 1 of every step (positions 0 .. 13 of a record: nearly always there): the eight atomics are issued together,
        // one wait for all of them instead of one LDS round trip per hit
        uint32_t bx[STEPS], by[STEPS], ox[STEPS], oy[STEPS], tx[STEPS], ty[STEPS];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) { bx[st] = rc[st].x - of1[st]; by[st] = rc[st].y - of1[st]; }
#pragma unroll
        for (int st = 0; st < STEPS; ++st) { ox[st] = arrive(rc[st].x, bx[st], tx[st]); oy[st] = arrive(rc[st].y, by[st], ty[st]); }
#pragma unroll
        for (int st = 0; st < STEPS; ++st) { settle(bx[st], ox[st], tx[st]); settle(by[st], oy[st], ty[st]); }
    }
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        if (__builtin_amdgcn_uicmp(rc[st].z, 0u, 33) != 0ull) pass1(rc[st].z, of1[st]);      // counts >= 15
        if (__builtin_amdgcn_uicmp(rc[st].w, 0u, 33) != 0ull) pass1(rc[st].w, of1[st]);      // counts >= 22
    }
    for (uint32_t d = 0; d < no_max; ++d) {           // seeds with more than GMB_C hits: 32 ranks of the suffix array per half and round
        const bool en = vote_h && d < n_ov_h;
        const uint32_t k = en ? S.ov_k[h][d] : 0u, n = en ? S.ov_n[h][d] : 0u, ot = en ? S.ov_ot[h][d] : 0u;
        for (uint32_t c = 0; __builtin_amdgcn_ballot_w64(32u * c < n) != 0ull; ++c) {
            const uint32_t idx = 32u * c + jj;
            pass1(idx < n ? ix.full_sa[k + idx] + 1u : 0u, ot & 0xFFFFu);
        }
    }
    tick(4);
    const unsigned long long anyd = __builtin_amdgcn_ballot_w64(lc_h != 0u);
    if (anyd == 0ull || (p.dbg & 1024)) continue;      // (GM_DBG 1024: timing experiment, stop after pass 1)                         // no second arrival on either strand: no window start with two votes
    __syncthreads();
    if (__builtin_amdgcn_ballot_w64(vote_h && lc_h > GMB_LCAP) != 0ull) {
        // more second arrivals than the key list holds (a repeat-rich read): the half goes to the list kernel after all
        const bool over_h = vote_h && lc_h > GMB_LCAP;
        write_rows(over_h);
        if (jj == 0u && over_h) hand_over();
        if (over_h) { vote_h = false; lc_h = 0; }
        if (__builtin_amdgcn_ballot_w64(lc_h != 0u) == 0ull) continue;
    }
    // ---- sweeps: one per distinct key of a half's list (both halves in the same instructions) ----
    uint32_t mykey = jj < lc_h ? keys[jj] : 0u;       // lane jj holds list entry jj of its half; 0 = none / done
    uint32_t n_em_h = 0;                              // candidates of this half so far
    GmCand first_c; first_c.rs = rs; first_c.b = 0; first_c.step = 0; first_c.flags = 4; first_c.pad = 0; first_c.score = 0.0f;
    const uint32_t shard = (2u * r + h) & (GM_NSHARD - 1);
    for (;;) {
        const unsigned long long rem = __builtin_amdgcn_ballot_w64(mykey != 0u);
        if (rem == 0ull) break;
        // the key of each half: its first remaining entry
        const uint32_t r0 = (uint32_t)rem, r1 = (uint32_t)(rem >> 32);
        const uint32_t k0 = r0 ? (uint32_t)__builtin_amdgcn_readlane((int)mykey, __builtin_ctz(r0)) : 0u;
        const uint32_t k1 = r1 ? (uint32_t)__builtin_amdgcn_readlane((int)mykey, 32 + __builtin_ctz(r1)) : 0u;
        const uint32_t key = gmb_pick(h, k0, k1);     // 0: this half has no key left (no hit has window start 0 here)
        if (mykey == key) mykey = 0u;                 // every list entry with this key is done
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            bool mt = (rc[st].x - of1[st]) == key;
            mt |= (rc[st].y - of1[st]) == key;
            if (__builtin_amdgcn_ballot_w64(rc[st].z != 0u) != 0ull) { mt |= (rc[st].z - of1[st]) == key; mt |= (rc[st].w - of1[st]) == key; }
            if (mt && key != 0u) atomicOr(&s_tagm[h], 1u << (4u * st + g));
        }
        for (uint32_t d = 0; d < no_max; ++d) {
            const bool en = vote_h && d < n_ov_h;
            const uint32_t k = en ? S.ov_k[h][d] : 0u, n = en ? S.ov_n[h][d] : 0u, ot = en ? S.ov_ot[h][d] : 0u;
            for (uint32_t c = 0; __builtin_amdgcn_ballot_w64(32u * c < n) != 0ull; ++c) {
                const uint32_t idx = 32u * c + jj;
                const uint32_t v = idx < n ? ix.full_sa[k + idx] + 1u : 0u;
                if (v != 0u && key != 0u && v - (ot & 0xFFFFu) == key) atomicOr(&s_tagm[h], 1u << (ot >> 16));
            }
        }
        __syncthreads();
        // the key's votes = the seeds that hold it
        const uint32_t tm = s_tagm[h];
        const uint32_t votes = (uint32_t)__popc(tm);
        const bool em = jj == 0u && key != 0u && votes >= (uint32_t)p.kmin;
        if (jj == 0u) s_tagm[h] = 0u;
        if (__builtin_amdgcn_ballot_w64(em) != 0ull) {
            uint32_t step;
            if (p.nw) { uint32_t mm = tm; for (int rr = 1; rr < p.kmin; ++rr) mm &= mm - 1; step = mm ? (uint32_t)(__ffs((int)mm) - 1) : 0u; }
            else step = votes;
            GmCand c;
            c.rs = rs; c.b = key; c.step = (uint16_t)step; c.flags = 4; c.pad = 0; c.score = 0.0f;           // key = (position + 1) - (read offset + 1): the window start
            if (p.dbg & 2048) { if (c.b == 0x7FFFFFF1u) b.counters[GMK_DBG1] = 1; }      // (GM_DBG 2048: timing experiment, no candidate stores)
            else if (em && b.fixed_cands != nullptr && n_em_h == 0u) first_c = c;          // slot 0 is stored last, with the count (one store for the usual single candidate)
            else if (em && b.fixed_cands != nullptr && n_em_h < GM_FIXED_C) b.fixed_cands[GM_FIXED_AT(b, rs, n_em_h)] = c;
            else if (em) {                            // more candidates than own slots (or no own slots): the shared list
                const uint32_t at = atomicAdd(&b.shard_cnt[shard * GM_SHARD_STRIDE], 1u);
                if (at < b.cand_region) b.cands[(size_t)shard * b.cand_region + at] = c;
            }
            if (em) ++n_em_h;
        }
        __syncthreads();
    }
    tick(5);
    if (jj == 0u && b.fixed_cands != nullptr && n_em_h != 0u && !(p.dbg & 2048)) {
        first_c.pad = (uint8_t)(n_em_h < GM_FIXED_C ? n_em_h : GM_FIXED_C);
        first_c.score = __uint_as_float(b.fixed_epoch);                            // k_cand_gather takes slots stamped with this launch only: no count array to zero, no second store
        b.fixed_cands[GM_FIXED_AT(b, rs, 0u)] = first_c;
    }
    tick(6);
    }   // next read of this wave
}

// ---- launchers ------------------------------------------------------------------------------------------------------------------------
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

int gmk_build_bucket(const uint2* tab, const uint32_t* full_sa, const uint8_t* pac, uint4* bucket, int T, int ctx, void* stream) {
    (void)pac; (void)ctx;
    const unsigned long long n = 1ull << (2 * T);
    const dim3 grid((uint32_t)std::min<unsigned long long>((n * 8 + 255) / 256, 256ull * 64));
    hipLaunchKernelGGL(k_build_bucket, grid, dim3(256), 0, S_(stream), tab, full_sa, bucket, n);
    return (int)hipGetLastError();
}

int gmk_bucket_max_seeds(void) { return 32; }

static uint32_t gm_bucket_grid(const GmDevParams& p, uint32_t max_reg) {
    const long long fixed = gm_opt_ll("GM_BUCKET_GRID", 0);
    if (fixed > 0) return (uint32_t)fixed;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const uint32_t per_simd = max_reg <= 16 ? 8u : max_reg <= 24 ? 5u : 4u;          // the kernels' launch bounds
    return (uint32_t)cus * 4u * per_simd * 4u;
}

// seeds per strand the launch has to hold: max_reg = ceil((longest read - mer) / jump)
int gmk_vote_bucket(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, uint32_t max_reg, const uint32_t* rlist, const uint32_t* n_rlist, void* stream) {
    if (b.n == 0) return 0;
    if (max_reg > 32 || p.bucket_ctx || p.mer != p.bucket_T) return (int)hipErrorInvalidValue;
    // persistent waves, each taking every grid-th read: 4 x as many one-wave workgroups as the chip holds at once (measured at 10 M reads:
    // exactly resident 10.3 ms - the dispatcher's placement is then final and uneven -, 2 x 9.7, 4 x 9.25, one read per wave 9.65)
    const dim3 grid(std::min<uint32_t>(b.n, gm_bucket_grid(p, max_reg))), blk(64);
    if (max_reg <= 8) hipLaunchKernelGGL((k_vote_bucket<2>), grid, blk, 0, S_(stream), ix, p, b, rlist, n_rlist);
    else if (max_reg <= 16) hipLaunchKernelGGL((k_vote_bucket<4>), grid, blk, 0, S_(stream), ix, p, b, rlist, n_rlist);
    else if (max_reg <= 24) hipLaunchKernelGGL((k_vote_bucket<6>), grid, blk, 0, S_(stream), ix, p, b, rlist, n_rlist);
    else hipLaunchKernelGGL((k_vote_bucket<8>), grid, blk, 0, S_(stream), ix, p, b, rlist, n_rlist);
    return (int)hipGetLastError();
}
