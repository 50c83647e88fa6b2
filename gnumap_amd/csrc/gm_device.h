// gm_device.h — device-side helpers shared by the kernel translation units of libgnumap_hip (gm_kernels.hip, gm_bucket.hip): rank
// queries on the two BWT layouts, the score-table row product, wave-level scans, the candidate emitters, the serial seed walk.
// Everything here is __device__ inline code; each translation unit gets its own copy (no relocatable device code).
#pragma once
#include <hip/hip_runtime.h>
#include "gm_internal.h"

#define GM_NEG_INF (-100000.0f)
#define GM_EMPTY 0xFFFFFFFFu
#define GM_WALK_CAP (1u << 24)

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int gm_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ uint32_t gm_nt4(uint32_t ch) {
    // nst_nt4_table src/bntseq.c:47-64 : ACGT/acgt -> 0..3, everything else 4
    uint32_t u = ch & 0xDFu;        // fold case
    return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 4u;
}

__device__ __forceinline__ unsigned long long gm_wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ void gm_count(const GmDevBatch& b, int which, unsigned long long v) {
    unsigned long long s = gm_wave_sum(v);
    if (gm_lane() == 0 && s) atomicAdd(&b.counters[which], s);
}

// number of bases == c among the first `take` (clamped to 0..32) bases of a word holding 32 bases MSB first
__device__ __forceinline__ uint32_t gm_count_base(unsigned long long w, uint32_t c, int take) {
    unsigned long long pat = ((c & 1u) ? 0x5555555555555555ull : 0ull) | ((c & 2u) ? 0xAAAAAAAAAAAAAAAAull : 0ull);
    unsigned long long x = w ^ pat;
    unsigned long long m = ~(x | (x >> 1)) & 0x5555555555555555ull;
    if (take <= 0) return 0;
    if (take < 32) m &= ~0ull << (2 * (32 - take));
    return (uint32_t)__popcll(m);
}

// L2[c] / L2[c+1]-L2[c] by select (c is per-lane: no dynamic indexing of the by-value index struct)
__device__ __forceinline__ uint32_t gm_L2(const GmDevIndex& ix, uint32_t c) {
    return c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : ix.L2[3];
}
__device__ __forceinline__ uint32_t gm_L2n(const GmDevIndex& ix, uint32_t c) {
    return c == 0 ? ix.L2[1] : c == 1 ? ix.L2[2] : c == 2 ? ix.L2[3] : ix.L2[4];
}

// bwt_occ src/bwt.c:107-129 : occurrences of base c in BWT[0..k].  One 64-byte block: a dword of the
// cumulative count + two 16-byte loads of packed bases, all from the same line.
__device__ __forceinline__ uint32_t gm_occ(const GmDevIndex& ix, uint32_t k, uint32_t c) {
    if (k == ix.seq_len) return gm_L2n(ix, c) - gm_L2(ix, c);
    if (k == 0xFFFFFFFFu) return 0;                         // the reference's (bwtint_t)-1
    k -= (k >= ix.primary) ? 1u : 0u;
    const uint32_t* blk = ix.bwt + ((size_t)(k >> 7) << 4);
    uint32_t n = blk[2 * c];                                // low half of the u64 count (seq_len < 2^32)
    const uint4 w0 = *reinterpret_cast<const uint4*>(blk + 8);
    const uint4 w1 = *reinterpret_cast<const uint4*>(blk + 12);
    int within = (int)(k & 127u) + 1;
    n += gm_count_base(((unsigned long long)w0.x << 32) | w0.y, c, within);
    n += gm_count_base(((unsigned long long)w0.z << 32) | w0.w, c, within - 32);
    n += gm_count_base(((unsigned long long)w1.x << 32) | w1.y, c, within - 64);
    n += gm_count_base(((unsigned long long)w1.z << 32) | w1.w, c, within - 96);
    return n;
}

// the same rank query on the bit-plane layout: one 16-byte load.  x-space = positions of the $-removed BWT.
__device__ __forceinline__ uint32_t gm_occ_plane(const GmDevIndex& ix, uint32_t k, uint32_t c) {
    if (k == ix.seq_len) return gm_L2n(ix, c) - gm_L2(ix, c);
    if (k == 0xFFFFFFFFu) return 0;
    uint32_t x = k - ((k >= ix.primary) ? 1u : 0u);
    uint32_t blk = x / 96u;
    int r = (int)(x - blk * 96u) + 1;                       // bits [0, r) of the granule count
    const uint4 v = ix.occ_planes[(size_t)c * ix.occ_nblk + blk];
    uint32_t m0 = r >= 32 ? 0xFFFFFFFFu : ((1u << r) - 1u);
    uint32_t m1 = r >= 64 ? 0xFFFFFFFFu : (r > 32 ? ((1u << (r - 32)) - 1u) : 0u);
    uint32_t m2 = r >= 96 ? 0xFFFFFFFFu : (r > 64 ? ((1u << (r - 64)) - 1u) : 0u);
    return v.x + (uint32_t)__popc(v.y & m0) + (uint32_t)__popc(v.z & m1) + (uint32_t)__popc(v.w & m2);
}

// bwt_invPsi src/bwt.c:53-59
__device__ __forceinline__ uint32_t gm_inv_psi(const GmDevIndex& ix, uint32_t k) {
    uint32_t x = k - ((k > ix.primary) ? 1u : 0u);
    uint32_t word = ix.bwt[((size_t)(x >> 7) << 4) + 8 + ((x & 0x7fu) >> 4)];
    uint32_t c = (word >> ((~x & 0xfu) << 1)) & 3u;
    uint32_t r = gm_L2(ix, c) + gm_occ(ix, k, c);
    return k == ix.primary ? 0u : r;
}

// bwt_sa src/bwt.c:86-96 ; *steps receives the number of LF steps taken
__device__ __forceinline__ uint32_t gm_locate_walk(const GmDevIndex& ix, uint32_t k, uint32_t* steps) {
    uint32_t sa = 0;
    while ((k & ix.sa_mask) && sa < GM_WALK_CAP) {
        ++sa;
        k = gm_inv_psi(ix, k);
    }
    *steps = sa;
    return sa + ix.sa_samples[k >> ix.sa_shift];
}

// bin_seq::get_val src/bin_seq.cpp:975-987 with the PWM row given as (called base, p, q)
__device__ __forceinline__ float gm_get_val(uint32_t code, float p, float q, const float* s) {
    float r0 = code == 0 ? p : q, r1 = code == 1 ? p : q, r2 = code == 2 ? p : q, r3 = code == 3 ? p : q;
    float a = __fadd_rn(__fmul_rn(r0, s[0]), __fmul_rn(r1, s[1]));
    a = __fadd_rn(a, __fmul_rn(r2, s[2]));
    a = __fadd_rn(a, __fmul_rn(r3, s[3]));
    return a;
}

// bin_seq::max_flt src/bin_seq.cpp:1013-1026
__device__ __forceinline__ float gm_max3(float a, float b, float c) {
    if (a >= b) return a >= c ? a : c;
    return b >= c ? b : c;
}

// neighbour exchange inside a row of 16 lanes without touching LDS (DPP row shifts)
__device__ __forceinline__ float gm_from_prev_lane(float v) {      // lane i <- lane i-1   (row_shr:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, false));
}
__device__ __forceinline__ float gm_from_next_lane(float v) {      // lane i <- lane i+1   (row_shl:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xF, 0xF, false));
}

// bns_pos2rid src/bntseq.c:349-363 : last contig whose offset <= pos
__device__ __forceinline__ uint32_t gm_pos2rid(const uint32_t* coff, uint32_t n_seqs, uint32_t pos) {
    uint32_t lo = 0, hi = n_seqs - 1;
    while (lo < hi) {
        uint32_t mid = (lo + hi + 1) >> 1;
        if (pos >= coff[mid]) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// GenomeBwt::GetString src/GenomeBwt.cpp:384-415 validity: the window must lie inside one contig
__device__ __forceinline__ bool gm_window_ok(const GmDevIndex& ix, const uint32_t* coff, uint32_t begin, uint32_t L) {
    if ((unsigned long long)begin + L > ix.l_pac) return false;
    return gm_pos2rid(coff, ix.n_seqs, begin) == gm_pos2rid(coff, ix.n_seqs, begin + L - 1);
}

// The adaptive k-mer walk of align_sequence (inc/align_seq2_raw.cpp:200-231) over one read x strand: rb = the read's bases (LDS or
// HBM), out = its seed row.  The counters are the caller's (added to, never reset).
__device__ __forceinline__ void gm_seed_walk(const GmDevIndex& ix, const GmDevParams& p, const unsigned char* rb, const uint32_t L, const uint32_t strand,
                                             GmSeed* out, const uint32_t max_seeds, unsigned long long& nk, unsigned long long& nocc, unsigned long long& nblk,
                                             unsigned long long& ntab, unsigned long long& nseed, unsigned long long& nent) {
    uint32_t last = L - (uint32_t)p.mer;
    uint32_t i = 0;
    while (i < last) {
        // bwt_match_exact on the k-mer at [i, i+mer), right to left
        uint32_t k = 0, l = ix.seq_len;
        int t = p.mer - 1;
        bool ok = true;
        ++nk;
        if (p.kmer_tab) {
            // the last kmer_T characters in one lookup of the memoised backward search
            uint32_t code = 0;
            for (int q = 0; q < p.kmer_T; ++q, --t) {
                uint32_t pos = i + (uint32_t)t;
                uint32_t c = gm_nt4(strand ? rb[L - 1 - pos] : rb[pos]);
                if (c > 3) { ok = false; break; }           // t = position of the rightmost non-ACGT
                if (strand) c = 3 - c;
                code |= c << (2 * q);
            }
            if (ok) {
                ++ntab;
                bool answered = false;
                if (p.kmer_ctab) {                           // 2 MB, L2 resident: start rank + byte counts of 8 consecutive codes
                    const uint4 rec = p.kmer_ctab[code >> 3];
                    const uint32_t sub = code & 7u;
                    const unsigned long long cw = (unsigned long long)rec.y | ((unsigned long long)rec.z << 32);
                    const uint32_t cnt = (uint32_t)(cw >> (8 * sub)) & 255u;
                    if (rec.w == 0u && cnt >= 224u) {        // empty, and the record says after how many characters: no second probe
                        ok = false; t = p.mer - (int)(cnt - 223u);
                        answered = true;
                    } else if (rec.w == 0u) {
                        unsigned long long below = sub ? (cw & (~0ull >> (64 - 8 * sub))) : 0ull;
                        // bytes >= 224 are empty codes, not counts: a byte's bit 7 survives iff its bits 7, 6 and 5 are all set
                        const unsigned long long emp = below & (below << 1) & (below << 2) & 0x8080808080808080ull;
                        below &= ~((emp >> 7) * 0xFFull);
                        unsigned long long s2 = (below & 0x00FF00FF00FF00FFull) + ((below >> 8) & 0x00FF00FF00FF00FFull);      // 4 x 16-bit sums
                        const uint32_t pre = (uint32_t)((s2 * 0x0001000100010001ull) >> 48);
                        k = rec.x + pre; l = k + cnt - 1;
                        answered = true;
                    }
                }
                if (!answered) {
                    const uint2 iv = p.kmer_tab[code];
                    if (iv.x == 0xFFFFFFFFu) { ok = false; t = p.mer - (int)iv.y; }   // the last iv.y characters do not occur
                    else { k = iv.x; l = iv.y; }
                }
            }
        }
        for (; ok && t >= 0; --t) {
            uint32_t pos = i + (uint32_t)t;
            uint32_t c = gm_nt4(strand ? rb[L - 1 - pos] : rb[pos]);
            if (c > 3) { ok = false; break; }
            if (strand) c = 3 - c;
            uint32_t ok_ = gm_occ_plane(ix, k - 1, c);
            uint32_t ol_ = gm_occ_plane(ix, l, c);
            nocc += 2;
            {   // 64-byte blocks of the REFERENCE layout this step touches (bwt_2occ src/bwt.c:132-163: one when k-1 and l
                // share a block) - kept as the unit of the algorithmic-bytes accounting
                uint32_t k1 = k - 1, kb = k1 - ((k1 >= ix.primary) ? 1u : 0u), lb = l - ((l >= ix.primary) ? 1u : 0u);
                bool ks = (k1 == 0xFFFFFFFFu) || (k1 == ix.seq_len), ls = (l == ix.seq_len);
                nblk += (ks ? 0u : 1u) + (ls ? 0u : 1u) - ((!ks && !ls && (kb >> 7) == (lb >> 7)) ? 1u : 0u);
            }
            k = gm_L2(ix, c) + ok_ + 1;
            l = gm_L2(ix, c) + ol_;
            if (k > l) { ok = false; break; }
        }
        if (!ok) {
            // the suffix [i+t, i+mer) of this k-mer does not occur (or holds a non-ACGT): every k-mer starting in
            // [i, i+t] contains it, so the reference's one-by-one slide (:200-231) fails on all of them too
            i += (uint32_t)t + 1;
            continue;
        }
        uint32_t cnt = l - k + 1;
        if (p.hcap > 0 && cnt > p.hcap) { i += 1; continue; }       // too many hits: slide by one (:213-217)
        if (nseed < max_seeds && !(p.dbg & 128)) { GmSeed sd; sd.k = k; sd.l = l; sd.pos = i; out[nseed] = sd; }
        ++nseed;
        nent += cnt;
        i += (uint32_t)p.jump;
    }
}

struct GmVoteSrc {                      // where the located coordinates of seed t come from
    const uint32_t* full_sa;
    const uint32_t* coords;
};

template <class Table>
__device__ __forceinline__ void gm_emit(const GmDevBatch& b, bool emit, uint32_t rs, uint32_t bpos, uint32_t step, uint8_t flags) {
    unsigned long long mask = __ballot(emit);
    if (mask == 0) return;
    int lane = gm_lane();
    int leader = __ffsll((long long)mask) - 1;
    const uint32_t shard = blockIdx.x & (GM_NSHARD - 1);   // a single bump counter saturates at ~90 M atomics/s: shard it
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&b.shard_cnt[shard * GM_SHARD_STRIDE], (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (emit) {
        uint32_t idx = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (idx < b.cand_region) {
            GmCand c;
            c.rs = rs; c.b = bpos; c.step = (uint16_t)step; c.flags = flags; c.pad = 0; c.score = 0.0f;
            b.cands[(size_t)shard * b.cand_region + idx] = c;
        }
    }
}

// consumers of the sharded candidate list: exclusive prefix of the shard fill counts (block-wide, into LDS) and the
// map from a flat work index to the candidate slot
__device__ __forceinline__ uint32_t gm_cand_prefix(const GmDevBatch& b, uint32_t* pre /* GM_NSHARD + 1 */) {
    const int tid = threadIdx.x;
    if (tid < 64) {                                  // wave 0: 16 shards per lane, wave scan
        uint32_t v[GM_NSHARD / 64], sum = 0;
#pragma unroll
        for (int q = 0; q < GM_NSHARD / 64; ++q) {
            uint32_t c = b.shard_cnt[(size_t)(tid * (GM_NSHARD / 64) + q) * GM_SHARD_STRIDE];
            v[q] = c < b.cand_region ? c : b.cand_region;
            sum += v[q];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off); if (tid >= off) incl += t; }
        uint32_t run = incl - sum;
#pragma unroll
        for (int q = 0; q < GM_NSHARD / 64; ++q) { pre[tid * (GM_NSHARD / 64) + q] = run; run += v[q]; }
        if (tid == 63) pre[GM_NSHARD] = run;
    }
    __syncthreads();
    return pre[GM_NSHARD];
}

__device__ __forceinline__ size_t gm_cand_slot(const GmDevBatch& b, const uint32_t* pre, uint32_t w) {
    uint32_t lo = 0, hi = GM_NSHARD;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= w) lo = mid; else hi = mid; }
    return (size_t)lo * b.cand_region + (w - pre[lo]);
}

struct GmLdsTable {
    uint32_t* keys; uint32_t* vals; uint32_t mask; int bits;
    __device__ __forceinline__ uint32_t slot0(uint32_t key) const { return (key * 0x85EBCA6Bu) >> (32 - bits); }
};

// insert `key`; returns slot or GM_EMPTY when the table is full.  *fresh = key was not present.
template <class T>
__device__ __forceinline__ uint32_t gm_table_insert(T& tb, uint32_t key, bool* fresh) {
    uint32_t slot = tb.slot0(key);
    *fresh = false;
    for (uint32_t probe = 0; probe <= tb.mask; ++probe) {
        uint32_t old = atomicCAS(&tb.keys[slot], GM_EMPTY, key);
        if (old == GM_EMPTY) { *fresh = true; return slot; }
        if (old == key) return slot;
        slot = (slot + 1) & tb.mask;
    }
    return GM_EMPTY;
}

__device__ __forceinline__ void gm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// inclusive prefix sum over the 64 lanes of a wavefront with DPP adds (no LDS permutes): within rows of 16 lanes by
// row_shr 1, 2, 3, 4, 8, then the row totals by row_bcast:15 / row_bcast:31 (lanes a move does not reach add 0)
__device__ __forceinline__ uint32_t gm_wave_scan_incl(uint32_t x) {
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);        // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);        // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xF, 0xF, true);        // row_shr:3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xE, true);        // row_shr:4, banks 1-3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xC, true);        // row_shr:8, banks 2-3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);        // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);        // row_bcast:31 -> rows 2, 3
    return v;
}

struct GmKArgs { GmDevIndex ix; GmDevParams p; GmDevBatch b; };
// gm_kargs() reads the kernel's arguments where the hardware put them: every kernel that uses it takes EXACTLY (GmDevIndex ix,
// GmDevParams p, GmDevBatch b) as its first three by-value arguments.  The kernarg segment packs by-value structs at their natural
// alignment, which is what this struct does as long as all three are 8-byte aligned and their sizes multiples of 8:
static_assert(alignof(GmDevIndex) == 8 && alignof(GmDevParams) == 8 && alignof(GmDevBatch) == 8, "kernarg mirror: alignment");
static_assert(sizeof(GmDevIndex) % 8 == 0 && sizeof(GmDevParams) % 8 == 0 && sizeof(GmDevBatch) % 8 == 0, "kernarg mirror: padding");
static_assert(offsetof(GmKArgs, p) == sizeof(GmDevIndex) && offsetof(GmKArgs, b) == sizeof(GmDevIndex) + sizeof(GmDevParams), "kernarg mirror: offsets");
// one lane.  out = where the seeds go (LDS), or null: the read x strand's row in HBM, for the kernel it is handed to.  count = add the
// failed k-mers to the work counters (k_heavy_collect counts one k-mer and one table probe per seed).  Returns the number of seeds.
static __device__ __attribute__((noinline)) __attribute__((unused)) uint32_t gm_seed_walk_ool(const GmKArgs* a, const uint32_t rs, GmSeed* out, const int count) {
    const GmDevBatch& b = a->b;
    unsigned long long nk = 0, nocc = 0, nblk = 0, ntab = 0, nseed = 0, nent = 0;
    const uint32_t r = rs >> 1;
    gm_seed_walk(a->ix, a->p, b.bases + (size_t)r * b.stride, b.len[r], rs & 1u, out ? out : b.seeds + (size_t)rs * b.max_seeds, b.max_seeds, nk, nocc, nblk, ntab,
                 nseed, nent);
    if (count) {
        if (nk > nseed) atomicAdd(&b.counters[GMK_KMERS], nk - nseed);
        if (ntab > nseed) atomicAdd(&b.counters[GMK_TAB_LOOKUPS], ntab - nseed);
        if (nocc) { atomicAdd(&b.counters[GMK_OCC], nocc); atomicAdd(&b.counters[GMK_OCC_BLOCKS], nblk); }
    }
    return (uint32_t)nseed;
}

__device__ __forceinline__ const GmKArgs* gm_kargs() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (const GmKArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    return nullptr;
#endif
}
