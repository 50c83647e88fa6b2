// gm_capset.hip — the exact SET of the k-mers that exceed -h, for seeds longer than the k-mer table.
//
// The reference drops a k-mer with more than -h hits and tries the next position (inc/align_seq2_raw.cpp:213-217).  For a seed of
// mer > T characters the count is only known after the table lookup AND mer - T backward-search steps (two rank queries each:
// 8 random lines for -m 20 over the 16-character table) - and inside a repeat family that is paid at EVERY position of the read: on
// the repeat-rich human-scale reference k_seed spends 34.8 ms against 19.6 ms on the i.i.d. one (-m 20 -j 10 -h 150).  The k-mers that
// exceed the cap are few (a repeat family contributes its consensus k-mers: ~10^6 of 4^20 codes), so they are enumerated once per
// (index, mer, -h) - every table entry above the cap is extended character by character while its count stays above it - and kept
// in an open-addressing table of their 2 mer-bit codes (a few tens of MB: resident in the Infinity Cache).  k_seed asks the table
// first: a member is dropped with one probe and no search.  Exact (a k-mer is in the set iff its count exceeds -h), so the seeds are
// the same ones.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "gm_internal.h"
#pragma clang diagnostic ignored "-Wunused-function"      // (gm_device.h carries the seed walk of the vote kernels)
#include "gm_device.h"

#define GMC_MAXD 8                        // characters in front of the table's k-mer (mer - T)

// one thread per T-mer code above the cap: depth-first over the characters in front of it
__global__ void __launch_bounds__(256) k_capset_collect(GmDevIndex ix, const uint2* __restrict__ tab, int T, int D, uint32_t hcap, unsigned long long* __restrict__ out,
                                                        unsigned long long cap, unsigned long long* __restrict__ n_out) {
    const unsigned long long n_codes = 1ull << (2 * T);
    for (unsigned long long code = (unsigned long long)blockIdx.x * 256 + threadIdx.x; code < n_codes; code += (unsigned long long)gridDim.x * 256) {
        const uint2 iv = tab[code];
        if (iv.x == 0xFFFFFFFFu || iv.y - iv.x + 1u <= hcap) continue;
        uint32_t ks[GMC_MAXD + 1], ls[GMC_MAXD + 1], cs[GMC_MAXD + 1];
        unsigned long long cd[GMC_MAXD + 1];
        int level = 0;
        ks[0] = iv.x; ls[0] = iv.y; cs[0] = 0; cd[0] = code;
        while (level >= 0) {
            if (cs[level] == 4u) { --level; continue; }
            const uint32_t c = cs[level]++;
            const uint32_t k = gm_L2(ix, c) + gm_occ_plane(ix, ks[level] - 1u, c) + 1u;      // bwt_match_exact src/bwt.c:183-200, one more character
            const uint32_t l = gm_L2(ix, c) + gm_occ_plane(ix, ls[level], c);
            if (k > l || l - k + 1u <= hcap) continue;
            const unsigned long long ext = cd[level] | ((unsigned long long)c << (2 * (T + level)));
            if (level + 1 == D) {
                const unsigned long long at = atomicAdd(n_out, 1ull);
                if (at < cap) out[at] = ext;
            } else { ++level; ks[level] = k; ls[level] = l; cs[level] = 0; cd[level] = ext; }
        }
    }
}

__device__ __forceinline__ uint32_t gm_capset_slot(unsigned long long code, uint32_t bits) { return (uint32_t)((code * 0x9E3779B97F4A7C15ull) >> (64u - bits)); }

__global__ void __launch_bounds__(256) k_capset_insert(const unsigned long long* __restrict__ list, unsigned long long n, unsigned long long* __restrict__ table, uint32_t bits) {
    const uint32_t mask = (1u << bits) - 1u;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        const unsigned long long key = list[i] + 1ull;                    // 0 = empty
        uint32_t s = gm_capset_slot(list[i], bits);
        while (atomicCAS(&table[s], 0ull, key) != 0ull) s = (s + 1u) & mask;      // (codes of the list are distinct; the table is at most half full)
    }
}

// ---- which W-mers occur at all: one bit per code ------------------------------------------------------------------------------------
// With seeds longer than the k-mer table most k-mers k_seed tries do not occur: the read's WRONG strand (a 20-mer that is not in the
// reference dies one or two characters behind the 16 the table holds - present by chance 51 % of the time at 3.1 Gbp), and the k-mers
// that hold a sequencing error.  Each of them costs a table probe and, after it, one or two search steps of two rank queries:
// ~3.6 random lines to learn that the k-mer is dead and after how many characters.  A bitmap over the codes of W = 18 characters
// (4^18 bits = 8.6 GB, filled from the text itself) says "the last 18 characters do not occur" in ONE line for 95 % of those
// k-mers; the walk then skips every k-mer that contains them (mer - 17 positions), exactly as it does with the depth the search dies at.
__global__ void __launch_bounds__(256) k_kbit_build(const uint8_t* __restrict__ pac, unsigned long long n_pos /* text positions that start a W-mer */, int W, uint32_t* __restrict__ bits) {
    const unsigned long long mask = (1ull << (2 * W)) - 1ull;
    // a thread takes 32 consecutive positions: the code slides by one character
    for (unsigned long long p0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 32ull; p0 < n_pos; p0 += (unsigned long long)gridDim.x * 256 * 32ull) {
        unsigned long long code = 0;
        for (int q = 0; q < W - 1; ++q) { const unsigned long long x = p0 + (unsigned long long)q; code = (code << 2) | ((pac[x >> 2] >> ((~x & 3ull) << 1)) & 3u); }
        for (unsigned long long p = p0; p < p0 + 32ull && p < n_pos; ++p) {
            const unsigned long long x = p + (unsigned long long)(W - 1);
            code = ((code << 2) | ((pac[x >> 2] >> ((~x & 3ull) << 1)) & 3u)) & mask;
            atomicOr(&bits[code >> 5], 1u << (code & 31ull));
        }
    }
}

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

int gmk_kbit_build(const uint8_t* pac, unsigned long long text_len, int W, uint32_t* bits, void* stream) {
    if (W < 8 || W > 18 || text_len < (unsigned long long)W) return (int)hipErrorInvalidValue;
    const unsigned long long n_pos = text_len - (unsigned long long)W + 1ull;
    hipLaunchKernelGGL(k_kbit_build, dim3((uint32_t)std::min<unsigned long long>((n_pos / 32 + 256) / 256, 256ull * 64)), dim3(256), 0, S_(stream), pac, n_pos, W, bits);
    return (int)hipGetLastError();
}

int gmk_capset_collect(const GmDevIndex& ix, const uint2* tab, int T, int D, uint32_t hcap, unsigned long long* out, unsigned long long cap, unsigned long long* n_out, void* stream) {
    if (D < 1 || D > GMC_MAXD) return (int)hipErrorInvalidValue;
    const unsigned long long n = 1ull << (2 * T);
    hipLaunchKernelGGL(k_capset_collect, dim3((uint32_t)std::min<unsigned long long>((n + 255) / 256, 256ull * 128)), dim3(256), 0, S_(stream), ix, tab, T, D, hcap, out, cap, n_out);
    return (int)hipGetLastError();
}

int gmk_capset_insert(const unsigned long long* list, unsigned long long n, unsigned long long* table, uint32_t bits, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_capset_insert, dim3((uint32_t)std::min<unsigned long long>((n + 255) / 256, 4096ull)), dim3(256), 0, S_(stream), list, n, table, bits);
    return (int)hipGetLastError();
}
