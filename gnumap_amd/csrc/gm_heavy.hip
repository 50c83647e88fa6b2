// gm_heavy.hip — the vote path for read x strands with very many SA hits (repeat seeds without a -h cap: 10^4 .. 10^7 hits each),
// bandwidth-proportional and without any per-read x strand hash table:
//
//   k_heavy_collect   read x strands whose seeds hold more than `heavy_min` SA hits are taken out of the ordinary vote kernels up front
//                     (their seed count is set to 0, the true count is kept in the heavy list) - routing by k_seed's own hit count,
//                     not after a kernel has overflowed
//   k_heavy_expand    every SA hit of a heavy read x strand becomes one 64-bit key  {index in the chunk | window start b | seed step}
//                     (coalesced reads of the suffix array, one 8-byte store per hit)
//   rocPRIM radix sort of the keys of a chunk (in HBM)
//   k_heavy_runs      in sorted order all votes for one window start are adjacent and ordered by seed step: the -k-th element of a
//                     run IS the vote with which the reference's counter reaches -k (inc/align_seq2_raw.cpp:262-274, process_hits
//                     :28-40), its step the step at which the position is aligned; --no_nw takes the run length as the score.
//                     The clamped position b = 0, the only one a single seed can vote for twice, needs no special case: duplicates
//                     of (b, step) are adjacent too.
//
// Cost: ~4 (SA read) + 8 (key) + radix passes x 16 bytes per SA hit, whatever the hit count of a single read x strand is; the
// global-memory CAS table of k_vote_retry (one dependent probe chain + one atomic per hit) stays for the moderate overflow cases only.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <algorithm>
#include "gm_internal.h"

// sum_counters: the seeds were looked up inside the vote kernel (GmDevParams::fused), which leaves the per read x strand counts only:
// the work counters k_seed keeps (one k-mer and one table probe per seed; SA hits) are summed here, one atomic per wave
__global__ void __launch_bounds__(256) k_heavy_collect(GmDevBatch b, uint32_t heavy_min, uint32_t* n_heavy, uint32_t* heavy_list /* {rs, n_seeds, SA hits} triples */,
                                                       int sum_counters) {
    unsigned long long a = 0, e = 0;
    const uint32_t n2 = 2 * b.n, n4 = n2 >> 2;           // four read x strands per thread and step: one 8-byte and one 16-byte load
    auto take = [&](uint32_t rs, uint32_t ns, uint32_t ne) {
        a += ns; e += ne;
        if (ns == 0 || ne <= heavy_min) return;
        const uint32_t j = atomicAdd(n_heavy, 1u);
        heavy_list[3 * j] = rs; heavy_list[3 * j + 1] = ns; heavy_list[3 * j + 2] = ne;
        b.n_seeds[rs] = 0;                               // the ordinary vote kernels see nothing to do
    };
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < n4; q += gridDim.x * 256) {
        const uint2 s4 = reinterpret_cast<const uint2*>(b.n_seeds)[q];
        const uint4 e4 = reinterpret_cast<const uint4*>(b.n_entries)[q];
        take(4 * q, s4.x & 0xFFFFu, e4.x); take(4 * q + 1, s4.x >> 16, e4.y); take(4 * q + 2, s4.y & 0xFFFFu, e4.z); take(4 * q + 3, s4.y >> 16, e4.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n2 & 3u)) { const uint32_t rs = 4 * n4 + threadIdx.x; take(rs, b.n_seeds[rs], b.n_entries[rs]); }
    if (sum_counters) {                                  // one set of atomics per workgroup of a small grid (they all hit one line)
        __shared__ unsigned long long s_a[4], s_e[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); e += __shfl_xor(e, off); }
        if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_e[threadIdx.x >> 6] = e; }
        __syncthreads();
        if (threadIdx.x == 0) {
            a = s_a[0] + s_a[1] + s_a[2] + s_a[3]; e = s_e[0] + s_e[1] + s_e[2] + s_e[3];
            if (a) {
                atomicAdd(&b.counters[GMK_KMERS], a); atomicAdd(&b.counters[GMK_TAB_LOOKUPS], a); atomicAdd(&b.counters[GMK_SEEDS], a);
                atomicAdd(&b.counters[GMK_SA_HITS], e);
            }
        }
    }
}

// one workgroup per heavy read x strand of the chunk [j0, j0 + nj)
__global__ void __launch_bounds__(256) k_heavy_expand(GmDevIndex ix, GmDevParams p, GmDevBatch b, int use_full_sa, const uint32_t* heavy_list, uint32_t j0,
                                                      const unsigned long long* key_off /* per chunk item */, unsigned long long* keys) {
    const uint32_t jl = blockIdx.x;
    const uint32_t rs = heavy_list[3 * (j0 + jl)];
    uint32_t ns = heavy_list[3 * (j0 + jl) + 1];
    if (p.nw && p.fast && ns > 1) ns = 1;
    const GmSeed* seeds = b.seeds + (size_t)rs * b.max_seeds;
    unsigned long long out = key_off[jl];
    unsigned long long flat = 0;                          // entry index inside the read x strand (sampled-SA mode: coords[] order)
    const uint32_t* coords = use_full_sa ? nullptr : b.coords + b.entry_off[rs];
    for (uint32_t t = 0; t < ns; ++t) {
        const GmSeed sd = seeds[t];
        const uint32_t cnt = sd.l - sd.k + 1;
        for (uint32_t e = threadIdx.x; e < cnt; e += 256) {
            const uint32_t c = use_full_sa ? ix.full_sa[sd.k + e] : coords[flat + e];
            const uint32_t bp = c <= sd.pos ? 0u : c - sd.pos;                               // :267-269
            keys[out + e] = ((unsigned long long)jl << 48) | ((unsigned long long)bp << 16) | (unsigned long long)t;
        }
        out += cnt; flat += cnt;
    }
    // --fast uses the first seed only while the hit count covers every seed: the rest of this item's key range is filled with keys
    // k_heavy_runs ignores (window start 0xFFFFFFFF never exists: coordinates are below 2^32 - 2)
    const unsigned long long end = key_off[jl] + heavy_list[3 * (j0 + jl) + 2];
    for (unsigned long long e = out + threadIdx.x; e < end; e += 256) keys[e] = ((unsigned long long)jl << 48) | (0xFFFFFFFFull << 16) | 0xFFFFull;
}

__global__ void __launch_bounds__(256) k_heavy_runs(GmDevBatch b, const uint32_t* heavy_list, uint32_t j0, const unsigned long long* keys, unsigned long long n,
                                                    uint32_t kmin, int nw) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool emit = false;
    uint32_t rs = 0, bp = 0, step = 0;
    if (i < n) {
        const unsigned long long k = keys[i];
        const unsigned long long grp = k >> 16;
        if (nw) {
            // element i is the kmin-th vote of its window start iff the kmin - 1 elements before it belong to the same run and the one
            // before those does not
            bool ok = i + 1 >= kmin;
            if (ok && kmin > 1) ok = (keys[i - (kmin - 1)] >> 16) == grp;
            if (ok && i >= kmin) ok = (keys[i - kmin] >> 16) != grp;
            emit = ok;
            step = (uint32_t)(k & 0xFFFFu);
        } else {
            const bool last = i + 1 == n || (keys[i + 1] >> 16) != grp;
            if (last) {
                unsigned long long cnt = 1;
                while (cnt <= i && cnt < 65536ull && (keys[i - cnt] >> 16) == grp) ++cnt;
                emit = cnt >= kmin;
                step = cnt > 65535ull ? 65535u : (uint32_t)cnt;
            }
        }
        if (emit) { rs = heavy_list[3 * (j0 + (uint32_t)(k >> 48))]; bp = (uint32_t)(k >> 16); emit = bp != 0xFFFFFFFFu; }
    }
    // one candidate reservation per wave
    const unsigned long long m = __builtin_amdgcn_ballot_w64(emit);
    if (m == 0ull) return;
    const uint32_t shard = (uint32_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) & (GM_NSHARD - 1);
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&b.shard_cnt[shard * GM_SHARD_STRIDE], (uint32_t)__popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (emit) {
        const uint32_t idx = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (idx < b.cand_region) {
            GmCand c;
            c.rs = rs; c.b = bp; c.step = (uint16_t)step; c.flags = 0; c.pad = 0; c.score = 0.0f;
            b.cands[(size_t)shard * b.cand_region + idx] = c;
        }
    }
}

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

int gmk_heavy_collect(const GmDevBatch& b, uint32_t heavy_min, uint32_t* n_heavy, uint32_t* heavy_list, int sum_counters, void* stream) {
    if (b.n == 0) return 0;
    hipLaunchKernelGGL(k_heavy_collect, dim3((uint32_t)std::min<unsigned long long>((2ull * b.n / 4 + 255) / 256 + 1, 1024ull)), dim3(256), 0, S_(stream), b, heavy_min, n_heavy, heavy_list, sum_counters);
    return (int)hipGetLastError();
}

size_t gmk_heavy_sort_temp_bytes(size_t n_keys) {
    size_t bytes = 0;
    rocprim::double_buffer<unsigned long long> kb(nullptr, nullptr);
    if (rocprim::radix_sort_keys(nullptr, bytes, kb, n_keys, 0u, 64u, (hipStream_t) nullptr) != hipSuccess) return 0;
    return bytes;
}

// expand + sort + runs for the chunk [j0, j0 + nj); keys0 / keys1: two buffers of n_keys keys; item_bits = bits the chunk index needs
int gmk_heavy_chunk(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, int use_full_sa, const uint32_t* heavy_list, uint32_t j0, uint32_t nj,
                    const unsigned long long* key_off, unsigned long long* keys0, unsigned long long* keys1, unsigned long long n_keys, void* tmp,
                    size_t tmp_bytes, unsigned item_bits, void* stream) {
    if (nj == 0 || n_keys == 0) return 0;
    hipLaunchKernelGGL(k_heavy_expand, dim3(nj), dim3(256), 0, S_(stream), ix, p, b, use_full_sa, heavy_list, j0, key_off, keys0);
    rocprim::double_buffer<unsigned long long> kb(keys0, keys1);
    // the step field is as wide as the seed count needs, then 32 bits of window start, then the chunk index
    hipError_t e = rocprim::radix_sort_keys(tmp, tmp_bytes, kb, (size_t)n_keys, 0u, 48u + item_bits, S_(stream));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_heavy_runs, dim3((uint32_t)((n_keys + 255) / 256)), dim3(256), 0, S_(stream), b, heavy_list, j0, kb.current(), n_keys,
                       (uint32_t)(p.kmin < 1 ? 1 : p.kmin), p.nw);
    return (int)hipGetLastError();
}
