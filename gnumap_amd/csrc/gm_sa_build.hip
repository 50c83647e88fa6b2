// gm_sa_build.hip — suffix array + BWT of the forward strand on the MI355X (the index builder).
//
// Replaces the SA stage of bwa_index (is_sa / is_bwt, src/is.c:53-223, called from bwt_pac2bwt src/bwtindex.c:60-104) with an
// HBM-resident prefix-doubling sort: the whole problem (keys, ranks, positions: ~29 bytes per base, 90 GB for a human-size
// reference) fits the 288 GB of one MI355X, so every round is a streaming pass + one radix sort over all suffixes:
//
//   round 0   key[i] = first 21 symbols of suffix i, 3 bits each ('$' = 0 < A..T = 1..4, so a shorter suffix sorts first)
//   round r   key[i] = (rank[i] << 32) | (rank[i + h] + 1, or 0 past the end), h = 21, 42, 84, ...   (Manber-Myers doubling)
//   rank      = index of the first suffix of the group in sorted order (head flags -> max-scan -> scatter by position)
//   stop      when every group is a single suffix
//
// The sorts are rocPRIM's device radix sort (a library primitive, like a library GEMM); the key builders, head flags, rank
// scatter, BWT gather, primary search and SA sampling are the kernels below.  The result is the same suffix array as the
// host SA-IS builder in gm_index.cpp produces (suffix arrays are unique), so the index files stay byte-identical.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/gnumap_hip.h"
#include "gm_host.h"

namespace {

constexpr int SA_T = 256;             // threads per workgroup
constexpr int SA_K0 = 21;             // symbols in the round-0 key

inline uint32_t sa_grid(uint64_t n) {
    uint64_t g = (n + SA_T - 1) / SA_T;
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(g, 1), 256u * 64u);       // grid-stride: <= 64 workgroups per CU
}

__global__ void __launch_bounds__(SA_T) k_sa_init_keys(const uint8_t* __restrict__ codes, uint64_t n, uint64_t* __restrict__ keys,
                                                       uint32_t* __restrict__ vals) {
    for (uint64_t i = (uint64_t)blockIdx.x * SA_T + threadIdx.x; i < n; i += (uint64_t)gridDim.x * SA_T) {
        uint64_t k = 0;
#pragma unroll
        for (int t = 0; t < SA_K0; ++t) {
            const uint64_t c = i + t < n ? codes[i + t] : 0;
            k = (k << 3) | c;
        }
        keys[i] = k;
        vals[i] = (uint32_t)i;
    }
}

// head[j] = j where a new group starts in sorted order, else 0; *n_groups += number of groups
__global__ void __launch_bounds__(SA_T) k_sa_heads(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ head,
                                                   unsigned long long* __restrict__ n_groups) {
    unsigned long long cnt = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * SA_T + threadIdx.x; j < n; j += (uint64_t)gridDim.x * SA_T) {
        const bool h = j == 0 || keys[j] != keys[j - 1];
        head[j] = h ? (uint32_t)j : 0u;
        cnt += h;
    }
    for (int o = 32; o; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(n_groups, cnt);
}

__global__ void __launch_bounds__(SA_T) k_sa_scatter_rank(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ grp, uint64_t n,
                                                          uint32_t* __restrict__ rank) {
    for (uint64_t j = (uint64_t)blockIdx.x * SA_T + threadIdx.x; j < n; j += (uint64_t)gridDim.x * SA_T) rank[sa[j]] = grp[j];
}

__global__ void __launch_bounds__(SA_T) k_sa_next_keys(const uint32_t* __restrict__ rank, uint64_t n, uint64_t h, uint64_t* __restrict__ keys,
                                                       uint32_t* __restrict__ vals) {
    for (uint64_t i = (uint64_t)blockIdx.x * SA_T + threadIdx.x; i < n; i += (uint64_t)gridDim.x * SA_T) {
        const uint64_t lo = i + h < n ? (uint64_t)rank[i + h] + 1u : 0u;
        keys[i] = ((uint64_t)rank[i] << 32) | lo;
        vals[i] = (uint32_t)i;
    }
}

__global__ void __launch_bounds__(SA_T) k_sa_find_zero(const uint32_t* __restrict__ sa, uint64_t n, unsigned long long* __restrict__ where) {
    for (uint64_t j = (uint64_t)blockIdx.x * SA_T + threadIdx.x; j < n; j += (uint64_t)gridDim.x * SA_T)
        if (sa[j] == 0) *where = j;
}

// BWT with '$' removed, 16 symbols per word, first symbol in the top bits (is_bwt src/is.c:208-223 + the packing loop of
// bwt_pac2bwt src/bwtindex.c:84-99).  Entry 0 of the (n+1)-row matrix is the '$' suffix (BWT symbol = last base); sorted suffix j
// is row j + 1; the row of suffix 0 (the primary) is dropped.
__global__ void __launch_bounds__(SA_T) k_sa_bwt(const uint8_t* __restrict__ codes, const uint32_t* __restrict__ sa, uint64_t n, uint64_t jprim,
                                                 uint32_t* __restrict__ plain, uint64_t n_words) {
    for (uint64_t w = (uint64_t)blockIdx.x * SA_T + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * SA_T) {
        uint32_t word = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint64_t o = w * 16 + t;
            if (o < n) {
                uint32_t c;
                if (o == 0) c = codes[n - 1];
                else {
                    uint64_t j = o - 1;
                    if (j >= jprim) j = o;
                    c = codes[(uint64_t)sa[j] - 1];
                }
                word |= (c - 1u) << ((15 - t) << 1);
            }
        }
        plain[w] = word;
    }
}

// sa[s * intv] of the (n+1)-row array for s = 1 .. n_sa-1 (bwt_cal_sa src/bwt.c:62-84 keeps every intv-th rank)
__global__ void __launch_bounds__(SA_T) k_sa_samples(const uint32_t* __restrict__ sa, uint64_t intv, uint64_t n_samples, uint64_t* __restrict__ out) {
    for (uint64_t s = (uint64_t)blockIdx.x * SA_T + threadIdx.x; s < n_samples; s += (uint64_t)gridDim.x * SA_T)
        out[s] = sa[(s + 1) * intv - 1];
}

struct DevMem {
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace

bool gm_device_available() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

// codes: n symbols in 1..4 (host).  Fills plain ((n+15)/16 words), primary (row of suffix 0 in the (n+1)-row matrix) and
// samples (n_sa - 1 values, sa[intv], sa[2 intv], ...).
int gm_device_sa_build(const uint8_t* codes, uint64_t n, int device_id, uint32_t intv, std::vector<uint32_t>& plain, uint64_t& primary,
                       std::vector<uint64_t>& samples, int* rounds_out, std::string& err) {
    if (n == 0 || n >= 0xFFFFFFFEull) { err = "reference too long for 32-bit ranks"; return GM_E_UNSUPPORTED; }
    if (hipSetDevice(device_id) != hipSuccess) { err = "hipSetDevice failed"; return GM_E_NO_DEVICE; }
    const uint64_t n_words = (n + 15) >> 4, n_sa = (n + intv) / intv, n_samples = n_sa - 1;
    DevMem d_codes, d_k0, d_k1, d_v0, d_v1, d_rank, d_tmp, d_cnt, d_small;
    size_t tmp_sort = 0, tmp_scan = 0;
    {
        rocprim::double_buffer<uint64_t> kb(nullptr, nullptr);
        rocprim::double_buffer<uint32_t> vb(nullptr, nullptr);
        if (rocprim::radix_sort_pairs(nullptr, tmp_sort, kb, vb, (size_t)n, 0u, 64u, (hipStream_t) nullptr) != hipSuccess ||
            rocprim::inclusive_scan(nullptr, tmp_scan, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, rocprim::maximum<uint32_t>(),
                                    (hipStream_t) nullptr) != hipSuccess) {
            err = "rocPRIM temporary-storage query failed";
            return GM_E_HIP;
        }
    }
    const size_t small_bytes = std::max<size_t>(n_words * 4, n_samples * 8);
    if (!d_codes.alloc(n) || !d_k0.alloc(n * 8) || !d_k1.alloc(n * 8) || !d_v0.alloc(n * 4) || !d_v1.alloc(n * 4) || !d_rank.alloc(n * 4) ||
        !d_tmp.alloc(std::max(tmp_sort, tmp_scan)) || !d_cnt.alloc(16) || !d_small.alloc(small_bytes)) {
        err = "out of HBM for the suffix array construction (" + std::to_string((29 * n) >> 20) + " MiB needed)";
        return GM_E_NOMEM;
    }
    auto bad = [&](const char* what) { err = std::string(what) + ": " + hipGetErrorString(hipGetLastError()); return GM_E_HIP; };
    if (hipMemcpy(d_codes.p, codes, n, hipMemcpyHostToDevice) != hipSuccess) return bad("codes upload");
    const uint32_t grid = sa_grid(n);
    rocprim::double_buffer<uint64_t> keys(d_k0.as<uint64_t>(), d_k1.as<uint64_t>());
    rocprim::double_buffer<uint32_t> vals(d_v0.as<uint32_t>(), d_v1.as<uint32_t>());
    size_t tmp_bytes = std::max(tmp_sort, tmp_scan);
    hipLaunchKernelGGL(k_sa_init_keys, dim3(grid), dim3(SA_T), 0, nullptr, d_codes.as<uint8_t>(), n, keys.current(), vals.current());
    if (rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, keys, vals, (size_t)n, 0u, (unsigned)(3 * SA_K0), (hipStream_t) nullptr) != hipSuccess)
        return bad("radix sort");
    int rounds = 0;
    for (uint64_t h = SA_K0;; h *= 2, ++rounds) {
        unsigned long long groups = 0;
        if (hipMemsetAsync(d_cnt.p, 0, 16, nullptr) != hipSuccess) return bad("memset");
        uint32_t* head = vals.alternate();                                 // both alternates are scratch between two sorts
        uint32_t* grp = reinterpret_cast<uint32_t*>(keys.alternate());
        hipLaunchKernelGGL(k_sa_heads, dim3(grid), dim3(SA_T), 0, nullptr, keys.current(), n, head, d_cnt.as<unsigned long long>());
        if (hipMemcpy(&groups, d_cnt.p, 8, hipMemcpyDeviceToHost) != hipSuccess) return bad("group count");
        if (groups == n) break;
        if (h >= 2 * n + 64) { err = "prefix doubling did not converge"; return GM_E_HIP; }
        tmp_bytes = std::max(tmp_sort, tmp_scan);
        if (rocprim::inclusive_scan(d_tmp.p, tmp_bytes, head, grp, (size_t)n, rocprim::maximum<uint32_t>(), (hipStream_t) nullptr) != hipSuccess)
            return bad("scan");
        hipLaunchKernelGGL(k_sa_scatter_rank, dim3(grid), dim3(SA_T), 0, nullptr, vals.current(), grp, n, d_rank.as<uint32_t>());
        hipLaunchKernelGGL(k_sa_next_keys, dim3(grid), dim3(SA_T), 0, nullptr, d_rank.as<uint32_t>(), n, h, keys.current(), vals.current());
        tmp_bytes = std::max(tmp_sort, tmp_scan);
        if (rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, keys, vals, (size_t)n, 0u, 64u, (hipStream_t) nullptr) != hipSuccess) return bad("radix sort");
    }
    if (rounds_out) *rounds_out = rounds;
    const uint32_t* sa = vals.current();
    unsigned long long jprim = ~0ull;
    if (hipMemcpy(d_cnt.p, &jprim, 8, hipMemcpyHostToDevice) != hipSuccess) return bad("primary init");
    hipLaunchKernelGGL(k_sa_find_zero, dim3(grid), dim3(SA_T), 0, nullptr, sa, n, d_cnt.as<unsigned long long>());
    if (hipMemcpy(&jprim, d_cnt.p, 8, hipMemcpyDeviceToHost) != hipSuccess || jprim == ~0ull) return bad("primary search");
    primary = jprim + 1;
    hipLaunchKernelGGL(k_sa_bwt, dim3(sa_grid(n_words)), dim3(SA_T), 0, nullptr, d_codes.as<uint8_t>(), sa, n, (uint64_t)jprim, d_small.as<uint32_t>(),
                       n_words);
    plain.resize(n_words);
    if (hipMemcpy(plain.data(), d_small.p, n_words * 4, hipMemcpyDeviceToHost) != hipSuccess) return bad("bwt download");
    samples.resize(n_samples);
    if (n_samples) {
        hipLaunchKernelGGL(k_sa_samples, dim3(sa_grid(n_samples)), dim3(SA_T), 0, nullptr, sa, (uint64_t)intv, n_samples, d_small.as<uint64_t>());
        if (hipMemcpy(samples.data(), d_small.p, n_samples * 8, hipMemcpyDeviceToHost) != hipSuccess) return bad("sample download");
    }
    if (hipDeviceSynchronize() != hipSuccess) return bad("suffix array construction");
    return GM_OK;
}
