// tools/ubench/gather.hip — what a random record fetch costs on MI355X by record size (design input for the bucket table of
// k_vote_bucket): every wave fetches random records of R = 16 * LPR bytes (LPR lanes x 16 B) from a table far larger than the
// Infinity Cache; records are aligned to ALIGN bytes (ALIGN >= R: the rest of the ALIGN block is never touched).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gather.hip -o /tmp/gather && /tmp/gather [table GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}

template <int LPR, int U>
__global__ void __launch_bounds__(64) k_gather(const uint4* __restrict__ tab, uint64_t n_blocks /* ALIGN-sized blocks */, uint32_t align16 /* ALIGN / 16 */, int iters, uint32_t* out) {
    const int lane = threadIdx.x;
    const uint32_t grp = lane / LPR, sub = lane % LPR;
    constexpr int GPW = 64 / LPR;                   // records per wave-instruction
    uint32_t acc = 0;
    uint64_t s = mix(((uint64_t)blockIdx.x << 20) ^ 0x1234567ull);
    for (int it = 0; it < iters; ++it) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t h = mix(s + (uint64_t)(it * U + u) * GPW + grp);
            const uint64_t blk = h % n_blocks;
            v[u] = tab[blk * align16 + sub];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x9e3779b9u) out[0] = acc;           // keeps the loads alive
}

template <int LPR, int U>
static void run(const uint4* tab, uint64_t bytes, uint32_t align, int iters, uint32_t* out, const char* label) {
    const uint64_t n_blocks = bytes / align;
    const int grid = 256 * 32 * 8;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_gather<LPR, U>), dim3(grid), dim3(64), 0, 0, tab, n_blocks, align / 16, 2, out);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL((k_gather<LPR, U>), dim3(grid), dim3(64), 0, 0, tab, n_blocks, align / 16, iters, out);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms = 0; CHK(hipEventElapsedTime(&ms, a, b));
    const double recs = (double)grid * iters * U * (64 / LPR);
    printf("%-44s R=%3d B align=%3u: %8.3f ms  %7.2f G records/s  useful %7.1f GB/s  aligned-block %7.1f GB/s\n", label, 16 * LPR, align, ms, recs / ms / 1e6,
           recs * 16 * LPR / ms / 1e6, recs * align / ms / 1e6);
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 32.0;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) & ~0xFFFull;
    uint4* tab; uint32_t* out;
    CHK(hipMalloc(&tab, bytes)); CHK(hipMalloc(&out, 64));
    CHK(hipMemset(tab, 1, bytes)); CHK(hipMemset(out, 0, 64));
    printf("table %.1f GiB\n", gib);
    const int it = 64;
    run<8, 4>(tab, bytes, 128, it, out, "8 lanes x 16 B = whole 128-B line");
    run<4, 4>(tab, bytes, 128, it * 2 / 2, out, "4 lanes x 16 B = first half of a 128-B line");
    run<4, 4>(tab, bytes, 64, it, out, "4 lanes x 16 B = 64-B records, packed");
    run<2, 4>(tab, bytes, 128, it, out, "2 lanes x 16 B = first 32 B of a 128-B line");
    run<2, 4>(tab, bytes, 32, it, out, "2 lanes x 16 B = 32-B records, packed");
    run<1, 4>(tab, bytes, 128, it, out, "1 lane x 16 B per 128-B line (table probe)");
    run<1, 4>(tab, bytes, 16, it, out, "1 lane x 16 B, packed");
    run<8, 8>(tab, bytes, 128, it / 2, out, "8 lanes x 16 B, 8 loads in flight per lane");
    run<4, 8>(tab, bytes, 64, it / 2, out, "4 lanes x 16 B packed, 8 loads in flight");
    run<8, 2>(tab, bytes, 128, it * 2, out, "8 lanes x 16 B, 2 loads in flight per lane");
    run<8, 4>(tab, bytes, 256, it, out, "8 lanes x 16 B, one line of every two");
    return 0;
}
