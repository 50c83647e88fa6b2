// gm_band.hip — the banded DP kernels for -M / --max_gap other than the default 3 (gMAX_GAP, inc/const_define.h).
//
// The specialised kernels of gm_kernels.hip keep the 7-cell band row of -M 3 in registers.  These two do the same arithmetic for
// any band half-width G = 1 .. 7 with the band rows in LDS ([cell][lane], conflict-free) and run-time loop bounds: one lane per
// candidate / per kept sequence, the recurrences written exactly as the reference states them
//   k_nw_band         bin_seq::get_align_score_begin(read, gen, end = L)   src/bin_seq.cpp:781-850  (backward DP, score = nm[0][0])
//   k_traceback_band  bin_seq::get_align_score_w_traceback                 src/bin_seq.cpp:445-718  (forward DP + move matrix; the
//                     2-bit moves of a row, up to 17 cells, go to a 64-bit word per row in HBM as [row][item], coalesced)
// and the same epilogues as k_nw_lane (accept test, top score, hit count) and k_traceback_lane (packed operations, CIGAR length).
// Slower than the -M 3 kernels (LDS band, no unrolling) - a rarely used option pays for itself, the default path is untouched.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "gm_internal.h"

#define GB_NEG_INF (-100000.0f)
#define GB_NT 128
#define GB_WMAX 17                      // 2 * 7 + 3

namespace {

__device__ __forceinline__ uint32_t gb_nt4(uint32_t ch) { uint32_t u = ch & 0xDFu; return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 4u; }

__device__ __forceinline__ float gb_get_val(uint32_t code, float p, float q, const float* s) {     // get_val src/bin_seq.cpp:975-987, no FMA
    float r0 = code == 0 ? p : q, r1 = code == 1 ? p : q, r2 = code == 2 ? p : q, r3 = code == 3 ? p : q;
    float a = __fadd_rn(__fmul_rn(r0, s[0]), __fmul_rn(r1, s[1]));
    a = __fadd_rn(a, __fmul_rn(r2, s[2]));
    a = __fadd_rn(a, __fmul_rn(r3, s[3]));
    return a;
}

__device__ __forceinline__ uint32_t gb_pos2rid(const uint32_t* coff, uint32_t n_seqs, uint32_t pos) {
    uint32_t lo = 0, hi = n_seqs - 1;
    while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (pos >= coff[mid]) lo = mid; else hi = mid - 1; }
    return lo;
}

__device__ __forceinline__ bool gb_window_ok(const GmDevIndex& ix, uint32_t begin, uint32_t L) {      // GenomeBwt::GetString src/GenomeBwt.cpp:384-415
    if ((unsigned long long)begin + L > ix.l_pac) return false;
    return gb_pos2rid(ix.contig_off, ix.n_seqs, begin) == gb_pos2rid(ix.contig_off, ix.n_seqs, begin + L - 1);
}

__device__ __forceinline__ uint32_t gb_ref(const uint8_t* pac, uint32_t g) { return (pac[g >> 2] >> ((~g & 3u) << 1)) & 3u; }      // _get_pac src/bntseq.c:225

// PWM row i of the read in strand orientation against the four reference bases (reverse_comp_cpy SequenceOperations.h:149-161)
__device__ __forceinline__ void gb_row(const GmDevBatch& b, const float2* lut, const float (*sg)[4], uint32_t r, uint32_t L, uint32_t strand, uint32_t i, float* v4) {
    const uint32_t src = strand ? L - 1u - i : i;
    const uint32_t ch = b.bases[(size_t)r * b.stride + src], qc = b.quals[(size_t)r * b.stride + src];
    uint32_t code = gb_nt4(ch);
    if (strand && code < 4) code = 3 - code;
    const float2 pq = lut[qc];
#pragma unroll
    for (int g = 0; g < 4; ++g) v4[g] = gb_get_val(code, pq.x, pq.y, sg[g]);
}
__device__ __forceinline__ float gb_pick(const float* v4, uint32_t wc) { return (wc & 2u) ? ((wc & 1u) ? v4[3] : v4[2]) : ((wc & 1u) ? v4[1] : v4[0]); }

__device__ __forceinline__ float gb_max3(float a, float b, float c) { if (a >= b) return a >= c ? a : c; return b >= c ? b : c; }     // max_flt :1013-1026

__device__ __forceinline__ uint32_t gb_digits(uint32_t v) { return v >= 10000 ? 5u : v >= 1000 ? 4u : v >= 100 ? 3u : v >= 10 ? 2u : 1u; }

}  // namespace

__global__ void __launch_bounds__(GB_NT) k_nw_band(GmDevIndex ix, GmDevParams p, GmDevBatch b, uint32_t n_cands, int G) {
    __shared__ float s_row[2][GB_WMAX][GB_NT];
    __shared__ uint32_t s_pre[GM_NSHARD + 4];
    // prefix of the shard counts (the candidate list is 1024 bump-allocated regions)
    for (uint32_t q = threadIdx.x; q < GM_NSHARD; q += GB_NT) { uint32_t c = b.shard_cnt[q * GM_SHARD_STRIDE]; s_pre[q + 1] = c < b.cand_region ? c : b.cand_region; }
    if (threadIdx.x == 0) s_pre[0] = 0;
    __syncthreads();
    if (threadIdx.x == 0) for (uint32_t q = 1; q <= GM_NSHARD; ++q) s_pre[q] += s_pre[q - 1];
    __syncthreads();
    float sg[4][4];
    for (int g = 0; g < 4; ++g) for (int k = 0; k < 4; ++k) sg[g][k] = p.S256[(size_t)("acgt"[g]) * 4 + k];
    const int W = 2 * G + 3, tid = threadIdx.x;
    const float gap = p.gap;
    const uint32_t total = s_pre[GM_NSHARD];
    unsigned long long cells = 0, accepted = 0;
    for (uint32_t wi = blockIdx.x * GB_NT + tid; wi < total; wi += gridDim.x * GB_NT) {
        uint32_t lo = 0, hi = GM_NSHARD - 1;                               // shard of candidate wi
        while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (s_pre[mid] <= wi) lo = mid; else hi = mid - 1; }
        const size_t ci = (size_t)lo * b.cand_region + (wi - s_pre[lo]);
        GmCand c = b.cands[ci];
        const uint32_t r = c.rs >> 1, strand = c.rs & 1;
        if ((c.flags & 4) && b.rs_overflow[c.rs]) continue;                // superseded by the retry kernel
        const uint32_t L = b.len[r];
        const bool ok = gb_window_ok(ix, c.b, L);
        float result = 0.0f;
        if (ok && p.nw) {
            const float2* lut = p.lut + ((r < b.illumina_until) ? 256 : 0);
            const int end = (int)L;
            int cur = 0;                                                    // s_row[cur] = row i + 1 while row i is computed into s_row[cur ^ 1]
            // row `end`: nm[end][j] = gGAP * (end - j) for the last G + 2 columns (bin_seq.cpp:805-808); offset = j - i + G + 1
            for (int off = 0; off < W; ++off) {
                const int j = end + off - (G + 1);
                s_row[cur][off][tid] = (j >= 0 && j <= end && end - j <= G + 1) ? __fmul_rn(gap, (float)(unsigned)(end - j)) : GB_NEG_INF;
            }
            for (int i = end - 1; i >= 0; --i) {
                const int nx = cur ^ 1;
                for (int off = 0; off < W; ++off) s_row[nx][off][tid] = GB_NEG_INF;
                if (end - i <= G + 1) s_row[nx][end - i + G + 1][tid] = __fmul_rn(gap, (float)(unsigned)(end - i));      // last column nm[i][end]
                float v4[4];
                gb_row(b, lut, sg, r, L, strand, (uint32_t)i, v4);
                for (int j = i + G; j >= i - G; --j) {
                    if (j >= end) continue;
                    if (j < 0) break;
                    const int off = j - i + G + 1;
                    const float mm = __fadd_rn(s_row[cur][off][tid], gb_pick(v4, gb_ref(ix.pac, c.b + (uint32_t)j)));
                    const float g1 = __fadd_rn(s_row[cur][off - 1][tid], gap);
                    const float g2 = __fadd_rn(s_row[nx][off + 1][tid], gap);
                    s_row[nx][off][tid] = gb_max3(mm, g1, g2);
                    ++cells;
                }
                cur = nx;
            }
            result = s_row[cur][G + 1][tid];                               // nm[0][0]
        } else if (!p.nw) {
            result = (float)c.step;                                         // --no_nw: the score is the vote count (:70-76)
        }
        uint8_t fl = c.flags & 4;
        if (ok) {
            fl |= GMC_VALID;
            if (result > 0.0f) atomicMax(reinterpret_cast<int*>(&b.top_score[r]), __float_as_int(result));   // top_align_score (:95-98)
            if ((double)result >= b.min_score[r]) { fl |= GMC_ACCEPT; atomicAdd(&b.hit_count[r], 1u); ++accepted; }
        }
        b.cands[ci].score = result;
        b.cands[ci].flags = fl;
    }
    atomicAdd(&b.counters[GMK_NW_CELLS], cells);
    atomicAdd(&b.counters[GMK_ACCEPTED], accepted);
}

// forward DP of get_align_score_w_traceback for items [i0, i0 + ni); moves[(row) * ni + item] = 2 bits per band cell (0 D, 1 U, 2 L)
__global__ void __launch_bounds__(GB_NT) k_traceback_band(GmDevIndex ix, GmDevParams p, GmDevBatch b, const GmCand* items, uint32_t i0, uint32_t ni,
                                                          unsigned long long* moves, unsigned long long* ops, uint32_t ops_words, uint16_t* ops_len,
                                                          const uint8_t* emit, uint32_t* cig_cnt, uint32_t* max_span, int G) {
    __shared__ float s_row[2][GB_WMAX][GB_NT];
    float sg[4][4];
    for (int g = 0; g < 4; ++g) for (int k = 0; k < 4; ++k) sg[g][k] = p.S256[(size_t)("acgt"[g]) * 4 + k];
    const int W = 2 * G + 3, tid = threadIdx.x;
    const float gap = p.gap;
    const uint32_t it = blockIdx.x * GB_NT + tid;
    uint32_t my_span = 0;
    if (it < ni) {
        const uint32_t ci = i0 + it;
        const GmCand c = items[ci];
        const uint32_t r = c.rs >> 1, strand = c.rs & 1;
        const uint32_t L = b.len[r];
        const bool ok = L > 0 && gb_window_ok(ix, c.b, L) && 2 * L <= 32 * ops_words;
        uint16_t outlen = 0;
        uint32_t ctext = 1;                                                 // "*" when there is no path
        if (ok) {
            const float2* lut = p.lut + ((r < b.illumina_until) ? 256 : 0);
            const int N = (int)L;
            int cur = 0;
            // row 0: nm[0][j] = gGAP * j ('L') for j <= G + 1 (bin_seq.cpp:503-511); offset = j - i + G + 1
            for (int off = 0; off < W; ++off) { const int j = off - (G + 1); s_row[cur][off][tid] = (j >= 0 && j <= G + 1 && j <= N) ? __fmul_rn(gap, (float)j) : GB_NEG_INF; }
            for (int i = 1; i <= N; ++i) {
                const int nx = cur ^ 1;
                for (int off = 0; off < W; ++off) s_row[nx][off][tid] = GB_NEG_INF;
                if (i <= G + 1) s_row[nx][G + 1 - i][tid] = __fmul_rn(gap, (float)i);          // first column nm[i][0] ('U')
                unsigned long long mrow = 0;
                float v4[4];
                gb_row(b, lut, sg, r, L, strand, (uint32_t)(i - 1), v4);
                for (int j = i - G; j <= i + G; ++j) {
                    if (j <= 0) continue;
                    if (j > N) break;
                    const int off = j - i + G + 1;
                    const float d = __fadd_rn(s_row[cur][off][tid], gb_pick(v4, gb_ref(ix.pac, c.b + (uint32_t)(j - 1))));
                    const float u = __fadd_rn(s_row[cur][off + 1][tid], gap);
                    const float l = __fadd_rn(s_row[nx][off - 1][tid], gap);
                    uint32_t m; float best;                                 // max_flt(char&,...) src/bin_seq.cpp:989-1011
                    if (d >= u) { if (d >= l) { m = 0; best = d; } else { m = 2; best = l; } }
                    else        { if (u >= l) { m = 1; best = u; } else { m = 2; best = l; } }
                    s_row[nx][off][tid] = best;
                    mrow |= (unsigned long long)m << (2 * off);
                }
                moves[(size_t)i * ni + it] = mrow;
                cur = nx;
            }
            // walk back from (L, N); cells outside the computed band cannot be reached (they are NEG_INF neighbours), the first row /
            // column end the walk
            unsigned long long* out = ops + (size_t)ci * ops_words;
            int i = N, j = N, nops = 0;
            while (i != 0 && j != 0) {
                const uint32_t m = (uint32_t)(moves[(size_t)i * ni + it] >> (2 * (j - i + G + 1))) & 3u;
                if (m == 0) { --i; --j; } else if (m == 1) { --i; } else { --j; }
                ++nops;
            }
            nops += i + j;
            int k = nops - 1;
            unsigned long long curw = 0;
            uint32_t run_code = 3, run_len = 0, text = 0; bool first_run = true;
            auto push = [&](uint32_t code) {
                curw |= (unsigned long long)code << (2 * (k & 31));
                if ((k & 31) == 0) { out[k >> 5] = curw; curw = 0; }
                --k;
                if (code == run_code) { ++run_len; return; }
                if (run_len) { if (!(first_run && run_code == 2)) text += gb_digits(run_len) + 1; first_run = false; }
                run_code = code; run_len = 1;
            };
            i = N; j = N;
            while (i != 0 && j != 0) {
                const uint32_t m = (uint32_t)(moves[(size_t)i * ni + it] >> (2 * (j - i + G + 1))) & 3u;
                push(m);
                if (m == 0) { --i; --j; } else if (m == 1) { --i; } else { --j; }
            }
            while (i > 0) { push(1); --i; }
            while (j > 0) { push(2); --j; }
            if (run_len && !(first_run && run_code == 2)) text += gb_digits(run_len) + 1;
            outlen = (uint16_t)nops;
            if (nops) ctext = text;
        }
        ops_len[ci] = outlen;
        my_span = outlen;
        if (cig_cnt) cig_cnt[ci] = (emit && !emit[ci]) ? 0u : (p.nw ? ctext : gb_digits(L) + 1u) + 1u;
    }
    if (max_span) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(my_span, off); my_span = o > my_span ? o : my_span; }
        if ((tid & 63) == 0 && my_span) atomicMax(max_span, my_span);
    }
}

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

int gmk_nw_band(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, uint32_t n_cands, int G, void* stream) {
    if (b.n == 0) return 0;
    const uint32_t grid = std::min<uint32_t>(16384u, std::max<uint32_t>(256u, n_cands / 512u));
    hipLaunchKernelGGL(k_nw_band, dim3(grid), dim3(GB_NT), 0, S_(stream), ix, p, b, n_cands, G);
    return (int)hipGetLastError();
}

// the move words of a launch live in b.band_moves ([row][item] for the launch's items): as many items per launch as fit
int gmk_traceback_band(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, const GmCand* items, uint32_t n, unsigned long long* ops,
                       uint32_t ops_words, uint16_t* ops_len, const uint8_t* emit, uint32_t* cig_cnt, uint32_t* max_span, int G, void* stream) {
    if (n == 0) return 0;
    const uint64_t rows = (uint64_t)b.stride + 1;
    const uint64_t per = b.band_moves ? b.band_moves_words / rows : 0;
    if (per == 0) return (int)hipErrorInvalidValue;
    for (uint64_t i0 = 0; i0 < n; i0 += per) {
        const uint32_t ni = (uint32_t)std::min<uint64_t>(per, n - i0);
        hipLaunchKernelGGL(k_traceback_band, dim3((ni + GB_NT - 1) / GB_NT), dim3(GB_NT), 0, S_(stream), ix, p, b, items, (uint32_t)i0, ni, b.band_moves, ops,
                           ops_words, ops_len, emit, cig_cnt, max_span, G);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}
