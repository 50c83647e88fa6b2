// gm_snp.hip — the --snp deposit of SNPScoredSeq::score (src/SNPScoredSeq.cpp:25-109): per kept sequence the pair HMM of the read (in the
// orientation of the sequence's first strand, with its argmax consensus) against the reference window - bin_seq::pairHMM,
// src/bin_seq.cpp:60-244 - gives 5 floats per window position (posterior weight of a, c, g, t, n); every place of the sequence then adds
// total_score to the coverage of its L positions (AddScore) and hmm[i][base] * total_score to the five per-nucleotide tracks
// (AddSeqScore GenomeBwt.cpp:496-551, plain build), places on the other strand with reverse_comp_cpy_phmm (SequenceOperations.h:164-181).
//
// Bit-exactness decides the shape of the kernel: the reference's three (n+1) x (m+1) x 3 matrices are fp64, its transition constants
// floats (products like PHMM_q * PHMM_Tmg are FLOAT products), the emission p_seq a float sum of float products, and the result rows
// take `float += double` one read position at a time.  Every cell is computed with exactly those types and that operation order (the
// library is built with -ffp-contract=off); the Y state is a recurrence along the row, so a row cannot be cut over lanes without
// changing the rounding - ONE LANE walks one kept sequence, 64 sequences per wavefront, the forward matrix of a lane in HBM scratch
// interleaved lane by lane ([cell][64 lanes]: every access of the wavefront is one 512-byte stretch).  The backward sweep keeps two
// rows, forms the posterior of a cell where it stands and stores it over the forward value it consumed; a last pass adds a window
// position's column up in read order.  This mode is a correctness path, not a throughput path (the reference spends ~0.3 ms of one
// core on a 100 x 100 pair HMM).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "gm_internal.h"
#include "gm_device.h"

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

// gPHMM_ALIGN_SCORES[window char][read base] (inc/a_matrices.c:92-120, inc/const_define.h:79-81); windows are acgt
__device__ __forceinline__ float gs_score(uint32_t g, uint32_t x) { return g == x ? 0.98f : ((g ^ x) == 2u ? 0.01f : 0.005f); }

// bin_seq::p_seq src/bin_seq.cpp:41-57: float sum += x[k] * pam_p(k, y) for k = a, c, g, t, then 3 * sum
__device__ __forceinline__ float gs_pseq(const float row[4], uint32_t g) {
    float sum = __fmul_rn(row[0], gs_score(g, 0u));
    sum = __fadd_rn(sum, __fmul_rn(row[1], gs_score(g, 1u)));
    sum = __fadd_rn(sum, __fmul_rn(row[2], gs_score(g, 2u)));
    sum = __fadd_rn(sum, __fmul_rn(row[3], gs_score(g, 3u)));
    return __fmul_rn(3.0f, sum);
}

// PWM row i of read r in the orientation of `fs` (reverse_comp_cpy for the reverse strand) and its argmax consensus code
// (ScoredSeq::max_char ScoredSeq.h:72-103: 'n' = 4 when all four are equal)
__device__ __forceinline__ uint32_t gs_row(const GmDevBatch& b, const float2* lut, uint32_t r, uint32_t L, uint32_t fs, uint32_t i, float row[4]) {
    const uint32_t src = fs ? L - 1u - i : i;
    const uint8_t ch = b.bases[(size_t)r * b.stride + src], q = b.quals[(size_t)r * b.stride + src];
    const float2 pq = lut[q];
    uint32_t code = gm_nt4(ch);
    if (fs && code < 4u) code = 3u - code;
    row[0] = row[1] = row[2] = row[3] = pq.y;
    if (code == 0u) row[0] = pq.x; else if (code == 1u) row[1] = pq.x; else if (code == 2u) row[2] = pq.x; else if (code == 3u) row[3] = pq.x;
    const float* c = row;
    if (c[0] == c[1] && c[0] == c[2] && c[0] == c[3]) return 4u;
    if (c[0] >= c[1]) { if (c[0] >= c[2]) return c[0] >= c[3] ? 0u : 3u; return c[2] >= c[3] ? 2u : 3u; }
    if (c[1] >= c[2]) return c[1] >= c[3] ? 1u : 3u;
    return c[2] >= c[3] ? 2u : 3u;
}

__device__ __forceinline__ uint32_t gs_base(const GmDevIndex& ix, uint32_t g) { return (uint32_t)(ix.pac[g >> 2] >> ((~g & 3u) << 1)) & 3u; }

// items[k] = { rs = read * 2 + first strand, b = window start }; hmm[k][Lmax][5]; scratch: per wavefront cells_wave doubles
__global__ void __launch_bounds__(64) k_pair_hmm(GmDevIndex ix, GmDevParams p, GmDevBatch b, const GmCand* items, uint32_t n, double* scratch, size_t cells_wave,
                                                 uint32_t Lmax, float* hmm) {
    const uint32_t k = blockIdx.x * 64 + threadIdx.x, lane = threadIdx.x;
    if (k >= n) return;
    const GmCand it = items[k];
    const uint32_t r = it.rs >> 1, fs = it.rs & 1u, L = b.len[r], w0 = it.b;
    float* const out = hmm + (size_t)k * Lmax * 5u;
    if (L == 0u || L > Lmax) return;
    const float2* lut = p.lut + ((r < b.illumina_until) ? 256 : 0);
    // inc/bin_seq.h:49-69, floats
    const float PH_q = 0.25f, PH_t = 0.05f, PH_d = 0.0025f, PH_e = 0.5f;
    const float Tmm = __fsub_rn(__fsub_rn(1.0f, __fmul_rn(2.0f, PH_d)), PH_t), Tgm = __fsub_rn(__fsub_rn(1.0f, PH_d), PH_t), Tmg = PH_d, Tgg = PH_e;
    const float qTmg = __fmul_rn(PH_q, Tmg), qTgg = __fmul_rn(PH_q, Tgg);
    // the lane's slices of the wavefront's scratch: F[(i, j)][3] for i, j in 0 .. L; two backward rows {M, X} per column; consensus codes
    const uint32_t W = Lmax + 1u;
    double* const base = scratch + (size_t)blockIdx.x * cells_wave;
    auto F = [&](uint32_t i, uint32_t j, uint32_t s) -> double& { return base[((size_t)(i * W + j) * 3u + s) * 64u + lane]; };
    double* const rows = base + (size_t)W * W * 3u * 64u;
    auto R = [&](uint32_t which, uint32_t j, uint32_t s) -> double& { return rows[((size_t)(which * W + j) * 2u + s) * 64u + lane]; };
    unsigned char* const codes = reinterpret_cast<unsigned char*>(rows + (size_t)2u * W * 2u * 64u);
    // ---- forward (bin_seq.cpp:140-157); row 0 and column 0 are the zeros of its memset, f[0][0].M = 1 ----
    for (uint32_t j = 0; j <= L; ++j) { F(0, j, 0) = j == 0u ? 1.0 : 0.0; F(0, j, 1) = 0.0; F(0, j, 2) = 0.0; }
    for (uint32_t i = 1; i <= L; ++i) {
        float row[4];
        codes[(size_t)(i - 1u) * 64u + lane] = (unsigned char)gs_row(b, lut, r, L, fs, i - 1u, row);
        F(i, 0, 0) = 0.0; F(i, 0, 1) = 0.0; F(i, 0, 2) = 0.0;
        double dM = F(i - 1, 0, 0), dX = F(i - 1, 0, 1), dY = F(i - 1, 0, 2);        // f[i-1][j-1]
        double lM = 0.0, lY = 0.0;                                                      // f[i][j-1]
        for (uint32_t j = 1; j <= L; ++j) {
            const double uM = F(i - 1, j, 0), uX = F(i - 1, j, 1), uY = F(i - 1, j, 2);
            const float ps = gs_pseq(row, gs_base(ix, w0 + j - 1u));
            const double M = __dmul_rn((double)ps, __dadd_rn(__dadd_rn(__dmul_rn((double)Tmm, dM), __dmul_rn((double)Tgm, dX)), __dmul_rn((double)Tgm, dY)));
            const double X = __dmul_rn((double)PH_q, __dadd_rn(__dmul_rn((double)Tmg, uM), __dmul_rn((double)Tgg, uX)));
            const double Y = __dmul_rn((double)PH_q, __dadd_rn(__dmul_rn((double)Tmg, lM), __dmul_rn((double)Tgg, lY)));
            F(i, j, 0) = M; F(i, j, 1) = X; F(i, j, 2) = Y;
            dM = uM; dX = uX; dY = uY; lM = M; lY = Y;
        }
    }
    const double fE = __dmul_rn((double)PH_t, __dadd_rn(__dadd_rn(F(L, L, 0), F(L, L, 1)), F(L, L, 2)));
    // ---- backward (:162-207) with the posterior (:209-219) formed in place: p[i][j] = f[i+1][j+1] * b[i][j] / fE over f[i+1][j+1] ----
    for (int i = (int)L - 1; i >= 0; --i) {
        const uint32_t cur = (uint32_t)i & 1u, nxt = cur ^ 1u;                        // R(nxt, ..) = row i + 1
        float row[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        if ((uint32_t)i + 1u < L) gs_row(b, lut, r, L, fs, (uint32_t)i + 1u, row);       // r.pwm[i+1]
        double nM = 0.0, nY = 0.0;                                                      // b[i][j+1]
        for (int j = (int)L - 1; j >= 0; --j) {
            double bM, bX, bY;
            if (j == (int)L - 1 && i == (int)L - 1) { bM = bX = bY = (double)PH_t; }
            else if (j == (int)L - 1) { const double xb = R(nxt, (uint32_t)j, 1); bM = __dmul_rn((double)qTmg, xb); bX = __dmul_rn((double)qTgg, xb); bY = 0.0; }
            else if (i == (int)L - 1) { bM = __dmul_rn((double)qTmg, nY); bY = __dmul_rn((double)qTgg, nY); bX = 0.0; }
            else {
                const float ps = gs_pseq(row, gs_base(ix, w0 + (uint32_t)j + 1u));
                const double mb = R(nxt, (uint32_t)j + 1u, 0), xb = R(nxt, (uint32_t)j, 1);
                bM = __dadd_rn(__dadd_rn(__dmul_rn((double)__fmul_rn(ps, Tmm), mb), __dmul_rn((double)qTmg, xb)), __dmul_rn((double)qTmg, nY));
                bX = __dadd_rn(__dmul_rn((double)__fmul_rn(ps, Tgm), mb), __dmul_rn((double)qTgg, xb));
                bY = __dadd_rn(__dmul_rn((double)__fmul_rn(ps, Tgm), mb), __dmul_rn((double)qTgg, nY));
            }
            R(cur, (uint32_t)j, 0) = bM; R(cur, (uint32_t)j, 1) = bX;
            nM = bM; nY = bY;
            const uint32_t fi = (uint32_t)i + 1u, fj = (uint32_t)j + 1u;
            F(fi, fj, 0) = __ddiv_rn(__dmul_rn(F(fi, fj, 0), bM), fE);
            F(fi, fj, 1) = __ddiv_rn(__dmul_rn(F(fi, fj, 1), bX), fE);
            F(fi, fj, 2) = __ddiv_rn(__dmul_rn(F(fi, fj, 2), bY), fE);
        }
        (void)nM;
    }
    // ---- pGenScore[window position][g_gen_CONVERSION[consensus[read position]]] += pY + pM, read positions in order (:222-241) ----
    for (uint32_t ig = 0; ig < L; ++ig) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, a4 = 0.0f;
        for (uint32_t jr = 0; jr < L; ++jr) {
            const double v = __dadd_rn(F(jr + 1u, ig + 1u, 2), F(jr + 1u, ig + 1u, 0));
            const uint32_t c = codes[(size_t)jr * 64u + lane];
            if (c == 0u) a0 = (float)__dadd_rn((double)a0, v); else if (c == 1u) a1 = (float)__dadd_rn((double)a1, v);
            else if (c == 2u) a2 = (float)__dadd_rn((double)a2, v); else if (c == 3u) a3 = (float)__dadd_rn((double)a3, v);
            else a4 = (float)__dadd_rn((double)a4, v);
        }
        float* o = out + (size_t)ig * 5u;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4;
    }
}

// one workgroup per kept sequence of the chunk: every place of it, every base t < L
__global__ void __launch_bounds__(128) k_snp_deposit(float* cov, float* nuc, uint64_t bins, uint32_t bin_size, GmDevBatch b, const GmDevMatch* matches,
                                                     const GmDevPos* positions, uint32_t m0, const float* post, const float* hmm, uint32_t Lmax) {
    const uint32_t m = m0 + blockIdx.x;
    const GmDevMatch mm = matches[m];
    const uint32_t r = mm.read - b.read_base, L = b.len[r];
    const float w = post[m];                                            // (float)total_score: AddScore's and AddSeqScore's float argument
    const float* h = hmm + (size_t)blockIdx.x * Lmax * 5u;
    for (uint32_t q = mm.pos_begin; q < mm.pos_end; ++q) {
        const GmDevPos pp = positions[q];
        const bool same = pp.strand == mm.first_strand;
        for (uint32_t t = threadIdx.x; t < L; t += blockDim.x) {
            const uint64_t bin = (pp.pos + t) / bin_size;
            if (bin >= bins) continue;
            atomicAdd(&cov[bin], w);
            const float* s5 = h + (size_t)(same ? t : L - 1u - t) * 5u;
            // reverse_comp_cpy_phmm: the row of the mirrored position with a <-> t, c <-> g swapped, n kept
            const float va = same ? s5[0] : s5[3], vc = same ? s5[1] : s5[2], vg = same ? s5[2] : s5[1], vt = same ? s5[3] : s5[0], vn = s5[4];
            atomicAdd(&nuc[0 * bins + bin], __fmul_rn(va, w)); atomicAdd(&nuc[1 * bins + bin], __fmul_rn(vc, w));
            atomicAdd(&nuc[2 * bins + bin], __fmul_rn(vg, w)); atomicAdd(&nuc[3 * bins + bin], __fmul_rn(vt, w));
            atomicAdd(&nuc[4 * bins + bin], __fmul_rn(vn, w));
        }
    }
}

// doubles of scratch one wavefront (64 kept sequences) needs for reads of up to Lmax bases
size_t gmk_pair_hmm_cells(uint32_t Lmax) {
    const size_t W = (size_t)Lmax + 1;
    return (W * W * 3 + 2 * W * 2) * 64 + (((size_t)Lmax * 64 + 7) / 8 + 8);
}

int gmk_pair_hmm(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, const GmCand* items, uint32_t n, double* scratch, uint32_t Lmax, float* hmm,
                 void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_pair_hmm, dim3((n + 63) / 64), dim3(64), 0, S_(stream), ix, p, b, items, n, scratch, gmk_pair_hmm_cells(Lmax), Lmax, hmm);
    return (int)hipGetLastError();
}

int gmk_snp_deposit(float* cov, float* nuc, uint64_t bins, uint32_t bin_size, const GmDevBatch& b, const GmDevMatch* matches, const GmDevPos* positions,
                    uint32_t m0, uint32_t count, const float* post, const float* hmm, uint32_t Lmax, void* stream) {
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_snp_deposit, dim3(count), dim3(128), 0, S_(stream), cov, nuc, bins, bin_size, b, matches, positions, m0, post, hmm, Lmax);
    return (int)hipGetLastError();
}
