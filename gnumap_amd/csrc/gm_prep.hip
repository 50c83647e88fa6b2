// gm_prep.hip — k_prep_rows: k_prep (gm_kernels.hip: read self score, -a threshold, status, 2-bit forms) with the read row in REGISTERS.
//
// k_prep stages a tile of reads into LDS with coalesced loads, computes, and stages the next: its wavefronts spent 72 % of their time
// waiting (profiles/r04_sq_counters_1M.txt: SQ_WAIT_ANY / SQ_WAVE_CYCLES), the vector units 35 % busy - the load phase of a tile is not
// overlapped with anything, and three workgroups per CU (50 KB of LDS each) is all that fits.  Here a lane loads its own row (bases and
// qualities: 2 x NCH 8-byte words) into registers up front - all loads of a lane go out back to back, every line is fetched once, the
// trick k_nw_lane's register form measured (4.17 -> 0.66 GB per 1 M candidates) - and works from the registers at compile-time byte
// positions: no tile, no barrier between load and compute, the loads of one wavefront overlap the arithmetic of the others (LDS only for
// the term / class tables and the output transpose).  The 2-bit forms of a wavefront's 64 reads are ONE contiguous stretch in HBM
// (64 x pack_words x 4 bytes): they are collected in LDS and written out as whole 16-byte pieces, coalesced, instead of word by word from
// a lane's strided row.
//
// Arithmetic and outputs are k_prep's, statement by statement (self score = fp32 sum in index order of the per-base terms - the term of a
// base is looked up in the (class, quality character) table built with the reference's expression, bin_seq.cpp:975-987; status / cutoff /
// min_score as Driver.cpp:432-502).  The host launches it for rows of up to 152 bytes; longer rows keep k_prep.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "gm_internal.h"
#include "gm_device.h"

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

template <int NCH>
__global__ void __launch_bounds__(256) k_prep_rows(GmDevIndex ix, GmDevParams p, GmDevBatch b) {
    __shared__ float s_term[2][10][128];                   // [phred table][char class][quality character & 127]
    __shared__ uint8_t s_cls[256];
    __shared__ int s_other_uniform;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_rows[];        // [4 waves][64 lanes][pack_words]: the 2-bit forms on their way out
    const float4* const S4 = reinterpret_cast<const float4*>(p.S256);
    if (threadIdx.x == 0) s_other_uniform = 1;
    __syncthreads();
    {
        const int ch = threadIdx.x;
        int cl = 8;
        switch (ch) { case 'A': cl = 0; break; case 'C': cl = 1; break; case 'G': cl = 2; break; case 'T': cl = 3; break;
                      case 'a': cl = 4; break; case 'c': cl = 5; break; case 'g': cl = 6; break; case 't': cl = 7; break; default: cl = 8; }
        s_cls[ch] = (uint8_t)cl;
        if (cl == 8) {
            const float4 r0 = S4[(int)'N'], r = S4[ch];
            if (r.x != r0.x || r.y != r0.y || r.z != r0.z || r.w != r0.w) atomicAnd(&s_other_uniform, 0);
        }
    }
    for (int e = threadIdx.x; e < 2 * 9 * 128; e += 256) {
        const int tab = e / (9 * 128), cl = (e / 128) % 9, qc = e & 127;
        const int ch = cl < 8 ? "ACGTacgt"[cl] : 'N';
        const float2 pq = p.lut[tab * 256 + qc];
        const float4 sv = S4[ch];
        const float sarr[4] = { sv.x, sv.y, sv.z, sv.w };
        s_term[tab][cl][qc] = gm_get_val(gm_nt4((uint32_t)ch), pq.x, pq.y, sarr);
    }
    __syncthreads();
    const bool uni = s_other_uniform != 0;
    const uint32_t pw = b.pack ? b.pack_words : 0u;
    uint32_t* const myrow = s_rows + (size_t)threadIdx.x * pw;
    unsigned long long bad = 0, high = 0;
    for (uint32_t base = blockIdx.x * 256u; base < b.n; base += gridDim.x * 256u) {
        const uint32_t r = base + threadIdx.x;
        const bool rv = r < b.n;
        const uint32_t L = rv ? b.len[r] : 0u;
        // ---- the row into registers: every load of the lane in flight at once ----
        uint2 RB[NCH], RQ[NCH];
        {
            const uint8_t* rb = b.bases + (size_t)(rv ? r : 0u) * b.stride;
            const uint8_t* rq = b.quals + (size_t)(rv ? r : 0u) * b.stride;
            // 16 bytes per load where the read reaches into both halves (rows are 8-byte aligned, the loads need 4): half as many address
            // cycles in the texture path, where a wavefront's 64 rows are 64 different lines for every load instruction
#pragma unroll
            for (int k = 0; k < NCH; k += 2) {
                RB[k] = make_uint2(0u, 0u); RQ[k] = make_uint2(0u, 0u);
                if (k + 1 < NCH) { RB[k + 1] = make_uint2(0u, 0u); RQ[k + 1] = make_uint2(0u, 0u); }
                if (k + 1 < NCH && (uint32_t)(8 * k + 8) < L) {
                    uint4 tb, tq;
                    __builtin_memcpy(&tb, rb + 8 * k, 16); __builtin_memcpy(&tq, rq + 8 * k, 16);
                    RB[k] = make_uint2(tb.x, tb.y); RQ[k] = make_uint2(tq.x, tq.y);
                    if (k + 1 < NCH) { RB[k + 1] = make_uint2(tb.z, tb.w); RQ[k + 1] = make_uint2(tq.z, tq.w); }
                } else if ((uint32_t)(8 * k) < L) {
                    RB[k] = *reinterpret_cast<const uint2*>(rb + 8 * k); RQ[k] = *reinterpret_cast<const uint2*>(rq + 8 * k);
                }
            }
        }
        if (pw) {                                         // the lane's row of the output stretch starts out zero
#pragma unroll 1
            for (uint32_t w = 0; w < pw; w += 4) *reinterpret_cast<uint4*>(myrow + w) = make_uint4(0u, 0u, 0u, 0u);
        }
        const int tab = (rv && r < b.illumina_until) ? 1 : 0;
        const float2* lut = p.lut + tab * 256;
        const float (*term)[128] = s_term[tab];
        float score = 0.0f;
        uint32_t pk = 0, any_n = 0, carry = 0;
        bool nan_l = false;
        uint32_t qor = 0;                                    // OR of the quality words (bit 7 of a byte: a character above 127)
        const uint32_t rsh = 2u * (16u * b.pack_w2 - L);     // the reverse-strand form is the complemented read shifted up to the top of its words
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if ((uint32_t)(8 * k) < L) {
                const uint2 bw = RB[k], qw = RQ[k];
                if ((uint32_t)(8 * k + 8) <= L) {
                    // a whole word of 8 bases: no per-base conditions, the 8 class reads go out together, then the 8 term reads
                    uint32_t cl[8];
                    float v[8];
#pragma unroll
                    for (uint32_t t = 0; t < 8; ++t) cl[t] = s_cls[((t < 4 ? bw.x : bw.y) >> ((t & 3) * 8)) & 255u];
#pragma unroll
                    for (uint32_t t = 0; t < 8; ++t) v[t] = term[cl[t]][((t < 4 ? qw.x : qw.y) >> ((t & 3) * 8)) & 127u];     // NaN when the probability is negative, like the direct form
                    qor |= qw.x | qw.y;
#pragma unroll
                    for (uint32_t t = 0; t < 8; ++t) {
                        pk |= (cl[t] & 3u) << ((((uint32_t)(8 * k) & 8u) + t) << 1); any_n |= cl[t];
                        score = __fadd_rn(score, v[t]);
                    }
#pragma unroll
                    for (uint32_t t = 0; t < 8; t += 2) nan_l |= __builtin_isunordered(v[t], v[t + 1]);
                } else {
#pragma unroll 1
                    for (uint32_t t = 0; t < L - (uint32_t)(8 * k); ++t) {              // the last, partial word
                        const uint32_t ch = ((t < 4 ? bw.x : bw.y) >> ((t & 3) * 8)) & 255u;
                        const uint32_t qc = ((t < 4 ? qw.x : qw.y) >> ((t & 3) * 8)) & 255u;
                        const uint32_t c1 = s_cls[ch];
                        const float v1 = term[c1][qc & 127u];
                        pk |= (c1 & 3u) << ((((uint32_t)(8 * k) & 8u) + t) << 1); any_n |= c1;
                        qor |= qc;
                        nan_l |= v1 != v1;
                        score = __fadd_rn(score, v1);
                    }
                }
                if (pw && ((k & 1) || (uint32_t)(8 * k + 8) >= L)) {          // a word of 16 bases is complete (or the read ends)
                    const uint32_t w = (uint32_t)k >> 1;
                    uint32_t rvw = __brev(pk);
                    rvw = ((rvw >> 1) & 0x55555555u) | ((rvw & 0x55555555u) << 1);       // 2-bit groups in reverse order
                    myrow[1u + b.pack_w2 - 1u - w] = rvw;
                    const uint32_t g = ~pk, bs = rsh & 31u;
                    uint32_t* const rform = myrow + b.pack_w2 + 2u + (rsh >> 5);
                    rform[w] = (g << bs) | carry;
                    carry = bs ? g >> (32u - bs) : 0u;
                    if ((uint32_t)(8 * k + 8) >= L) rform[w + 1u] = carry;   // the word above the last one takes what was shifted out (its own padding word when bs = 0)
                    pk = 0;
                }
            }
        }
        any_n >>= 3;                                         // (classes are 0..8: bit 3 of their OR = some character outside ACGTacgt)
        if ((qor & 0x80808080u) || (any_n && !uni)) {
            // a quality character above 127, or a row of S the caller edited (-S file), somewhere in the read: the whole sum again with the
            // direct form for those bases (rare; one copy of this code instead of one per unrolled base - the row is still in cache)
            const uint8_t* rb = b.bases + (size_t)r * b.stride;
            const uint8_t* rq = b.quals + (size_t)r * b.stride;
            score = 0.0f; nan_l = false;
#pragma unroll 1
            for (uint32_t i = 0; i < L; ++i) {
                const uint32_t ch = rb[i], qc = rq[i];
                const uint32_t cl = s_cls[ch];
                float v;
                if (qc < 128u && (cl < 8u || uni)) {
                    v = term[cl][qc];
                } else {
                    const float2 pq = lut[qc];
                    const float4 sv = S4[ch];
                    const float s4[4] = { sv.x, sv.y, sv.z, sv.w };
                    v = gm_get_val(gm_nt4(ch), pq.x, pq.y, s4);
                }
                nan_l |= v != v;
                if (qc >= 128u) high = 1;                    // (k_nw_rows' table of row values covers the characters below 128)
                score = __fadd_rn(score, v);
            }
        }
        if (nan_l) bad = 1;                                  // negative probability (SeqReader.cpp:1171-1189)
        if (rv) {
            int8_t st = 0;
            float self = 0.0f;
            double mn;
            if (L < (uint32_t)p.mer) {
                st = -2;                                     // READ_TOO_SHORT
                mn = 0.0;
            } else if (p.nw) {
                float v = 0.0f;                              // get_align_score(read,cons,0,L-1): begin(=0) + mid + end(=0)
                v = __fadd_rn(v, 0.0f); v = __fadd_rn(v, score); v = __fadd_rn(v, 0.0f);
                self = v;
                if ((double)self < (double)p.cutoff) st = -3;    // READ_TOO_POOR
                mn = p.align_is_fraction ? (double)p.align_score * (double)self : (double)p.align_score;
            } else {
                mn = (double)p.kmin;                         // Driver.cpp:502
            }
            b.status[r] = st;
            if (pw) myrow[0] = L | (any_n << 16) | ((st != 0 ? 1u : 0u) << 17);
            b.self_score[r] = self;
            b.min_score[r] = mn;
            b.top_score[r] = 0.0f;
            b.hit_count[r] = 0;
            b.hit_cursor[r] = 0;
        }
        if (pw) {
            // the wavefront's 64 rows are one stretch of pack[]: out in 16-byte pieces, lane after lane
            __syncthreads();
            const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
            const uint32_t r0 = base + 64u * wave;
            if (r0 < b.n) {
                const uint32_t rows = b.n - r0 < 64u ? b.n - r0 : 64u;
                const uint32_t pieces = rows * pw / 4u;                         // (pack_words is a multiple of 4)
                const uint4* src = reinterpret_cast<const uint4*>(s_rows + (size_t)wave * 64u * pw);
                uint4* dst = reinterpret_cast<uint4*>(b.pack + (size_t)r0 * pw);
                for (uint32_t q = lane; q < pieces; q += 64u) dst[q] = src[q];
            }
            __syncthreads();
        }
    }
    gm_count(b, GMK_BAD_QUAL, bad);
    gm_count(b, GMK_HIGH_QUAL, high);
}

int gmk_prep_rows(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, void* stream) {
    if (b.n == 0) return 0;
    if (b.stride > 152) return (int)hipErrorInvalidValue;
    const size_t lds = b.pack ? (size_t)256 * b.pack_words * 4 : 0;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)b.n + 255) / 256, 256ull * 5 * 4);
    if (b.stride <= 104) hipLaunchKernelGGL((k_prep_rows<13>), dim3(grid), dim3(256), lds, S_(stream), ix, p, b);
    else hipLaunchKernelGGL((k_prep_rows<19>), dim3(grid), dim3(256), lds, S_(stream), ix, p, b);
    return (int)hipGetLastError();
}
