// gm_nw.hip — k_nw_rows: the banded probabilistic Needleman-Wunsch score (bin_seq::get_align_score, src/bin_seq.cpp:761-850,
// get_val :975-987, max_flt :1013-1026) for blocks whose reads all have ONE length, one LANE per candidate.
//
// Same arithmetic as k_nw_lane (gm_kernels.hip): the 7-cell band row in registers, swept from the bottom-right row by row, cells
// j = i+3 .. i-3, three fp32 adds and the reference's 3-way max per cell, no contraction - the score bits are the reference's.  What
// differs is everything AROUND the adds, which is where k_nw_lane spent its ~170 vector instructions per DP row (it ran at the vector
// issue rate, HBM idle):
//   * the four substitution values of a PWM row, val(row, g) for g = a, c, g, t, depend only on (called base, quality character): they
//     are looked up in a per-workgroup LDS table built once with the reference's own expression (gm_get_val; bit-identical, the same
//     function value looked up instead of recomputed): one ds_read_b128 instead of 28 multiplies / adds + the LUT read;
//   * the read row and the reference window are brought into DP ORDER when they are loaded (forward-strand reads byte-reversed, the
//     2-bit window bit-reversed and funnel-shifted), so that DP row t of EVERY lane - whatever its strand and window start - finds its
//     base, its quality and its new window column at the same register and bit position: compile-time shifts in rows unrolled 8 at a
//     time, no per-row address arithmetic, no 13-way register pick, no strand-dependent indexing;
//   * a band column's base is kept as its byte offset inside a value record (4 x its 2-bit code); a cell's value is ONE ds_read_b32 at
//     record + column offset (the first form picked it out of the row's float4 with three v_cndmask on two lane masks per column: 21 of
//     the 57 instructions of a row; as LDS reads the row is 45 and the kernel 5 % faster - the LDS has the room);
//     sliding the band is register renaming inside the unrolled rows;
//   * the first 8 and the last 9 .. 16 rows (which touch row L, column L or column -1) run a generic row; the rows between them test nothing.
// The host launches it only for blocks it has checked: one read length L (24 <= L <= 8 NCH), no quality character above 127, -M 3.
#include <hip/hip_runtime.h>
#include "gm_device.h"

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define GM_NWR_NCOFF 1024u                  // contig offsets cached in LDS when there are at most this many
#define GM_NWR_QSTRIDE 144u                 // bytes per quality character in the value table: 8 class records of 16 bytes + 16 bytes of skew (banks)
#define GM_NWR_TAB_BYTES (128u * GM_NWR_QSTRIDE)

template <int NCH>
__global__ void __launch_bounds__(256, NCH <= 13 ? 4 : 3) k_nw_rows(GmDevIndex ix, GmDevParams p, GmDevBatch b, const uint32_t L, const uint32_t ntab) {
    constexpr int NHW = (8 * NCH + 3) / 16 + 1;                                // 16-column words of the window stream
    extern __shared__ __attribute__((aligned(16))) unsigned char s_tab[];       // [ntab][128 quality characters][8 classes] float4 {val(a), val(c), val(g), val(t)}
    __shared__ uint16_t s_cls[2][2][256];                                        // [phred table][strand][read character] -> byte offset of its class record (+ its table's)
    __shared__ uint32_t s_coff[GM_NWR_NCOFF];
    __shared__ uint32_t s_pre[GM_NSHARD + 4];
    {
        float sg[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) sg[g][k] = p.S256[(size_t)("acgt"[g]) * 4 + k];     // windows are lowercase acgt (GetString)
        for (uint32_t e = threadIdx.x; e < ntab * 1024u; e += 256) {
            const uint32_t tab = e >> 10, q = (e >> 3) & 127u, cl = e & 7u;
            const uint32_t code = cl < 4u ? cl : 4u;                             // classes 4 .. 7: any character outside ACGTacgt
            const float2 pq = p.lut[tab * 256u + q];
            float4 v;
            v.x = gm_get_val(code, pq.x, pq.y, sg[0]); v.y = gm_get_val(code, pq.x, pq.y, sg[1]);
            v.z = gm_get_val(code, pq.x, pq.y, sg[2]); v.w = gm_get_val(code, pq.x, pq.y, sg[3]);
            *reinterpret_cast<float4*>(s_tab + tab * GM_NWR_TAB_BYTES + q * GM_NWR_QSTRIDE + cl * 16u) = v;
        }
        const uint32_t ch = threadIdx.x, code = gm_nt4(ch);
#pragma unroll
        for (uint32_t tab = 0; tab < 2; ++tab)
#pragma unroll
            for (uint32_t st = 0; st < 2; ++st)                                   // the reverse strand reads the complemented PWM row (reverse_comp_cpy)
                s_cls[tab][st][ch] = (uint16_t)((tab < ntab ? tab : 0u) * GM_NWR_TAB_BYTES + ((code < 4u && st) ? 3u - code : code) * 16u);
    }
    const bool lds_coff = ix.n_seqs + 1 <= GM_NWR_NCOFF;
    if (lds_coff) for (uint32_t q = threadIdx.x; q <= ix.n_seqs; q += 256) s_coff[q] = ix.contig_off[q];
    const uint32_t* coff = lds_coff ? s_coff : ix.contig_off;
    const uint32_t n_cands = gm_cand_prefix(b, s_pre);          // includes the barrier that publishes the tables
    const float gap = p.gap, gap4 = __fmul_rn(p.gap, 4.0f);
    const float ninf_gap = __fadd_rn(GM_NEG_INF, gap);
    const uint32_t* pac32 = reinterpret_cast<const uint32_t*>(ix.pac);
    const int Li = (int)L;
    const int nchunk = (Li + 7) >> 3;
    const uint32_t sh = (uint32_t)(8 * nchunk - Li);            // bytes the reversed row of a forward-strand read is shifted down by
    unsigned long long cells = 0, accepted = 0;
    for (uint32_t wi = blockIdx.x * 256 + threadIdx.x; wi < n_cands; wi += gridDim.x * 256) {
        const size_t ci = gm_cand_slot(b, s_pre, wi);
        GmCand c = b.cands[ci];
        const uint32_t r = c.rs >> 1, strand = c.rs & 1;
        if ((c.flags & 4) && b.rs_overflow[c.rs]) continue;                 // superseded by the retry kernel
        const bool ok = gm_window_ok(ix, coff, c.b, L);
        float result = 0.0f;
        if (ok && p.nw) {
            // ---- the read in DP order: element t belongs to DP row i = L - 1 - t.  Reverse strand: t = the read's own index (complemented
            //      through s_cls); forward strand: the read backwards ----
            uint2 XB[NCH], XQ[NCH];
            {
                const uint8_t* rb = b.bases + (size_t)r * b.stride;
                const uint8_t* rq = b.quals + (size_t)r * b.stride;
                const uint32_t selx = strand ? 0x03020100u : 0x04050607u, sely = strand ? 0x07060504u : 0x00010203u;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    XB[k] = make_uint2(0u, 0u); XQ[k] = make_uint2(0u, 0u);
                    if (k < nchunk) {
                        const int fo = Li - 8 - 8 * k;
                        const uint32_t off = strand ? (uint32_t)(8 * k) : (uint32_t)(fo > 0 ? fo : 0);
                        uint2 tb, tq;
                        __builtin_memcpy(&tb, rb + off, 8); __builtin_memcpy(&tq, rq + off, 8);
                        uint2 ob = make_uint2(__builtin_amdgcn_perm(tb.y, tb.x, selx), __builtin_amdgcn_perm(tb.y, tb.x, sely));
                        uint2 oq = make_uint2(__builtin_amdgcn_perm(tq.y, tq.x, selx), __builtin_amdgcn_perm(tq.y, tq.x, sely));
                        if (k == nchunk - 1 && sh != 0u) {                      // the last word of a reversed row was loaded from offset 0: its elements sit sh bytes up
                            const uint32_t s8 = strand ? 0u : 8u * sh;
                            const unsigned long long wb = (((unsigned long long)ob.y << 32) | ob.x) >> s8, wq = (((unsigned long long)oq.y << 32) | oq.x) >> s8;
                            ob = make_uint2((uint32_t)wb, (uint32_t)(wb >> 32)); oq = make_uint2((uint32_t)wq, (uint32_t)(wq >> 32));
                        }
                        XB[k] = ob; XQ[k] = oq;
                    }
                }
            }
            // ---- the window in DP order: H[u] = w[L - 1 - u], 2 bits each, 16 per word, u ascending from bit 0.  The reference packs 4
            //      bases per byte MSB first (_get_pac, src/bntseq.c:225): in a byte-swapped 32-bit word base g sits at bit 30 - 2 (g & 15),
            //      so descending positions are ascending bits and the stream is the words n0, n0 - 1, .. funnel-shifted by 30 - 2 (g0 & 15) ----
            uint32_t HW[NHW];
            {
                const uint32_t g0 = c.b + L - 1u, n0 = g0 >> 4, s = 30u - 2u * (g0 & 15u);
                uint32_t W[NHW + 1];
#pragma unroll
                for (int j = 0; j <= NHW; ++j) {
                    const uint32_t idx = n0 >= (uint32_t)j ? n0 - (uint32_t)j : 0u;          // below the reference's start only columns j < 0 are fed
                    const uint32_t x = pac32[idx];
                    W[j] = __builtin_amdgcn_perm(x, x, 0x00010203u);
                }
#pragma unroll
                for (int m = 0; m < NHW; ++m) HW[m] = __builtin_amdgcn_alignbit(W[m + 1], W[m], s);
            }
            const uint32_t tab = (ntab > 1u && r < b.illumina_until) ? 1u : 0u;
            const uint16_t* const clsrow = s_cls[tab][strand];
            // band row of i + 1: P[d] = nm[i+1][i+1+d-3].  Row L: gGAP * (L - j) for j <= L (bin_seq.cpp:805-808)
            float P[7];
#pragma unroll
            for (int d = 0; d < 7; ++d) P[d] = d <= 3 ? __fmul_rn(gap, (float)(3 - d)) : GM_NEG_INF;
            // the base of band column d as 4 x its 2-bit code (its byte offset in a value record); row L - 1: columns j = L-4+d, valid for d <= 3
            uint32_t cc[7];
#pragma unroll
            for (int d = 0; d < 7; ++d) cc[d] = d <= 3 ? ((HW[0] >> (2 * (3 - d))) & 3u) << 2 : 0u;
            // a row's value record {val(a), val(c), val(g), val(t)} as its LDS byte offset
            auto row_vals = [&](uint32_t bword, uint32_t qword, uint32_t shift) -> uint32_t {
                const uint32_t chv = (bword >> shift) & 255u;
                const uint32_t co = clsrow[chv];
                const uint32_t qv = (qword >> shift) & 255u;
                return qv * GM_NWR_QSTRIDE + co;
            };
            auto cell_val = [&](const uint32_t v, int d) -> float { return *reinterpret_cast<const float*>(s_tab + v + cc[d]); };
            auto slide = [&](uint32_t hw, uint32_t pos) {                       // band columns of the next row; the new one is H[t + 4] (j = i - 4)
#pragma unroll
                for (int d = 6; d >= 1; --d) cc[d] = cc[d - 1];
                cc[0] = ((hw >> pos) & 3u) << 2;
            };
            // an interior row (4 <= i <= L - 5): every cell inside the matrix, band edges are NEG_INF
            auto row_int = [&](const uint32_t v, uint32_t hw, uint32_t pos) {
                float val[7], mm[7], g1[7];
#pragma unroll
                for (int d = 0; d < 7; ++d) val[d] = cell_val(v, d);
#pragma unroll
                for (int d = 0; d < 7; ++d) { mm[d] = __fadd_rn(P[d], val[d]); g1[d] = d > 0 ? __fadd_rn(P[d - 1], gap) : ninf_gap; }
                float left = ninf_gap;                                          // nm[i][j+1] + gGAP beyond the band
#pragma unroll
                for (int d = 6; d >= 0; --d) {
                    const float best = fmaxf(fmaxf(mm[d], g1[d]), left);       // max_flt (:1013-1026) on finite operands = v_max3_f32
                    P[d] = best;
                    left = __fadd_rn(best, gap);
                }
                slide(hw, pos);
            };
            // a row near a matrix edge: exactly k_nw_lane's EDGE row
            auto row_edge = [&](const int t, const uint32_t v, uint32_t hw, uint32_t pos) {
                const int i = Li - 1 - t;
                const float lastcol = __fmul_rn(gap, (float)(unsigned)(Li - i));                  // nm[i][L] = gGAP * (L - i)
#pragma unroll
                for (int d = 6; d >= 0; --d) {
                    const int j = i + d - 3;
                    const float val = cell_val(v, d);
                    const float up = d > 0 ? P[d - 1] : ((i + 1 == Li) ? gap4 : GM_NEG_INF);     // nm[i+1][j]
                    const float left = d < 6 ? P[d + 1] : ((j + 1 == Li) ? gap4 : GM_NEG_INF);   // nm[i][j+1]
                    const float best = fmaxf(fmaxf(__fadd_rn(P[d], val), __fadd_rn(up, gap)), __fadd_rn(left, gap));
                    P[d] = (j >= 0 && j < Li) ? best : (j == Li ? lastcol : GM_NEG_INF);
                }
                slide(hw, pos);
            };
            auto pick2 = [&](const uint2* X, int k) -> uint2 {                  // k is wave-uniform: a branch tree, one move per word
                uint2 o = X[0];
                switch (k) {
#define GM_NWR_CASE(q) case q: if (q < NCH) o = X[q < NCH ? q : 0]; break;
                    GM_NWR_CASE(1) GM_NWR_CASE(2) GM_NWR_CASE(3) GM_NWR_CASE(4) GM_NWR_CASE(5) GM_NWR_CASE(6) GM_NWR_CASE(7) GM_NWR_CASE(8) GM_NWR_CASE(9)
                    GM_NWR_CASE(10) GM_NWR_CASE(11) GM_NWR_CASE(12) GM_NWR_CASE(13) GM_NWR_CASE(14) GM_NWR_CASE(15) GM_NWR_CASE(16) GM_NWR_CASE(17) GM_NWR_CASE(18)
#undef GM_NWR_CASE
                    default: break;
                }
                return o;
            };
            auto pick_hw = [&](int m) -> uint32_t {
                uint32_t o = HW[0];
                switch (m) {
#define GM_NWR_CASE(q) case q: if (q < NHW) o = HW[q < NHW ? q : 0]; break;
                    GM_NWR_CASE(1) GM_NWR_CASE(2) GM_NWR_CASE(3) GM_NWR_CASE(4) GM_NWR_CASE(5) GM_NWR_CASE(6) GM_NWR_CASE(7) GM_NWR_CASE(8) GM_NWR_CASE(9) GM_NWR_CASE(10)
#undef GM_NWR_CASE
                    default: break;
                }
                return o;
            };
            // rows [t0, t1) of chunk k at run-time positions (the first rows, the tail): EDGE = they touch row L, column L or column -1
            auto tail_rows = [&](const int k, const int t0, const int t1, auto edge_tag) {
                constexpr bool EDGE = decltype(edge_tag)::value;
                const uint2 bw = pick2(XB, k), qw = pick2(XQ, k);
                for (int t = t0; t < t1; ++t) {
                    const uint32_t bsel = (t & 4) ? bw.y : bw.x, qsel = (t & 4) ? qw.y : qw.x;
                    const uint32_t v = row_vals(bsel, qsel, (uint32_t)(t & 3) << 3);
                    const int u = t + 4;
                    if (EDGE) row_edge(t, v, pick_hw(u >> 4), 2u * (uint32_t)(u & 15));
                    else row_int(v, pick_hw(u >> 4), 2u * (uint32_t)(u & 15));
                }
            };
            // interior rows [B0, 8) of chunk k at compile-time positions; PAR = k & 1 fixes where the rows' new columns sit in the window
            // stream.  The class offsets of all rows are requested first (one LDS round trip for the chunk)
            auto rows8 = [&](const int k, auto par_tag, auto b0_tag) {
                constexpr int PAR = decltype(par_tag)::value, B0 = decltype(b0_tag)::value;
                const uint2 bw = pick2(XB, k), qw = pick2(XQ, k);
                const int m = (8 * k + 4) >> 4;
                const uint32_t hw_lo = pick_hw(m), hw_hi = PAR ? pick_hw(m + 1) : 0u;
                uint32_t co[8];
#pragma unroll
                for (int bb = B0; bb < 8; ++bb) co[bb] = clsrow[((bb < 4 ? bw.x : bw.y) >> ((bb & 3) << 3)) & 255u];
#pragma unroll
                for (int bb = B0; bb < 8; ++bb) {
                    const uint32_t v = (((bb < 4 ? qw.x : qw.y) >> ((bb & 3) << 3)) & 255u) * GM_NWR_QSTRIDE + co[bb];
                    if (PAR) { if (bb < 4) row_int(v, hw_lo, 2u * (12u + bb)); else row_int(v, hw_hi, 2u * (bb - 4u)); }
                    else row_int(v, hw_lo, 2u * (4u + bb));
                }
            };
            if (nchunk >= 4) {
                // rows 0 .. 3 touch row L / column L, rows L - 4 .. L - 1 column -1 (row i = 3 slides column -1 in); all others are interior
                using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I4 = std::integral_constant<int, 4>;
                tail_rows(0, 0, 4, std::true_type{});
                rows8(0, I0{}, I4{});
                const int k_last = nchunk - 3;
                for (int k = 1; k <= k_last; ++k) { if (k & 1) rows8(k, I1{}, I0{}); else rows8(k, I0{}, I0{}); }
                for (int k = k_last + 1; k < nchunk; ++k) {
                    const int lo = 8 * k, hi = (8 * k + 8 < Li) ? 8 * k + 8 : Li, cut = Li - 4;
                    if (lo < cut) tail_rows(k, lo, hi < cut ? hi : cut, std::false_type{});
                    if (hi > cut) tail_rows(k, lo > cut ? lo : cut, hi, std::true_type{});
                }
            } else {
                for (int k = 0; k < nchunk; ++k) tail_rows(k, 8 * k, (8 * k + 8 < Li) ? 8 * k + 8 : Li, std::true_type{});
            }
            // cells inside the band (counter only)
            if (Li >= 7) cells += (unsigned long long)(7 * Li - 12);
            else for (int i = 0; i < Li; ++i) { int lo = i - 3 < 0 ? 0 : i - 3, hi = i + 3 >= Li ? Li - 1 : i + 3; cells += (unsigned long long)(hi - lo + 1); }
            result = P[3];                                                      // nm[0][0]
        } else if (!p.nw) {
            result = (float)c.step;                                             // --no_nw: the score is the vote count (:70-76)
        }
        uint8_t fl = c.flags & 4;
        if (ok) {
            fl |= GMC_VALID;
            if (result > 0.0f) atomicMax(reinterpret_cast<int*>(&b.top_score[r]), __float_as_int(result));   // top_align_score (:95-98)
            if ((double)result >= b.min_score[r]) {                            // :102
                fl |= GMC_ACCEPT;
                atomicAdd(&b.hit_count[r], 1u);
                ++accepted;
            }
        }
        b.cands[ci].score = result;
        b.cands[ci].flags = fl;
    }
    gm_count(b, GMK_NW_CELLS, cells);
    gm_count(b, GMK_ACCEPTED, accepted);
}

// L = the one read length of the block; illumina = some reads of the block use the Phred+64 table (both tables are then resident)
int gmk_nw_rows(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, uint32_t n_cands, uint32_t L, void* stream) {
    if (b.n == 0) return 0;
    const uint32_t ntab = b.illumina_until ? 2u : 1u;
    const size_t lds = (size_t)ntab * GM_NWR_TAB_BYTES;
    const uint32_t nw_fixed = (uint32_t)gm_opt_ll("GM_NW_GRID", 0);
    const uint32_t grid = nw_fixed ? nw_fixed : std::min<uint32_t>(16384u, std::max<uint32_t>(2048u, n_cands / 1024u));
    if (L <= 104) hipLaunchKernelGGL((k_nw_rows<13>), dim3(grid), dim3(256), lds, S_(stream), ix, p, b, L, ntab);
    else hipLaunchKernelGGL((k_nw_rows<19>), dim3(grid), dim3(256), lds, S_(stream), ix, p, b, L, ntab);
    return (int)hipGetLastError();
}
