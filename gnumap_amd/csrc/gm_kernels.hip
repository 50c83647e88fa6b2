// gm_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the GNUMAP seed-and-extend hot path.
//
// Kernels, in pipeline order (reference file:line each one stands for):
//   k_expand_full_sa   once per index: rank-sampled SA -> full 32-bit SA by LF walks     (bwt_sa/bwt_invPsi src/bwt.c:53-96)
//   k_prep             read self score, -a threshold, status; 2-bit forms of the read for the fused seed lookup
//                                                                                        (set_top_matches src/Driver.cpp:446-503,
//                                                                                         get_align_score_mid src/bin_seq.cpp:860-893)
//   k_seed             adaptive k-mer walk + backward search, one lane per read x strand (align_sequence inc/align_seq2_raw.cpp:192-243,
//                                                                                         bwt_match_exact/bwt_2occ/bwt_occ src/bwt.c:107-239)
//                      - or, GmDevParams::fused, the same walk INSIDE k_vote_slots / k_vote_tiny / k_vote_tiny2 (gm_tiny_seeds<true>:
//                      lane j takes the k-mer at j * jump; the serial walk gm_seed_walk only for the read x strands where that fails)
//   k_locate_sampled   faithful locate: LF walk to the next sampled rank                  (bwt_sa src/bwt.c:86-96)
//   k_vote_*           locate + vote: k_vote_slots (dense seeds, 2 waves per read x strand), k_vote_tiny / _tiny2 (one wave, few hits in
//                      short runs), k_vote_sparse + k_vote_fast_list, k_vote, k_vote_block, k_vote_retry (exact table in HBM);
//                      k_cand_gather moves the candidates the vote waves left in their own slots into the shards
//                                                                                         (inc/align_seq2_raw.cpp:262-274, process_hits :28-40)
//   k_nw_lane, k_nw    banded probabilistic NW score: 1 lane / 8 lanes per candidate      (get_align_score_begin src/bin_seq.cpp:781-850)
//   k_scatter_hits     accepted candidates -> per-read CSR
//   k_traceback_lane, k_traceback   forward banded DP with move bits + traceback: 1 lane / 8 lanes per kept sequence
//                                                                                         (get_align_score_w_traceback src/bin_seq.cpp:445-718)
//   k_coverage_add     amount_genome[(pos+i)/bin] += w for caller-given deposits (gm_coverage_add)   (GenomeBwt::AddScore src/GenomeBwt.cpp:483-490)
// The unique map, CIGAR text, SAM rows and the coverage deposit of a batch are in gm_output.hip; the sorted-key vote path for read x
// strands with very many SA hits is in gm_heavy.hip.
//
// Arithmetic: ranks/coordinates are u32 (reference < 2^32-1 ranks); scores are fp32 with the reference's operation
// order and NO fused multiply-add (built with -ffp-contract=off; mul/add also go through __fmul_rn/__fadd_rn).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include "gm_internal.h"
#include "gm_device.h"


// ------------------------------------------------------------------------------------------------
// full SA expansion (one thread per SA sample)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_expand_full_sa(GmDevIndex ix, uint32_t* full_sa, uint32_t n_sa) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_sa) return;
    uint32_t k = s << ix.sa_shift;
    if (k > ix.seq_len) return;
    uint32_t v = (s == 0) ? ix.seq_len : ix.sa_samples[s];
    full_sa[k] = v;
    for (uint32_t it = 0; it < GM_WALK_CAP; ++it) {
        k = gm_inv_psi(ix, k);
        --v;
        if ((k & ix.sa_mask) == 0) break;
        full_sa[k] = v;
    }
}

// one thread per 96-position granule: derives the four bit planes from the reference-format BWT
__global__ void __launch_bounds__(256) k_build_occ_planes(GmDevIndex ix, uint4* planes, uint32_t nblk) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nblk) return;
    uint32_t x0 = j * 96u;
    uint32_t w[4][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
    for (uint32_t t = 0; t < 96; ++t) {
        uint32_t x = x0 + t;
        if (x >= ix.seq_len) break;
        uint32_t word = ix.bwt[((size_t)(x >> 7) << 4) + 8 + ((x & 0x7fu) >> 4)];     // bwt_B0 inc/bwt.h:72-78
        uint32_t c = (word >> ((~x & 0xfu) << 1)) & 3u;
        w[c][t >> 5] |= 1u << (t & 31);
    }
    for (uint32_t c = 0; c < 4; ++c) {
        uint32_t cnt = 0;
        if (j > 0) {
            uint32_t xl = x0 - 1;                            // last position before the granule, back in rank space
            uint32_t k = xl < ix.primary ? xl : xl + 1;
            cnt = gm_occ(ix, k, c);
        }
        uint4 v; v.x = cnt; v.y = w[c][0]; v.z = w[c][1]; v.w = w[c][2];
        planes[(size_t)c * nblk + j] = v;
    }
}

// SA interval of every T-mer (memoised bwt_match_exact on the last T characters of a seed k-mer): one thread per T-mer.
// code = sum over characters, leftmost character in the highest bits.  Empty: {0xFFFFFFFF, d} with d = number of
// characters (from the right end) after which the backward search died.
__global__ void __launch_bounds__(256) k_build_kmer_table(GmDevIndex ix, uint2* tab, int T) {
    // grid-stride over the codes: T = 16 has 2^32 of them, more work-items than one launch dimension takes
    for (uint64_t code = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; code < (1ull << (2 * T)); code += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t k = 0, l = ix.seq_len;
        uint2 out;
        out.x = 0; out.y = 0;
        bool ok = true;
        for (int t = 0; t < T; ++t) {                    // t-th character from the right end
            uint32_t c = (uint32_t)(code >> (2 * t)) & 3u;
            uint32_t ok_ = gm_occ_plane(ix, k - 1, c), ol_ = gm_occ_plane(ix, l, c);
            k = gm_L2(ix, c) + ok_ + 1;
            l = gm_L2(ix, c) + ol_;
            if (k > l) { ok = false; out.x = 0xFFFFFFFFu; out.y = (uint32_t)t + 1; break; }
        }
        if (ok) { out.x = k; out.y = l; }
        tab[code] = out;
    }
}

// One more character on the left of every (T-1)-mer of `prev`: the T-mer table from the (T-1)-mer table with ONE backward-search
// step per entry (the direct build walks all T steps per entry; at T = 14 that is 268 M x 14 steps against 268 M x 1)
__global__ void __launch_bounds__(256) k_extend_kmer_table(GmDevIndex ix, const uint2* prev, uint2* next, int T) {
    for (uint64_t code = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; code < (1ull << (2 * T)); code += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)(code >> (2 * (T - 1)));            // the leftmost character is searched last
        const uint2 pv = prev[code & ((1ull << (2 * (T - 1))) - 1ull)];
        uint2 out = pv;                                                  // an empty suffix stays empty, with the depth it died at
        if (pv.x != 0xFFFFFFFFu) {
            const uint32_t k = gm_L2(ix, c) + gm_occ_plane(ix, pv.x - 1, c) + 1;
            const uint32_t l = gm_L2(ix, c) + gm_occ_plane(ix, pv.y, c);
            if (k > l) { out.x = 0xFFFFFFFFu; out.y = (uint32_t)T; } else { out.x = k; out.y = l; }
        }
        next[code] = out;
    }
}

// Compact form of the k-mer table for the seed kernel: the SA intervals of lexicographically consecutive T-mers are adjacent
// (apart from the few suffixes shorter than T), so 8 consecutive codes need one start rank and 8 hit counts.  16 bytes per 8
// codes instead of 64: the 10-mer table shrinks from 8 MB to 2 MB and stays in the L2 of every XCD - the seed kernel's random
// lookups stop being HBM traffic.  A count byte of 224..239 marks an empty code and carries the depth at which its backward
// search died (what the seed walk needs to slide on), so the k-mers of the wrong strand - most of the probes with long seeds -
// cost ONE 128-byte line, not two.  A record is marked "escape" when a count does not fit (>= 224) or its intervals are not
// adjacent; escapes are answered from the full table, so results never depend on this form.
__global__ void __launch_bounds__(256) k_build_kmer_compact(const uint2* tab, uint4* ctab, int T) {
    for (uint64_t rec = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; rec < (1ull << (2 * T - 3)); rec += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t start = 0, next = 0, w0 = 0, w1 = 0, esc = 0;
        bool have = false;
        for (uint32_t i = 0; i < 8; ++i) {
            const uint2 iv = tab[rec * 8 + i];
            uint32_t cnt;
            if (iv.x != 0xFFFFFFFFu) {
                cnt = iv.y - iv.x + 1;
                if (!have) { start = iv.x; have = true; }
                else if (iv.x != next) esc = 1;              // a shorter suffix sorts in between
                next = iv.y + 1;
                if (cnt >= 224u) esc = 1;
            } else {
                cnt = 223u + iv.y;                           // empty: 224..239 = the search died after iv.y (1..16) characters
            }
            if (i < 4) w0 |= (cnt & 255u) << (8 * i); else w1 |= (cnt & 255u) << (8 * (i - 4));
        }
        ctab[rec] = make_uint4(start, w0, w1, esc);
    }
}

// ------------------------------------------------------------------------------------------------
// prep: one thread per read
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_prep(GmDevIndex ix, GmDevParams p, GmDevBatch b, uint32_t tile_reads) {
    // the term a base adds to the self score depends only on (its character's row of the score table, its quality character): one
    // table per workgroup, computed ONCE with the reference's expression (get_val, bin_seq.cpp:975-987), turns ~25 instructions per
    // base into one LDS read + one add - bit-identical, the same function value is only looked up instead of recomputed.
    // Rows: the 8 ACGT/acgt characters keep their own rows (the -b / -d edits touch the lowercase rows only), every other
    // character shares the row of its first occurrence class: codes 8.. are built on demand below (class 8 = any other char).
    __shared__ float s_term[2][10][128];                   // [phred table][char class][quality character & 127]
    __shared__ uint8_t s_cls[256];
    const float4* const S4 = reinterpret_cast<const float4*>(p.S256);
    __shared__ int s_other_uniform;
    if (threadIdx.x == 0) s_other_uniform = 1;
    __syncthreads();
    {
        // all characters outside ACGTacgt have the same row in the reference's table (transversion score for every base,
        // a_matrices.c:55-83) unless a caller edited gm_params.S by hand (-S file): then those bases take the direct path
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
    // a lane walking its own row with 8-byte loads pulls a whole 128-byte line per load and finds it evicted by the next one
    // (measured: 13 x the useful HBM traffic).  So the 256 reads of a tile are staged into LDS with coalesced 16-byte loads.
    // tile_reads = reads per tile that fit the LDS budget (256 at 100 bp); 0 = rows too long to stage, lanes read HBM directly
    extern __shared__ __attribute__((aligned(16))) unsigned char s_tile[];          // bases of a tile, then quals
    unsigned long long bad = 0, high = 0;
    const bool staged = tile_reads != 0;
    const uint32_t TR = staged ? tile_reads : 256u;
    const uint32_t n_tiles = (b.n + TR - 1u) / TR;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint32_t r0 = tile * TR;
    const uint32_t nr = b.n - r0 < TR ? b.n - r0 : TR;
    const size_t bytes = (size_t)nr * b.stride;
    unsigned char* const s_b = s_tile; unsigned char* const s_q = s_tile + (size_t)TR * b.stride;
    __syncthreads();                                 // the previous tile (and the tables) are done with
#pragma unroll
    for (int which = 0; which < 2 && staged; ++which) {
        const unsigned char* src = (which ? b.quals : b.bases) + (size_t)r0 * b.stride;
        unsigned char* dst = which ? s_q : s_b;
        if ((((size_t)src) & 15) == 0) {
            for (size_t o = (size_t)threadIdx.x * 16; o < bytes; o += 256 * 16) {
                if (o + 16 <= bytes) *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(src + o);
                else for (size_t q = o; q < bytes; ++q) dst[q] = src[q];
            }
        } else {
            for (size_t o = (size_t)threadIdx.x * 8; o < bytes; o += 256 * 8) *reinterpret_cast<uint2*>(dst + o) = *reinterpret_cast<const uint2*>(src + o);
        }
    }
    __syncthreads();
    const uint32_t r = r0 + threadIdx.x;
    if (threadIdx.x < TR && r < b.n) {
        uint32_t L = b.len[r];
        const uint8_t* rb = staged ? s_b + (size_t)threadIdx.x * b.stride : b.bases + (size_t)r * b.stride;
        const uint8_t* rq = staged ? s_q + (size_t)threadIdx.x * b.stride : b.quals + (size_t)r * b.stride;
        const int tab = (r < b.illumina_until) ? 1 : 0;
        const float2* lut = p.lut + tab * 256;
        const float (*term)[128] = s_term[tab];
        const bool uni = s_other_uniform != 0;
        int8_t st = 0;
        float self = 0.0f;
        double mn;
        float score = 0.0f;
        // 2-bit forms of the read for the fused seed lookup (GmDevBatch::pack): 16 bases = one word of each form
        uint32_t* const prow = b.pack ? b.pack + (size_t)r * b.pack_words : nullptr;
        uint32_t pk = 0, any_n = 0, carry = 0;
        const uint32_t rsh = 2u * (16u * b.pack_w2 - L);     // the reverse-strand form is the complemented read shifted up to the top of its words
        for (uint32_t i0 = 0; i0 < L; i0 += 8) {             // rows are 8-byte aligned (stride is a multiple of 8)
            const uint2 bw = *reinterpret_cast<const uint2*>(rb + i0);
            const uint2 qw = *reinterpret_cast<const uint2*>(rq + i0);
#pragma unroll
            for (uint32_t t = 0; t < 8; ++t) {
                if (i0 + t < L) {
                    uint32_t ch = ((t < 4 ? bw.x : bw.y) >> ((t & 3) * 8)) & 255u;
                    uint32_t qc = ((t < 4 ? qw.x : qw.y) >> ((t & 3) * 8)) & 255u;
                    const uint32_t cl = s_cls[ch];
                    if (prow) { pk |= (cl & 3u) << (((i0 & 8u) + t) << 1); any_n |= cl >> 3; }
                    float v;
                    if (qc < 128u && (cl < 8u || uni)) {
                        v = term[cl][qc];                    // NaN when the probability is negative, like the direct form
                    } else {
                        float2 pq = lut[qc];
                        const float4 sv = S4[ch];
                        const float s4[4] = { sv.x, sv.y, sv.z, sv.w };
                        v = gm_get_val(gm_nt4(ch), pq.x, pq.y, s4);
                    }
                    if (v != v) bad = 1;                     // negative probability (SeqReader.cpp:1171-1189)
                    if (qc >= 128u) high = 1;                // (k_nw_rows' table of row values covers the characters below 128)
                    score = __fadd_rn(score, v);
                }
            }
            if (prow) {
                if ((i0 & 8u) || i0 + 8 >= L) {              // a word of 16 bases is complete (or the read ends)
                    const uint32_t w = i0 >> 4;
                    uint32_t rv = __brev(pk);
                    rv = ((rv >> 1) & 0x55555555u) | ((rv & 0x55555555u) << 1);       // 2-bit groups in reverse order
                    prow[1u + b.pack_w2 - 1u - w] = rv;
                    const uint32_t g = ~pk, bs = rsh & 31u;
                    uint32_t* const rform = prow + b.pack_w2 + 2u + (rsh >> 5);
                    rform[w] = (g << bs) | carry;
                    carry = bs ? g >> (32u - bs) : 0u;
                    if (i0 + 8 >= L) rform[w + 1u] = carry;   // the word above the last one takes what was shifted out (its own padding word when bs = 0)
                    pk = 0;
                }
            }
        }
        if (L < (uint32_t)p.mer) {
            st = -2;                                         // READ_TOO_SHORT
            mn = 0.0;
        } else if (p.nw) {
            float v = 0.0f;                                  // get_align_score(read,cons,0,L-1): begin(=0) + mid + end(=0)
            v = __fadd_rn(v, 0.0f); v = __fadd_rn(v, score); v = __fadd_rn(v, 0.0f);
            self = v;
            if ((double)self < (double)p.cutoff) st = -3;    // READ_TOO_POOR
            mn = p.align_is_fraction ? (double)p.align_score * (double)self : (double)p.align_score;
        } else {
            mn = (double)p.kmin;                             // Driver.cpp:502
        }
        b.status[r] = st;
        if (prow) prow[0] = L | (any_n << 16) | ((st != 0 ? 1u : 0u) << 17);
        b.self_score[r] = self;
        b.min_score[r] = mn;
        b.top_score[r] = 0.0f;
        b.hit_count[r] = 0;
        b.hit_cursor[r] = 0;
    }
    }
    gm_count(b, GMK_BAD_QUAL, bad);
    gm_count(b, GMK_HIGH_QUAL, high);
}


// ------------------------------------------------------------------------------------------------
// seed: one lane per read x strand walks the read exactly like align_sequence does
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 8) k_seed(GmDevIndex ix, GmDevParams p, GmDevBatch b, uint32_t TR) {
    // the TR reads of a tile (2 lanes per read: + and - strand; TR = 128 unless long reads leave less room in LDS: gmk_seed) are
    // staged into LDS with coalesced 16-byte loads
    extern __shared__ __attribute__((aligned(16))) unsigned char s_reads[];
    unsigned long long nk = 0, nocc = 0, nblk = 0, ntab = 0, nseed_all = 0, nent_all = 0;
    // persistent workgroups: a tile is 128 reads; the work counters go to HBM once per wave at the very end (one atomic per
    // wave per tile on six shared addresses used to cost more than the seed search itself: ~90 M atomics/s per address)
    const uint32_t n_tiles = (b.n + TR - 1u) / TR;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint32_t r0 = tile * TR;
    {
        const uint32_t nr = b.n - r0 < TR ? b.n - r0 : TR;
        const size_t bytes = (size_t)nr * b.stride;                       // stride is a multiple of 8; r0 * stride of 16 when stride % 16 == 0
        const unsigned char* src = b.bases + (size_t)r0 * b.stride;
        if ((((size_t)src) & 15) == 0) {
            for (size_t o = (size_t)threadIdx.x * 16; o < bytes; o += 256 * 16) {
                if (o + 16 <= bytes) *reinterpret_cast<uint4*>(s_reads + o) = *reinterpret_cast<const uint4*>(src + o);
                else for (size_t q = o; q < bytes; ++q) s_reads[q] = src[q];
            }
        } else {
            for (size_t o = (size_t)threadIdx.x * 8; o < bytes; o += 256 * 8) *reinterpret_cast<uint2*>(s_reads + o) = *reinterpret_cast<const uint2*>(src + o);
        }
    }
    __syncthreads();
    const uint32_t rs = tile * 2u * TR + threadIdx.x;
    unsigned long long nseed = 0, nent = 0;
    if (threadIdx.x < 2u * TR && rs < 2 * b.n) {
        uint32_t r = rs >> 1, strand = rs & 1;
        bool on = b.status[r] == 0 && (strand ? p.neg_strand : p.pos_strand);
        if (on) {
            uint32_t L = b.len[r];
            const unsigned char* rb = s_reads + (size_t)(r - r0) * b.stride;
            gm_seed_walk(ix, p, rb, L, strand, b.seeds + (size_t)rs * b.max_seeds, b.max_seeds, nk, nocc, nblk, ntab, nseed, nent);
        }
        b.n_seeds[rs] = (uint16_t)(nseed < b.max_seeds ? nseed : b.max_seeds);
        b.n_entries[rs] = nent > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nent;
    }
    nseed_all += nseed; nent_all += nent;
    __syncthreads();                                 // everyone is done with this tile's reads before the next one is staged
    }
    gm_count(b, GMK_KMERS, nk);
    gm_count(b, GMK_OCC, nocc);
    gm_count(b, GMK_OCC_BLOCKS, nblk);
    gm_count(b, GMK_TAB_LOOKUPS, ntab);
    gm_count(b, GMK_SEEDS, nseed_all);
    gm_count(b, GMK_SA_HITS, nent_all);
}

// ------------------------------------------------------------------------------------------------
// exclusive scan u32 -> u64 (three small passes; 1024 items per block)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_scan_block_sums(const uint32_t* in, uint64_t n, unsigned long long* block_sums) {
    __shared__ unsigned long long ws[4];
    uint64_t base = (uint64_t)blockIdx.x * 1024 + (uint64_t)threadIdx.x * 4;
    unsigned long long s = 0;
    for (int q = 0; q < 4; ++q) if (base + q < n) s += in[base + q];
    s = gm_wave_sum(s);
    if (gm_lane() == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ void __launch_bounds__(1024) k_scan_sums(unsigned long long* block_sums, uint32_t nb) {
    // one workgroup of 1024 threads, a contiguous run of the block sums each (10 at 10 M reads): all loads of the run are in flight at
    // once.  (Round 1-3: one wavefront, 64 sums per dependent round trip - 82 us at 10 M reads for 80 KB.)
    __shared__ unsigned long long s_w[16];
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t lo = min(nb, threadIdx.x * per), hi = min(nb, lo + per);
    unsigned long long s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += block_sums[i];
    const int lane = gm_lane(), wave = threadIdx.x >> 6;
    unsigned long long incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned long long run = incl - s, total = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) run += s_w[w]; total += s_w[w]; }
    for (uint32_t i = lo; i < hi; ++i) { const unsigned long long v = block_sums[i]; block_sums[i] = run; run += v; }
    if (threadIdx.x == 0) block_sums[nb] = total;
}

__global__ void __launch_bounds__(256) k_scan_final(const uint32_t* in, uint64_t n, const unsigned long long* block_sums, uint64_t* out) {
    __shared__ unsigned long long ws[4];
    int lane = gm_lane(), wave = threadIdx.x >> 6;
    uint64_t base = (uint64_t)blockIdx.x * 1024 + (uint64_t)threadIdx.x * 4;
    uint32_t v[4];
    unsigned long long s = 0;
    for (int q = 0; q < 4; ++q) { v[q] = base + q < n ? in[base + q] : 0; s += v[q]; }
    unsigned long long incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    unsigned long long pre = block_sums[blockIdx.x];
    for (int w = 0; w < wave; ++w) pre += ws[w];
    unsigned long long run = pre + incl - s;
    for (int q = 0; q < 4; ++q) { if (base + q < n) out[base + q] = run; run += v[q]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = block_sums[gridDim.x];
}

// ------------------------------------------------------------------------------------------------
// faithful locate (sampled SA): one wavefront per read x strand, lanes = SA ranks of the current seed
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_locate_sampled(GmDevIndex ix, GmDevBatch b) {
    uint32_t rs = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = gm_lane();
    unsigned long long steps_total = 0;
    if (rs < 2 * b.n) {
        uint32_t ns = b.n_seeds[rs];
        const GmSeed* seeds = b.seeds + (size_t)rs * b.max_seeds;
        uint64_t off = b.entry_off[rs];
        for (uint32_t t = 0; t < ns; ++t) {
            GmSeed sd = seeds[t];
            uint32_t cnt = sd.l - sd.k + 1;
            for (uint32_t base = 0; base < cnt; base += 64) {
                uint32_t idx = base + lane;
                if (idx < cnt) {
                    uint32_t st;
                    b.coords[off + idx] = gm_locate_walk(ix, sd.k + idx, &st);
                    steps_total += st;
                }
            }
            off += cnt;
        }
    }
    gm_count(b, GMK_LF_STEPS, steps_total);
}

// ------------------------------------------------------------------------------------------------
// faithful locate, the form that fills the lanes: k_locate_sampled above gives a wavefront to a read x strand and its lanes to the SA
// ranks of ONE seed at a time - 12 of 64 lanes at -m 14 on 3.1 Gbp, 1 .. 4 at -m 20, and every seed waits for its longest walk (the
// walks are geometric: mean 31 steps, the longest of 64 ~130).  Here the hits of the whole block are one flat list (coords[], filled
// with the START RANKS by k_locate_ranks) and every LANE walks its own hit; a lane whose walk has reached a sampled rank stores the
// coordinate over the rank and takes its next hit at once, so all lanes stay on LF steps until the list is empty.  One LF step
// (bwt_invPsi, src/bwt.c:53-59: the base at k and bwt_occ(k, base)) reads the 64-byte occ block of k ONCE - the four 16-byte loads of
// one line go out together, the base is picked out of the loaded words: one HBM round trip per step instead of two dependent ones.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_locate_ranks(GmDevBatch b) {
    const uint32_t rs = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = (uint32_t)gm_lane();
    if (rs >= 2 * b.n) return;
    const uint32_t ns = b.n_seeds[rs];
    const GmSeed* seeds = b.seeds + (size_t)rs * b.max_seeds;
    uint64_t off = b.entry_off[rs];
    for (uint32_t t = 0; t < ns; ++t) {
        const GmSeed sd = seeds[t];
        const uint32_t cnt = sd.l - sd.k + 1;
        for (uint32_t idx = lane; idx < cnt; idx += 64) b.coords[off + idx] = sd.k + idx;
        off += cnt;
    }
}

// one LF step with one line fetch: k -> L2[c] + occ(k, c), c = the BWT character at k (k != primary is the caller's business)
__device__ __forceinline__ uint32_t gm_lf_step(const GmDevIndex& ix, const uint32_t k) {
    const uint32_t x = k - ((k > ix.primary) ? 1u : 0u);               // bwt_B0's index (bwt_invPsi :55)
    // bwt_occ(k, c) :107-129 counts bases == c among x' = k - (k >= primary) .. that is x for k != primary: [block start, x]
    const uint4* blk = reinterpret_cast<const uint4*>(ix.bwt + ((size_t)(x >> 7) << 4));
    const uint4 c0 = blk[0], c1 = blk[1], w0 = blk[2], w1 = blk[3];    // cumulative counts A, C (lo, hi each) | G, T | 64 bases | 64 bases
    const uint32_t wi = (x & 0x7fu) >> 4;
    const uint32_t word = wi < 4u ? (wi == 0u ? w0.x : wi == 1u ? w0.y : wi == 2u ? w0.z : w0.w) : (wi == 4u ? w1.x : wi == 5u ? w1.y : wi == 6u ? w1.z : w1.w);
    const uint32_t c = (word >> ((~x & 0xfu) << 1)) & 3u;
    uint32_t n = c == 0u ? c0.x : c == 1u ? c0.z : c == 2u ? c1.x : c1.z;              // low halves of the u64 counts (seq_len < 2^32)
    const int within = (int)(x & 127u) + 1;
    n += gm_count_base(((unsigned long long)w0.x << 32) | w0.y, c, within);
    n += gm_count_base(((unsigned long long)w0.z << 32) | w0.w, c, within - 32);
    n += gm_count_base(((unsigned long long)w1.x << 32) | w1.y, c, within - 64);
    n += gm_count_base(((unsigned long long)w1.z << 32) | w1.w, c, within - 96);
    return gm_L2(ix, c) + n;
}

__global__ void __launch_bounds__(256) k_locate_flat(GmDevIndex ix, GmDevBatch b, const unsigned long long n_entries) {
    const unsigned long long stride = (unsigned long long)gridDim.x * 256ull;
    unsigned long long e = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    unsigned long long steps_total = 0;
    bool have = e < n_entries;
    uint32_t k = have ? b.coords[e] : 0u, sa = 0;
    while (__builtin_amdgcn_ballot_w64(have) != 0ull) {
        if (have) {
            if ((k & ix.sa_mask) == 0u || sa >= GM_WALK_CAP) {              // a sampled rank: bwt_sa's loop ends (src/bwt.c:89-95)
                b.coords[e] = sa + ix.sa_samples[k >> ix.sa_shift];
                steps_total += sa;
                e += stride;
                have = e < n_entries;
                if (have) { k = b.coords[e]; sa = 0; }
            } else {
                k = (k == ix.primary) ? 0u : gm_lf_step(ix, k);             // bwt_invPsi: the rank of $ maps to 0
                ++sa;
            }
        }
    }
    gm_count(b, GMK_LF_STEPS, steps_total);
}

// ------------------------------------------------------------------------------------------------
// locate + vote.  One wavefront owns one read x strand.  Votes are counted in an LDS table; because nearly all
// located positions are singletons, a two-level bit filter (A: seen, B: seen twice) keeps the exact table small:
// only positions whose filter bit was hit at least twice can reach kmin >= 2 votes.
// ------------------------------------------------------------------------------------------------
#define GMV_FBITS 14                    // 16384 filter bits = 512 words
#define GMV_FWORDS (1 << (GMV_FBITS - 5))
#define GMV_TBITS 9                     // 512 exact slots
#define GMV_TSIZE (1 << GMV_TBITS)
#define GMV_TLIMIT (GMV_TSIZE * 3 / 4)


__device__ __forceinline__ uint32_t gm_coord(const GmDevIndex& ix, const GmDevBatch& b, int use_full_sa, const GmSeed& sd,
                                             uint32_t idx, uint64_t coord_off) {
    return use_full_sa ? ix.full_sa[sd.k + idx] : b.coords[coord_off + idx];
}

__global__ void __launch_bounds__(256) k_vote(GmDevIndex ix, GmDevParams p, GmDevBatch b, int use_full_sa) {
    __shared__ uint32_t s_A[4][GMV_FWORDS];
    __shared__ uint32_t s_B[4][GMV_FWORDS];
    __shared__ uint32_t s_keys[4][GMV_TSIZE];
    __shared__ uint32_t s_vals[4][GMV_TSIZE];
    int wave = threadIdx.x >> 6, lane = gm_lane();
    uint32_t rs = blockIdx.x * 4 + wave;
    if (rs >= 2 * b.n) return;                       // wave-uniform; no block barriers below
    uint32_t ns = b.n_seeds[rs];
    if (ns == 0) return;
    uint32_t E = b.n_entries[rs];
    const GmSeed* seeds = b.seeds + (size_t)rs * b.max_seeds;
    uint64_t coff0 = use_full_sa ? 0 : b.entry_off[rs];
    bool filter = p.kmin >= 2 && E > GMV_TSIZE / 2;
    if (p.nw && p.fast) ns = 1;                      // --fast: only the first seed is looked at (:309-312)
    uint32_t* A = s_A[wave]; uint32_t* B = s_B[wave];
    GmLdsTable tb; tb.keys = s_keys[wave]; tb.vals = s_vals[wave]; tb.mask = GMV_TSIZE - 1; tb.bits = GMV_TBITS;
    for (int q = lane; q < GMV_TSIZE; q += 64) { tb.keys[q] = GM_EMPTY; tb.vals[q] = 0; }
    if (filter) {
        for (int q = lane; q < GMV_FWORDS; q += 64) { A[q] = 0; B[q] = 0; }
        uint64_t coff = coff0;
        for (uint32_t t = 0; t < ns; ++t) {          // pass 1: which filter bits are hit twice
            GmSeed sd = seeds[t];
            uint32_t cnt = sd.l - sd.k + 1;
            for (uint32_t base = 0; base < cnt; base += 64) {
                uint32_t idx = base + lane;
                if (idx < cnt) {
                    uint32_t c = gm_coord(ix, b, use_full_sa, sd, idx, coff);
                    uint32_t bp = (c <= sd.pos) ? 0u : c - sd.pos;
                    uint32_t h = (bp * 0x9E3779B1u) >> (32 - GMV_FBITS);
                    uint32_t bit = 1u << (h & 31);
                    uint32_t old = atomicOr(&A[h >> 5], bit);
                    if (old & bit) atomicOr(&B[h >> 5], bit);
                }
            }
            coff += cnt;
        }
    }
    uint32_t nkeys = 0;
    bool overflow = false;
    uint64_t coff = coff0;
    for (uint32_t t = 0; t < ns && !overflow; ++t) { // pass 2: exact votes, in seed order
        GmSeed sd = seeds[t];
        uint32_t cnt = sd.l - sd.k + 1;
        for (uint32_t base = 0; base < cnt && !overflow; base += 64) {
            uint32_t idx = base + lane;
            bool emit = false, fresh = false, full = false;
            uint32_t bp = 0;
            if (idx < cnt) {
                uint32_t c = gm_coord(ix, b, use_full_sa, sd, idx, coff);
                bp = (c <= sd.pos) ? 0u : c - sd.pos;               // :267
                bool take = true;
                if (filter) {
                    uint32_t h = (bp * 0x9E3779B1u) >> (32 - GMV_FBITS);
                    take = (B[h >> 5] >> (h & 31)) & 1u;
                }
                if (take) {
                    uint32_t slot = gm_table_insert(tb, bp, &fresh);
                    if (slot == GM_EMPTY) full = true;
                    else {
                        uint32_t v = atomicAdd(&tb.vals[slot], 1u) + 1u;
                        emit = p.nw && v == (uint32_t)p.kmin;       // reaches -k votes at this seed -> NW now (:28-40)
                    }
                }
            }
            nkeys += (uint32_t)__popcll(__ballot(fresh));
            if (__ballot(full) != 0 || nkeys > GMV_TLIMIT) overflow = true;
            if (!overflow) gm_emit<GmLdsTable>(b, emit, rs, bp, t, 4);
        }
        coff += cnt;
    }
    if (overflow) {                                  // hand this read x strand to the global-table kernel
        if (lane == 0) {
            b.rs_overflow[rs] = 1;
            uint32_t j = atomicAdd(b.n_retry, 1u);
            uint32_t need = 2 * E; uint32_t sz = 1024; while (sz < need && sz < 0x80000000u) sz <<= 1;
            unsigned long long off = atomicAdd(&b.counters[GMK_HEAVY_SLOTS], (unsigned long long)sz);
            b.retry_list[j] = rs;
            b.retry_off[j] = off;
            atomicAdd(&b.counters[GMK_OVERFLOW_RS], 1ull);
        }
        return;
    }
    if (!p.nw) {                                     // --no_nw: one pass over the final counts (:317-325)
        for (int q = lane; q < GMV_TSIZE; q += 64) {
            uint32_t key = tb.keys[q], v = tb.vals[q];
            bool emit = key != GM_EMPTY && v >= (uint32_t)p.kmin;
            gm_emit<GmLdsTable>(b, emit, rs, key, v > 65535u ? 65535u : v, 4);
        }
    }
}

// ---- order-free vote kernel (the fast path) ---------------------------------------------------------------------
// The seed order only matters for WHEN a position reaches kmin votes (that is the seed step at which the reference
// runs NW on it).  Per exact-table slot we therefore keep, besides the vote count, a bit mask of the seed steps that
// voted: all votes of one position come from different seeds (one SA interval holds each text position once), so the
// step at which the count reaches kmin is the kmin-th lowest set bit.  The only position that can collect several votes
// from ONE seed is the clamped b = 0 (c <= i, :267); it gets its own per-step counters.  With the order gone, the SA
// hits of all seeds are one flat list: every lane keeps GMV_U coalesced loads in flight instead of one.
#define GMV_U 4
template <bool MASK64>
__device__ __forceinline__ void gm_vote_fast_rs(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, int use_full_sa, uint32_t rs) {
    __shared__ uint32_t s_A[4][GMV_FWORDS];          // pass 1: "seen" bits; afterwards reused as the low step masks
    __shared__ uint32_t s_B[4][GMV_FWORDS];          // "seen twice" bits
    __shared__ uint32_t s_keys[4][GMV_TSIZE];
    __shared__ uint32_t s_vals[4][GMV_TSIZE];
    __shared__ uint32_t s_hi[MASK64 ? 4 : 1][MASK64 ? GMV_TSIZE : 1];
    __shared__ uint32_t s_pre[4][66];                // exclusive prefix of the seeds' hit counts
    __shared__ uint32_t s_cnt0[4][64];               // votes of b = 0 per seed step
    const int wave = threadIdx.x >> 6, lane = gm_lane();
    if (rs >= 2 * b.n) return;                       // wave-uniform; no block barriers below
    uint32_t ns = b.n_seeds[rs];
    if (ns == 0) return;
    if (p.nw && p.fast) ns = 1;                      // --fast: only the first seed is looked at (:309-312)
    const GmSeed* seeds = b.seeds + (size_t)rs * b.max_seeds;
    uint32_t* A = s_A[wave]; uint32_t* B = s_B[wave]; uint32_t* pre = s_pre[wave]; uint32_t* cnt0 = s_cnt0[wave];
    uint32_t* mlo = A; uint32_t* mhi = s_hi[MASK64 ? wave : 0];
    GmLdsTable tb; tb.keys = s_keys[wave]; tb.vals = s_vals[wave]; tb.mask = GMV_TSIZE - 1; tb.bits = GMV_TBITS;
    // seeds of this read x strand live in lanes 0..ns-1 (ns <= 64 on this path)
    GmSeed mine; mine.k = 0; mine.l = 0; mine.pos = 0;
    uint32_t cnt = 0;
    if ((uint32_t)lane < ns) { mine = seeds[lane]; cnt = mine.l - mine.k + 1; }
    uint32_t incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    pre[lane + 1] = incl;
    if (lane == 0) pre[0] = 0;
    cnt0[lane] = 0;
    const uint32_t E = __shfl(incl, 63);
    const uint64_t coff0 = use_full_sa ? 0 : b.entry_off[rs];
    const bool filter = p.kmin >= 2 && E > GMV_TSIZE / 2;
    for (int q = lane; q < GMV_TSIZE; q += 64) { tb.keys[q] = GM_EMPTY; tb.vals[q] = 0; if (MASK64) mhi[q] = 0; }
    for (int q = lane; q < GMV_FWORDS; q += 64) { A[q] = 0; B[q] = 0; }

    // one sweep over the flat hit list; PASS 1 fills the filter, PASS 2 votes
    auto sweep = [&](const int pass, bool& overflow) {
        uint32_t nkeys = 0;
        uint32_t t0 = 0;                             // seed of the first entry of the current group (wave-uniform)
        for (uint32_t base = 0; base < E && !overflow; base += 64 * GMV_U) {
            while (t0 + 1 < ns && pre[t0 + 1] <= base) ++t0;
            uint32_t cc[GMV_U], tt[GMV_U];
            bool vv[GMV_U];
#pragma unroll
            for (int u = 0; u < GMV_U; ++u) {        // which seed does each entry belong to
                uint32_t e = base + (uint32_t)u * 64 + (uint32_t)lane;
                vv[u] = e < E;
                uint32_t t = t0;
                if (vv[u]) { while (t + 1 < ns && pre[t + 1] <= e) ++t; }
                tt[u] = t;
                cc[u] = 0;
            }
#pragma unroll
            for (int u = 0; u < GMV_U; ++u) {
                uint32_t e = base + (uint32_t)u * 64 + (uint32_t)lane;
                uint32_t kk = __shfl(mine.k, (int)tt[u]);      // every lane takes part in the shuffle; all loads of the group are issued here
                if (vv[u]) cc[u] = use_full_sa ? ix.full_sa[kk + (e - pre[tt[u]])] : b.coords[coff0 + e];
            }
#pragma unroll
            for (int u = 0; u < GMV_U; ++u) {
                uint32_t sp = __shfl(mine.pos, (int)tt[u]);
                bool fresh = false, full = false;
                if (vv[u]) {
                    uint32_t c = cc[u];
                    uint32_t bp = (c <= sp) ? 0u : c - sp;                      // :267
                    uint32_t h = (bp * 0x9E3779B1u) >> (32 - GMV_FBITS);
                    uint32_t bit = 1u << (h & 31);
                    if (pass == 1) {
                        if (bp != 0) { uint32_t old = atomicOr(&A[h >> 5], bit); if (old & bit) atomicOr(&B[h >> 5], bit); }
                    } else if (bp == 0) {
                        atomicAdd(&cnt0[tt[u]], 1u);
                    } else if (!filter || (B[h >> 5] & bit)) {
                        uint32_t slot = gm_table_insert(tb, bp, &fresh);
                        if (slot == GM_EMPTY) full = true;
                        else {
                            atomicAdd(&tb.vals[slot], 1u);
                            if (tt[u] < 32) atomicOr(&mlo[slot], 1u << tt[u]);
                            else if (MASK64) atomicOr(&mhi[slot], 1u << (tt[u] - 32));
                        }
                    }
                }
                if (pass == 2) {
                    nkeys += (uint32_t)__popcll(__ballot(fresh));
                    if (__ballot(full) != 0 || nkeys > GMV_TLIMIT) overflow = true;
                }
            }
        }
    };
    bool overflow = false;
    if (filter) sweep(1, overflow);
    // the low step masks reuse A: clear it now that pass 1 is over
    for (int q = lane; q < GMV_TSIZE; q += 64) mlo[q] = 0;
    sweep(2, overflow);
    if (overflow) {                                  // hand this read x strand to the global-table kernel
        if (lane == 0) {
            b.rs_overflow[rs] = 1;
            uint32_t j = atomicAdd(b.n_retry, 1u);
            uint32_t need = 2 * E; uint32_t sz = 1024; while (sz < need && sz < 0x80000000u) sz <<= 1;
            unsigned long long off = atomicAdd(&b.counters[GMK_HEAVY_SLOTS], (unsigned long long)sz);
            b.retry_list[j] = rs;
            b.retry_off[j] = off;
            atomicAdd(&b.counters[GMK_OVERFLOW_RS], 1ull);
        }
        return;
    }
    // emit every position with >= kmin votes; its NW step is the kmin-th lowest step that voted for it
    for (int q = lane; q < GMV_TSIZE; q += 64) {
        uint32_t key = tb.keys[q], v = tb.vals[q];
        bool emit = key != GM_EMPTY && v >= (uint32_t)p.kmin;
        uint32_t step = 0;
        if (emit) {
            if (p.nw) {
                unsigned long long m = (unsigned long long)mlo[q] | (MASK64 ? ((unsigned long long)mhi[q] << 32) : 0ull);
                for (int r = 1; r < p.kmin && m; ++r) m &= m - 1;
                step = m ? (uint32_t)(__ffsll((long long)m) - 1) : 0u;
            } else step = v > 65535u ? 65535u : v;
        }
        gm_emit<GmLdsTable>(b, emit, rs, key, step, 4);
    }
    {   // b = 0: cumulative per-step counts
        uint32_t c0 = cnt0[lane], run = c0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(run, off); if (lane >= off) run += t; }
        uint32_t total = __shfl(run, 63);
        unsigned long long reached = __ballot(run >= (uint32_t)p.kmin);
        bool emit = lane == 0 && total >= (uint32_t)p.kmin;
        uint32_t step = p.nw ? (uint32_t)(__ffsll((long long)reached) - 1) : (total > 65535u ? 65535u : total);
        gm_emit<GmLdsTable>(b, emit, rs, 0u, step, 4);
    }
}

template <bool MASK64>
__global__ void __launch_bounds__(256) k_vote_fast(GmDevIndex ix, GmDevParams p, GmDevBatch b, int use_full_sa) {
    gm_vote_fast_rs<MASK64>(ix, p, b, use_full_sa, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// the same, driven by the list of read x strands that k_vote_sparse found too large for its sub-wave groups
template <bool MASK64>
__global__ void __launch_bounds__(256) k_vote_fast_list(GmDevIndex ix, GmDevParams p, GmDevBatch b, int use_full_sa) {
    const uint32_t n_big = *b.n_big;
    for (uint32_t it = blockIdx.x * 4 + (threadIdx.x >> 6); it < n_big; it += gridDim.x * 4)
        gm_vote_fast_rs<MASK64>(ix, p, b, use_full_sa, b.big_list[it]);
}

// ---- sparse seeds (a handful of SA hits per read x strand): 16 lanes per read x strand, votes by all-pairs compare ----
// Lane e of a group holds hit e of the flat (seed-major) hit list.  rank = equal positions earlier in the list, cnt = equal
// positions in all; the hit with rank == kmin-1 is the vote that makes the position reach -k, and its seed is the NW step
// (flat order is seed order; this also covers the clamped b = 0 with several hits of one seed).  No LDS tables at all.
__global__ void __launch_bounds__(256) k_vote_sparse(GmDevIndex ix, GmDevParams p, GmDevBatch b, int use_full_sa) {
    const int lane = gm_lane(), l = lane & 15;
    const uint32_t rs = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (uint32_t)(lane >> 4);
    const bool in = rs < 2 * b.n;
    uint32_t ns = in ? b.n_seeds[rs] : 0;
    if (p.nw && p.fast && ns > 1) ns = 1;
    GmSeed mine; mine.k = 0; mine.l = 0; mine.pos = 0;
    uint32_t cnt = 0;
    if ((uint32_t)l < ns && ns <= 16) { mine = b.seeds[(size_t)rs * b.max_seeds + l]; cnt = mine.l - mine.k + 1; }
    uint32_t incl = cnt;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) { uint32_t t = __shfl_up(incl, off, 16); if (l >= off) incl += t; }
    const uint32_t E = __shfl(incl, 15, 16);
    const bool big = ns > 16 || E > 16;              // group-uniform
    if (big && l == 0) { uint32_t at = atomicAdd(b.n_big, 1u); b.big_list[at] = rs; }
    // which seed does hit e = l belong to
    uint32_t t = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { uint32_t iq = __shfl(incl, q, 16); if ((uint32_t)q < ns && iq <= (uint32_t)l) t = (uint32_t)q + 1; }
    const bool valid = in && !big && (uint32_t)l < E;
    const uint32_t tt = t < 16 ? t : 15;
    const uint32_t kt = __shfl(mine.k, (int)tt, 16), pt = __shfl(mine.pos, (int)tt, 16);
    const uint32_t pre_t = __shfl(incl - cnt, (int)tt, 16);
    uint32_t bp = 0;
    if (valid) {
        uint32_t c = use_full_sa ? ix.full_sa[kt + ((uint32_t)l - pre_t)] : b.coords[b.entry_off[rs] + (uint32_t)l];
        bp = (c <= pt) ? 0u : c - pt;                 // :267
    }
    uint32_t rank = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        uint32_t ob = __shfl(bp, q, 16);
        int ov = __shfl((int)valid, q, 16);
        bool eq = valid && ov && ob == bp;
        total += eq ? 1u : 0u;
        rank += (eq && q < l) ? 1u : 0u;
    }
    bool emit; uint32_t step;
    if (p.nw) { emit = valid && total >= (uint32_t)p.kmin && rank == (uint32_t)p.kmin - 1; step = tt; }
    else { emit = valid && total >= (uint32_t)p.kmin && rank == 0; step = total; }
    gm_emit<GmLdsTable>(b, emit, rs, bp, step, 0);
}

// ---- order-free vote kernel, one WORKGROUP per read x strand (dense seeds: tens of SA hits per seed and more) ----
// Same algorithm as k_vote_fast, but the flat hit list of the read x strand is spread over NT lanes: every lane issues
// all of its loads at once - one HBM latency per read x strand - and keeps the located positions in registers.
// LDS atomics run at roughly one wave-instruction per 25 cycles per CU, so they are rationed:
//   pass 1   one NON-returning ds_add per hit into a counting filter (8192 slots x 8 bit)
//   pass 2a  hits whose filter slot counted >= 2 (about one in five) are compacted into an LDS list with ballots and plain
//            stores - no atomics
//   pass 2b  only the compacted list goes through the exact table (CAS + two non-returning atomics), with full waves
// The filter region is dead after pass 2a and becomes the exact table.  Reads x strands with more than GMB_ENTRIES hits
// take the round-based path with the bit filter (as k_vote_fast).
#define GMB_ENTRIES 2048        // hits a workgroup keeps in registers (NT lanes x GMB_ENTRIES/NT each)
#define GMB_FBITS 13            // counting filter slots
#define GMB_FWORDS (1 << (GMB_FBITS - 2))
#define GMB_TBITS 10            // exact table of the one-round path
#define GMB_TSIZE (1 << GMB_TBITS)
#define GMB_TLIMIT (GMB_TSIZE * 3 / 4)
#define GMB_LCAP 768            // compacted list entries per workgroup
#define GMB_FSAT 200u           // a filter byte this high could have wrapped: hand the read x strand to the retry kernel
template <bool MASK64, int NT, int TB>
__global__ void __launch_bounds__(NT) k_vote_block(GmDevIndex ix, GmDevParams p, GmDevBatch b, int use_full_sa) {
    constexpr int GMB_U = GMB_ENTRIES / NT;
    constexpr int NW = NT / 64;
    constexpr int LSEG = GMB_LCAP / NW;
    constexpr int TS1 = 1 << TB;                     // exact table of the one-round path: keys | vals | low masks live in s_r0
    constexpr uint32_t TLIM1 = TB >= 10 ? TS1 * 3 / 4 : TS1 * 7 / 8;
    static_assert(3 * TS1 <= 2048 || TB == 10, "table must fit the filter region");
    __shared__ uint32_t s_r0[2048];                  // 8 KB: counting filter, then keys[1024] + vals[1024]  (round path: A,B,keys,vals x 512)
    __shared__ uint32_t s_mlo[TB >= 10 ? GMB_TSIZE : 1];              // low step masks (TB = 9: in s_r0 / in the unused list memory on the round path)
    __shared__ uint32_t s_mhi[MASK64 ? GMB_TSIZE : 1];
    __shared__ uint32_t s_lbp[GMB_LCAP];
    __shared__ uint8_t s_lt[GMB_LCAP];
    __shared__ uint32_t s_pre[66], s_k[64], s_pos[64], s_cnt0[64], s_chunk[64];
    __shared__ uint32_t s_nkeys, s_full, s_lcnt[NW];
    const uint32_t rs = blockIdx.x;                  // grid = 2n
    const int tid = threadIdx.x, lane = gm_lane(), wave = tid >> 6;
    uint32_t ns = b.n_seeds[rs];
    if (ns == 0) return;                             // block-uniform
    if (p.nw && p.fast) ns = 1;
    const GmSeed* seeds = b.seeds + (size_t)rs * b.max_seeds;
    if (tid < 64) {                                  // wave 0: seeds -> LDS, exclusive prefix of their hit counts
        GmSeed mine; mine.k = 0; mine.l = 0; mine.pos = 0;
        uint32_t cnt = 0;
        if ((uint32_t)tid < ns) { mine = seeds[tid]; cnt = mine.l - mine.k + 1; }
        uint32_t incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        s_pre[tid + 1] = incl; s_k[tid] = mine.k; s_pos[tid] = mine.pos; s_cnt0[tid] = 0;
        if (tid == 0) { s_pre[0] = 0; s_nkeys = 0; s_full = 0; }
        // seed of the first entry of every 32-entry chunk (the per-entry search then needs one or two steps); s_pre was
        // written by this same wave just above, LDS operations of one wave complete in order
        const uint32_t E0 = __shfl(incl, 63);
        uint32_t e = (uint32_t)tid * 32u, t = 0;
        if (e < E0) { uint32_t lo = 0, hi = ns; while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (s_pre[mid] <= e) lo = mid; else hi = mid; } t = lo; }
        s_chunk[tid] = t;
    }
    for (int q = tid; q < 2048; q += NT) s_r0[q] = 0;
    __syncthreads();
    const uint32_t E = s_pre[ns];
    const uint64_t coff0 = use_full_sa ? 0 : b.entry_off[rs];
    uint32_t bpv[GMB_U], ttv[GMB_U];

    auto load_round = [&](uint32_t base) {           // located position (as window start) of up to GMB_U entries per lane
        uint32_t ee[GMB_U];
#pragma unroll
        for (int u = 0; u < GMB_U; ++u) {
            uint32_t e = base + (uint32_t)u * (uint32_t)NT + (uint32_t)tid;
            ee[u] = e;
            uint32_t t = 0xFFFFFFFFu;
            if (e < E) {
                if (base == 0) { t = s_chunk[e >> 5]; while (t + 1 < ns && s_pre[t + 1] <= e) ++t; }
                else { uint32_t lo = 0, hi = ns; while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (s_pre[mid] <= e) lo = mid; else hi = mid; } t = lo; }
            }
            ttv[u] = t;
        }
        uint32_t cc[GMB_U];
#pragma unroll
        for (int u = 0; u < GMB_U; ++u) {            // all loads in flight together
            cc[u] = 0;
            if (ttv[u] != 0xFFFFFFFFu) cc[u] = use_full_sa ? ix.full_sa[s_k[ttv[u]] + (ee[u] - s_pre[ttv[u]])] : b.coords[coff0 + ee[u]];
        }
#pragma unroll
        for (int u = 0; u < GMB_U; ++u) {
            uint32_t sp = ttv[u] != 0xFFFFFFFFu ? s_pos[ttv[u]] : 0u;
            bpv[u] = (cc[u] <= sp) ? 0u : cc[u] - sp;                       // :267
        }
    };
    if (p.dbg & 1) {                                 // timing experiment: loads only
        uint32_t acc = 0;
        for (uint32_t base = 0; base < E; base += (uint32_t)NT * GMB_U) { load_round(base);
#pragma unroll
            for (int u = 0; u < GMB_U; ++u) acc ^= bpv[u]; }
        if (acc == 0x12345678u) b.rs_overflow[rs] = 3;
        return;
    }

    GmLdsTable tb;
    uint32_t* mlo = s_mlo;
    bool failed = false;
    if (E <= (uint32_t)NT * GMB_U) {
        // ------------------------------------------------------------ one round: everything stays in registers
        const bool filter = p.kmin >= 2 && E > 256;
        load_round(0);
        if (filter && !(p.dbg & 2)) {
#pragma unroll
            for (int u = 0; u < GMB_U; ++u)
                if (ttv[u] != 0xFFFFFFFFu && bpv[u] != 0) {
                    uint32_t h = (bpv[u] * 0x9E3779B1u) >> (32 - GMB_FBITS);
                    atomicAdd(&s_r0[h >> 2], 1u << ((h & 3) << 3));         // result unused: non-returning ds_add
                }
        }
        __syncthreads();
        uint32_t wcount = 0;                         // wave-uniform fill of this wave's list segment
#pragma unroll
        for (int u = 0; u < GMB_U; ++u) {
            bool pass = false;
            if (ttv[u] != 0xFFFFFFFFu) {
                if (bpv[u] == 0) atomicAdd(&s_cnt0[ttv[u]], 1u);
                else if (!filter) pass = true;
                else {
                    uint32_t h = (bpv[u] * 0x9E3779B1u) >> (32 - GMB_FBITS);
                    pass = ((s_r0[h >> 2] >> ((h & 3) << 3)) & 255u) >= 2u;
                }
            }
            unsigned long long m = __ballot(pass);
            if (pass) {
                uint32_t at = wcount + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (at < (uint32_t)LSEG) { s_lbp[wave * LSEG + at] = bpv[u]; s_lt[wave * LSEG + at] = (uint8_t)ttv[u]; }
            }
            wcount += (uint32_t)__popcll(m);
        }
        if (lane == 0) { s_lcnt[wave] = wcount < (uint32_t)LSEG ? wcount : (uint32_t)LSEG; if (wcount > (uint32_t)LSEG) s_full = 1; }
        __syncthreads();
        // the filter is dead: check it for saturation while turning its memory into the exact table
        for (int q = tid; q < 2048; q += NT) {
            uint32_t w = s_r0[q];
            if (filter && ((w & 255u) >= GMB_FSAT || ((w >> 8) & 255u) >= GMB_FSAT || ((w >> 16) & 255u) >= GMB_FSAT || (w >> 24) >= GMB_FSAT)) s_full = 1;
            s_r0[q] = q < TS1 ? GM_EMPTY : 0u;
        }
        if (TB >= 10) for (int q = tid; q < TS1; q += NT) s_mlo[q] = 0;
        if (MASK64) for (int q = tid; q < TS1; q += NT) s_mhi[q] = 0;
        __syncthreads();
        tb.keys = s_r0; tb.vals = s_r0 + TS1; tb.mask = TS1 - 1; tb.bits = TB;
        mlo = TB >= 10 ? s_mlo : s_r0 + 2 * TS1;
        if (!(p.dbg & 4)) {
            const uint32_t n_l = s_lcnt[wave];
            uint32_t nfresh = 0; bool full = false;
            for (uint32_t i0 = 0; i0 < n_l; i0 += 64) {
                uint32_t i = i0 + (uint32_t)lane;
                bool fresh = false;
                if (i < n_l) {
                    uint32_t bp = s_lbp[wave * LSEG + i], t = s_lt[wave * LSEG + i];
                    uint32_t slot = gm_table_insert(tb, bp, &fresh);
                    if (slot == GM_EMPTY) full = true;
                    else {
                        atomicAdd(&tb.vals[slot], 1u);
                        if (t < 32) atomicOr(&mlo[slot], 1u << t);
                        else if (MASK64) atomicOr(&s_mhi[slot], 1u << (t - 32));
                    }
                }
                nfresh += (uint32_t)__popcll(__ballot(fresh));
            }
            if (lane == 0 && nfresh) atomicAdd(&s_nkeys, nfresh);
            if (full) s_full = 1;
        }
        __syncthreads();
        failed = s_full || s_nkeys > TLIM1;
    } else {
        // ------------------------------------------------------------ rounds of GMB_ENTRIES hits, bit filter (as k_vote_fast)
        uint32_t* A = s_r0; uint32_t* B = s_r0 + 512;
        tb.keys = s_r0 + 1024; tb.vals = s_r0 + 1536; tb.mask = GMV_TSIZE - 1; tb.bits = GMV_TBITS;
        mlo = TB >= 10 ? s_mlo : s_lbp;
        const bool filter = p.kmin >= 2;
        for (int q = tid; q < GMV_TSIZE; q += NT) { tb.keys[q] = GM_EMPTY; mlo[q] = 0; if (MASK64) s_mhi[q] = 0; }
        __syncthreads();
        if (filter)
            for (uint32_t base = 0; base < E; base += (uint32_t)NT * GMB_U) {
                load_round(base);
#pragma unroll
                for (int u = 0; u < GMB_U; ++u)
                    if (ttv[u] != 0xFFFFFFFFu && bpv[u] != 0) {
                        uint32_t h = (bpv[u] * 0x9E3779B1u) >> (32 - GMV_FBITS), bit = 1u << (h & 31);
                        uint32_t old = atomicOr(&A[h >> 5], bit);
                        if (old & bit) atomicOr(&B[h >> 5], bit);
                    }
            }
        __syncthreads();
        for (uint32_t base = 0; base < E; base += (uint32_t)NT * GMB_U) {
            load_round(base);
            uint32_t nfresh = 0; bool full = false;
#pragma unroll
            for (int u = 0; u < GMB_U; ++u)
                if (ttv[u] != 0xFFFFFFFFu) {
                    uint32_t bp = bpv[u], t = ttv[u];
                    uint32_t h = (bp * 0x9E3779B1u) >> (32 - GMV_FBITS), bit = 1u << (h & 31);
                    if (bp == 0) atomicAdd(&s_cnt0[t], 1u);
                    else if (!filter || (B[h >> 5] & bit)) {
                        bool fresh;
                        uint32_t slot = gm_table_insert(tb, bp, &fresh);
                        if (slot == GM_EMPTY) full = true;
                        else {
                            nfresh += fresh ? 1u : 0u;
                            atomicAdd(&tb.vals[slot], 1u);
                            if (t < 32) atomicOr(&mlo[slot], 1u << t);
                            else if (MASK64) atomicOr(&s_mhi[slot], 1u << (t - 32));
                        }
                    }
                }
            if (nfresh) atomicAdd(&s_nkeys, nfresh);
            if (full) s_full = 1;
            __syncthreads();
            if (s_full || s_nkeys > GMV_TLIMIT) break;                       // block-uniform
        }
        __syncthreads();
        failed = s_full || s_nkeys > GMV_TLIMIT;
    }
    if (failed) {                                    // hand this read x strand to the global-table kernel
        if (tid == 0) {
            b.rs_overflow[rs] = 1;
            uint32_t j = atomicAdd(b.n_retry, 1u);
            uint32_t need = 2 * E; uint32_t sz = 1024; while (sz < need && sz < 0x80000000u) sz <<= 1;
            unsigned long long off = atomicAdd(&b.counters[GMK_HEAVY_SLOTS], (unsigned long long)sz);
            b.retry_list[j] = rs;
            b.retry_off[j] = off;
            atomicAdd(&b.counters[GMK_OVERFLOW_RS], 1ull);
        }
        return;
    }
    for (uint32_t q = (uint32_t)tid; q <= tb.mask; q += NT) {      // emit: NW step = kmin-th lowest step that voted
        uint32_t key = tb.keys[q], v = tb.vals[q];
        bool emit = key != GM_EMPTY && v >= (uint32_t)p.kmin;
        uint32_t step = 0;
        if (emit) {
            if (p.nw) {
                unsigned long long m = (unsigned long long)mlo[q] | (MASK64 ? ((unsigned long long)s_mhi[q] << 32) : 0ull);
                for (int r = 1; r < p.kmin && m; ++r) m &= m - 1;
                step = m ? (uint32_t)(__ffsll((long long)m) - 1) : 0u;
            } else step = v > 65535u ? 65535u : v;
        }
        gm_emit<GmLdsTable>(b, emit, rs, key, step, 4);
    }
    if (tid < 64) {                                  // b = 0: cumulative per-step counts
        uint32_t run = s_cnt0[tid];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(run, off); if (lane >= off) run += t; }
        uint32_t total = __shfl(run, 63);
        unsigned long long reached = __ballot(run >= (uint32_t)p.kmin);
        bool emit = tid == 0 && total >= (uint32_t)p.kmin;
        uint32_t step = p.nw ? (uint32_t)(__ffsll((long long)reached) - 1) : (total > 65535u ? 65535u : total);
        gm_emit<GmLdsTable>(b, emit, rs, 0u, step, 4);
    }
}


// ---- k_vote_slots: the dense-seed vote kernel with wave-uniform seeds ------------------------------------------------
// Counters (rocprofv3 --pmc, 1 M reads) show k_vote_block / k_vote_pipe issue ~85 vector and ~95 scalar instructions per
// wave per 64 hits; most of it is per-lane bookkeeping (which seed does entry e belong to, is the entry valid, exec-mask
// juggling).  Here the hit list of a read x strand is cut into SLOTS of up to 64 consecutive SA ranks of ONE seed
// (ceil(hits / 64) slots per seed, seed-major).  Wave 0 turns the seeds into slot descriptors in LDS; a wave then handles
// whole slots, so the seed (SA rank of lane 0, read offset, step tag, number of valid lanes) is wave-uniform: it lives in
// scalar registers, the load address is `scalar base + 4 * lane`, validity is one lane compare, the b = 0 votes are one
// ballot + popcount per slot, and every per-slot branch is a scalar branch.  The LDS phases are those of k_vote_pipe:
// counting filter (non-returning ds_add) -> compacted list -> exact table with key 0 = empty and a list of inserted slots.
// Read x strands that need more than GMS_SMAX slots go to k_vote_fast_list through b.big_list.

// ---- seeds of the vote kernels that take one read x strand per wave / workgroup ---------------------------------------------------------------------------------------
// SEED = false: the read x strand's row of b.seeds, written by k_seed.
// SEED = true (GmDevParams::fused; full SA, the k-mer table covers the whole seed): the wave looks its seeds up itself and no seed
// row goes through HBM.  While nothing fails the adaptive walk (gm_seed_walk) visits i = 0, jump, 2 jump ... < L - mer, so lane j
// takes the k-mer at j * jump: its table code is 2 * mer bits of the read's 2-bit form (k_prep: GmDevBatch::pack), one 16-byte
// table record gives the interval.  A k-mer that does not occur or exceeds -h changes the positions of all later ones, and so does
// a non-ACGT base: then lane 0 runs the walk itself, exactly as k_seed does, into LDS.  Either way the seeds are those of k_seed.
// n_seeds / n_entries of every read x strand still go to HBM (6 bytes): the heavy-path routing and the work counters
// (k_heavy_collect) read them; the kernels a read x strand may be handed to (list, retry, heavy) get its seed row written first.
// The serial walk of the rare cases is kept OUT OF LINE: inlined, the index fields it needs were loaded at the top of every wave
// and pushed ~40 scalar registers into vector lanes (one vector instruction each, on a kernel that is bound by vector issue).
// The callee finds the kernel's arguments where the hardware put them (the kernarg segment: ix, p, b in this order).
// The walk of a read x strand whose regular positions do not all succeed, by the WHOLE wave (all 64 lanes call it): round after
// round, lane j probes the k-mer at pos0 + j * jump; the seeds before the first k-mer that changes the walk are final, the walk
// resumes behind it exactly as gm_seed_walk does (a k-mer whose last d characters do not occur: i += mer - d + 1; one above -h:
// i += 1).  One probe round trip per failing k-mer instead of one per k-mer.  Reads with a non-ACGT base are not handled here
// (the 2-bit forms do not say where it is): they take gm_seed_walk_ool.  out: LDS, max_seeds seeds.  Returns the number of seeds.
__device__ __attribute__((noinline)) uint32_t gm_seed_rewalk_ool(const GmKArgs* a, const uint32_t rs, const int lane, GmSeed* out) {
    const GmDevParams& p = a->p;
    const GmDevBatch& b = a->b;
    const uint32_t m = (uint32_t)p.mer, jump = (uint32_t)p.jump, w2 = b.pack_w2;
    const uint32_t* const row = b.pack + (size_t)(rs >> 1) * b.pack_words;
    const uint32_t* const form = row + ((rs & 1u) ? w2 + 2u : 1u);
    const uint32_t L = row[0] & 0xFFFFu, last = L - m;
    const uint32_t cmask = m >= 16u ? 0xFFFFFFFFu : ((1u << (2u * m)) - 1u);
    uint32_t pos0 = 0, nseed = 0;
    unsigned long long extra = 0;                    // k-mers searched beyond one per seed
    while (pos0 < last) {
        const uint32_t i = pos0 + (uint32_t)lane * jump;
        const bool act = i < last;
        uint32_t k = 0, cnt = 0, t = 0;
        bool fail = false, capped = false;
        if (act) {
            const uint32_t o = 2u * (16u * w2 - i - m);
            const uint32_t code = (uint32_t)((((unsigned long long)form[(o >> 5) + 1u] << 32) | form[o >> 5]) >> (o & 31u)) & cmask;
            bool answered = false;
            if (p.kmer_ctab) {                       // the table probe of gm_seed_walk
                const uint4 rec = p.kmer_ctab[code >> 3];
                const uint32_t sub = code & 7u;
                const unsigned long long cw = (unsigned long long)rec.y | ((unsigned long long)rec.z << 32);
                cnt = (uint32_t)(cw >> (8 * sub)) & 255u;
                if (rec.w == 0u && cnt >= 224u) { fail = true; t = m - (cnt - 223u); answered = true; }
                else if (rec.w == 0u) {
                    unsigned long long below = sub ? (cw & (~0ull >> (64 - 8 * sub))) : 0ull;
                    const unsigned long long emp = below & (below << 1) & (below << 2) & 0x8080808080808080ull;
                    below &= ~((emp >> 7) * 0xFFull);
                    const unsigned long long s2 = (below & 0x00FF00FF00FF00FFull) + ((below >> 8) & 0x00FF00FF00FF00FFull);
                    k = rec.x + (uint32_t)((s2 * 0x0001000100010001ull) >> 48);
                    answered = true;
                }
            }
            if (!answered) {
                const uint2 iv = p.kmer_tab[code];
                if (iv.x == 0xFFFFFFFFu) { fail = true; t = m - iv.y; }
                else { k = iv.x; cnt = iv.y - iv.x + 1u; }
            }
            if (!fail && p.hcap > 0 && cnt > p.hcap) { fail = true; capped = true; }
        }
        const unsigned long long fm = __builtin_amdgcn_ballot_w64(fail);
        const uint32_t nact = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(act));
        const uint32_t f = fm ? (uint32_t)(__ffsll((long long)fm) - 1) : nact;            // lanes before f hold final seeds
        if ((uint32_t)lane < f && nseed + (uint32_t)lane < b.max_seeds) { GmSeed sd; sd.k = k; sd.l = k + cnt - 1u; sd.pos = i; out[nseed + (uint32_t)lane] = sd; }
        nseed += f;
        if (!fm) break;
        ++extra;
        const uint32_t t_f = (uint32_t)__builtin_amdgcn_readlane((int)t, (int)f), cap_f = (uint32_t)__builtin_amdgcn_readlane((int)(capped ? 1u : 0u), (int)f);
        pos0 = pos0 + f * jump + (cap_f ? 1u : t_f + 1u);
    }
    if (lane == 0 && extra) { atomicAdd(&b.counters[GMK_KMERS], extra); atomicAdd(&b.counters[GMK_TAB_LOOKUPS], extra); }
    return nseed < b.max_seeds ? nseed : b.max_seeds;
}

// returns false when the wave has nothing more to do (no seeds, or handed to the heavy path)
template <bool SEED>
__device__ __forceinline__ bool gm_tiny_seeds(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, const uint32_t rs, const int lane,
                                              GmSeed* scratch /* LDS, 64 seeds, free to use */, GmSeed& sd, uint32_t& ns, uint32_t& ie_all) {
    sd.k = 0; sd.l = 0; sd.pos = 0;
    ie_all = 0xFFFFFFFFu;                             // inclusive scan of the seeds' hit counts when the function has computed it
    if (!SEED) {
        if ((uint32_t)lane < b.max_seeds) sd = b.seeds[(size_t)rs * b.max_seeds + lane];
        ns = b.n_seeds[rs];
        return true;
    }
    const uint32_t r = rs >> 1, strand = rs & 1u;
    ns = 0;
    // one round trip: the row's header and the two words that hold this lane's k-mer (their address does not depend on the length)
    const uint32_t m = (uint32_t)p.mer, w2 = b.pack_w2;
    const uint32_t i = (uint32_t)lane * (uint32_t)p.jump;
    const uint32_t* const row = b.pack + (size_t)r * b.pack_words;
    const bool inrow = i + m <= 16u * w2;
    const uint32_t o = inrow ? 2u * (16u * w2 - i - m) : 0u;
    const uint32_t* const form = row + (strand ? w2 + 2u : 1u);
    const uint32_t hdr = row[0], f0 = form[o >> 5], f1 = form[(o >> 5) + 1u];
    asm volatile("" :: "v"(f0), "v"(f1));             // issued here, beside the header, not after the branch on it
    const uint32_t L = hdr & 0xFFFFu;
    const bool on = !((hdr >> 17) & 1u) && (strand ? p.neg_strand : p.pos_strand);      // wave-uniform
    if (on) {
        const bool act = i + m < L;                                                      // lanes 0 .. ceil((L - mer) / jump) - 1
        const uint32_t nreg = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(act));
        bool bad = act && ((hdr >> 16) & 1u);                                            // a read with a non-ACGT base takes the serial walk
        uint32_t k = 0, cnt = 0;
        if (act && !bad) {
            const uint32_t code = (uint32_t)((((unsigned long long)f1 << 32) | f0) >> (o & 31u)) & (m >= 16u ? 0xFFFFFFFFu : ((1u << (2u * m)) - 1u));
            {                                                                            // the table probe of gm_seed_walk
                bool answered = false;
                if (p.kmer_ctab) {
                    const uint4 rec = p.kmer_ctab[code >> 3];
                    asm volatile("" :: "v"(rec.x), "v"(rec.y), "v"(rec.z), "v"(rec.w));      // ONE 16-byte load (the compiler would fetch rec.x in a second trip, where it is used)
                    // the record of gm_seed_walk, decoded per 32-bit half: the bytes below this code's (empty codes, >= 224, count 0) add up
                    // in two byte-sum instructions
                    const uint32_t sub = code & 7u, sh = (sub & 3u) << 3;
                    const uint32_t own = sub < 4u ? rec.y : rec.z;
                    cnt = (own >> sh) & 255u;
                    if (rec.w == 0u && cnt >= 224u) { bad = true; answered = true; }
                    else if (rec.w == 0u) {
                        const uint32_t part = own & ((1u << sh) - 1u);                  // bytes below the code's inside its own half
                        uint32_t lo = sub < 4u ? part : rec.y, hi = sub < 4u ? 0u : part;
                        lo &= ~(((lo & (lo << 1) & (lo << 2) & 0x80808080u) >> 7) * 0xFFu);
                        hi &= ~(((hi & (hi << 1) & (hi << 2) & 0x80808080u) >> 7) * 0xFFu);
                        k = rec.x + __builtin_amdgcn_sad_u8(lo, 0u, __builtin_amdgcn_sad_u8(hi, 0u, 0u));
                        answered = true;
                    }
                }
                if (!answered) {
                    const uint2 iv = p.kmer_tab[code];
                    if (iv.x == 0xFFFFFFFFu) bad = true;
                    else { k = iv.x; cnt = iv.y - iv.x + 1u; }
                }
                if (p.hcap > 0 && cnt > p.hcap) bad = true;
            }
        }
        if (__builtin_amdgcn_ballot_w64(bad) == 0ull) {
            ns = nreg;
            if (act) { sd.k = k; sd.l = k + cnt - 1u; sd.pos = i; }
        } else {                                                                          // rare: the wave walks again round by round; a non-ACGT base: the serial walk, by lane 0
            uint32_t nseed = 0;
            if ((hdr >> 16) & 1u) { if (lane == 0) nseed = gm_seed_walk_ool(gm_kargs(), rs, scratch, 1); }
            else nseed = gm_seed_rewalk_ool(gm_kargs(), rs, lane, scratch);
            // ONE wave runs this function (a one-wave workgroup, or wave 0 of k_vote_slots): the callee's stores to LDS have to
            // be complete before the wave's other lanes read them, no workgroup barrier
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            ns = (uint32_t)__builtin_amdgcn_readlane((int)nseed, 0);
            if ((uint32_t)lane < ns) sd = scratch[lane];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
    }
    // n_seeds / n_entries (saturating, like k_seed) of this read x strand
    const uint32_t c_all = (uint32_t)lane < ns ? sd.l - sd.k + 1u : 0u;
    uint32_t e_all;
    if (__builtin_amdgcn_ballot_w64(c_all > (1u << 24)) != 0ull) {                       // the 32-bit sum could overflow
        const unsigned long long e64 = gm_wave_sum((unsigned long long)c_all);
        e_all = e64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)e64;
        if (lane == 0 && e64 > 0xFFFFFFFFull) atomicAdd(&b.counters[GMK_SA_HITS], e64 - 0xFFFFFFFFull);     // k_heavy_collect adds n_entries
    } else {
        ie_all = gm_wave_scan_incl(c_all);                                               // the caller's scan of the hit counts, unless --fast cuts the seeds
        e_all = (uint32_t)__builtin_amdgcn_readlane((int)ie_all, 63);
    }
    if (lane == 0) { b.n_seeds[rs] = (uint16_t)ns; b.n_entries[rs] = e_all; }
    if (ns == 0) return false;
    if (e_all > p.heavy_min) {                                                            // sorted-key path: it reads the seed row
        if ((uint32_t)lane < ns) b.seeds[(size_t)rs * b.max_seeds + lane] = sd;
        return false;
    }
    return true;
}

#define GMS_SMAX 40                      // slots (64 lanes each) a read x strand may take in this kernel
#ifndef GMS_QS
#define GMS_QS 4                         // wave steps of the list walked per round of the second filter / the table
#endif
#define GMS_LCAP 540                     // compacted list entries per workgroup (more -> retry kernel); 488 / 440 (11.0 / 10.7 KB of LDS) measured: no change
// SMAX = slots a read x strand may take: 16 / 24 / 40 for few seeds or few hits per seed (every slot of the form is walked,
// used or not, so the small forms are the fast ones there), 64 = BIG, the form for 41..64 slots (e.g. 10-mers on a 150 Mbp
// reference: ~150 hits per seed): 32 slots per wave, a longer list, the second filter takes the whole zeroed region
// (4096 x 16 bit) and the table its own 4 KB.
template <bool MASK64, bool FULL, int SMAX, bool SEED>
__global__ void __launch_bounds__(128, SMAX == 64 ? 5 : SMAX == 16 ? 8 : 7) k_vote_slots(GmDevIndex ix, GmDevParams p, GmDevBatch b) {
    constexpr bool BIG = SMAX == 64;
    constexpr int LCAP = BIG ? 1280 : SMAX == 16 ? 320 : GMS_LCAP;      // 16-slot form: 10.0 KB of LDS, 16 workgroups per CU
    static_assert(SMAX == 16 || SMAX == 24 || SMAX == GMS_SMAX || SMAX == 64, "instantiated forms");
    constexpr int NT = 128, NW = 2, U = SMAX / NW, ZK = 512 / NT;
    static_assert(NW == 2, "the list is two stacks growing towards each other");
    __shared__ uint4 s_r0v[512];                     // 8 KB: counting filter (8192 x 8 bit), then keys | vals | low masks | high masks x 512
    __shared__ uint32_t s_lbp[LCAP];
    __shared__ uint8_t s_lt[LCAP];
    __shared__ uint2 s_desc[SMAX];               // {SA rank (flat entry index if !FULL) of lane 0, read offset | tag << 16 | (lanes - 1) << 24}
    __shared__ uint32_t s_cnt0[64];
    __shared__ uint32_t s_tab[BIG ? 1024 : 1];        // BIG: the exact table (256 x key | votes | low mask | high mask)
    __shared__ uint32_t s_nslots, s_E, s_nkeys, s_full, s_any0, s_lcnt[NW];
    uint32_t* const s_r0 = reinterpret_cast<uint32_t*>(s_r0v);
    const uint32_t rs = blockIdx.x;                  // grid = 2n
    const int tid = threadIdx.x, lane = gm_lane();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool prof = (p.dbg & 64) && (blockIdx.x & 63u) == 0 && tid == 0;      // sampled phase clocks (GM_DBG=64)
    long long tck = prof ? clock64() : 0;
    auto tick = [&](int slot) { if (prof) { long long now = clock64(); atomicAdd(&b.counters[GMK_DBG0 + slot], (unsigned long long)(now - tck)); tck = now; } };
    if (wave == 0) {                                 // seeds -> slot descriptors
        // SEED: the wave looks the seeds up itself (gm_tiny_seeds: lane j takes the k-mer at j * jump; no k_seed launch)
        GmSeed sd; uint32_t ns, ie_all;
        const bool go = gm_tiny_seeds<SEED>(ix, p, b, rs, lane, reinterpret_cast<GmSeed*>(s_lbp), sd, ns, ie_all);      // SEED = false: the seed row, not waiting for n_seeds: one round trip
        if (!go) ns = 0;                         // nothing to vote on, or left to the heavy path
        if (p.nw && p.fast && ns > 1) ns = 1;
        const uint32_t cnt = (uint32_t)lane < ns ? sd.l - sd.k + 1 : 0u;
        const uint32_t nsl = (cnt + 63u) >> 6;
        const uint32_t ie = gm_wave_scan_incl(cnt), is = gm_wave_scan_incl(nsl);
        uint32_t E0 = __builtin_amdgcn_readlane(ie, 63), S0 = __builtin_amdgcn_readlane(is, 63);
        if (S0 > SMAX) {                         // wave-uniform: hand over to the list kernel
            if (SEED && (uint32_t)lane < b.max_seeds) b.seeds[(size_t)rs * b.max_seeds + lane] = sd;
            if (lane == 0) { const uint32_t at = atomicAdd(b.n_big, 1u); b.big_list[at] = rs; }
            S0 = 0; E0 = 0;
        } else {
            const uint32_t s0 = is - nsl, e0 = ie - cnt;
            for (uint32_t j = 0; j < nsl; ++j) {
                const uint32_t left = cnt - 64u * j;
                s_desc[s0 + j] = make_uint2((FULL ? sd.k : e0) + 64u * j, sd.pos | ((uint32_t)lane << 16) | ((left < 64u ? left : 64u) << 24));
            }
        }
        if ((uint32_t)lane >= S0 && lane < SMAX) s_desc[lane] = make_uint2(0u, 0u);      // unused slots: no valid lane
        s_cnt0[lane] = 0;
        if (BIG) for (int q = lane; q < 1024; q += 64) s_tab[q] = 0;
        if (lane == 0) { s_nslots = S0; s_E = E0; s_nkeys = 0; s_full = 0; s_any0 = 0; }
    }
#pragma unroll
    for (int k = 0; k < ZK; ++k) s_r0v[tid + NT * k] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    tick(0);
    const uint32_t nslots = s_nslots, E = s_E;
    if (nslots == 0) return;                         // block-uniform: nothing to vote on, or handed over
    const uint32_t* const src = FULL ? ix.full_sa : b.coords + b.entry_off[rs];
    // ---- loads: slot s = j * NW + wave; everything about the slot is wave-uniform
    uint32_t bpv[U], meta[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        // no branches on the slot index below: an unused slot has 0 valid lanes (its loads hit rank 0.., harmless).  Guarding
        // the register-array updates with scalar branches makes the compiler treat the arrays as one wide vector and spill it.
        const uint2 d = s_desc[j * NW + wave];
        const uint32_t r0 = __builtin_amdgcn_readfirstlane(d.x);
        meta[j] = __builtin_amdgcn_readfirstlane(d.y);
        const uint32_t* const sp = src + r0;       // scalar base + 4 * lane
        bpv[j] = sp[lane];                         // lanes past the seed's last rank read the next ranks (buffers are padded); masked below
    }
    // ---- window starts; lanes without an entry become 0
    uint32_t nvalid = 0;                             // entries this wave holds (scalar)
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const uint32_t pos = meta[j] & 0xFFFFu, nl = meta[j] >> 24;
        nvalid += nl;
        bpv[j] = (uint32_t)lane < nl ? __builtin_elementwise_sub_sat(bpv[j], pos) : 0u;      // :267
    }
    tick(1);
    const bool filter = p.kmin >= 2 && E > 256;
    const uint32_t thr1 = (uint32_t)(p.kmin < 100 ? p.kmin : 100);      // filter bytes saturate (>= 128 -> retry kernel), so cap the test
    uint32_t wcount = 0;                             // wave-uniform fill of this wave's list segment
    uint32_t nnz = 0;                                // non-zero window starts of this wave (scalar)
    const int lorg = wave ? LCAP - 1 : 0, ldir = wave ? -1 : 1;       // the list is two stacks (see below)
    uint32_t lorg4 = (uint32_t)lorg * 4u, wbase = 0;                  // (lorg4: in a vector register, see pass 2a)
    asm volatile("" : "+v"(lorg4));
    const int ldir4 = ldir * 4;
    if (filter) {
        // ---- pass 1: one non-returning ds_add per hit into the counting filter
#pragma unroll
        for (int j = 0; j < U; ++j) {
            // slot = the low 13 bits of the window start as they are, used as the BYTE address of its counter (SA hits are
            // spread evenly; positions a multiple of 8192 apart share a slot, which only sends them on to the second filter):
            // the add needs the word and a shift, the test in pass 2a is one byte load.  No branch around the lanes without a
            // hit: they add 0 to a word of their own (s_cnt0[lane], which stays what it is).
            // (round 4: the kernel issues vector instructions 89 % of the time, so the lanes without a hit are masked off instead: one
            // compare - the one the popcount needs anyway - against a compare and two selects)
            asm volatile("" : "+v"(bpv[j]));            // (the compare is made HERE: folded into the subtraction above it leaves twenty lane masks in scalar registers)
            const uint32_t h = bpv[j];
            const bool nz = h != 0u;
            nnz += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(nz));
            asm volatile("" : "+s"(nnz));               // (counted here, not from twenty masks kept until the end of the loop)
            if (nz) atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(s_r0) + (h & 0x1FFCu)), 1u << ((h & 3u) << 3));
        }
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) nnz += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(bpv[j] != 0u));
    }
    // b = 0 (a hit closer to the reference start than the seed's read offset) is the one position that can take several
    // votes from one seed: counted per step.  It is rare, so it is only looked for when the number of non-zero window
    // starts (popcount of the compare pass 1 makes anyway) falls short of the number of entries.
    if (nnz != nvalid) {                             // wave-uniform, rare
        if (lane == 0) s_any0 = 1;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint32_t m = __builtin_amdgcn_readfirstlane(s_desc[j * NW + wave].y);
            const uint32_t nl = m >> 24, t = (m >> 16) & 63u;
            const unsigned long long z = __builtin_amdgcn_ballot_w64((uint32_t)lane < nl && bpv[j] == 0u);
            if (lane == 0 && z != 0ull) atomicAdd(&s_cnt0[t], (uint32_t)__popcll(z));
        }
    }
    if (filter) {
        __syncthreads();
        tick(2);
        // ---- pass 2a: hits whose slot counted >= kmin -> list (ballot + plain stores)
        // every lane reads its counter byte (lanes without a hit read byte 0 - a broadcast).  ALL the slots' bytes are requested before the
        // first is looked at: read and test slot by slot, the wave waits out one LDS round trip per slot (twenty in a row; the phase clocks
        // of GM_DBG=64 had this pass at 280 cycles per slot)
        uint32_t cnts[U];
#pragma unroll
        for (int j = 0; j < U; ++j) cnts[j] = reinterpret_cast<const unsigned char*>(s_r0)[bpv[j] & 8191u];
#pragma unroll
        for (int j = 0; j < U; ++j) asm volatile("" : "+v"(cnts[j]));
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint32_t cnt = cnts[j];
            const uint32_t c2 = bpv[j] != 0u ? cnt : 0u;                           // (one select + ONE compare that feeds the ballot and the branch)
            const bool pass = c2 >= thr1;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
            if (pass) {
                // wbase = min(wcount, LCAP - 64) (scalar), so the index needs no clamp; past the end the read x strand goes to the retry kernel (lfull)
                const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, wbase));
                const uint32_t li4 = (uint32_t)((int)lorg4 + ldir4 * (int)at);     // byte address of the entry (one multiply-add: lorg4 lives in a vector register)
                *reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(s_lbp) + li4) = bpv[j];
                s_lt[li4 >> 2] = (uint8_t)(j * NW + wave);                         // slot id; its seed's step is looked up later, for the few that survive
            }
            wcount += (uint32_t)__popcll(m);
            wbase = wcount < (uint32_t)(LCAP - 64) ? wcount : (uint32_t)(LCAP - 64);
        }
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const bool pass = bpv[j] != 0;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
            if (pass) {
                const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, wcount));
                if (at < (uint32_t)LCAP) { const uint32_t li = (uint32_t)(lorg + ldir * (int)at); s_lbp[li] = bpv[j]; s_lt[li] = (uint8_t)(j * NW + wave); }
            }
            wcount += (uint32_t)__popcll(m);
        }
    }
    // the list is one array used as two stacks: wave 0 grows up from 0, wave 1 down from the top; they only meet when the
    // two together exceed LCAP, and then the read x strand goes to the retry kernel anyway
    if (lane == 0) s_lcnt[wave] = wcount;
    __syncthreads();
    tick(3);
    if (filter) {
        // ---- the filter is dead: check it for saturation (a byte >= 128) and zero it - that is the empty table
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < ZK; ++k) {
            const uint4 v = s_r0v[tid + NT * k];
            acc |= v.x | v.y | v.z | v.w;
            s_r0v[tid + NT * k] = make_uint4(0u, 0u, 0u, 0u);
        }
        if (acc & 0x80808080u) s_full = 1;
        __syncthreads();
        tick(4);
    }
    // ---- pass 2b: the list is mostly single positions that shared a filter slot with another one; a linear-probing table
    // loaded with all of them makes every wave step wait for its unluckiest lane (tens of dependent CAS round trips).  So
    // the list goes through a second filter first (other hash, two bits per slot: seen / seen again), and only entries whose
    // second slot was hit again enter the exact table, which then stays nearly empty.
    // Both live in the zeroed filter memory: words [0,1024) = the second filter (16 384 slots of two bits), words [1024,2048) = 256 slots of
    // key | votes | low step mask | high step mask (key 0 = empty; b = 0 never gets here).
    constexpr int T2 = 256;
    constexpr uint32_t F2W = BIG ? 2047u : 1023u;     // second filter: words of sixteen 2-bit slots
    constexpr int F2S = BIG ? 11 : 10;
    uint32_t* const keys = BIG ? s_tab : s_r0 + 1024; uint32_t* const vals = keys + T2; uint32_t* const mlo = keys + 2 * T2; uint32_t* const mhi = keys + 3 * T2;
    const bool lfull = s_lcnt[0] + s_lcnt[1] > (uint32_t)LCAP;       // block-uniform
    const uint32_t n_l = lfull ? 0u : s_lcnt[wave];
    const uint32_t thr = (uint32_t)(p.kmin < 1 ? 1 : p.kmin);
    for (uint32_t i0 = 0; i0 < n_l; i0 += 64 * GMS_QS) {     // GMS_QS wave steps at a time: their list reads are in flight together
        uint32_t bp4[GMS_QS];
#pragma unroll
        for (int q = 0; q < GMS_QS; ++q) {
            const uint32_t i = i0 + 64u * q + (uint32_t)lane;
            bp4[q] = i < n_l ? s_lbp[wave ? LCAP - 1 - i : i] : 0u;
        }
#pragma unroll
        for (int q = 0; q < GMS_QS; ++q)
            if (bp4[q] != 0u) {
                // two bits per slot, seen / seen again (as in k_vote_tiny): 16 384 (BIG: 32 768) slots in the words that held 2048
                // (4096) 16-bit counters - 8 x fewer entries reach the CAS loop of the table by sharing a slot
                const uint32_t h2 = (bp4[q] * 0x85EBCA6Bu) >> (BIG ? 17 : 18), sh = (h2 >> F2S) << 1;
                const uint32_t old = atomicOr(&s_r0[h2 & F2W], 1u << sh);
                if ((old >> sh) & 1u) atomicOr(&s_r0[h2 & F2W], 2u << sh);
            }
    }
    __syncthreads();
    tick(5);
    {
        bool full = false;
        uint32_t nfresh = 0;
        for (uint32_t i0 = 0; i0 < n_l; i0 += 64 * GMS_QS) {
            uint32_t bp4[GMS_QS], c4[GMS_QS];
#pragma unroll
            for (int q = 0; q < GMS_QS; ++q) {
                const uint32_t i = i0 + 64u * q + (uint32_t)lane;
                bp4[q] = i < n_l ? s_lbp[wave ? LCAP - 1 - i : i] : 0u;
            }
#pragma unroll
            for (int q = 0; q < GMS_QS; ++q) {
                const uint32_t h2 = (bp4[q] * 0x85EBCA6Bu) >> (BIG ? 17 : 18);
                c4[q] = (s_r0[h2 & F2W] >> ((h2 >> F2S) << 1)) & (thr >= 2u ? 2u : 1u);
            }
#pragma unroll
            for (int q = 0; q < GMS_QS; ++q) {
                bool fresh = false;
                if (bp4[q] != 0u && c4[q] != 0u) {
                    const uint32_t i = i0 + 64u * q + (uint32_t)lane;
                    const uint32_t bp = bp4[q], t = (s_desc[s_lt[wave ? LCAP - 1 - i : i]].y >> 16) & 63u;
                    uint32_t slot = (bp * 0x9E3779B1u) >> 24;
                    uint32_t old;
                    int probes = 0;
                    // one exit test per probe (the table is kept under 3/4 full, so an empty slot always turns up)
                    while ((old = atomicCAS(&keys[slot], 0u, bp)) != 0u && old != bp && ++probes < T2) slot = (slot + 1) & (T2 - 1);
                    fresh = old == 0u;
                    const bool found = old == 0u || old == bp;
                    if (!found) full = true;
                    else {
                        atomicAdd(&vals[slot], 1u);
                        if (t < 32) atomicOr(&mlo[slot], 1u << t);
                        else if (MASK64) atomicOr(&mhi[slot], 1u << (t - 32));
                    }
                }
                nfresh += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fresh));
            }
        }
        if (lane == 0 && nfresh) atomicAdd(&s_nkeys, nfresh);
        if (full) s_full = 1;
    }
    __syncthreads();
    tick(6);
    const bool failed = lfull || s_full || s_nkeys > (uint32_t)(T2 * 3 / 4);
    if (failed) {                                    // hand this read x strand to the global-table kernel
        if (tid == 0) {
            if (SEED) (void)gm_seed_walk_ool(gm_kargs(), rs, nullptr, 0);      // the retry kernel reads the seed row
            b.rs_overflow[rs] = 1;
            const uint32_t j = atomicAdd(b.n_retry, 1u);
            const uint32_t need = 2 * E; uint32_t sz = 1024; while (sz < need && sz < 0x80000000u) sz <<= 1;
            const unsigned long long off = atomicAdd(&b.counters[GMK_HEAVY_SLOTS], (unsigned long long)sz);
            b.retry_list[j] = rs;
            b.retry_off[j] = off;
            atomicAdd(&b.counters[GMK_OVERFLOW_RS], 1ull);
        }
        return;
    }
    // ---- emit: NW step = kmin-th lowest step that voted.  Wave 0 looks at all four table slots per lane and leaves the candidates
    // in the read x strand's OWN slots (plain stores, no bump counter to wait for; k_cand_gather moves them into the shards and keeps
    // the candidates of neighbouring reads together for the DP kernel); more than GM_FIXED_C of them: one returning atomic as before
    if (tid < 64) {
        static_assert(T2 == 256, "four table slots per lane of wave 0");
        bool em[4]; uint32_t ky[4], st[4], nb[4];
        unsigned long long mk[4];
        uint32_t total = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t slot = (uint32_t)(q * 64 + lane);
            const uint32_t key = keys[slot], v = vals[slot];
            em[q] = key != 0u && v >= (uint32_t)p.kmin;
            ky[q] = key; st[q] = 0;
            if (em[q]) {
                if (p.nw) {
                    unsigned long long m = (unsigned long long)mlo[slot] | (MASK64 ? ((unsigned long long)mhi[slot] << 32) : 0ull);
                    for (int r = 1; r < p.kmin; ++r) m &= m - 1;          // uniform trip count; 0 stays 0
                    st[q] = m ? (uint32_t)(__ffsll((long long)m) - 1) : 0u;
                } else st[q] = v > 65535u ? 65535u : v;
            }
            mk[q] = __builtin_amdgcn_ballot_w64(em[q]);
            nb[q] = total;
            total += (uint32_t)__popcll(mk[q]);
        }
        if (total != 0u && b.fixed_cands && total <= GM_FIXED_C) {            // wave-uniform
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (em[q]) {
                    const uint32_t idx = nb[q] + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk[q] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk[q], 0u));
                    GmCand c;
                    c.rs = rs; c.b = ky[q]; c.step = (uint16_t)st[q]; c.flags = 4; c.pad = 0; c.score = 0.0f;
                    b.fixed_cands[GM_FIXED_AT(b, rs, idx)] = c;
                }
            if (lane == 0) b.fixed_cnt[rs] = (uint8_t)total;
        } else if (total != 0u) {                    // wave-uniform
            const uint32_t shard = blockIdx.x & (GM_NSHARD - 1);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&b.shard_cnt[shard * GM_SHARD_STRIDE], total);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (em[q]) {
                    const uint32_t idx = base + nb[q] + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk[q] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk[q], 0u));
                    if (idx < b.cand_region) {
                        GmCand c;
                        c.rs = rs; c.b = ky[q]; c.step = (uint16_t)st[q]; c.flags = 4; c.pad = 0; c.score = 0.0f;
                        b.cands[(size_t)shard * b.cand_region + idx] = c;
                    }
                }
        }
    }
    if (tid < 64 && s_any0) {                        // b = 0 (only if some wave saw such a vote): cumulative per-step counts
        const uint32_t run = gm_wave_scan_incl(s_cnt0[tid]);
        const uint32_t total = __builtin_amdgcn_readlane(run, 63);
        const unsigned long long reached = __builtin_amdgcn_ballot_w64(run >= (uint32_t)p.kmin);
        const bool emit = tid == 0 && total >= (uint32_t)p.kmin;
        const uint32_t step = p.nw ? (uint32_t)(__ffsll((long long)reached) - 1) : (total > 65535u ? 65535u : total);
        gm_emit<GmLdsTable>(b, emit, rs, 0u, step, 4);
    }
    tick(7);
    if (prof) atomicAdd(&b.counters[GMK_DBG8], 1ull);
}

// ---- k_vote_slots_pp: k_vote_slots (full SA, seeds looked up in the kernel) with the loads of the read x strands to come in flight -------
// k_vote_slots spends a third of a workgroup's life in three dependent round trips before its first vote (the 2-bit row, the k-mer
// table record, the SA ranks: profiles/r04_slots_phase_clocks_100Mbp.txt), and its LDS leaves no room for more workgroups to wait them
// out.  Here a workgroup takes `chunk` consecutive read x strands and keeps three of them on the way at once: while it votes on read x
// strand c, the SA ranks of c + 1 are landing in a second register set, the table record of c + 2 and the 2-bit words of c + 3 are in
// flight.  Every load is issued one iteration before its use and is unconditional (an address clamped onto the chunk's last read x
// strand instead of a branch around the load: a branch makes the compiler merge registers at the join, i.e. wait at once); both waves
// issue the (identical) seed loads so that no wave-dependent branch surrounds them, wave 0 alone turns records into slot descriptors.
// The slot descriptors are double-buffered; the filter / table memory is zeroed at the END of a vote (wave 1 its first half while wave
// 0 emits from the second), so an iteration starts with one barrier.  Votes, hand-overs and results are those of k_vote_slots.
// the 64-slot form of the pipelined kernel keeps its exact table in the upper half of the filter memory, as the smaller forms do (0; 1: in 4 KB
// of its own, as k_vote_slots<64>): 15.9 instead of 20 KB of LDS = 10 workgroups per CU where the registers allow them (seed rows from
// k_seed: 91 registers, vote 94.9 -> 90.0 ms at the chrX shape with -h 150; seeds looked up in the kernel: 96 + 11 spilled at 5 waves per
// SIMD, 112.1 ms either way)
#ifndef GMS_BIG_TAB
#define GMS_BIG_TAB 0
#endif
#ifndef GMS_PP_WAVES64
#define GMS_PP_WAVES64 5
#endif
#ifndef GMS_PP_WAVES
#define GMS_PP_WAVES 5                   // wavefronts per SIMD the pipelined form is compiled for (its 40-slot form: 96 registers)
#endif
template <bool MASK64, int SMAX, bool SEED>
__global__ void __launch_bounds__(128, SMAX == 64 ? GMS_PP_WAVES64 : GMS_PP_WAVES) k_vote_slots_pp(GmDevIndex ix, GmDevParams p, GmDevBatch b, const uint32_t chunk) {
    constexpr bool BIG = SMAX == 64;
    constexpr bool BIGT = BIG && GMS_BIG_TAB != 0;      // the 64-slot form's exact table in 4 KB of its own (second filter: 32 768 slots) or, as in the smaller forms, in the upper half of the filter memory
    constexpr int LCAP = BIG ? 1280 : SMAX == 16 ? 320 : GMS_LCAP;
    static_assert(SMAX == 16 || SMAX == 24 || SMAX == GMS_SMAX || SMAX == 64, "instantiated forms");
    constexpr int NT = 128, NW = 2, U = SMAX / NW, ZK = 512 / NT;
    __shared__ uint4 s_r0v[512];                     // as in k_vote_slots
    __shared__ uint32_t s_lbp[LCAP];
    __shared__ uint8_t s_lt[LCAP];
    __shared__ uint2 s_desc[2][SMAX];                // [read x strand parity within the chunk]
    __shared__ uint32_t s_cnt0[64];
    __shared__ uint4 s_tabv[BIGT ? 256 : 1];          // BIG: the exact table (256 x key | votes | low mask | high mask)
    __shared__ uint32_t s_nslots[2], s_E[2], s_nkeys, s_full, s_any0, s_lcnt[NW];
    uint32_t* const s_r0 = reinterpret_cast<uint32_t*>(s_r0v);
    uint32_t* const s_tab = reinterpret_cast<uint32_t*>(s_tabv);
    const int tid = threadIdx.x, lane0 = gm_lane();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n2 = 2u * b.n, rs0 = blockIdx.x * chunk;
    if (rs0 >= n2) return;
    const int n_it = (int)(n2 - rs0 < chunk ? n2 - rs0 : chunk);
    const uint32_t rs_last = rs0 + (uint32_t)n_it - 1u;
    // the seed lookup of gm_tiny_seeds<true>, cut into its three round trips; what does not depend on the read x strand:
    const uint32_t m = (uint32_t)p.mer, w2 = b.pack_w2;
    const uint32_t cmask = m >= 16u ? 0xFFFFFFFFu : ((1u << (2u * m)) - 1u);
#pragma unroll
    for (int k = 0; k < ZK; ++k) s_r0v[tid + NT * k] = make_uint4(0u, 0u, 0u, 0u);
    if (BIGT) { s_tabv[tid] = make_uint4(0u, 0u, 0u, 0u); s_tabv[tid + NT] = make_uint4(0u, 0u, 0u, 0u); }
    uint32_t vz = 0;                                 // zero, opaque: the header is loaded per lane and stays in a vector register until stage A makes
    asm volatile("" : "+v"(vz));                     // it scalar (loaded from a uniform address it is made scalar at once, i.e. waited for at once)
    uint32_t F_hdr = 0, F_f0 = 0, F_f1 = 0;          // c + 3: header and the two words that hold this lane's k-mer
    uint32_t P_hdr = 0, P_code = 0;                  // c + 2: its header, this lane's code and the table record of the code
    uint4 P_rec = make_uint4(0u, 0u, 0u, 0u);
    GmSeed R_sd; R_sd.k = 0; R_sd.l = 0; R_sd.pos = 0;      // SEED = false (seed rows of k_seed): the lane's seed of c + 2 and the row's length
    uint32_t R_ns = 0;
    uint32_t bpv[U];                                 // SA ranks of c + 1 on their way (from the point where c's have gone into the list) -> window starts of c
#pragma unroll
    for (int j = 0; j < U; ++j) bpv[j] = 0;
    for (int it = -3; it < n_it; ++it) {
        const uint32_t cb = (uint32_t)it & 1u, nb = cb ^ 1u;
        // the wave number as a value the compiler cannot see through: what is derived from it (the 20 slot numbers j * NW + wave of the list
        // entries) is then computed where it is used instead of being kept in 20 registers across the loop
        uint32_t wv = (uint32_t)wave, lv = (uint32_t)lane0;
        asm volatile("" : "+v"(wv), "+v"(lv));
        const int lane = (int)lv;                        // (likewise: the lane's seed offset, shift and word index are three instructions, not three registers)
        const uint32_t si = lv * (uint32_t)p.jump;
        const uint32_t o = si + m <= 16u * w2 ? 2u * (16u * w2 - si - m) : 0u;
        if (wave == 0) {
            // ---- c + 1: record -> seeds -> slot descriptors (gm_tiny_seeds<true> from its probe on, then wave 0's part of k_vote_slots)
            const uint32_t rs = rs0 + (uint32_t)(it + 1);
            uint32_t S0 = 0, E0 = 0;
            if (it + 1 >= 0 && it + 1 < n_it) {          // wave-uniform
                GmSeed sd; sd.k = 0; sd.l = 0; sd.pos = 0;
                uint32_t ns = 0, ie_all = 0xFFFFFFFFu;
                if (!SEED) {                             // the seed row has arrived: nothing to look up, n_seeds / n_entries are k_seed's
                    ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)R_ns);
                    if ((uint32_t)lane < b.max_seeds) sd = R_sd;
                }
                const uint32_t hdr = (uint32_t)__builtin_amdgcn_readfirstlane((int)P_hdr), strand = rs & 1u;
                const uint32_t L = hdr & 0xFFFFu;
                const bool on = SEED && !((hdr >> 17) & 1u) && (strand ? p.neg_strand : p.pos_strand);      // wave-uniform
                if (on) {
                    const bool act = si + m < L;
                    const uint32_t nreg = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(act));
                    bool bad = act && ((hdr >> 16) & 1u);
                    uint32_t k = 0, cnt = 0;
                    if (act && !bad) {
                        const uint32_t code = P_code;
                        const uint4 rec = P_rec;
                        bool answered = false;
                        const uint32_t sub = code & 7u, sh = (sub & 3u) << 3;
                        const uint32_t own = sub < 4u ? rec.y : rec.z;
                        cnt = (own >> sh) & 255u;
                        if (rec.w == 0u && cnt >= 224u) { bad = true; answered = true; }
                        else if (rec.w == 0u) {
                            const uint32_t part = own & ((1u << sh) - 1u);
                            uint32_t lo = sub < 4u ? part : rec.y, hi = sub < 4u ? 0u : part;
                            lo &= ~(((lo & (lo << 1) & (lo << 2) & 0x80808080u) >> 7) * 0xFFu);
                            hi &= ~(((hi & (hi << 1) & (hi << 2) & 0x80808080u) >> 7) * 0xFFu);
                            k = rec.x + __builtin_amdgcn_sad_u8(lo, 0u, __builtin_amdgcn_sad_u8(hi, 0u, 0u));
                            answered = true;
                        }
                        if (!answered) {                 // an escaped record (a count that does not fit a byte): the full table, now
                            const uint2 iv = p.kmer_tab[code];
                            if (iv.x == 0xFFFFFFFFu) bad = true;
                            else { k = iv.x; cnt = iv.y - iv.x + 1u; }
                        }
                        if (p.hcap > 0 && cnt > p.hcap) bad = true;
                    }
                    if (__builtin_amdgcn_ballot_w64(bad) == 0ull) {
                        ns = nreg;
                        if (act) { sd.k = k; sd.l = k + cnt - 1u; sd.pos = si; }
                    } else {                             // rare: walked again round by round; a non-ACGT base: the serial walk, by lane 0
                        GmSeed* const scratch = reinterpret_cast<GmSeed*>(s_lbp);      // (the list of the vote before is dead, this one's not begun)
                        uint32_t nseed = 0;
                        if ((hdr >> 16) & 1u) { if (lane == 0) nseed = gm_seed_walk_ool(gm_kargs(), rs, scratch, 1); }
                        else nseed = gm_seed_rewalk_ool(gm_kargs(), rs, lane, scratch);
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                        ns = (uint32_t)__builtin_amdgcn_readlane((int)nseed, 0);
                        if ((uint32_t)lane < ns) sd = scratch[lane];
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    }
                }
                if (SEED) {
                const uint32_t c_all = (uint32_t)lane < ns ? sd.l - sd.k + 1u : 0u;
                uint32_t e_all;
                if (__builtin_amdgcn_ballot_w64(c_all > (1u << 24)) != 0ull) {
                    const unsigned long long e64 = gm_wave_sum((unsigned long long)c_all);
                    e_all = e64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)e64;
                    if (lane == 0 && e64 > 0xFFFFFFFFull) atomicAdd(&b.counters[GMK_SA_HITS], e64 - 0xFFFFFFFFull);
                } else {
                    ie_all = gm_wave_scan_incl(c_all);
                    e_all = (uint32_t)__builtin_amdgcn_readlane((int)ie_all, 63);
                }
                if (lane == 0) { b.n_seeds[rs] = (uint16_t)ns; b.n_entries[rs] = e_all; }
                if (ns != 0 && e_all > p.heavy_min) {    // sorted-key path: it reads the seed row
                    if ((uint32_t)lane < ns) b.seeds[(size_t)rs * b.max_seeds + lane] = sd;
                    ns = 0;
                }
                }
                if (p.nw && p.fast && ns > 1) ns = 1;
                const uint32_t cnt = (uint32_t)lane < ns ? sd.l - sd.k + 1 : 0u;
                const uint32_t nsl = (cnt + 63u) >> 6;
                const uint32_t ie = gm_wave_scan_incl(cnt), is = gm_wave_scan_incl(nsl);
                E0 = __builtin_amdgcn_readlane(ie, 63); S0 = __builtin_amdgcn_readlane(is, 63);
                if (S0 > SMAX) {                         // wave-uniform: hand over to the list kernel
                    if (SEED && (uint32_t)lane < b.max_seeds) b.seeds[(size_t)rs * b.max_seeds + lane] = sd;
                    if (lane == 0) { const uint32_t at = atomicAdd(b.n_big, 1u); b.big_list[at] = rs; }
                    S0 = 0; E0 = 0;
                } else {
                    const uint32_t s0 = is - nsl;
                    for (uint32_t j = 0; j < nsl; ++j) {
                        const uint32_t left = cnt - 64u * j;
                        s_desc[nb][s0 + j] = make_uint2(sd.k + 64u * j, sd.pos | ((uint32_t)lane << 16) | ((left < 64u ? left : 64u) << 24));
                    }
                }
            }
            if ((uint32_t)lane >= S0 && lane < SMAX) s_desc[nb][lane] = make_uint2(0u, 0u);      // unused slots: no valid lane
            s_cnt0[lane] = 0;                             // (the state of THIS iteration's vote)
            if (lane == 0) { s_nslots[nb] = S0; s_E[nb] = E0; s_nkeys = 0; s_full = 0; s_any0 = 0; }
        }
        if (!SEED) {   // ---- c + 2: the lane's seed of the row k_seed wrote, and the row's length (as a lane value: see vz)
            const uint32_t lo = rs0 + (uint32_t)(it + 2 > 0 ? it + 2 : 0);
            const uint32_t rs = lo < rs_last ? lo : rs_last;
            R_sd = b.seeds[(size_t)rs * b.max_seeds + ((uint32_t)lane < b.max_seeds ? (uint32_t)lane : 0u)];
            R_ns = b.n_seeds[(size_t)rs + vz];
        }
        if (SEED) {   // ---- c + 2: this lane's code from the words that have arrived -> its table record (one 16-byte load)
            const uint32_t code = (uint32_t)((((unsigned long long)F_f1 << 32) | F_f0) >> (o & 31u)) & cmask;
            P_hdr = F_hdr; P_code = code;
            P_rec = p.kmer_ctab[code >> 3];          // (no use here: it is wanted one iteration on)
        }
        if (SEED) {   // ---- c + 3: the row's header and the two words that hold this lane's k-mer
            const uint32_t rs = rs0 + (uint32_t)(it + 3) < rs_last ? rs0 + (uint32_t)(it + 3) : rs_last;
            const uint32_t* const row = b.pack + (size_t)(rs >> 1) * b.pack_words;
            const uint32_t* const form = row + ((rs & 1u) ? w2 + 2u : 1u);
            F_hdr = row[vz]; F_f0 = form[o >> 5]; F_f1 = form[(o >> 5) + 1u];
        }
        gm_lds_barrier();
        // ---- c: the vote of k_vote_slots
        const uint32_t rs = rs0 + (uint32_t)it;
        const uint32_t nslots = it >= 0 ? s_nslots[cb] : 0u, E = s_E[cb];
        const bool voting = nslots != 0;                 // block-uniform; false: nothing to vote on, or handed over (the filter memory is still zero)
        uint32_t nvalid = 0;
        uint32_t wcount = 0, nnz = 0;
        const bool filter = p.kmin >= 2 && E > 256;
        const uint32_t thr1 = (uint32_t)(p.kmin < 100 ? p.kmin : 100);
        const int lorg = wave ? LCAP - 1 : 0, ldir = wave ? -1 : 1;
        uint32_t lorg4 = (uint32_t)lorg * 4u + (lv & 0u), wbase = 0;      // (per lane on purpose: a vector register for the list address' multiply-add)
        const int ldir4 = ldir * 4;
        if (voting) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint32_t mt = __builtin_amdgcn_readfirstlane(s_desc[cb][j * NW + wave].y);
            const uint32_t pos = mt & 0xFFFFu, nl = mt >> 24;
            nvalid += nl;
            bpv[j] = (uint32_t)lane < nl ? __builtin_elementwise_sub_sat(bpv[j], pos) : 0u;      // :267
        }
        if (filter) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                asm volatile("" : "+v"(bpv[j]));
                const uint32_t h = bpv[j];
                const bool nz = h != 0u;
                nnz += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(nz));
                asm volatile("" : "+s"(nnz));
                if (nz) atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(s_r0) + (h & 0x1FFCu)), 1u << ((h & 3u) << 3));
            }
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) nnz += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(bpv[j] != 0u));
        }
        if (nnz != nvalid) {                             // wave-uniform, rare: votes for b = 0, counted per step
            if (lane == 0) s_any0 = 1;
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const uint32_t mt = __builtin_amdgcn_readfirstlane(s_desc[cb][j * NW + wave].y);
                const uint32_t nl = mt >> 24, t = (mt >> 16) & 63u;
                const unsigned long long z = __builtin_amdgcn_ballot_w64((uint32_t)lane < nl && bpv[j] == 0u);
                if (lane == 0 && z != 0ull) atomicAdd(&s_cnt0[t], (uint32_t)__popcll(z));
            }
        }
        if (filter) {
            gm_lds_barrier();
            // (the counter bytes of half the slots requested before the first is looked at: see k_vote_slots)
            constexpr int HB = U / 2;
            static_assert(U % 2 == 0, "two batches of filter bytes");
#pragma unroll
            for (int j0 = 0; j0 < U; j0 += HB) {
            uint32_t cnts[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) cnts[j] = reinterpret_cast<const unsigned char*>(s_r0)[bpv[j0 + j] & 8191u];
#pragma unroll
            for (int j = 0; j < HB; ++j) asm volatile("" : "+v"(cnts[j]));
#pragma unroll
            for (int jj = 0; jj < HB; ++jj) {
                const int j = j0 + jj;
                const uint32_t cnt = cnts[jj];
                const uint32_t c2 = bpv[j] != 0u ? cnt : 0u;
                const bool pass = c2 >= thr1;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(pass);
                if (pass) {
                    const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, wbase));
                    const uint32_t li4 = (uint32_t)((int)lorg4 + ldir4 * (int)at);
                    *reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(s_lbp) + li4) = bpv[j];
                    s_lt[li4 >> 2] = (uint8_t)((uint32_t)(j * NW) + wv);
                }
                wcount += (uint32_t)__popcll(mk);
                wbase = wcount < (uint32_t)(LCAP - 64) ? wcount : (uint32_t)(LCAP - 64);
            }
            }
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const bool pass = bpv[j] != 0;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(pass);
                if (pass) {
                    const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, wcount));
                    if (at < (uint32_t)LCAP) { const uint32_t li = (uint32_t)(lorg + ldir * (int)at); s_lbp[li] = bpv[j]; s_lt[li] = (uint8_t)((uint32_t)(j * NW) + wv); }
                }
                wcount += (uint32_t)__popcll(mk);
            }
        }
        if (lane == 0) s_lcnt[wave] = wcount;
        }
        // ---- c + 1: its SA ranks into the registers whose window starts have just gone into the list: slot s = j * NW + wave (scalar base
        // + 4 * lane; an unused slot reads rank 0.., harmless); they land during the table phases and the emit
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint32_t r0 = __builtin_amdgcn_readfirstlane(s_desc[nb][j * NW + wave].x);
            bpv[j] = (ix.full_sa + r0)[lane];
        }
        if (!voting) continue;
        gm_lds_barrier();
        if (filter) {                                    // the filter is dead: check it for saturation and zero it - that is the empty table
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < ZK; ++k) {
                const uint4 v = s_r0v[tid + NT * k];
                acc |= v.x | v.y | v.z | v.w;
                s_r0v[tid + NT * k] = make_uint4(0u, 0u, 0u, 0u);
            }
            if (acc & 0x80808080u) s_full = 1;
            gm_lds_barrier();
        }
        constexpr int T2 = 256;
        constexpr uint32_t F2W = BIGT ? 2047u : 1023u;
        constexpr int F2S = BIGT ? 11 : 10;
        uint32_t* const keys = BIGT ? s_tab : s_r0 + 1024; uint32_t* const vals = keys + T2; uint32_t* const mlo = keys + 2 * T2; uint32_t* const mhi = keys + 3 * T2;
        const bool lfull = s_lcnt[0] + s_lcnt[1] > (uint32_t)LCAP;       // block-uniform
        const uint32_t n_l = lfull ? 0u : s_lcnt[wave];
        const uint32_t thr = (uint32_t)(p.kmin < 1 ? 1 : p.kmin);
        for (uint32_t i0 = 0; i0 < n_l; i0 += 64 * GMS_QS) {     // GMS_QS wave steps at a time: their list reads are in flight together
            uint32_t bp4[GMS_QS];
#pragma unroll
            for (int q = 0; q < GMS_QS; ++q) {
                const uint32_t i = i0 + 64u * q + (uint32_t)lane;
                bp4[q] = i < n_l ? s_lbp[wave ? LCAP - 1 - i : i] : 0u;
            }
#pragma unroll
            for (int q = 0; q < GMS_QS; ++q)
                if (bp4[q] != 0u) {
                    // two bits per slot, seen / seen again (as in k_vote_tiny): 16 384 (BIG: 32 768) slots in the words that held 2048
                    // (4096) 16-bit counters - 8 x fewer entries reach the CAS loop of the table by sharing a slot
                    const uint32_t h2 = (bp4[q] * 0x85EBCA6Bu) >> (BIGT ? 17 : 18), sh = (h2 >> F2S) << 1;
                    const uint32_t old = atomicOr(&s_r0[h2 & F2W], 1u << sh);
                    if ((old >> sh) & 1u) atomicOr(&s_r0[h2 & F2W], 2u << sh);
                }
        }
        gm_lds_barrier();
        {
            bool full = false;
            uint32_t nfresh = 0;
            for (uint32_t i0 = 0; i0 < n_l; i0 += 64 * GMS_QS) {
                uint32_t bp4[GMS_QS], c4[GMS_QS];
#pragma unroll
                for (int q = 0; q < GMS_QS; ++q) {
                    const uint32_t i = i0 + 64u * q + (uint32_t)lane;
                    bp4[q] = i < n_l ? s_lbp[wave ? LCAP - 1 - i : i] : 0u;
                }
#pragma unroll
                for (int q = 0; q < GMS_QS; ++q) {
                    const uint32_t h2 = (bp4[q] * 0x85EBCA6Bu) >> (BIGT ? 17 : 18);
                    c4[q] = (s_r0[h2 & F2W] >> ((h2 >> F2S) << 1)) & (thr >= 2u ? 2u : 1u);
                }
#pragma unroll
                for (int q = 0; q < GMS_QS; ++q) {
                    bool fresh = false;
                    if (bp4[q] != 0u && c4[q] != 0u) {
                        const uint32_t i = i0 + 64u * q + (uint32_t)lane;
                        const uint32_t bp = bp4[q], t = (s_desc[cb][s_lt[wave ? LCAP - 1 - i : i]].y >> 16) & 63u;
                        uint32_t slot = (bp * 0x9E3779B1u) >> 24;
                        uint32_t old;
                        int probes = 0;
                        // one exit test per probe (the table is kept under 3/4 full, so an empty slot always turns up)
                        while ((old = atomicCAS(&keys[slot], 0u, bp)) != 0u && old != bp && ++probes < T2) slot = (slot + 1) & (T2 - 1);
                        fresh = old == 0u;
                        const bool found = old == 0u || old == bp;
                        if (!found) full = true;
                        else {
                            atomicAdd(&vals[slot], 1u);
                            if (t < 32) atomicOr(&mlo[slot], 1u << t);
                            else if (MASK64) atomicOr(&mhi[slot], 1u << (t - 32));
                        }
                    }
                    nfresh += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fresh));
                }
            }
            if (lane == 0 && nfresh) atomicAdd(&s_nkeys, nfresh);
            if (full) s_full = 1;
        }
        gm_lds_barrier();
        const bool failed = lfull || s_full || s_nkeys > (uint32_t)(T2 * 3 / 4);      // block-uniform
        if (wave != 0) {                                 // wave 1 is done with this vote: it zeroes what wave 0 does not emit from
#pragma unroll
            for (int k = 0; k < (BIGT ? 8 : 4); ++k) s_r0v[lane + 64 * k] = make_uint4(0u, 0u, 0u, 0u);
            continue;
        }
        if (failed) {                                    // hand this read x strand to the global-table kernel
            if (lane == 0) {
                if (SEED) (void)gm_seed_walk_ool(gm_kargs(), rs, nullptr, 0);      // the retry kernel reads the seed row
                b.rs_overflow[rs] = 1;
                const uint32_t j = atomicAdd(b.n_retry, 1u);
                const uint32_t need = 2 * E; uint32_t sz = 1024; while (sz < need && sz < 0x80000000u) sz <<= 1;
                const unsigned long long off = atomicAdd(&b.counters[GMK_HEAVY_SLOTS], (unsigned long long)sz);
                b.retry_list[j] = rs;
                b.retry_off[j] = off;
                atomicAdd(&b.counters[GMK_OVERFLOW_RS], 1ull);
            }
        } else {
            // ---- emit (wave 0): as k_vote_slots
            static_assert(T2 == 256, "four table slots per lane of wave 0");
            bool em[4]; uint32_t ky[4], st[4], nbs[4];
            unsigned long long mk[4];
            uint32_t total = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t slot = (uint32_t)(q * 64 + lane);
                const uint32_t key = keys[slot], v = vals[slot];
                em[q] = key != 0u && v >= (uint32_t)p.kmin;
                ky[q] = key; st[q] = 0;
                if (em[q]) {
                    if (p.nw) {
                        unsigned long long mm = (unsigned long long)mlo[slot] | (MASK64 ? ((unsigned long long)mhi[slot] << 32) : 0ull);
                        for (int r = 1; r < p.kmin; ++r) mm &= mm - 1;
                        st[q] = mm ? (uint32_t)(__ffsll((long long)mm) - 1) : 0u;
                    } else st[q] = v > 65535u ? 65535u : v;
                }
                mk[q] = __builtin_amdgcn_ballot_w64(em[q]);
                nbs[q] = total;
                total += (uint32_t)__popcll(mk[q]);
            }
            if (total != 0u && b.fixed_cands && total <= GM_FIXED_C) {            // wave-uniform
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (em[q]) {
                        const uint32_t idx = nbs[q] + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk[q] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk[q], 0u));
                        GmCand c;
                        c.rs = rs; c.b = ky[q]; c.step = (uint16_t)st[q]; c.flags = 4; c.pad = 0; c.score = 0.0f;
                        b.fixed_cands[GM_FIXED_AT(b, rs, idx)] = c;
                    }
                if (lane == 0) b.fixed_cnt[rs] = (uint8_t)total;
            } else if (total != 0u) {                    // wave-uniform
                const uint32_t shard = rs & (GM_NSHARD - 1);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&b.shard_cnt[shard * GM_SHARD_STRIDE], total);
                base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (em[q]) {
                        const uint32_t idx = base + nbs[q] + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk[q] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk[q], 0u));
                        if (idx < b.cand_region) {
                            GmCand c;
                            c.rs = rs; c.b = ky[q]; c.step = (uint16_t)st[q]; c.flags = 4; c.pad = 0; c.score = 0.0f;
                            b.cands[(size_t)shard * b.cand_region + idx] = c;
                        }
                    }
            }
            if (s_any0) {                                // b = 0 (only if some wave saw such a vote): cumulative per-step counts
                const uint32_t run = gm_wave_scan_incl(s_cnt0[lane]);
                const uint32_t total0 = __builtin_amdgcn_readlane(run, 63);
                const unsigned long long reached = __builtin_amdgcn_ballot_w64(run >= (uint32_t)p.kmin);
                const bool emit = lane == 0 && total0 >= (uint32_t)p.kmin;
                const uint32_t step = p.nw ? (uint32_t)(__ffsll((long long)reached) - 1) : (total0 > 65535u ? 65535u : total0);
                gm_emit<GmLdsTable>(b, emit, rs, 0u, step, 4);
            }
        }
        // wave 0 zeroes the half it emitted from (BIG: the table)
        if (BIGT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_tabv[lane + 64 * k] = make_uint4(0u, 0u, 0u, 0u);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_r0v[256 + lane + 64 * k] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

// ---- k_vote_tiny: one WAVEFRONT per read x strand, for at most 256 SA hits in at most 32 groups of 16 ranks ----------------
// Long seeds on a large reference (-m 14 on 3.1 Gbp: 13 seeds x ~12 hits) leave k_vote_slots latency-bound: a workgroup's life is
// three dependent HBM round trips (seeds, SA ranks, candidate reservation) and 10 KB of LDS allow 16 of them per CU.  Here a read x
// strand is one wave and 4.9 KB of LDS (32 per CU: every wave slot of the CU), a descriptor covers 16 consecutive ranks of one seed (a 64-lane step holds
// four seeds' hits: ~75 % of the lanes carry a hit instead of ~20 %), there is no first counting filter: the hits go through
// the second filter (4096 x 2 bit: seen / seen again) into a 128-slot exact table as in k_vote_slots - straight from their load
// steps when there are four of those (at most 16 groups), through a dense list of 64-hit chunks when there are eight.
// More hits or groups than that -> b.big_list -> k_vote_fast_list; a table that fills up -> the retry kernel.
#define GMT_Q 32                         // 16-rank groups per read x strand
#define GMT_LCAP 256                     // hits per read x strand
// the part of k_vote_tiny after its descriptors are in LDS, unrolled for UU load steps (4 groups of 16 ranks each) and LQ list
// chunks of 64 hits (LQ = 0: no list): the kernel is vector-issue bound, and a read x strand of 13 seeds x ~12 hits fills 4 of the
// 8 steps - the step counts are wave-uniform, so the kernel picks the instantiation instead of walking empty steps
template <bool MASK64, bool FULL, bool SEED, int UU, int LQ>
__device__ __forceinline__ void gm_vote_tiny_body(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, const uint32_t rs, const int lane, const uint32_t E,
                                                  uint4* s_r0v, uint32_t* s_lbp, uint8_t* s_lt, const uint2* s_desc, uint32_t* s_cnt0) {
    constexpr int T2 = 128;
    uint32_t* const s_r0 = reinterpret_cast<uint32_t*>(s_r0v);
    const uint32_t* const src = FULL ? ix.full_sa : b.coords + b.entry_off[rs];
    // ---- loads: step j holds groups 4j .. 4j+3, lane -> group 4j + lane / 16, rank = first rank + lane % 16
    uint32_t bpv[UU], tg[UU];
    const uint32_t sub = (uint32_t)lane & 15u;
#pragma unroll
    for (int j = 0; j < UU; ++j) {
        const uint2 d = s_desc[4 * j + (lane >> 4)];
        const bool valid = sub < (d.y >> 24);
        tg[j] = (d.y >> 16) & 63u;
        uint32_t v = 0;
        if (valid) v = src[d.x + sub];
        bpv[j] = valid ? __builtin_elementwise_sub_sat(v, d.y & 0xFFFFu) : 0xFFFFFFFFu;      // :267; 0xFFFFFFFF = no hit (never a window start)
    }
    // ---- LQ > 0: every hit goes to the list (dense chunks of 64); LQ = 0: the hits stay in their load steps.  b = 0 votes are counted per step tag
    constexpr int NP = LQ == 0 ? UU : LQ;
    uint32_t wcount = 0;
    bool any0 = false;
    uint32_t bp4[NP];
#pragma unroll
    for (int j = 0; j < UU; ++j) {
        const bool hit = bpv[j] != 0xFFFFFFFFu;
        const bool z = hit && bpv[j] == 0u;
        if (z) atomicAdd(&s_cnt0[tg[j]], 1u);
        any0 |= z;
        const bool pass = hit && !z;
        if constexpr (LQ == 0) bp4[j] = pass ? bpv[j] : 0u;
        else {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
            if (pass) {
                const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, wcount));      // < E <= GMT_LCAP
                s_lbp[at] = bpv[j]; s_lt[at] = (uint8_t)tg[j];
            }
            wcount += (uint32_t)__popcll(m);
        }
    }
    const bool wave_any0 = __builtin_amdgcn_ballot_w64(any0) != 0ull;
    if constexpr (LQ != 0) __syncthreads();
    uint32_t* const keys = s_r0 + 256; uint32_t* const vals = keys + T2; uint32_t* const mlo = keys + 2 * T2; uint32_t* const mhi = keys + 3 * T2;
    const uint32_t n_l = wcount;
    const uint32_t thr = (uint32_t)(p.kmin < 1 ? 1 : p.kmin);
    const bool few = E <= 136u;                      // wave-uniform: which filter (measured: counters 8 % ahead at ~120 hits, two bits 5 % at ~156)
#pragma unroll
    for (int q = 0; q < NP; ++q)
        if constexpr (LQ != 0) { const uint32_t i = 64u * q + (uint32_t)lane; bp4[q] = i < n_l ? s_lbp[i] : 0u; }
    if (few) {                                       // 512 counters of 16 bits: one non-returning atomic per hit
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (bp4[q] != 0u) {
                const uint32_t h2 = (bp4[q] * 0x85EBCA6Bu) >> 23;
                atomicAdd(&s_r0[h2 & 255u], 1u << ((h2 >> 8) << 4));
            }
    } else {
        // 4096 slots of two bits in the same 256 words - "seen" and "seen again".  At ~160 hits a slot is shared by chance by ~4 % of
        // them (512 counters: ~30 %, and every such hit goes through the CAS loop of the table); the returning atomic costs ~7
        // instructions per pass, which is why the counters stay for few hits (wave-uniform choice)
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (bp4[q] != 0u) {
                const uint32_t h2 = (bp4[q] * 0x85EBCA6Bu) >> 20, sh = (h2 >> 8) << 1;
                const uint32_t old = atomicOr(&s_r0[h2 & 255u], 1u << sh);
                if ((old >> sh) & 1u) atomicOr(&s_r0[h2 & 255u], 2u << sh);
            }
    }
    __syncthreads();
    bool reached[NP];
    if (few) {
#pragma unroll
        for (int q = 0; q < NP; ++q) { const uint32_t h2 = (bp4[q] * 0x85EBCA6Bu) >> 23; reached[q] = ((s_r0[h2 & 255u] >> ((h2 >> 8) << 4)) & 0xFFFFu) >= thr; }
    } else {
#pragma unroll
        for (int q = 0; q < NP; ++q) { const uint32_t h2 = (bp4[q] * 0x85EBCA6Bu) >> 20; reached[q] = ((s_r0[h2 & 255u] >> ((h2 >> 8) << 1)) & (thr >= 2u ? 2u : 1u)) != 0u; }
    }
    bool full = false;
    uint32_t nkeys = 0;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        bool fresh = false;
        if (bp4[q] != 0u && reached[q]) {
            uint32_t t;
            if constexpr (LQ == 0) t = tg[q]; else t = s_lt[64u * q + (uint32_t)lane];
            const uint32_t bp = bp4[q];
            uint32_t slot = (bp * 0x9E3779B1u) >> 25;
            uint32_t old;
            int probes = 0;
            while ((old = atomicCAS(&keys[slot], 0u, bp)) != 0u && old != bp && ++probes < T2) slot = (slot + 1) & (T2 - 1);
            fresh = old == 0u;
            if (!(old == 0u || old == bp)) full = true;
            else {
                atomicAdd(&vals[slot], 1u);
                if (t < 32) atomicOr(&mlo[slot], 1u << t);
                else if (MASK64) atomicOr(&mhi[slot], 1u << (t - 32));
            }
        }
        nkeys += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fresh));
    }
    __syncthreads();
    if (__builtin_amdgcn_ballot_w64(full) != 0ull || nkeys > (uint32_t)(T2 * 3 / 4)) {     // hand this read x strand to the global-table kernel
        if (lane == 0) {
            if (SEED) (void)gm_seed_walk_ool(gm_kargs(), rs, nullptr, 0);      // the retry kernel reads the seed row
            b.rs_overflow[rs] = 1;
            const uint32_t j = atomicAdd(b.n_retry, 1u);
            const uint32_t need = 2 * E; uint32_t sz = 1024; while (sz < need && sz < 0x80000000u) sz <<= 1;
            const unsigned long long off = atomicAdd(&b.counters[GMK_HEAVY_SLOTS], (unsigned long long)sz);
            b.retry_list[j] = rs;
            b.retry_off[j] = off;
            atomicAdd(&b.counters[GMK_OVERFLOW_RS], 1ull);
        }
        return;
    }
    {   // ---- emit: two table slots per lane, one candidate reservation per wave (as in k_vote_slots)
        bool em[2]; uint32_t ky[2], st[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t slot = (uint32_t)(q * 64 + lane);
            const uint32_t key = keys[slot], v = vals[slot];
            em[q] = key != 0u && v >= (uint32_t)p.kmin;
            ky[q] = key; st[q] = 0;
            if (em[q]) {
                if (p.nw) {
                    unsigned long long m = (unsigned long long)mlo[slot] | (MASK64 ? ((unsigned long long)mhi[slot] << 32) : 0ull);
                    for (int r = 1; r < p.kmin; ++r) m &= m - 1;
                    st[q] = m ? (uint32_t)(__ffsll((long long)m) - 1) : 0u;
                } else st[q] = v > 65535u ? 65535u : v;
            }
        }
        const unsigned long long m0 = __builtin_amdgcn_ballot_w64(em[0]), m1 = __builtin_amdgcn_ballot_w64(em[1]);
        const uint32_t n0 = (uint32_t)__popcll(m0), n1 = (uint32_t)__popcll(m1);
        if (n0 + n1 != 0u && b.fixed_cands && n0 + n1 <= GM_FIXED_C) {       // wave-uniform: own slots, no counter to wait for
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (em[q]) {
                    const unsigned long long mq = q ? m1 : m0;
                    const uint32_t idx = (q ? n0 : 0u) + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
                    GmCand c;
                    c.rs = rs; c.b = ky[q]; c.step = (uint16_t)st[q]; c.flags = 4; c.pad = 0; c.score = 0.0f;
                    b.fixed_cands[GM_FIXED_AT(b, rs, idx)] = c;
                }
            if (lane == 0) b.fixed_cnt[rs] = (uint8_t)(n0 + n1);
        } else if (n0 + n1 != 0u) {                  // wave-uniform
            const uint32_t shard = blockIdx.x & (GM_NSHARD - 1);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&b.shard_cnt[shard * GM_SHARD_STRIDE], n0 + n1);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (em[q]) {
                    const unsigned long long mq = q ? m1 : m0;
                    const uint32_t idx = base + (q ? n0 : 0u) + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
                    if (idx < b.cand_region) {
                        GmCand c;
                        c.rs = rs; c.b = ky[q]; c.step = (uint16_t)st[q]; c.flags = 4; c.pad = 0; c.score = 0.0f;
                        b.cands[(size_t)shard * b.cand_region + idx] = c;
                    }
                }
        }
    }
    if (wave_any0) {                                  // b = 0: cumulative per-step counts
        const uint32_t run = gm_wave_scan_incl(s_cnt0[lane]);
        const uint32_t total = __builtin_amdgcn_readlane(run, 63);
        const unsigned long long reached = __builtin_amdgcn_ballot_w64(run >= (uint32_t)p.kmin);
        const bool emit = lane == 0 && total >= (uint32_t)p.kmin;
        const uint32_t step = p.nw ? (uint32_t)(__ffsll((long long)reached) - 1) : (total > 65535u ? 65535u : total);
        gm_emit<GmLdsTable>(b, emit, rs, 0u, step, 4);
    }
}

template <bool MASK64, bool FULL, bool SEED>
__global__ void __launch_bounds__(64, 8) k_vote_tiny(GmDevIndex ix, GmDevParams p, GmDevBatch b) {
    __shared__ uint4 s_r0v[192];                      // 3 KB: words [0,256) = 4096 x 2-bit filter, [256,768) = 128 x key | votes | low mask | high mask
    __shared__ uint32_t s_lbp[GMT_LCAP];
    __shared__ uint8_t s_lt[GMT_LCAP];
    __shared__ uint2 s_desc[GMT_Q];                   // {SA rank (flat entry index if !FULL) of the group's first hit, read offset | tag << 16 | hits << 24}
    __shared__ uint32_t s_cnt0[64];
    const uint32_t rs = blockIdx.x;                   // grid = 2n
    const int lane = threadIdx.x;
    GmSeed sd; uint32_t ns, ie_all;
    if (!gm_tiny_seeds<SEED>(ix, p, b, rs, lane, reinterpret_cast<GmSeed*>(s_r0v), sd, ns, ie_all)) return;
    const bool cut = p.nw && p.fast && ns > 1;
    if (cut) ns = 1;
    const uint32_t cnt = (uint32_t)lane < ns ? sd.l - sd.k + 1 : 0u;
    const uint32_t nq = (cnt + 15u) >> 4;
    uint32_t ie;
    if (SEED && !cut && __builtin_amdgcn_ballot_w64(ie_all == 0xFFFFFFFFu) == 0ull) ie = ie_all;      // wave-uniform
    else ie = gm_wave_scan_incl(cnt);
    const uint32_t iq = gm_wave_scan_incl(nq);
    const uint32_t E = __builtin_amdgcn_readlane(ie, 63), Q = __builtin_amdgcn_readlane(iq, 63);
    if (Q == 0) return;                               // wave-uniform: nothing to vote on
    if (Q > GMT_Q || E > GMT_LCAP) {                  // wave-uniform: hand over to the list kernel
        if (SEED && (uint32_t)lane < b.max_seeds) b.seeds[(size_t)rs * b.max_seeds + lane] = sd;
        if (lane == 0) { const uint32_t at = atomicAdd(b.n_big, 1u); b.big_list[at] = rs; }
        return;
    }
    if (lane < GMT_Q) s_desc[lane] = make_uint2(0u, 0u);
    s_cnt0[lane] = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) s_r0v[lane + 64 * k] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    {
        const uint32_t q0 = iq - nq, e0 = ie - cnt;
        for (uint32_t j = 0; j < nq; ++j) {
            const uint32_t left = cnt - 16u * j;
            s_desc[q0 + j] = make_uint2((FULL ? sd.k : e0) + 16u * j, sd.pos | ((uint32_t)lane << 16) | ((left < 16u ? left : 16u) << 24));
        }
    }
    __syncthreads();
    if (Q <= 16u) {
        // four load steps: the hits stay where they were loaded (no list: with the two-bit filter few hits reach the table, and
        // the 3 - 4 dense chunks the list buys cost more than the fourth sparse step; measured 16.6 -> 16.45 ms)
        if (E <= 64u) gm_vote_tiny_body<MASK64, FULL, SEED, 4, 1>(ix, p, b, rs, lane, E, s_r0v, s_lbp, s_lt, s_desc, s_cnt0);      // one dense chunk
        else gm_vote_tiny_body<MASK64, FULL, SEED, 4, 0>(ix, p, b, rs, lane, E, s_r0v, s_lbp, s_lt, s_desc, s_cnt0);
    } else {
        if (E <= 192u) gm_vote_tiny_body<MASK64, FULL, SEED, 8, 3>(ix, p, b, rs, lane, E, s_r0v, s_lbp, s_lt, s_desc, s_cnt0);
        else gm_vote_tiny_body<MASK64, FULL, SEED, 8, 4>(ix, p, b, rs, lane, E, s_r0v, s_lbp, s_lt, s_desc, s_cnt0);
    }
}

// ---- k_vote_tiny2: the same one-wave form for up to 384 hits in up to 64 groups (150-bp reads on a human-size reference: 20
// seeds x ~12 hits; 10-mers on 20 Mbp).  No list at all: the hits stay in registers (16 steps), go through the 8192 x 2-bit
// filter and, where their slot was hit again, straight into a 256-slot exact table.  6.8 KB of LDS: 23 read x strands per CU.
#define GMT2_Q 64
#define GMT2_E 384
template <bool MASK64, bool FULL, bool SEED>
__global__ void __launch_bounds__(64, 6) k_vote_tiny2(GmDevIndex ix, GmDevParams p, GmDevBatch b) {
    constexpr int U = GMT2_Q / 4, T2 = 256;
    __shared__ uint4 s_r0v[384];                      // 6 KB: words [0,512) = 8192 x 2-bit filter, [512,1536) = 256 x key | votes | low mask | high mask
    __shared__ uint2 s_desc[GMT2_Q];
    __shared__ uint32_t s_cnt0[64];
    uint32_t* const s_r0 = reinterpret_cast<uint32_t*>(s_r0v);
    const uint32_t rs = blockIdx.x;                   // grid = 2n
    const int lane = threadIdx.x;
    GmSeed sd; uint32_t ns, ie_all;
    if (!gm_tiny_seeds<SEED>(ix, p, b, rs, lane, reinterpret_cast<GmSeed*>(s_r0v), sd, ns, ie_all)) return;
    const bool cut = p.nw && p.fast && ns > 1;
    if (cut) ns = 1;
    const uint32_t cnt = (uint32_t)lane < ns ? sd.l - sd.k + 1 : 0u;
    const uint32_t nq = (cnt + 15u) >> 4;
    uint32_t ie;
    if (SEED && !cut && __builtin_amdgcn_ballot_w64(ie_all == 0xFFFFFFFFu) == 0ull) ie = ie_all;      // wave-uniform
    else ie = gm_wave_scan_incl(cnt);
    const uint32_t iq = gm_wave_scan_incl(nq);
    const uint32_t E = __builtin_amdgcn_readlane(ie, 63), Q = __builtin_amdgcn_readlane(iq, 63);
    if (Q == 0) return;                               // wave-uniform: nothing to vote on
    if (Q > GMT2_Q || E > GMT2_E) {                   // wave-uniform: hand over to the list kernel
        if (SEED && (uint32_t)lane < b.max_seeds) b.seeds[(size_t)rs * b.max_seeds + lane] = sd;
        if (lane == 0) { const uint32_t at = atomicAdd(b.n_big, 1u); b.big_list[at] = rs; }
        return;
    }
    s_desc[lane] = make_uint2(0u, 0u);
    s_cnt0[lane] = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s_r0v[lane + 64 * k] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    {
        const uint32_t q0 = iq - nq, e0 = ie - cnt;
        for (uint32_t j = 0; j < nq; ++j) {
            const uint32_t left = cnt - 16u * j;
            s_desc[q0 + j] = make_uint2((FULL ? sd.k : e0) + 16u * j, sd.pos | ((uint32_t)lane << 16) | ((left < 16u ? left : 16u) << 24));
        }
    }
    __syncthreads();
    const uint32_t* const src = FULL ? ix.full_sa : b.coords + b.entry_off[rs];
    uint32_t bpv[U], tg[U];
    const uint32_t sub = (uint32_t)lane & 15u;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const uint2 d = s_desc[4 * j + (lane >> 4)];
        const bool valid = sub < (d.y >> 24);
        tg[j] = (d.y >> 16) & 63u;
        uint32_t v = 0;
        if (valid) v = src[d.x + sub];
        bpv[j] = valid ? __builtin_elementwise_sub_sat(v, d.y & 0xFFFFu) : 0xFFFFFFFFu;      // :267; 0xFFFFFFFF = no hit
    }
    // ---- pass A: b = 0 votes per step tag; every other hit counts in the filter.  Afterwards bpv = 0 means "nothing here".
    bool any0 = false;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const bool hit = bpv[j] != 0xFFFFFFFFu;
        const bool z = hit && bpv[j] == 0u;
        if (z) atomicAdd(&s_cnt0[tg[j]], 1u);
        any0 |= z;
        if (!hit) bpv[j] = 0u;
        if (bpv[j] != 0u) {
            const uint32_t h2 = (bpv[j] * 0x85EBCA6Bu) >> 19, sh = (h2 >> 9) << 1;       // 8192 slots of two bits: seen / seen again (k_vote_tiny)
            const uint32_t old = atomicOr(&s_r0[h2 & 511u], 1u << sh);
            if ((old >> sh) & 1u) atomicOr(&s_r0[h2 & 511u], 2u << sh);
        }
    }
    const bool wave_any0 = __builtin_amdgcn_ballot_w64(any0) != 0ull;
    __syncthreads();
    // ---- pass B: hits whose filter slot reached -k enter the exact table
    uint32_t* const keys = s_r0 + 512; uint32_t* const vals = keys + T2; uint32_t* const mlo = keys + 2 * T2; uint32_t* const mhi = keys + 3 * T2;
    const uint32_t thr = (uint32_t)(p.kmin < 1 ? 1 : p.kmin);
    bool full = false;
    uint32_t nkeys = 0;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const uint32_t bp = bpv[j];
        const uint32_t h2 = (bp * 0x85EBCA6Bu) >> 19;
        const uint32_t c = (s_r0[h2 & 511u] >> ((h2 >> 9) << 1)) & 3u;
        bool fresh = false;
        if (bp != 0u && (c & (thr >= 2u ? 2u : 1u))) {
            const uint32_t t = tg[j];
            uint32_t slot = (bp * 0x9E3779B1u) >> 24;
            uint32_t old;
            int probes = 0;
            while ((old = atomicCAS(&keys[slot], 0u, bp)) != 0u && old != bp && ++probes < T2) slot = (slot + 1) & (T2 - 1);
            fresh = old == 0u;
            if (!(old == 0u || old == bp)) full = true;
            else {
                atomicAdd(&vals[slot], 1u);
                if (t < 32) atomicOr(&mlo[slot], 1u << t);
                else if (MASK64) atomicOr(&mhi[slot], 1u << (t - 32));
            }
        }
        nkeys += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fresh));
    }
    __syncthreads();
    if (__builtin_amdgcn_ballot_w64(full) != 0ull || nkeys > (uint32_t)(T2 * 3 / 4)) {     // hand this read x strand to the global-table kernel
        if (lane == 0) {
            if (SEED) (void)gm_seed_walk_ool(gm_kargs(), rs, nullptr, 0);      // the retry kernel reads the seed row
            b.rs_overflow[rs] = 1;
            const uint32_t j = atomicAdd(b.n_retry, 1u);
            const uint32_t need = 2 * E; uint32_t sz = 1024; while (sz < need && sz < 0x80000000u) sz <<= 1;
            const unsigned long long off = atomicAdd(&b.counters[GMK_HEAVY_SLOTS], (unsigned long long)sz);
            b.retry_list[j] = rs;
            b.retry_off[j] = off;
            atomicAdd(&b.counters[GMK_OVERFLOW_RS], 1ull);
        }
        return;
    }
    {   // ---- emit: four table slots per lane, one candidate reservation per wave
        bool em[4]; uint32_t ky[4], st[4], nb[4];
        unsigned long long mk[4];
        uint32_t total = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t slot = (uint32_t)(q * 64 + lane);
            const uint32_t key = keys[slot], v = vals[slot];
            em[q] = key != 0u && v >= (uint32_t)p.kmin;
            ky[q] = key; st[q] = 0;
            if (em[q]) {
                if (p.nw) {
                    unsigned long long m = (unsigned long long)mlo[slot] | (MASK64 ? ((unsigned long long)mhi[slot] << 32) : 0ull);
                    for (int r = 1; r < p.kmin; ++r) m &= m - 1;
                    st[q] = m ? (uint32_t)(__ffsll((long long)m) - 1) : 0u;
                } else st[q] = v > 65535u ? 65535u : v;
            }
            mk[q] = __builtin_amdgcn_ballot_w64(em[q]);
            nb[q] = total;
            total += (uint32_t)__popcll(mk[q]);
        }
        if (total != 0u && b.fixed_cands && total <= GM_FIXED_C) {            // wave-uniform: own slots, no counter to wait for
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (em[q]) {
                    const uint32_t idx = nb[q] + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk[q] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk[q], 0u));
                    GmCand c;
                    c.rs = rs; c.b = ky[q]; c.step = (uint16_t)st[q]; c.flags = 4; c.pad = 0; c.score = 0.0f;
                    b.fixed_cands[GM_FIXED_AT(b, rs, idx)] = c;
                }
            if (lane == 0) b.fixed_cnt[rs] = (uint8_t)total;
        } else if (total != 0u) {                    // wave-uniform
            const uint32_t shard = blockIdx.x & (GM_NSHARD - 1);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&b.shard_cnt[shard * GM_SHARD_STRIDE], total);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (em[q]) {
                    const uint32_t idx = base + nb[q] + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk[q] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk[q], 0u));
                    if (idx < b.cand_region) {
                        GmCand c;
                        c.rs = rs; c.b = ky[q]; c.step = (uint16_t)st[q]; c.flags = 4; c.pad = 0; c.score = 0.0f;
                        b.cands[(size_t)shard * b.cand_region + idx] = c;
                    }
                }
        }
    }
    if (wave_any0) {                                  // b = 0: cumulative per-step counts
        const uint32_t run = gm_wave_scan_incl(s_cnt0[lane]);
        const uint32_t total = __builtin_amdgcn_readlane(run, 63);
        const unsigned long long reached = __builtin_amdgcn_ballot_w64(run >= (uint32_t)p.kmin);
        const bool emit = lane == 0 && total >= (uint32_t)p.kmin;
        const uint32_t step = p.nw ? (uint32_t)(__ffsll((long long)reached) - 1) : (total > 65535u ? 65535u : total);
        gm_emit<GmLdsTable>(b, emit, rs, 0u, step, 4);
    }
}


// retry path: one workgroup per overflowed read x strand, exact vote table in HBM (pre-set to EMPTY/0 by the host)
struct GmGlobalTable {
    uint32_t* keys; uint32_t* vals; uint32_t mask; int bits;
    __device__ __forceinline__ uint32_t slot0(uint32_t key) const { return (key * 0x85EBCA6Bu) >> (32 - bits); }
};

__global__ void __launch_bounds__(256) k_vote_retry(GmDevIndex ix, GmDevParams p, GmDevBatch b, int use_full_sa, uint32_t j0, uint32_t n_retry) {
    if (blockIdx.x >= n_retry) return;
    uint32_t j = j0 + blockIdx.x;                    // entries [j0, j0 + n_retry) of the retry list
    uint32_t rs = b.retry_list[j];
    uint32_t ns = b.n_seeds[rs];
    uint32_t E = b.n_entries[rs];
    uint32_t need = 2 * E; uint32_t sz = 1024; int bits = 10;
    while (sz < need && sz < 0x80000000u) { sz <<= 1; ++bits; }
    GmGlobalTable tb; tb.keys = b.gtab_keys + b.retry_off[j]; tb.vals = b.gtab_vals + b.retry_off[j]; tb.mask = sz - 1; tb.bits = bits;
    const GmSeed* seeds = b.seeds + (size_t)rs * b.max_seeds;
    uint64_t coff = use_full_sa ? 0 : b.entry_off[rs];
    if (p.nw && p.fast) ns = 1;
    for (uint32_t t = 0; t < ns; ++t) {
        GmSeed sd = seeds[t];
        uint32_t cnt = sd.l - sd.k + 1;
        for (uint32_t base = 0; base < cnt; base += 256) {
            uint32_t idx = base + threadIdx.x;
            bool emit = false, fresh;
            uint32_t bp = 0;
            if (idx < cnt) {
                uint32_t c = gm_coord(ix, b, use_full_sa, sd, idx, coff);
                bp = (c <= sd.pos) ? 0u : c - sd.pos;
                uint32_t slot = gm_table_insert(tb, bp, &fresh);
                if (slot != GM_EMPTY) {
                    uint32_t v = atomicAdd(&tb.vals[slot], 1u) + 1u;
                    emit = p.nw && v == (uint32_t)p.kmin;
                }
            }
            gm_emit<GmGlobalTable>(b, emit, rs, bp, t, 0);
        }
        coff += cnt;
        __syncthreads();                             // all votes of seed t are in before seed t+1 starts
    }
    if (!p.nw) {
        for (uint32_t q = threadIdx.x; q < sz; q += 256) {
            uint32_t key = tb.keys[q], v = tb.vals[q];
            bool emit = key != GM_EMPTY && v >= (uint32_t)p.kmin;
            gm_emit<GmGlobalTable>(b, emit, rs, key, v > 65535u ? 65535u : v, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// banded NW score: 8 lanes per candidate (7 band diagonals + 1 idle), anti-diagonal sweep.
// Lane d holds the newest cell of diagonal delta = j - i = d - 3; cells of one anti-diagonal i+j = s are independent,
// their three inputs are this lane's previous cell (s+2) and the two neighbour lanes' cells (s+1), exchanged by
// wave shuffles.  The read rows (called base, quality) and the 2-bit window are staged in LDS.
// ------------------------------------------------------------------------------------------------
#define GM_NW_NCOFF 1024        // contig offsets cached in LDS when there are at most this many
#define GM_NW_HDR (512 * sizeof(float2) + 16 * sizeof(float) + GM_NW_NCOFF * sizeof(uint32_t) + (GM_NSHARD + 4) * sizeof(uint32_t))
struct GmNwLds {
    float2* lut;            // 512 entries
    float* sg;              // 4 x 4 substitution rows a,c,g,t
    uint32_t* coff;         // contig offsets
    uint32_t* pre;          // prefix of the candidate shard counts
    uint16_t* rows;         // G groups x Lp   (G = blockDim / 8 candidates per workgroup)
    uint8_t* win;           // G groups x Lp
};

__device__ __forceinline__ GmNwLds gm_nw_lds(unsigned char* raw, uint32_t Lp, uint32_t G) {
    GmNwLds s;
    s.lut = reinterpret_cast<float2*>(raw);
    s.sg = reinterpret_cast<float*>(raw + 512 * sizeof(float2));
    s.coff = reinterpret_cast<uint32_t*>(raw + 512 * sizeof(float2) + 16 * sizeof(float));
    s.pre = s.coff + GM_NW_NCOFF;
    s.rows = reinterpret_cast<uint16_t*>(raw + GM_NW_HDR);
    s.win = raw + GM_NW_HDR + (size_t)G * Lp * 2;
    return s;
}

// stage read rows (in strand orientation) and the reference window of one candidate; 8 lanes cooperate.
// All global loads of a lane are independent 8-byte (read) / 1-byte (packed reference) loads issued back to back.
__device__ __forceinline__ void gm_stage(const GmDevIndex& ix, const GmDevBatch& b, uint16_t* rows, uint8_t* win,
                                         uint32_t r, uint32_t strand, uint32_t L, uint32_t b0, int d) {
    const uint8_t* rb = b.bases + (size_t)r * b.stride;
    const uint8_t* rq = b.quals + (size_t)r * b.stride;
    for (uint32_t o = (uint32_t)d * 8; o < L; o += 64) {                  // rows are 8-byte aligned
        const uint2 bw = *reinterpret_cast<const uint2*>(rb + o);
        const uint2 qw = *reinterpret_cast<const uint2*>(rq + o);
#pragma unroll
        for (uint32_t t = 0; t < 8; ++t) {
            uint32_t src = o + t;
            if (src < L) {
                uint32_t ch = ((t < 4 ? bw.x : bw.y) >> ((t & 3) * 8)) & 255u;
                uint32_t qc = ((t < 4 ? qw.x : qw.y) >> ((t & 3) * 8)) & 255u;
                uint32_t code = gm_nt4(ch);
                if (strand && code < 4) code = 3 - code;
                rows[strand ? L - 1 - src : src] = (uint16_t)((code << 8) | qc);   // reverse_comp_cpy SequenceOperations.h:149-161
            }
        }
    }
    const uint32_t y0 = b0 >> 2, y1 = (b0 + L - 1) >> 2;                   // bytes of the 2-bit reference holding the window
    for (uint32_t y = y0 + (uint32_t)d * 4; y <= y1; y += 32) {
        uint32_t by[4];
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) by[t] = (y + t <= y1) ? ix.pac[y + t] : 0u;
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t)
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                uint32_t pp = ((y + t) << 2) + q;                           // _get_pac src/bntseq.c:225
                if (pp >= b0 && pp < b0 + L) win[pp - b0] = (uint8_t)((by[t] >> ((~pp & 3u) << 1)) & 3u);
            }
    }
}

__global__ void __launch_bounds__(256) k_nw(GmDevIndex ix, GmDevParams p, GmDevBatch b, uint32_t Lp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const uint32_t G = blockDim.x >> 3;
    GmNwLds S = gm_nw_lds(s_raw, Lp, G);
    for (int q = threadIdx.x; q < 512; q += blockDim.x) S.lut[q] = p.lut[q];
    if (threadIdx.x < 16) S.sg[threadIdx.x] = p.S256[(size_t)("acgt"[threadIdx.x >> 2]) * 4 + (threadIdx.x & 3)];
    const bool lds_coff = ix.n_seqs + 1 <= GM_NW_NCOFF;
    if (lds_coff) for (uint32_t q = threadIdx.x; q <= ix.n_seqs; q += blockDim.x) S.coff[q] = ix.contig_off[q];
    const uint32_t* coff = lds_coff ? S.coff : ix.contig_off;
    __syncthreads();
    const int lane = gm_lane(), wave = threadIdx.x >> 6;
    const int g = (threadIdx.x >> 3), d = lane & 7, delta = d - 3;
    uint16_t* rows = S.rows + (size_t)g * Lp;
    uint8_t* win = S.win + (size_t)g * Lp;
    const uint32_t n_cands = gm_cand_prefix(b, S.pre);
    const float gap = p.gap, gap4 = __fmul_rn(p.gap, 4.0f);
    unsigned long long cells = 0, accepted = 0;
    (void)wave;
    for (uint32_t base = blockIdx.x * G; base < n_cands; base += gridDim.x * G) {       // block-uniform trip count
        uint32_t wi = base + g;
        bool have = wi < n_cands;
        size_t ci = have ? gm_cand_slot(b, S.pre, wi) : 0;
        GmCand c; c.rs = 0; c.b = 0; c.step = 0; c.flags = 0; c.score = 0;
        if (have) c = b.cands[ci];
        uint32_t r = c.rs >> 1, strand = c.rs & 1;
        if (have && (c.flags & 4) && b.rs_overflow[c.rs]) have = false;                // superseded by the retry kernel
        uint32_t L = have ? b.len[r] : 0;
        bool ok = have && gm_window_ok(ix, coff, c.b, L);
        if (ok && p.nw) gm_stage(ix, b, rows, win, r, strand, L, c.b, d);
        __syncthreads();
        float result = 0.0f;
        if (p.nw) {
            const float2* lut = S.lut + ((r < b.illumina_until) ? 256 : 0);
            int absd = delta < 0 ? -delta : delta;
            float own = __fmul_rn(gap, (float)absd);      // boundary row/column: gGAP * (end - i), bin_seq.cpp:801-808
            int Lw = (int)L;                              // wave-uniform sweep length = longest read in the wave
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) { int o = __shfl_xor(Lw, off); Lw = o > Lw ? o : Lw; }
            // lane delta walks its diagonal from the bottom-right: cells (i, i+delta), i = i0 .. imin; it is due at s = 2i+delta,
            // i.e. every other step, so the substitution score of its NEXT cell is fetched from LDS in the idle step
            const int imin = delta < 0 ? -delta : 0;
            int icur = (ok && d < 7) ? ((int)L - 1 - (delta > 0 ? delta : 0)) : -1;
            float vcur = 0.0f;
            if (icur >= imin) {
                uint32_t row = rows[icur];
                float2 pq = lut[row & 255u];
                vcur = gm_get_val(row >> 8, pq.x, pq.y, S.sg + 4 * win[icur + delta]);
            }
            for (int s = 2 * Lw - 2; s >= 0; --s) {
                float a = gm_from_prev_lane(own);         // diagonal delta-1, cell (i+1, j)
                float bb = gm_from_next_lane(own);        // diagonal delta+1, cell (i, j+1)
                const bool act = icur >= imin && s == 2 * icur + delta;
                if (d == 0) a = (icur + 1 == (int)L) ? gap4 : GM_NEG_INF;             // outside the band except on the last row
                if (d == 6) bb = (icur + delta + 1 == (int)L) ? gap4 : GM_NEG_INF;    // ... or the last column
                if (act) {
                    float mm = __fadd_rn(own, vcur);
                    float g1 = __fadd_rn(a, gap);
                    float g2 = __fadd_rn(bb, gap);
                    own = gm_max3(mm, g1, g2);
                    ++cells;
                    --icur;
                    if (icur >= imin) {
                        uint32_t row = rows[icur];
                        float2 pq = lut[row & 255u];
                        vcur = gm_get_val(row >> 8, pq.x, pq.y, S.sg + 4 * win[icur + delta]);
                    }
                }
            }
            result = __shfl(own, (lane & ~7) + 3);        // nm[0][0] lives on diagonal 0
        } else {
            result = (float)c.step;                       // --no_nw: the score is the vote count (:70-76)
        }
        if (have && d == 3) {
            uint8_t fl = c.flags & 4;
            if (ok) {
                fl |= GMC_VALID;
                if (result > 0.0f) atomicMax(reinterpret_cast<int*>(&b.top_score[r]), __float_as_int(result));   // top_align_score (:95-98)
                if ((double)result >= b.min_score[r]) {  // :102
                    fl |= GMC_ACCEPT;
                    atomicAdd(&b.hit_count[r], 1u);
                    ++accepted;
                }
            }
            b.cands[ci].score = result;
            b.cands[ci].flags = fl;
        }
        __syncthreads();
    }
    gm_count(b, GMK_NW_CELLS, cells);
    gm_count(b, GMK_ACCEPTED, accepted);
}

// ------------------------------------------------------------------------------------------------
// banded NW score, one LANE per candidate (the throughput form).  The 7-cell band row lives in registers and is
// swept row by row from the bottom-right exactly like get_align_score_begin (src/bin_seq.cpp:819-843): cell order
// j = i+3 .. i-3, three adds and the reference's 3-way max per cell, fp32 without contraction.  Per row the four
// possible substitution scores are formed once (get_val for g = a,c,g,t) and each cell selects by its window base.
// Reads stream through two 8-byte words, the 2-bit reference through two 32-bit words; nothing is staged in LDS
// except the quality LUT.  (k_nw below is the 8-lanes-per-candidate anti-diagonal form; GM_NW=wave selects it.)
// ------------------------------------------------------------------------------------------------
#ifndef GM_NW_OCC
#define GM_NW_OCC 5       // measured: 4 -> 3.09 ms, 5 -> 2.97, 6 -> 3.03, 8 -> 3.16 (2 M reads)
#endif
// NCH > 0: the candidate's read row (bases and qualities, NCH 8-byte words each) is loaded into registers up front - every load of
// a lane goes out back to back and each 128-byte line is fetched once - instead of streaming one word per 8 DP rows (which re-fetched
// the lines ~9 times: 3.9 KB of HBM traffic per candidate for 208 bytes of read).  NCH = 0 keeps the streaming form (long reads).
// LDSR (NCH > 0): the candidate's read row waits in LDS instead of 4 NCH registers - [chunk][thread] 8-byte slots, so that a thread's
// own slot of ANY chunk sits on the same two banks (conflict-free whatever chunk each lane is at) and the row pick of every eighth DP
// row is two ds_read_b64 instead of a 13-way mask-and-or over 52 registers (229 vector instructions per pick, a sixth of the kernel)
// Waves per SIMD (NCH = 13): the kernel is a chain of dependent adds and max3 (a row's cells depend on each other), so it needs waves to
// fill the issue slots more than it needs registers - 4 waves with 12 spilled registers: 4.19 ms, 3 waves without spills: 4.52 ms.
template <int NCH, bool LDSR>
__global__ void __launch_bounds__(256, NCH > 0 ? (LDSR ? 2 : NCH <= 13 ? 4 : 3) : GM_NW_OCC) k_nw_lane(GmDevIndex ix, GmDevParams p, GmDevBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint2 s_rows[];          // LDSR: [2][NCH][256] bases, then qualities
    __shared__ float2 s_lut[512];
    __shared__ uint32_t s_coff[GM_NW_NCOFF];
    __shared__ uint32_t s_pre[GM_NSHARD + 4];
    for (int q = threadIdx.x; q < 512; q += 256) s_lut[q] = p.lut[q];
    const bool lds_coff = ix.n_seqs + 1 <= GM_NW_NCOFF;
    if (lds_coff) for (uint32_t q = threadIdx.x; q <= ix.n_seqs; q += 256) s_coff[q] = ix.contig_off[q];
    const uint32_t* coff = lds_coff ? s_coff : ix.contig_off;
    const uint32_t n_cands = gm_cand_prefix(b, s_pre);          // includes the barrier that publishes s_lut / s_coff
    float sg[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) sg[g][k] = p.S256[(size_t)("acgt"[g]) * 4 + k];
    const float gap = p.gap, gap4 = __fmul_rn(p.gap, 4.0f);
    const uint32_t* pac32 = reinterpret_cast<const uint32_t*>(ix.pac);
    unsigned long long cells = 0, accepted = 0;
    for (uint32_t wi = blockIdx.x * 256 + threadIdx.x; wi < n_cands; wi += gridDim.x * 256) {
        const size_t ci = gm_cand_slot(b, s_pre, wi);
        GmCand c = b.cands[ci];
        const uint32_t r = c.rs >> 1, strand = c.rs & 1;
        if ((c.flags & 4) && b.rs_overflow[c.rs]) continue;                 // superseded by the retry kernel
        const uint32_t L = b.len[r];
        const bool ok = gm_window_ok(ix, coff, c.b, L);
        float result = 0.0f;
        if (ok && p.nw) {
            const float2* lut = s_lut + ((r < b.illumina_until) ? 256 : 0);
            const uint8_t* rb = b.bases + (size_t)r * b.stride;
            const uint8_t* rq = b.quals + (size_t)r * b.stride;
            const int Li = (int)L;
            // band row of i+1: P[d] = nm[i+1][i+1+delta], delta = d-3.  Row L: gGAP * (L - j) for j <= L (bin_seq.cpp:805-808)
            float P[7];
#pragma unroll
            for (int d = 0; d < 7; ++d) P[d] = d <= 3 ? __fmul_rn(gap, (float)(3 - d)) : GM_NEG_INF;
            // window bases of the current row: W[d] = w[i+delta]; start with row L-1
            // the base of a band column as two v_perm_b32 selectors (bit 0 and bit 1 of its 2-bit code: take the low or the high
            // source register whole): picking a cell's value among the row's four is three byte-permutes, no compares
            uint32_t S1[7], S2[7];
            auto set_col = [&](int d, uint32_t code) { S1[d] = 0x03020100u + (code & 1u) * 0x04040404u; S2[d] = 0x03020100u + (code >> 1) * 0x04040404u; };
            uint32_t wword_lo = 0, wword_hi = 0; int wbase = -1;          // 16-base words [wbase*16, +16) and the next one
            auto wcode = [&](int j) -> uint32_t {                          // 2-bit code of the reference at window offset j
                uint32_t g = c.b + (uint32_t)j;
                int wi16 = (int)(g >> 4);
                if (wi16 != wbase && wi16 != wbase + 1) { wbase = wi16; wword_lo = pac32[wi16]; wword_hi = pac32[wi16 + 1]; }
                uint32_t word = wi16 == wbase ? wword_lo : wword_hi;
                return (word >> ((((g >> 2) & 3u) << 3) + ((~g & 3u) << 1))) & 3u;     // _get_pac src/bntseq.c:225 on a little-endian word
            };
#pragma unroll
            for (int d = 0; d < 7; ++d) { int j = Li - 1 + d - 3; set_col(d, (j >= 0 && j < Li) ? wcode(j) : 0u); }
            // the read streams through 8-byte words; the NEXT word is requested one chunk ahead so its latency hides under 8 rows
            const int src0 = strand ? 0 : Li - 1, cstep = strand ? 1 : -1, nchunk = (Li + 7) >> 3;
            int chunk = src0 >> 3;
            uint2 RB[(NCH > 0 && !LDSR) ? NCH : 1], RQ[(NCH > 0 && !LDSR) ? NCH : 1];
            auto pick = [&](int cch, uint2& bo, uint2& qo) {           // keeps the rows in registers (a chain of `cch == k ?` selects became a scratch array)
                if constexpr (LDSR) { bo = s_rows[(size_t)cch * 256 + threadIdx.x]; qo = s_rows[(size_t)(NCH + cch) * 256 + threadIdx.x]; return; }
                // a binary tree of selects over the bits of the chunk number: 12 selects per word for 13 chunks, where OR-ing the
                // chunks under 13 compare masks took 9 instructions per chunk (229 per pick, a sixth of the kernel)
                constexpr int N = NCH > 0 ? NCH : 1;
                uint32_t out[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {          // one word at a time: the tree's intermediate values of one word only are alive
                    uint32_t t[N];
#pragma unroll
                    for (int k = 0; k < N; ++k) t[k] = c == 0 ? RB[k].x : c == 1 ? RB[k].y : c == 2 ? RQ[k].x : RQ[k].y;
                    int n = N;
#pragma unroll
                    for (int lvl = 0; lvl < 5; ++lvl) {
                        if (n <= 1) break;
                        const bool hi = (((uint32_t)cch >> lvl) & 1u) != 0u;
                        const int m = (n + 1) / 2;
#pragma unroll
                        for (int i = 0; i < m; ++i) t[i] = (2 * i + 1 < n) ? (hi ? t[2 * i + 1] : t[2 * i]) : t[2 * i];
                        n = m;
                    }
                    out[c] = t[0];
                }
                bo = make_uint2(out[0], out[1]); qo = make_uint2(out[2], out[3]);
            };
            uint2 bw, qw, bn, qn;
            if constexpr (NCH > 0 && LDSR) {
                uint2 tb[NCH], tq[NCH];                                // all loads of the lane in flight, then its own LDS slots (read back by this thread only: no barrier)
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    tb[k] = make_uint2(0u, 0u); tq[k] = make_uint2(0u, 0u);
                    if (k < nchunk) { tb[k] = *reinterpret_cast<const uint2*>(rb + (k << 3)); tq[k] = *reinterpret_cast<const uint2*>(rq + (k << 3)); }
                }
#pragma unroll
                for (int k = 0; k < NCH; ++k) { s_rows[(size_t)k * 256 + threadIdx.x] = tb[k]; s_rows[(size_t)(NCH + k) * 256 + threadIdx.x] = tq[k]; }
                pick(chunk, bw, qw);
                bn = bw; qn = qw;
            } else if constexpr (NCH > 0) {
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    RB[k] = make_uint2(0u, 0u); RQ[k] = make_uint2(0u, 0u);
                    if (k < nchunk) { RB[k] = *reinterpret_cast<const uint2*>(rb + (k << 3)); RQ[k] = *reinterpret_cast<const uint2*>(rq + (k << 3)); }
                }
                pick(chunk, bw, qw);
                bn = bw; qn = qw;
            } else {
                bw = *reinterpret_cast<const uint2*>(rb + (chunk << 3)); qw = *reinterpret_cast<const uint2*>(rq + (chunk << 3));
                bn = bw; qn = qw;
                int nc = chunk + cstep; if (nc >= 0 && nc < nchunk) { bn = *reinterpret_cast<const uint2*>(rb + (nc << 3)); qn = *reinterpret_cast<const uint2*>(rq + (nc << 3)); }
            }
            // one DP row.  EDGE = the row touches column L, row L or column -1 (the first 4 and the last 3 rows); the rows in
            // between - nearly all of them - need none of those tests
            // (Measured and dropped: rows taken seven at a time with the cells moving over fixed selector slots instead of the selectors
            // sliding - 24 moves per row less, but seven copies of the row body: 4.50 -> 4.83 ms per 10.7 M candidates.)
            auto dp_row = [&](const int i, auto edge_tag) {
                constexpr bool EDGE = decltype(edge_tag)::value;
                // PWM row i in strand orientation (reverse_comp_cpy SequenceOperations.h:149-161)
                const int src = strand ? Li - 1 - i : i;
                if ((src >> 3) != chunk) {
                    chunk = src >> 3;
                    if constexpr (NCH > 0) pick(chunk, bw, qw);
                    else {
                        bw = bn; qw = qn;
                        int nc = chunk + cstep;
                        if (nc >= 0 && nc < nchunk) { bn = *reinterpret_cast<const uint2*>(rb + (nc << 3)); qn = *reinterpret_cast<const uint2*>(rq + (nc << 3)); }
                    }
                }
                const uint32_t sh = (uint32_t)(src & 3) << 3;
                const uint32_t ch = (((src & 4) ? bw.y : bw.x) >> sh) & 255u;
                const uint32_t qc = (((src & 4) ? qw.y : qw.x) >> sh) & 255u;
                uint32_t code = gm_nt4(ch);
                if (strand && code < 4) code = 3 - code;
                const float2 pq = lut[qc];
                float v4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) v4[g] = gm_get_val(code, pq.x, pq.y, sg[g]);
                const float lastcol = EDGE ? __fmul_rn(gap, (float)(unsigned)(Li - i)) : 0.0f;      // nm[i][L] = gGAP * (L - i)
#pragma unroll
                for (int d = 6; d >= 0; --d) {
                    const int j = i + d - 3;
                    const uint32_t lo = __builtin_amdgcn_perm(__float_as_uint(v4[1]), __float_as_uint(v4[0]), S1[d]);
                    const uint32_t hi = __builtin_amdgcn_perm(__float_as_uint(v4[3]), __float_as_uint(v4[2]), S1[d]);
                    const float val = __uint_as_float(__builtin_amdgcn_perm(hi, lo, S2[d]));
                    // the row is updated in place, from the last column down: P[d] (diagonal) and P[d - 1] (above) still hold row i + 1,
                    // P[d + 1] already row i
                    const float up = d > 0 ? P[d - 1] : ((EDGE && i + 1 == Li) ? gap4 : GM_NEG_INF);        // nm[i+1][j]
                    const float left = d < 6 ? P[d + 1] : ((EDGE && j + 1 == Li) ? gap4 : GM_NEG_INF);       // nm[i][j+1]
                    const float mm = __fadd_rn(P[d], val);
                    const float g1 = __fadd_rn(up, gap);
                    const float g2 = __fadd_rn(left, gap);
                    // bin_seq::max_flt (src/bin_seq.cpp:1013-1026) is a compare chain; on finite, never-negative-zero operands it
                    // returns the same bits as max(): one v_max3_f32
                    const float best = fmaxf(fmaxf(mm, g1), g2);
                    if (EDGE) P[d] = (j >= 0 && j < Li) ? best : (j == Li ? lastcol : GM_NEG_INF);
                    else P[d] = best;
                }
                // slide the window bases to row i-1
#pragma unroll
                for (int d = 6; d >= 1; --d) { S1[d] = S1[d - 1]; S2[d] = S2[d - 1]; }
                set_col(0, (!EDGE || i - 4 >= 0) ? wcode(i - 4) : 0u);
            };
            {
                int i = Li - 1;
                const int mid_hi = Li - 5, mid_lo = 4;                          // rows [mid_lo, mid_hi] are interior (also wcode(i - 4) is in range)
                for (; i >= 0 && i > mid_hi; --i) dp_row(i, std::true_type{});
                for (; i >= mid_lo; --i) dp_row(i, std::false_type{});
                for (; i >= 0; --i) dp_row(i, std::true_type{});
            }
            // cells inside the band (counter only): sum over rows of min(i+3, L-1) - max(i-3, 0) + 1
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

// candidates the one-wave vote kernels left in their own slots -> the shards the DP kernel reads: one thread per read x strand, the
// wave's candidates are counted by a scan and reserved with ONE atomic (64 read x strands per atomic instead of one each, and no
// vote wave waits for it).  A wave's candidates stay together: 64 consecutive candidates belong to ~60 consecutive reads, so the
// DP kernel's lanes and the hit scatter touch neighbouring rows.
__global__ void __launch_bounds__(256) k_cand_gather(GmDevBatch b) {
    const uint32_t rs = blockIdx.x * 256 + threadIdx.x;
    const int lane = gm_lane();
    uint32_t c = 0;
    if (rs < 2 * b.n) {
        if (b.fixed_epoch) { const GmCand c0 = b.fixed_cands[GM_FIXED_AT(b, rs, 0u)]; c = __float_as_uint(c0.score) == b.fixed_epoch ? min((uint32_t)c0.pad, (uint32_t)GM_FIXED_C) : 0u; }      // k_vote_bucket: count + launch stamp in slot 0
        else c = b.fixed_cnt[rs];
    }
    const uint32_t incl = gm_wave_scan_incl(c);
    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0u) return;                             // wave-uniform
    const uint32_t shard = (rs >> 6) & (GM_NSHARD - 1);
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&b.shard_cnt[shard * GM_SHARD_STRIDE], total);
    base = __builtin_amdgcn_readfirstlane(base) + incl - c;
    for (uint32_t k = 0; k < c; ++k)
        if (base + k < b.cand_region) { GmCand cc = b.fixed_cands[GM_FIXED_AT(b, rs, k)]; cc.pad = 0; cc.score = 0.0f; b.cands[(size_t)shard * b.cand_region + base + k] = cc; }
}

// total and maximum of the candidate shards' fill counts, for the host's sizing decision: 8 bytes come back instead of the 128 KB the
// counters are spread over (one 128-byte line each)
__global__ void __launch_bounds__(256) k_shard_stats(GmDevBatch b, uint32_t* out /* {total, max} */) {
    __shared__ uint32_t s_sum[4], s_max[4];
    uint32_t sum = 0, mx = 0;
    unsigned long long dropped = 0;                       // word 1 of a shard: k-mers tried and dropped by the walks of k_vote_bucket (taken out as they are summed)
    for (uint32_t q = threadIdx.x; q < GM_NSHARD; q += 256) {
        const uint32_t c = b.shard_cnt[(size_t)q * GM_SHARD_STRIDE]; sum += c; mx = c > mx ? c : mx;
        const uint32_t d = b.shard_cnt[(size_t)q * GM_SHARD_STRIDE + 1]; if (d) { dropped += d; b.shard_cnt[(size_t)q * GM_SHARD_STRIDE + 1] = 0; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sum += __shfl_xor(sum, off); const uint32_t o = __shfl_xor(mx, off); mx = o > mx ? o : mx; dropped += __shfl_xor(dropped, off); }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_max[threadIdx.x >> 6] = mx; if (dropped) { atomicAdd(&b.counters[GMK_KMERS], dropped); atomicAdd(&b.counters[GMK_TAB_LOOKUPS], dropped); } }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        out[1] = std::max(std::max(s_max[0], s_max[1]), std::max(s_max[2], s_max[3]));
    }
}

__global__ void __launch_bounds__(256) k_scatter_hits(GmDevBatch b) {
    __shared__ uint32_t s_pre[GM_NSHARD + 4];
    const uint32_t n_cands = gm_cand_prefix(b, s_pre);
    for (uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x; wi < n_cands; wi += gridDim.x * blockDim.x) {
        GmCand c = b.cands[gm_cand_slot(b, s_pre, wi)];
        if (!(c.flags & GMC_ACCEPT)) continue;
        uint32_t r = c.rs >> 1;
        uint64_t slot = b.hit_begin[r] + atomicAdd(&b.hit_cursor[r], 1u);
        if (slot < b.raw_cap) {
            GmRawHit h;
            h.read = b.read_base + r; h.pos = c.b; h.score = c.score; h.step = c.step; h.strand = (uint8_t)(c.rs & 1); h.pad = 0;
            b.raw_hits[slot] = h;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// traceback: forward banded DP (same 8-lane anti-diagonal sweep) recording one 2-bit move per cell in LDS,
// then lane 0 of the group walks the moves back from (L,L).
// ------------------------------------------------------------------------------------------------
// Operations are written 2 bits each (0 = M, 1 = I, 2 = D), operation k in word k / 32 at bits 2 (k % 32): 8-byte stores instead of
// one byte store per operation.  With `emit` the lane that walks the path also adds up the length of the run-length CIGAR text
// (bin_seq.cpp:578-698; one trailing run of D stripped, SequenceOperations.h:32-42) for the matches that will be printed, and the
// longest aligned length of the launch (the grid of the coverage deposit).
__device__ __forceinline__ uint32_t gm_digits(uint32_t v) { return v >= 10000 ? 5u : v >= 1000 ? 4u : v >= 100 ? 3u : v >= 10 ? 2u : 1u; }

__global__ void __launch_bounds__(256) k_traceback(GmDevIndex ix, GmDevParams p, GmDevBatch b, const GmCand* items, uint32_t n,
                                                   unsigned long long* ops, uint32_t ops_words, uint16_t* ops_len, uint32_t Lp, uint32_t mvw,
                                                   const uint8_t* emit, uint32_t* cig_cnt, uint32_t* max_span) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const uint32_t G = blockDim.x >> 3;
    GmNwLds S = gm_nw_lds(s_raw, Lp, G);
    uint32_t* mv_all = reinterpret_cast<uint32_t*>(s_raw + GM_NW_HDR + (size_t)G * Lp * 3);   // G groups x 7 x mvw words
    for (int q = threadIdx.x; q < 512; q += blockDim.x) S.lut[q] = p.lut[q];
    if (threadIdx.x < 16) S.sg[threadIdx.x] = p.S256[(size_t)("acgt"[threadIdx.x >> 2]) * 4 + (threadIdx.x & 3)];
    const bool lds_coff = ix.n_seqs + 1 <= GM_NW_NCOFF;
    if (lds_coff) for (uint32_t q = threadIdx.x; q <= ix.n_seqs; q += blockDim.x) S.coff[q] = ix.contig_off[q];
    const uint32_t* coff = lds_coff ? S.coff : ix.contig_off;
    __syncthreads();
    const int lane = gm_lane();
    const int g = (threadIdx.x >> 3), d = lane & 7, delta = d - 3;
    uint16_t* rows = S.rows + (size_t)g * Lp;
    uint8_t* win = S.win + (size_t)g * Lp;
    uint32_t* mvg = mv_all + (size_t)g * 7 * mvw;
    uint32_t* mv = mvg + (size_t)(d < 7 ? d : 0) * mvw;
    const float gap = p.gap, gap4 = __fmul_rn(p.gap, 4.0f);
    for (uint32_t base = blockIdx.x * G; base < n; base += gridDim.x * G) {
        uint32_t ci = base + g;
        bool have = ci < n;
        GmCand c; c.rs = 0; c.b = 0;
        if (have) c = items[ci];
        uint32_t r = c.rs >> 1, strand = c.rs & 1;
        uint32_t L = have ? b.len[r] : 0;
        bool ok = have && L > 0 && gm_window_ok(ix, coff, c.b, L) && 2 * L <= 32 * ops_words;
        if (ok) gm_stage(ix, b, rows, win, r, strand, L, c.b, d);
        if (d < 7) for (uint32_t q = 0; q < mvw; ++q) mv[q] = 0;
        __syncthreads();
        const float2* lut = S.lut + ((r < b.illumina_until) ? 256 : 0);
        int absd = delta < 0 ? -delta : delta;
        float own = __fmul_rn(gap, (float)absd);          // first row / column: gGAP * t (bin_seq.cpp:503-511)
        int Lw = (int)L;
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) { int o = __shfl_xor(Lw, off); Lw = o > Lw ? o : Lw; }
        // lane delta walks its diagonal from the top-left: cells (i, i+delta), i = imin .. imax, due at s = 2i+delta
        const int imin = delta < 0 ? 1 - delta : 1;
        const int imax = (int)L - (delta > 0 ? delta : 0);
        int icur = (ok && d < 7) ? imin : 0x7fffffff;
        float vcur = 0.0f;
        if (icur <= imax) {
            uint32_t row = rows[icur - 1];
            float2 pq = lut[row & 255u];
            vcur = gm_get_val(row >> 8, pq.x, pq.y, S.sg + 4 * win[icur + delta - 1]);
        }
        for (int s = 2; s <= 2 * Lw; ++s) {
            float u_nb = gm_from_next_lane(own);          // diagonal delta+1, cell (i-1, j)
            float l_nb = gm_from_prev_lane(own);          // diagonal delta-1, cell (i, j-1)
            const bool act = icur <= imax && s == 2 * icur + delta;
            if (d == 6) u_nb = (icur - 1 == 0) ? gap4 : GM_NEG_INF;
            if (d == 0) l_nb = (icur + delta - 1 == 0) ? gap4 : GM_NEG_INF;
            if (act) {
                float dd = __fadd_rn(own, vcur);
                float u = __fadd_rn(u_nb, gap);
                float l = __fadd_rn(l_nb, gap);
                uint32_t m; float best;                   // max_flt(char&,...) src/bin_seq.cpp:989-1011
                if (dd >= u) { if (dd >= l) { m = 0; best = dd; } else { m = 2; best = l; } }
                else         { if (u >= l)  { m = 1; best = u; }  else { m = 2; best = l; } }
                own = best;
                mv[icur >> 4] |= m << ((icur & 15) << 1);
                ++icur;
                if (icur <= imax) {
                    uint32_t row = rows[icur - 1];
                    float2 pq = lut[row & 255u];
                    vcur = gm_get_val(row >> 8, pq.x, pq.y, S.sg + 4 * win[icur + delta - 1]);
                }
            }
        }
        __syncthreads();
        uint32_t my_span = 0;
        if (have && d == 0) {
            uint16_t outlen = 0;
            uint32_t ctext = 1;                                       // "*" when there is no path (ScoredSeq.h:318-324)
            if (ok) {
                unsigned long long* out = ops + (size_t)ci * ops_words;
                // pass 1: path length
                int i = (int)L, j = (int)L, nops = 0; bool bad = false;
                while (i != 0 && j != 0) {
                    int dl = j - i;
                    if (dl < -3 || dl > 3) { bad = true; break; }
                    uint32_t m = (mvg[(size_t)(dl + 3) * mvw + (i >> 4)] >> ((i & 15) << 1)) & 3u;
                    if (m == 0) { --i; --j; } else if (m == 1) { --i; } else { --j; }
                    ++nops;
                }
                if (!bad) {
                    nops += i + j;
                    // pass 2: the operations from the last to the first
                    int k = nops - 1;
                    unsigned long long cur = 0;
                    uint32_t run_code = 3, run_len = 0, text = 0; bool first_run = true;
                    auto push = [&](uint32_t code) {
                        cur |= (unsigned long long)code << (2 * (k & 31));
                        if ((k & 31) == 0) { out[k >> 5] = cur; cur = 0; }
                        --k;
                        if (code == run_code) { ++run_len; return; }
                        if (run_len) { if (!(first_run && run_code == 2)) text += gm_digits(run_len) + 1; first_run = false; }
                        run_code = code; run_len = 1;
                    };
                    i = (int)L; j = (int)L;
                    while (i != 0 && j != 0) {
                        int dl = j - i;
                        uint32_t m = (mvg[(size_t)(dl + 3) * mvw + (i >> 4)] >> ((i & 15) << 1)) & 3u;
                        push(m);
                        if (m == 0) { --i; --j; } else if (m == 1) { --i; } else { --j; }
                    }
                    while (i > 0) { push(1); --i; }
                    while (j > 0) { push(2); --j; }
                    if (run_len && !(first_run && run_code == 2)) text += gm_digits(run_len) + 1;
                    outlen = (uint16_t)nops;
                    if (nops) ctext = text;
                }
            }
            ops_len[ci] = outlen;
            my_span = outlen;
            if (cig_cnt) cig_cnt[ci] = (emit && !emit[ci]) ? 0u : (p.nw ? ctext : gm_digits(L) + 1u) + 1u;       // emit == null: the length for every item       // --no_nw prints "<L>M" (ScoredSeq.h:330-340)
        }
        if (max_span) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(my_span, off); my_span = o > my_span ? o : my_span; }
            if (lane == 0 && my_span) atomicMax(max_span, my_span);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// traceback, lane form (reads up to 511 bases): ONE lane per kept sequence runs the forward banded DP with the 7-cell band row in
// registers (the scheme of k_nw_lane) and keeps the 14 move bits of every row in LDS as [row][lane] 16-bit words (conflict-free: a
// wave's row is 128 consecutive bytes); the same lane then walks the moves back from (L, L) and emits the packed operations, the
// CIGAR text length and the aligned length exactly like k_traceback (which stays for longer reads).  Same arithmetic, same
// tie-breaks (max_flt(char&,...) src/bin_seq.cpp:989-1011), so the operations are identical.
// ------------------------------------------------------------------------------------------------
#define GM_TBV_ENTRIES (5u * 96u)            // k_traceback_lane's value table: codes 0..4 x quality characters 32..127
template <int NT>
__global__ void __launch_bounds__(NT) k_traceback_lane(GmDevIndex ix, GmDevParams p, GmDevBatch b, const GmCand* items, uint32_t n,
                                                       unsigned long long* ops, uint32_t ops_words, uint16_t* ops_len, uint32_t Lp,
                                                       const uint8_t* emit, uint32_t* cig_cnt, uint32_t* max_span, uint32_t ntab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    uint16_t* s_mv = reinterpret_cast<uint16_t*>(s_dyn);                 // (Lp + 1) rows x NT lanes
    // the four substitution values of a PWM row depend only on (called base, quality character): per-workgroup table built once with the
    // reference's expression, as in k_nw_rows - [phred table][code 0..4][quality character 32..127] -> {val(a), val(c), val(g), val(t)}
    float4* const s_val = reinterpret_cast<float4*>(s_dyn + (((size_t)(Lp + 1) * NT * 2 + 15) & ~(size_t)15));
    __shared__ uint32_t s_coff[GM_NW_NCOFF];
    const bool lds_coff = ix.n_seqs + 1 <= GM_NW_NCOFF;
    if (lds_coff) for (uint32_t q = threadIdx.x; q <= ix.n_seqs; q += NT) s_coff[q] = ix.contig_off[q];
    const uint32_t* coff = lds_coff ? s_coff : ix.contig_off;
    __syncthreads();
    float sg[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) sg[g][k] = p.S256[(size_t)("acgt"[g]) * 4 + k];
    for (uint32_t e = threadIdx.x; e < ntab * GM_TBV_ENTRIES; e += NT) {
        const uint32_t tab = e / GM_TBV_ENTRIES, code = (e % GM_TBV_ENTRIES) / 96u, q = 32u + e % 96u;
        const float2 pq = p.lut[tab * 256u + q];
        s_val[e] = make_float4(gm_get_val(code, pq.x, pq.y, sg[0]), gm_get_val(code, pq.x, pq.y, sg[1]), gm_get_val(code, pq.x, pq.y, sg[2]), gm_get_val(code, pq.x, pq.y, sg[3]));
    }
    __syncthreads();
    const float gap = p.gap, gap4 = __fmul_rn(p.gap, 4.0f);
    const uint32_t* pac32 = reinterpret_cast<const uint32_t*>(ix.pac);
    const int lane = gm_lane();
    uint16_t* mv = s_mv + threadIdx.x;
    for (uint32_t base = blockIdx.x * NT; base < n; base += gridDim.x * NT) {
        const uint32_t ci = base + threadIdx.x;
        const bool have = ci < n;
        GmCand c; c.rs = 0; c.b = 0;
        if (have) c = items[ci];
        const uint32_t r = c.rs >> 1, strand = c.rs & 1;
        const uint32_t L = have ? b.len[r] : 0;
        const bool ok = have && L > 0 && L <= Lp && gm_window_ok(ix, coff, c.b, L) && 2 * L <= 32 * ops_words;
        uint16_t outlen = 0;
        uint32_t ctext = 1;                                              // "*" when there is no path
        if (ok) {
            const uint32_t tab = (r < b.illumina_until) ? 1u : 0u;
            const float2* lut = p.lut + tab * 256u;
            const float4* const vtab = s_val + (tab < ntab ? tab : 0u) * GM_TBV_ENTRIES;
            const uint8_t* rb = b.bases + (size_t)r * b.stride;
            const uint8_t* rq = b.quals + (size_t)r * b.stride;
            const int Li = (int)L;
            uint32_t wword_lo = 0, wword_hi = 0; int wbase = -1;
            auto wcode = [&](int j) -> uint32_t {                        // 2-bit code of the reference at window offset j
                uint32_t g = c.b + (uint32_t)j;
                int wi16 = (int)(g >> 4);
                if (wi16 != wbase && wi16 != wbase + 1) { wbase = wi16; wword_lo = pac32[wi16]; wword_hi = pac32[wi16 + 1]; }
                uint32_t word = wi16 == wbase ? wword_lo : wword_hi;
                return (word >> ((((g >> 2) & 3u) << 3) + ((~g & 3u) << 1))) & 3u;
            };
            // row 0: nm[0][j] = gGAP * j for the band's columns (bin_seq.cpp:503-511)
            float P[7], C[7];
#pragma unroll
            for (int d = 0; d < 7; ++d) P[d] = d >= 3 ? __fmul_rn(gap, (float)(d - 3)) : GM_NEG_INF;
            // window codes of the band columns of row i: W[d] = w[i + d - 4] (column j - 1 of cell (i, j = i + d - 3)); start at row 1
            uint32_t W[7];
#pragma unroll
            for (int d = 0; d < 7; ++d) { int j = 1 + d - 4; W[d] = (j >= 0 && j < Li) ? wcode(j) : 0u; }
            // the read streams through 8-byte words, the next word requested one chunk ahead
            const int src0 = strand ? Li - 1 : 0, cstep = strand ? -1 : 1, nchunk = (Li + 7) >> 3;
            int chunk = src0 >> 3;
            uint2 bw = *reinterpret_cast<const uint2*>(rb + (chunk << 3)), qw = *reinterpret_cast<const uint2*>(rq + (chunk << 3));
            uint2 bn = bw, qn = qw;
            { int nc = chunk + cstep; if (nc >= 0 && nc < nchunk) { bn = *reinterpret_cast<const uint2*>(rb + (nc << 3)); qn = *reinterpret_cast<const uint2*>(rq + (nc << 3)); } }
            for (int i = 1; i <= Li; ++i) {
                const int src = strand ? Li - i : i - 1;                 // PWM row i-1 in strand orientation (reverse_comp_cpy)
                if ((src >> 3) != chunk) {
                    chunk = src >> 3;
                    bw = bn; qw = qn;
                    int nc = chunk + cstep;
                    if (nc >= 0 && nc < nchunk) { bn = *reinterpret_cast<const uint2*>(rb + (nc << 3)); qn = *reinterpret_cast<const uint2*>(rq + (nc << 3)); }
                }
                const uint32_t sh = (uint32_t)(src & 3) << 3;
                const uint32_t ch = (((src & 4) ? bw.y : bw.x) >> sh) & 255u;
                const uint32_t qc = (((src & 4) ? qw.y : qw.x) >> sh) & 255u;
                uint32_t code = gm_nt4(ch);
                if (strand && code < 4) code = 3 - code;
                float v4[4];
                if (tab < ntab && qc - 32u < 96u) {
                    const float4 t4 = vtab[code * 96u + qc - 32u];
                    v4[0] = t4.x; v4[1] = t4.y; v4[2] = t4.z; v4[3] = t4.w;
                } else {                                                 // a character outside the table (or no table): the direct form
                    const float2 pq = lut[qc];
#pragma unroll
                    for (int g = 0; g < 4; ++g) v4[g] = gm_get_val(code, pq.x, pq.y, sg[g]);
                }
                uint32_t mrow = 0;
#pragma unroll
                for (int d = 0; d < 7; ++d) {
                    const int j = i + d - 3;
                    if (j <= 0 || j > Li) { C[d] = j == 0 ? __fmul_rn(gap, (float)i) : GM_NEG_INF; continue; }     // first column: gGAP * i
                    const uint32_t wc = W[d];
                    const float val = (wc & 2u) ? ((wc & 1u) ? v4[3] : v4[2]) : ((wc & 1u) ? v4[1] : v4[0]);
                    const float up = d < 6 ? P[d + 1] : (i - 1 == 0 ? gap4 : GM_NEG_INF);        // nm[i-1][j]
                    const float left = d > 0 ? C[d - 1] : (j - 1 == 0 ? gap4 : GM_NEG_INF);       // nm[i][j-1]
                    const float dd = __fadd_rn(P[d], val);
                    const float u = __fadd_rn(up, gap);
                    const float l = __fadd_rn(left, gap);
                    uint32_t m; float best;                              // max_flt(char&,...) src/bin_seq.cpp:989-1011
                    if (dd >= u) { if (dd >= l) { m = 0; best = dd; } else { m = 2; best = l; } }
                    else         { if (u >= l)  { m = 1; best = u; }  else { m = 2; best = l; } }
                    C[d] = best;
                    mrow |= m << (2 * d);
                }
#pragma unroll
                for (int d = 0; d < 7; ++d) P[d] = C[d];
                mv[(size_t)i * NT] = (uint16_t)mrow;
                // slide the window codes to row i + 1: column offsets (i + 1) + d - 4
#pragma unroll
                for (int d = 0; d < 6; ++d) W[d] = W[d + 1];
                { const int j = i + 3; W[6] = j < Li ? wcode(j) : 0u; }
            }
            // walk back from (L, L)
            unsigned long long* out = ops + (size_t)ci * ops_words;
            int i = Li, j = Li, nops = 0; bool bad = false;
            while (i != 0 && j != 0) {
                const int dl = j - i;
                if (dl < -3 || dl > 3) { bad = true; break; }
                const uint32_t m = ((uint32_t)mv[(size_t)i * NT] >> (2 * (dl + 3))) & 3u;
                if (m == 0) { --i; --j; } else if (m == 1) { --i; } else { --j; }
                ++nops;
            }
            if (!bad) {
                nops += i + j;
                int k = nops - 1;
                unsigned long long cur = 0;
                uint32_t run_code = 3, run_len = 0, text = 0; bool first_run = true;
                auto push = [&](uint32_t code) {
                    cur |= (unsigned long long)code << (2 * (k & 31));
                    if ((k & 31) == 0) { out[k >> 5] = cur; cur = 0; }
                    --k;
                    if (code == run_code) { ++run_len; return; }
                    if (run_len) { if (!(first_run && run_code == 2)) text += gm_digits(run_len) + 1; first_run = false; }
                    run_code = code; run_len = 1;
                };
                i = Li; j = Li;
                while (i != 0 && j != 0) {
                    const int dl = j - i;
                    const uint32_t m = ((uint32_t)mv[(size_t)i * NT] >> (2 * (dl + 3))) & 3u;
                    push(m);
                    if (m == 0) { --i; --j; } else if (m == 1) { --i; } else { --j; }
                }
                while (i > 0) { push(1); --i; }
                while (j > 0) { push(2); --j; }
                if (run_len && !(first_run && run_code == 2)) text += gm_digits(run_len) + 1;
                outlen = (uint16_t)nops;
                if (nops) ctext = text;
            }
        }
        if (have) {
            ops_len[ci] = outlen;
            if (cig_cnt) cig_cnt[ci] = (emit && !emit[ci]) ? 0u : (p.nw ? ctext : gm_digits(L) + 1u) + 1u;       // emit == null: the length for every item
        }
        if (max_span) {
            uint32_t my_span = outlen;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(my_span, off); my_span = o > my_span ? o : my_span; }
            if (lane == 0 && my_span) atomicMax(max_span, my_span);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// coverage: one thread per deposited base
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_coverage_add(float* cov, uint64_t bins, uint32_t bin_size, const uint64_t* pos,
                                                      const uint32_t* span, const float* w, uint32_t n, uint32_t max_span,
                                                      float* nuc, const uint8_t* codes, const uint64_t* code_off) {
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t dep = (uint32_t)(gid / max_span), t = (uint32_t)(gid % max_span);
    if (dep >= n || t >= span[dep]) return;
    uint64_t bin = (pos[dep] + t) / bin_size;
    if (bin >= bins) return;
    atomicAdd(&cov[bin], w[dep]);
    if (nuc) {                                       // -b / -d: reads[base][loc] += w (GenomeBwt::AddSeqScore src/GenomeBwt.cpp:556-603)
        uint32_t c = codes[code_off[dep] + t];
        if (c < 5) atomicAdd(&nuc[(size_t)c * bins + bin], w[dep]);
    }
}

// ------------------------------------------------------------------------------------------------
// unit-level kernels (parity tests of the building blocks)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_sa_interval(GmDevIndex ix, const uint8_t* kmers, uint32_t n, uint32_t m, uint32_t* start, uint32_t* end) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const uint8_t* km = kmers + (size_t)q * m;
    uint32_t k = 0, l = ix.seq_len;
    bool ok = true;
    for (int t = (int)m - 1; t >= 0; --t) {
        uint32_t c = gm_nt4(km[t]);
        if (c > 3) { ok = false; break; }
        uint32_t ok_ = gm_occ_plane(ix, k - 1, c), ol_ = gm_occ_plane(ix, l, c);
        k = gm_L2(ix, c) + ok_ + 1;
        l = gm_L2(ix, c) + ol_;
        if (k > l) { ok = false; break; }
    }
    start[q] = ok ? k : 0;
    end[q] = ok ? l : 0;
}

__global__ void __launch_bounds__(256) k_locate(GmDevIndex ix, const uint32_t* ranks, uint32_t n, int use_full_sa, uint32_t* out) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    uint32_t st;
    out[q] = use_full_sa ? ix.full_sa[ranks[q]] : gm_locate_walk(ix, ranks[q], &st);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// grid of a persistent kernel: exactly the workgroups the device keeps resident (CUs x workgroups per CU for this kernel and
// its dynamic LDS).  A larger static grid makes the surplus workgroups run as a second, half-empty round.
template <class K>
static uint32_t resident_grid(K kernel, int threads, size_t dyn_lds, uint32_t fallback) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fallback;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return fallback;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, dyn_lds) != hipSuccess || per_cu <= 0) return fallback;
    return (uint32_t)cus * (uint32_t)per_cu;
}


int gmk_expand_full_sa(const GmDevIndex& ix, uint32_t* full_sa, void* stream) {
    uint32_t n_sa = (uint32_t)(((uint64_t)ix.seq_len + ix.sa_mask + 1) >> ix.sa_shift);
    hipLaunchKernelGGL(k_expand_full_sa, dim3(cdiv(n_sa, 256)), dim3(256), 0, S_(stream), ix, full_sa, n_sa);
    return (int)hipGetLastError();
}

int gmk_build_occ_planes(const GmDevIndex& ix, uint4* planes, uint32_t nblk, void* stream) {
    hipLaunchKernelGGL(k_build_occ_planes, dim3(cdiv(nblk, 256)), dim3(256), 0, S_(stream), ix, planes, nblk);
    return (int)hipGetLastError();
}

int gmk_build_kmer_table(const GmDevIndex& ix, uint2* tab, int T, void* stream) {
    const uint64_t n = 1ull << (2 * T);
    hipLaunchKernelGGL(k_build_kmer_table, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 1u << 22)), dim3(256), 0, S_(stream), ix, tab, T);
    return (int)hipGetLastError();
}

int gmk_extend_kmer_table(const GmDevIndex& ix, const uint2* prev, uint2* next, int T, void* stream) {
    const uint64_t n = 1ull << (2 * T);
    hipLaunchKernelGGL(k_extend_kmer_table, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 1u << 22)), dim3(256), 0, S_(stream), ix, prev, next, T);
    return (int)hipGetLastError();
}

int gmk_build_kmer_compact(const uint2* tab, uint4* ctab, int T, void* stream) {
    const uint64_t n = 1ull << (2 * T - 3);
    hipLaunchKernelGGL(k_build_kmer_compact, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 1u << 22)), dim3(256), 0, S_(stream), tab, ctab, T);
    return (int)hipGetLastError();
}

int gmk_prep(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, void* stream) {
    if (b.n == 0) return 0;
    // rows of up to 152 bytes: the form with the row in registers (gm_prep.hip); GM_PREP=tile keeps the LDS-tile form for A/B runs
    if (b.stride <= 152 && !gm_opt_is("GM_PREP", "tile")) return gmk_prep_rows(ix, p, b, stream);
    // reads per LDS tile: 48 KB for bases + quals, whole waves; rows too long for at least one wave per tile are read directly
    uint32_t tr = (uint32_t)(49152 / (2 * (size_t)b.stride));
    tr = tr >= 256 ? 256 : (tr / 64) * 64;
    const uint32_t per = tr ? tr : 256;
    const uint32_t pg = resident_grid(k_prep, 256, (size_t)2 * tr * b.stride, 256 * 3);
    hipLaunchKernelGGL(k_prep, dim3((uint32_t)std::min<uint64_t>(cdiv(b.n, per), pg)), dim3(256), (size_t)2 * tr * b.stride, S_(stream), ix, p, b, tr);
    return (int)hipGetLastError();
}

int gmk_seed(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, void* stream) {
    if (b.n == 0) return 0;
    // (measured: a grid of exactly the resident workgroups is slower here, 8.7 against 7.6 ms at 10 M reads)
    const uint32_t seed_grid = (uint32_t)gm_opt_ll("GM_SEED_GRID", 256 * 12);
    // reads per tile: 128 (every lane of the 256 holds a read x strand) while the tile fits 64 KB of LDS, fewer whole waves for long reads
    uint32_t TR = 128;
    while (TR > 32 && (size_t)TR * b.stride > 64u * 1024u) TR -= 32;
    if ((size_t)TR * b.stride > 64u * 1024u) return (int)hipErrorInvalidValue;     // stride > 2048: gm_batch_upload refuses it
    hipLaunchKernelGGL(k_seed, dim3((uint32_t)std::min<uint64_t>(cdiv(b.n, TR), seed_grid)), dim3(256), (size_t)TR * b.stride, S_(stream), ix, p, b, TR);
    return (int)hipGetLastError();
}

static int scan_u32(const uint32_t* in, uint64_t n, uint64_t* out, unsigned long long* tmp, void* stream) {
    uint32_t nb = cdiv(n, 1024);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(256), 0, S_(stream), in, n, tmp);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, S_(stream), tmp, nb);
    hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(256), 0, S_(stream), in, n, tmp, out);
    return (int)hipGetLastError();
}

int gmk_scan_u32(const uint32_t* in, uint64_t n, uint64_t* out, unsigned long long* tmp, void* stream) {
    if (n == 0) return (int)hipMemsetAsync(out, 0, 8, S_(stream));
    return scan_u32(in, n, out, tmp, stream);
}

int gmk_scan_entries(const GmDevBatch& b, void* stream) {
    if (b.n == 0) return 0;
    // scan scratch: the tail of hit_begin's sibling buffer is not available yet, so the host gives us retry_off as scratch
    return scan_u32(b.n_entries, 2ull * b.n, b.entry_off, reinterpret_cast<unsigned long long*>(b.retry_off), stream);
}

int gmk_locate_sampled(const GmDevIndex& ix, const GmDevBatch& b, unsigned long long n_entries, void* stream) {
    if (b.n == 0) return 0;
    if (gm_opt_is("GM_LOCATE", "wave")) {           // the first form: a wavefront per read x strand, its lanes on one seed's ranks at a time
        hipLaunchKernelGGL(k_locate_sampled, dim3(cdiv(2ull * b.n, 4)), dim3(256), 0, S_(stream), ix, b);
        return (int)hipGetLastError();
    }
    if (n_entries == 0) return 0;
    hipLaunchKernelGGL(k_locate_ranks, dim3(cdiv(2ull * b.n, 4)), dim3(256), 0, S_(stream), b);
    // every lane walks: 8 workgroups of 256 lanes per CU resident, each lane taking every (grid x 256)-th hit of the list
    const uint32_t grid = (uint32_t)std::min<unsigned long long>(cdiv(n_entries, 256ull), (unsigned long long)gm_opt_ll("GM_LOCATE_GRID", 256 * 8));
    hipLaunchKernelGGL(k_locate_flat, dim3(grid), dim3(256), 0, S_(stream), ix, b, n_entries);
    return (int)hipGetLastError();
}

int gmk_vote(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, int use_full_sa, int dense, int slots_hint, void* stream) {
    if (b.n == 0) return 0;
    // dense seeds (many SA hits per read x strand): one workgroup per read x strand
    if (dense && b.max_seeds <= 64) {
        const int nt = [] { const int v = (int)gm_opt_ll("GM_VOTE_NT", 128); return v == 64 || v == 256 ? v : 128; }();
        const bool m64 = b.max_seeds > 32;
        const char* kenv = gm_opt("GM_VOTE_KERNEL"); if (kenv && !*kenv) kenv = nullptr;
        const bool slots_form = kenv ? !strcmp(kenv, "slots") : dense <= 2;      // dense == 2: the 64-slot form; 3: rounds of the block form
        if (slots_form) {                           // default: wave-uniform seed slots + the list kernel for what it hands over
            const int slot_form = dense == 2 ? 64 : slots_hint == 0 ? 0 : slots_hint < 0 ? -1 : slots_hint <= 14 ? 16 : slots_hint <= 22 ? 24 : GMS_SMAX;      // 0 = k_vote_tiny, -1 = k_vote_tiny2
            const uint32_t lgrid = (uint32_t)std::min<uint64_t>(cdiv(2ull * b.n, 4), 256 * 20);
            // the fused form as persistent workgroups with the next read x strands' loads in flight (k_vote_slots_pp): `pp_chunk` consecutive read x
            // strands per workgroup - the pipeline's fill (three round trips) against ~64 votes.  Default for the 64-slot form only (its 20 KB of LDS
            // leave 8 workgroups per CU to hide the three round trips: 133.9 -> 112.8 ms at the chrX shape); the smaller forms have 13 - 16
            // workgroups per CU and lose them to the pipelined form's registers (40 slots: 59.9 against 68.1 ms at configs[1]).
            // GM_SLOTS_PIPE=0 / 1: never / every form.  GM_DBG=64 (phase clocks) runs k_vote_slots.
            const int pp_env = (int)gm_opt_ll("GM_SLOTS_PIPE", -1);
            const uint32_t pp_chunk = ((pp_env < 0 ? slot_form == 64 : pp_env != 0) && !(p.dbg & 64)) ? (uint32_t)std::max<long long>(2, gm_opt_ll("GM_SLOTS_CHUNK", 64)) : 0u;
#define GM_LAUNCH_VSL1(M, F, S) do { if (F && p.fused && pp_chunk) hipLaunchKernelGGL((k_vote_slots_pp<M, S, true>), dim3((uint32_t)cdiv(2ull * b.n, pp_chunk)), dim3(128), 0, S_(stream), ix, p, b, pp_chunk); \
                                     else if (F && pp_chunk) hipLaunchKernelGGL((k_vote_slots_pp<M, S, false>), dim3((uint32_t)cdiv(2ull * b.n, pp_chunk)), dim3(128), 0, S_(stream), ix, p, b, pp_chunk); \
                                     else if (F && p.fused) hipLaunchKernelGGL((k_vote_slots<M, true, S, true>), dim3(2 * b.n), dim3(128), 0, S_(stream), ix, p, b); \
                                     else hipLaunchKernelGGL((k_vote_slots<M, F, S, false>), dim3(2 * b.n), dim3(128), 0, S_(stream), ix, p, b); } while (0)
#define GM_LAUNCH_VSL(M, F) do { if (slot_form == 0 && F && p.fused) hipLaunchKernelGGL((k_vote_tiny<M, true, true>), dim3(2 * b.n), dim3(64), 0, S_(stream), ix, p, b); \
                                 else if (slot_form == -1 && F && p.fused) hipLaunchKernelGGL((k_vote_tiny2<M, true, true>), dim3(2 * b.n), dim3(64), 0, S_(stream), ix, p, b); \
                                 else if (slot_form == 0) hipLaunchKernelGGL((k_vote_tiny<M, F, false>), dim3(2 * b.n), dim3(64), 0, S_(stream), ix, p, b); \
                                 else if (slot_form == -1) hipLaunchKernelGGL((k_vote_tiny2<M, F, false>), dim3(2 * b.n), dim3(64), 0, S_(stream), ix, p, b); \
                                 else if (slot_form == 64) GM_LAUNCH_VSL1(M, F, 64); else if (slot_form == 16) GM_LAUNCH_VSL1(M, F, 16); \
                                 else if (slot_form == 24) GM_LAUNCH_VSL1(M, F, 24); else GM_LAUNCH_VSL1(M, F, GMS_SMAX); } while (0)
            if (m64) { if (use_full_sa) GM_LAUNCH_VSL(true, true); else GM_LAUNCH_VSL(true, false); }
            else { if (use_full_sa) GM_LAUNCH_VSL(false, true); else GM_LAUNCH_VSL(false, false); }
#undef GM_LAUNCH_VSL1
#undef GM_LAUNCH_VSL
            if (m64) hipLaunchKernelGGL(k_vote_fast_list<true>, dim3(lgrid), dim3(256), 0, S_(stream), ix, p, b, use_full_sa);
            else hipLaunchKernelGGL(k_vote_fast_list<false>, dim3(lgrid), dim3(256), 0, S_(stream), ix, p, b, use_full_sa);
            return (int)hipGetLastError();
        }
        // one-round exact table: 512 slots (13 KB of LDS per workgroup, 12 workgroups per CU) or 1024 (17 KB, 9 per CU)
        const int tb = gm_opt_ll("GM_VOTE_TB", 9) == 10 ? 10 : 9;
#define GM_LAUNCH_VB(M, N, T) hipLaunchKernelGGL((k_vote_block<M, N, T>), dim3(2 * b.n), dim3(N), 0, S_(stream), ix, p, b, use_full_sa)
#define GM_LAUNCH_VB2(M, N) do { if (tb == 10) GM_LAUNCH_VB(M, N, 10); else GM_LAUNCH_VB(M, N, 9); } while (0)
        if (nt == 64) { if (m64) GM_LAUNCH_VB2(true, 64); else GM_LAUNCH_VB2(false, 64); }
        else if (nt == 256) { if (m64) GM_LAUNCH_VB2(true, 256); else GM_LAUNCH_VB2(false, 256); }
        else { if (m64) GM_LAUNCH_VB2(true, 128); else GM_LAUNCH_VB2(false, 128); }
#undef GM_LAUNCH_VB2
#undef GM_LAUNCH_VB
        return (int)hipGetLastError();
    }
    // sparse seeds: sub-wave groups first; what does not fit 16 hits goes to the wave-per-read x strand kernel through a list
    const bool sparse_form = !gm_opt_is("GM_VOTE_SPARSE", "0");
    if (sparse_form && b.max_seeds <= 64) {
        hipLaunchKernelGGL(k_vote_sparse, dim3(cdiv(2ull * b.n, 16)), dim3(256), 0, S_(stream), ix, p, b, use_full_sa);
        const uint32_t grid = (uint32_t)std::min<uint64_t>(cdiv(2ull * b.n, 4), 256 * 20);
        if (b.max_seeds <= 32) hipLaunchKernelGGL(k_vote_fast_list<false>, dim3(grid), dim3(256), 0, S_(stream), ix, p, b, use_full_sa);
        else hipLaunchKernelGGL(k_vote_fast_list<true>, dim3(grid), dim3(256), 0, S_(stream), ix, p, b, use_full_sa);
        return (int)hipGetLastError();
    }
    // order-free fast path while the seed steps fit a 64-bit mask; the ordered kernel is the general form
    if (b.max_seeds <= 32) hipLaunchKernelGGL(k_vote_fast<false>, dim3(cdiv(2ull * b.n, 4)), dim3(256), 0, S_(stream), ix, p, b, use_full_sa);
    else if (b.max_seeds <= 64) hipLaunchKernelGGL(k_vote_fast<true>, dim3(cdiv(2ull * b.n, 4)), dim3(256), 0, S_(stream), ix, p, b, use_full_sa);
    else hipLaunchKernelGGL(k_vote, dim3(cdiv(2ull * b.n, 4)), dim3(256), 0, S_(stream), ix, p, b, use_full_sa);
    return (int)hipGetLastError();
}

// read x strands flagged for the list kernel (k_vote_bucket: b.fixed_cnt[rs] = 1) -> b.big_list: one thread per 16 flags (one 16-byte
// load, no loop: the flags are few and the kernel is all latency), one atomic per wavefront that has any
__global__ void __launch_bounds__(256) k_big_collect(GmDevBatch b) {
    const uint32_t n2 = 2u * b.n, q = blockIdx.x * 256u + threadIdx.x;
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (16u * q < n2) w = reinterpret_cast<const uint4*>(b.fixed_cnt)[q];                // (the buffer is padded beyond 2n and zeroed with it)
    const uint32_t any = (w.x | w.y | w.z | w.w) & 0x01010101u;
    if (__builtin_amdgcn_ballot_w64(any != 0u) == 0ull) return;                           // wave-uniform: nearly every wave
    const uint32_t ws[4] = { w.x & 0x01010101u, w.y & 0x01010101u, w.z & 0x01010101u, w.w & 0x01010101u };
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) c += (uint32_t)__popc(ws[k]);
    uint32_t left = 16u * q < n2 ? n2 - 16u * q : 0u;                                     // flags of this thread that are read x strands
    if (left < 16u) {                                                                      // the last thread: count only those
        c = 0;
        for (uint32_t t = 0; t < left; ++t) c += (ws[t >> 2] >> (8u * (t & 3u))) & 1u;
    } else left = 16u;
    const uint32_t incl = gm_wave_scan_incl(c);
    if (__builtin_amdgcn_readlane((int)incl, 63) == 0) return;                             // nothing flagged in this wavefront (the usual case): no atomic
    uint32_t at = 0;
    if (gm_lane() == 63) at = atomicAdd(b.n_big, incl);
    at = (uint32_t)__builtin_amdgcn_readlane((int)at, 63) + incl - c;
    for (uint32_t t = 0; t < left; ++t) if ((ws[t >> 2] >> (8u * (t & 3u))) & 1u) b.big_list[at++] = 16u * q + t;
}

int gmk_vote_list(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, int use_full_sa, void* stream) {
    if (b.n == 0) return 0;
    if (p.bucket != nullptr && b.fixed_cnt != nullptr)
        hipLaunchKernelGGL(k_big_collect, dim3((uint32_t)cdiv((2ull * b.n + 15) / 16, 256)), dim3(256), 0, S_(stream), b);
    const uint32_t lgrid = (uint32_t)std::min<uint64_t>(cdiv(2ull * b.n, 4), 256 * 20);
    if (b.max_seeds > 32) hipLaunchKernelGGL(k_vote_fast_list<true>, dim3(lgrid), dim3(256), 0, S_(stream), ix, p, b, use_full_sa);
    else hipLaunchKernelGGL(k_vote_fast_list<false>, dim3(lgrid), dim3(256), 0, S_(stream), ix, p, b, use_full_sa);
    return (int)hipGetLastError();
}

int gmk_shard_stats(const GmDevBatch& b, uint32_t* out, void* stream) {
    hipLaunchKernelGGL(k_shard_stats, dim3(1), dim3(256), 0, S_(stream), b, out);
    return (int)hipGetLastError();
}

int gmk_cand_gather(const GmDevBatch& b, void* stream) {
    if (b.n == 0 || !b.fixed_cands) return 0;
    hipLaunchKernelGGL(k_cand_gather, dim3(cdiv(2ull * b.n, 256)), dim3(256), 0, S_(stream), b);
    return (int)hipGetLastError();
}

int gmk_vote_retry(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, int use_full_sa, uint32_t j0, uint32_t n_retry, void* stream) {
    if (n_retry == 0) return 0;
    hipLaunchKernelGGL(k_vote_retry, dim3(n_retry), dim3(256), 0, S_(stream), ix, p, b, use_full_sa, j0, n_retry);
    return (int)hipGetLastError();
}

static inline uint32_t lp_of(uint32_t stride) { return (stride + 7u) & ~7u; }

// blocks of ONE read length (what a sequencer writes): the rows-in-DP-order kernel of gm_nw.hip.  GM_NW=lane keeps k_nw_lane
static bool nw_rows_ok(const GmDevParams& p, const GmDevBatch& b, uint32_t rows_len) {
    return p.max_gap == 3 && p.nw && !gm_opt_is("GM_NW", "wave") && !gm_opt_is("GM_NW", "lane") && rows_len >= 24 && rows_len <= 152 && rows_len <= b.stride;
}
const char* gmk_nw_form(const GmDevParams& p, const GmDevBatch& b, uint32_t n_cands, uint32_t rows_len) {
    (void)n_cands;
    if (p.max_gap != 3) return "k_nw_band";
    if (nw_rows_ok(p, b, rows_len)) return "k_nw_rows";
    return gm_opt_is("GM_NW", "wave") ? "k_nw" : "k_nw_lane";
}

int gmk_nw(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, uint32_t n_cands, uint32_t rows_len, void* stream) {
    if (b.n == 0) return 0;
    if (p.max_gap != 3) return gmk_nw_band(ix, p, b, n_cands, p.max_gap, stream);
    const bool wave_form = gm_opt_is("GM_NW", "wave");
    if (nw_rows_ok(p, b, rows_len)) return gmk_nw_rows(ix, p, b, n_cands, rows_len, stream);
    if (!wave_form) {
        // ~4 candidates per lane: fewer, larger workgroups leave a long tail (measured at 17 M candidates: 2048 workgroups 6.1 ms,
        // 16384 5.6 ms), more, smaller ones pay their set-up (LDS tables, shard prefix) too often (2 M candidates: 0.77 against 1.02 ms)
        const uint32_t nw_fixed = (uint32_t)gm_opt_ll("GM_NW_GRID", 0);
        uint32_t nw_grid = nw_fixed ? nw_fixed : std::min<uint32_t>(16384u, std::max<uint32_t>(2048u, n_cands / 1024u));
        // rows in registers while the candidates are sparse (about one per read: every candidate touches its own lines); with several
        // candidates per read (dense seeds) neighbouring lanes share the lines and the streaming form with its 5 waves per SIMD wins
        // (measured at configs[1], 4.3 candidates per read: 2.8 against 3.2 ms).  GM_NW_ROWS=0: streaming form always
        // GM_NW_ROWS=1 (default): the row waits in registers; 2: in LDS (measured: 6.7 against 4.7 ms - two workgroups per CU instead of
        // three cost more than the 13-way register pick it removes)
        const int rows_in_regs = (int)gm_opt_ll("GM_NW_ROWS", 1);
        const bool sparse = rows_in_regs && n_cands < 2.5 * b.n;
        // 53 / 78 KB of rows + 12 KB of tables: beyond the 64 KB a launch gets by default.  The attribute is per DEVICE (--gpus=N): set once
        // for each device this process launches the LDS-row form on
        bool big_lds = true;
        if (rows_in_regs == 2) {
            static std::atomic<unsigned long long> lds_set{ 0 };
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
            if (!((lds_set.load() >> dev) & 1ull)) {
                big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nw_lane<13, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 13 * 256 * 8) == hipSuccess &&
                          hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nw_lane<19, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 19 * 256 * 8) == hipSuccess;
                if (big_lds) lds_set.fetch_or(1ull << dev);
            }
        }
        if (!big_lds && rows_in_regs == 2) return (int)hipErrorInvalidValue;
        if (sparse && rows_in_regs == 2 && b.stride <= 104) hipLaunchKernelGGL((k_nw_lane<13, true>), dim3(nw_grid), dim3(256), (size_t)2 * 13 * 256 * 8, S_(stream), ix, p, b);
        else if (sparse && rows_in_regs == 2 && b.stride <= 152) hipLaunchKernelGGL((k_nw_lane<19, true>), dim3(nw_grid), dim3(256), (size_t)2 * 19 * 256 * 8, S_(stream), ix, p, b);
        else if (sparse && b.stride <= 104) hipLaunchKernelGGL((k_nw_lane<13, false>), dim3(nw_grid), dim3(256), 0, S_(stream), ix, p, b);
        else if (sparse && b.stride <= 152) hipLaunchKernelGGL((k_nw_lane<19, false>), dim3(nw_grid), dim3(256), 0, S_(stream), ix, p, b);
        else hipLaunchKernelGGL((k_nw_lane<0, false>), dim3(nw_grid), dim3(256), 0, S_(stream), ix, p, b);
        return (int)hipGetLastError();
    }
    uint32_t Lp = lp_of(b.stride);
    uint32_t threads = 256;                                  // 32 candidates per workgroup; long reads: 8, to stay inside LDS
    if (GM_NW_HDR + (size_t)32 * Lp * 3 > 60 * 1024) threads = 64;
    size_t lds = GM_NW_HDR + (size_t)(threads / 8) * Lp * 3;
    hipLaunchKernelGGL(k_nw, dim3(threads == 256 ? 2048 : 8192), dim3(threads), lds, S_(stream), ix, p, b, Lp);
    return (int)hipGetLastError();
}

int gmk_scan_hits(const GmDevBatch& b, void* stream) {
    if (b.n == 0) return 0;
    return scan_u32(b.hit_count, b.n, b.hit_begin, reinterpret_cast<unsigned long long*>(b.retry_off), stream);
}

int gmk_scatter(const GmDevBatch& b, uint32_t grid, void* stream) {
    if (b.n == 0) return 0;
    hipLaunchKernelGGL(k_scatter_hits, dim3(grid), dim3(256), 0, S_(stream), b);
    return (int)hipGetLastError();
}

int gmk_compact(const GmDevBatch& b, void* stream) {
    int e = gmk_scan_hits(b, stream);
    if (e) return e;
    return gmk_scatter(b, 2048, stream);
}

int gmk_sa_interval(const GmDevIndex& ix, const uint8_t* kmers, uint32_t n, uint32_t m, uint32_t* start, uint32_t* end, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_sa_interval, dim3(cdiv(n, 256)), dim3(256), 0, S_(stream), ix, kmers, n, m, start, end);
    return (int)hipGetLastError();
}

int gmk_locate(const GmDevIndex& ix, const uint32_t* ranks, uint32_t n, int use_full_sa, uint32_t* out, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_locate, dim3(cdiv(n, 256)), dim3(256), 0, S_(stream), ix, ranks, n, use_full_sa, out);
    return (int)hipGetLastError();
}

int gmk_traceback(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, const GmCand* items, uint32_t n,
                  unsigned long long* ops, uint32_t ops_words, uint16_t* ops_len, const uint8_t* emit, uint32_t* cig_cnt, uint32_t* max_span, void* stream) {
    if (n == 0) return 0;
    if (p.max_gap != 3) return gmk_traceback_band(ix, p, b, items, n, ops, ops_words, ops_len, emit, cig_cnt, max_span, p.max_gap, stream);
    uint32_t Lp = lp_of(b.stride);
    const bool group_form = gm_opt_is("GM_TRACEBACK", "group");   // GM_TRACEBACK=group: the 8-lane form for every length (tests)
    if (!group_form && Lp <= 511) {              // lane form: one lane per item, move rows in LDS
        uint32_t ntab = gm_opt_is("GM_TRACEBACK", "direct") ? 0u : (b.illumina_until ? 2u : 1u);      // value tables in LDS (direct: none, A/B runs)
        if (Lp <= 255) {
            const size_t mv_bytes = ((size_t)(Lp + 1) * 128 * 2 + 15) & ~(size_t)15;
            if (mv_bytes + (size_t)ntab * GM_TBV_ENTRIES * 16 > 65536) ntab = 0;          // (the default dynamic-LDS limit: long rows go without the table)
            const size_t lds = mv_bytes + (size_t)ntab * GM_TBV_ENTRIES * 16;
            uint32_t grid = cdiv(n, 128); if (grid > 16384) grid = 16384;
            hipLaunchKernelGGL((k_traceback_lane<128>), dim3(grid), dim3(128), lds, S_(stream), ix, p, b, items, n, ops, ops_words, ops_len, Lp, emit, cig_cnt, max_span, ntab);
        } else {
            const size_t mv_bytes = ((size_t)(Lp + 1) * 64 * 2 + 15) & ~(size_t)15;
            if (mv_bytes + (size_t)ntab * GM_TBV_ENTRIES * 16 > 65536) ntab = 0;
            const size_t lds = mv_bytes + (size_t)ntab * GM_TBV_ENTRIES * 16;
            uint32_t grid = cdiv(n, 64); if (grid > 16384) grid = 16384;
            hipLaunchKernelGGL((k_traceback_lane<64>), dim3(grid), dim3(64), lds, S_(stream), ix, p, b, items, n, ops, ops_words, ops_len, Lp, emit, cig_cnt, max_span, ntab);
        }
        return (int)hipGetLastError();
    }
    uint32_t mvw = (Lp + 1 + 15) / 16 + 1;
    uint32_t threads = 256;
    if (GM_NW_HDR + (size_t)32 * Lp * 3 + (size_t)32 * 7 * mvw * 4 > 60 * 1024) threads = 64;
    const uint32_t G = threads / 8;
    size_t lds = GM_NW_HDR + (size_t)G * Lp * 3 + (size_t)G * 7 * mvw * 4;
    uint32_t grid = cdiv(n, G); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_traceback, dim3(grid), dim3(threads), lds, S_(stream), ix, p, b, items, n, ops, ops_words, ops_len, Lp, mvw, emit, cig_cnt, max_span);
    return (int)hipGetLastError();
}

int gmk_coverage_add(float* cov, uint64_t bins, uint32_t bin_size, const uint64_t* pos, const uint32_t* span, const float* w,
                     uint32_t n, uint32_t max_span, float* nuc, const uint8_t* codes, const uint64_t* code_off, void* stream) {
    if (n == 0 || max_span == 0) return 0;
    uint64_t total = (uint64_t)n * max_span;
    hipLaunchKernelGGL(k_coverage_add, dim3(cdiv(total, 256)), dim3(256), 0, S_(stream), cov, bins, bin_size, pos, span, w, n, max_span,
                       nuc, codes, code_off);
    return (int)hipGetLastError();
}
