// gm_output.hip — the device half of what the reference does AFTER a candidate has been scored:
//
//   k_group_*        process_hits' unique-sequence map (inc/align_seq2_raw.cpp:102-165) and the -T / -u exits of
//                    align_sequence (:299-306, :146-152) / set_top_matches (src/Driver.cpp:506-610): the accepted hits of a read
//                    are put in the reference's processing order (POS strand pass, then NEG; seed step; ascending position),
//                    grouped by their window string in READ orientation (2-bit packed compare, no strings), the groups ranked in
//                    std::map<string> (lexicographic) order, and written as gm_match / gm_pos records in HBM.
//   k_out_*          create_match_output (src/Driver.cpp:614-753): SAM rows (TopReadOutput, inc/ScoredSeq.h:375-401) with their
//                    run-length CIGAR built from the traceback operations (src/bin_seq.cpp:578-698, fix_CIGAR_for_deletions
//                    inc/SequenceOperations.h:32-42), and the coverage deposit of every kept sequence
//                    (NormalScoredSeq::score src/NormalScoredSeq.cpp:66-75, BSScoredSeq::score src/BSScoredSeq.cpp:24-88).
//
// What stays on the host (gm_api.cpp) is only the order-dependent fp64 libm work on a flat array of scores:
// denominator += exp(score) in processing order, posterior = exp(score) / denominator, MAPQ = round(-10 log10(1 - p)).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "gm_internal.h"

namespace {

__device__ __forceinline__ uint32_t go_pos2rid(const uint32_t* coff, uint32_t n_seqs, uint32_t pos) {   // bns_pos2rid src/bntseq.c:349-363
    uint32_t lo = 0, hi = n_seqs - 1;
    while (lo < hi) {
        uint32_t mid = (lo + hi + 1) >> 1;
        if (pos >= coff[mid]) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// 32 reference bases starting at g, MSB first (2 bits each); the device pac buffer is padded, bytes past the end are never used
// by a valid window
__device__ __forceinline__ unsigned long long go_ref_word(const uint8_t* pac, uint32_t g) {
    const uint8_t* q = pac + (g >> 2);
    unsigned long long w = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) w = (w << 8) | q[k];
    const uint32_t sh = (g & 3u) * 2u;
    if (sh) w = (w << sh) | ((unsigned long long)q[8] >> (8u - sh));
    return w;
}

// reverse the order of the 32 two-bit groups of a word
__device__ __forceinline__ unsigned long long go_rev_pairs(unsigned long long x) {
    x = __brevll(x);
    return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}

// word j (bases 32j .. 32j+31, MSB first, zero padded) of the unique-map key of a hit: the window in READ orientation, i.e. the
// window itself for a POS-strand hit and reverse_comp(window) for a NEG-strand hit (align_seq2_raw.cpp:125-128).  Base codes
// a<c<g<t are 0<1<2<3, so comparing the words as unsigned integers is the std::string comparison of the keys.
__device__ __forceinline__ unsigned long long go_key_word(const uint8_t* pac, uint32_t pos, uint32_t L, uint32_t strand, uint32_t j) {
    const uint32_t done = 32u * j;
    const uint32_t cnt = L - done < 32u ? L - done : 32u;
    unsigned long long k;
    if (!strand) {
        k = go_ref_word(pac, pos + done);
        if (cnt < 32u) k &= ~0ull << (2u * (32u - cnt));
    } else {
        const uint32_t gstart = pos + L - done - cnt;       // lowest reference position of this word's bases
        unsigned long long w = go_ref_word(pac, gstart);    // its top `cnt` groups are the ones wanted
        // the reversal moves them to the low 2*cnt bits in read order; complement (3 - c), then up to the top (zeros shift in)
        k = ~go_rev_pairs(w);
        if (cnt < 32u) k <<= 2u * (32u - cnt);
    }
    return k;
}

// 64-bit hash of a key.  Every word goes through a full-avalanche mixer (the murmur3 finaliser): repeat copies differ from each other
// in a handful of bases, and with a multiply-only step a difference in the top bits of one word stays in the top bits and is cancelled
// by a difference at the same place in the next word - near-identical windows then collide by the dozen (seen on a repeat-rich
// reference: 53 reads of 2424 handed back for "collisions").
__device__ __forceinline__ unsigned long long go_mix64(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ unsigned long long go_key_hash(const uint8_t* pac, uint32_t pos, uint32_t L, uint32_t strand) {
    unsigned long long h = 0x243F6A8885A308D3ull;
    const uint32_t nw = (L + 31u) >> 5;
    for (uint32_t j = 0; j < nw; ++j) h = go_mix64(h ^ go_key_word(pac, pos, L, strand, j)) + 0x9E3779B97F4A7C15ull;
    return h;
}

// <0, 0, >0: key(a) vs key(b) in std::string order
__device__ __forceinline__ int go_key_cmp(const uint8_t* pac, uint32_t L, uint32_t pa, uint32_t sa, uint32_t pb, uint32_t sb) {
    if (pa == pb && sa == sb) return 0;
    const uint32_t nw = (L + 31u) >> 5;
    for (uint32_t j = 0; j < nw; ++j) {
        unsigned long long x = go_key_word(pac, pa, L, sa, j), y = go_key_word(pac, pb, L, sb, j);
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}

// the reference's processing order of accepted hits: POS strand pass before NEG (Driver.cpp:506-556); inside a pass the seed step
// at which the position reached -k votes (align_sequence loop), then ascending position (std::map<unsigned long,int>, process_hits).
// --no_nw scores every position of a strand in ONE process_hits call at the end (:317-325): strand, then position.
__device__ __forceinline__ bool go_hit_less(const GmRawHit& a, const GmRawHit& b, int nw) {
    if (a.strand != b.strand) return a.strand < b.strand;
    if (nw && a.step != b.step) return a.step < b.step;
    return a.pos < b.pos;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// grouping, pass 1: reads with 0 or 1 accepted hits are finished by one thread; the others go to a list
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_group_single(GmDevBatch b, GmDevGroup g, int nw, uint32_t max_matches, uint32_t big_min) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b.n) return;
    const uint64_t hb = b.hit_begin[r];
    const uint32_t k = (uint32_t)(b.hit_begin[r + 1] - hb);
    uint32_t nm = 0;
    if (b.status[r] == 0) {
        if (k == 0) b.status[r] = 2;                                     // GM_READ_NONE: the unique map stayed empty (Driver.cpp:595-601)
        else if (k == 1) {
            const GmRawHit h = b.raw_hits[hb];
            g.sorted[hb] = h; g.ord_score[hb] = h.score; g.lead[hb] = 0; g.krank[hb] = 0;
            if (nw && 1u > max_matches) b.status[r] = 1;                  // unique.size() > gMAX_MATCHES (align_seq2_raw.cpp:299-306)
            else nm = 1;
        } else {
            if (k > big_min) { const uint32_t at = atomicAdd(g.n_big, 1u); g.big_list[at] = r; g.big_done[at] = 0; }
            else g.multi_list[atomicAdd(g.n_multi, 1u)] = r;
            return;                                                      // n_match is written by k_group_multi / k_group_big
        }
    }
    g.n_match[r] = nm;
}

// one wavefront per read with >= 2 accepted hits
__global__ void __launch_bounds__(64) k_group_multi(GmDevIndex ix, GmDevBatch b, GmDevGroup g, int nw, int unique_only, uint32_t max_matches) {
    __shared__ uint32_t s_cnt[2];
    const int lane = threadIdx.x;
    const uint32_t n_multi = *g.n_multi;
    for (uint32_t li = blockIdx.x; li < n_multi; li += gridDim.x) {
        const uint32_t r = g.multi_list[li];
        const uint64_t hb = b.hit_begin[r];
        const uint32_t k = (uint32_t)(b.hit_begin[r + 1] - hb);
        const uint32_t L = b.len[r];
        const GmRawHit* raw = b.raw_hits + hb;
        GmRawHit* srt = g.sorted + hb;
        if (lane < 2) s_cnt[lane] = 0;
        // 1. processing order: rank of every hit among the read's hits ((strand, pos) is unique, so the order is total)
        for (uint32_t a = lane; a < k; a += 64) {
            const GmRawHit ha = raw[a];
            uint32_t rank = 0;
            for (uint32_t c = 0; c < k; ++c) rank += go_hit_less(raw[c], ha, nw) ? 1u : 0u;
            srt[rank] = ha;
        }
        __syncthreads();
        // 2. key hash of every hit
        for (uint32_t s = lane; s < k; s += 64) g.khash[hb + s] = go_key_hash(ix.pac, srt[s].pos, L, srt[s].strand);
        __syncthreads();
        // 3. leader = the first hit in processing order with the same key (hash, then the 2-bit words themselves)
        for (uint32_t s = lane; s < k; s += 64) {
            const unsigned long long hs = g.khash[hb + s];
            const GmRawHit h = srt[s];
            uint32_t l = s;
            for (uint32_t t = 0; t < s; ++t)
                if (g.khash[hb + t] == hs && go_key_cmp(ix.pac, L, srt[t].pos, srt[t].strand, h.pos, h.strand) == 0) { l = t; break; }
            g.lead[hb + s] = l;
            g.ord_score[hb + s] = h.score;
        }
        __syncthreads();
        bool too_many = false;
        if (unique_only) {
            if (nw) {
                // -u: an existing key makes align_sequence return false -> READ_TOO_MANY (align_seq2_raw.cpp:146-152, Driver.cpp:512-525)
                bool dup = false;
                for (uint32_t s = lane; s < k; s += 64) dup |= g.lead[hb + s] != s;
                too_many = __ballot(dup) != 0ull;
            } else if (lane == 0) {
                // --no_nw ignores process_hits' return value (:317-325): the rest of THAT strand's single pass is dropped, nothing else
                bool stopped = false; uint32_t cur = 2;
                for (uint32_t s = 0; s < k; ++s) {
                    if (srt[s].strand != cur) { cur = srt[s].strand; stopped = false; }
                    bool drop = stopped;
                    if (!drop) {
                        for (uint32_t t = 0; t < s; ++t)
                            if (g.lead[hb + t] != 0xFFFFFFFFu && g.khash[hb + t] == g.khash[hb + s] &&
                                go_key_cmp(ix.pac, L, srt[t].pos, srt[t].strand, srt[s].pos, srt[s].strand) == 0) { drop = true; stopped = true; break; }
                    }
                    if (drop) { g.lead[hb + s] = 0xFFFFFFFFu; g.ord_score[hb + s] = -INFINITY; }
                    else g.lead[hb + s] = s;
                }
            }
            __syncthreads();
        }
        // 4. number of keys first: a read beyond -T is READ_TOO_MANY whatever the order of its keys is (the map only grows, so the
        //    check after the last seed decides, align_seq2_raw.cpp:299-306) - and ranking thousands of repeat copies would be the
        //    one quadratic step with full key compares
        uint32_t my_leaders = 0;
        for (uint32_t s = lane; s < k; s += 64) my_leaders += g.lead[hb + s] == s ? 1u : 0u;
        if (my_leaders) atomicAdd(&s_cnt[0], my_leaders);
        __syncthreads();
        const uint32_t u = s_cnt[0];
        if (nw && u > max_matches) too_many = true;
        // 5. rank of every key among the read's keys (std::map<string> order): first words first, the whole key only on a tie
        if (!too_many) {
            for (uint32_t s = lane; s < k; s += 64)
                if (g.lead[hb + s] == s) g.khash[hb + s] = go_key_word(ix.pac, srt[s].pos, L, srt[s].strand, 0);      // the hash has done its work
            __syncthreads();
            for (uint32_t s = lane; s < k; s += 64) {
                if (g.lead[hb + s] != s) continue;
                const GmRawHit h = srt[s];
                const unsigned long long w0 = g.khash[hb + s];
                uint32_t rank = 0;
                for (uint32_t t = 0; t < k; ++t) {
                    if (t == s || g.lead[hb + t] != t) continue;
                    const unsigned long long wt = g.khash[hb + t];
                    if (wt < w0 || (wt == w0 && go_key_cmp(ix.pac, L, srt[t].pos, srt[t].strand, h.pos, h.strand) < 0)) ++rank;
                }
                g.krank[hb + s] = rank;
            }
        }
        if (lane == 0) {
            if (too_many) { b.status[r] = 1; g.n_match[r] = 0; }
            else if (u == 0) { b.status[r] = 2; g.n_match[r] = 0; }
            else g.n_match[r] = u;
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------------
// grouping of reads with MANY accepted hits (repeat copies: up to 10^5 .. 10^6 per read), linear in the hits instead of the
// all-pairs steps of k_group_multi:
//   * distinct keys are counted in an LDS hash set keyed by the 64-bit key hash, value = the smallest processing-order key among
//     the hits with that hash.  More distinct hashes than -T means more distinct keys than -T: READ_TOO_MANY as soon as the count
//     passes -T, after a few thousand insertions (what the reference finds out seed by seed, align_seq2_raw.cpp:299-306)
//   * otherwise the hits are put in processing order by a wave-level LSD radix sort in HBM, the leader of a hit is found through the
//     set (and verified on the 2-bit windows: a hash collision hands the read back to the all-pairs kernel), the few keys are ranked
//     all-pairs, and a second sort by (key rank, position, strand) lays out the matches' position sets (k_group_write_big)
// ------------------------------------------------------------------------------------------------
#define GO_SET_SLOTS 4096u
#define GO_SET_LIMIT 3000u

// stable LSD radix sort (4-bit digits) of (key, value) pairs by one wavefront, ping-pong between (k0, v0) and (k1, v1); returns 0 / 1 =
// which pair of buffers holds the result.  Needs blockDim.x == 64 (uses __syncthreads between passes).
__device__ int go_wave_sort(unsigned long long* k0, uint32_t* v0, unsigned long long* k1, uint32_t* v1, uint32_t n, int nbits, int lane) {
    unsigned long long* kin = k0; uint32_t* vin = v0; unsigned long long* kout = k1; uint32_t* vout = v1;
    int cur = 0;
    for (int shift = 0; shift < nbits; shift += 4) {
        // histogram: lane d (< 16) ends up with the number of keys whose digit is d
        uint32_t total = 0;
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t i = base + lane;
            const uint32_t dig = i < n ? (uint32_t)(kin[i] >> shift) & 15u : 16u;
#pragma unroll
            for (uint32_t d = 0; d < 16; ++d) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(dig == d);
                if ((uint32_t)lane == d) total += (uint32_t)__popcll(m);
            }
        }
        // exclusive prefix over the 16 digits (lanes 0..15)
        uint32_t incl = (uint32_t)lane < 16u ? total : 0u;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        uint32_t run = incl - ((uint32_t)lane < 16u ? total : 0u);       // lane d: where the next key with digit d goes
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t i = base + lane;
            unsigned long long key = 0; uint32_t val = 0;
            uint32_t dig = 16u;
            if (i < n) { key = kin[i]; val = vin[i]; dig = (uint32_t)(key >> shift) & 15u; }
            uint32_t my_rank = 0;
#pragma unroll
            for (uint32_t d = 0; d < 16; ++d) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(dig == d);
                const uint32_t start = __shfl(run, (int)d);
                if (dig == d) my_rank = start + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if ((uint32_t)lane == d) run += (uint32_t)__popcll(m);
            }
            if (i < n) { kout[my_rank] = key; vout[my_rank] = val; }
        }
        __syncthreads();
        unsigned long long* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
        cur ^= 1;
    }
    return cur;
}

__global__ void __launch_bounds__(64) k_group_big(GmDevIndex ix, GmDevBatch b, GmDevGroup g, int nw, int unique_only, uint32_t max_matches) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_big[];
    unsigned long long* s_keys = reinterpret_cast<unsigned long long*>(s_big);           // GO_SET_SLOTS hashes (0 = empty)
    unsigned long long* s_vals = s_keys + GO_SET_SLOTS;                                  // smallest order key of the hash
    uint32_t* s_lead = reinterpret_cast<uint32_t*>(s_vals + GO_SET_SLOTS);               // up to GO_SET_LIMIT leaders (sorted index)
    __shared__ uint32_t s_n[4];                                                          // distinct hashes, leaders, flags
    const int lane = threadIdx.x;
    const uint32_t n_big = *g.n_big;
    for (uint32_t li = blockIdx.x; li < n_big; li += gridDim.x) {
        const uint32_t r = g.big_list[li];
        const uint64_t hb = b.hit_begin[r];
        const uint32_t k = (uint32_t)(b.hit_begin[r + 1] - hb);
        const uint32_t L = b.len[r];
        const GmRawHit* raw = b.raw_hits + hb;
        GmRawHit* srt = g.sorted + hb;
        auto hand_back = [&](int why) { if (lane == 0) { g.multi_list[atomicAdd(g.n_multi, 1u)] = r; atomicAdd(g.n_big + 1 + why, 1u); } };    // the all-pairs kernel runs after this one; n_big[1..3] count the reasons
        if (unique_only && !nw) { hand_back(0); continue; }                // per-strand drop rule of --no_nw -u: rare, all-pairs kernel
        for (uint32_t q = lane; q < GO_SET_SLOTS; q += 64) { s_keys[q] = 0ull; s_vals[q] = ~0ull; }
        if (lane < 4) s_n[lane] = 0;
        __syncthreads();
        const uint32_t limit = nw && max_matches < GO_SET_LIMIT ? max_matches : GO_SET_LIMIT;
        // 1. hash + processing-order key of every hit; distinct hashes into the set (stops as soon as the count passes the limit)
        for (uint32_t base = 0; base < k; base += 64) {
            const uint32_t a = base + lane;
            if (a < k) {
                const GmRawHit h = raw[a];
                const unsigned long long hs = go_key_hash(ix.pac, h.pos, L, h.strand) | 1ull;
                const unsigned long long ok = nw ? ((unsigned long long)h.strand << 48) | ((unsigned long long)h.step << 32) | h.pos
                                                 : ((unsigned long long)h.strand << 48) | h.pos;
                g.khash[hb + a] = hs; g.sk0[hb + a] = ok; g.si0[hb + a] = a;
                uint32_t slot = (uint32_t)(hs >> 17) & (GO_SET_SLOTS - 1u);
                for (uint32_t probe = 0; probe < GO_SET_SLOTS; ++probe) {
                    const unsigned long long old = atomicCAS(&s_keys[slot], 0ull, hs);
                    if (old == 0ull) atomicAdd(&s_n[0], 1u);
                    if (old == 0ull || old == hs) { atomicMin(&s_vals[slot], ok); break; }
                    slot = (slot + 1u) & (GO_SET_SLOTS - 1u);
                }
            }
            __syncthreads();
            if (s_n[0] > limit) break;                                    // uniform: read after the barrier
        }
        __syncthreads();
        const uint32_t distinct = s_n[0];
        if (distinct > limit) {
            if (nw && limit == max_matches) {                             // more distinct keys than -T: READ_TOO_MANY (Driver.cpp:512-525)
                if (lane == 0) { b.status[r] = 1; g.n_match[r] = 0; g.big_done[li] = 1; }
            } else hand_back(1);                                          // more keys than the set holds and no -T to stop at: all-pairs kernel
            continue;
        }
        // 2. processing order
        const int where = go_wave_sort(g.sk0 + hb, g.si0 + hb, g.sk1 + hb, g.si1 + hb, k, 52, lane);
        const unsigned long long* sk = (where ? g.sk1 : g.sk0) + hb;
        const uint32_t* si = (where ? g.si1 : g.si0) + hb;
        unsigned long long* spare_k = (where ? g.sk0 : g.sk1) + hb;
        for (uint32_t i = lane; i < k; i += 64) { const GmRawHit h = raw[si[i]]; srt[i] = h; g.ord_score[hb + i] = h.score; }
        __syncthreads();
        // 3. leader of every hit = the hit with the smallest order key among those with its hash (verified on the windows)
        bool dup = false, collide = false;
        for (uint32_t i = lane; i < k; i += 64) {
            const unsigned long long hs = g.khash[hb + si[i]];
            uint32_t slot = (uint32_t)(hs >> 17) & (GO_SET_SLOTS - 1u);
            while (s_keys[slot] != hs) slot = (slot + 1u) & (GO_SET_SLOTS - 1u);
            const unsigned long long lk = s_vals[slot];
            uint32_t lo = 0, hi = k - 1;                                  // order keys are unique: binary search for the leader's place
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sk[mid] < lk) lo = mid + 1; else hi = mid; }
            const GmRawHit h = srt[i], hl = srt[lo];
            if (lo != i && go_key_cmp(ix.pac, L, hl.pos, hl.strand, h.pos, h.strand) != 0) collide = true;
            g.lead[hb + i] = lo;
            if (lo != i) dup = true;
            else { const uint32_t at = atomicAdd(&s_n[1], 1u); if (at < GO_SET_LIMIT) s_lead[at] = i; }
        }
        __syncthreads();
        if (__builtin_amdgcn_ballot_w64(collide) != 0ull || s_n[1] != distinct) { hand_back(2); continue; }    // two keys under one hash: all-pairs kernel
        if (unique_only && nw && __builtin_amdgcn_ballot_w64(dup) != 0ull) {     // -u: an existing key -> READ_TOO_MANY (align_seq2_raw.cpp:146-152)
            if (lane == 0) { b.status[r] = 1; g.n_match[r] = 0; g.big_done[li] = 1; }
            continue;
        }
        // 4. rank of the (few) keys in std::map<string> order: first words first, the whole key on a tie
        for (uint32_t q = lane; q < distinct; q += 64) { const GmRawHit h = srt[s_lead[q]]; spare_k[q] = go_key_word(ix.pac, h.pos, L, h.strand, 0); }
        __syncthreads();
        for (uint32_t q = lane; q < distinct; q += 64) {
            const uint32_t i = s_lead[q];
            const GmRawHit h = srt[i];
            const unsigned long long w0 = spare_k[q];
            uint32_t rank = 0;
            for (uint32_t t = 0; t < distinct; ++t) {
                if (t == q) continue;
                const unsigned long long wt = spare_k[t];
                if (wt < w0) ++rank;
                else if (wt == w0) { const GmRawHit ht = srt[s_lead[t]]; if (go_key_cmp(ix.pac, L, ht.pos, ht.strand, h.pos, h.strand) < 0) ++rank; }
            }
            g.krank[hb + i] = rank;
        }
        if (lane == 0) { g.n_match[r] = distinct; g.big_done[li] = 1; }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(64) k_group_write_big(GmDevBatch b, GmDevGroup g) {
    const int lane = threadIdx.x;
    const uint32_t n_big = *g.n_big;
    for (uint32_t li = blockIdx.x; li < n_big; li += gridDim.x) {
        const uint32_t r = g.big_list[li];
        if (!g.big_done[li] || g.n_match[r] == 0) continue;
        const uint64_t hb = b.hit_begin[r], mb = g.match_begin[r];
        const uint32_t k = (uint32_t)(b.hit_begin[r + 1] - hb);
        const GmRawHit* srt = g.sorted + hb;
        // second sort: (key rank, position, strand) = the matches in std::map order, each with its set<(pos,strand)> in order
        for (uint32_t i = lane; i < k; i += 64) {
            const GmRawHit h = srt[i];
            g.sk0[hb + i] = ((unsigned long long)g.krank[hb + g.lead[hb + i]] << 33) | ((unsigned long long)h.pos << 1) | h.strand;
            g.si0[hb + i] = i;
        }
        __syncthreads();
        const int where = go_wave_sort(g.sk0 + hb, g.si0 + hb, g.sk1 + hb, g.si1 + hb, k, 48, lane);
        const unsigned long long* sk = (where ? g.sk1 : g.sk0) + hb;
        const uint32_t* si = (where ? g.si1 : g.si0) + hb;
        for (uint32_t j = lane; j < k; j += 64) {
            const uint32_t i = si[j];
            const GmRawHit h = srt[i];
            const uint32_t grp = (uint32_t)(sk[j] >> 33);
            GmDevPos p; p.pos = h.pos; p.strand = h.strand; for (int q = 0; q < 7; ++q) p.pad[q] = 0;
            g.positions[hb + j] = p;
            GmDevMatch* m = g.matches + mb + grp;
            if (j == 0 || (uint32_t)(sk[j - 1] >> 33) != grp) {
                const uint32_t l = g.lead[hb + i];
                const GmRawHit hl = srt[l];                               // the FIRST hit of a key gives the ScoredSeq its sequence, score and strand
                m->read = b.read_base + r; m->score = hl.score; m->first_pos = hl.pos; m->first_strand = hl.strand;
                m->pad[0] = m->pad[1] = m->pad[2] = 0; m->tail = 0;
                m->pos_begin = (uint32_t)(hb + j);
                g.match_hit[mb + grp] = (uint32_t)(hb + l);
            }
            if (j + 1 == k || (uint32_t)(sk[j + 1] >> 33) != grp) m->pos_end = (uint32_t)(hb + j + 1);
        }
        __syncthreads();
    }
}

// grouping, pass 2 (after the scan of n_match): gm_match records in key order, their positions as the ordered set<(pos,strand)>
__global__ void __launch_bounds__(256) k_group_write_single(GmDevBatch b, GmDevGroup g) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b.n) return;
    const uint64_t hb = b.hit_begin[r];
    if (g.n_match[r] != 1 || b.hit_begin[r + 1] - hb != 1) return;
    const GmRawHit h = g.sorted[hb];
    GmDevMatch m;
    m.read = b.read_base + r; m.score = h.score; m.first_pos = h.pos; m.first_strand = h.strand; m.pad[0] = m.pad[1] = m.pad[2] = 0;
    m.pos_begin = (uint32_t)hb; m.pos_end = (uint32_t)hb + 1u;
    g.matches[g.match_begin[r]] = m;
    g.match_hit[g.match_begin[r]] = (uint32_t)hb;
    GmDevPos p; p.pos = h.pos; p.strand = h.strand; for (int q = 0; q < 7; ++q) p.pad[q] = 0;
    g.positions[hb] = p;
}

__global__ void __launch_bounds__(64) k_group_write_multi(GmDevBatch b, GmDevGroup g) {
    const int lane = threadIdx.x;
    const uint32_t n_multi = *g.n_multi;
    for (uint32_t li = blockIdx.x; li < n_multi; li += gridDim.x) {
        const uint32_t r = g.multi_list[li];
        if (g.n_match[r] == 0) continue;
        const uint64_t hb = b.hit_begin[r], mb = g.match_begin[r];
        const uint32_t k = (uint32_t)(b.hit_begin[r + 1] - hb);
        const GmRawHit* srt = g.sorted + hb;
        for (uint32_t s = lane; s < k; s += 64) {
            const uint32_t l = g.lead[hb + s];
            if (l == 0xFFFFFFFFu) continue;
            const uint32_t grp = g.krank[hb + l];
            const GmRawHit h = srt[s];
            uint32_t within = 0, goff = 0, gsize = 0;
            for (uint32_t t = 0; t < k; ++t) {
                const uint32_t lt = g.lead[hb + t];
                if (lt == 0xFFFFFFFFu) continue;
                if (lt == l) {
                    ++gsize;
                    const GmRawHit ht = srt[t];
                    within += (ht.pos < h.pos || (ht.pos == h.pos && ht.strand < h.strand)) ? 1u : 0u;     // set<pair<pos,strand>> order
                } else if (g.krank[hb + lt] < grp) ++goff;
            }
            GmDevPos p; p.pos = h.pos; p.strand = h.strand; for (int q = 0; q < 7; ++q) p.pad[q] = 0;
            g.positions[hb + goff + within] = p;
            if (s == l) {                                                 // the FIRST hit of a key gives the ScoredSeq its sequence, score and strand
                GmDevMatch m;
                m.read = b.read_base + r; m.score = h.score; m.first_pos = h.pos; m.first_strand = h.strand; m.pad[0] = m.pad[1] = m.pad[2] = 0;
                m.pos_begin = (uint32_t)(hb + goff); m.pos_end = (uint32_t)(hb + goff + gsize);
                g.matches[mb + grp] = m;
                g.match_hit[mb + grp] = (uint32_t)(hb + s);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// output stage
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_out_items(const GmDevMatch* matches, uint32_t n_m, uint32_t read_base, GmCand* items, uint32_t* pos_match) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_m) return;
    const GmDevMatch mm = matches[m];
    GmCand c; c.rs = (mm.read - read_base) * 2u + mm.first_strand; c.b = (uint32_t)mm.first_pos; c.step = 0; c.flags = 0; c.pad = 0; c.score = 0;
    items[m] = c;
    for (uint32_t q = mm.pos_begin; q < mm.pos_end; ++q) pos_match[q] = m;
}

// once the host pass has said which sequences are printed: SAM rows and CIGAR bytes per match (the traceback kernel left the CIGAR
// length of EVERY match, it ran while the host was still computing)
__global__ void __launch_bounds__(256) k_out_sizes(const GmDevMatch* matches, uint32_t n_m, const uint8_t* emit, const uint32_t* cig_all, uint32_t* rec_cnt,
                                                   uint32_t* cig_cnt) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_m) return;
    const bool e = emit[m] != 0;
    rec_cnt[m] = e ? matches[m].pos_end - matches[m].pos_begin : 0u;     // one SAM row per place of a printed sequence (ScoredSeq.h:375-401)
    cig_cnt[m] = e ? cig_all[m] : 0u;
}

// run-length CIGAR text of a traceback (bin_seq.cpp:578-698) from the packed operations (2 bits each: 0 M, 1 I, 2 D); "" -> "*",
// otherwise one trailing run of D is stripped (fix_CIGAR_for_deletions, SequenceOperations.h:32-42).  Returns the length written.
__device__ __forceinline__ uint32_t go_op(const unsigned long long* ops, uint32_t k) { return (uint32_t)(ops[k >> 5] >> (2u * (k & 31u))) & 3u; }

__device__ __forceinline__ uint32_t go_digits(uint32_t v) { return v >= 10000 ? 5u : v >= 1000 ? 4u : v >= 100 ? 3u : v >= 10 ? 2u : 1u; }

__device__ __forceinline__ uint32_t go_cigar(const unsigned long long* ops, uint32_t n_op, char* out) {
    if (n_op == 0) { out[0] = '*'; return 1; }
    uint32_t end = n_op;
    if (go_op(ops, n_op - 1) == 2u) { while (end > 0 && go_op(ops, end - 1) == 2u) --end; }
    uint32_t w = 0, i = 0;
    unsigned long long word = 0;
    while (i < end) {
        if ((i & 31u) == 0 || i == 0) word = ops[i >> 5];
        const uint32_t op = (uint32_t)(word >> (2u * (i & 31u))) & 3u;
        uint32_t j = i + 1;
        while (j < end) {
            if ((j & 31u) == 0) word = ops[j >> 5];
            if (((uint32_t)(word >> (2u * (j & 31u))) & 3u) != op) break;
            ++j;
        }
        const uint32_t run = j - i, digits = go_digits(run);
        uint32_t v = run;
        for (uint32_t d = digits; d > 0; --d) { out[w + d - 1] = (char)('0' + v % 10u); v /= 10u; }
        out[w + digits] = op == 0 ? 'M' : op == 1 ? 'I' : 'D';
        w += digits + 1;
        i = j;
    }
    return w;
}

__global__ void __launch_bounds__(256) k_out_write(GmDevIndex ix, GmDevBatch b, const GmDevMatch* matches, const GmDevPos* positions, uint32_t n_m,
                                                   const uint8_t* emit, const int32_t* mapq, const float* post, const unsigned long long* ops, uint32_t ops_words,
                                                   const uint16_t* ops_len, int nw, const uint64_t* rec_off, const uint64_t* cig_off,
                                                   GmDevSamRec* recs, char* pool) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_m || !emit[m]) return;
    const GmDevMatch mm = matches[m];
    const uint64_t co = cig_off[m];
    char* cg = pool + co;
    uint32_t cl;
    if (nw) cl = go_cigar(ops + (size_t)m * ops_words, ops_len[m], cg);
    else {                                                              // --no_nw: "<L>M" (ScoredSeq.h:330-340)
        uint32_t L = b.len[mm.read - b.read_base], d = go_digits(L), v = L;
        for (uint32_t q = d; q > 0; --q) { cg[q - 1] = (char)('0' + v % 10u); v /= 10u; }
        cg[d] = 'M'; cl = d + 1;
    }
    cg[cl] = 0;
    uint64_t ro = rec_off[m];
    for (uint32_t q = mm.pos_begin; q < mm.pos_end; ++q, ++ro) {
        const GmDevPos p = positions[q];
        GmDevSamRec s;
        s.read = mm.read; s.pad0 = 0; s.pos = p.pos;
        s.contig = go_pos2rid(ix.contig_off, ix.n_seqs, (uint32_t)p.pos); s.pad1 = 0;
        const int base = (int)((uint32_t)p.pos - ix.contig_off[s.contig]);          // int chr_base_pos, GenomeBwt.cpp:632
        s.chr_pos = (unsigned long long)(long long)base + 1ull;
        s.strand = p.strand; s.pad2[0] = s.pad2[1] = s.pad2[2] = 0;
        s.mapq = mapq[m]; s.a_score = mm.score; s.post_prob = post[m];
        s.sim_matches = (int32_t)(mm.pos_end - mm.pos_begin); s.cigar_off = (uint32_t)co;
        recs[ro] = s;
    }
}

// -b / -d: the gapped read string of a kept sequence as g_gen_CONVERSION codes (a,c,g,t,n = 0..4; 6 = the NUL the reference reads one
// past the consensus, which AddSeqScore ignores), built from the traceback operations on the argmax consensus of the read in the
// orientation of the match's first strand (ScoredSeq.h:57-103 max_char, bin_seq.cpp:578-698 incl. the consense[i] quirk at :607)
__device__ __forceinline__ uint32_t go_cons_code(const GmDevBatch& b, const float2* lut, uint32_t r, uint32_t L, uint32_t first_strand, uint32_t i) {
    const uint32_t src = first_strand ? L - 1u - i : i;
    const uint8_t ch = b.bases[(size_t)r * b.stride + src], q = b.quals[(size_t)r * b.stride + src];
    const float2 pq = lut[q];
    int code;
    switch (ch) { case 'a': case 'A': code = 0; break; case 'c': case 'C': code = 1; break; case 'g': case 'G': code = 2; break;
                  case 't': case 'T': code = 3; break; default: code = 4; }
    if (first_strand && code < 4) code = 3 - code;
    float c[4] = { pq.y, pq.y, pq.y, pq.y };
    if (code < 4) c[code] = pq.x;
    if (c[0] == c[1] && c[0] == c[2] && c[0] == c[3]) return 4;        // 'n'
    if (c[0] >= c[1]) { if (c[0] >= c[2]) return c[0] >= c[3] ? 0u : 3u; return c[2] >= c[3] ? 2u : 3u; }
    if (c[1] >= c[2]) return c[1] >= c[3] ? 1u : 3u;
    return c[2] >= c[3] ? 2u : 3u;
}

__global__ void __launch_bounds__(256) k_out_codes(GmDevBatch b, GmDevParams p, const GmDevMatch* matches, uint32_t n_m, const unsigned long long* ops,
                                                   uint32_t ops_words, const uint16_t* ops_len, uint8_t* codes, uint32_t codes_stride) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_m) return;
    const GmDevMatch mm = matches[m];
    const uint32_t r = mm.read - b.read_base, L = b.len[r];
    const float2* lut = p.lut + ((r < b.illumina_until) ? 256 : 0);
    const unsigned long long* op = ops + (size_t)m * ops_words;
    uint8_t* out = codes + (size_t)m * codes_stride;
    const uint32_t n_op = ops_len[m];
    uint32_t rr = 0;
    for (uint32_t k = 0; k < n_op; ++k) {
        uint8_t cv;
        const uint32_t o = go_op(op, k);
        if (o == 0) { cv = rr < L ? (uint8_t)go_cons_code(b, lut, r, L, mm.first_strand, rr) : 6; ++rr; }
        else if (o == 1) { cv = rr + 1 < L ? (uint8_t)go_cons_code(b, lut, r, L, mm.first_strand, rr + 1) : 6; ++rr; }
        else cv = 5;                                                    // '-'
        out[k] = cv;
    }
}

// coverage deposit of every kept sequence at every one of its places: amount_genome[(pos+t)/bin] += (float)posterior for t < aligned
// length (GenomeBwt::AddScore src/GenomeBwt.cpp:483-490).  One thread per (place, bin): the bases of a place that fall into one bin
// are added up in a register in base order - exactly the fp32 sequence the reference runs through on an untouched bin - and reach
// HBM as ONE atomic per bin instead of one per base (8 x fewer atomics at the default bin size).  -b / -d (bin size 1) additionally
// reads[base][pos+t] += posterior with the gapped read string reverse-complemented for places on the other strand than the first
// (BSScoredSeq.cpp:50-84).
__global__ void __launch_bounds__(256) k_out_deposit(float* cov, uint64_t bins, uint32_t bin_size, const GmDevMatch* matches, const GmDevPos* positions,
                                                     const uint32_t* pos_match, uint64_t n_p, const uint16_t* ops_len, const float* post, uint32_t bins_per_place,
                                                     float* nuc, const uint8_t* codes, uint32_t codes_stride) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t q = gid / bins_per_place; const uint32_t bi = (uint32_t)(gid % bins_per_place);
    if (q >= n_p) return;
    const uint32_t m = pos_match[q];
    if (m == 0xFFFFFFFFu) return;
    const uint32_t span = ops_len[m];
    const GmDevPos p = positions[q];
    const uint64_t bin = p.pos / bin_size + bi;
    if (bin >= bins || span == 0) return;
    // bases t of this place inside the bin: pos + t in [bin * bin_size, (bin + 1) * bin_size)
    const uint64_t lo = bin * bin_size > p.pos ? bin * bin_size - p.pos : 0;
    uint64_t hi = (bin + 1) * bin_size - p.pos;
    if (hi > span) hi = span;
    if (lo >= hi) return;
    const float w = post[m];
    float acc = 0.0f;
    for (uint64_t t = lo; t < hi; ++t) acc += w;
    atomicAdd(&cov[bin], acc);
    if (nuc) {
        const uint8_t* cd = codes + (size_t)m * codes_stride;
        const bool same = p.strand == matches[m].first_strand;
        for (uint32_t t = (uint32_t)lo; t < (uint32_t)hi; ++t) {
            // codes: 0..3 acgt, 4 'n', 5 '-', 6 the NUL read past the consensus.  Same strand: '-' falls into the N slot (g_gen_CONVERSION
            // default), NUL is ignored.  Other strand: reverse_comp first (SequenceOperations.h:56-96: acgt swapped, everything else 'n')
            uint32_t c;
            if (same) { c = cd[t]; if (c == 5u) c = 4u; }
            else { c = cd[span - 1u - t]; c = c < 4u ? 3u - c : 4u; }
            if (c < 5u) atomicAdd(&nuc[(size_t)c * bins + (p.pos + t) / bin_size], w);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

int gmk_group_count(const GmDevIndex& ix, const GmDevBatch& b, const GmDevGroup& g, int nw, int unique_only, uint32_t max_matches, void* stream) {
    if (b.n == 0) return 0;
    const uint32_t big_min = (uint32_t)gm_opt_ll("GM_GROUP_BIG_MIN", GM_GROUP_BIG);     // test switch
    hipLaunchKernelGGL(k_group_single, dim3(cdiv(b.n, 256)), dim3(256), 0, S_(stream), b, g, nw, max_matches, big_min);
    {   // reads with many accepted hits first: what this kernel cannot finish goes to the all-pairs kernel's list
        const size_t lds = (size_t)GO_SET_SLOTS * 16 + (size_t)GO_SET_LIMIT * 4 + 64;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_group_big), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipLaunchKernelGGL(k_group_big, dim3(std::min<uint32_t>(b.n, 4096u)), dim3(64), lds, S_(stream), ix, b, g, nw, unique_only, max_matches);
    }
    hipLaunchKernelGGL(k_group_multi, dim3(std::min<uint32_t>(b.n, 16384u)), dim3(64), 0, S_(stream), ix, b, g, nw, unique_only, max_matches);
    return (int)hipGetLastError();
}

int gmk_group_write(const GmDevBatch& b, const GmDevGroup& g, void* stream) {
    if (b.n == 0) return 0;
    hipLaunchKernelGGL(k_group_write_single, dim3(cdiv(b.n, 256)), dim3(256), 0, S_(stream), b, g);
    hipLaunchKernelGGL(k_group_write_multi, dim3(std::min<uint32_t>(b.n, 16384u)), dim3(64), 0, S_(stream), b, g);
    hipLaunchKernelGGL(k_group_write_big, dim3(std::min<uint32_t>(b.n, 4096u)), dim3(64), 0, S_(stream), b, g);
    return (int)hipGetLastError();
}

int gmk_out_items(const GmDevMatch* matches, uint32_t n_m, uint32_t read_base, GmCand* items, uint32_t* pos_match, void* stream) {
    if (n_m == 0) return 0;
    hipLaunchKernelGGL(k_out_items, dim3(cdiv(n_m, 256)), dim3(256), 0, S_(stream), matches, n_m, read_base, items, pos_match);
    return (int)hipGetLastError();
}

int gmk_out_sizes(const GmDevMatch* matches, uint32_t n_m, const uint8_t* emit, const uint32_t* cig_all, uint32_t* rec_cnt, uint32_t* cig_cnt, void* stream) {
    if (n_m == 0) return 0;
    hipLaunchKernelGGL(k_out_sizes, dim3(cdiv(n_m, 256)), dim3(256), 0, S_(stream), matches, n_m, emit, cig_all, rec_cnt, cig_cnt);
    return (int)hipGetLastError();
}

int gmk_out_write(const GmDevIndex& ix, const GmDevBatch& b, const GmDevMatch* matches, const GmDevPos* positions, uint32_t n_m, const uint8_t* emit,
                  const int32_t* mapq, const float* post, const unsigned long long* ops, uint32_t ops_words, const uint16_t* ops_len, int nw,
                  const uint64_t* rec_off, const uint64_t* cig_off, GmDevSamRec* recs, char* pool, void* stream) {
    if (n_m == 0) return 0;
    hipLaunchKernelGGL(k_out_write, dim3(cdiv(n_m, 256)), dim3(256), 0, S_(stream), ix, b, matches, positions, n_m, emit, mapq, post, ops, ops_words, ops_len,
                       nw, rec_off, cig_off, recs, pool);
    return (int)hipGetLastError();
}

int gmk_out_codes(const GmDevBatch& b, const GmDevParams& p, const GmDevMatch* matches, uint32_t n_m, const unsigned long long* ops, uint32_t ops_words,
                  const uint16_t* ops_len, uint8_t* codes, uint32_t codes_stride, void* stream) {
    if (n_m == 0) return 0;
    hipLaunchKernelGGL(k_out_codes, dim3(cdiv(n_m, 256)), dim3(256), 0, S_(stream), b, p, matches, n_m, ops, ops_words, ops_len, codes, codes_stride);
    return (int)hipGetLastError();
}

int gmk_out_deposit(float* cov, uint64_t bins, uint32_t bin_size, const GmDevMatch* matches, const GmDevPos* positions, const uint32_t* pos_match,
                    uint64_t n_p, const uint16_t* ops_len, const float* post, uint32_t max_span, float* nuc, const uint8_t* codes, uint32_t codes_stride,
                    void* stream) {
    if (n_p == 0 || max_span == 0) return 0;
    const uint32_t bins_per_place = (max_span + bin_size - 1) / bin_size + 1;          // a place may start in the middle of a bin
    const uint64_t total = n_p * bins_per_place;
    hipLaunchKernelGGL(k_out_deposit, dim3(cdiv(total, 256)), dim3(256), 0, S_(stream), cov, bins, bin_size, matches, positions, pos_match, n_p, ops_len,
                       post, bins_per_place, nuc, codes, codes_stride);
    return (int)hipGetLastError();
}
