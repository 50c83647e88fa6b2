// gm_pair.hip — k_vote_pair: the common case of k_vote_bucket (gm_bucket.hip) with TWO reads per wavefront.
//
// k_vote_bucket runs at the vector issue rate (profiles/r04_*: ~550 vector instructions per read, every one of them 64 lanes wide, the
// SIMDs ~95 % busy; prefetching the next read's words and cutting the filter pass' instructions moved the time by what they removed,
// not more).  A good part of those instructions is per-READ work that a whole wavefront executes for one read: the read's header,
// its k-mer codes, the record addresses, hit counts, routing (~220 of the 550), and the passes over the records' positions run with a
// third of the lanes holding a hit.  Here a wavefront takes two reads: a read is 32 lanes, a strand a QUARTER (16 lanes = one DPP row),
// a seed's record is fetched by 8 lanes as before, two seeds per strand and step.  The per-read instructions are shared by two reads
// and every pass over the positions serves four strands.
//
// Only the common case lives here: every k-mer at the regular positions occurs (and stays within -h), no record beyond its 28 inline
// positions, no early position, at most 384 hits per strand, at most 16 seeds per strand, ACGT only, at most GMP_LCAP second arrivals.
// A read with anything else is FLAGGED (one byte) and left untouched; k_pair_collect lists the flagged reads and k_vote_bucket - which
// knows how to walk again, follow the suffix array, hand over to the list / heavy kernels - takes exactly those.  Same candidates,
// same counters: the flagged reads are a few per thousand on an i.i.d. reference, more on repeats, and parity does not depend on how many.
//
// Votes as in k_vote_bucket: one returning LDS atomic per hit into a 16384-bit filter of its strand (slot = low bits of the window
// start), a second arrival's window start goes to the strand's key list unless it is the strand's latest key, one sweep per distinct key
// ORs the tag bits of the seeds that hold it: votes = popcount, NW step = its -k-th lowest bit (inc/align_seq2_raw.cpp:262-274,
// process_hits :28-40).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "gm_internal.h"
#include "gm_device.h"

#define GMP_LCAP 16                      // second-or-later arrivals listed per strand; more: the read is flagged
#define GMP_ECAP 384u                    // SA hits per strand voted on here
#define GMP_FWORDS 512                   // filter words per strand

struct GmPairLds {
    uint4 filt[4 * GMP_FWORDS / 4];      // [4 strands][512 words]
    uint32_t keys[4][GMP_LCAP];
    uint32_t tagm[4];
    uint32_t pad[12];
    // results of 16 consecutive pairs (= 64 read x strands, whose own candidate slots are one 1 KB stretch): written out together, whole
    // lines at a time - a 16-byte store per read x strand is a partial-line write each, and those were the most expensive thing left
    GmCand cbuf[64];
    uint32_t nebuf[64];
    uint16_t nsbuf[64];
};

__device__ __forceinline__ int gmp_lane_again() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ uint4 gmp_zero4() {        // (made here: a hoisted zero tuple is spilled, see gmb_zero4)
    uint4 z;
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z.x), "=v"(z.y), "=v"(z.z), "=v"(z.w));
    return z;
}
// the 16 bits of a wave mask that belong to this lane's quarter (qsh = 16 x quarter)
__device__ __forceinline__ uint32_t gmp_qbits(unsigned long long m, uint32_t qsh) { return (uint32_t)(m >> qsh) & 0xFFFFu; }

template <int STEPS>
__global__ void __launch_bounds__(64, 4) k_vote_pair(GmDevIndex ix, GmDevParams p, GmDevBatch b, uint8_t* fallback, const uint32_t chunk /* pairs per workgroup, a multiple of 16 */) {
    __shared__ GmPairLds S;
    const uint32_t m = (uint32_t)p.mer, jump = (uint32_t)p.jump, w2 = b.pack_w2;
    const uint32_t cmask = m >= 16u ? 0xFFFFFFFFu : ((1u << (2u * m)) - 1u);
    const uint32_t zero_code = cmask + 1u;
    const uint4* const bucket = reinterpret_cast<const uint4*>(p.bucket);
    const uint32_t npairs = (b.n + 1u) >> 1;
    const uint32_t pr0 = blockIdx.x * chunk, pend = pr0 + chunk < npairs ? pr0 + chunk : npairs;
    // Software pipeline over the CONSECUTIVE pairs of a workgroup (pairs pr0 .. pend - 1): while pair i is voted
    // on, the RECORDS of pair i + 1 are in flight (their codes come from the 2-bit words requested an iteration earlier) and the words of
    // pair i + 2 are requested - the vote phase's instructions and the two dependent HBM trips of a read no longer take turns.
    uint32_t n_hdr = 0, n_f0 = 0, n_f1 = 0;
    auto request_forms = [&](const uint32_t pr) {
        const int ln = gmp_lane_again();
        const uint32_t r = 2u * pr + ((uint32_t)ln >> 5), st = ((uint32_t)ln >> 4) & 1u, ireg = ((uint32_t)ln & 15u) * jump;
        const uint32_t oo = ireg + m <= 16u * w2 ? 2u * (16u * w2 - ireg - m) : 0u;
        n_hdr = 1u << 17; n_f0 = 0u; n_f1 = 0u;                                      // a read beyond the block: "nothing to do"
        if (pr < pend && r < b.n) {
            const uint32_t* const rw = b.pack + (size_t)r * b.pack_words;
            const uint32_t* const fm_ = rw + (st ? w2 + 2u : 1u);
            n_hdr = rw[0]; n_f0 = fm_[oo >> 5]; n_f1 = fm_[(oo >> 5) + 1u];
        }
    };
    // state of the pair whose records are in flight (N) / being voted on (C)
    uint32_t N_hdr = 0, N_ns = 0; bool N_fb = false; uint4 N_rc[STEPS]; uint4 N_hd4 = make_uint4(0u, 0u, 0u, 0u);
    // codes of the pair from its words (which must have been requested before), its record loads, and the request of the words two pairs on
    auto issue = [&](const uint32_t prn) {
        const int lane = gmp_lane_again();
        const uint32_t sd = ((uint32_t)lane >> 4) & 1u, qt = (uint32_t)lane >> 4, jj = (uint32_t)lane & 15u, g = ((uint32_t)lane >> 3) & 1u,
                       q = (uint32_t)lane & 7u, qbase = (uint32_t)lane & 48u, qsh = 16u * qt;
        const uint32_t hdr = n_hdr, f0 = n_f0, f1 = n_f1;
        const uint32_t L = hdr & 0xFFFFu;
        const bool dead = ((hdr >> 17) & 1u) != 0u;                                  // status != 0 (or no read): nothing to look up
        const bool has_n = !dead && ((hdr >> 16) & 1u) != 0u;                         // a base that is not ACGT: the 2-bit forms cannot say where
        const bool strand_on = sd ? (p.neg_strand != 0) : (p.pos_strand != 0);
        const uint32_t last = L - m, i_reg = jj * jump;
        const bool act = !dead && !has_n && strand_on && i_reg < last && i_reg + m <= 16u * w2;
        const uint32_t o = i_reg + m <= 16u * w2 ? 2u * (16u * w2 - i_reg - m) : 0u;
        const uint32_t code = act ? (uint32_t)((((unsigned long long)f1 << 32) | f0) >> (o & 31u)) & cmask : 0u;
        const uint32_t ns_q = (uint32_t)__popc(gmp_qbits(__builtin_amdgcn_ballot_w64(act), qsh));        // regular positions 0, jump, .. < last
        N_fb = has_n || (strand_on && !dead && (ns_q > 2u * STEPS || (jj == 15u && act && i_reg + jump < last)));       // more seeds than this form holds
        N_hdr = hdr; N_ns = ns_q;
        // the record of seed 2 st + g of the strand: 8 lanes, one 16-byte load each; lanes q < STEPS also fetch lane 0's part of the record of
        // seed 2 q + g - all headers of a strand are looked at once
        const int bp_base = (int)((qbase + g) << 2);
        const uint32_t hslot = 2u * q + g;
        const uint32_t hcode = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((qbase + hslot) << 2), (int)code);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const uint32_t cdv = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_base + 8 * st, (int)code);
            N_rc[st] = bucket[(size_t)((2u * (uint32_t)st + g < ns_q && q != 0u) ? cdv : zero_code) * 8u + q];
        }
        N_hd4 = bucket[(size_t)((q < (uint32_t)STEPS && hslot < ns_q) ? hcode : zero_code) * 8u];
        // (last: the words just consumed are dead by now, the new ones land in their registers - requested earlier, the loop-carried copy
        // of the old values makes the wavefront wait for the new loads right away)
        request_forms(prn + 1u);
    };
    if (pr0 < pend) { request_forms(pr0); issue(pr0); }
    for (uint32_t pr = pr0; pr < pend; ++pr) {
        const int lane = gmp_lane_again();
        const uint32_t rd = (uint32_t)lane >> 5, sd = ((uint32_t)lane >> 4) & 1u, qt = (uint32_t)lane >> 4, jj = (uint32_t)lane & 15u, g = ((uint32_t)lane >> 3) & 1u,
                       q = (uint32_t)lane & 7u, qbase = (uint32_t)lane & 48u, qsh = 16u * qt;
        const uint32_t r = 2u * pr + rd, rs = 2u * r + sd;
        // this pair's state out of the pipeline registers, then the next pair's loads go out
        const uint32_t hdr = N_hdr, ns_q = N_ns;
        bool fb = N_fb;
        uint4 rc[STEPS];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) rc[st] = N_rc[st];
        const uint4 hd4 = N_hd4;
        issue(pr + 1u);                               // (unconditional: beyond the last pair it loads the zero record - a branch here makes the compiler copy the
                                                      //  freshly loaded words at the join, i.e. wait for them at once)
        {
            const uint4 z = gmp_zero4();
#pragma unroll
            for (int k = 0; k < 8; ++k) S.filt[lane + 64 * k] = z;
            if (lane < (int)((sizeof(S.keys) + sizeof(S.tagm) + sizeof(S.pad)) / 16)) reinterpret_cast<uint4*>(&S.keys[0][0])[lane] = z;
            if ((pr & 15u) == 0u) { reinterpret_cast<uint4*>(&S.cbuf[0])[lane] = z; S.nebuf[lane] = 0u; S.nsbuf[lane] = 0; }
        }
        do {                                              // (the pair's work; `break` = done with it - the group's results are flushed behind the block)
        const bool rv = r < b.n;
        const bool dead = ((hdr >> 17) & 1u) != 0u;
        uint32_t of1[STEPS];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) of1[st] = (2u * (uint32_t)st + g) * jump + 1u;
        const uint32_t hslot = 2u * q + g;
        const bool hv = q < (uint32_t)STEPS && hslot < ns_q;
        // ---- headers: a k-mer that does not occur, exceeds -h, keeps its hits in the suffix array or has an early position: not here ----
        const uint32_t hc = hd4.x & 0xFFFFu;
        fb |= hv && ((hd4.x & 0xE0000000u) != 0u || (p.hcap > 0 && hc > (uint32_t)p.hcap));
        uint32_t c_lane = hv ? hc : 0u;
        {   // hits of the strand: sum over its 16 lanes (one DPP row)
            uint32_t v = c_lane;
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
            c_lane = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((qbase + 15u) << 2), (int)v);
        }
        const uint32_t E_q = c_lane;
        fb |= E_q > GMP_ECAP || E_q > p.heavy_min;
        // a flagged read leaves with nothing written but its flag
        unsigned long long fbm = __builtin_amdgcn_ballot_w64(fb);
        bool fb_read = (rd ? (uint32_t)(fbm >> 32) : (uint32_t)fbm) != 0u;
        if (rv && !fb_read && jj == 0u) { S.nsbuf[rs & 63u] = (uint16_t)(dead ? 0u : ns_q); S.nebuf[rs & 63u] = dead ? 0u : E_q; }
        bool vote = rv && !dead && !fb_read && ns_q != 0u;
        if (__builtin_amdgcn_ballot_w64(vote) == 0ull) {
            if (rv && fb_read && jj == 0u && sd == 0u) fallback[r] = 1;
            break;
        }
        if (p.dbg & 32768) break;                      // (GM_DBG 32768: timing experiment, stop before the votes)
        __syncthreads();                                  // zeroed structures (one wave: a wait, not a rendezvous)
        // ---- pass 1: the filter ----
        uint32_t lc_q = 0, known = 0;
        const uint32_t filt_off = (uint32_t)(size_t)(reinterpret_cast<unsigned char*>(&S.filt[0]) - reinterpret_cast<unsigned char*>(&S)) + qt * (GMP_FWORDS * 4u);
        const uint32_t qm_lo = rd ? 0u : (0xFFFFu << (16u * sd)), qm_hi = rd ? (0xFFFFu << (16u * sd)) : 0u;       // this quarter's lanes in the two mask halves
        uint32_t* const keys = &S.keys[qt][0];
        auto arrive = [&](const uint32_t v, const uint32_t bp, uint32_t& bit) -> uint32_t {
            uint32_t old = 0;
            bit = 1u << (bp & 31u);
            if (vote && v != 0u) {
                uint32_t addr;
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(addr) : "v"(bp >> 3), "s"(0x7FCu), "v"(filt_off));
                old = atomicOr(reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(&S) + addr), bit);
            }
            return old;
        };
        auto settle = [&](const uint32_t bp, const uint32_t old, const uint32_t bit) {
            const unsigned long long dm = __builtin_amdgcn_uicmp(old & bit, 0u, 33 /* != */) & __builtin_amdgcn_uicmp(bp, known, 33 /* != */);
            if (dm != 0ull) {                             // rare: once per distinct window start that is met again
                const bool dup = (old & bit) != 0u && bp != known;
                const uint32_t dlo = (uint32_t)dm & qm_lo, dhi = (uint32_t)(dm >> 32) & qm_hi;
                const uint32_t at = lc_q + __builtin_amdgcn_mbcnt_hi(dhi, __builtin_amdgcn_mbcnt_lo(dlo, 0u));
                if (dup && at < GMP_LCAP) keys[at] = bp;
                lc_q += (uint32_t)__popc(dlo) + (uint32_t)__popc(dhi);
                const uint32_t qb = dlo | dhi;            // (one of the two is 0)
                const uint32_t src = (rd << 5) | (qb ? (uint32_t)__builtin_ctz(qb) : 0u);
                const uint32_t nk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)bp);
                if (qb != 0u) known = nk;
            }
        };
        {
            uint32_t bx[STEPS], by[STEPS], ox[STEPS], oy[STEPS], tx[STEPS], ty[STEPS];
#pragma unroll
            for (int st = 0; st < STEPS; ++st) { bx[st] = rc[st].x - of1[st]; by[st] = rc[st].y - of1[st]; }
#pragma unroll
            for (int st = 0; st < STEPS; ++st) { ox[st] = arrive(rc[st].x, bx[st], tx[st]); oy[st] = arrive(rc[st].y, by[st], ty[st]); }
#pragma unroll
            for (int st = 0; st < STEPS; ++st) { settle(bx[st], ox[st], tx[st]); settle(by[st], oy[st], ty[st]); }
        }
        {   // register 2 (positions 14 .. 20: a fifth of the records has some) of every step in ONE round trip as well - a branch and a
            // dependent LDS trip per step is what a wavefront's iteration time is made of here, not the instructions
            // (is there any such register in the wavefront: the OR of the steps' words, one compare each - not a compare per step)
            uint32_t orz = 0u, orw = 0u;
#pragma unroll
            for (int st = 0; st < STEPS; ++st) { orz |= rc[st].z; orw |= rc[st].w; }
            const unsigned long long anyz = __builtin_amdgcn_uicmp(vote ? orz : 0u, 0u, 33), anyw = __builtin_amdgcn_uicmp(vote ? orw : 0u, 0u, 33);
            if (anyz != 0ull) {
                uint32_t bz[STEPS], oz[STEPS], tz[STEPS];
#pragma unroll
                for (int st = 0; st < STEPS; ++st) bz[st] = rc[st].z - of1[st];
#pragma unroll
                for (int st = 0; st < STEPS; ++st) oz[st] = arrive(rc[st].z, bz[st], tz[st]);
#pragma unroll
                for (int st = 0; st < STEPS; ++st) settle(bz[st], oz[st], tz[st]);
            }
            if (anyw != 0ull) {
#pragma unroll
                for (int st = 0; st < STEPS; ++st)
                    if (__builtin_amdgcn_uicmp(vote ? rc[st].w : 0u, 0u, 33) != 0ull) { const uint32_t bp = rc[st].w - of1[st]; uint32_t bit; const uint32_t old = arrive(rc[st].w, bp, bit); settle(bp, old, bit); }
            }
        }
        // more second arrivals than the key list holds (a repeat-rich read): flagged after all
        {
            const unsigned long long om = __builtin_amdgcn_ballot_w64(vote && lc_q > GMP_LCAP);
            if (om != 0ull) {
                const bool over_read = (rd ? (uint32_t)(om >> 32) : (uint32_t)om) != 0u;
                if (over_read) { fb_read = true; vote = false; lc_q = 0; }
            }
        }
        if (rv && fb_read && jj == 0u && sd == 0u) fallback[r] = 1;
        if (__builtin_amdgcn_ballot_w64(vote && lc_q != 0u) == 0ull || (p.dbg & 16384)) break;      // (GM_DBG 16384: timing experiment, no sweeps)       // no second arrival anywhere: no window start with two votes
        __syncthreads();
        // ---- sweeps: one per distinct key of a strand's list (all four strands in the same instructions) ----
        uint32_t mykey = (vote && jj < lc_q) ? keys[jj] : 0u;
        uint32_t n_em = 0;
        GmCand first_c; first_c.rs = rs; first_c.b = 0; first_c.step = 0; first_c.flags = 4; first_c.pad = 0; first_c.score = 0.0f;
        const uint32_t shard = rs & (GM_NSHARD - 1);
        for (;;) {
            const unsigned long long rem = __builtin_amdgcn_ballot_w64(mykey != 0u);
            if (rem == 0ull) break;
            const uint32_t rq = gmp_qbits(rem, qsh);
            const uint32_t src = qbase | (rq ? (uint32_t)__builtin_ctz(rq) : 0u);
            const uint32_t kk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)mykey);
            const uint32_t key = rq ? kk : 0u;            // 0: this strand has no key left (no hit has window start 0 here: early positions are flagged)
            if (mykey == key) mykey = 0u;
            // the seeds that hold the key: every lane collects the tag bits of its own matches, the strand's 16 lanes (one DPP row) OR them
            // together - no LDS atomics, no barrier, no read-back on the wavefront's path
            uint32_t tl = 0;
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                bool mt = (rc[st].x - of1[st]) == key;
                mt |= (rc[st].y - of1[st]) == key;
                mt |= (rc[st].z - of1[st]) == key;
                if (__builtin_amdgcn_uicmp(rc[st].w, 0u, 33) != 0ull) mt |= (rc[st].w - of1[st]) == key;
                tl |= mt ? (1u << (2u * (uint32_t)st + g)) : 0u;
            }
            tl |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tl, 0x111, 0xF, 0xF, true);
            tl |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tl, 0x112, 0xF, 0xF, true);
            tl |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tl, 0x114, 0xF, 0xF, true);
            tl |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tl, 0x118, 0xF, 0xF, true);
            const uint32_t tm = key != 0u ? (uint32_t)__builtin_amdgcn_ds_bpermute((int)((qbase + 15u) << 2), (int)tl) : 0u;
            const uint32_t votes = (uint32_t)__popc(tm);
            const bool em = jj == 0u && key != 0u && votes >= (uint32_t)p.kmin;
            if (__builtin_amdgcn_ballot_w64(em) != 0ull) {
                uint32_t step;
                if (p.nw) { uint32_t mm = tm; for (int rr = 1; rr < p.kmin; ++rr) mm &= mm - 1; step = mm ? (uint32_t)(__ffs((int)mm) - 1) : 0u; }
                else step = votes;
                GmCand c;
                c.rs = rs; c.b = key; c.step = (uint16_t)step; c.flags = 4; c.pad = 0; c.score = 0.0f;
                if (em && b.fixed_cands != nullptr && n_em == 0u) first_c = c;
                else if (em && b.fixed_cands != nullptr && n_em < GM_FIXED_C) b.fixed_cands[GM_FIXED_AT(b, rs, n_em)] = c;
                else if (em) {
                    const uint32_t at = atomicAdd(&b.shard_cnt[shard * GM_SHARD_STRIDE], 1u);
                    if (at < b.cand_region) b.cands[(size_t)shard * b.cand_region + at] = c;
                }
                if (em) ++n_em;
            }
        }
        if (jj == 0u && b.fixed_cands != nullptr && n_em != 0u) {
            first_c.pad = (uint8_t)(n_em < GM_FIXED_C ? n_em : GM_FIXED_C);
            first_c.score = __uint_as_float(b.fixed_epoch);       // k_cand_gather takes slots stamped with this launch only
            S.cbuf[rs & 63u] = first_c;
        }
        } while (0);
        // ---- the group's results: slot 0 of 64 read x strands is one 1 KB stretch (GM_FIXED_AT), their seed / hit counts 128 / 256 bytes ----
        if ((pr & 15u) == 15u || pr + 1u == pend) {
            __syncthreads();
            const uint32_t rs_l = ((pr >> 4) << 6) + (uint32_t)lane;
            if (rs_l < 2u * b.n && !(p.dbg & 4096)) { b.n_seeds[rs_l] = S.nsbuf[lane]; b.n_entries[rs_l] = S.nebuf[lane]; }
            if (rs_l < 2u * b.n && b.fixed_cands != nullptr && !(p.dbg & 8192)) b.fixed_cands[GM_FIXED_AT(b, rs_l, 0u)] = S.cbuf[lane];
        }
    }
}

// the flagged reads, as a list for k_vote_bucket (any order).  16 flags per thread, 4096 per workgroup, ONE atomic per workgroup that has
// any: the first form (a thread per read, an atomic per wavefront) spent 116 us at 10 M reads on 150 000 atomics to one address
__global__ void __launch_bounds__(256) k_pair_collect(const uint8_t* fallback, uint32_t n, uint32_t* list, uint32_t* n_list) {
    __shared__ uint32_t s_w[4], s_base;
    const uint32_t r0 = blockIdx.x * 4096u + threadIdx.x * 16u;
    uint32_t w[4] = { 0u, 0u, 0u, 0u };
    if (r0 < n) { const uint4 v = *reinterpret_cast<const uint4*>(fallback + r0); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }      // (the buffer is padded: gm_api.cpp)
    uint32_t mask = 0;                                   // bit t: read r0 + t is flagged
#pragma unroll
    for (uint32_t t = 0; t < 16; ++t) if (((w[t >> 2] >> ((t & 3u) << 3)) & 255u) != 0u && r0 + t < n) mask |= 1u << t;
    const uint32_t c = (uint32_t)__popc(mask);
    const uint32_t incl = gm_wave_scan_incl(c);
    const int lane = gm_lane(), wave = (int)(threadIdx.x >> 6);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - c, total = 0;
    for (int q = 0; q < 4; ++q) { if (q < wave) pre += s_w[q]; total += s_w[q]; }
    if (total == 0u) return;                             // workgroup-uniform
    if (threadIdx.x == 0) s_base = atomicAdd(n_list, total);
    __syncthreads();
    uint32_t at = s_base + pre;
    while (mask) { const uint32_t t = (uint32_t)__builtin_ctz(mask); mask &= mask - 1u; list[at++] = r0 + t; }
}

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

// max_reg = seeds per strand the longest read needs; fallback: n bytes, zeroed by the caller; list / n_list: n words + 1 counter (zeroed)
int gmk_vote_pair(const GmDevIndex& ix, const GmDevParams& p, const GmDevBatch& b, uint32_t max_reg, uint8_t* fallback, uint32_t* list, uint32_t* n_list, void* stream) {
    if (b.n == 0) return 0;
    if (max_reg > 16 || p.bucket_ctx || p.mer != p.bucket_T) return (int)hipErrorInvalidValue;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    (void)cus;
    // a workgroup (one wavefront) takes `chunk` consecutive pairs: long enough that the pipeline's fill (one pair) does not matter, short enough
    // that small blocks still fill the chip
    const uint32_t npairs = (b.n + 1u) / 2u;
    uint32_t chunk = (uint32_t)std::max<long long>(16, gm_opt_ll("GM_PAIR_CHUNK", 32)) / 16u * 16u;
    while (chunk > 16u && (npairs + chunk - 1u) / chunk < 4096u) chunk -= 16u;
    const dim3 grid((npairs + chunk - 1u) / chunk), blk(64);
    if (max_reg <= 8) hipLaunchKernelGGL((k_vote_pair<4>), grid, blk, 0, S_(stream), ix, p, b, fallback, chunk);
    else if (max_reg <= 14) hipLaunchKernelGGL((k_vote_pair<7>), grid, blk, 0, S_(stream), ix, p, b, fallback, chunk);
    else hipLaunchKernelGGL((k_vote_pair<8>), grid, blk, 0, S_(stream), ix, p, b, fallback, chunk);
    hipLaunchKernelGGL(k_pair_collect, dim3((b.n + 4095u) / 4096u), dim3(256), 0, S_(stream), fallback, b.n, list, n_list);
    return (int)hipGetLastError();
}
