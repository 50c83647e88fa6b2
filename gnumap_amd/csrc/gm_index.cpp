// gm_index.cpp — host side of the index: reading <fa>.gnumap.{bwt,sa,pac,ann,amb} and building them.
//
// File formats (byte-compatible with the reference, so either side can read the other's index):
//   .bwt  primary u64, L2[1..4] u64, then u32 words of the occ-interleaved BWT   (bwt_dump_bwt src/bwt.c:385-393,
//         bwt_bwtupdate_core src/bwtindex.c:128-150)
//   .sa   primary, L2[1..4], sa_intv, seq_len (u64 each), then SA[32], SA[64], ... as u64       (bwt_dump_sa src/bwt.c:395-407)
//   .pac  2 bit/base MSB first + padding byte(s), last byte = l_pac % 4               (bns_fasta2bntseq src/bntseq.c:296-310)
//   .ann/.amb  text                                                                    (bns_dump src/bntseq.c:66-96)
// The BWT itself is defined mathematically (suffix array of the forward strand + '$'), so a from-scratch SA-IS
// gives the same bytes as the reference's IS / BWT-SW builders.
#include "gm_host.h"
#include <zlib.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>

// ------------------------------------------------------------------------------------------------
// loading
// ------------------------------------------------------------------------------------------------
static bool slurp(const std::string& fn, std::vector<uint8_t>& out) {
    FILE* f = fopen(fn.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize((size_t)n);
    bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

bool gm_host_index_files_exist(const std::string& fa) {
    // bwa_idx_infer_prefix src/GenomeBwt.cpp:59-88 (the ".64" variant is never written by this build)
    FILE* f = fopen((fa + ".gnumap.bwt").c_str(), "rb");
    if (!f) return false;
    fclose(f);
    return true;
}

int gm_host_index_load(const std::string& fa, GmHostIndex& ix, std::string& err) {
    std::vector<uint8_t> raw;
    if (!slurp(fa + ".gnumap.bwt", raw) || raw.size() < 40) { err = "cannot read " + fa + ".gnumap.bwt"; return GM_E_IO; }
    memcpy(&ix.primary, raw.data(), 8);
    ix.L2[0] = 0;
    memcpy(&ix.L2[1], raw.data() + 8, 32);
    ix.seq_len = ix.L2[4];
    size_t words = (raw.size() - 40) >> 2;
    ix.bwt.assign(words + 16, 0);                       // 64 bytes of slack: occ reads whole 64-byte blocks
    memcpy(ix.bwt.data(), raw.data() + 40, words * 4);
    ix.bwt_words = words;
    if (!slurp(fa + ".gnumap.sa", raw) || raw.size() < 56) { err = "cannot read " + fa + ".gnumap.sa"; return GM_E_IO; }
    uint64_t primary, seq_len, intv;
    memcpy(&primary, raw.data(), 8);
    memcpy(&intv, raw.data() + 40, 8);
    memcpy(&seq_len, raw.data() + 48, 8);
    if (primary != ix.primary) { err = "SA-BWT inconsistency: primary is not the same."; return GM_E_IO; }      // bwt.c:429
    if (seq_len != ix.seq_len) { err = "SA-BWT inconsistency: seq_len is not the same."; return GM_E_IO; }      // bwt.c:433
    if (intv == 0 || (intv & (intv - 1))) { err = "SA sample interval is not a power of 2."; return GM_E_IO; }
    ix.sa_intv = (uint32_t)intv;
    ix.n_sa = (ix.seq_len + intv) / intv;
    if (raw.size() - 56 < (ix.n_sa - 1) * 8) { err = "truncated .sa file"; return GM_E_IO; }
    ix.sa.assign(ix.n_sa, 0);
    ix.sa[0] = (uint64_t)-1;
    memcpy(ix.sa.data() + 1, raw.data() + 56, (ix.n_sa - 1) * 8);
    // .ann
    FILE* f = fopen((fa + ".gnumap.ann").c_str(), "r");
    if (!f) { err = "cannot read " + fa + ".gnumap.ann"; return GM_E_IO; }
    long long xx; int n_seqs; unsigned seed;
    if (fscanf(f, "%lld%d%u", &xx, &n_seqs, &seed) != 3 || n_seqs <= 0) { fclose(f); err = "Parse error reading .ann"; return GM_E_IO; }
    ix.l_pac = (uint64_t)xx;
    ix.contigs.resize((size_t)n_seqs);
    for (int i = 0; i < n_seqs; ++i) {
        unsigned gi; char str[8192]; int c, len, nambs;
        if (fscanf(f, "%u%8191s", &gi, str) != 2) { fclose(f); err = "Parse error reading .ann"; return GM_E_IO; }
        ix.contigs[i].name = str;
        while ((c = fgetc(f)) != '\n' && c != EOF) {}
        if (fscanf(f, "%lld%d%d", &xx, &len, &nambs) != 3) { fclose(f); err = "Parse error reading .ann"; return GM_E_IO; }
        ix.contigs[i].offset = (uint64_t)xx;
        ix.contigs[i].len = (uint32_t)len;
    }
    fclose(f);
    if (!slurp(fa + ".gnumap.pac", raw) || raw.size() < ix.l_pac / 4 + 1) { err = "cannot read " + fa + ".gnumap.pac"; return GM_E_IO; }
    ix.pac.assign(raw.begin(), raw.end());
    ix.pac.resize(ix.pac.size() + 64, 0);
    if (ix.l_pac != ix.seq_len) { err = "index holds both strands; GNUMAP indexes the forward strand only"; return GM_E_IO; }
    return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// suffix array by induced sorting (SA-IS, Nong/Zhang/Chan 2009), templated on the index width so that
// references beyond 2^31 symbols use 64-bit indices
// ------------------------------------------------------------------------------------------------
namespace {

template <class Int, class Sym>
struct Sais {
    static inline bool tget(const std::vector<uint64_t>& t, Int i) { return (t[(size_t)i >> 6] >> ((size_t)i & 63)) & 1; }
    static inline void tset(std::vector<uint64_t>& t, Int i, bool v) {
        if (v) t[(size_t)i >> 6] |= 1ull << ((size_t)i & 63); else t[(size_t)i >> 6] &= ~(1ull << ((size_t)i & 63));
    }
    static inline bool is_lms(const std::vector<uint64_t>& t, Int i) { return i > 0 && tget(t, i) && !tget(t, i - 1); }

    static void buckets(const Sym* s, std::vector<Int>& bkt, Int n, Int K, bool end) {
        std::fill(bkt.begin(), bkt.end(), (Int)0);
        for (Int i = 0; i < n; ++i) bkt[(size_t)s[i]]++;
        Int sum = 0;
        for (Int i = 0; i < K; ++i) { sum += bkt[(size_t)i]; bkt[(size_t)i] = end ? sum : sum - bkt[(size_t)i]; }
    }
    static void induce_l(const std::vector<uint64_t>& t, Int* SA, const Sym* s, std::vector<Int>& bkt, Int n, Int K) {
        buckets(s, bkt, n, K, false);
        for (Int i = 0; i < n; ++i) {
            Int j = SA[i] - 1;
            if (SA[i] > 0 && !tget(t, j)) SA[(size_t)bkt[(size_t)s[j]]++] = j;
        }
    }
    static void induce_s(const std::vector<uint64_t>& t, Int* SA, const Sym* s, std::vector<Int>& bkt, Int n, Int K) {
        buckets(s, bkt, n, K, true);
        for (Int i = n - 1; i >= 0; --i) {
            Int j = SA[i] - 1;
            if (SA[i] > 0 && tget(t, j)) SA[(size_t)--bkt[(size_t)s[j]]] = j;
        }
    }

    // s[0..n-1] with s[n-1] = 0 the unique smallest symbol; symbols in [0,K)
    static void run(const Sym* s, Int* SA, Int n, Int K) {
        std::vector<uint64_t> t(((size_t)n >> 6) + 1, 0);
        tset(t, n - 1, true);
        if (n >= 2) tset(t, n - 2, false);
        for (Int i = n - 3; i >= 0; --i)
            tset(t, i, s[i] < s[i + 1] || (s[i] == s[i + 1] && tget(t, i + 1)));
        std::vector<Int> bkt((size_t)K);
        // stage 1: sort the LMS substrings
        buckets(s, bkt, n, K, true);
        for (Int i = 0; i < n; ++i) SA[i] = -1;
        for (Int i = 1; i < n; ++i) if (is_lms(t, i)) SA[(size_t)--bkt[(size_t)s[i]]] = i;
        induce_l(t, SA, s, bkt, n, K);
        induce_s(t, SA, s, bkt, n, K);
        Int n1 = 0;
        for (Int i = 0; i < n; ++i) if (is_lms(t, SA[i])) SA[n1++] = SA[i];
        for (Int i = n1; i < n; ++i) SA[i] = -1;
        Int name = 0, prev = -1;
        for (Int i = 0; i < n1; ++i) {
            Int pos = SA[i];
            bool diff = false;
            for (Int d = 0; d < n; ++d) {
                if (prev == -1 || s[pos + d] != s[prev + d] || tget(t, pos + d) != tget(t, prev + d)) { diff = true; break; }
                else if (d > 0 && (is_lms(t, pos + d) || is_lms(t, prev + d))) break;
            }
            if (diff) { ++name; prev = pos; }
            SA[(size_t)(n1 + pos / 2)] = name - 1;
        }
        for (Int i = n - 1, j = n - 1; i >= n1; --i) if (SA[i] >= 0) SA[j--] = SA[i];
        // stage 2: the reduced problem
        Int* SA1 = SA; Int* s1 = SA + n - n1;
        if (name < n1) Sais<Int, Int>::run(s1, SA1, n1, name);
        else for (Int i = 0; i < n1; ++i) SA1[(size_t)s1[i]] = i;
        // stage 3: induce the final order
        buckets(s, bkt, n, K, true);
        for (Int i = 1, j = 0; i < n; ++i) if (is_lms(t, i)) s1[j++] = i;
        for (Int i = 0; i < n1; ++i) SA1[i] = s1[(size_t)SA1[i]];
        for (Int i = n1; i < n; ++i) SA[i] = -1;
        for (Int i = n1 - 1; i >= 0; --i) { Int j = SA[i]; SA[i] = -1; SA[(size_t)--bkt[(size_t)s[j]]] = j; }
        induce_l(t, SA, s, bkt, n, K);
        induce_s(t, SA, s, bkt, n, K);
    }
};

// glibc lrand48 (the reference fills N with lrand48()&3 after srand48(11), src/bntseq.c:261,290-291)
struct Rand48 {
    uint64_t x;
    explicit Rand48(uint32_t seed) : x(((uint64_t)seed << 16) | 0x330E) {}
    uint32_t next() { x = (0x5DEECE66DULL * x + 0xB) & ((1ULL << 48) - 1); return (uint32_t)(x >> 17); }
};

inline int nt4(unsigned char c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

struct Hole { uint64_t offset; uint32_t len; char amb; };
struct Ann { std::string name, anno; uint64_t offset; uint32_t len; uint32_t n_ambs; };

// SA stage on the host: SA-IS, then the BWT with '$' removed (is_bwt src/is.c:208-223, bwt_pac2bwt src/bwtindex.c:60-104) and the
// rank-sampled SA (bwt_cal_sa src/bwt.c:62-84: sa[isa/intv] = SA[isa]; sa[0] is not stored)
template <class Int>
int host_sa_stage(std::vector<uint8_t>& codes /* n symbols 1..4 + sentinel 0 */, uint64_t intv, std::vector<uint32_t>& plain, uint64_t& primary,
                  std::vector<uint64_t>& samples, std::string& err) {
    const uint64_t n = codes.size() - 1;
    std::vector<Int> SA;
    try { SA.resize(n + 1); } catch (...) { err = "out of memory for the suffix array"; return GM_E_NOMEM; }
    Sais<Int, uint8_t>::run(codes.data(), SA.data(), (Int)(n + 1), (Int)5);
    plain.assign((n + 15) >> 4, 0);
    primary = 0;
    uint64_t o = 0;
    for (uint64_t i = 0; i <= n; ++i) {
        uint64_t sa = (uint64_t)SA[i];
        if (sa == 0) { primary = i; continue; }
        uint32_t c = (uint32_t)codes[sa - 1] - 1;
        plain[o >> 4] |= c << ((15 - (o & 15)) << 1);
        ++o;
    }
    const uint64_t n_sa = (n + intv) / intv;
    samples.resize(n_sa - 1);
    for (uint64_t s = 1; s < n_sa; ++s) samples[s - 1] = (uint64_t)SA[s * intv];
    return GM_OK;
}

// .bwt and .sa from the SA stage's products (the same bytes whichever device ran the stage)
int write_bwt_sa(const std::string& fa, const std::vector<uint8_t>& codes, uint64_t intv, const std::vector<uint32_t>& plain, uint64_t primary,
                 const std::vector<uint64_t>& samples, std::string& err) {
    const uint64_t n = codes.size() - 1;
    uint64_t L2[5] = { 0, 0, 0, 0, 0 };
    const uint64_t n_words = (n + 15) >> 4;
    // interleave the occurrence counts every 128 bases (bwt_bwtupdate_core src/bwtindex.c:128-150)
    const uint64_t n_occ = (n + 127) / 128 + 1;
    std::vector<uint32_t> bwt(n_words + n_occ * 8, 0);
    {
        uint64_t c[4] = { 0, 0, 0, 0 }, k = 0;
        const uint64_t full = n >> 4;                                  // words with 16 real symbols
        for (uint64_t w = 0; w < n_words; ++w) {
            if ((w & 7) == 0) { memcpy(&bwt[k], c, 32); k += 8; }
            const uint32_t x = plain[w];
            bwt[k++] = x;
            if (w < full) {                                            // 2-bit symbol histogram of one word
                const uint32_t lo = x & 0x55555555u, hi = (x >> 1) & 0x55555555u;
                const uint32_t c3 = (uint32_t)__builtin_popcount(hi & lo), c2 = (uint32_t)__builtin_popcount(hi & ~lo),
                               c1 = (uint32_t)__builtin_popcount(lo & ~hi);
                c[3] += c3; c[2] += c2; c[1] += c1; c[0] += 16 - c1 - c2 - c3;
            } else {
                for (uint64_t i = w << 4; i < n; ++i) ++c[(x >> ((~i & 15) << 1)) & 3];
            }
        }
        memcpy(&bwt[k], c, 32);
        k += 8;
        if (k != bwt.size()) { err = "inconsistent bwt_size"; return GM_E_IO; }
        for (int i = 1; i <= 4; ++i) L2[i] = L2[i - 1] + c[i - 1];     // the BWT is a permutation of the text: its totals are the text's
        if (L2[4] != n) { err = "inconsistent symbol counts"; return GM_E_IO; }
    }
    FILE* f = fopen((fa + ".gnumap.bwt").c_str(), "wb");
    if (!f) { err = "cannot write " + fa + ".gnumap.bwt"; return GM_E_IO; }
    fwrite(&primary, 8, 1, f); fwrite(&L2[1], 8, 4, f); fwrite(bwt.data(), 4, bwt.size(), f);
    fclose(f);
    f = fopen((fa + ".gnumap.sa").c_str(), "wb");
    if (!f) { err = "cannot write " + fa + ".gnumap.sa"; return GM_E_IO; }
    fwrite(&primary, 8, 1, f); fwrite(&L2[1], 8, 4, f); fwrite(&intv, 8, 1, f); fwrite(&n, 8, 1, f);
    if (!samples.empty()) fwrite(samples.data(), 8, samples.size(), f);
    fclose(f);
    return GM_OK;
}

}  // namespace

int gm_host_index_build(const std::string& fa, int where, int device_id, std::string& err) {
    const bool timing = getenv("GM_TIMING") && atoi(getenv("GM_TIMING"));            // GM_TIMING=1: stage times on stderr
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        auto t = std::chrono::steady_clock::now();
        if (timing) fprintf(stderr, "[gm_timing] index build: %s %.2f s\n", what, std::chrono::duration<double>(t - t_last).count());
        t_last = t;
    };
    gzFile fp = gzopen(fa.c_str(), "r");
    if (!fp) { err = "cannot open " + fa; return GM_E_IO; }
    // FASTA -> 2-bit codes (+1), annotations and holes: bns_fasta2bntseq/add1 src/bntseq.c:227-328 with for_only = 1
    std::vector<uint8_t> codes;
    std::vector<Ann> anns;
    std::vector<Hole> holes;
    Rand48 rng(11);
    {
        // byte-stream state machine over large gzread blocks: a '>' at the start of a line opens a header line (name = up to the
        // first white space, comment = the rest: kseq), every other byte after the first header except '\n' / '\r' is a base (kseq keeps
        // blanks inside a sequence line, so they become ambiguity holes like any non-ACGT byte)
        static const struct Lut { uint8_t v[256]; Lut() { for (int c = 0; c < 256; ++c) v[c] = c == '\r' ? 5 : (uint8_t)nt4((unsigned char)c); } } lut;
        std::vector<unsigned char> buf(8u << 20);
        std::string header;
        bool in_header = false, line_start = true, in_seq = false;
        int lasts = 0;
        auto take_header = [&](std::string& ln) {
            if (!ln.empty() && ln.back() == '\r') ln.pop_back();
            Ann a;
            size_t e = 0;
            while (e < ln.size() && !isspace((unsigned char)ln[e])) ++e;
            a.name = ln.substr(0, e);
            size_t c = e;
            if (c < ln.size()) ++c;                                       // kseq: the comment is what follows the first separator
            a.anno = c < ln.size() ? ln.substr(c) : std::string();
            if (a.anno.empty()) a.anno = "(null)";                        // bntseq.c:238
            a.offset = codes.size(); a.len = 0; a.n_ambs = 0;
            anns.push_back(a);
            lasts = 0;
            in_seq = true;
            ln.clear();
        };
        int got;
        while ((got = gzread(fp, buf.data(), (unsigned)buf.size())) > 0) {
            const size_t base = codes.size();
            codes.resize(base + (size_t)got);                             // at most one code per input byte
            uint8_t* out = codes.data() + base;
            for (int i = 0; i < got; ++i) {
                const unsigned char ch = buf[i];
                if (in_header) {
                    if (ch == '\n') { codes.resize((size_t)(out - codes.data())); take_header(header); in_header = false; line_start = true;
                                      const size_t at = codes.size(); codes.resize(at + (size_t)(got - i)); out = codes.data() + at; }
                    else header.push_back((char)ch);
                    continue;
                }
                if (ch == '\n') { line_start = true; continue; }
                if (line_start && ch == '>') { in_header = true; line_start = false; continue; }
                line_start = false;
                if (!in_seq) continue;
                int c = lut.v[ch];
                if (c == 5) continue;                                     // '\r' of a CRLF line end
                if (c >= 4) {
                    Ann& a = anns.back();
                    const uint64_t at = (uint64_t)(out - codes.data());
                    if (lasts == ch) ++holes.back().len;                  // contiguous run of the same ambiguity code
                    else { Hole h; h.offset = at; h.len = 1; h.amb = (char)ch; holes.push_back(h); ++a.n_ambs; }
                    c = (int)(rng.next() & 3);
                }
                lasts = ch;
                *out++ = (uint8_t)(c + 1);
            }
            codes.resize((size_t)(out - codes.data()));
        }
        if (in_header) take_header(header);
        // contig lengths from the offsets
        for (size_t k = 0; k < anns.size(); ++k) anns[k].len = (uint32_t)((k + 1 < anns.size() ? anns[k + 1].offset : codes.size()) - anns[k].offset);
    }
    gzclose(fp);
    const uint64_t l_pac = codes.size();
    if (l_pac == 0 || anns.empty()) { err = "no sequence in " + fa; return GM_E_IO; }
    // .pac
    {
        std::vector<uint8_t> pac((l_pac >> 2) + 1, 0);
        const uint64_t quads = l_pac >> 2;
        for (uint64_t q = 0; q < quads; ++q) {
            const uint8_t* c4 = &codes[q << 2];
            pac[q] = (uint8_t)(((c4[0] - 1) << 6) | ((c4[1] - 1) << 4) | ((c4[2] - 1) << 2) | (c4[3] - 1));
        }
        for (uint64_t l = quads << 2; l < l_pac; ++l) pac[l >> 2] |= (uint8_t)((codes[l] - 1) << ((~l & 3) << 1));
        FILE* f = fopen((fa + ".gnumap.pac").c_str(), "wb");
        if (!f) { err = "cannot write " + fa + ".gnumap.pac"; return GM_E_IO; }
        fwrite(pac.data(), 1, (l_pac >> 2) + ((l_pac & 3) == 0 ? 0 : 1), f);
        uint8_t ct = 0;
        if (l_pac % 4 == 0) fwrite(&ct, 1, 1, f);
        ct = (uint8_t)(l_pac % 4);
        fwrite(&ct, 1, 1, f);
        fclose(f);
    }
    // .ann / .amb
    {
        FILE* f = fopen((fa + ".gnumap.ann").c_str(), "w");
        if (!f) { err = "cannot write " + fa + ".gnumap.ann"; return GM_E_IO; }
        fprintf(f, "%lld %d %u\n", (long long)l_pac, (int)anns.size(), 11u);
        for (const Ann& a : anns) {
            fprintf(f, "%d %s", 0, a.name.c_str());
            if (!a.anno.empty()) fprintf(f, " %s\n", a.anno.c_str()); else fprintf(f, "\n");
            fprintf(f, "%lld %d %d\n", (long long)a.offset, (int)a.len, (int)a.n_ambs);
        }
        fclose(f);
        f = fopen((fa + ".gnumap.amb").c_str(), "w");
        if (!f) { err = "cannot write " + fa + ".gnumap.amb"; return GM_E_IO; }
        fprintf(f, "%lld %d %u\n", (long long)l_pac, (int)anns.size(), (unsigned)holes.size());
        for (const Hole& h : holes) fprintf(f, "%lld %d %c\n", (long long)h.offset, (int)h.len, h.amb);
        fclose(f);
    }
    codes.push_back(0);                                                     // the '$' sentinel
    lap("FASTA -> codes, .pac/.ann/.amb");
    // SA stage: on the MI355X when there is one (gm_sa_build.hip), else SA-IS on the host; GM_INDEX_BUILD=host|device overrides "auto"
    if (where == GM_BUILD_AUTO) {
        const char* e = getenv("GM_INDEX_BUILD");
        if (e && !strcmp(e, "host")) where = GM_BUILD_HOST;
        else if (e && !strcmp(e, "device")) where = GM_BUILD_DEVICE;
        else where = gm_device_available() && l_pac >= 4096 ? GM_BUILD_DEVICE : GM_BUILD_HOST;
    }
    const uint64_t intv = 32;
    std::vector<uint32_t> plain;
    std::vector<uint64_t> samples;
    uint64_t primary = 0;
    int rc;
    if (where == GM_BUILD_DEVICE) {
        if (!gm_device_available()) { err = "no usable HIP device for the device index build"; return GM_E_NO_DEVICE; }
        int rounds = 0;
        rc = gm_device_sa_build(codes.data(), l_pac, device_id, (uint32_t)intv, plain, primary, samples, &rounds, err);
        if (timing) fprintf(stderr, "[gm_timing] index build: %d doubling rounds after the 21-symbol sort\n", rounds);
    } else if (l_pac + 1 < (1ull << 31)) rc = host_sa_stage<int32_t>(codes, intv, plain, primary, samples, err);
    else rc = host_sa_stage<int64_t>(codes, intv, plain, primary, samples, err);
    if (rc) return rc;
    lap(where == GM_BUILD_DEVICE ? "suffix array + BWT + samples (device)" : "suffix array + BWT + samples (host SA-IS)");
    rc = write_bwt_sa(fa, codes, intv, plain, primary, samples, err);
    lap("occ interleave, .bwt/.sa");
    return rc;
}
