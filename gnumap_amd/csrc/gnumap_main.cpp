// gnumap_main.cpp — the C++ host driver: keeps the reference's command line (src/Driver.cpp:172-250, 2658-3251),
// SAM format (src/Driver.cpp:2146-2217, 2317-2329) and .sgr output (src/GenomeBwt.cpp:1212-1273), and drives the hot
// path through the C ABI of libgnumap_hip.so only.  Per batch it mirrors parallel_thread_run (src/Driver.cpp:2303-2407):
//     parse FASTQ block -> gm_map_batch (= loop over set_top_matches) -> gm_output_batch (= loop over create_match_output)
//     -> SAM text.
// I/O at rate: the FASTQ file is memory-mapped and cut into blocks by one scanner thread (memchr only, no
// copies); per GPU `--workers` host threads (default 3, each with its own gm_batch and HIP stream) pack a block, run the
// two batch calls and hand the records to formatter threads; SAM text is written in block order by one writer thread.
// The number of blocks in flight is bounded by a fixed pool of Block objects.
// Multi-GPU (--gpus N): one index replica per GPU, blocks dealt to whichever worker is free, coverage tracks combined
// with one RCCL all-reduce (gm_coverage_allreduce).
#include "gnumap_hip.h"
#include "gm_fmt.h"
#include <algorithm>
#include <atomic>
#include <charconv>
#include <condition_variable>
#include <deque>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define MAX_NAME_SZ 1024        // inc/const_include.h:46

struct Options {
    std::string genome, output = "gnumap_out", reads, subst;
    gm_params p;
    int gpus = 1, locate_sampled = 0, verbose = 1;
    uint32_t batch = 262144;
    bool batch_set = false;     // --batch given: blocks of exactly that many reads (one sequential scanner); otherwise byte-range chunks parsed by the workers
    int workers = 3;            // host threads (and gm_batch objects) per GPU (measured: 2 -> 9.6, 3 -> 11.4, 4 -> 11.0 M reads/s)
    int fmt_threads = 0;        // SAM formatter threads per block (0 = min(8, cores))
    int threads = 1;            // -c: accepted for compatibility (the GPU replaces the pthread pool)
};

static void usage(int rc, const char* msg) {
    if (msg && *msg) fprintf(stderr, "%s\n", msg);
    fprintf(stderr,
            "Usage: gnumap [options] <file_to_parse>\n"
            "  -g, --genome=STRING          Genome .fa file\n"
            "  -o, --output=STRING          Output file prefix (<out>.sam, <out>.sgr)\n"
            "  -a, --align_score=DOUBLE     Limit for sequence alignment (default: 0.9)\n"
            "  -r, --raw                    Use raw score when determining alignment cutoff\n"
            "  -q, --read_quality=DOUBLE    Read quality cutoff\n"
            "  -m, --mer_size=INT           Mer size (default: 10)\n"
            "  -j, --jump=INT               Number of bases to jump in the sequence indexing (default: mer_size/2)\n"
            "  -k, --num_seed=INT           Minimum number of seed matches per location (default: 2)\n"
            "  -h, --max_kmer=INT           Kmers occuring more often than this are skipped (default: no limit)\n"
            "  -T, --max_match=INT          Maximum number of matches per read (default: 1000)\n"
            "  -u X                         Only match sequences to one position (like the reference, -u swallows the next word)\n"
            "  -G, --gap_penalty=DOUBLE     Gap penalty (default: -4)\n"
            "  -S, --subst_file=STRING      5 x 4 substitution matrix file (rows a c g t n; scores are then used unscaled)\n"
            "  -c, --num_proc=INT           accepted for compatibility (the GPU path ignores it)\n"
            "  -b, --bs_seq / --b2 / -d, --a_to_g   bisulfite / A-to-G scoring\n"
            "      --snp                    pair-HMM per-nucleotide deposit (SNPScoredSeq); <out>.gmp without the likelihood-ratio columns\n"
            "      --no_nw                  use k-mer hit counts instead of Needleman-Wunsch alignments\n"
            "      --fast, --print_all_sam, --illumina, --up_strand, --down_strand, --bin_size=INT\n"
            "  MI355X options: --gpus=N  --batch=N (blocks of exactly N reads)  --chunk_reads=N  --workers=N  --fmt_threads=N  --locate=sampled|full\n");
    exit(rc);
}

static bool starts(const char* s, const char* pre) { return strncmp(s, pre, strlen(pre)) == 0; }

static void parse_args(int argc, char** argv, Options& o) {
    gm_params_default(&o.p);
    bool mer_set = false;
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        if (a[0] != '-') { if (!o.reads.empty()) usage(1, "Please specify only a single sequence file"); o.reads = a; continue; }
        if (a[1] == '-') {
            const char* s = a + 2;
            if (starts(s, "genome=")) o.genome = s + 7;
            else if (starts(s, "output=")) o.output = s + 7;
            else if (starts(s, "align_score=")) o.p.align_score = (float)atof(s + 12);
            else if (!strcmp(s, "percent")) o.p.align_is_fraction = 1;
            else if (!strcmp(s, "raw")) o.p.align_is_fraction = 0;
            else if (!strcmp(s, "no_nw")) o.p.nw = 0;
            else if (starts(s, "read_quality=")) o.p.cutoff = (float)atof(s + 13);
            else if (starts(s, "verbose=")) o.verbose = atoi(s + 8);
            else if (starts(s, "num_proc=")) o.threads = atoi(s + 9);
            else if (starts(s, "mer_size=")) { o.p.mer = atoi(s + 9); mer_set = true; }
            else if (starts(s, "max_match=")) o.p.max_matches = (uint32_t)atoi(s + 10);
            else if (starts(s, "max_kmer=")) o.p.max_kmer_hits = (uint32_t)atoi(s + 9);
            else if (starts(s, "gap_penalty=")) o.p.gap = (float)atof(s + 12);
            else if (starts(s, "subst_file=")) o.subst = s + 11;
            else if (starts(s, "max_gap=")) o.p.max_gap = atoi(s + 8);
            else if (!strcmp(s, "unique")) o.p.unique_only = 1;
            else if (!strcmp(s, "print_all_sam")) o.p.print_all_sam = 1;
            else if (!strcmp(s, "bs_seq")) o.p.mode = GM_MODE_BS;
            else if (!strcmp(s, "b2")) o.p.mode = GM_MODE_BS2;
            else if (!strcmp(s, "a_to_g")) o.p.mode = GM_MODE_ATOG;
            else if (!strcmp(s, "snp")) o.p.mode = GM_MODE_SNP;             // Driver.cpp:3207-3211: gSNP, bin size 1
            else if (!strcmp(s, "fast")) o.p.fast = 1;
            else if (starts(s, "bin_size=")) o.p.bin_size = atoi(s + 9);
            else if (starts(s, "jump=")) o.p.jump = atoi(s + 5);
            else if (starts(s, "num_seed=")) o.p.min_seed_hits = atoi(s + 9);
            else if (!strcmp(s, "illumina")) o.p.illumina = 1;
            else if (!strcmp(s, "up_strand")) { o.p.pos_strand = 1; o.p.neg_strand = 0; }
            else if (!strcmp(s, "down_strand")) { o.p.pos_strand = 0; o.p.neg_strand = 1; }
            else if (starts(s, "gpus=")) o.gpus = atoi(s + 5);
            else if (starts(s, "batch=")) { o.batch = (uint32_t)atoi(s + 6); o.batch_set = true; }
            else if (starts(s, "chunk_reads=")) o.batch = (uint32_t)atoi(s + 12);       // byte-range chunks of about this many records (default 262144)
            else if (starts(s, "workers=")) o.workers = atoi(s + 8);
            else if (starts(s, "fmt_threads=")) o.fmt_threads = atoi(s + 12);
            else if (starts(s, "locate=")) o.locate_sampled = !strcmp(s + 7, "sampled");
            else if (!strcmp(s, "help")) usage(0, "");
            else { fprintf(stderr, "No matching arg in: %s\n", a); exit(1); }
            continue;
        }
        if (strlen(a) > 2) { fprintf(stderr, "Irregular Parameter in: %s\n", a); exit(1); }
        char c = a[1];
        // -u takes no value but the reference's parser consumes the next argv all the same (src/Driver.cpp:2768-2770 lacks the
        // count--): a command line written for the reference has a filler word there, so the same is done here
        bool flag = strchr("prbd0", c) != nullptr;
        const char* v = nullptr;
        if (!flag) { if (i + 1 >= argc) usage(1, "missing value"); v = argv[++i]; }
        switch (c) {
            case 'g': o.genome = v; break;
            case 'o': o.output = v; break;
            case 'a': o.p.align_score = (float)atof(v); break;
            case 'p': o.p.align_is_fraction = 1; break;
            case 'r': o.p.align_is_fraction = 0; break;
            case 'q': o.p.cutoff = (float)atof(v); break;
            case 'v': o.verbose = atoi(v); break;
            case 'c': o.threads = atoi(v); break;
            case 'm': o.p.mer = atoi(v); mer_set = true; break;
            case 'u': o.p.unique_only = 1; break;
            case 'T': o.p.max_matches = (uint32_t)atoi(v); break;
            case 'h': o.p.max_kmer_hits = (uint32_t)atoi(v); break;
            case 'G': o.p.gap = (float)atof(v); break;
            case 'M': o.p.max_gap = atoi(v); break;
            case 'b': o.p.mode = GM_MODE_BS; break;
            case 'd': o.p.mode = GM_MODE_ATOG; break;
            case '0': break;
            case 'j': o.p.jump = atoi(v); break;
            case 'k': o.p.min_seed_hits = atoi(v); break;
            case 'S': o.subst = v; break;
            case 'l': case 'B': case 's': case 'A': fprintf(stderr, "option -%c is outside the hot path of this build\n", c); exit(1);
            case '?': usage(0, "");
            default: fprintf(stderr, "Irregular Parameter in: %s\n", a); exit(1);
        }
    }
    if (o.reads.empty()) usage(1, "Specify a single file e.g., sequences.fa\n");
    if (o.genome.empty()) usage(1, "Specify a genome to map to with the -g flag\n");
    if (o.p.fast) { if (!mer_set) o.p.mer = 14; o.p.jump = o.p.mer; }      // Driver.cpp:1163-1171
    if ((o.p.mode == GM_MODE_BS || o.p.mode == GM_MODE_BS2) && !o.p.pos_strand) o.p.mode = GM_MODE_BS2;    // Driver.cpp:1260-1281
    if (o.p.mode == GM_MODE_ATOG && !o.p.pos_strand) o.p.mode = GM_MODE_ATOG2;
    if (gm_params_finalize(&o.p) != GM_OK) { fprintf(stderr, "%s\n", gm_last_error()); exit(1); }
    if (!o.subst.empty() && gm_params_load_subst(&o.p, o.subst.c_str()) != GM_OK) { fprintf(stderr, "ERROR: \n\t%s\n", gm_last_error()); exit(1); }
    if (o.gpus < 1) o.gpus = 1;
    if (o.batch < 1) o.batch = 1;
    if (o.workers < 1) o.workers = 1;
    if (o.fmt_threads < 1) o.fmt_threads = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
}

// page-locked host array (gm_host_alloc): the library's copies to and from it are DMA at link rate, no staging through the CPU
template <class T> struct PinVec {
    T* p = nullptr; size_t cap = 0;
    PinVec() = default;
    PinVec(const PinVec&) = delete; PinVec& operator=(const PinVec&) = delete;
    ~PinVec() { if (p) gm_host_free(p); }
    void ensure(size_t n) {                       // contents are NOT kept
        if (n <= cap) return;
        if (p) gm_host_free(p);
        cap = n + n / 8 + 64;
        p = (T*)gm_host_alloc(cap * sizeof(T));
        if (!p) { fprintf(stderr, "ERROR: out of page-locked host memory (%zu bytes)\n", cap * sizeof(T)); exit(1); }
    }
    T* data() { return p; } const T* data() const { return p; } size_t size() const { return cap; }
    T& operator[](size_t i) { return p[i]; } const T& operator[](size_t i) const { return p[i]; }
};

// SAM text of one formatter slice: plain bytes that are never value-initialised (std::string::resize would write every byte of the
// 8.7 GB of a 32 M-read run once before the formatter writes it again)
struct TextBuf {
    std::unique_ptr<char[]> p; size_t len = 0, cap = 0;
    void clear() { len = 0; }
    bool empty() const { return len == 0; }
    size_t size() const { return len; }
    const char* data() const { return p.get(); }
    char* room(size_t extra) {                        // at least `extra` writable bytes at the end
        if (len + extra > cap) {
            const size_t nc = std::max(len + extra, cap + cap / 2 + 4096);
            std::unique_ptr<char[]> q(new char[nc]);
            if (len) memcpy(q.get(), p.get(), len);
            p = std::move(q); cap = nc;
        }
        return p.get() + len;
    }
};

// ---- blocks ------------------------------------------------------------------------------------------------------
struct Block {
    uint64_t index = 0;
    uint32_t n = 0, maxlen = 0, stride = 8;
    // views into the memory-mapped FASTQ text (no per-read strings)
    std::vector<const char*> name, seq, qual;
    std::vector<uint32_t> name_len, qual_len;
    std::vector<uint16_t> len;
    // packed for gm_reads (page-locked)
    PinVec<uint8_t> bases, qbuf; PinVec<uint16_t> plen;
    // results of the two batch calls (page-locked)
    PinVec<gm_sam_rec> recs; PinVec<char> pool; uint64_t n_recs = 0;
    int gpu = 0;
    size_t text_lo = 0, text_hi = 0; bool lazy = false;   // chunk mode: the byte range of the FASTQ text a worker still has to cut into records
    bool malformed = false;                 // chunk mode: a malformed record ended this block; the rest of the input is read again in file order with the reference's recovery
    int illumina = 0;                       // --illumina still in force when this block starts (the fallback is sticky, SeqReader.cpp:1171-1180)
    std::vector<TextBuf> text;              // SAM text, one piece per formatter thread
    bool failed = false;
};

template <class T> struct Queue {             // unbounded MPMC queue; the Block pool bounds what is in flight
    std::mutex mu; std::condition_variable cv; std::deque<T> q; bool closed = false;
    void push(T v) { { std::lock_guard<std::mutex> lk(mu); q.push_back(v); } cv.notify_one(); }
    bool pop(T& v) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        v = q.front(); q.pop_front();
        return true;
    }
    void close() { { std::lock_guard<std::mutex> lk(mu); closed = true; } cv.notify_all(); }
};

// ---- FASTQ scanner (SeqReader::get_more_fastq src/SeqReader.cpp:1023-1292: 4-line records, blank lines before a name skipped)
struct FastqScanner {
    const char* base = nullptr; size_t size = 0, at = 0;
    int fd = -1; bool mapped = false; std::vector<char> heap;
    bool done = false;
    bool open(const std::string& fn) {
        fd = ::open(fn.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            size = (size_t)st.st_size;
            if (size == 0) { base = ""; return true; }
            void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (p != MAP_FAILED) { base = (const char*)p; mapped = true; madvise(p, size, MADV_SEQUENTIAL); return true; }
        }
        char buf[1 << 16]; ssize_t k;                       // pipes and the like: slurp
        while ((k = ::read(fd, buf, sizeof buf)) > 0) heap.insert(heap.end(), buf, buf + k);
        base = heap.data(); size = heap.size();
        return true;
    }
    ~FastqScanner() { if (mapped) munmap((void*)base, size); if (fd >= 0) ::close(fd); }
    // one line [p, p+len), without the newline; false at end of input
    inline bool line(const char*& p, size_t& len) {
        if (at >= size) return false;
        const char* s = base + at;
        const char* e = (const char*)memchr(s, '\n', size - at);
        if (!e) { p = s; len = size - at; at = size; return true; }
        p = s; len = (size_t)(e - s); at = (size_t)(e - base) + 1;
        return true;
    }
    // Chunk mode: cut base[cur, limit) into records until max_reads (thread-safe: no scanner state).  Only well-formed records are
    // taken here; at the first one that is not, `stop` is set and `bad_at` is where its name line starts: the driver then reads the rest
    // of the input again from there in file order (parse_resync), because what the reference does with a malformed record depends on
    // the lines that follow it.  `too_long` = a read longer than the kernels take (this implementation's limit, not the reference's).
    static bool parse_range(const char* base, size_t& cur, size_t limit, Block& b, uint32_t max_reads, bool* stop, size_t* bad_at, bool* too_long) {
        b.n = 0; b.maxlen = 0;
        b.name.clear(); b.seq.clear(); b.qual.clear(); b.name_len.clear(); b.qual_len.clear(); b.len.clear();
        auto line = [&](const char*& p, size_t& len) -> bool {       // one line without its newline; false at the end of the range
            if (cur >= limit) return false;
            const char* s0 = base + cur;
            const char* e = (const char*)memchr(s0, '\n', limit - cur);
            if (!e) { p = s0; len = limit - cur; cur = limit; return true; }
            p = s0; len = (size_t)(e - s0); cur = (size_t)(e - base) + 1;
            return true;
        };
        while (b.n < max_reads) {
            const char *nm, *sq = nullptr, *pl = nullptr, *ql = nullptr; size_t nl, sl = 0, pll = 0, qll = 0;
            if (!line(nm, nl)) break;
            bool eof = false;
            while (nl == 0) { if (!line(nm, nl)) { eof = true; break; } }
            if (eof) break;
            const size_t rec_at = (size_t)(nm - base);
            const bool l2 = line(sq, sl), l3 = line(pl, pll), l4 = line(ql, qll);
            if (!l2 || !l3 || !l4 || nm[0] != '@' || pll == 0 || pl[0] != '+' || sl > qll) { *stop = true; *bad_at = rec_at; break; }
            if (sl > 2048) { fprintf(stderr, "ERROR: read %.*s is longer than 2048 bases\n", (int)std::min<size_t>(nl, 200), nm); *too_long = true; break; }
            b.name.push_back(nm + 1); b.name_len.push_back((uint32_t)(nl - 1));
            b.seq.push_back(sq); b.len.push_back((uint16_t)sl);
            b.qual.push_back(ql); b.qual_len.push_back((uint32_t)qll);
            b.maxlen = std::max<uint32_t>(b.maxlen, (uint32_t)sl);
            ++b.n;
        }
        b.stride = std::max<uint32_t>(8, (b.maxlen + 7u) & ~7u);
        return b.n > 0;
    }
    // File order, with the recovery of SeqReader::get_more_fastq (src/SeqReader.cpp:1060-1150): four lines are read; while the first does
    // not start with '@' or the third not with '+', the lines are shifted up by one and one more is read; a quality line shorter than
    // its sequence makes the reader take the next four lines instead.  The lines come from std::getline on an ifstream, whose end-of-input
    // behaviour decides what happens to a record the input ends in: a stream that is no longer good fails WITHOUT touching the string,
    // the end of input empties the string and sets eofbit + failbit, a last line without a newline is delivered and sets eofbit; every
    // end-of-input test inside the recovery ends the input (a record completed by a line without a newline is lost there).
    struct Line { const char* p = ""; size_t n = 0; char first() const { return n ? p[0] : 0; } };
    bool rs_eof = false, rs_fail = false;
    void rs_getline(Line& s) {
        if (rs_eof || rs_fail) { rs_fail = true; return; }
        if (at >= size) { s = Line(); rs_eof = rs_fail = true; return; }
        const char* s0 = base + at;
        const char* e = (const char*)memchr(s0, '\n', size - at);
        s.p = s0;
        if (e) { s.n = (size_t)(e - s0); at = (size_t)(e - base) + 1; } else { s.n = size - at; at = size; rs_eof = true; }
    }
    bool parse_resync(Block& b, uint32_t max_reads, bool* too_long) {
        b.n = 0; b.maxlen = 0;
        b.name.clear(); b.seq.clear(); b.qual.clear(); b.name_len.clear(); b.qual_len.clear(); b.len.clear();
        Line name, seq, plus, qual;
        while (b.n < max_reads) {
            if (rs_eof) { done = true; break; }                                               // :1061
            rs_getline(name);
            while (name.n == 0 && !rs_eof) rs_getline(name);                                  // blank lines before a name :1076-1079
            if (rs_eof) { done = true; break; }                                               // :1082
            rs_getline(seq); rs_getline(plus); rs_getline(qual);
            while (name.first() != '@' || plus.first() != '+' || seq.n > qual.n) {            // :1095
                fprintf(stderr, "--Found error...atteming to resolve...\n");
                if (name.first() != '@' || plus.first() != '+') {
                    fprintf(stderr, "--ERROR at sequence %.*s (name[s] formatted incorrectly)\n\tTrying to recover...\n", (int)std::min<size_t>(name.n, 200), name.p);
                    while ((name.first() != '@' || plus.first() != '+') && !rs_eof) {         // :1101-1115
                        name = seq; seq = plus; plus = qual; rs_getline(qual);
                        if (rs_eof) { done = true; break; }
                    }
                    if (done || rs_eof) { done = true; break; }
                }
                if (seq.n > qual.n) {                                                         // :1128-1142
                    fprintf(stderr, "--ERROR at sequence %.*s (length of fastq and sequence not equal)\n\tTrying to recover...\n", (int)std::min<size_t>(name.n, 200), name.p);
                    rs_getline(name); rs_getline(seq); rs_getline(plus); rs_getline(qual);
                    if (rs_eof) { done = true; break; }
                }
            }
            if (done) break;
            if (seq.n > 2048) { fprintf(stderr, "ERROR: read %.*s is longer than 2048 bases\n", (int)std::min<size_t>(name.n, 200), name.p); *too_long = true; done = true; break; }
            b.name.push_back(name.p + 1); b.name_len.push_back((uint32_t)(name.n - 1));
            b.seq.push_back(seq.p); b.len.push_back((uint16_t)seq.n);
            b.qual.push_back(qual.p); b.qual_len.push_back((uint32_t)qual.n);
            b.maxlen = std::max<uint32_t>(b.maxlen, (uint32_t)seq.n);
            ++b.n;
        }
        b.stride = std::max<uint32_t>(8, (b.maxlen + 7u) & ~7u);
        return b.n > 0;
    }
    bool next(Block& b, uint32_t max_reads, bool* too_long) {    // file-order mode: blocks of exactly max_reads reads
        if (done) return false;
        return parse_resync(b, max_reads, too_long);
    }
    // chunk mode: the start of the first record at or after `off` (a line starting with '@' whose next-but-one line starts with
    // '+': a quality line may start with '@' too, but then the line two below is a sequence, never '+')
    size_t record_start(size_t off) const {
        if (off == 0) return 0;
        if (off >= size) return size;
        const char* e = (const char*)memchr(base + off - 1, '\n', size - off + 1);      // the line containing off-1 ends here
        size_t p = e ? (size_t)(e - base) + 1 : size;
        while (p < size) {
            const char* l1 = (const char*)memchr(base + p, '\n', size - p);
            if (!l1) return size;
            const char* l2 = (const char*)memchr(l1 + 1, '\n', size - (size_t)(l1 + 1 - base));
            if (!l2) return size;
            if (base[p] == '@' && (size_t)(l2 + 1 - base) < size && l2[1] == '+') return p;
            p = (size_t)(l1 - base) + 1;
        }
        return size;
    }
    size_t chunk_bytes = 0;
    bool next_chunk(Block& b, uint32_t target_reads) {       // a byte range of about target_reads records, cut at record starts
        if (done || at >= size) { done = true; return false; }
        if (!chunk_bytes) {                                  // record size from the first records of the file
            size_t p = 0; uint32_t recs = 0;
            while (recs < 4000 && p < size) { const char* e = (const char*)memchr(base + p, '\n', size - p); if (!e) break; p = (size_t)(e - base) + 1; ++recs; }
            const double per_rec = recs >= 4 ? (double)p / (recs / 4) : 256.0;
            chunk_bytes = (size_t)std::max(4096.0, per_rec * target_reads);
        }
        const size_t lo = at, hi = record_start(std::min(size, at + chunk_bytes));
        b.text_lo = lo; b.text_hi = hi > lo ? hi : size; b.lazy = true; b.n = 0;
        at = b.text_hi;
        if (at >= size) done = true;
        return true;
    }
};

template <class F> static void run_slices(uint32_t n, int threads, uint32_t grain, F&& fn) {          // fn(slice, lo, hi)
    int T = (int)std::min<uint64_t>((uint64_t)threads, std::max<uint32_t>(1, n / std::max<uint32_t>(1, grain)));
    if (T <= 1) { fn(0, 0u, n); return; }
    const uint64_t per = (n + (uint64_t)T - 1) / (uint64_t)T;
    std::vector<std::thread> th;
    for (int s = 1; s < T; ++s)
        th.emplace_back([&, s] { fn(s, (uint32_t)std::min<uint64_t>(n, s * per), (uint32_t)std::min<uint64_t>(n, (s + 1) * per)); });
    fn(0, 0u, (uint32_t)std::min<uint64_t>(n, per));
    for (auto& x : th) x.join();
}

static void pack_block(Block& b, int threads) {          // rows of `stride` bytes, zero padded, as gm_reads wants them
    const size_t bytes = (size_t)b.n * b.stride;
    b.bases.ensure(bytes); b.qbuf.ensure(bytes); b.plen.ensure(b.n);
    memcpy(b.plen.data(), b.len.data(), (size_t)b.n * 2);
    run_slices(b.n, threads, 8192, [&](int, uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; ++i) {
            uint8_t* pb = &b.bases[(size_t)i * b.stride]; uint8_t* pq = &b.qbuf[(size_t)i * b.stride];
            const uint32_t L = b.len[i];
            memcpy(pb, b.seq[i], L); memset(pb + L, 0, b.stride - L);
            memcpy(pq, b.qual[i], L); memset(pq + L, 0, b.stride - L);
        }
    });
}

// ---- SAM text (src/Driver.cpp:2146-2217) ---------------------------------------------------------------------------
static inline char comp_char(char c) {                   // reverse_comp, SequenceOperations.h:56-96
    switch (c) {
        case 'a': return 't'; case 'A': return 'T'; case 't': return 'a'; case 'T': return 'A';
        case 'c': return 'g'; case 'C': return 'G'; case 'g': return 'c'; case 'G': return 'C';
        case '-': return '-'; default: return 'n';
    }
}

static size_t reverse_cigar(const char* s, char* out) {  // SequenceOperations.h:109-123 (':' counts as a digit there, 48..58)
    const size_t n = strlen(s);
    size_t w = n, tok = 0;
    for (size_t i = 0; i < n; ++i) {
        if (s[i] >= 48 && s[i] <= 58) continue;
        const size_t len = i + 1 - tok;                    // digits + the operation character
        w -= len;
        memcpy(out + w, s + tok, len);
        tok = i + 1;
    }
    // trailing digits without an operation are dropped by the reference
    const size_t used = n - w;
    memmove(out, out + w, used);
    return used;
}

static inline char* put_u64(char* p, uint64_t v) { return gm_put_u64(p, v); }

static const struct CompLut { char t[256]; CompLut() { for (int c = 0; c < 256; ++c) t[c] = comp_char((char)c); } } g_comp;

// dst[0..n) = src[n-1..0], 8 bytes at a time
static inline void reverse_bytes(char* dst, const char* src, uint32_t n) {
    uint32_t k = 0;
    for (; k + 8 <= n; k += 8) { uint64_t x; memcpy(&x, src + n - 8 - k, 8); x = __builtin_bswap64(x); memcpy(dst + k, &x, 8); }
    for (; k < n; ++k) dst[k] = src[n - 1 - k];
}

static void format_sam(TextBuf& out, const gm_index* ix, const gm_params& p, const gm_sam_rec& r, const char* cigar, const Block& b) {
    const uint32_t i = r.read;
    const uint32_t L = b.len[i], QL = b.qual_len[i];
    const uint32_t nl = std::min<uint32_t>(b.name_len[i], MAX_NAME_SZ - 1);
    const char* cn = gm_index_contig_name(ix, r.contig);
    const size_t cl = strlen(cn), gl = strlen(cigar);
    char* const w0 = out.room(nl + cl + gl + L + QL + 160);
    char* w = w0;
    memcpy(w, b.name[i], nl); w += nl;
    if (r.strand == GM_POS_STRAND) { memcpy(w, "\t0\t", 3); w += 3; } else { memcpy(w, "\t16\t", 4); w += 4; }
    memcpy(w, cn, cl); w += cl;
    *w++ = '\t'; w = put_u64(w, r.chr_pos); *w++ = '\t';
    if (r.mapq < 0) { *w++ = '-'; w = put_u64(w, (uint64_t)(-(int64_t)r.mapq)); } else w = put_u64(w, (uint64_t)r.mapq);
    *w++ = '\t';
    if (r.strand == GM_POS_STRAND) {
        memcpy(w, cigar, gl); w += gl;
        memcpy(w, "\t*\t0\t0\t", 7); w += 7;
        memcpy(w, b.seq[i], L); w += L; *w++ = '\t';
        memcpy(w, b.qual[i], QL); w += QL; *w++ = '\t';
    } else {
        w += reverse_cigar(cigar, w);
        memcpy(w, "\t*\t0\t0\t", 7); w += 7;
        reverse_bytes(w, b.seq[i], L);
        for (uint32_t k = 0; k < L; ++k) w[k] = g_comp.t[(unsigned char)w[k]];
        w += L; *w++ = '\t';
        reverse_bytes(w, b.qual[i], QL);
        w += QL; *w++ = '\t';
    }
    // "XA:f:%g\tXP:f:%g\tX0:i:%d\n" (gm_fmt.h: exactly printf's %g)
    memcpy(w, "XA:f:", 5); w += 5;
    w = gm_put_g6(w, (double)(float)r.a_score * (1.0 / p.adjust));
    memcpy(w, "\tXP:f:", 6); w += 6;
    if (r.post_prob == 1.0f) *w++ = '1'; else w = gm_put_g6(w, (double)(float)r.post_prob);
    memcpy(w, "\tX0:i:", 6); w += 6;
    if (r.sim_matches < 0) { *w++ = '-'; w = put_u64(w, (uint64_t)(-(int64_t)r.sim_matches)); } else w = put_u64(w, (uint64_t)r.sim_matches);
    *w++ = '\n';
    out.len += (size_t)(w - w0);
}

// ---- per worker: the two batch calls ---------------------------------------------------------------------------------
struct Worker {
    int gpu = 0;
    gm_index* ix = nullptr;
    gm_batch* batch = nullptr;
    void* stream = nullptr;                 // this worker's own HIP stream: its device work overlaps the other workers'
    PinVec<int8_t> status; PinVec<float> self_score; PinVec<double> top, den; PinVec<uint64_t> mbegin;
    PinVec<gm_match> matches; PinVec<gm_pos> positions;
    uint64_t n_reads = 0, n_matched = 0, n_records = 0;
    double t_pack = 0, t_map = 0, t_out = 0, t_scan = 0;
};

static double secs_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }

static int process_block(Worker& w, const Options& o, Block& b) {
    const uint32_t n = b.n;
    gm_params bp = o.p; bp.illumina = b.illumina;
    auto c0 = std::chrono::steady_clock::now();
    pack_block(b, 4);
    gm_reads reads; reads.n = n; reads.stride = b.stride; reads.bases = b.bases.data(); reads.quals = b.qbuf.data(); reads.len = b.plen.data();
    w.status.ensure(n); w.self_score.ensure(n); w.top.ensure(n); w.den.ensure(n); w.mbegin.ensure((size_t)n + 1);
    w.matches.ensure(2 * (size_t)n + 64); w.positions.ensure(2 * (size_t)n + 64);
    gm_hits hits{};
    auto c1 = std::chrono::steady_clock::now();
    for (;;) {
        hits.n = n; hits.status = w.status.data(); hits.self_score = w.self_score.data(); hits.top_score = w.top.data();
        hits.denominator = w.den.data(); hits.match_begin = w.mbegin.data();
        hits.matches = w.matches.data(); hits.matches_cap = w.matches.size();
        hits.positions = w.positions.data(); hits.positions_cap = w.positions.size();
        int rc = gm_map_batch(w.ix, &bp, w.batch, &reads, &hits, w.stream);
        if (rc == GM_E_CAPACITY) { w.matches.ensure(hits.matches_cap + 64); w.positions.ensure(hits.positions_cap + 64); continue; }
        if (rc == GM_E_BATCH_TOO_LARGE) { fprintf(stderr, "note: block of %u reads is mapped in halves (%s)\n", n, gm_last_error()); return rc; }
        if (rc != GM_OK) { fprintf(stderr, "ERROR: gm_map_batch: %s\n", gm_last_error()); return rc; }
        break;
    }
    auto c2 = std::chrono::steady_clock::now();
    b.recs.ensure((size_t)n + (size_t)n / 4 + 64);
    b.pool.ensure(8 * (size_t)n + 1024);
    gm_sam_out so;
    for (;;) {
        so.recs = b.recs.data(); so.recs_cap = b.recs.size(); so.cigar_pool = b.pool.data(); so.cigar_cap = b.pool.size();
        int rc = gm_output_batch(w.ix, &bp, w.batch, &reads, &hits, &so, w.stream);
        if (rc == GM_E_CAPACITY) { b.recs.ensure(so.recs_cap + 64); b.pool.ensure(so.cigar_cap + 64); continue; }
        if (rc == GM_E_BATCH_TOO_LARGE) { fprintf(stderr, "note: block of %u reads is written in halves (%s)\n", n, gm_last_error()); return rc; }
        if (rc != GM_OK) { fprintf(stderr, "ERROR: gm_output_batch: %s\n", gm_last_error()); return rc; }
        break;
    }
    b.n_recs = so.n_recs; b.gpu = w.gpu;
    w.t_pack += std::chrono::duration<double>(c1 - c0).count(); w.t_map += std::chrono::duration<double>(c2 - c1).count(); w.t_out += secs_since(c2);
    w.n_reads += n; w.n_records += so.n_recs;
    for (uint32_t i = 0; i < n; ++i) w.n_matched += (w.status[i] == GM_READ_OK || w.status[i] == GM_READ_TOO_MANY);
    return GM_OK;
}

// a block whose intermediate lists outgrow one launch (GM_E_BATCH_TOO_LARGE: e.g. > 2^31 candidates on a repeat-rich reference
// without -h) is mapped as two halves, recursively; the halves' records are put back together in read order
static int process_block_split(Worker& w, const Options& o, Block& b, int depth) {
    int rc = process_block(w, o, b);
    if (rc != GM_E_BATCH_TOO_LARGE || b.n < 2 || depth > 20) return rc;
    const uint32_t half = b.n / 2;
    uint64_t total = 0; size_t pool_len = 0;
    std::vector<gm_sam_rec> recs; std::vector<char> pool;
    for (int part = 0; part < 2; ++part) {
        const uint32_t lo = part ? half : 0, hi = part ? b.n : half;
        Block c;
        c.index = b.index; c.n = hi - lo; c.maxlen = b.maxlen; c.stride = b.stride; c.illumina = b.illumina;
        c.name.assign(b.name.begin() + lo, b.name.begin() + hi); c.seq.assign(b.seq.begin() + lo, b.seq.begin() + hi);
        c.qual.assign(b.qual.begin() + lo, b.qual.begin() + hi); c.name_len.assign(b.name_len.begin() + lo, b.name_len.begin() + hi);
        c.qual_len.assign(b.qual_len.begin() + lo, b.qual_len.begin() + hi); c.len.assign(b.len.begin() + lo, b.len.begin() + hi);
        if (part && b.illumina) {                           // the fallback may have happened inside the first half
            for (uint32_t i = 0; i < half && c.illumina; ++i)
                for (uint32_t t = 0; t < b.len[i]; ++t) if ((signed char)b.qual[i][t] < 64) { c.illumina = 0; break; }      // (the reference's quality characters are signed chars)
        }
        rc = process_block_split(w, o, c, depth + 1);
        if (rc != GM_OK) return rc;
        for (uint64_t k = 0; k < c.n_recs; ++k) { gm_sam_rec r = c.recs[k]; r.read += lo; r.cigar_off += (uint32_t)pool_len; recs.push_back(r); }
        size_t used = 0;
        for (uint64_t k = 0; k < c.n_recs; ++k) used = std::max<size_t>(used, c.recs[k].cigar_off + strlen(c.pool.data() + c.recs[k].cigar_off) + 1);
        pool.insert(pool.end(), c.pool.data(), c.pool.data() + used);
        pool_len += used; total += c.n_recs;
    }
    b.recs.ensure(recs.size() + 1); b.pool.ensure(pool.size() + 1);
    if (!recs.empty()) memcpy(b.recs.data(), recs.data(), recs.size() * sizeof(gm_sam_rec));
    if (!pool.empty()) memcpy(b.pool.data(), pool.data(), pool.size());
    b.n_recs = total; b.gpu = w.gpu;
    return GM_OK;
}

int main(int argc, char** argv) {
    Options o;
    parse_args(argc, argv, o);
    std::string cl;
    for (int i = 0; i < argc; ++i) { cl += argv[i]; cl += " "; }      // Driver.cpp:1032-1039
    auto t0 = std::chrono::steady_clock::now();
    const int flags = GM_INDEX_BUILD | (o.locate_sampled ? 0 : GM_INDEX_FULL_SA);
    std::vector<gm_index*> gpu_ix((size_t)o.gpus, nullptr);
    {   // build once if missing, then one replica per GPU
        if (gm_index_open(o.genome.c_str(), 0, flags, &gpu_ix[0]) != GM_OK) { fprintf(stderr, "ERROR: %s\n", gm_last_error()); return 1; }
        for (int g = 1; g < o.gpus; ++g)
            if (gm_index_open(o.genome.c_str(), g, flags, &gpu_ix[(size_t)g]) != GM_OK) { fprintf(stderr, "ERROR: GPU %d: %s\n", g, gm_last_error()); return 1; }
    }
    for (int g = 0; g < o.gpus; ++g)                                   // k-mer tables / records for these parameters: part of staging the index
        if (gm_index_prepare(gpu_ix[(size_t)g], &o.p) != GM_OK) { fprintf(stderr, "ERROR: GPU %d: %s\n", g, gm_last_error()); return 1; }
    std::vector<Worker> workers((size_t)o.gpus * (size_t)o.workers);
    for (int g = 0; g < o.gpus; ++g) {
        if (gm_coverage_reset(gpu_ix[(size_t)g], (uint32_t)o.p.bin_size) != GM_OK ||
            (o.p.mode != GM_MODE_NORMAL && gm_coverage_enable_nuc(gpu_ix[(size_t)g]) != GM_OK)) { fprintf(stderr, "ERROR: %s\n", gm_last_error()); return 1; }
        for (int k = 0; k < o.workers; ++k) {
            Worker& w = workers[(size_t)g * (size_t)o.workers + (size_t)k];
            w.gpu = g; w.ix = gpu_ix[(size_t)g];
            if (gm_batch_create(w.ix, o.batch_set || o.p.illumina ? o.batch : 16000000u, 2048, &w.batch) != GM_OK || gm_stream_create(w.ix, &w.stream) != GM_OK) { fprintf(stderr, "ERROR: %s\n", gm_last_error()); return 1; }
        }
    }
    gm_index_info info;
    gm_index_get_info(gpu_ix[0], &info);
    const double t_index = secs_since(t0);
    if (o.verbose > 0)
        fprintf(stderr, "gnumap-mi355x: genome %s (%lu bp, %u contigs), %s locate, %d GPU(s), index %.1f MB in HBM\n", o.genome.c_str(),
                (unsigned long)info.l_pac, info.n_seqs, info.full_sa ? "full-SA" : "sampled-SA", o.gpus, info.hbm_bytes / 1e6);
    const int ofd = ::open((o.output + ".sam").c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (ofd < 0) { fprintf(stderr, "ERROR: cannot write %s.sam\n", o.output.c_str()); return 1; }
    auto write_all = [&](const char* p, size_t n) {
        while (n) { ssize_t k = ::write(ofd, p, n); if (k <= 0) return false; p += k; n -= (size_t)k; }
        return true;
    };
    {
        std::ostringstream hd;
        for (uint32_t i = 0; i < info.n_seqs; ++i)                     // Driver.cpp:2322-2327
            hd << "@SQ\tSN:" << gm_index_contig_name(gpu_ix[0], i) << "\tLN:"
               << (gm_index_contig_offset(gpu_ix[0], i + 1) - gm_index_contig_offset(gpu_ix[0], i)) << "\n";
        hd << "@PG\tID:gnumap\tPN:gnumap\tVN:4.0.0 BETA\tCL:" << cl << "\n";
        const std::string s = hd.str();
        if (!write_all(s.data(), s.size())) { fprintf(stderr, "ERROR: write failed\n"); return 1; }
    }

    auto t_pipe0 = std::chrono::steady_clock::now();
    FastqScanner fq;
    if (!fq.open(o.reads)) { fprintf(stderr, "ERROR: cannot open %s\n", o.reads.c_str()); return 1; }
    std::atomic<int> failed{ 0 };
    double t_scan = 0, t_fmt = 0, t_write = 0;
    uint64_t file_off = 0;
    { struct stat hs; if (fstat(ofd, &hs) == 0) file_off = (uint64_t)lseek(ofd, 0, SEEK_CUR); }
    auto pwrite_all = [&](const char* p, size_t n, uint64_t off) {
        while (n) { ssize_t k = ::pwrite(ofd, p, n, (off_t)off); if (k <= 0) return false; p += k; n -= (size_t)k; off += (uint64_t)k; }
        return true;
    };
    // (Tried in round 4: the pieces copied in through shared mappings of the file's byte ranges instead - no inode write lock, which is
    // what holds buffered pwrite()s to ONE file to the rate of one thread, 8.4-8.8 GB/s whatever the number of writers.  4 x SLOWER on the
    // GPU box's file system: 2.1 M first-touch page faults of a growing file cost more than the copies.  The write calls stay.)
    const size_t npos = ~(size_t)0;
    // One pass of the pipeline over the FASTQ text from fq.at on.  chunks = byte ranges cut into records by the workers in parallel
    // (well-formed records only); returns where a malformed record starts (npos: none) - everything before it has been mapped and
    // written, nothing after it has touched the SAM file or the coverage track: a block is mapped only once every block before it is
    // known to be well-formed.  !chunks = file order with the reference's recovery from malformed records (--batch=N, --illumina - its
    // fallback is a property of the reads in file order -, and the rest of an input in which a malformed record was found).
    auto run_pass = [&](const bool chunks) -> size_t {
        const size_t n_blocks = workers.size() * 2 + 4;                    // blocks in flight
        std::vector<Block> blocks(n_blocks);
        Queue<Block*> free_q, map_q, fmt_q;
        for (auto& b : blocks) free_q.push(&b);
        std::mutex fmt_mu;
        // the gate of chunk mode: `parsed` = every block below it has been cut into records; `stop` = first block with a malformed record
        struct { std::mutex mu; std::condition_variable cv; uint64_t parsed = 0; std::set<uint64_t> done; uint64_t stop = ~0ull; size_t bad_at = ~(size_t)0; } gate;
        std::atomic<uint64_t> stop_block{ ~0ull };

        std::thread scanner([&] {
            uint64_t idx = 0;
            int ill_state = o.p.illumina;
            Block* b;
            while (!failed && stop_block.load() == ~0ull && free_q.pop(b)) {
                auto s0 = std::chrono::steady_clock::now();
                b->lazy = false; b->malformed = false;
                bool too_long = false;
                const bool more = chunks ? fq.next_chunk(*b, o.batch) : fq.next(*b, o.batch, &too_long);
                t_scan += secs_since(s0);
                if (too_long) failed = 1;
                if (!more || failed) { free_q.push(b); break; }
                b->index = idx++; b->failed = false;
                b->illumina = ill_state;
                if (ill_state)                              // gILLUMINA is cleared for the rest of the run by the first quality below '@'
                    for (uint32_t i = 0; i < b->n && ill_state; ++i)
                        for (uint32_t t = 0; t < b->len[i]; ++t) if ((signed char)b->qual[i][t] < 64) { ill_state = 0; break; }
                map_q.push(b);
            }
            map_q.close();
        });
        std::vector<std::thread> wth;
        std::atomic<int> live_workers{ (int)workers.size() };
        for (size_t k = 0; k < workers.size(); ++k)
            wth.emplace_back([&, k] {
                Worker& w = workers[k];
                Block* b;
                while (map_q.pop(b)) {
                    if (b->lazy) {                              // chunk mode: this worker cuts its byte range into records
                        auto s0 = std::chrono::steady_clock::now();
                        size_t cur = b->text_lo, bad_at = npos; bool stop = false, too_long = false;
                        FastqScanner::parse_range(fq.base, cur, b->text_hi, *b, 16000000u, &stop, &bad_at, &too_long);
                        if (too_long) failed = 1;
                        w.t_scan += secs_since(s0);
                        bool drop;
                        {
                            std::unique_lock<std::mutex> lk(gate.mu);
                            if (stop && b->index < gate.stop) { gate.stop = b->index; gate.bad_at = bad_at; stop_block = b->index; b->malformed = true; }
                            gate.done.insert(b->index);
                            while (gate.done.count(gate.parsed)) { gate.done.erase(gate.parsed); ++gate.parsed; }
                            gate.cv.notify_all();
                            // nothing of this block reaches the GPU before every block below it is known to be well-formed
                            gate.cv.wait(lk, [&] { return gate.parsed >= b->index || gate.stop < b->index || failed.load(); });
                            drop = gate.stop < b->index;
                        }
                        if (drop) b->n = 0;
                    }
                    if (b->n == 0) { b->n_recs = 0; fmt_q.push(b); continue; }
                    if (!failed && process_block_split(w, o, *b, 0) != GM_OK) { failed = 1; gate.cv.notify_all(); }
                    if (failed) { b->failed = true; b->n_recs = 0; }
                    fmt_q.push(b);
                }
                if (--live_workers == 0) fmt_q.close();
            });
        // formatters: a block is cut into slices of records, one string each; the writer emits blocks in index order
        std::mutex wr_mu; std::condition_variable wr_cv; std::map<uint64_t, Block*> ready; bool fmt_done = false;
        const int n_fmt = 2;
        std::atomic<int> live_fmt{ n_fmt };
        std::vector<std::thread> fth;
        for (int f = 0; f < n_fmt; ++f)
            fth.emplace_back([&] {
                Block* b;
                while (fmt_q.pop(b)) {
                    auto f0 = std::chrono::steady_clock::now();
                    const gm_index* ix = gpu_ix[(size_t)b->gpu];
                    const uint32_t nr = (uint32_t)b->n_recs;
                    const int T = o.fmt_threads;
                    if ((int)b->text.size() < T) b->text.resize((size_t)T);
                    for (auto& s : b->text) s.clear();
                    run_slices(nr, T, 4096, [&](int s, uint32_t lo, uint32_t hi) {
                        TextBuf& out = b->text[(size_t)s];
                        out.room((size_t)(hi - lo) * 320);
                        for (uint32_t k = lo; k < hi; ++k) { const gm_sam_rec& r = b->recs[k]; format_sam(out, ix, o.p, r, b->pool.data() + r.cigar_off, *b); }
                    });
                    { std::lock_guard<std::mutex> lk(fmt_mu); t_fmt += secs_since(f0); }
                    { std::lock_guard<std::mutex> lk(wr_mu); ready[b->index] = b; }
                    wr_cv.notify_all();
                }
                if (--live_fmt == 0) { { std::lock_guard<std::mutex> lk(wr_mu); fmt_done = true; } wr_cv.notify_all(); }
            });
        // the writer only hands out file offsets in block order; the text pieces of a block (one per formatter slice) are written by
        // pwrite() from a few threads at once (a single write() stream tops out at ~8 GB/s of page-cache copies)
        std::thread writer([&] {
            uint64_t next = 0;
            for (;;) {
                Block* b = nullptr;
                {
                    std::unique_lock<std::mutex> lk(wr_mu);
                    wr_cv.wait(lk, [&] { return (!ready.empty() && ready.begin()->first == next) || (fmt_done && ready.empty()) || (fmt_done && failed); });
                    if (ready.empty() || ready.begin()->first != next) break;
                    b = ready.begin()->second; ready.erase(ready.begin());
                }
                auto w0 = std::chrono::steady_clock::now();
                if (!b->failed && b->n_recs) {
                    std::vector<uint64_t> offs(b->text.size());
                    for (size_t k = 0; k < b->text.size(); ++k) { offs[k] = file_off; file_off += b->text[k].size(); }
                    const int nt = (int)std::min<size_t>(8, b->text.size());
                    std::atomic<size_t> nextp{ 0 }; std::atomic<int> bad{ 0 };
                    auto job = [&] { for (size_t k; (k = nextp++) < b->text.size();) if (!b->text[k].empty() && !pwrite_all(b->text[k].data(), b->text[k].size(), offs[k])) bad = 1; };
                    std::vector<std::thread> ws;
                    for (int t = 1; t < nt; ++t) ws.emplace_back(job);
                    job();
                    for (auto& x : ws) x.join();
                    if (bad) { fprintf(stderr, "ERROR: write failed\n"); failed = 1; }
                }
                t_write += secs_since(w0);
                ++next;
                free_q.push(b);
            }
            free_q.close();
        });
        scanner.join();
        for (auto& x : wth) x.join();
        for (auto& x : fth) x.join();
        writer.join();
        return gate.bad_at;
    };
    const bool file_order = o.batch_set || o.p.illumina;
    const size_t bad_at = run_pass(!file_order);
    if (!failed && bad_at != npos) {                   // a malformed record: the reference's recovery needs the lines in file order from there on
        fq.at = bad_at; fq.done = false;
        run_pass(false);
    }
    ::close(ofd);
    if (failed) return 1;
    const double t_pipe = secs_since(t_pipe0);
    auto t_cov0 = std::chrono::steady_clock::now();
    // coverage: all-reduce over the GPUs, then PrintFinalSGR / PrintFinalBisulfite
    if (gm_coverage_allreduce(gpu_ix.data(), o.gpus) != GM_OK) { fprintf(stderr, "ERROR: %s\n", gm_last_error()); return 1; }
    PinVec<float> cov; cov.ensure(gm_coverage_bins(gpu_ix[0]));                         // page-locked: the 1.5 GB of a human track come down at link rate
    if (gm_coverage_download(gpu_ix[0], cov.data()) != GM_OK) { fprintf(stderr, "ERROR: %s\n", gm_last_error()); return 1; }
    if (o.p.mode == GM_MODE_NORMAL) {                                  // GenomeBwt::PrintFinal src/GenomeBwt.cpp:915-926
        if (gm_coverage_write_sgr(gpu_ix[0], cov.data(), (o.output + ".sgr").c_str(), 0) != GM_OK) { fprintf(stderr, "ERROR: %s\n", gm_last_error()); return 1; }
    } else {
        std::vector<float> nuc(5 * (size_t)gm_coverage_bins(gpu_ix[0]));
        if (gm_coverage_download_nuc(gpu_ix[0], nuc.data()) != GM_OK ||
            gm_coverage_write_gmp(gpu_ix[0], &o.p, cov.data(), nuc.data(), (o.output + ".gmp").c_str(), 0) != GM_OK) {
            fprintf(stderr, "ERROR: %s\n", gm_last_error());
            return 1;
        }
    }
    uint64_t n_reads = 0, n_matched = 0, n_records = 0;
    double t_pack = 0, t_map = 0, t_out = 0;
    for (auto& w : workers) {
        n_reads += w.n_reads; n_matched += w.n_matched; n_records += w.n_records; t_pack += w.t_pack; t_map += w.t_map; t_out += w.t_out; t_scan += w.t_scan;
        gm_batch_destroy(w.batch);
        gm_stream_destroy(w.ix, w.stream);
    }
    const double t_cov = secs_since(t_cov0);
    for (auto ix : gpu_ix) gm_index_close(ix);
    double secs = secs_since(t0);
    if (o.verbose > 0) {
        fprintf(stderr, "stage seconds (summed over threads): index %.2f, scan %.2f, pack %.2f, gm_map_batch %.2f, gm_output_batch %.2f, SAM format %.2f, write %.2f\n",
                t_index, t_scan, t_pack, t_map, t_out, t_fmt, t_write);
        fprintf(stderr, "wall seconds: index %.2f, FASTQ->SAM pipeline %.2f (%.3f M reads/s), coverage all-reduce + track file %.2f\n", t_index, t_pipe,
                n_reads / std::max(t_pipe, 1e-9) / 1e6, t_cov);
        fprintf(stderr, "Finished: %lu reads, %lu matched, %lu SAM records, %.2f s total (%.2f s after the index was resident)\n", (unsigned long)n_reads,
                (unsigned long)n_matched, (unsigned long)n_records, secs, secs - t_index);
    }
    return 0;
}
