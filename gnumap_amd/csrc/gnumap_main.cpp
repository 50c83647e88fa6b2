// gnumap_main.cpp — the C++ host driver: keeps the reference's command line (src/Driver.cpp:172-250, 2658-3251),
// SAM format (src/Driver.cpp:2146-2217, 2317-2329) and .sgr output (src/GenomeBwt.cpp:1212-1273), and drives the hot
// path through the C ABI of libgnumap_hip.so only.  Per batch it mirrors parallel_thread_run (src/Driver.cpp:2303-2407):
//     parse FASTQ block -> gm_map_batch (= loop over set_top_matches) -> gm_output_batch (= loop over create_match_output)
//     -> SAM text.
// Multi-GPU (--gpus N): one host thread and one index replica per GPU, read blocks dealt round-robin, per-GPU SAM
// text concatenated in block order, coverage tracks combined with one RCCL all-reduce (gm_coverage_allreduce).
#include "gnumap_hip.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#define MAX_NAME_SZ 1024        // inc/const_include.h:46

struct Options {
    std::string genome, output = "gnumap_out", reads;
    gm_params p;
    int gpus = 1, locate_sampled = 0, verbose = 1;
    uint32_t batch = 262144;
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
            "  -u, --unique                 Only match sequences to one position\n"
            "  -G, --gap_penalty=DOUBLE     Gap penalty (default: -4)\n"
            "  -c, --num_proc=INT           accepted for compatibility (the GPU path ignores it)\n"
            "  -b, --bs_seq / --b2 / -d, --a_to_g   bisulfite / A-to-G scoring\n"
            "      --no_nw                  use k-mer hit counts instead of Needleman-Wunsch alignments\n"
            "      --fast, --print_all_sam, --illumina, --up_strand, --down_strand, --bin_size=INT\n"
            "  MI355X options: --gpus=N  --batch=N  --locate=sampled|full\n");
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
            else if (starts(s, "max_gap=")) o.p.max_gap = atoi(s + 8);
            else if (!strcmp(s, "unique")) o.p.unique_only = 1;
            else if (!strcmp(s, "print_all_sam")) o.p.print_all_sam = 1;
            else if (!strcmp(s, "bs_seq")) o.p.mode = GM_MODE_BS;
            else if (!strcmp(s, "b2")) o.p.mode = GM_MODE_BS2;
            else if (!strcmp(s, "a_to_g")) o.p.mode = GM_MODE_ATOG;
            else if (!strcmp(s, "fast")) o.p.fast = 1;
            else if (starts(s, "bin_size=")) o.p.bin_size = atoi(s + 9);
            else if (starts(s, "jump=")) o.p.jump = atoi(s + 5);
            else if (starts(s, "num_seed=")) o.p.min_seed_hits = atoi(s + 9);
            else if (!strcmp(s, "illumina")) o.p.illumina = 1;
            else if (!strcmp(s, "up_strand")) { o.p.pos_strand = 1; o.p.neg_strand = 0; }
            else if (!strcmp(s, "down_strand")) { o.p.pos_strand = 0; o.p.neg_strand = 1; }
            else if (starts(s, "gpus=")) o.gpus = atoi(s + 5);
            else if (starts(s, "batch=")) o.batch = (uint32_t)atoi(s + 6);
            else if (starts(s, "locate=")) o.locate_sampled = !strcmp(s + 7, "sampled");
            else if (!strcmp(s, "help")) usage(0, "");
            else { fprintf(stderr, "No matching arg in: %s\n", a); exit(1); }
            continue;
        }
        if (strlen(a) > 2) { fprintf(stderr, "Irregular Parameter in: %s\n", a); exit(1); }
        char c = a[1];
        bool flag = strchr("prubd0", c) != nullptr;
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
            case 'l': case 'B': case 's': case 'A': case 'S': fprintf(stderr, "option -%c is outside the hot path of this build\n", c); exit(1);
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
    if (o.gpus < 1) o.gpus = 1;
    if (o.batch < 1) o.batch = 1;
}

// ---- FASTQ block reader (SeqReader::get_more_fastq src/SeqReader.cpp:1023-1292: 4-line records, blank names skipped) ----
struct Block {
    uint64_t index = 0;
    std::vector<std::string> names, seqs, quals;
    std::vector<uint8_t> bases, qbuf;
    std::vector<uint16_t> len;
    uint32_t stride = 0;
};

struct FastqReader {
    std::ifstream in;
    bool done = false;
    explicit FastqReader(const std::string& fn) : in(fn.c_str()) {}
    bool ok() const { return in.is_open(); }
    bool next(Block& b, uint32_t max_reads) {
        b.names.clear(); b.seqs.clear(); b.quals.clear();
        if (done) return false;
        std::string name, seq, plus, qual;
        uint32_t maxlen = 0;
        while (b.names.size() < max_reads) {
            if (!std::getline(in, name)) { done = true; break; }
            while (name.empty() && !in.eof()) std::getline(in, name);
            if (in.eof() && name.empty()) { done = true; break; }
            std::getline(in, seq); std::getline(in, plus); std::getline(in, qual);
            if (name.empty() || name[0] != '@' || plus.empty() || plus[0] != '+' || seq.size() > qual.size()) {
                fprintf(stderr, "--ERROR at sequence %s (malformed FASTQ record); stopping here\n", name.c_str());
                done = true;
                break;
            }
            if (seq.size() > 2048) { fprintf(stderr, "read %s longer than 2048 bases\n", name.c_str()); done = true; break; }
            b.names.push_back(name.substr(1));
            maxlen = std::max<uint32_t>(maxlen, (uint32_t)seq.size());
            b.seqs.push_back(seq); b.quals.push_back(qual);
        }
        const size_t n = b.names.size();
        b.stride = std::max<uint32_t>(8, (maxlen + 7u) & ~7u);
        b.bases.assign(n * b.stride, 0); b.qbuf.assign(n * b.stride, 0); b.len.resize(n);
        for (size_t i = 0; i < n; ++i) {
            b.len[i] = (uint16_t)b.seqs[i].size();
            memcpy(&b.bases[i * b.stride], b.seqs[i].data(), b.seqs[i].size());
            memcpy(&b.qbuf[i * b.stride], b.quals[i].data(), b.seqs[i].size());
        }
        return n > 0;
    }
};

static std::string reverse_comp(const std::string& s) {           // SequenceOperations.h:56-96
    std::string t(s.size(), 'n');
    for (size_t i = 0; i < s.size(); ++i) {
        char c = s[s.size() - 1 - i], r;
        switch (c) {
            case 'a': r = 't'; break; case 'A': r = 'T'; break; case 't': r = 'a'; break; case 'T': r = 'A'; break;
            case 'c': r = 'g'; break; case 'C': r = 'G'; break; case 'g': r = 'c'; break; case 'G': r = 'C'; break;
            case '-': r = '-'; break; default: r = 'n'; break;
        }
        t[i] = r;
    }
    return t;
}

static std::string reverse_cigar(const char* s) {                  // SequenceOperations.h:109-123
    std::string out, number;
    for (size_t i = 0; i < strlen(s); ++i) {
        if (s[i] >= 48 && s[i] <= 58) number += s[i];
        else { out = number + s[i] + out; number.clear(); }
    }
    return out;
}

static void format_sam(std::string& out, const gm_index* ix, const gm_params& p, const gm_sam_rec& r, const char* cigar,
                       const std::string& name, const std::string& seq, const std::string& qual) {
    char buf[256];
    out.append(name, 0, std::min<size_t>(name.size(), MAX_NAME_SZ - 1));
    out += r.strand == GM_POS_STRAND ? "\t0\t" : "\t16\t";
    out += gm_index_contig_name(ix, r.contig);
    snprintf(buf, sizeof buf, "\t%lu\t%d\t", (unsigned long)r.chr_pos, r.mapq);
    out += buf;
    if (r.strand == GM_POS_STRAND) { out += cigar; out += "\t*\t0\t0\t"; out += seq; out += '\t'; out += qual; out += '\t'; }
    else {
        out += reverse_cigar(cigar); out += "\t*\t0\t0\t"; out += reverse_comp(seq); out += '\t';
        out.append(qual.rbegin(), qual.rend()); out += '\t';
    }
    snprintf(buf, sizeof buf, "XA:f:%g\tXP:f:%g\tX0:i:%d\n", (double)(float)r.a_score * (1.0 / p.adjust), (double)(float)r.post_prob, r.sim_matches);
    out += buf;
}

struct Worker {
    gm_index* ix = nullptr;
    gm_batch* batch = nullptr;
    std::vector<int8_t> status; std::vector<float> self_score; std::vector<double> top, den; std::vector<uint64_t> mbegin;
    std::vector<gm_match> matches; std::vector<gm_pos> positions; std::vector<gm_sam_rec> recs; std::vector<char> pool;
    uint64_t n_reads = 0, n_matched = 0, n_records = 0;
};

static int process_block(Worker& w, const Options& o, const Block& b, std::string& sam) {
    const uint32_t n = (uint32_t)b.names.size();
    gm_reads reads; reads.n = n; reads.stride = b.stride; reads.bases = b.bases.data(); reads.quals = b.qbuf.data(); reads.len = b.len.data();
    w.status.resize(n); w.self_score.resize(n); w.top.resize(n); w.den.resize(n); w.mbegin.resize(n + 1);
    if (w.matches.size() < 4 * (size_t)n + 64) w.matches.resize(4 * (size_t)n + 64);
    if (w.positions.size() < 8 * (size_t)n + 64) w.positions.resize(8 * (size_t)n + 64);
    gm_hits hits;
    for (;;) {
        hits.n = n; hits.status = w.status.data(); hits.self_score = w.self_score.data(); hits.top_score = w.top.data();
        hits.denominator = w.den.data(); hits.match_begin = w.mbegin.data();
        hits.matches = w.matches.data(); hits.matches_cap = w.matches.size();
        hits.positions = w.positions.data(); hits.positions_cap = w.positions.size();
        int rc = gm_map_batch(w.ix, &o.p, w.batch, &reads, &hits, nullptr);
        if (rc == GM_E_CAPACITY) { w.matches.resize(hits.matches_cap + 64); w.positions.resize(hits.positions_cap + 64); continue; }
        if (rc != GM_OK) { fprintf(stderr, "ERROR: gm_map_batch: %s\n", gm_last_error()); return rc; }
        break;
    }
    if (w.recs.size() < 2 * (size_t)n + 64) w.recs.resize(2 * (size_t)n + 64);
    if (w.pool.size() < 16 * (size_t)n + 1024) w.pool.resize(16 * (size_t)n + 1024);
    gm_sam_out so;
    for (;;) {
        so.recs = w.recs.data(); so.recs_cap = w.recs.size(); so.cigar_pool = w.pool.data(); so.cigar_cap = w.pool.size();
        int rc = gm_output_batch(w.ix, &o.p, w.batch, &reads, &hits, &so, nullptr);
        if (rc == GM_E_CAPACITY) { w.recs.resize(so.recs_cap + 64); w.pool.resize(so.cigar_cap + 64); continue; }
        if (rc != GM_OK) { fprintf(stderr, "ERROR: gm_output_batch: %s\n", gm_last_error()); return rc; }
        break;
    }
    for (uint64_t k = 0; k < so.n_recs; ++k) {
        const gm_sam_rec& r = w.recs[k];
        format_sam(sam, w.ix, o.p, r, w.pool.data() + r.cigar_off, b.names[r.read], b.seqs[r.read], b.quals[r.read]);
    }
    w.n_reads += n; w.n_records += so.n_recs;
    for (uint32_t i = 0; i < n; ++i) w.n_matched += (w.status[i] == GM_READ_OK || w.status[i] == GM_READ_TOO_MANY);
    return GM_OK;
}

int main(int argc, char** argv) {
    Options o;
    parse_args(argc, argv, o);
    std::string cl;
    for (int i = 0; i < argc; ++i) { cl += argv[i]; cl += " "; }      // Driver.cpp:1032-1039
    auto t0 = std::chrono::steady_clock::now();
    const int flags = GM_INDEX_BUILD | (o.locate_sampled ? 0 : GM_INDEX_FULL_SA);
    std::vector<Worker> workers((size_t)o.gpus);
    {   // build once if missing, then one replica per GPU
        gm_index* probe = nullptr;
        int rc = gm_index_open(o.genome.c_str(), 0, flags, &probe);
        if (rc != GM_OK) { fprintf(stderr, "ERROR: %s\n", gm_last_error()); return 1; }
        workers[0].ix = probe;
        for (int g = 1; g < o.gpus; ++g)
            if (gm_index_open(o.genome.c_str(), g, flags, &workers[(size_t)g].ix) != GM_OK) { fprintf(stderr, "ERROR: GPU %d: %s\n", g, gm_last_error()); return 1; }
    }
    for (int g = 0; g < o.gpus; ++g) {
        if (gm_coverage_reset(workers[(size_t)g].ix, (uint32_t)o.p.bin_size) != GM_OK ||
            (o.p.mode != GM_MODE_NORMAL && gm_coverage_enable_nuc(workers[(size_t)g].ix) != GM_OK) ||
            gm_batch_create(workers[(size_t)g].ix, o.batch, 4096, &workers[(size_t)g].batch) != GM_OK) { fprintf(stderr, "ERROR: %s\n", gm_last_error()); return 1; }
    }
    gm_index_info info;
    gm_index_get_info(workers[0].ix, &info);
    if (o.verbose > 0)
        fprintf(stderr, "gnumap-mi355x: genome %s (%lu bp, %u contigs), %s locate, %d GPU(s), index %.1f MB in HBM\n", o.genome.c_str(),
                (unsigned long)info.l_pac, info.n_seqs, info.full_sa ? "full-SA" : "sampled-SA", o.gpus, info.hbm_bytes / 1e6);
    std::ofstream of((o.output + ".sam").c_str(), std::ofstream::out | std::ofstream::binary);
    if (!of) { fprintf(stderr, "ERROR: cannot write %s.sam\n", o.output.c_str()); return 1; }
    for (uint32_t i = 0; i < info.n_seqs; ++i)                         // Driver.cpp:2322-2327
        of << "@SQ\tSN:" << gm_index_contig_name(workers[0].ix, i) << "\tLN:"
           << (gm_index_contig_offset(workers[0].ix, i + 1) - gm_index_contig_offset(workers[0].ix, i)) << "\n";
    of << "@PG\tID:gnumap\tPN:gnumap\tVN:4.0.0 BETA\tCL:" << cl << std::endl;

    FastqReader fq(o.reads);
    if (!fq.ok()) { fprintf(stderr, "ERROR: cannot open %s\n", o.reads.c_str()); return 1; }
    std::mutex rd_mu, wr_mu;
    std::map<uint64_t, std::string> pending;
    uint64_t next_block = 0, next_write = 0;
    std::atomic<int> failed{ 0 };
    auto run = [&](int g) {
        Worker& w = workers[(size_t)g];
        Block b;
        for (;;) {
            {
                std::lock_guard<std::mutex> lk(rd_mu);                // the reference serialises FASTQ parsing too (read_lock)
                if (!fq.next(b, o.batch)) break;
                b.index = next_block++;
            }
            std::string sam;
            if (process_block(w, o, b, sam) != GM_OK) { failed = 1; break; }
            std::lock_guard<std::mutex> lk(wr_mu);
            pending[b.index] = std::move(sam);
            while (!pending.empty() && pending.begin()->first == next_write) {
                of << pending.begin()->second;
                pending.erase(pending.begin());
                ++next_write;
            }
        }
    };
    std::vector<std::thread> th;
    for (int g = 0; g < o.gpus; ++g) th.emplace_back(run, g);
    for (auto& t : th) t.join();
    if (failed) return 1;
    of.close();
    // coverage: all-reduce over the GPUs, then PrintFinalSGR
    std::vector<gm_index*> all;
    for (auto& w : workers) all.push_back(w.ix);
    if (gm_coverage_allreduce(all.data(), o.gpus) != GM_OK) { fprintf(stderr, "ERROR: %s\n", gm_last_error()); return 1; }
    std::vector<float> cov(gm_coverage_bins(workers[0].ix));
    if (gm_coverage_download(workers[0].ix, cov.data()) != GM_OK) { fprintf(stderr, "ERROR: %s\n", gm_last_error()); return 1; }
    if (o.p.mode == GM_MODE_NORMAL) {                                  // GenomeBwt::PrintFinal src/GenomeBwt.cpp:915-926
        if (gm_coverage_write_sgr(workers[0].ix, cov.data(), (o.output + ".sgr").c_str(), 0) != GM_OK) { fprintf(stderr, "ERROR: %s\n", gm_last_error()); return 1; }
    } else {
        std::vector<float> nuc(5 * cov.size());
        if (gm_coverage_download_nuc(workers[0].ix, nuc.data()) != GM_OK ||
            gm_coverage_write_gmp(workers[0].ix, &o.p, cov.data(), nuc.data(), (o.output + ".gmp").c_str(), 0) != GM_OK) {
            fprintf(stderr, "ERROR: %s\n", gm_last_error());
            return 1;
        }
    }
    uint64_t n_reads = 0, n_matched = 0, n_records = 0;
    for (auto& w : workers) { n_reads += w.n_reads; n_matched += w.n_matched; n_records += w.n_records; gm_batch_destroy(w.batch); gm_index_close(w.ix); }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (o.verbose > 0)
        fprintf(stderr, "Finished: %lu reads, %lu matched, %lu SAM records, %.2f s total\n", (unsigned long)n_reads, (unsigned long)n_matched,
                (unsigned long)n_records, secs);
    return 0;
}
