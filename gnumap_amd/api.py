"""ctypes mirror of include/gnumap_hip.h."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

GM_INDEX_FULL_SA, GM_INDEX_BUILD, GM_INDEX_HOST_ONLY = 1, 2, 4
GM_READ_OK, GM_READ_TOO_MANY, GM_READ_NONE, GM_READ_TOO_SHORT, GM_READ_TOO_POOR = 0, 1, 2, -2, -3
GM_E_CAPACITY = -5
GM_E_BATCH_TOO_LARGE = -9

u8p = C.POINTER(C.c_uint8)
u64 = C.c_uint64


class GnumapError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gnumap_hip error {code}: {msg}")
        self.code = code


class gm_index_info(C.Structure):
    _fields_ = [("l_pac", u64), ("seq_len", u64), ("primary", u64), ("bwt_words", u64), ("n_sa", u64),
                ("sa_intv", C.c_uint32), ("n_seqs", C.c_uint32), ("device_id", C.c_int), ("full_sa", C.c_int), ("hbm_bytes", u64)]


class gm_params(C.Structure):
    _fields_ = [("mer", C.c_int), ("jump", C.c_int), ("min_seed_hits", C.c_int), ("max_kmer_hits", C.c_uint32), ("max_matches", C.c_uint32),
                ("max_gap", C.c_int), ("nw", C.c_int), ("fast", C.c_int), ("unique_only", C.c_int), ("pos_strand", C.c_int), ("neg_strand", C.c_int),
                ("mode", C.c_int), ("align_score", C.c_float), ("align_is_fraction", C.c_int), ("cutoff", C.c_float),
                ("adjust", C.c_float), ("match", C.c_float), ("transition", C.c_float), ("transversion", C.c_float), ("gap", C.c_float),
                ("S", (C.c_float * 4) * 256), ("bin_size", C.c_int), ("print_all_sam", C.c_int), ("illumina", C.c_int), ("finalized", C.c_int)]


class gm_reads(C.Structure):
    _fields_ = [("n", C.c_uint32), ("stride", C.c_uint32), ("bases", C.c_void_p), ("quals", C.c_void_p), ("len", C.c_void_p)]


class gm_pos(C.Structure):
    _fields_ = [("pos", u64), ("strand", C.c_uint8)]


class gm_match(C.Structure):
    _fields_ = [("read", C.c_uint32), ("score", C.c_float), ("first_pos", u64), ("first_strand", C.c_uint8),
                ("pos_begin", C.c_uint32), ("pos_end", C.c_uint32)]


class gm_hits(C.Structure):
    _fields_ = [("n", C.c_uint32), ("status", C.c_void_p), ("self_score", C.c_void_p), ("top_score", C.c_void_p), ("denominator", C.c_void_p),
                ("match_begin", C.c_void_p), ("matches", C.c_void_p), ("matches_cap", u64), ("positions", C.c_void_p), ("positions_cap", u64),
                ("stamp", u64)]


class gm_sam_rec(C.Structure):
    _fields_ = [("read", C.c_uint32), ("pos", u64), ("contig", C.c_uint32), ("chr_pos", u64), ("strand", C.c_uint8), ("mapq", C.c_int32),
                ("a_score", C.c_float), ("post_prob", C.c_float), ("sim_matches", C.c_int32), ("cigar_off", C.c_uint32)]


class gm_sam_out(C.Structure):
    _fields_ = [("recs", C.c_void_p), ("recs_cap", u64), ("n_recs", u64), ("cigar_pool", C.c_void_p), ("cigar_cap", u64), ("cigar_len", u64)]


class gm_counters(C.Structure):
    _fields_ = [(n, u64) for n in ("reads", "kmers_searched", "occ_calls", "occ_blocks", "seeds_used", "sa_hits", "lf_steps", "candidates",
                                   "nw_cells", "accepted", "vote_retries", "table_lookups")]

GM_K_COUNT = 7


RAW_HIT_DTYPE = np.dtype([("read", "<u4"), ("pos", "<u4"), ("score", "<f4"), ("step", "<u2"), ("strand", "u1"), ("pad", "u1")])
MATCH_DTYPE = np.dtype([("read", "<u4"), ("score", "<f4"), ("first_pos", "<u8"), ("first_strand", "u1"), ("pos_begin", "<u4"), ("pos_end", "<u4")],
                       align=True)
POS_DTYPE = np.dtype([("pos", "<u8"), ("strand", "u1")], align=True)
SAM_DTYPE = np.dtype([("read", "<u4"), ("pos", "<u8"), ("contig", "<u4"), ("chr_pos", "<u8"), ("strand", "u1"), ("mapq", "<i4"),
                      ("a_score", "<f4"), ("post_prob", "<f4"), ("sim_matches", "<i4"), ("cigar_off", "<u4")], align=True)

# every symbol include/gnumap_hip.h declares
EXPORTS = ["gm_last_error", "gm_version", "gm_set_option", "gm_selftest_pass_parallel", "gm_index_build", "gm_index_build_on", "gm_index_open", "gm_index_close", "gm_index_prepare", "gm_index_get_info", "gm_index_contig_name",
           "gm_index_contig_offset", "gm_index_window", "gm_params_default", "gm_params_finalize", "gm_params_load_subst", "gm_batch_create", "gm_batch_destroy",
           "gm_batch_upload", "gm_map_batch_device", "gm_batch_counters", "gm_batch_path", "gm_batch_set_profiling", "gm_batch_kernel_times", "gm_kernel_name",
           "gm_batch_raw_hits", "gm_stream_create", "gm_stream_destroy", "gm_host_alloc", "gm_host_free", "gm_map_batch", "gm_output_batch",
           "gm_map_batch_enqueue", "gm_output_batch_enqueue", "gm_batch_wait",
           "gm_dev_sa_interval", "gm_dev_locate", "gm_dev_nw_score", "gm_dev_traceback", "gm_dev_pair_hmm", "gm_coverage_reset", "gm_coverage_bins",
           "gm_coverage_device_ptr", "gm_coverage_add", "gm_coverage_download", "gm_coverage_allreduce", "gm_coverage_write_sgr", "gm_coverage_enable_nuc", "gm_coverage_nuc_device_ptr",
           "gm_coverage_download_nuc", "gm_coverage_write_gmp"]


def library_path():
    return os.path.join(HERE, "libgnumap_hip.so")


def load_library():
    """Load the in-tree libgnumap_hip.so.  Fails loudly when it has not been built (no silent fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise GnumapError(-3, f"{path} is missing: build it with `make -C gnumap_amd` (or __graft_entry__.build())")
    L = C.CDLL(path)
    L.gm_last_error.restype = C.c_char_p
    L.gm_version.restype = C.c_char_p
    L.gm_set_option.argtypes = [C.c_char_p, C.c_char_p]
    L.gm_index_build.argtypes = [C.c_char_p]
    L.gm_index_build_on.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.gm_index_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.gm_index_close.argtypes = [C.c_void_p]; L.gm_index_close.restype = None
    L.gm_index_prepare.argtypes = [C.c_void_p, C.POINTER(gm_params)]
    L.gm_index_get_info.argtypes = [C.c_void_p, C.POINTER(gm_index_info)]
    L.gm_index_contig_name.argtypes = [C.c_void_p, C.c_uint32]; L.gm_index_contig_name.restype = C.c_char_p
    L.gm_index_contig_offset.argtypes = [C.c_void_p, C.c_uint32]; L.gm_index_contig_offset.restype = u64
    L.gm_index_window.argtypes = [C.c_void_p, u64, C.c_uint32, C.c_char_p]
    L.gm_params_default.argtypes = [C.POINTER(gm_params)]; L.gm_params_default.restype = None
    L.gm_params_finalize.argtypes = [C.POINTER(gm_params)]
    L.gm_params_load_subst.argtypes = [C.POINTER(gm_params), C.c_char_p]
    L.gm_batch_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.gm_batch_destroy.argtypes = [C.c_void_p]; L.gm_batch_destroy.restype = None
    L.gm_batch_upload.argtypes = [C.c_void_p, C.POINTER(gm_params), C.POINTER(gm_reads), C.c_void_p]
    L.gm_map_batch_device.argtypes = [C.c_void_p, C.POINTER(gm_params), C.c_void_p, C.c_void_p]
    L.gm_batch_counters.argtypes = [C.c_void_p, C.POINTER(gm_counters)]
    L.gm_batch_path.argtypes = [C.c_void_p]; L.gm_batch_path.restype = C.c_char_p
    L.gm_map_batch_enqueue.argtypes = [C.c_void_p, C.POINTER(gm_params), C.c_void_p, C.POINTER(gm_reads), C.POINTER(gm_hits), C.c_void_p]
    L.gm_output_batch_enqueue.argtypes = [C.c_void_p, C.POINTER(gm_params), C.c_void_p, C.POINTER(gm_reads), C.POINTER(gm_hits), C.POINTER(gm_sam_out), C.c_void_p]
    L.gm_batch_wait.argtypes = [C.c_void_p]
    L.gm_batch_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.gm_batch_kernel_times.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.gm_kernel_name.argtypes = [C.c_int]; L.gm_kernel_name.restype = C.c_char_p
    L.gm_batch_raw_hits.argtypes = [C.c_void_p, C.c_void_p, u64, C.POINTER(u64), C.c_void_p, C.c_void_p, C.c_void_p]
    L.gm_stream_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.gm_stream_destroy.argtypes = [C.c_void_p, C.c_void_p]; L.gm_stream_destroy.restype = None
    L.gm_host_alloc.argtypes = [C.c_size_t]; L.gm_host_alloc.restype = C.c_void_p
    L.gm_host_free.argtypes = [C.c_void_p]; L.gm_host_free.restype = None
    L.gm_map_batch.argtypes = [C.c_void_p, C.POINTER(gm_params), C.c_void_p, C.POINTER(gm_reads), C.POINTER(gm_hits), C.c_void_p]
    L.gm_output_batch.argtypes = [C.c_void_p, C.POINTER(gm_params), C.c_void_p, C.POINTER(gm_reads), C.POINTER(gm_hits), C.POINTER(gm_sam_out), C.c_void_p]
    L.gm_dev_sa_interval.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.gm_dev_locate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    L.gm_dev_nw_score.argtypes = [C.c_void_p, C.POINTER(gm_params), C.POINTER(gm_reads), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.gm_dev_traceback.argtypes = [C.c_void_p, C.POINTER(gm_params), C.POINTER(gm_reads), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                   C.c_void_p, C.c_uint32, C.c_void_p]
    L.gm_dev_pair_hmm.argtypes = [C.c_void_p, C.POINTER(gm_params), C.POINTER(gm_reads), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.gm_coverage_reset.argtypes = [C.c_void_p, C.c_uint32]
    L.gm_coverage_bins.argtypes = [C.c_void_p]; L.gm_coverage_bins.restype = u64
    L.gm_coverage_device_ptr.argtypes = [C.c_void_p]; L.gm_coverage_device_ptr.restype = C.c_void_p
    L.gm_coverage_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.gm_coverage_download.argtypes = [C.c_void_p, C.c_void_p]
    L.gm_coverage_allreduce.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.gm_coverage_write_sgr.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    L.gm_coverage_enable_nuc.argtypes = [C.c_void_p]
    L.gm_coverage_nuc_device_ptr.argtypes = [C.c_void_p]; L.gm_coverage_nuc_device_ptr.restype = C.c_void_p
    L.gm_coverage_download_nuc.argtypes = [C.c_void_p, C.c_void_p]
    L.gm_coverage_write_gmp.argtypes = [C.c_void_p, C.POINTER(gm_params), C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    _LIB = L
    return L


def lib():
    return load_library()


def _chk(rc):
    if rc != 0:
        raise GnumapError(rc, lib().gm_last_error().decode())


def version():
    return lib().gm_version().decode()


def set_option(name, value):
    """gm_set_option: override a GM_* run-time switch for the calls that follow (None: back to the environment)"""
    _chk(lib().gm_set_option(name.encode(), None if value is None else str(value).encode()))


GM_BUILD_AUTO, GM_BUILD_HOST, GM_BUILD_DEVICE = 0, 1, 2


def index_build(fasta, where=GM_BUILD_AUTO, device=0):
    """write <fasta>.gnumap.{pac,ann,amb,bwt,sa}; the suffix-array stage runs on the MI355X when there is one (same bytes)"""
    _chk(lib().gm_index_build_on(os.fsencode(fasta), where, device))


class Params:
    """gm_params with the reference's defaults (inc/const_define.h); keyword overrides, then finalize."""

    def __init__(self, **kw):
        self.c = gm_params()
        lib().gm_params_default(C.byref(self.c))
        for k, v in kw.items():
            if not hasattr(self.c, k):
                raise AttributeError(k)
            setattr(self.c, k, v)
        _chk(lib().gm_params_finalize(C.byref(self.c)))

    def load_subst(self, path):
        """-S / --subst_file: overwrite the lowercase rows of the score table from a 5 x 4 file (readPWM, Driver.cpp:768-859)"""
        _chk(lib().gm_params_load_subst(C.byref(self.c), os.fsencode(path)))
        return self

    def __getattr__(self, k):
        return getattr(self.c, k)


def pack_reads(seqs, quals, stride=None):
    """list of bytes -> (bases[n,stride] u8, quals[n,stride] u8, len[n] u16)"""
    n = len(seqs)
    mx = max([len(s) for s in seqs] + [1])
    stride = stride or ((mx + 7) // 8) * 8
    B = np.zeros((n, stride), np.uint8); Q = np.zeros((n, stride), np.uint8); Ln = np.zeros(n, np.uint16)
    for i, (s, q) in enumerate(zip(seqs, quals)):
        B[i, :len(s)] = np.frombuffer(s, np.uint8); Q[i, :len(s)] = np.frombuffer(q[:len(s)], np.uint8); Ln[i] = len(s)
    return B, Q, Ln


def _reads_struct(B, Q, Ln):
    r = gm_reads()
    r.n = B.shape[0]; r.stride = B.shape[1] if B.ndim == 2 else 0
    r.bases = B.ctypes.data; r.quals = Q.ctypes.data; r.len = Ln.ctypes.data
    return r


class Index:
    def __init__(self, fasta, device=0, flags=GM_INDEX_FULL_SA):
        self.h = C.c_void_p()
        _chk(lib().gm_index_open(os.fsencode(fasta), device, flags, C.byref(self.h)))
        self.info = gm_index_info()
        _chk(lib().gm_index_get_info(self.h, C.byref(self.info)))

    def close(self):
        if self.h:
            lib().gm_index_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prepare(self, params):
        """build the k-mer tables / records for these parameters now (gm_index_prepare) instead of inside the first map call"""
        _chk(lib().gm_index_prepare(self.h, C.byref(params.c)))

    def contigs(self):
        return [(lib().gm_index_contig_name(self.h, i).decode(), lib().gm_index_contig_offset(self.h, i)) for i in range(self.info.n_seqs)]

    def window(self, begin, L):
        buf = C.create_string_buffer(L + 1)
        lib().gm_index_window(self.h, begin, L, buf)
        return buf.value

    # ---- unit-level device entry points ----
    def dev_sa_interval(self, kmers):
        m = len(kmers[0]); n = len(kmers)
        flat = b"".join(kmers)
        s = np.zeros(n, np.uint64); e = np.zeros(n, np.uint64)
        _chk(lib().gm_dev_sa_interval(self.h, flat, n, m, s.ctypes.data, e.ctypes.data))
        return s, e

    def dev_locate(self, ranks, use_full_sa):
        ranks = np.ascontiguousarray(ranks, np.uint64); out = np.zeros(len(ranks), np.uint64)
        _chk(lib().gm_dev_locate(self.h, ranks.ctypes.data, len(ranks), int(use_full_sa), out.ctypes.data))
        return out

    def dev_nw_score(self, params, B, Q, Ln, read_idx, strand, pos):
        r = _reads_struct(B, Q, Ln)
        read_idx = np.ascontiguousarray(read_idx, np.uint32); strand = np.ascontiguousarray(strand, np.uint8); pos = np.ascontiguousarray(pos, np.uint64)
        n = len(read_idx); score = np.zeros(n, np.float32); valid = np.zeros(n, np.uint8)
        _chk(lib().gm_dev_nw_score(self.h, C.byref(params.c), C.byref(r), read_idx.ctypes.data, strand.ctypes.data, pos.ctypes.data, n,
                                   score.ctypes.data, valid.ctypes.data))
        return score, valid

    def dev_traceback(self, params, B, Q, Ln, read_idx, strand, pos):
        r = _reads_struct(B, Q, Ln)
        read_idx = np.ascontiguousarray(read_idx, np.uint32); strand = np.ascontiguousarray(strand, np.uint8); pos = np.ascontiguousarray(pos, np.uint64)
        n = len(read_idx); stride = 2 * B.shape[1] + 16
        ops = np.zeros((n, stride), np.uint8); ln = np.zeros(n, np.uint16)
        _chk(lib().gm_dev_traceback(self.h, C.byref(params.c), C.byref(r), read_idx.ctypes.data, strand.ctypes.data, pos.ctypes.data, n,
                                    ops.ctypes.data, stride, ln.ctypes.data))
        return [ops[i, :ln[i]].tobytes() for i in range(n)]

    def dev_pair_hmm(self, params, B, Q, Ln, read_idx, strand, pos):
        """bin_seq::pairHMM of read read_idx[k] (oriented by strand[k]) against the window at pos[k]: [n][stride][5] floats"""
        r = _reads_struct(B, Q, Ln)
        read_idx = np.ascontiguousarray(read_idx, np.uint32); strand = np.ascontiguousarray(strand, np.uint8); pos = np.ascontiguousarray(pos, np.uint64)
        out = np.zeros((len(read_idx), B.shape[1], 5), np.float32)
        _chk(lib().gm_dev_pair_hmm(self.h, C.byref(params.c), C.byref(r), read_idx.ctypes.data, strand.ctypes.data, pos.ctypes.data, len(read_idx), out.ctypes.data))
        return out

    # ---- coverage ----
    def coverage_reset(self, bin_size):
        _chk(lib().gm_coverage_reset(self.h, bin_size))

    def coverage_bins(self):
        return lib().gm_coverage_bins(self.h)

    def coverage_device_ptr(self):
        return lib().gm_coverage_device_ptr(self.h)

    def coverage_enable_nuc(self):
        """-b / -d: the five per-nucleotide arrays reads[A,C,G,T,N][loc] in HBM (call after coverage_reset)"""
        _chk(lib().gm_coverage_enable_nuc(self.h))

    def coverage_nuc_device_ptr(self):
        return lib().gm_coverage_nuc_device_ptr(self.h)

    def coverage_add(self, pos, span, w, stream=None):
        """gm_coverage_add: amount_genome[(pos+t)/bin] += w for t < span (GenomeBwt::AddScore)"""
        pos = np.ascontiguousarray(pos, np.uint64); span = np.ascontiguousarray(span, np.uint32); w = np.ascontiguousarray(w, np.float32)
        _chk(lib().gm_coverage_add(self.h, pos.ctypes.data, span.ctypes.data, w.ctypes.data, len(pos), stream))

    def coverage_download(self):
        out = np.zeros(self.coverage_bins(), np.float32)
        _chk(lib().gm_coverage_download(self.h, out.ctypes.data))
        return out

    def coverage_download_nuc(self):
        """the five per-nucleotide tracks (a, c, g, t, n), [5 * bins]"""
        out = np.zeros(5 * self.coverage_bins(), np.float32)
        _chk(lib().gm_coverage_download_nuc(self.h, out.ctypes.data))
        return out

    def coverage_write_sgr(self, bins, path):
        bins = np.ascontiguousarray(bins, np.float32)
        _chk(lib().gm_coverage_write_sgr(self.h, bins.ctypes.data, os.fsencode(path), 0))


class Batch:
    def __init__(self, index, max_reads, max_len):
        self.index = index
        self.h = C.c_void_p()
        _chk(lib().gm_batch_create(index.h, max_reads, max_len, C.byref(self.h)))
        self._keep = None

    def destroy(self):
        if self.h:
            lib().gm_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def upload(self, params, B, Q, Ln, stream=None):
        self._keep = (B, Q, Ln)
        r = _reads_struct(B, Q, Ln)
        _chk(lib().gm_batch_upload(self.h, C.byref(params.c), C.byref(r), stream))
        self.n = B.shape[0]

    def map_device(self, params, stream=None):
        _chk(lib().gm_map_batch_device(self.index.h, C.byref(params.c), self.h, stream))

    def counters(self):
        c = gm_counters()
        _chk(lib().gm_batch_counters(self.h, C.byref(c)))
        return {n: getattr(c, n) for n, _ in gm_counters._fields_}

    def path(self):
        """which kernels the last map_device / map chose"""
        return lib().gm_batch_path(self.h).decode()

    def set_profiling(self, on=True):
        _chk(lib().gm_batch_set_profiling(self.h, int(on)))

    def kernel_times(self):
        """{kernel name: (total ms, launches)} accumulated since the last call"""
        ms = np.zeros(GM_K_COUNT, np.float64); ln = np.zeros(GM_K_COUNT, np.uint64)
        _chk(lib().gm_batch_kernel_times(self.h, ms.ctypes.data, ln.ctypes.data))
        return {lib().gm_kernel_name(i).decode(): (float(ms[i]), int(ln[i])) for i in range(GM_K_COUNT)}

    def raw_hits(self):
        n = self.n
        status = np.zeros(n, np.int8); self_score = np.zeros(n, np.float32); top = np.zeros(n, np.float32)
        cap = 1024
        while True:
            out = np.zeros(cap, RAW_HIT_DTYPE); got = u64()
            rc = lib().gm_batch_raw_hits(self.h, out.ctypes.data, cap, C.byref(got), status.ctypes.data, self_score.ctypes.data, top.ctypes.data)
            if rc == GM_E_CAPACITY:
                cap = int(got.value) + 16
                continue
            _chk(rc)
            return out[:got.value], status, self_score, top

    def map(self, params, B, Q, Ln, stream=None):
        """gm_map_batch: returns dict(status, self_score, top_score, denominator, match_begin, matches, positions)"""
        n = B.shape[0]
        self._keep = (B, Q, Ln); self.n = n
        r = _reads_struct(B, Q, Ln)
        status = np.zeros(n, np.int8); self_score = np.zeros(n, np.float32); top = np.zeros(n, np.float64); den = np.zeros(n, np.float64)
        mbegin = np.zeros(n + 1, np.uint64)
        mcap, pcap = 4 * n + 64, 8 * n + 64
        while True:
            matches = np.zeros(mcap, MATCH_DTYPE); positions = np.zeros(pcap, POS_DTYPE)
            h = gm_hits()
            h.n = n; h.status = status.ctypes.data; h.self_score = self_score.ctypes.data; h.top_score = top.ctypes.data
            h.denominator = den.ctypes.data; h.match_begin = mbegin.ctypes.data
            h.matches = matches.ctypes.data; h.matches_cap = mcap; h.positions = positions.ctypes.data; h.positions_cap = pcap
            rc = lib().gm_map_batch(self.index.h, C.byref(params.c), self.h, C.byref(r), C.byref(h), stream)
            if rc == GM_E_CAPACITY:
                mcap, pcap = int(h.matches_cap) + 64, int(h.positions_cap) + 64
                continue
            _chk(rc)
            break
        res = dict(status=status, self_score=self_score, top_score=top, denominator=den, match_begin=mbegin,
                   matches=matches[:int(mbegin[n])], positions=positions, _struct=h, _reads=r)
        return res

    def output(self, params, res, stream=None):
        """gm_output_batch on the result of map(): returns (records ndarray, list of CIGAR bytes per record)"""
        rcap, ccap = 2 * self.n + 64, 32 * self.n + 1024
        while True:
            recs = np.zeros(rcap, SAM_DTYPE); pool = np.zeros(ccap, np.uint8)
            so = gm_sam_out()
            so.recs = recs.ctypes.data; so.recs_cap = rcap; so.cigar_pool = pool.ctypes.data; so.cigar_cap = ccap
            rc = lib().gm_output_batch(self.index.h, C.byref(params.c), self.h, C.byref(res["_reads"]), C.byref(res["_struct"]), C.byref(so), stream)
            if rc == GM_E_CAPACITY:
                rcap, ccap = int(so.recs_cap) + 64, int(so.cigar_cap) + 64
                continue
            _chk(rc)
            break
        recs = recs[:so.n_recs]
        raw = pool.tobytes()
        cigars = [raw[o:raw.index(b"\0", o)] for o in recs["cigar_off"]]
        return recs, cigars


def pinned_empty(shape, dtype):
    """numpy array over page-locked host memory from gm_host_alloc (freed when the array's owner object is collected)"""
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dt.itemsize
    ptr = lib().gm_host_alloc(max(nbytes, 1))
    if not ptr:
        raise GnumapError(-7, lib().gm_last_error().decode())
    buf = (C.c_uint8 * max(nbytes, 1)).from_address(ptr)
    arr = np.frombuffer(buf, dtype=np.uint8, count=nbytes).view(dt).reshape(shape)
    _PINNED[id(buf)] = (buf, ptr)
    return arr


_PINNED = {}


class BlockRunner:
    """The two ABI calls of one block loop iteration (gm_map_batch + gm_output_batch) on caller-owned, page-locked, REUSED buffers:
    what a host driver thread does per block.  One BlockRunner = one gm_batch + one HIP stream + one set of output buffers."""

    def __init__(self, index, params, max_reads, stride, stream=None):
        self.index, self.params, self.max_reads, self.stride, self.stream = index, params, max_reads, stride, stream
        self.batch = Batch(index, max_reads, stride)
        n = max_reads
        self.status = pinned_empty(n, np.int8); self.self_score = pinned_empty(n, np.float32)
        self.top = pinned_empty(n, np.float64); self.den = pinned_empty(n, np.float64); self.mbegin = pinned_empty(n + 1, np.uint64)
        self._alloc_hits(2 * n + 64, 2 * n + 64)
        self._alloc_out(2 * n + 64, 16 * n + 1024)

    def _alloc_hits(self, mcap, pcap):
        self.matches = pinned_empty(mcap, MATCH_DTYPE); self.positions = pinned_empty(pcap, POS_DTYPE)

    def _alloc_out(self, rcap, ccap):
        self.recs = pinned_empty(rcap, SAM_DTYPE); self.pool = pinned_empty(ccap, np.uint8)

    def run(self, B, Q, Ln):
        """B, Q: [n, stride] uint8 (ideally page-locked), Ln: [n] uint16.  Returns (n_matches, n_records)."""
        n = B.shape[0]
        r = _reads_struct(B, Q, Ln)
        L = lib()
        while True:
            h = gm_hits()
            h.n = n; h.status = self.status.ctypes.data; h.self_score = self.self_score.ctypes.data; h.top_score = self.top.ctypes.data
            h.denominator = self.den.ctypes.data; h.match_begin = self.mbegin.ctypes.data
            h.matches = self.matches.ctypes.data; h.matches_cap = len(self.matches)
            h.positions = self.positions.ctypes.data; h.positions_cap = len(self.positions)
            rc = L.gm_map_batch(self.index.h, C.byref(self.params.c), self.batch.h, C.byref(r), C.byref(h), self.stream)
            if rc == GM_E_CAPACITY:
                self._alloc_hits(int(h.matches_cap) + 64, int(h.positions_cap) + 64)
                continue
            _chk(rc)
            break
        while True:
            so = gm_sam_out()
            so.recs = self.recs.ctypes.data; so.recs_cap = len(self.recs); so.cigar_pool = self.pool.ctypes.data; so.cigar_cap = len(self.pool)
            rc = L.gm_output_batch(self.index.h, C.byref(self.params.c), self.batch.h, C.byref(r), C.byref(h), C.byref(so), self.stream)
            if rc == GM_E_CAPACITY:
                self._alloc_out(int(so.recs_cap) + 64, int(so.cigar_cap) + 64)
                continue
            _chk(rc)
            break
        return int(self.mbegin[n]), int(so.n_recs)

    def run_async(self, B, Q, Ln):
        """queue gm_map_batch + gm_output_batch of this block on the batch's service thread (gm_*_enqueue) and return at once; wait()
        gives (n_matches, n_records).  The buffers must have room (a first synchronous run() sizes them)."""
        n = B.shape[0]
        self._n = n
        self._r = _reads_struct(B, Q, Ln); self._keep = (B, Q, Ln)
        h = self._h = gm_hits()
        h.n = n; h.status = self.status.ctypes.data; h.self_score = self.self_score.ctypes.data; h.top_score = self.top.ctypes.data
        h.denominator = self.den.ctypes.data; h.match_begin = self.mbegin.ctypes.data
        h.matches = self.matches.ctypes.data; h.matches_cap = len(self.matches)
        h.positions = self.positions.ctypes.data; h.positions_cap = len(self.positions)
        so = self._so = gm_sam_out()
        so.recs = self.recs.ctypes.data; so.recs_cap = len(self.recs); so.cigar_pool = self.pool.ctypes.data; so.cigar_cap = len(self.pool)
        L = lib()
        _chk(L.gm_map_batch_enqueue(self.index.h, C.byref(self.params.c), self.batch.h, C.byref(self._r), C.byref(h), self.stream))
        _chk(L.gm_output_batch_enqueue(self.index.h, C.byref(self.params.c), self.batch.h, C.byref(self._r), C.byref(h), C.byref(so), self.stream))

    def wait(self):
        rc = lib().gm_batch_wait(self.batch.h)
        if rc == GM_E_CAPACITY:                      # the buffers were too small after all: size them and run the block synchronously
            self._alloc_hits(max(int(self._h.matches_cap), len(self.matches)) + 64, max(int(self._h.positions_cap), len(self.positions)) + 64)
            self._alloc_out(max(int(self._so.recs_cap), len(self.recs)) + 64, max(int(self._so.cigar_cap), len(self.pool)) + 64)
            return self.run(*self._keep)
        _chk(rc)
        return int(self.mbegin[self._n]), int(self._so.n_recs)
