"""ctypes bindings for the TEST-ONLY checkers: oracle/_ref/libgnumap_ref.so (the unmodified reference
functions, built by oracle/Makefile where /root/reference exists) and oracle/libgm_oracle.so (the CPU
restatement).  Never imported by the product."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libgnumap_ref.so")
ORACLE_SO = os.path.join(ROOT, "oracle", "libgm_oracle.so")

u64 = C.c_uint64
fptr = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def revcomp_pwm(P):
    return np.ascontiguousarray(P[::-1, ::-1])


_COMP = bytes.maketrans(b"acgtACGT", b"tgcaTGCA")


def revcomp_str(s: bytes) -> bytes:
    out = bytearray()
    for ch in reversed(s):
        c = bytes([ch])
        if c in b"acgtACGT":
            out += c.translate(_COMP)
        elif c == b"-":
            out += b"-"
        else:
            out += b"n"
    return bytes(out)


def have_ref():
    return os.path.exists(REF_SO)


class RefLib:
    _setup_mode = None

    def __init__(self):
        self.lib = L = C.CDLL(REF_SO)
        L.ref_setup.argtypes = [C.c_int]
        L.ref_get_scores.argtypes = [fptr, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.ref_index_build.argtypes = [C.c_char_p]
        L.ref_index_load.argtypes = [C.c_char_p]; L.ref_index_load.restype = C.c_void_p
        L.ref_index_free.argtypes = [C.c_void_p]
        for f in ("ref_seq_len", "ref_primary"):
            getattr(L, f).argtypes = [C.c_void_p]; getattr(L, f).restype = u64
        L.ref_occ.argtypes = [C.c_void_p, u64, C.c_int]; L.ref_occ.restype = u64
        L.ref_sa_interval.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(u64), C.POINTER(u64)]
        L.ref_sa_coord.argtypes = [C.c_void_p, u64]; L.ref_sa_coord.restype = u64
        L.ref_window.argtypes = [C.c_void_p, u64, C.c_uint, C.c_char_p]
        L.ref_pos2rid.argtypes = [C.c_void_p, C.c_int64]
        L.ref_nw_score.argtypes = [fptr, C.c_int, C.c_char_p]; L.ref_nw_score.restype = C.c_float
        L.ref_self_score.argtypes = [fptr, C.c_int, C.c_char_p]; L.ref_self_score.restype = C.c_float
        L.ref_align_score_be.argtypes = [fptr, C.c_int, C.c_char_p, C.c_uint, C.c_uint]; L.ref_align_score_be.restype = C.c_float
        L.ref_traceback.argtypes = [fptr, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_char_p]
        if hasattr(L, "ref_pair_hmm"):
            L.ref_pair_hmm.argtypes = [fptr, C.c_int, C.c_char_p, C.c_char_p, fptr]
        L.ref_read_fastq.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, fptr,
                                     np.ctypeslib.ndpointer(np.int32), C.c_char_p, C.c_char_p, C.c_char_p]
        L.ref_reverse_comp.argtypes = [C.c_char_p, C.c_char_p]
        L.ref_reverse_cigar.argtypes = [C.c_char_p, C.c_char_p]
        L.ref_fix_cigar.argtypes = [C.c_char_p, C.c_char_p]

    def setup(self, mode):
        # the reference scales its globals in place, so one process = one mode
        if RefLib._setup_mode is None:
            assert self.lib.ref_setup(mode) == 0
            RefLib._setup_mode = mode
        assert RefLib._setup_mode == mode, "reference tables already set up for another mode in this process"

    def get_scores(self):
        t = np.zeros((256, 4), np.float32); g = C.c_float(); m = C.c_int()
        self.lib.ref_get_scores(t, C.byref(g), C.byref(m))
        return t, g.value, m.value

    def index_build(self, fa):
        return self.lib.ref_index_build(fa.encode())

    def index_load(self, fa):
        return self.lib.ref_index_load(fa.encode())

    def sa_interval(self, ix, kmer: bytes):
        s = u64(); e = u64()
        self.lib.ref_sa_interval(ix, kmer, len(kmer), C.byref(s), C.byref(e))
        return s.value, e.value

    def window(self, ix, begin, L):
        buf = C.create_string_buffer(L + 8)
        self.lib.ref_window(ix, begin, L, buf)
        return buf.value

    def nw_score(self, P, w: bytes):
        P = np.ascontiguousarray(P, np.float32)
        return self.lib.ref_nw_score(P, len(P), w)

    def self_score(self, P, cons: bytes):
        P = np.ascontiguousarray(P, np.float32)
        return self.lib.ref_self_score(P, len(P), cons)

    def traceback(self, P, cons: bytes, w: bytes):
        P = np.ascontiguousarray(P, np.float32)
        L = len(P)
        al = C.create_string_buffer(2 * L + 8); n = C.c_int(); cg = C.create_string_buffer(1024)
        self.lib.ref_traceback(P, L, cons, w, al, C.byref(n), cg)
        return al.raw[:n.value], n.value, cg.value

    def pair_hmm(self, P, cons: bytes, w: bytes):
        P = np.ascontiguousarray(P, np.float32)
        out = np.zeros((len(w), 5), np.float32)
        self.lib.ref_pair_hmm(P, len(P), cons, w, out)
        return out

    def read_fastq(self, fq, illumina, cap, max_len):
        pwm = np.zeros((cap, max_len, 4), np.float32); lens = np.zeros(cap, np.int32)
        seq = C.create_string_buffer(cap * (max_len + 1)); q = C.create_string_buffer(cap * (max_len + 1))
        names = C.create_string_buffer(cap * 256)
        n = self.lib.ref_read_fastq(fq.encode(), illumina, cap, max_len, pwm, lens, seq, q, names)
        assert n >= 0, n
        S = [seq.raw[i * (max_len + 1):(i + 1) * (max_len + 1)].split(b"\0")[0] for i in range(n)]
        Q = [q.raw[i * (max_len + 1):(i + 1) * (max_len + 1)].split(b"\0")[0] for i in range(n)]
        N = [names.raw[i * 256:(i + 1) * 256].split(b"\0")[0].decode() for i in range(n)]
        return n, lens, pwm, S, Q, N


# ---------------------------------------------------------------------------------------------
class GmoContig(C.Structure):
    _fields_ = [("name", C.c_char_p), ("offset", u64), ("len", C.c_uint32)]


class GmoIndex(C.Structure):
    _fields_ = [("primary", u64), ("L2", u64 * 5), ("seq_len", u64), ("bwt_size", u64), ("bwt", C.POINTER(C.c_uint32)),
                ("sa_intv", u64), ("n_sa", u64), ("sa", C.POINTER(u64)), ("l_pac", u64), ("pac", C.POINTER(C.c_uint8)),
                ("n_seqs", C.c_int), ("contigs", C.POINTER(GmoContig))]


class GmoParams(C.Structure):
    _fields_ = [("mer", C.c_int), ("jump", C.c_int), ("min_seed_hits", C.c_int),
                ("max_kmer_hits", C.c_uint32), ("max_matches", C.c_uint32),
                ("max_gap", C.c_int), ("nw", C.c_int), ("fast", C.c_int), ("unique_only", C.c_int),
                ("pos_strand", C.c_int), ("neg_strand", C.c_int), ("mode", C.c_int),
                ("align_score", C.c_float), ("align_is_fraction", C.c_int), ("cutoff", C.c_float),
                ("adjust", C.c_float), ("match", C.c_float), ("transition", C.c_float), ("transversion", C.c_float), ("gap", C.c_float),
                ("S", (C.c_float * 4) * 256),
                ("bin_size", C.c_int), ("print_all_sam", C.c_int), ("illumina", C.c_int)]


class GmoCounters(C.Structure):
    _fields_ = [(n, u64) for n in ("kmers", "occ_calls", "occ_blocks", "locates", "lf_steps", "nw", "tracebacks")]


class GmoPos(C.Structure):
    _fields_ = [("pos", u64), ("strand", C.c_int)]


class GmoHit(C.Structure):
    _fields_ = [("key", C.c_char_p), ("seq", C.c_char_p), ("score", C.c_double), ("first_strand", C.c_int),
                ("pos", C.POINTER(GmoPos)), ("n_pos", C.c_int), ("cap_pos", C.c_int)]


class GmoResult(C.Structure):
    _fields_ = [("status", C.c_int), ("self_score", C.c_float), ("min_score", C.c_double), ("top_score", C.c_double),
                ("denominator", C.c_double), ("hits", C.POINTER(GmoHit)), ("n_hits", C.c_int), ("cap_hits", C.c_int),
                ("ctr", GmoCounters)]


class GmoSam(C.Structure):
    _fields_ = [("pos", u64), ("strand", C.c_int), ("contig", C.c_int), ("chr_pos", u64), ("mapq", C.c_int), ("cigar", C.c_char * 1024),
                ("a_score", C.c_float), ("post_prob", C.c_float), ("sim_matches", C.c_int)]


class GmoDeposit(C.Structure):
    _fields_ = [("pos", u64), ("span", C.c_uint32), ("w", C.c_float), ("codes", C.POINTER(C.c_uint8)), ("hmm", C.POINTER(C.c_float))]


class GmoRunStats(C.Structure):
    _fields_ = [("n_reads", u64), ("n_matched", u64), ("n_records", u64), ("map_seconds", C.c_double), ("ctr", GmoCounters)]


class OracleLib:
    def __init__(self):
        self.lib = L = C.CDLL(ORACLE_SO)
        L.gmo_index_load.argtypes = [C.c_char_p]; L.gmo_index_load.restype = C.POINTER(GmoIndex)
        L.gmo_index_free.argtypes = [C.POINTER(GmoIndex)]
        L.gmo_occ.argtypes = [C.POINTER(GmoIndex), u64, C.c_int, C.c_void_p]; L.gmo_occ.restype = u64
        L.gmo_sa_interval.argtypes = [C.POINTER(GmoIndex), C.c_char_p, C.c_int, C.POINTER(u64), C.POINTER(u64), C.c_void_p]
        L.gmo_locate.argtypes = [C.POINTER(GmoIndex), u64, C.c_void_p]; L.gmo_locate.restype = u64
        L.gmo_window.argtypes = [C.POINTER(GmoIndex), u64, C.c_uint32, C.c_char_p]
        L.gmo_params_default.argtypes = [C.POINTER(GmoParams)]
        L.gmo_params_finalize.argtypes = [C.POINTER(GmoParams)]
        L.gmo_pwm_from_fastq.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), fptr]
        L.gmo_self_score.argtypes = [C.POINTER(GmoParams), fptr, C.c_char_p, C.c_int]; L.gmo_self_score.restype = C.c_float
        L.gmo_nw_score.argtypes = [C.POINTER(GmoParams), fptr, C.c_int, C.c_char_p]; L.gmo_nw_score.restype = C.c_float
        L.gmo_align_score_be.argtypes = [C.POINTER(GmoParams), fptr, C.c_int, C.c_char_p, C.c_uint, C.c_uint]
        L.gmo_align_score_be.restype = C.c_float
        L.gmo_traceback.argtypes = [C.POINTER(GmoParams), fptr, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_char_p]
        L.gmo_pair_hmm.argtypes = [fptr, C.c_int, C.c_char_p, C.c_char_p, C.c_int, fptr]
        L.gmo_fix_cigar.argtypes = [C.c_char_p]
        L.gmo_reverse_cigar.argtypes = [C.c_char_p, C.c_char_p]
        L.gmo_revcomp_str.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        L.gmo_map_read.argtypes = [C.POINTER(GmoIndex), C.POINTER(GmoParams), fptr, C.c_char_p, C.c_int, C.POINTER(GmoResult)]
        L.gmo_result_free.argtypes = [C.POINTER(GmoResult)]
        L.gmo_read_output.argtypes = [C.POINTER(GmoIndex), C.POINTER(GmoParams), C.POINTER(GmoResult), fptr, C.c_char_p, C.c_int,
                                      C.POINTER(C.POINTER(GmoSam)), C.POINTER(C.POINTER(GmoDeposit)), C.POINTER(C.c_int), C.c_void_p]
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]
        L.gmo_run.argtypes = [C.POINTER(GmoIndex), C.POINTER(GmoParams), C.c_char_p, C.c_char_p, C.c_int, u64, C.c_char_p, C.POINTER(GmoRunStats)]

    def params(self, **kw):
        p = GmoParams()
        self.lib.gmo_params_default(C.byref(p))
        for k, v in kw.items():
            assert hasattr(p, k), k
            setattr(p, k, v)
        self.lib.gmo_params_finalize(C.byref(p))
        return p

    @staticmethod
    def apply_subst(p, path):
        """-S file on finalized oracle parameters: readPWM (Driver.cpp:768-859) overwrites the lowercase rows a,c,g,t,n unscaled and
        sets gADJUST = 1 (the gap keeps its scaled value)"""
        rows = []
        for line in open(path):
            f = line.split()
            if len(f) == 4 and f[0][0].lower() == "a" and f[1][0].lower() == "c":
                continue
            vals = [float(x) for x in (f[1:] if len(f) == 5 else f)]
            assert len(vals) == 4, line
            rows.append(vals)
        assert len(rows) >= 5
        for ch, vals in zip("acgtn", rows[:5]):
            for b, v in enumerate(vals):
                p.S[ord(ch)][b] = v
        p.adjust = 1.0
        return p

    def index_load(self, fa):
        ix = self.lib.gmo_index_load(fa.encode())
        assert ix, f"cannot load index {fa}"
        return ix

    def sa_interval(self, ix, kmer: bytes):
        s = u64(); e = u64()
        self.lib.gmo_sa_interval(ix, kmer, len(kmer), C.byref(s), C.byref(e), None)
        return s.value, e.value

    def window(self, ix, begin, L):
        buf = C.create_string_buffer(L + 8)
        self.lib.gmo_window(ix, begin, L, buf)
        return buf.value

    def pwm(self, seq: bytes, qual: bytes, illumina=0):
        L = len(seq)
        P = np.zeros((max(L, 1), 4), np.float32); ill = C.c_int(illumina)
        r = self.lib.gmo_pwm_from_fastq(seq, qual, L, C.byref(ill), P)
        assert r == 0
        return P[:L]

    def pair_hmm(self, P, cons: bytes, w: bytes):
        P = np.ascontiguousarray(P, np.float32)
        out = np.zeros((len(w), 5), np.float32)
        self.lib.gmo_pair_hmm(P, len(P), cons, w, len(w), out)
        return out

    def traceback(self, p, P, cons: bytes, w: bytes):
        P = np.ascontiguousarray(P, np.float32)
        L = len(P)
        al = C.create_string_buffer(2 * L + 8); n = C.c_int(); cg = C.create_string_buffer(1024)
        self.lib.gmo_traceback(C.byref(p), P, L, cons, w, al, C.byref(n), cg)
        return al.raw[:n.value], n.value, cg.value

    def map_read(self, ix, p, P, cons: bytes):
        P = np.ascontiguousarray(P, np.float32)
        r = GmoResult()
        self.lib.gmo_map_read(ix, C.byref(p), P, cons, len(cons), C.byref(r))
        hits = []
        for i in range(r.n_hits):
            h = r.hits[i]
            hits.append(dict(key=h.key, seq=h.seq, score=h.score, first_strand=h.first_strand,
                             pos=[(h.pos[j].pos, h.pos[j].strand) for j in range(h.n_pos)]))
        out = dict(status=r.status, self_score=r.self_score, min_score=r.min_score, top_score=r.top_score,
                   denominator=r.denominator, hits=hits,
                   ctr={n: getattr(r.ctr, n) for n, _ in GmoCounters._fields_})
        self.lib.gmo_result_free(C.byref(r))
        return out

    def read_output(self, ix, p, P, cons: bytes):
        """gmo_map_read + gmo_read_output of one read: (status, SAM records, coverage deposits [(pos, span, w, codes|None)])"""
        P = np.ascontiguousarray(P, np.float32)
        r = GmoResult()
        self.lib.gmo_map_read(ix, C.byref(p), P, cons, len(cons), C.byref(r))
        recs, deps = [], []
        if r.status == 0:
            pr = C.POINTER(GmoSam)(); pd = C.POINTER(GmoDeposit)(); nd = C.c_int()
            n = self.lib.gmo_read_output(ix, C.byref(p), C.byref(r), P, cons, len(cons), C.byref(pr), C.byref(pd), C.byref(nd), None)
            for k in range(n):
                q = pr[k]
                recs.append(dict(pos=q.pos, strand=q.strand, contig=q.contig, chr_pos=q.chr_pos, mapq=q.mapq, cigar=bytes(q.cigar),
                                 a_score=q.a_score, post_prob=q.post_prob, sim_matches=q.sim_matches))
            for k in range(nd.value):
                d = pd[k]
                codes = bytes(d.codes[t] for t in range(d.span)) if d.codes else None
                if d.hmm:               # --snp: span x 5 posteriors instead of codes
                    codes = np.ctypeslib.as_array(d.hmm, shape=(d.span, 5)).copy()
                    self.libc.free(d.hmm)
                deps.append((d.pos, d.span, d.w, codes))
                if d.codes:
                    self.libc.free(d.codes)
            if n:
                self.libc.free(pr)
            if nd.value:
                self.libc.free(pd)
        st = r.status
        self.lib.gmo_result_free(C.byref(r))
        return st, recs, deps

    def run(self, ix, p, fastq, out_prefix, threads=1, max_reads=0, cmdline=""):
        st = GmoRunStats()
        rc = self.lib.gmo_run(ix, C.byref(p), fastq.encode(), out_prefix.encode() if out_prefix else None, threads, max_reads,
                              cmdline.encode(), C.byref(st))
        assert rc == 0, rc
        return st
