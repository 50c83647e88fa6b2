"""CPU-side checks of the drop-in boundary: libgnumap_hip.so loads, exports every symbol include/gnumap_hip.h
declares, and refuses to compute without a gfx950 device (no silent CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import gnumap_amd as g
from gnumap_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gnumap_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = g.load_library()
    names = _declared()
    assert len(names) >= 28
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/gnumap_hip.h but not exported"
    assert sorted(api.EXPORTS) == names


def test_struct_layouts_match_numpy_views():
    assert C.sizeof(api.gm_match) == api.MATCH_DTYPE.itemsize
    assert C.sizeof(api.gm_pos) == api.POS_DTYPE.itemsize
    assert C.sizeof(api.gm_sam_rec) == api.SAM_DTYPE.itemsize
    assert api.RAW_HIT_DTYPE.itemsize == 16


def test_params_follow_reference_defaults():
    p = g.Params()
    # inc/const_define.h + a_matrices.c: match .75, transition -.5, transversion -.75, gap -1, jump = mer/2
    assert (p.mer, p.jump, p.min_seed_hits, p.max_matches, p.max_kmer_hits, p.bin_size) == (10, 5, 2, 1000, 0, 8)
    assert (p.match, p.transition, p.transversion, p.gap) == (0.75, -0.5, -0.75, -1.0)
    S = np.ctypeslib.as_array(p.c.S).reshape(256, 4)
    assert S[ord("a")].tolist() == [0.75, -0.75, -0.5, -0.75] and S[ord("T")].tolist() == [-0.75, -0.5, -0.75, 0.75]
    assert S[ord("n")].tolist() == [-0.75] * 4
    b = g.Params(mode=1)            # -b: lowercase row only, bin size forced to 1 (Driver.cpp:1266, 2805-2810)
    Sb = np.ctypeslib.as_array(b.c.S).reshape(256, 4)
    assert Sb[ord("c")][3] == 0.75 and Sb[ord("C")][3] == -0.5 and b.bin_size == 1
    with pytest.raises(g.GnumapError):
        g.Params(min_seed_hits=0)


def test_host_only_index_and_window(syn_fa, golden):
    ix = g.Index(syn_fa, flags=g.GM_INDEX_HOST_ONLY)
    assert ix.info.seq_len == int(golden["seq_len"]) and ix.info.primary == int(golden["primary"])
    assert ix.contigs() == [("chrA", 0), ("chrB", 150000), ("chrC", 250000)]
    for b, L, w in zip(golden["win_begin"], golden["win_len"], golden["win_out"]):
        assert ix.window(int(b), int(L)) == bytes(w)
    with pytest.raises(g.GnumapError, match="host-only|no usable HIP device"):
        g.Batch(ix, 16, 100)


def test_compute_refuses_without_device(syn_fa):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(g.GnumapError, match="no usable HIP device"):
        g.Index(syn_fa)
    exe = os.path.join(ROOT, "gnumap_amd", "bin", "gnumap")
    if os.path.exists(exe):
        r = subprocess.run([exe, "-g", syn_fa, "-o", "/tmp/none", os.path.join(ROOT, "tests", "golden", "syn.fq")], capture_output=True, text=True)
        assert r.returncode != 0 and "no usable HIP device" in r.stderr


def test_product_does_not_touch_the_oracle():
    """the product (gnumap_amd/, include/) must never reference oracle/ — it is test infrastructure"""
    for base in ("gnumap_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                    txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                    assert "gm_oracle" not in txt and "libgnumap_ref" not in txt and "oracle/" not in txt, os.path.join(dirpath, fn)


def test_slice_pool_survives_concurrent_callers():
    """pass_parallel (gm_api.cpp): the fp64 passes of gm_map_batch / gm_output_batch cut a block over a process-wide helper pool from
    several calling threads at once; the completion state must outlive the caller's frame (a helper notifying a condition
    variable on a dead stack).  Many short passes from 6 callers - items visited exactly once, nothing hangs."""
    L = g.load_library()
    L.gm_selftest_pass_parallel.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.gm_selftest_pass_parallel.restype = C.c_int
    assert L.gm_selftest_pass_parallel(4096, 256, 6, 3000) == 0
    assert L.gm_selftest_pass_parallel(100, 256, 2, 10) == 0          # below the grain: the caller alone
    assert L.gm_selftest_pass_parallel(0, 256, 2, 10) == 0
