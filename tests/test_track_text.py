"""The track writers (gm_coverage_write_sgr / _gmp: GenomeBwt::PrintFinalSGR / PrintFinalBisulfite, src/GenomeBwt.cpp:1092-1273) format
their numbers without printf; the text must be what printf("%.5f") / ("%f") gives for every float, ties and thresholds included.  CPU
only: a host-only index gives the contig geometry, the bins are made up."""
import ctypes as C
import os

import numpy as np

import gnumap_amd as g
from gnumap_amd import api


def _values(n, rng):
    v = rng.random(n).astype(np.float32) * np.float32(40.0)
    v[::7] = (rng.integers(0, 1 << 20, len(v[::7])) / np.float32(1 << 14)).astype(np.float32)        # k / 2^14: many exact decimal ties at 5 and 6 places
    v[::11] = np.float32(0.001)                                                                        # the print threshold itself (not above it)
    v[1::11] = np.nextafter(np.float32(0.001), np.float32(1))
    v[2::11] = np.float32(0.0)
    v[3::11] = np.float32(0.015625); v[4::11] = np.float32(2.5e-6); v[5::11] = np.float32(123456.789); v[6::11] = np.float32(8.0)
    v[7::11] = np.float32(0.000015); v[8::11] = np.float32(99999.995)
    return v


def test_sgr_and_gmp_text_equal_printf(syn_fa, tmp_path):
    h = C.c_void_p()
    assert g.lib().gm_index_open(os.fsencode(syn_fa), 0, api.GM_INDEX_HOST_ONLY, C.byref(h)) == 0
    L = g.lib()
    L.gm_coverage_write_sgr.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    L.gm_coverage_write_gmp.argtypes = [C.c_void_p, C.POINTER(api.gm_params), C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    L.gm_coverage_bins.restype = C.c_uint64; L.gm_coverage_bins.argtypes = [C.c_void_p]
    contigs = [(L.gm_index_contig_name(h, i).decode(), L.gm_index_contig_offset(h, i)) for i in range(3)]
    l_pac = L.gm_index_contig_offset(h, 3)
    rng = np.random.default_rng(3)
    for bs in (8, 1):
        assert L.gm_coverage_reset(h, bs) == 0
        nb = L.gm_coverage_bins(h)
        bins = _values(nb, rng)
        out = str(tmp_path / f"t{bs}.sgr")
        assert L.gm_coverage_write_sgr(h, bins.ctypes.data, out.encode(), 0) == 0
        want = []
        for k in range((l_pac + bs - 1) // bs):
            count = k * bs
            i = max(j for j, (_, off) in enumerate(contigs) if off <= count)
            if float(bins[k]) > 0.001:
                want.append("%s\t%d\t%.5f\n" % (contigs[i][0], count - contigs[i][1] + 1, float(bins[k])))
        got = open(out).read()
        assert got == "".join(want) and len(want) > 1000
        # appending writes behind what is there
        assert L.gm_coverage_write_sgr(h, bins.ctypes.data, out.encode(), 1) == 0
        assert open(out).read() == 2 * "".join(want)
        # several slices written concurrently (slices of 4096 bins), new file and append: the text does not depend on the slicing
        assert L.gm_set_option(b"GM_TRACK_SLICE", b"4096") == 0
        try:
            assert L.gm_coverage_write_sgr(h, bins.ctypes.data, out.encode(), 0) == 0
            assert open(out).read() == "".join(want)
            assert L.gm_coverage_write_sgr(h, bins.ctypes.data, out.encode(), 1) == 0
            assert L.gm_coverage_write_sgr(h, bins.ctypes.data, out.encode(), 1) == 0
            assert open(out).read() == 3 * "".join(want)
        finally:
            assert L.gm_set_option(b"GM_TRACK_SLICE", None) == 0
    # .gmp: bin size 1, only the positions whose reference base is 'c' (-b), "%f" for the total, "%.5f" for the five tracks
    p = g.Params(mode=1)
    assert L.gm_coverage_reset(h, 1) == 0
    nb = L.gm_coverage_bins(h)
    bins = _values(nb, rng); nuc = _values(5 * nb, rng)
    out = str(tmp_path / "t.gmp")
    assert L.gm_coverage_write_gmp(h, C.byref(p.c), bins.ctypes.data, nuc.ctypes.data, out.encode(), 0) == 0
    pac = np.fromfile(syn_fa + ".gnumap.pac", np.uint8)
    want = []
    for k in range(l_pac):
        base = (pac[k >> 2] >> ((~k & 3) << 1)) & 3
        if base != 1 or not float(bins[k]) > 0.0:
            continue
        i = max(j for j, (_, off) in enumerate(contigs) if off <= k)
        want.append("%s\t%d\t%f\t%.5f\t%.5f\t%.5f\t%.5f\t%.5f\n" % ((contigs[i][0], k - contigs[i][1] + 1, float(bins[k])) + tuple(float(nuc[q * nb + k]) for q in range(5))))
    assert open(out).read() == "".join(want) and len(want) > 10000
    L.gm_index_close(h)
