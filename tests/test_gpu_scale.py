"""Maximum sizes: a reference with more than 2^31 positions (2.3 Gbp synthetic, 7 contigs).  The index is built on the box
(suffix-array stage on the device), loaded into HBM, and reads drawn mostly from beyond coordinate 2^31 are compared read by read
with the oracle; exact reads must recover their origin.  tools/scale_check.py is the same check as a command-line tool."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_reference_beyond_2_31_positions(tmp_path):
    import scale_check
    out = scale_check.main(["--mbp", "2300", "--contigs", "7", "--mer", "14", "--reads", "200000", "--sample", "48", "--steps", "1",
                            "--workdir", str(tmp_path)])
    assert out["l_pac"] == 2_300_000_000 and out["l_pac"] > 2 ** 31
    assert out["oracle_sample"] == 48 and out["oracle_mismatches"] == 0
    assert out["oracle_tail_reads"] >= 10                     # sampled reads that lie beyond 2^31
    assert out["exact_reads_checked"] > 1000 and out["exact_reads_origin_found"] == out["exact_reads_checked"]
    assert out["max_reported_pos"] > 2 ** 31
    assert out["vote_retries"] == 0
