"""CPU, build container only: every committed fixture under tests/golden/ is what the committed generator
produces from the reference TODAY -- `python oracle/gen_golden.py --check` regenerates all eight families into a
scratch directory (importing /root/reference, never touching a GPU) and fails on any byte of any array that
differs.  Skipped where the reference is absent (the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "seesaw")), reason="reference checkout not present")
def test_all_golden_families_regenerate_byte_for_byte():
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""  # the generator must not need a GPU
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), "--check"], cwd=ROOT,
                          env=env, capture_output=True, text=True, timeout=900)
    tail = "\n".join((proc.stdout + proc.stderr).splitlines()[-40:])
    assert proc.returncode == 0, tail
    for family in ("scan_topk", "multiscale_query", "labelprop", "rank_loss", "logreg", "multireg", "bench_loop", "lknn"):
        assert f"{family}: reproduced byte for byte" in proc.stdout, tail
