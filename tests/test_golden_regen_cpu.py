"""CPU, build container only: every committed fixture under tests/golden/ is what the committed generator
produces from the reference TODAY -- `python oracle/gen_golden.py --check` regenerates all FOURTEEN families into a
scratch directory (importing /root/reference, never touching a GPU) and fails on any byte of any array that
differs.  The thirteen light families run in one generator process, `c5_sequence` in a second one beside it.  That family is
the reference's own eight 30-round sessions at C5's small size (about four minutes): by default the second process
regenerates the four sessions of the first torch seed -- every loop once: plain, knn_prop2, multi_reg, pseudo_lr
(SSW_C5_QUICK; the arrays of the other two seeds are then not compared) -- and all eight with SSW_GOLDEN_ALL=1, so the
default suite stays at the light families' six minutes.  The generator processes keep their default thread counts: the
reference's torch-CPU fits change in the last digits with the thread count.
Skipped where the reference is absent (the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
LIGHT = ("tiling", "sliding", "lknn", "scan_topk", "multiscale_query", "labelprop", "rank_loss", "logreg", "multireg", "bench_loop",
         "multiregneg", "contweighted", "multireg_det")
HEAVY = ("c5_sequence",)

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "seesaw")), reason="reference checkout not present")


@pytest.fixture(scope="module")
def regen():
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""  # the generator must not need a GPU
    gen = os.path.join(ROOT, "oracle", "gen_golden.py")
    procs = {"light": subprocess.Popen([sys.executable, gen, "--check"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                       stderr=subprocess.STDOUT, text=True),
             "heavy": subprocess.Popen([sys.executable, gen, "--check", *HEAVY], cwd=ROOT,
                                       env=env if os.environ.get("SSW_GOLDEN_ALL") else dict(env, SSW_C5_QUICK="1"),
                                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)}
    out = {}
    for tag, proc in procs.items():
        try:
            text, _ = proc.communicate(timeout=1500)
        except subprocess.TimeoutExpired:
            proc.kill()
            text, _ = proc.communicate()
            text += "\n[timed out]"
        out[tag] = (proc.returncode, text)
    return out


def test_the_generator_knows_exactly_these_families():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import gen_golden
    finally:
        sys.path.pop(0)
    assert set(gen_golden.FAMILIES) == set(LIGHT) | set(HEAVY) and set(gen_golden.HEAVY) == set(HEAVY)
    have = {f[:-4] for f in os.listdir(os.path.join(ROOT, "tests", "golden")) if f.endswith(".npz")}
    assert set(LIGHT) | set(HEAVY) <= have, sorted((set(LIGHT) | set(HEAVY)) - have)


def test_all_light_golden_families_regenerate_byte_for_byte(regen):
    rc, text = regen["light"]
    tail = "\n".join(text.splitlines()[-40:])
    assert rc == 0, tail
    for family in LIGHT:
        assert f"{family}: reproduced byte for byte" in text, (family, tail)


def test_c5_sequence_regenerates_byte_for_byte(regen):
    """the reference's own 30-round sessions at C5's small size (1 109 images x 13 tiles), all four loops (first torch seed;
    all three seeds of the fitting loops with SSW_GOLDEN_ALL=1)"""
    rc, text = regen["heavy"]
    tail = "\n".join(text.splitlines()[-40:])
    assert rc == 0, tail
    for family in HEAVY:
        assert f"{family}: reproduced byte for byte" in text, (family, tail)
