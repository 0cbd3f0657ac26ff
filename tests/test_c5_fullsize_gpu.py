"""GPU: BASELINE config C5 at its FULL size -- 120 000 images x 13 tiles = 1.56 M vectors, batch 1, shortlist 50 -- as a
`-m gpu` test (VERDICT r3 configs_untested: this size was checked only by bench.py's sequence_check).

No fixture of the reference exists at this size (its own session takes ~0.5 s a round here and its k-NN graph comes from
pynndescent, which this image lacks), so the check is the one the task statement prescribes for full sizes: the HIP
sessions against the CPU oracle (oracle/cpu_loop.py: the reference's numpy / scipy / torch-CPU expressions, pinned at the
small sizes by tests/golden/c5_sequence.npz and bench_loop.npz) on the same dataset, the same exact k-NN graph and the same
numpy / torch seeds -- the image returned in every round must be the same, over all 30 rounds of the reference's standard
session (scripts/configs/std_bench.yaml:6-15) for every loop (round 4 stopped the graph loops at 4
rounds; the CPU legs cost 0.25-0.5 s a round here, so 30 rounds of all four are about a minute)."""
import contextlib
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MATRIX = dict(knn_path="nndescent60", symmetric=True, self_edges=False, normalized_weights=False, knn_k=10, edist=0.05)
LP = dict(matrix_options=MATRIX, normalize_scores=False, sigmoid_before_propagate=True, calib_a=10.0, calib_b=-0.4, prior_weight=1.0)
OPTIONS = {
    "plain": None,
    "multi_reg": dict(label_loss_type="ce_loss", rank_loss_margin=0.2, use_qvec_norm=None, reg_data_lambda=0.0, reg_norm_lambda=100.0,
                      reg_query_lambda=0.0, verbose=False, max_iter=200, pos_weight="balanced", lr=1.0, matrix_options=MATRIX),
    "knn_prop2": LP,
    "pseudo_lr": dict(switch_over=True, real_sample_weight=1.0, sample_size=10000, label_prop_params=LP,
                      log_reg_params=dict(class_weights=1.0, scale="centered", reg_lambda=1.0, max_iter=200.0, lr=1, fit_intercept=False)),
}
ROUNDS = {"plain": 30, "multi_reg": 30, "knn_prop2": 30, "pseudo_lr": 30}


@pytest.fixture(scope="module")
def full():
    import torch
    from seesaw_amd.synthetic import GlobalDataManager, make_dataset
    free, _ = torch.cuda.mem_get_info(0)
    if free < (40 << 30):
        pytest.skip("needs 40 GB of device memory")
    ds = make_dataset("lvis", n_images=120000, tiles_per_image=13, n_categories=2, positive_frac=0.05, seed=11, knn_k=10, device=0)
    ds.embedding.noise = 1.2
    with contextlib.redirect_stdout(io.StringIO()):
        ds.knn_graph()  # the exact 10-NN graph of 1.56 M vectors, built by ssw_knn_build
    yield GlobalDataManager().add(ds), ds
    ds.load_index()._dev.close()


@pytest.mark.parametrize("name", ["plain", "multi_reg", "knn_prop2", "pseudo_lr"])
def test_full_size_session_equals_the_cpu_oracle(full, name):
    import torch
    from oracle import cpu_loop
    from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.seesaw_bench import benchmark_loop
    from seesaw_amd.seesaw_session import make_session
    gdm, ds = full
    assert ds.vectors.shape == (1_560_000, 512)
    rounds = ROUNDS[name]
    boxes, _ = ds.load_ground_truth()
    p = SessionParams(index_spec=IndexSpec(d_name="lvis", i_name="multiscale"), interactive=name, interactive_options=OPTIONS[name],
                      batch_size=1, shortlist_size=50, agg_method="plain_score", aug_larger="greater",
                      start_policy="after_first_batch", index_options={"use_vec_index": False})
    b = BenchParams(name=name, ground_truth_category="c1", qstr="a c1", n_batches=rounds, max_results=10 ** 6)
    with contextlib.redirect_stdout(io.StringIO()):
        ret = make_session(gdm, p, b=b)
        np.random.seed(0)
        torch.manual_seed(0)
        g = benchmark_loop(session=ret["session"], box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
        hip = [int(v) for a in ret["session"].acc_indices for v in np.asarray(a).reshape(-1)]
        qvec = ds.load_index().string2vec("a c1")
        np.random.seed(0)
        torch.manual_seed(0)
        c = cpu_loop.run_session(ds.vectors, ds.vector_meta, boxes, "c1", qvec, loop=name, n_batches=rounds, max_results=10 ** 6,
                                 knn_df=ds.knn_graph().restrict_k(k=10).knn_df if name in ("knn_prop2", "pseudo_lr") else None)
    cpu = [int(v) for v in c["shown"]]
    assert len(hip) == len(cpu) == rounds
    assert hip == cpu, (name, hip, cpu)
    assert g["nfound"] == c["nfound"]
    print(f"C5 full size, {name}: {rounds} rounds identical; HIP {1e3 * float(np.mean(g['latencies'])):.2f} ms / round, "
          f"CPU oracle {1e3 * float(np.mean(c['latencies'])):.0f} ms / round")


def test_graph_loop_rounds_are_fused_incremental_updates(full, monkeypatch):
    """at the benchmark's size a knn_prop2 session must actually TAKE the fast path -- of its 29
    propagating rounds at least 27 are incremental updates, each of them one C-ABI call (ssw_labelprop_round) with one host
    wait -- and the session it shows is the one the three-call path (SSW_NO_FUSED_ROUND=1) shows."""
    import torch
    from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.label_propagation import LabelPropagation
    from seesaw_amd.seesaw_bench import benchmark_loop
    from seesaw_amd.seesaw_session import make_session
    gdm, ds = full
    boxes, _ = ds.load_ground_truth()
    p = SessionParams(index_spec=IndexSpec(d_name="lvis", i_name="multiscale"), interactive="knn_prop2", interactive_options=OPTIONS["knn_prop2"],
                      batch_size=1, shortlist_size=50, agg_method="plain_score", aug_larger="greater",
                      start_policy="after_first_batch", index_options={"use_vec_index": False})
    b = BenchParams(name="knn_prop2", ground_truth_category="c1", qstr="a c1", n_batches=30, max_results=10 ** 6)
    kinds = []
    for entry in ("round", "fit_resident"):
        orig = getattr(LabelPropagation, entry)

        def recorded(self, *a, _orig=orig, _entry=entry, **k):
            out = _orig(self, *a, **k)
            self._read_run_info()
            kinds.append((_entry, self.last_mode, self.last_host_syncs))
            return out
        monkeypatch.setattr(LabelPropagation, entry, recorded)
    shown = {}
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("SSW_NO_FUSED_ROUND", "1")
        kinds.clear()
        with contextlib.redirect_stdout(io.StringIO()):
            ret = make_session(gdm, p, b=b)
            np.random.seed(0)
            torch.manual_seed(0)
            benchmark_loop(session=ret["session"], box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
        shown[fused] = [int(v) for a in ret["session"].acc_indices for v in np.asarray(a).reshape(-1)]
        if fused:
            assert len(kinds) == 29 and all(e == "round" for e, _, _ in kinds), kinds
            propagating = [(m, s) for _, m, s in kinds if m != 3]
            assert sum(m == 1 for m, _ in propagating) >= len(propagating) - 2 >= 25, kinds
            assert all(s == 1 for m, s in propagating if m == 1), kinds
        else:
            assert all(e == "fit_resident" for e, _, _ in kinds) and len(kinds) >= 27, kinds
    assert shown[True] == shown[False] and len(shown[True]) == 30
