"""GPU tests of the kernels that are actually hot: the fused round kernels (round_p1/p2/p3 and the
compile-time-blocked weight-streaming chains inside them), compared with the oracle tensor by tensor
through the debug tap (``dsat_session_set_tap``), the stand-alone fused chain (``dsat_op_mlp_ex``), and
the early exit at a MIDDLE round under graph replay.

The single-operator tests of tests/test_gpu_parity.py exercise the per-layer kernels
(kernels_gemm / kernels_sparse / kernels_misc); those are the DSAT_FUSED=0 path.  Here every
intermediate of the fused path -- query, literals, 4*clauses_loss, clause_messages,
variables_loss_all, new_clause_value, variables_grad, variables_loss, new_variables and the two carried
states -- is read out of the running kernels and held to 1e-4 relative against the oracle's trace.
"""
import os

import numpy as np
import pytest

from helpers import draws, make_all, oracle_weights
from oracle import querysat_oracle as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "early_exit_golden.npz")
TAP_RTOL = 1e-4


@pytest.fixture(scope="module")
def rt():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    from diffusionsat_amd import runtime
    return runtime


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _rel_close(got, ref, what, rtol=TAP_RTOL):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert np.all(np.isfinite(got)), what
    scale = max(float(np.abs(ref).max()), 1e-30)
    # 1e-4 relative, with a floor of a tenth of the tensor's scale under entries that cancel to ~0
    # (PairNorm divides by the row's RMS: absolute fp32 noise there is a few 1e-6 of the scale)
    err = np.abs(got - ref) / (np.abs(ref) + 1e-1 * scale)
    assert err.max() < rtol, "%s: max rel err %.3e at %s (ref scale %.3e)" % (
        what, err.max(), np.unravel_index(np.argmax(err), err.shape), scale)


def _instances(kind):
    if kind == "irregular":
        from diffusionsat_amd.batch import make_batch
        from diffusionsat_amd.instances import irregular_cnf
        insts = [irregular_cnf(300, 1500, seed=3, zipf_a=1.1)[:2], irregular_cnf(120, 500, seed=4)[:2]]
        return insts, make_batch(insts), O.create_adj_matrices(insts)
    return make_all(kind)


# the three forms the round runs in: fused tiles with the clause / update MLPs on the bf16 pipe (fp32 products from
# bf16 slices, the default), fused tiles on v_mfma_f32_16x16x4_f32 only, and the per-layer kernels
# ... and "fused_a2": as "fused", with P1's lit_mlp and P3 on 32-row split-product tiles (what batches of more than one
# wave of workgroups run; sessions told to prefer throughput get the first)
VARIANTS = ["fused", "fused_a2", "fused_f32mfma", "per_layer"]


def _session(rt, batch, ws, max_rounds, fused):
    env = {"per_layer": {"DSAT_FUSED": "0"}, "fused_a2": {"DSAT_P1_A2": "1", "DSAT_P3_BM": "32"},
           "fused_f32mfma": {"DSAT_P2_SPLIT": "0", "DSAT_P3_SPLIT": "0", "DSAT_P1_A2": "0", "DSAT_P1_BSPLIT": "0", "DSAT_P1_ASPLIT": "0"}}.get(fused, {})
    if fused is False:
        env = {"DSAT_FUSED": "0"}
    os.environ.update(env)
    try:
        s = rt.Session(rt.Graph(batch), rt.Weights(ws), max_rounds=max_rounds)
    finally:
        for k in env:
            os.environ.pop(k, None)
    if fused == "fused_f32mfma":
        assert s.split_mask() == 0
    if fused == "fused":
        assert s.split_mask() & 203 == 203    # the released layout: both split-product tiles, role B's two chains, role D of P2
    if fused == "fused_a2":
        assert s.split_mask() & 23 == 23      # ... and P3 on 32-row tiles
    return s


# oracle trace key -> (tap key, which rows)
TAP_MAP = [("query", "query"), ("literals", "literals"), ("clauses_loss4", "clauses_loss4"),
           ("clause_messages", "clause_messages"), ("variables_loss_all", "variables_loss_all"),
           ("new_clause_value", "new_clause_value"), ("variables_grad", "variables_grad"),
           ("variables_loss", "variables_loss"), ("new_variables", "new_variables"),
           ("variables", "variables"), ("clause_state", "clause_state")]


@pytest.mark.parametrize("fused", VARIANTS)
@pytest.mark.parametrize("kind", ["3sat", "ragged", "medium", "irregular"])
def test_round_intermediates_match_oracle_trace(rt, kind, fused):
    """Every intermediate tensor of rounds 0, 1 and 31 against the oracle trace.  The step runs 33
    rounds so that the clause state of round 31 (finished while round 32 runs on the fused path) is
    available; rounds are causal, so rounds 0..31 are those of a 32-round step."""
    if fused == "per_layer" and kind in ("medium", "irregular"):
        pytest.skip("per-layer path: two graph kinds are enough (it has its own operator tests)")
    insts, batch, data = _instances(kind)
    ws = oracle_weights(7)
    rounds = 33
    d = draws(batch.n_vars, rounds, seed=23)
    noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, np.float32), d["uniform"])
    ref = O.diffusion_step(ws, data, np.float32(0.4), noisy, d["normal"], d["labels"], rounds, early_exit=False,
                           dt=np.float32, trace=True)["trace"]
    s = _session(rt, batch, ws, rounds, fused)
    for r in (0, 1, 31):
        out = s.diffusion_step(0.4, _dev(noisy), _dev(d["normal"]), _dev(d["labels"]), rounds=rounds,
                               early_exit=False, tap_round=r)
        torch.cuda.synchronize()
        tap = out["tap"]
        for okey, tkey in TAP_MAP:
            # rounds 0 and 1 isolate one round's kernels (1e-4); by round 31 the two fp32 recurrences
            # (different summation orders) have drifted apart by up to ~1.5e-4 of a tensor's scale
            _rel_close(tap[tkey].cpu().numpy(), ref[okey][r], "%s round %d (%s)" % (tkey, r, kind),
                       TAP_RTOL if r < 2 else 5 * TAP_RTOL)
    # and a tapped run leaves no trace: the next untapped step is the ordinary one
    a = s.diffusion_step(0.4, _dev(noisy), _dev(d["normal"]), _dev(d["labels"]), rounds=rounds, early_exit=False,
                         want_logits=True)
    assert a["tap"] is None
    assert np.abs(a["logits"].cpu().numpy() - ref["logits"][-1]).max() < 2e-4


def test_tap_rejects_bad_rounds(rt):
    from diffusionsat_amd import _lib
    _, batch, _ = make_all("3sat")
    s = _session(rt, batch, oracle_weights(0), 4, True)
    noisy = _dev(np.tile(np.array([[1, 0]], np.float32), (batch.n_vars, 1)))
    with pytest.raises(_lib.DsatError):
        s.diffusion_step(0.5, noisy, rounds=4, early_exit=False, tap_round=4)      # beyond the step
    out = s.diffusion_step(0.5, noisy, rounds=4, early_exit=False, tap_round=3)    # last round: no clause_state
    assert "clause_state" not in out["tap"] and out["tap"]["variables"].abs().sum().item() > 0


@pytest.mark.parametrize("mlp", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("rb,nwaves", [(1, 4), (1, 8), (2, 4), (2, 8), (4, 8)])
def test_op_mlp_ex_fused_chain(rt, mlp, rb, nwaves):
    """``dsat_op_mlp_ex(impl=1)``: the row-tile fused chain (kernels_chain.hip, the generic per-layer
    weight-streaming path of fused_common.h) for every MLP and every tile shape it is built for."""
    _, batch, _ = make_all("medium")
    ws = oracle_weights(3)
    s = _session(rt, batch, ws, 2, True)
    name = O.MLP_ORDER[mlp]
    rows = batch.n_clauses if mlp == 2 else batch.n_vars
    rng = np.random.default_rng(10 + mlp)
    x = rng.standard_normal((rows, ws[name][0][0].shape[0])).astype(np.float32)
    from diffusionsat_amd import _lib
    try:
        y = s.op_mlp(mlp, _dev(x), impl=1, rb=rb, nwaves=nwaves).cpu().numpy()
    except _lib.DsatError:
        # 64-row tiles of the two widest MLPs (lit_mlp, update_gate) exceed the 160 KiB of LDS: the
        # library refuses the launch instead of running it
        assert rb == 4 and mlp in (1, 3)
        return
    ref = O.mlp(x.astype(np.float64), [(w.astype(np.float64), b.astype(np.float64)) for w, b in ws[name]])
    assert np.all(np.isfinite(y))
    assert np.abs(y - ref).max() < 1e-5 + 1e-4 * np.abs(ref).max()
    y0 = s.op_mlp(mlp, _dev(x), impl=0).cpu().numpy()                 # the per-layer GEMM path agrees
    assert np.abs(y - y0).max() < 1e-5 + 1e-4 * np.abs(ref).max()


@pytest.mark.parametrize("fused", VARIANTS)
@pytest.mark.parametrize("case", [0, 1])
def test_mid_loop_early_exit_matches_fixture(rt, case, fused):
    """is_batch_sat fires at a middle round (model/query_sat.py:330-338): ``steps_taken``, the logits of
    THAT round, the selected prediction, the argmin map and the rounded assignment equal the committed
    oracle fixture (tests/golden/make_early_exit_golden.py), eager and under graph replay.  On the fused
    path select/SAT of round r rides in the clause kernel of round r+1, so kernels of round r+1 have run
    when the flag lands; nothing they wrote may reach the outputs."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_early_exit_golden import unflatten
    from diffusionsat_amd.batch import make_batch
    g = np.load(GOLD, allow_pickle=False)
    p = "case%d_" % case
    insts = unflatten(g[p + "n_vars"], g[p + "n_clauses"], g[p + "rowptr"], g[p + "lits"])
    batch = make_batch(insts)
    rounds = int(g["rounds"])
    want_steps = int(g[p + "steps_taken"])
    assert 0 < want_steps < rounds - 1
    ws = oracle_weights(int(g["weight_seed"]))
    # the fixture is what the oracle computes today (guards against a stale file)
    ref = O.diffusion_step(ws, O.create_adj_matrices(insts), np.float32(g["noise_scale"]), g[p + "noisy"],
                           g[p + "normal"], g[p + "labels"], rounds, early_exit=True, dt=np.float32)
    assert ref["steps_taken"] == want_steps and np.array_equal(ref["logits"], g[p + "logits"])
    s = _session(rt, batch, ws, rounds, fused)
    for graph_replay in (False, True, True):          # the second replay reuses the captured graph
        s.set_graph_replay(graph_replay)
        out = s.diffusion_step(float(g["noise_scale"]), _dev(g[p + "noisy"]), _dev(g[p + "normal"]),
                               _dev(g[p + "labels"]), rounds=rounds, early_exit=True, want_logits=True)
        torch.cuda.synchronize()
        assert int(out["steps_taken"].item()) == want_steps
        assert np.abs(out["logits"].cpu().numpy() - g[p + "logits"]).max() < 2e-4
        pred = out["prediction"].cpu().numpy()
        assert np.abs(pred - g[p + "prediction"]).max() < 2e-4
        assert np.array_equal(out["best_map"].cpu().numpy(), g[p + "best_map"])
        assert np.array_equal(np.round(O.sigmoid(pred)).astype(np.uint8), g[p + "assignment"])
    # with the exit disarmed the same inputs run all rounds and end elsewhere
    out = s.diffusion_step(float(g["noise_scale"]), _dev(g[p + "noisy"]), _dev(g[p + "normal"]),
                           _dev(g[p + "labels"]), rounds=rounds, early_exit=False, want_logits=True)
    assert int(out["steps_taken"].item()) == rounds - 1
    assert np.abs(out["logits"].cpu().numpy() - g[p + "logits"]).max() > 1e-3
