"""GPU tests at BASELINE.json's full sizes through size-independent properties, plus the
committed fixture and the Python facade (QuerySAT / DiffusionSampler surface)."""
import os

import numpy as np
import pytest

from helpers import draws, make_all, oracle_weights
from oracle import querysat_oracle as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_golden.npz")


@pytest.fixture(scope="module")
def rt():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    from diffusionsat_amd import runtime
    return runtime


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_committed_fixture(rt):
    from diffusionsat_amd.batch import make_batch
    g = np.load(GOLD, allow_pickle=False)
    for kind in ("3sat", "ragged"):
        insts, batch, _ = make_all(kind)
        rounds = g[kind + "_normal"].shape[0]
        s = rt.Session(rt.Graph(batch), rt.Weights(oracle_weights(5)), max_rounds=rounds)
        s.set_graph_replay(False)
        out = s.diffusion_step(float(g[kind + "_noise_scale"]), _dev(g[kind + "_noisy"]), _dev(g[kind + "_normal"]),
                               _dev(g[kind + "_labels"]), rounds=rounds, early_exit=False, want_round_logits=True)
        got = out["round_logits"].cpu().numpy()
        assert np.abs(got - g[kind + "_round_logits_f64"]).max() < 2e-4
        assert np.array_equal(out["best_map"].cpu().numpy(), g[kind + "_best_map"])


def test_config2_replicas_agree_and_match_a_single_graph(rt):
    """Config-2 size (31 x n=100, 19 468 nodes): 31 copies of ONE formula with identical noise must
    give 31 identical answers (block-diagonal batching), equal to the formula run alone, and the
    single graph is small enough for the oracle."""
    from diffusionsat_amd.batch import make_batch
    from diffusionsat_amd.instances import hard_3sat
    n, clauses, _ = hard_3sat(100, seed=2000)
    G, rounds = 31, 32
    ws = oracle_weights(8)
    d1 = draws(n, rounds, seed=5)
    rep = lambda a, axis: np.concatenate([a] * G, axis=axis)
    noisy1 = O.randomized_rounding(np.full((n, 2), 0.5, np.float32), d1["uniform"])
    big = make_batch([(n, clauses)] * G)
    assert big.n_nodes == 19468
    sb = rt.Session(rt.Graph(big), rt.Weights(ws), max_rounds=rounds)
    ob = sb.diffusion_step(0.5, _dev(rep(noisy1, 0)), _dev(rep(d1["normal"], 1)), _dev(rep(d1["labels"], 0)),
                           rounds=rounds, early_exit=False, want_logits=True)
    lb = ob["logits"].cpu().numpy().reshape(G, n, 8)
    assert np.abs(lb - lb[0]).max() < 2e-5          # tiles straddle graphs differently: not bitwise
    s1 = rt.Session(rt.Graph(make_batch([(n, clauses)])), rt.Weights(ws), max_rounds=rounds)
    o1 = s1.diffusion_step(0.5, _dev(noisy1), _dev(d1["normal"]), _dev(d1["labels"]), rounds=rounds,
                           early_exit=False, want_logits=True)
    assert np.abs(o1["logits"].cpu().numpy() - lb[0]).max() < 2e-5
    ref = O.diffusion_step(ws, O.create_adj_matrices([(n, clauses)]), np.float32(0.5), noisy1, d1["normal"],
                           d1["labels"], rounds, early_exit=False, dt=np.float32)
    assert np.abs(lb[0] - ref["logits"]).max() < 2e-4


def test_config2_is_deterministic_and_fused_equals_per_layer_path(rt, monkeypatch):
    """Run-to-run bitwise reproducibility (no float atomics anywhere), and the fused round
    kernels against the one-launch-per-layer path on the full config-2 batch."""
    from diffusionsat_amd.batch import make_batch
    from diffusionsat_amd.instances import config2_instances
    batch = make_batch(config2_instances(31))
    ws = oracle_weights(3)
    rounds = 32
    d = draws(batch.n_vars, rounds, seed=2)
    noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, np.float32), d["uniform"])
    args = (0.25, _dev(noisy), _dev(d["normal"]), _dev(d["labels"]))
    s = rt.Session(rt.Graph(batch), rt.Weights(ws), max_rounds=rounds)
    a = s.diffusion_step(*args, rounds=rounds, early_exit=True, want_logits=True)
    b = s.diffusion_step(*args, rounds=rounds, early_exit=True, want_logits=True)
    assert torch.equal(a["logits"], b["logits"]) and torch.equal(a["prediction"], b["prediction"])
    s.set_graph_replay(False)
    c = s.diffusion_step(*args, rounds=rounds, early_exit=True, want_logits=True)
    assert torch.equal(a["logits"], c["logits"])            # graph replay == eager
    monkeypatch.setenv("DSAT_FUSED", "0")
    s0 = rt.Session(rt.Graph(batch), rt.Weights(ws), max_rounds=rounds)
    e = s0.diffusion_step(*args, rounds=rounds, early_exit=True, want_logits=True)
    assert (a["logits"] - e["logits"]).abs().max().item() < 1e-4
    assert int(a["steps_taken"].item()) == int(e["steps_taken"].item()) == rounds - 1


def test_config3_size_sampler_properties(rt):
    """Config-3 size (256 copies of one n=100 formula = 160 768 nodes), device RNG: outputs are
    {0,1}, frozen solutions really satisfy their formula, solved flags agree with a host re-check."""
    from diffusionsat_amd.batch import make_batch
    from diffusionsat_amd.dimacs import VariableAssignment
    from diffusionsat_amd.instances import hard_3sat
    n, clauses, _ = hard_3sat(100, seed=2000)
    G = 256
    batch = make_batch([(n, clauses)] * G)
    assert batch.n_nodes == 160768
    s = rt.Session(rt.Graph(batch), rt.Weights(oracle_weights(1)), max_rounds=4)
    out = s.sample(3, 4, seed=7, early_exit=True)
    asg = out["assignment"].cpu().numpy()
    solved = out["solved"].cpu().numpy()
    assert set(np.unique(asg)) <= {0, 1} and asg.shape == (G * n,)
    va = VariableAssignment(n_vars=n, clauses=clauses)
    for g in range(G):
        va.assign_all_from_bit_list(asg[g * n:(g + 1) * n])
        if solved[g]:
            assert va.satisfiable()
    assert abs(out["solved_fraction"] - solved.mean()) < 1e-6
    again = s.sample(3, 4, seed=7, early_exit=True)
    assert torch.equal(again["assignment"], out["assignment"])


def test_facade_query_sat_and_sampler(rt, tmp_path):
    """The reference surface: QuerySAT(...).diffusion_step(...) -> {"steps_taken","loss","prediction"},
    MLP / PairNorm layer calls, and DiffusionSampler(model_path, dimacs).samples(n) -> dict[int,int]."""
    from diffusionsat_amd import weights as W
    from diffusionsat_amd.diffusion_sampler import DiffusionSampler
    from diffusionsat_amd.dimacs import DimacsFile, VariableAssignment
    from diffusionsat_amd.query_sat import GraphInputs, QuerySAT
    insts, batch, data = make_all("3sat")
    ws = oracle_weights(5)
    model = QuerySAT(optimizer=None, test_rounds=8)
    model.set_weights(ws)
    gi = GraphInputs(batch)
    d = draws(batch.n_vars, 8, seed=11)
    noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, np.float32), d["uniform"])
    out = model.diffusion_step(gi, None, None, None, 0.5, _dev(noisy), normal=_dev(d["normal"]),
                               labels=_dev(d["labels"]))
    assert set(out) == {"steps_taken", "loss", "prediction"}
    ref = O.diffusion_step(ws, data, np.float32(0.5), noisy, d["normal"], d["labels"], 8, dt=np.float32, trace=True)
    assert out["prediction"].shape == (batch.n_vars,)
    assert int(out["steps_taken"].item()) == ref["steps_taken"]
    # the prediction is the selected logit map: compare on graphs whose argmin margin is clear
    srt = np.sort(ref["trace"]["per_graph_loss"][-1], axis=1)
    clear = ((srt[:, 1] - srt[:, 0]) > 1e-3)[data["variables_mask"]]
    assert clear.any()
    assert np.abs(out["prediction"].cpu().numpy() - ref["prediction"])[clear].max() < 2e-4
    x = np.random.default_rng(0).standard_normal((batch.n_vars, 128)).astype(np.float32)
    y = model.variables_output(_dev(x)).cpu().numpy()                   # MLP.call(inputs), model/mlp.py:43
    assert np.abs(y - O.mlp(x, ws["variables_output"])).max() < 1e-4
    vg = data["variables_graph_adj"]
    yn = model.variables_norm(_dev(x), vg.multiply(1.0 / vg.sum(axis=1))).cpu().numpy()   # PairNorm.call(inputs, graph)
    assert abs(float((yn ** 2).mean()) - 1.0) < 1e-3
    # sampler: a formula every assignment of which is a model -> n samples after one batch
    cnf = tmp_path / "taut.cnf"
    DimacsFile(str(cnf), clauses=[[1, -1], [2, -2, 3], [-3, 3], [4, -4]]).store()
    npz = tmp_path / "w.npz"
    W.save_npz(str(npz), ws)
    smp = DiffusionSampler(str(npz), str(cnf))
    assert smp.restored
    res = smp.samples(25)
    assert sum(res.values()) == 25 and all(isinstance(k, int) and 0 <= k < 16 for k in res)
    assert smp.success_rate == 1.0
    # the device's per-copy SAT flags give the same dictionary as the reference's host re-check
    smp_h = DiffusionSampler(str(npz), str(cnf))
    smp_h.host_recheck = True
    assert smp_h.samples(25) == res
    # a formula with few models (x1 & (x2|x3) & -x4): some copies fail, both decoders agree
    few = tmp_path / "few.cnf"
    DimacsFile(str(few), clauses=[[1], [2, 3], [-4]]).store()
    a, b = DiffusionSampler(str(npz), str(few)), DiffusionSampler(str(npz), str(few))
    b.host_recheck = True
    ra, rb = a.samples(40), b.samples(40)
    assert ra == rb and a.success_rate == b.success_rate
    c = DiffusionSampler(str(npz), str(few))
    c.batches_in_flight = 1                                  # sequential == pipelined
    assert c.samples(40) == ra and c.success_rate == a.success_rate
    assert all(k & 1 and not (k >> 3) & 1 and (k >> 1) & 3 for k in ra)
    # an unsatisfiable formula stops by the 0.5 % rule and returns an empty dict
    bad = tmp_path / "unsat.cnf"
    DimacsFile(str(bad), clauses=[[1], [-1], [2, 1]]).store()
    smp2 = DiffusionSampler(None, str(bad))
    assert smp2.samples(3) == {} and smp2.success_rate == 0.0


def test_irregular_degree_graph_matches_oracle(rt):
    """Config-5-like stress: clause lengths 2..8, power-law literal degrees (hundreds of clauses on
    the busiest literals, far beyond the LDS-staged CSR capacity of a 16-variable tile, so the
    un-staged gather path runs too).  3 rounds against the oracle."""
    from diffusionsat_amd.batch import make_batch
    from diffusionsat_amd.instances import irregular_cnf
    insts = [irregular_cnf(1500, 9000, seed=5, zipf_a=1.1)[:2], irregular_cnf(700, 3000, seed=6)[:2]]
    batch = make_batch(insts)
    data = O.create_adj_matrices(insts)
    deg = np.asarray(data["adjacency_matrix"].sum(axis=1)).ravel()
    assert deg.max() > 600                       # really irregular
    ws = oracle_weights(4)
    rounds = 3
    d = draws(batch.n_vars, rounds, seed=8)
    noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, np.float32), d["uniform"])
    ref = O.diffusion_step(ws, data, np.float32(0.3), noisy, d["normal"], d["labels"], rounds, early_exit=False,
                           dt=np.float32, trace=True)
    s = rt.Session(rt.Graph(batch), rt.Weights(ws), max_rounds=rounds)
    out = s.diffusion_step(0.3, _dev(noisy), _dev(d["normal"]), _dev(d["labels"]), rounds=rounds, early_exit=False,
                           want_logits=True, want_round_logits=True)
    got = out["round_logits"].cpu().numpy()
    for r in range(rounds):
        assert np.abs(got[r] - ref["trace"]["logits"][r]).max() < 5e-4, r


def test_two_batches_in_flight_do_not_disturb_each_other(rt):
    """Two sessions over the same device graph and weights, on two HIP streams at once (what
    samples() and bench.py --batches-in-flight do): each reproduces its solo run bit for bit."""
    from diffusionsat_amd.batch import make_batch
    from diffusionsat_amd.instances import config2_instances
    batch = make_batch(config2_instances(31))
    g, w = rt.Graph(batch), rt.Weights(oracle_weights(2))
    T, R = 3, 32
    solo = []
    for seed in (11, 12):
        s = rt.Session(g, w, max_rounds=R)
        out = s.sample(T, R, seed=seed, early_exit=True, want_predictions=True, want_solved_fraction=False)
        torch.cuda.synchronize()
        solo.append((out["assignment"].clone(), out["predictions"].clone()))
    assert not torch.equal(solo[0][1], solo[1][1])              # different seeds, different trajectories
    sess = [rt.Session(g, w, max_rounds=R) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(2):                                         # rep 0 captures the graphs, rep 1 replays them
        outs = []
        for s, st, seed in zip(sess, streams, (11, 12)):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs.append(s.sample(T, R, seed=seed, early_exit=True, want_predictions=True,
                                     want_solved_fraction=False))
        torch.cuda.synchronize()
        for o, (a, p) in zip(outs, solo):
            assert torch.equal(o["assignment"], a) and torch.equal(o["predictions"], p)
