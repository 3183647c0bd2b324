"""GPU parity tests: HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs.

Tolerances: north_star asks 1e-3 abs on the per-round literal logits in fp32; the kernels
accumulate in fp32 in a different order than NumPy/TF, and the fp32-vs-fp64 oracle gap
is ~5e-6 (tests/test_oracle.py), so these tests hold the HIP path to 2e-4 abs on logits
(logit magnitudes are O(0.1-1)) and 1e-4 relative on single operators.
Discrete outputs (argmin map, rounded bits) are compared only where the oracle's
pre-threshold margin exceeds the tolerance.
"""
import os
import numpy as np
import pytest

from helpers import draws, make_all, oracle_weights
from oracle import querysat_oracle as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-4
OP_RTOL = 1e-4


@pytest.fixture(scope="module")
def rt():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    from diffusionsat_amd import runtime
    return runtime


def _session(rt, batch, ws, max_rounds=32):
    g = rt.Graph(batch)
    w = rt.Weights(ws)
    return rt.Session(g, w, max_rounds=max_rounds)


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _close(a, b, rtol, atol):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    assert np.all(np.isfinite(a)), "HIP output has non-finite values"
    lim = atol + rtol * np.abs(b)
    assert np.all(err <= lim), "max err %.3e (limit %.3e) at %s" % (
        err.max(), lim.flat[np.argmax(err - lim)], np.unravel_index(np.argmax(err - lim), err.shape))


@pytest.mark.parametrize("kind", ["3sat", "ragged"])
@pytest.mark.parametrize("mlp", [0, 1, 2, 3, 4])
def test_op_mlp(rt, kind, mlp):
    """MLP.call (model/mlp.py:42-50) for each of the five MLPs."""
    _, batch, _ = make_all(kind)
    ws = oracle_weights(3)
    s = _session(rt, batch, ws)
    name = O.MLP_ORDER[mlp]
    rows = batch.n_clauses if mlp == 2 else batch.n_vars
    rng = np.random.default_rng(mlp)
    x = rng.standard_normal((rows, ws[name][0][0].shape[0])).astype(np.float32)
    y = s.op_mlp(mlp, _dev(x)).cpu().numpy()
    ref = O.mlp(x.astype(np.float64), [(w.astype(np.float64), b.astype(np.float64)) for w, b in ws[name]])
    _close(y, ref, OP_RTOL, 1e-5)


@pytest.mark.parametrize("kind", ["3sat", "ragged", "medium"])
def test_op_sparse_and_norm(rt, kind):
    """The two sparse half-rounds and PairNorm, one operator at a time."""
    _, batch, data = make_all(kind)
    s = _session(rt, batch, oracle_weights(1))
    n, c = batch.n_vars, batch.n_clauses
    rng = np.random.default_rng(7)
    q = rng.standard_normal((n, 128)).astype(np.float32) * 2
    lits = rng.standard_normal((2 * n, 128)).astype(np.float32)
    adj = data["adjacency_matrix"].astype(np.float64)
    cl_adj = adj.T.tocsr()
    # literal -> clause: loss/sat.py:125-137 and query_sat.py:255-256
    closs, cmsg = s.op_clause_gather(_dev(q), _dev(lits))
    ref_closs = O.softplus_loss_adj(q.astype(np.float64), cl_adj)
    rev = 1.0 / np.sqrt(np.maximum(np.asarray(cl_adj.sum(axis=1)), 1.0))
    ref_cmsg = (cl_adj @ lits.astype(np.float64)) * rev
    _close(closs.cpu().numpy(), ref_closs, OP_RTOL, 1e-6)
    _close(cmsg.cpu().numpy(), ref_cmsg, OP_RTOL, 1e-5)
    # clause -> literal: query_sat.py:242-246 (closed form) and :269-273
    vla = rng.standard_normal((c, 128)).astype(np.float32)
    cl = np.abs(rng.standard_normal((c, 128))).astype(np.float32)
    grad, vloss = s.op_literal_gather(_dev(q), _dev(cl), _dev(vla))
    deg = np.asarray(adj.sum(axis=1))
    m = adj @ cl.astype(np.float64)
    vdw = 4.0 / np.sqrt(np.maximum(deg[:n] + deg[n:], 1.0))
    q64 = q.astype(np.float64)
    ref_grad = (-O.sigmoid(q64) * m[:n] + O.sigmoid(-q64) * m[n:]) * vdw
    ref_vloss = (adj @ vla.astype(np.float64)) / np.sqrt(np.maximum(deg, 1.0))
    _close(grad.cpu().numpy(), ref_grad, OP_RTOL, 1e-5)
    _close(vloss.cpu().numpy(), ref_vloss, OP_RTOL, 1e-5)
    # PairNorm (layers/normalization.py:43-71), both sides
    for side, rows, gk, mk in ((0, n, "variables_graph_adj", "variables_mask"),
                               (1, c, "clauses_graph_adj", "clauses_mask")):
        x = (rng.standard_normal((rows, 128)) * 3 + 1).astype(np.float32)
        gm = data[gk].astype(np.float64)
        import scipy.sparse as sp
        gn = sp.diags(1.0 / np.asarray(gm.sum(axis=1)).ravel()) @ gm
        ref = O.pair_norm(x.astype(np.float64), gn.tocsr(), data[mk])
        y = s.op_pair_norm(side, _dev(x))
        _close(y.cpu().numpy(), ref, OP_RTOL, 1e-5)


@pytest.mark.parametrize("kind,noise_scale", [("3sat", 0.5), ("ragged", 1.0 / 32), ("uf20", 1.0), ("medium", 0.25)])
def test_diffusion_step_round_logits(rt, kind, noise_scale):
    """Per-round [N,8] logits of QuerySAT.loop (query_sat.py:186-373) for 32 rounds, the
    returned prediction and the selected logit map, all randomness injected."""
    _, batch, data = make_all(kind)
    ws = oracle_weights(5)
    rounds = 32
    d = draws(batch.n_vars, rounds, seed=11)
    x = np.full((batch.n_vars, 2), 0.5, dtype=np.float32)
    noisy = O.randomized_rounding(x, d["uniform"])
    ref = O.diffusion_step(ws, data, np.float32(noise_scale), noisy, d["normal"], d["labels"], rounds,
                           early_exit=False, dt=np.float32, trace=True)
    s = _session(rt, batch, ws, max_rounds=rounds)
    for graph_replay in (False, True):
        s.set_graph_replay(graph_replay)
        out = s.diffusion_step(noise_scale, _dev(noisy), _dev(d["normal"]), _dev(d["labels"]), rounds=rounds,
                               early_exit=False, want_logits=True, want_round_logits=not graph_replay)
        torch.cuda.synchronize()
        if not graph_replay:
            rl = out["round_logits"].cpu().numpy()
            for r in range(rounds):
                _close(rl[r], ref["trace"]["logits"][r], 0, LOGIT_TOL)
        _close(out["logits"].cpu().numpy(), ref["logits"], 0, LOGIT_TOL)
        assert int(out["steps_taken"].item()) == ref["steps_taken"] == rounds - 1
        # argmin map: compare where the oracle's margin between best and runner-up is clear
        pgl = ref["trace"]["per_graph_loss"][-1]
        srt = np.sort(pgl, axis=1)
        clear = (srt[:, 1] - srt[:, 0]) > 1e-3
        got = out["best_map"].cpu().numpy()
        assert np.array_equal(got[clear], ref["best_map"][clear])
        if noise_scale == 1.0:   # every KL is exactly 0 at t=1: first index must win
            assert np.all(got == 0)
        vmask = data["variables_mask"]
        sel = clear[vmask]
        _close(out["prediction"].cpu().numpy()[sel], ref["prediction"][sel], 0, LOGIT_TOL)


def test_early_exit_and_is_batch_sat(rt):
    """is_batch_sat (utils/sat.py:118-124) and the break of query_sat.py:330-338: a batch
    whose clauses are tautologies is satisfied by any assignment -> exit in round 0."""
    from diffusionsat_amd.batch import make_batch
    insts = [(5, [[1, -1], [2, -2, 3], [-4, 4], [5, -5]]), (3, [[1, -1], [2, -2], [3, -3]])]
    batch = make_batch(insts)
    data = O.create_adj_matrices(insts)
    ws = oracle_weights(2)
    d = draws(batch.n_vars, 8, seed=3)
    noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, dtype=np.float32), d["uniform"])
    ref = O.diffusion_step(ws, data, np.float32(0.5), noisy, d["normal"], d["labels"], 8, early_exit=True,
                           dt=np.float32)
    assert ref["steps_taken"] == 0
    s = _session(rt, batch, ws, max_rounds=8)
    for graph_replay in (False, True):
        s.set_graph_replay(graph_replay)
        out = s.diffusion_step(0.5, _dev(noisy), _dev(d["normal"]), _dev(d["labels"]), rounds=8, early_exit=True,
                               want_logits=True)
        assert int(out["steps_taken"].item()) == 0
        _close(out["logits"].cpu().numpy(), ref["logits"], 0, LOGIT_TOL)
    # a formula no assignment of which satisfies all clauses never exits early
    insts2 = [(2, [[1], [-1], [2, 1]])]
    b2 = make_batch(insts2)
    s2 = _session(rt, b2, ws, max_rounds=4)
    d2 = draws(2, 4, seed=1)
    out = s2.diffusion_step(0.5, _dev(np.array([[1, 0], [0, 1]], dtype=np.float32)), _dev(d2["normal"]),
                            _dev(d2["labels"]), rounds=4, early_exit=True)
    assert int(out["steps_taken"].item()) == 3
    # is_batch_sat on explicit logits
    _, batch3, data3 = make_all("3sat")
    s3 = _session(rt, batch3, ws)
    planted = []
    from diffusionsat_amd.instances import planted_ksat
    for i in range(3):
        planted.append(planted_ksat(12 + 3 * i, 40 + 7 * i, 3, seed=i)[2])
    bits = np.concatenate(planted).astype(np.float32)
    good = (bits * 2 - 1) * 3.0
    cl_adj = data3["adjacency_matrix"].T.tocsr()
    assert s3.op_is_batch_sat(_dev(good)) == 1 == int(O.is_batch_sat(good.reshape(-1, 1), cl_adj))
    bad = -good
    assert s3.op_is_batch_sat(_dev(bad)) == int(O.is_batch_sat(bad.reshape(-1, 1), cl_adj))


@pytest.mark.parametrize("kind", ["3sat", "ragged"])
def test_sample_matches_oracle_diffusion(rt, kind, self_cond=False):
    """dsat_sample vs diffusion() (satuniformity/DiffusionSampler.py:78-191), T=6 steps of
    R=8 rounds with all noise injected; compared step by step on the prediction logits while
    the discrete trajectory (randomized rounding) agrees, which the margins make robust."""
    insts, batch, data = make_all(kind)
    ws = oracle_weights(9)
    T, R = 6, 8
    d = draws(batch.n_vars, R, steps=T, seed=21)
    acc, final, steps = O.diffusion(ws, data, T, d["normal"], d["labels"], d["uniform"], rounds=R, early_exit=True,
                                    dt=np.float32, trace=True, self_supervised=self_cond)
    s = _session(rt, batch, ws, max_rounds=R)
    for graph_replay in (False, True):
        s.set_graph_replay(graph_replay)
        out = s.sample(T, R, seed=0, early_exit=True, uniform=_dev(d["uniform"].reshape(T, -1)),
                       normal=_dev(d["normal"]), labels=_dev(d["labels"]), want_predictions=True,
                       self_conditioning=self_cond)
        preds = out["predictions"].cpu().numpy()
        for t in range(T):
            # selected-map logits only where the oracle's map choice was clear-cut
            _close(preds[t], steps[t]["logits"], 0, 5e-4)
        assert np.array_equal(out["assignment"].cpu().numpy().astype(np.float32), final.astype(np.float32))
        assert abs(out["solved_fraction"] - acc) < 1e-6


def test_sample_self_conditioned_matches_oracle(rt):
    """The sampler with the reference's `self_supervised` switch (DiffusionSampler.py:21, :110-113):
    step t >= 1 is diffusion_step_self on sigmoid(prediction) of step t-1."""
    test_sample_matches_oracle_diffusion(rt, "3sat", self_cond=True)
    # and the switch changes the trajectory (the input is not ignored)
    _, batch, data = make_all("3sat")
    ws = oracle_weights(9)
    d = draws(batch.n_vars, 8, steps=6, seed=21)
    a = O.diffusion(ws, data, 6, d["normal"], d["labels"], d["uniform"], rounds=8, trace=True)[2]
    b = O.diffusion(ws, data, 6, d["normal"], d["labels"], d["uniform"], rounds=8, trace=True,
                    self_supervised=True)[2]
    assert np.abs(a[0]["logits"] - b[0]["logits"]).max() == 0 and np.abs(a[1]["logits"] - b[1]["logits"]).max() > 1e-3


def test_device_rng_matches_philox_mirror(rt):
    """Device Philox4x32-10 draws vs the NumPy mirror (oracle/philox.py): integers bit-exact,
    Box-Muller normals to 1e-5."""
    from oracle import philox
    _, batch, _ = make_all("3sat")
    s = _session(rt, batch, oracle_weights(0))
    normal, labels, uniform = s.op_rng(seed=0x1234567890ABCDEF, step_index=5, rounds=3)
    n = batch.n_vars
    ref_n = philox.normal(0x1234567890ABCDEF, 5, 3, n)
    ref_l = philox.labels(0x1234567890ABCDEF, 5, n)
    ref_u = philox.uniform(0x1234567890ABCDEF, 5, n)
    assert np.array_equal(labels.cpu().numpy(), ref_l)
    assert np.array_equal(uniform.cpu().numpy(), ref_u)
    _close(normal.cpu().numpy(), ref_n, 1e-5, 1e-5)


def test_diffusion_step_self_conditioning(rt):
    """QuerySAT.diffusion_step_self (query_sat.py:483-498): denoised_num enters v1 as
    [denoised, 1 - denoised]; fused and per-layer paths against the oracle."""
    import os
    _, batch, data = make_all("3sat")
    ws = oracle_weights(6)
    rounds = 6
    d = draws(batch.n_vars, rounds, seed=13)
    noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, dtype=np.float32), d["uniform"])
    den = np.random.default_rng(3).random((batch.n_vars, 1)).astype(np.float32)
    ref = O.diffusion_step(ws, data, np.float32(0.4), noisy, d["normal"], d["labels"], rounds, early_exit=False,
                           dt=np.float32, denoised_num=den)
    plain = O.diffusion_step(ws, data, np.float32(0.4), noisy, d["normal"], d["labels"], rounds, early_exit=False,
                             dt=np.float32)
    assert np.abs(ref["logits"] - plain["logits"]).max() > 1e-3          # the input matters
    for fused in ("1", "0"):
        os.environ["DSAT_FUSED"] = fused
        try:
            s = _session(rt, batch, ws, max_rounds=rounds)
        finally:
            os.environ.pop("DSAT_FUSED", None)
        out = s.diffusion_step(0.4, _dev(noisy), _dev(d["normal"]), _dev(d["labels"]), rounds=rounds,
                               early_exit=False, want_logits=True, denoised=_dev(den.reshape(-1)))
        _close(out["logits"].cpu().numpy(), ref["logits"], 0, LOGIT_TOL)


def test_facade_predict_step_and_self_conditioning(rt):
    """QuerySAT.predict_step (model/query_sat.py:419-451, prediction_tries == 1) with its host draws
    and the device Philox normals reproduced on the oracle side, and QuerySAT.diffusion_step_self
    (:483-498) through the facade."""
    from diffusionsat_amd.query_sat import GraphInputs, QuerySAT
    from oracle import philox
    _, batch, data = make_all("3sat")
    ws = oracle_weights(4)
    R, seed = 8, 5
    model = QuerySAT(optimizer=None, test_rounds=R, seed=seed)
    model.set_weights(ws)
    gi = GraphInputs(batch)
    out = model.predict_step(gi)
    assert set(out) == {"steps_taken", "loss", "prediction"}
    n = batch.n_vars
    rng = np.random.default_rng(seed + 7919 * 1)                 # the draws of predict_step's first call
    noise_scale = float(rng.random())
    labels = rng.integers(0, 2, n).astype(np.int32)
    one_hot = np.stack([1 - labels, labels], axis=1).astype(np.float32)
    t = np.float32(noise_scale) ** np.float32(0.5)
    num_at_t = one_hot * (1 - t) + t / 2
    rounded = np.floor(num_at_t[:, 0:1] + rng.random((n, 1)).astype(np.float32))
    noisy = np.concatenate([rounded, 1 - rounded], axis=1).astype(np.float32)
    normal = philox.normal(seed, 1, R, n)                        # step_index = call counter
    ref = O.diffusion_step(ws, data, np.float32(noise_scale), noisy, normal, labels, R, early_exit=True,
                           dt=np.float32)
    assert int(out["steps_taken"].item()) == ref["steps_taken"]
    _close(out["prediction"].cpu().numpy(), ref["prediction"], 0, 5e-4)
    # self-conditioning through the facade == the oracle with denoised_num
    d = draws(n, R, seed=3)
    den = np.random.default_rng(1).random((n, 1)).astype(np.float32)
    out2 = model.diffusion_step_self(gi, None, None, None, 0.3, _dev(noisy), _dev(den), normal=_dev(d["normal"]),
                                     labels=_dev(d["labels"]))
    ref2 = O.diffusion_step(ws, data, np.float32(0.3), noisy, d["normal"], d["labels"], R, early_exit=True,
                            dt=np.float32, denoised_num=den)
    _close(out2["prediction"].cpu().numpy(), ref2["prediction"], 0, 5e-4)


def test_predict_step_with_several_tries_and_is_graph_sat(rt):
    """QuerySAT.predict_step with prediction_tries > 1 (model/query_sat.py:423-443): each try is a fresh
    call; a graph keeps the prediction of the first try that satisfies it (is_graph_sat,
    utils/sat.py:165-180), unsolved graphs stay at zero.  Tiny formulas so that random weights do solve
    some graphs; the host draws and device Philox normals are reproduced on the oracle side."""
    from diffusionsat_amd.batch import make_batch
    from diffusionsat_amd.query_sat import GraphInputs, QuerySAT
    from oracle import philox
    insts = [(3, [[1, 2], [-1, 3]]), (4, [[1, -2, 3], [2, 4], [-3, -4]]), (2, [[1], [-1]]), (3, [[1, -1], [2, -2, 3]])]
    batch = make_batch(insts)
    data = O.create_adj_matrices(insts)
    ws = oracle_weights(4)
    R, seed, tries = 6, 9, 4
    model = QuerySAT(optimizer=None, test_rounds=R, seed=seed)
    model.set_weights(ws)
    model.prediction_tries = tries
    gi = GraphInputs(batch)
    out = model.predict_step(gi)
    n, G = batch.n_vars, batch.n_graphs
    vmask = data["variables_mask"]
    final = np.zeros(n, np.float32)
    solved = np.zeros(G, np.float32)
    adj = data["adjacency_matrix"]
    for call in range(1, tries + 1):
        rng = np.random.default_rng(seed + 7919 * call)
        noise_scale = float(rng.random())
        labels = rng.integers(0, 2, n).astype(np.int32)
        one_hot = np.stack([1 - labels, labels], axis=1).astype(np.float32)
        t = np.float32(noise_scale) ** np.float32(0.5)
        rounded = np.floor((one_hot * (1 - t) + t / 2)[:, 0:1] + rng.random((n, 1)).astype(np.float32))
        noisy = np.concatenate([rounded, 1 - rounded], axis=1).astype(np.float32)
        ref = O.diffusion_step(ws, data, np.float32(noise_scale), noisy, philox.normal(seed, call, R, n), labels, R,
                               early_exit=True, dt=np.float32)
        sat = O.is_graph_sat(ref["prediction"].astype(np.float32), adj, data["clauses_graph_adj"])
        new = np.clip(sat - solved, 0, 1)
        final += ref["prediction"] * new[vmask]
        solved += new
        # the single operator on the same logits
        s = model._session(gi)
        got = s.op_is_graph_sat(_dev(ref["prediction"].astype(np.float32))).cpu().numpy()
        assert np.array_equal(got, sat.astype(np.uint8))
    assert solved[2] == 0 and solved[3] == 1            # x & -x is never solved, the tautology always is
    _close(out["prediction"].cpu().numpy(), final, 0, 5e-4)
    assert np.all(out["prediction"].cpu().numpy()[vmask == 2] == 0)


def test_device_packed_assignment_bits(rt):
    """dsat_op_pack_assignment: per graph, little-endian bits (bit i <-> x_{i+1}) -- the int encoding of
    utils/VariableAssignment.py:63-69 as bytes; graphs of different sizes, sizes not multiples of 8."""
    from diffusionsat_amd.dimacs import VariableAssignment
    insts, batch, _ = make_all("ragged")
    s = _session(rt, batch, oracle_weights(0))
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2, batch.n_vars).astype(np.uint8)
    packed = s.pack_assignment(_dev(bits)).cpu().numpy()
    off = s.packed_offsets()
    assert off[-1] == packed.size == sum((n + 7) // 8 for n, _ in insts)
    v0 = 0
    for g, (n, clauses) in enumerate(insts):
        chunk = packed[off[g]:off[g + 1]]
        assert np.array_equal(chunk, np.packbits(bits[v0:v0 + n], bitorder="little"))
        va = VariableAssignment(n_vars=n, clauses=clauses)
        va.assign_all_from_bit_list(bits[v0:v0 + n])
        assert int.from_bytes(chunk.tobytes(), "little") == int(va)          # the reference's int encoding
        v0 += n


def test_hip_path_symmetries(rt):
    """The exact symmetries of the path, on the HIP path itself (tests/test_oracle.py has them for the oracle):
    renumbering the variables, reordering the clauses and shuffling the literals inside clauses -- what
    the reference's invariance_shuffle_literals feeds the model (diffusion_training.py:455-463) -- leave
    the per-variable logits unchanged up to summation order."""
    from diffusionsat_amd.batch import make_batch
    (n, clauses), = make_all("uf20")[0]
    ws = oracle_weights(6)
    rounds = 6
    d = draws(n, rounds, seed=9)
    noisy = O.randomized_rounding(np.full((n, 2), 0.5, dtype=np.float32), d["uniform"])

    def run(cl, perm=None):
        inv = np.arange(n) if perm is None else np.argsort(perm)
        s = _session(rt, make_batch([(n, cl)]), ws, max_rounds=rounds)
        out = s.diffusion_step(0.6, _dev(noisy[inv]), _dev(d["normal"][:, inv]), _dev(d["labels"][inv]), rounds=rounds,
                               early_exit=False, want_logits=True)
        return out["logits"].cpu().numpy()

    base = run(clauses)
    rng = np.random.default_rng(1)
    perm = rng.permutation(n)
    pc = [[int(np.sign(l)) * int(perm[abs(l) - 1] + 1) for l in c] for c in clauses]
    assert np.abs(run(pc, perm)[perm] - base).max() < 2e-5
    order = rng.permutation(len(clauses))
    sc = [[int(x) for x in rng.permutation(clauses[i])] for i in order]
    assert np.abs(run(sc) - base).max() < 2e-5
    # inverting every literal is not a symmetry of the network (see the oracle test): really different
    assert np.abs(run([[-l for l in c] for c in clauses]) - base).max() > 1e-3


def test_objects_are_bound_to_their_device(rt):
    """Handles remember the HIP device they were created on; a call made with another device current is
    refused (DSAT_E_STATE) instead of launching on the wrong GPU.  Needs two GPUs to provoke."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    import ctypes as C
    from diffusionsat_amd import _lib
    _, batch, _ = make_all("3sat")
    g0 = rt.Graph(batch, device="cuda:0")
    w1 = rt.Weights(oracle_weights(0), device="cuda:1")
    with pytest.raises(ValueError):
        rt.Session(g0, w1)
    h = C.c_void_p()
    with torch.cuda.device(0):
        rc = _lib.load().dsat_session_create(g0.handle, w1.handle, 4, C.byref(h))
    assert rc == -4 and b"device" in _lib.load().dsat_last_error()
    s0 = rt.Session(g0, rt.Weights(oracle_weights(0), device="cuda:0"))
    pred = torch.empty(batch.n_vars, device="cuda:0")
    noisy = torch.zeros(batch.n_vars, 2, device="cuda:0")
    with torch.cuda.device(1):      # the raw call with the wrong device current
        rc = _lib.load().dsat_diffusion_step(s0.handle, 0.5, noisy.data_ptr(), None, None, 0, 0, 4, 0, None, None,
                                              pred.data_ptr(), None, None, None)
    assert rc == -4
    out = s0.diffusion_step(0.5, noisy, rounds=4)            # the Python handle switches devices itself
    assert out["prediction"].device.index == 0


@pytest.mark.parametrize("hidden,layers", [((100, 300, 150, 200, 64), (2, 2, 2, 3, 2)),
                                           ((153, 512, 204, 230, 128), (2, 4, 2, 2, 3)),
                                           ((40, 64, 48, 72, 24), (3, 3, 3, 3, 2)),
                                           ((64, 96, 256, 256, 32), (2, 2, 2, 2, 2)),     # 16 column tiles, 2-layer update
                                           ((64, 96, 17, 33, 32), (2, 2, 2, 3, 2))])      # narrow split layers
@pytest.mark.parametrize("wide", [False, True])
def test_other_layer_counts_and_hidden_widths(rt, hidden, layers, wide):
    """The reference constructor derives layer counts and hidden widths from its arguments
    (model/query_sat.py:101-122: msg_layers, the *_scale factors).  Anything but the released shapes leaves
    the compile-time-blocked chains and runs the generic per-layer weight-streaming path (and, for P1,
    the in-tile output/query pair): 6 rounds of logits against the oracle for three such layouts."""
    from diffusionsat_amd import _lib
    _, batch, data = make_all("medium")
    lay = _lib.default_layout()
    for m in range(5):
        lay.n_layers[m] = layers[m]
        lay.hidden[m] = hidden[m]
    v1 = 128 + 9
    ins = (v1, v1, 384, 128 + v1 + 256, 128)
    outs = (128, 256, 256, 128, 8)
    dims = {name: [ins[m]] + [hidden[m]] * (layers[m] - 1) + [outs[m]] for m, name in enumerate(O.MLP_ORDER)}
    ws = O.make_weights(13, np.float32, bias_scale=0.05, dims=dims)
    rounds = 6
    d = draws(batch.n_vars, rounds, seed=17)
    noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, dtype=np.float32), d["uniform"])
    ref = O.diffusion_step(ws, data, np.float32(0.5), noisy, d["normal"], d["labels"], rounds, early_exit=False,
                           dt=np.float32, trace=("logits",))
    # wide: P3 on 32-row tiles where the layout allows the split-product tile (what large batches run); a 2-layer
    # update MLP then reads its only hidden tile in the last layer, so the stage of the column sums must be elsewhere
    env = {"DSAT_P3_BM": "32", "DSAT_P1_A2": "1"} if wide else {}
    os.environ.update(env)
    try:
        s = rt.Session(rt.Graph(batch), rt.Weights(ws, lay), max_rounds=rounds)
    finally:
        for k in env:
            os.environ.pop(k, None)
    if wide and hidden == (64, 96, 256, 256, 32):
        assert s.split_mask() & 16          # the 2-layer update MLP does run on the 32-row split-product tile
    out = s.diffusion_step(0.5, _dev(noisy), _dev(d["normal"]), _dev(d["labels"]), rounds=rounds, early_exit=False,
                           want_round_logits=True)
    got = out["round_logits"].cpu().numpy()
    for r in range(rounds):
        _close(got[r], ref["trace"]["logits"][r], 0, LOGIT_TOL)


def _fuzz_instances(seed):
    """Random batch structure: many graphs of 1..70 variables (so a 16-row tile can span up to 16 graphs and a
    64-row clause tile dozens), clause widths 1..7 with repeated and complementary literals, variables
    that occur nowhere, graphs with a single clause."""
    rng = np.random.default_rng(1000 + seed)
    style = seed % 4
    n_graphs = int(rng.integers(1, 60))
    insts = []
    for _ in range(n_graphs):
        if style == 0:
            n = int(rng.integers(1, 4))                 # tiny graphs only
        elif style == 1:
            n = int(rng.integers(1, 70))
        elif style == 2:
            n = int(rng.choice([1, 2, 16, 17, 31, 32, 33, 64, 65]))   # tile-boundary sizes
        else:
            n = int(rng.integers(1, 12)) if rng.random() < 0.8 else int(rng.integers(100, 300))
        m = int(rng.integers(1, max(2, 5 * n)))
        clauses = []
        for _ in range(m):
            k = int(rng.integers(1, 8))
            lits = rng.integers(1, n + 1, k) * rng.choice([-1, 1], k)
            clauses.append([int(x) for x in lits])
        insts.append((n, clauses))
    return insts


@pytest.mark.parametrize("seed", range(8))
def test_random_batch_structures_match_oracle(rt, seed):
    """Structure fuzz (the reference batches whatever the dataset yields, data/k_sat.py:60-117: any mix of
    sizes): 3 rounds of logits against the oracle on the four kernel forms (fused tiles with split products,
    the same with lit_mlp on 32-row split tiles, fused tiles on the fp32 MFMA, per-layer kernels), then is_graph_sat and the packed bits."""
    insts = _fuzz_instances(seed)
    from diffusionsat_amd.batch import make_batch
    batch = make_batch(insts)
    data = O.create_adj_matrices(insts)
    ws = oracle_weights(seed)
    rounds = 3
    d = draws(batch.n_vars, rounds, seed=seed)
    noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, dtype=np.float32), d["uniform"])
    ref = O.diffusion_step(ws, data, np.float32(0.3), noisy, d["normal"], d["labels"], rounds, early_exit=False,
                           dt=np.float32, trace=("logits",))
    g = rt.Graph(batch)
    w = rt.Weights(ws)
    for env in ({}, {"DSAT_P1_A2": "1", "DSAT_P3_BM": "32"}, {"DSAT_P2_SPLIT": "0", "DSAT_P3_SPLIT": "0", "DSAT_P1_A2": "0", "DSAT_P1_BSPLIT": "0", "DSAT_P1_ASPLIT": "0"},
                {"DSAT_FUSED": "0"}):
        os.environ.update(env)
        try:
            s = rt.Session(g, w, max_rounds=rounds)
            out = s.diffusion_step(0.3, _dev(noisy), _dev(d["normal"]), _dev(d["labels"]), rounds=rounds,
                                   early_exit=False, want_round_logits=True)
        finally:
            for k in env:
                os.environ.pop(k, None)
        got = out["round_logits"].cpu().numpy()
        for r in range(rounds):
            _close(got[r], ref["trace"]["logits"][r], 0, LOGIT_TOL)
    # is_graph_sat on a random +-1 prediction: integer work, exact
    pred = np.random.default_rng(seed).choice([-2.0, 2.0], batch.n_vars).astype(np.float32)
    adj = data["adjacency_matrix"]
    sat_ref = O.is_graph_sat(pred, adj, data["clauses_graph_adj"])
    got = s.op_is_graph_sat(_dev(pred)).cpu().numpy()
    assert np.array_equal(got.astype(np.int64), np.asarray(sat_ref).astype(np.int64))
    bits = (pred > 0).astype(np.uint8)
    packed = s.pack_assignment(_dev(bits)).cpu().numpy()
    off = s.packed_offsets()
    v0 = 0
    for gi, (n, _) in enumerate(insts):
        assert np.array_equal(packed[off[gi]:off[gi + 1]], np.packbits(bits[v0:v0 + n], bitorder="little"))
        v0 += n


def test_throughput_preference_switches_the_tile_form_not_the_result(rt):
    """dsat_session_set_throughput: a session told that several batches are in flight runs lit_mlp on 32-row
    split-product tiles (bit 2 of dsat_session_split_mask); toggling it between steps -- replayed graphs are cached
    per form -- changes nothing beyond fp32 rounding, and each form is reproducible."""
    _, batch, data = make_all("medium")
    ws = oracle_weights(3)
    rounds = 8
    d = draws(batch.n_vars, rounds, seed=31)
    noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, dtype=np.float32), d["uniform"])
    ref = O.diffusion_step(ws, data, np.float32(0.5), noisy, d["normal"], d["labels"], rounds, early_exit=False,
                           dt=np.float32)["logits"]
    s = rt.Session(rt.Graph(batch), rt.Weights(ws), max_rounds=rounds)
    got = {}
    for i, on in enumerate((False, True, False, True)):
        s.set_throughput(on)
        assert bool(s.split_mask() & 4) == on
        out = s.diffusion_step(0.5, _dev(noisy), _dev(d["normal"]), _dev(d["labels"]), rounds=rounds, early_exit=False,
                               want_logits=True)["logits"].cpu().numpy()
        _close(out, ref, 0, LOGIT_TOL)
        if on in got:
            assert np.array_equal(got[on], out)          # the same form, the same bits
        got[on] = out


def test_update_gate_partial_sums_from_p2s_idle_cus(rt):
    """Role D of P2: the `variables` k-steps of update_gate's first layer (model/query_sat.py:277-278)
    are formed by extra workgroups of P2 and P3 starts its accumulators from them (bit 6 of
    dsat_session_split_mask).  Off (DSAT_P3_PART=0, or a session told that batches overlap) P3 streams the whole
    layer itself: another summation order, the same logits to fp32 rounding; a ragged batch whose last 64-row group
    and last 16-row tile are partial."""
    _, batch, data = make_all("ragged")
    ws = oracle_weights(5)
    rounds = 8
    d = draws(batch.n_vars, rounds, seed=41)
    noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, dtype=np.float32), d["uniform"])
    ref = O.diffusion_step(ws, data, np.float32(0.5), noisy, d["normal"], d["labels"], rounds, early_exit=True,
                           dt=np.float32)
    outs = {}
    for env in ({}, {"DSAT_P3_PART": "0"}):
        os.environ.update(env)
        try:
            s = rt.Session(rt.Graph(batch), rt.Weights(ws), max_rounds=rounds)
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert bool(s.split_mask() & 64) == (not env)
        for on in (False, True):      # ... and the session-level switch
            s.set_throughput(on)
            assert bool(s.split_mask() & 64) == (not env and not on)
            out = s.diffusion_step(0.5, _dev(noisy), _dev(d["normal"]), _dev(d["labels"]), rounds=rounds,
                                   early_exit=True, want_logits=True)
            _close(out["logits"].cpu().numpy(), ref["logits"], 0, LOGIT_TOL)
            assert int(out["steps_taken"].item()) == ref["steps_taken"]
            outs[(bool(env), on)] = out["logits"].cpu().numpy()
    # told that batches overlap, both sessions run the same code (no role D, lit_mlp on 32-row tiles): the same bits
    assert np.array_equal(outs[(True, True)], outs[(False, True)])


@pytest.mark.parametrize("fm,qm", [(64, 64), (128, 64), (96, 128), (32, 48), (100, 20)])
def test_other_feature_and_query_widths(rt, fm, qm):
    """feature_maps / query_maps other than the released 128 / 128 (the reference constructor takes them,
    model/query_sat.py:86-89, and scales the hidden widths with them, :101-122): states, queries, messages and losses
    narrower than 128 live zero-padded in the kernels' 128-wide rows, the weights are packed to match, PairNorm's
    per-row variance is over the real features.  6 rounds of logits against the oracle, fused and per-layer."""
    from diffusionsat_amd import _lib
    _, batch, data = make_all("medium")
    dims = O.mlp_dims(feature_maps=fm, query_maps=qm)
    lay = _lib.default_layout()
    lay.feature_maps, lay.query_maps = fm, qm
    for m, name in enumerate(O.MLP_ORDER):
        lay.n_layers[m] = len(dims[name]) - 1
        lay.hidden[m] = dims[name][1]
    ws = O.make_weights(21, np.float32, bias_scale=0.05, dims=dims)
    rounds = 6
    d = draws(batch.n_vars, rounds, seed=19)
    noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, dtype=np.float32), d["uniform"])
    ref = O.diffusion_step(ws, data, np.float32(0.5), noisy, d["normal"], d["labels"], rounds, early_exit=False,
                           dt=np.float32, trace=("logits",))
    for env in ({}, {"DSAT_P3_BM": "32"}, {"DSAT_FUSED": "0"}):
        os.environ.update(env)
        try:
            s = rt.Session(rt.Graph(batch), rt.Weights(ws, lay), max_rounds=rounds)
        finally:
            for k in env:
                os.environ.pop(k, None)
        out = s.diffusion_step(0.5, _dev(noisy), _dev(d["normal"]), _dev(d["labels"]), rounds=rounds, early_exit=False,
                               want_round_logits=True)
        got = out["round_logits"].cpu().numpy()
        for r in range(rounds):
            _close(got[r], ref["trace"]["logits"][r], 0, LOGIT_TOL)


def test_facade_with_narrower_widths(rt):
    """QuerySAT(feature_maps=64, query_maps=96) through the facade: the constructor derives the hidden widths as the
    reference does (model/query_sat.py:101-122), and diffusion_step_self matches the oracle run on the same weights."""
    from diffusionsat_amd.query_sat import GraphInputs, QuerySAT
    _, batch, data = make_all("3sat")
    R = 6
    model = QuerySAT(optimizer=None, test_rounds=R, seed=2, feature_maps=64, query_maps=96)
    dims = O.mlp_dims(feature_maps=64, query_maps=96)
    assert [int(model.layout.hidden[m]) for m in range(5)] == [dims[k][1] for k in O.MLP_ORDER]
    ws = O.make_weights(9, np.float32, bias_scale=0.05, dims=dims)
    model.set_weights(ws)
    gi = GraphInputs(batch)
    n = batch.n_vars
    d = draws(n, R, seed=3)
    noisy = O.randomized_rounding(np.full((n, 2), 0.5, dtype=np.float32), d["uniform"])
    den = np.random.default_rng(1).random((n, 1)).astype(np.float32)
    out = model.diffusion_step_self(gi, None, None, None, 0.3, _dev(noisy), _dev(den), normal=_dev(d["normal"]),
                                    labels=_dev(d["labels"]))
    ref = O.diffusion_step(ws, data, np.float32(0.3), noisy, d["normal"], d["labels"], R, early_exit=True,
                           dt=np.float32, denoised_num=den)
    assert int(out["steps_taken"].item()) == ref["steps_taken"]
    _close(out["prediction"].cpu().numpy(), ref["prediction"], 0, 5e-4)
