"""The Python boundary is the reference's: ``diffusion_step`` takes the three sparse tensors of
``model/query_sat.py:467-481`` exactly as ``data/SatSpecifics.py:21-69`` builds them, ``MLP`` / ``PairNorm`` have
the argument lists of ``model/mlp.py:13-19,42-50`` and ``layers/normalization.py:31,43``."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import querysat_oracle as O
from tests.helpers import draws, make_all, oracle_weights

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _reference_tensors(insts):
    """(indices, values, dense_shape) triples in the reference's own entry order: positives clause by clause,
    then negatives shifted by N (data/dimac.py:14-18, data/SatSpecifics.py:25-35, :40-57)."""
    batched, off = [], 0
    for n, clauses in insts:
        batched += O.shift_clause(clauses, off)
        off += n
    n, c, g = off, len(batched), len(insts)
    pos, neg = O.compute_adj_indices(batched)
    pos = np.asarray(pos, np.int64).reshape(-1, 2)
    neg = np.asarray(neg, np.int64).reshape(-1, 2) + np.array([[n, 0]], np.int64)
    idx = np.concatenate([pos, neg], axis=0)
    vmask = np.repeat(np.arange(g), [i[0] for i in insts])
    cmask = np.repeat(np.arange(g), [len(i[1]) for i in insts])
    adj = (idx, np.ones(len(idx), np.float32), (2 * n, c))
    cg = (np.stack([cmask, np.arange(c)], 1).astype(np.int64), np.ones(c, np.float32), (g, c))
    vg = (np.stack([vmask, np.arange(n)], 1).astype(np.int64), np.ones(n, np.float32), (g, n))
    return adj, cg, vg


@pytest.mark.parametrize("kind", ["3sat", "ragged", "medium"])
def test_diffusion_step_takes_the_reference_tensors_bitwise(kind):
    from diffusionsat_amd.query_sat import GraphInputs, QuerySAT
    insts, batch, data = make_all(kind)
    ws = oracle_weights(3)
    model = QuerySAT(optimizer=None, test_rounds=8)
    model.set_weights(ws)
    d = draws(batch.n_vars, 8, seed=4)
    noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, np.float32), d["uniform"])
    kw = dict(normal=_dev(d["normal"]), labels=_dev(d["labels"]))
    ref = model.diffusion_step(GraphInputs(batch), None, None, None, 0.5, _dev(noisy), **kw)
    adj, cg, vg = _reference_tensors(insts)
    for name, triple in [
        ("reference order", (adj, cg, vg)),
        ("oracle's scipy matrices (row-major entries, duplicates summed to 2.0)",
         (data["adjacency_matrix"], data["clauses_graph_adj"], data["variables_graph_adj"])),
        ("torch sparse COO", tuple(torch.sparse_coo_tensor(torch.from_numpy(t[0].T.copy()), torch.from_numpy(t[1]), t[2])
                                   for t in (adj, cg, vg))),
        ("shuffled entries", tuple((t[0][p], t[1][p], t[2]) for t in (adj, cg, vg)
                                   for p in [np.random.default_rng(1).permutation(len(t[1]))])),
    ]:
        out = model.diffusion_step(triple[0], triple[1], triple[2], None, 0.5, noisy, **kw)   # noisy_num: host array, too
        assert torch.equal(out["prediction"], ref["prediction"]), name
        assert int(out["steps_taken"].item()) == int(ref["steps_taken"].item()), name
        again = model.diffusion_step(triple[0], triple[1], triple[2], None, 0.5, _dev(noisy), **kw)
        assert torch.equal(again["prediction"], ref["prediction"]), name          # cached graph, same result
    assert len(model._coo_inputs) <= 4
    # and against the oracle, as every other route
    o = O.diffusion_step(ws, data, np.float32(0.5), noisy, d["normal"], d["labels"], 8, dt=np.float32, trace=True)
    srt = np.sort(o["trace"]["per_graph_loss"][-1], axis=1)
    clear = ((srt[:, 1] - srt[:, 0]) > 1e-3)[data["variables_mask"]]
    assert np.abs(ref["prediction"].cpu().numpy() - o["prediction"])[clear].max() < 2e-4


def test_sparse_create_adj_matrices_feeds_predict_like_the_reference():
    """``predict()`` of satuniformity/DiffusionSampler.py:40-63 passes ``step_data`` entries straight through."""
    from diffusionsat_amd import sparse
    from diffusionsat_amd.query_sat import GraphInputs, QuerySAT
    insts, batch, data = make_all("3sat")
    step_data = sparse.create_adj_matrices(batch)
    model = QuerySAT(optimizer=None, test_rounds=4)
    noisy = np.tile(np.array([[1.0, 0.0]], np.float32), (batch.n_vars, 1))
    lab = _dev(np.zeros(batch.n_vars, np.int32))
    nrm = _dev(np.zeros((4, batch.n_vars, 4), np.float32))
    out = model.diffusion_step(adj_matrix=step_data["adjacency_matrix"], clauses_graph=step_data["clauses_graph_adj"],
                               variables_graph=step_data["variables_graph_adj"], solutions=step_data["solutions"],
                               noise_scale=0.25, noisy_num=noisy, labels=lab, normal=nrm)
    ref = model.diffusion_step(GraphInputs(batch), noise_scale=0.25, noisy_num=_dev(noisy), labels=lab, normal=nrm)
    assert torch.equal(out["prediction"], ref["prediction"])
    with pytest.raises(TypeError):
        model.diffusion_step(step_data["adjacency_matrix"].indices, None, None, None, 0.25, noisy)


def test_mlp_has_the_reference_signature():
    """MLP(layer_count, hidden_nmap, out_nmap, activation=leaky_relu, ...)(inputs, training=None), model/mlp.py:13-50."""
    from diffusionsat_amd.query_sat import MLP, QuerySAT, leaky_relu
    rng = np.random.default_rng(0)
    for layer_count, hidden, out, width, rows in [(3, 230, 128, 521, 777), (2, 7, 3, 5, 1), (1, 9, 33, 130, 65),
                                                  (4, 100, 17, 64, 300)]:
        mlp = MLP(layer_count, hidden, out, activation=leaky_relu, name="m", do_layer_norm=False)
        x = rng.standard_normal((rows, width)).astype(np.float32)
        y = mlp(_dev(x), training=False).cpu().numpy()                  # weights built at the first call (glorot, zero bias)
        wts = mlp.get_weights()
        assert len(mlp.dense_layers) == layer_count and wts[0].shape == (width, hidden if layer_count > 1 else out)
        assert all(np.all(b == 0) for b in wts[1::2])
        layers = list(zip(wts[0::2], wts[1::2]))
        ref = O.mlp(x.astype(np.float64), [(k.astype(np.float64), b.astype(np.float64)) for k, b in layers])
        assert y.shape == (rows, out) and np.abs(y - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
        new = [rng.standard_normal(a.shape).astype(np.float32) * 0.1 for a in wts]     # Keras set_weights
        mlp.set_weights(new)
        ref = O.mlp(x.astype(np.float64), [(k.astype(np.float64), b.astype(np.float64)) for k, b in zip(new[0::2], new[1::2])])
        assert np.abs(mlp(_dev(x)).cpu().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    with pytest.raises(NotImplementedError):
        MLP(2, 8, 8, activation=torch.relu)
    with pytest.raises(NotImplementedError):
        MLP(2, 8, 8, do_layer_norm=True)
    with pytest.raises(NotImplementedError):
        mlp(_dev(x), training=True)
    ob = MLP(2, 8, 4, out_bias=0.5, init_zero=True)
    assert np.allclose(ob(_dev(rng.standard_normal((3, 6)).astype(np.float32))).cpu().numpy(), 0.5)
    # the model's five MLPs are these objects, on the model's weights
    ws = oracle_weights(2)
    model = QuerySAT(optimizer=None)
    model.set_weights(ws)
    assert model.clause_mlp.name == "clause_update" and model.lit_mlp.name == "lit_query"     # Keras names, :120-122
    for attr, width in [("variables_query", 137), ("lit_mlp", 137), ("clause_mlp", 384), ("update_gate", 521),
                        ("variables_output", 128)]:
        x = rng.standard_normal((50, width)).astype(np.float32)
        y = getattr(model, attr)(_dev(x)).cpu().numpy()
        assert np.abs(y - O.mlp(x, ws[attr])).max() < 1e-4
        assert getattr(model, attr).dense_layers[0].kernel is ws[attr][0][0]


def test_pair_norm_has_the_reference_signature():
    """PairNorm(epsilon=1e-6, subtract_mean=False)(inputs, graph=None), layers/normalization.py:31-71."""
    from diffusionsat_amd.query_sat import PairNorm
    rng = np.random.default_rng(1)
    sizes = [5, 1, 300, 40, 2000]
    rows = sum(sizes)
    gid = np.repeat(np.arange(len(sizes)), sizes)
    for width in (128, 7, 200):
        x = (rng.standard_normal((rows, width)) * 3 + 1).astype(np.float32)
        x64 = x.astype(np.float64)
        graph = sp.coo_matrix((np.ones(rows), (gid, np.arange(rows))), shape=(len(sizes), rows)).tocsr()
        graph_norm = graph.multiply(1.0 / graph.sum(axis=1)).tocsr()                    # model/query_sat.py:206-211
        # subtract_mean=False (the default): rows scaled by their RMS only, graph ignored
        y = PairNorm()(_dev(x), graph_norm).cpu().numpy()
        ref = x64 / np.sqrt((x64 ** 2).mean(axis=1, keepdims=True) + 1e-6)
        assert np.abs(y - ref).max() < 2e-5
        # subtract_mean=True with the normalised graph = the oracle's pair_norm
        y = PairNorm(subtract_mean=True)(_dev(x), graph_norm).cpu().numpy()
        ref = O.pair_norm(x64, graph_norm.astype(np.float64), gid)
        assert np.abs(y - ref).max() < 2e-5
        # the tensor's own values are used (an un-normalised membership subtracts the column SUM, as TF would)
        y = PairNorm(subtract_mean=True)(_dev(x), graph).cpu().numpy()
        ref = O.pair_norm(x64, graph.astype(np.float64), gid)
        assert np.abs(y - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
        # graph=None: one graph per batch (normalization.py:55-57)
        y = PairNorm(epsilon=1e-3, subtract_mean=True)(_dev(x)).cpu().numpy()
        c = x64 - x64.mean(axis=0, keepdims=True)
        assert np.abs(y - c / np.sqrt((c ** 2).mean(axis=1, keepdims=True) + 1e-3)).max() < 2e-5


@pytest.mark.parametrize("kind", ["3sat", "ragged", "medium"])
def test_unsupervised_selection_matches_oracle(kind, monkeypatch):
    """QuerySAT(supervised=False): the logit map of a graph is picked by the unsupervised clause loss of its own
    clauses, sqrt(sum softplus_mixed_loss_adj + 1e-6) - sqrt(1e-6) (model/query_sat.py:293-309, loss/sat.py:92-100),
    on the fused kernels and on the per-layer path; the labels play no part."""
    from diffusionsat_amd import runtime as rt
    from diffusionsat_amd.query_sat import GraphInputs, QuerySAT
    insts, batch, data = make_all(kind)
    ws = oracle_weights(6)
    rounds = 6
    d = draws(batch.n_vars, rounds, seed=8)
    noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, np.float32), d["uniform"])
    ref = O.diffusion_step(ws, data, np.float32(0.4), noisy, d["normal"], d["labels"], rounds, early_exit=False,
                           dt=np.float32, trace=("per_graph_loss", "best_map", "logits"), supervised=False)
    sup = O.diffusion_step(ws, data, np.float32(0.4), noisy, d["normal"], d["labels"], rounds, early_exit=False,
                           dt=np.float32, trace=("per_graph_loss",), supervised=True)
    ssrt = np.sort(sup["trace"]["per_graph_loss"][-1], axis=1)
    sclear = (ssrt[:, 1] - ssrt[:, 0]) > 1e-3
    pgl = ref["trace"]["per_graph_loss"][-1]
    srt = np.sort(pgl, axis=1)
    clear = (srt[:, 1] - srt[:, 0]) > 1e-4 * np.maximum(1.0, srt[:, 0])
    assert clear.any()
    for fused in ("1", "0"):
        monkeypatch.setenv("DSAT_FUSED", fused)
        g, w = rt.Graph(batch), rt.Weights(ws)
        s = rt.Session(g, w, max_rounds=rounds)
        s.set_supervised(False)
        args = (0.4, _dev(noisy), _dev(d["normal"]))
        out = s.diffusion_step(*args, _dev(d["labels"]), rounds=rounds, early_exit=False, want_logits=True)
        best = out["best_map"].cpu().numpy()
        assert np.array_equal(best[clear], ref["best_map"][clear]), fused
        assert np.abs(out["logits"].cpu().numpy() - ref["logits"]).max() < 2e-4
        vclear = clear[data["variables_mask"]]
        assert np.abs(out["prediction"].cpu().numpy() - ref["prediction"])[vclear].max() < 2e-4
        flipped = s.diffusion_step(*args, _dev(1 - d["labels"]), rounds=rounds, early_exit=False)     # labels are unused
        assert torch.equal(flipped["best_map"], out["best_map"]) and torch.equal(flipped["prediction"], out["prediction"])
        s.set_supervised(True)                                        # and back: the supervised pick again
        back = s.diffusion_step(*args, _dev(d["labels"]), rounds=rounds, early_exit=False)
        assert np.array_equal(back["best_map"].cpu().numpy()[sclear], sup["best_map"][sclear])
    monkeypatch.delenv("DSAT_FUSED")
    assert not np.array_equal(ref["best_map"], sup["best_map"])      # the two rules really pick differently
    # through the facade
    model = QuerySAT(optimizer=None, test_rounds=rounds, supervised=False)
    model.set_weights(ws)
    o = model.diffusion_step(GraphInputs(batch), None, None, None, 0.4, _dev(noisy), normal=_dev(d["normal"]),
                             labels=_dev(d["labels"]), early_exit=False)
    vclear = clear[data["variables_mask"]]
    assert np.abs(o["prediction"].cpu().numpy() - ref["prediction"])[vclear].max() < 2e-4


def test_unsupervised_early_exit_on_a_tautology():
    """is_batch_sat still ends the loop (model/query_sat.py:330-338) when the unsupervised rule picks the map."""
    from diffusionsat_amd import runtime as rt
    from diffusionsat_amd.batch import make_batch
    batch = make_batch([(6, [[1, -1, 2], [3, -3], [4, 5, -4], [6, -6, 1]])] * 3)
    s = rt.Session(rt.Graph(batch), rt.Weights(oracle_weights(1)), max_rounds=5)
    s.set_supervised(False)
    noisy = _dev(np.tile(np.array([[1, 0]], np.float32), (batch.n_vars, 1)))
    out = s.diffusion_step(0.5, noisy, rounds=5, early_exit=True, seed=3, step_index=1)
    assert int(out["steps_taken"].item()) == 0
    smp = s.sample(3, 5, seed=2, early_exit=True)
    assert smp["solved_fraction"] == 1.0
