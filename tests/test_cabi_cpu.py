"""The C-ABI library on a machine without a GPU: it loads, exports every symbol include/dsat.h
declares, answers the layout queries and validates graphs before touching the device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from diffusionsat_amd import _lib, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "dsat.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsat_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libdsat.so does not export %s" % n
        assert n in _lib.SIGNATURES, "%s has no ctypes prototype in diffusionsat_amd/_lib.py" % n
    assert set(_lib.SIGNATURES) <= set(names)


def test_layout_queries(lib):
    lay = _lib.default_layout()
    assert (lay.feature_maps, lay.query_maps, lay.logit_maps) == (128, 128, 8)
    assert list(lay.n_layers) == [2, 3, 2, 3, 2] and list(lay.hidden) == [153, 512, 204, 230, 128]
    assert lib.dsat_layout_blob_floats(C.byref(lay)) == 856788
    dims = dict(W.layer_dims())
    assert dims["variables_query"] == [(137, 153), (153, 128)]
    assert dims["lit_mlp"] == [(137, 512), (512, 512), (512, 256)]
    assert dims["clause_mlp"] == [(384, 204), (204, 256)]
    assert dims["update_gate"] == [(521, 230), (230, 230), (230, 128)]
    assert dims["variables_output"] == [(128, 128), (128, 8)]
    i, o = C.c_int32(), C.c_int32()
    assert lib.dsat_layout_layer_dims(C.byref(lay), 9, 0, C.byref(i), C.byref(o)) == -1
    assert b"out of range" in lib.dsat_last_error()


def test_weight_blob_roundtrip(tmp_path):
    ws = W.glorot_weights(1, bias_scale=0.1)
    blob = W.to_blob(ws)
    assert blob.size == 856788
    back = W.from_blob(blob)
    for n in W.MLP_NAMES:
        for (a, b), (c, d) in zip(ws[n], back[n]):
            assert np.array_equal(a, c) and np.array_equal(b, d)
    W.save_npz(str(tmp_path / "w.npz"), ws)
    ws2 = W.load_npz(str(tmp_path / "w.npz"))
    assert all(np.array_equal(ws[n][i][0], ws2[n][i][0]) for n in W.MLP_NAMES for i in range(len(ws[n])))
    bad = dict(ws)
    bad["lit_mlp"] = ws["lit_mlp"][:2]
    with pytest.raises(ValueError):
        W.to_blob(bad)


def test_tf_checkpoint_roundtrip(tmp_path):
    from diffusionsat_amd import tf_checkpoint as T
    ws = W.glorot_weights(3, bias_scale=0.1)
    T.write_checkpoint(str(tmp_path), ws, step=167000)
    assert T.latest_checkpoint(str(tmp_path)).endswith("ckpt-167000")
    back = T.load_queriesat_weights(str(tmp_path))
    for n in W.MLP_NAMES:
        for (a, b), (c, d) in zip(ws[n], back[n]):
            assert np.array_equal(a, c) and np.array_equal(b, d)
    assert T.load_queriesat_weights(str(tmp_path / "nothing_here")) is None


def _graph_create(lib, rowptr, lits, vgid, cgid, n, c, g):
    arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (rowptr, lits, vgid, cgid)]
    h = C.c_void_p()
    rc = lib.dsat_graph_create(*[a.ctypes.data_as(C.c_void_p) for a in arrs], n, c, g, C.byref(h))
    return rc, lib.dsat_last_error().decode(), h


def test_graph_validation_happens_before_any_device_call(lib):
    ok = ([0, 2, 3], [1, -2, 2], [0, 0], [0, 0])
    cases = [
        (([1, 2, 3], ok[1], ok[2], ok[3], 2, 2, 1), "rowptr[0]"),
        (([0, 3, 2], ok[1], ok[2], ok[3], 2, 2, 1), "non-decreasing"),
        (([0, 2, 3], [1, -3, 2], ok[2], ok[3], 2, 2, 1), "literal out of range"),
        (([0, 2, 3], [1, 0, 2], ok[2], ok[3], 2, 2, 1), "literal out of range"),
        ((ok[0], ok[1], [0, 1], [0, 0], 2, 2, 2), "another graph"),
        ((ok[0], ok[1], [1, 0], ok[3], 2, 2, 2), "non-decreasing"),
        ((ok[0], ok[1], [0, 0], [0, 0], 2, 2, 2), "no variables"),
        ((ok[0], ok[1], ok[2], ok[3], 0, 2, 1), "n_vars>0"),
    ]
    for args, msg in cases:
        rc, err, _ = _graph_create(lib, *args)
        assert rc == -1 and msg in err, (rc, err, msg)


def test_weights_load_rejects_bad_blobs(lib):
    lay = _lib.default_layout()
    h = C.c_void_p()
    blob = np.zeros(10, dtype=np.float32)
    rc = lib.dsat_weights_load(blob.ctypes.data_as(C.c_void_p), 10, C.byref(lay), C.byref(h))
    assert rc == -1 and b"layout needs 856788" in lib.dsat_last_error()
    lay.feature_maps = 64                       # narrower widths are supported: the blob is just too short for them too
    rc = lib.dsat_weights_load(blob.ctypes.data_as(C.c_void_p), 10, C.byref(lay), C.byref(h))
    assert rc == -1 and b"layout needs" in lib.dsat_last_error()
    lay.feature_maps = 256                      # wider than the kernels' 128-wide rows
    rc = lib.dsat_weights_load(blob.ctypes.data_as(C.c_void_p), 10, C.byref(lay), C.byref(h))
    assert rc == -2 and b"128" in lib.dsat_last_error()


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_graph_size_guard_for_32bit_row_offsets(lib):
    """The fused gathers address rows with 32-bit float offsets: (2N+1)*128 must stay below 2^32.
    N = 2^24 is the first size that does not; the call is refused before any device work."""
    n = 1 << 24
    rc, err, _ = _graph_create(lib, [0], [], np.zeros(n, np.int32), [], n, 0, 1)
    assert rc == -1 and "32-bit row offsets" in err


def _graph_create_coo(lib, adj, vals, shape, cg, vg, g):
    adj = np.ascontiguousarray(adj, dtype=np.int64).reshape(-1, 2)
    cg = np.ascontiguousarray(cg, dtype=np.int64).reshape(-1, 2)
    vg = np.ascontiguousarray(vg, dtype=np.int64).reshape(-1, 2)
    vals = None if vals is None else np.ascontiguousarray(vals, dtype=np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None
    h = C.c_void_p()
    rc = lib.dsat_graph_create_coo(p(adj), p(vals), len(adj), shape[0], shape[1], p(cg), len(cg), p(vg), len(vg), g,
                                   C.byref(h))
    return rc, lib.dsat_last_error().decode()


def test_coo_graph_validation_happens_before_any_device_call(lib):
    """dsat_graph_create_coo takes the tensors of data/SatSpecifics.py:21-69; malformed ones are refused on the host."""
    adj = [[0, 0], [3, 0], [1, 1]]            # x1 | -x2 ; x2      (N = 2: rows 0,1 positive, 2,3 negative)
    cg, vg = [[0, 0], [0, 1]], [[0, 0], [0, 1]]
    cases = [
        ((adj, None, (3, 2), cg, vg, 1), -1, "2*n_vars rows"),
        (([[0, 0], [4, 0]], None, (4, 2), cg, vg, 1), -1, "adj_matrix index out of range"),
        (([[0, 2]], None, (4, 2), cg, vg, 1), -1, "adj_matrix index out of range"),
        ((adj, [1.0, 0.5, 1.0], (4, 2), cg, vg, 1), -2, "whole multiplicities"),
        ((adj, None, (4, 2), cg[:1], vg, 1), -1, "one entry per clause"),
        ((adj, None, (4, 2), cg, vg + [[0, 1]], 1), -1, "one entry per variable"),
        ((adj, None, (4, 2), [[0, 0], [0, 0]], vg, 1), -1, "two graphs"),
        ((adj, None, (4, 2), cg, [[0, 0], [1, 1]], 1), -1, "variables_graph index out of range"),
        ((adj, None, (4, 2), cg, [[1, 0], [0, 1]], 2), -1, "non-decreasing"),          # graphs must be contiguous
        ((adj, None, (4, 2), [[0, 0], [1, 1]], [[0, 0], [1, 1]], 2), -1, "another graph"),   # clause 0 uses x2 of graph 1
    ]
    for args, status, msg in cases:
        rc, err = _graph_create_coo(lib, *args)
        assert rc == status and msg in err, (rc, err, msg)


def test_layer_objects_validate_on_the_host(lib):
    h = C.c_void_p()
    dims = np.asarray([4, 3, 2], dtype=np.int32)
    blob = np.zeros(4 * 3 + 3 + 3 * 2 + 2, dtype=np.float32)
    assert lib.dsat_mlp_create(2, dims.ctypes.data_as(C.c_void_p), blob.ctypes.data_as(C.c_void_p), blob.size - 1,
                               C.byref(h)) == -1 and b"layers need 23" in lib.dsat_last_error()
    assert lib.dsat_mlp_create(0, dims.ctypes.data_as(C.c_void_p), blob.ctypes.data_as(C.c_void_p), blob.size,
                               C.byref(h)) == -1
    bad = np.asarray([4, 0, 2], dtype=np.int32)
    assert lib.dsat_mlp_create(2, bad.ctypes.data_as(C.c_void_p), blob.ctypes.data_as(C.c_void_p), blob.size,
                               C.byref(h)) == -2
    assert lib.dsat_pair_norm_workspace_floats(100, 128, 3) > 3 * 128
    assert lib.dsat_pair_norm_workspace_floats(100, 0, 3) == -1
    x = np.zeros(8, dtype=np.float32)
    seg = np.asarray([0, 3, 2], dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.dsat_pair_norm(vp(x), 2, 4, vp(seg), 2, None, 1e-6, 1, vp(x), vp(x), None) == -1
    assert b"seg must" in lib.dsat_last_error()


def test_sparse_tensors_follow_sat_specifics():
    """sparse.create_adj_matrices = data/SatSpecifics.py:21-69 (keys, entry order, shapes); as_coo reads every carrier."""
    import scipy.sparse as sp
    import torch
    from diffusionsat_amd import sparse
    from diffusionsat_amd.batch import make_batch
    from oracle import querysat_oracle as O
    insts = [(3, [[1, -2], [2, 3, -1], [-3]]), (2, [[1, 2], [-1, -1, 2]])]
    batch = make_batch(insts)
    d = sparse.create_adj_matrices(batch)
    assert {"adjacency_matrix_pos", "adjacency_matrix_neg", "adjacency_matrix", "clauses", "variables_in_graph",
            "normal_clauses", "clauses_graph_adj", "variables_graph_adj", "solutions"} == set(d)
    adj = d["adjacency_matrix"]
    # positives clause by clause, then negatives shifted by N = 5 (data/dimac.py:14-18, SatSpecifics.py:25-35)
    assert adj.indices.tolist() == [[0, 0], [1, 1], [2, 1], [3, 3], [4, 3], [4, 4],
                                    [6, 0], [5, 1], [7, 2], [8, 4], [8, 4]]
    assert adj.dense_shape == (10, 5) and np.all(adj.values == 1) and adj.indices.dtype == np.int64
    assert d["clauses_graph_adj"].indices.tolist() == [[0, 0], [0, 1], [0, 2], [1, 3], [1, 4]]
    assert d["variables_graph_adj"].indices.tolist() == [[0, 0], [0, 1], [0, 2], [1, 3], [1, 4]]
    ref = O.create_adj_matrices(insts)
    dense = np.zeros(adj.dense_shape)
    np.add.at(dense, (adj.indices[:, 0], adj.indices[:, 1]), adj.values)
    assert np.array_equal(dense, ref["adjacency_matrix"].toarray())
    for carrier in (ref["adjacency_matrix"], sp.coo_matrix(dense),
                    torch.sparse_coo_tensor(torch.from_numpy(adj.indices.T.copy()), torch.from_numpy(adj.values), adj.dense_shape),
                    (adj.indices, adj.values, adj.dense_shape)):
        got = sparse.as_coo(carrier)
        back = np.zeros(got.dense_shape)
        np.add.at(back, (got.indices[:, 0], got.indices[:, 1]), got.values)
        assert np.array_equal(back, dense) and got.indices.dtype == np.int64 and got.values.dtype == np.float32
    pos, neg = sparse.compute_adj_indices([[1, -2], [2, 3, -1], [-3]])
    rp, rn = O.compute_adj_indices([[1, -2], [2, 3, -1], [-3]])
    assert pos.tolist() == rp and neg.tolist() == rn
    with pytest.raises(TypeError):
        sparse.as_coo(np.zeros((2, 2)))


def test_foreign_model_and_layer_options_raise():
    from diffusionsat_amd.query_sat import MLP, PairNorm, QuerySAT
    with pytest.raises(NotImplementedError, match="use_linear_loss"):
        QuerySAT(optimizer=None, supervised=False, use_linear_loss=True)
    with pytest.raises(NotImplementedError, match="Optuna"):
        QuerySAT(optimizer=None, trial=object())
    assert QuerySAT(optimizer=None, supervised=False).supervised is False        # built: tests/test_gpu_boundary.py
    with pytest.raises(NotImplementedError):
        MLP(3, 16, 4, out_activation=lambda x: x)
    m = MLP(3, 16, 4, name="m")
    assert [d.units for d in m.dense_layers] == [16, 16, 4] and not m.built
    m.build(7)
    assert [a.shape for a in m.get_weights()] == [(7, 16), (16,), (16, 16), (16,), (16, 4), (4,)]
    with pytest.raises(ValueError):
        m.set_weights(m.get_weights()[:-1])
    p = PairNorm()
    assert p.epsilon == 1e-6 and p.subtract_mean is False           # the reference's defaults, normalization.py:31
