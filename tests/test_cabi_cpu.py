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
