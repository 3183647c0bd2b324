"""Weight containers for the five QuerySAT MLPs (host side, NumPy).

A weight set is ``{mlp_name: [(kernel[in,out], bias[out]), ...]}`` with Keras ``Dense``
layouts (model/mlp.py:22-39), MLP names as the attributes of the reference model
(model/query_sat.py:117-122).  ``to_blob`` flattens it in the order the C-ABI expects
(include/dsat.h, ``dsat_weights_load``).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import _lib

MLP_NAMES = _lib.MLP_NAMES
WeightSet = Dict[str, List[Tuple[np.ndarray, np.ndarray]]]

# checkpoint attribute name of each MLP (model/query_sat.py:117-122); lit_mlp and
# clause_mlp are the attribute names, "lit_query"/"clause_update" the Keras layer names
CHECKPOINT_ATTR = {"variables_query": "variables_query", "lit_mlp": "lit_mlp", "clause_mlp": "clause_mlp",
                   "update_gate": "update_gate", "variables_output": "variables_output"}


def layer_dims(layout=None):
    """[(mlp_name, [(in, out), ...]), ...] for a layout (default: the reference defaults)."""
    import ctypes as C
    lib = _lib.load()
    layout = layout or _lib.default_layout()
    out = []
    for m, name in enumerate(MLP_NAMES):
        dims = []
        for l in range(layout.n_layers[m]):
            i, o = C.c_int32(), C.c_int32()
            _lib.check(lib.dsat_layout_layer_dims(C.byref(layout), m, l, C.byref(i), C.byref(o)))
            dims.append((i.value, o.value))
        out.append((name, dims))
    return out


def glorot_weights(seed: int = 0, bias_scale: float = 0.0, layout=None) -> WeightSet:
    """Keras defaults: glorot_uniform kernels, zero biases (model/mlp.py:24,37-39).
    ``bias_scale`` > 0 draws normal biases (to exercise the bias path)."""
    rng = np.random.default_rng(seed)
    ws: WeightSet = {}
    for name, dims in layer_dims(layout):
        layers = []
        for fi, fo in dims:
            lim = np.sqrt(6.0 / (fi + fo))
            w = rng.uniform(-lim, lim, size=(fi, fo)).astype(np.float32)
            b = (rng.standard_normal(fo) * bias_scale).astype(np.float32)
            layers.append((w, b))
        ws[name] = layers
    return ws


def to_blob(ws: WeightSet, layout=None) -> np.ndarray:
    parts = []
    for name, dims in layer_dims(layout):
        layers = ws[name]
        if len(layers) != len(dims):
            raise ValueError("%s: %d layers given, layout has %d" % (name, len(layers), len(dims)))
        for (w, b), (fi, fo) in zip(layers, dims):
            w = np.asarray(w, dtype=np.float32)
            b = np.asarray(b, dtype=np.float32)
            if w.shape != (fi, fo) or b.shape != (fo,):
                raise ValueError("%s: kernel %s / bias %s, expected (%d,%d) / (%d,)"
                                 % (name, w.shape, b.shape, fi, fo, fo))
            parts.append(np.ascontiguousarray(w).ravel())
            parts.append(np.ascontiguousarray(b).ravel())
    return np.concatenate(parts)


def from_blob(blob: np.ndarray, layout=None) -> WeightSet:
    ws: WeightSet = {}
    off = 0
    for name, dims in layer_dims(layout):
        layers = []
        for fi, fo in dims:
            w = blob[off:off + fi * fo].reshape(fi, fo).copy(); off += fi * fo
            b = blob[off:off + fo].copy(); off += fo
            layers.append((w, b))
        ws[name] = layers
    if off != blob.size:
        raise ValueError("blob has %d floats, layout needs %d" % (blob.size, off))
    return ws


def save_npz(path: str, ws: WeightSet):
    """Documented portable format: ``<mlp>/dense_layers/<i>/{kernel,bias}`` arrays --
    the variable paths of the reference's TF2 checkpoints
    (``model/<mlp>/dense_layers/<i>/{kernel,bias}``) without the object-graph suffix."""
    arrs = {}
    for name, layers in ws.items():
        for i, (w, b) in enumerate(layers):
            arrs["%s/dense_layers/%d/kernel" % (name, i)] = np.asarray(w, dtype=np.float32)
            arrs["%s/dense_layers/%d/bias" % (name, i)] = np.asarray(b, dtype=np.float32)
    np.savez(path, **arrs)


def load_npz(path: str) -> WeightSet:
    data = np.load(path, allow_pickle=False)
    ws: WeightSet = {}
    for name in MLP_NAMES:
        layers = []
        i = 0
        while "%s/dense_layers/%d/kernel" % (name, i) in data.files:
            layers.append((data["%s/dense_layers/%d/kernel" % (name, i)],
                           data["%s/dense_layers/%d/bias" % (name, i)]))
            i += 1
        if not layers:
            raise ValueError("%s: no layers for MLP %s" % (path, name))
        ws[name] = layers
    return ws
