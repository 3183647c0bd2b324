"""The sparse tensors ``QuerySAT.diffusion_step`` is called with in the reference.

``model/query_sat.py:467-481`` declares three ``tf.SparseTensorSpec`` inputs (``adj_matrix`` [2N x C],
``clauses_graph`` [G x C], ``variables_graph`` [G x N]); ``data/SatSpecifics.py:21-69`` builds them from a
batch.  ``SparseTensor`` here is the same triple (``indices`` int64 [nnz, 2], ``values`` float32 [nnz],
``dense_shape``) as plain NumPy; ``as_coo`` also takes a ``tf.SparseTensor`` (duck-typed: TensorFlow is not a
dependency), a ``torch`` sparse COO tensor, a SciPy sparse matrix or an ``(indices, values, dense_shape)`` tuple,
so the reference's own ``step_data`` entries can be handed to this package's ``diffusion_step`` unchanged.
"""
from __future__ import annotations

from typing import NamedTuple, Sequence

import numpy as np

from .batch import Batch, make_batch


class SparseTensor(NamedTuple):
    indices: np.ndarray       # int64 [nnz, 2]
    values: np.ndarray        # float32 [nnz]
    dense_shape: tuple        # (rows, cols)


def _np(x):
    """numpy view of a TF / torch / numpy array-like."""
    if hasattr(x, "numpy") and not isinstance(x, np.ndarray):
        x = x.detach().cpu().numpy() if hasattr(x, "detach") else x.numpy()
    return np.asarray(x)


def as_coo(t) -> SparseTensor:
    """``indices, values, dense_shape`` of whatever sparse object the caller holds."""
    if isinstance(t, SparseTensor):
        return t
    if isinstance(t, tuple) and len(t) == 3:
        idx, val, shape = t
    elif hasattr(t, "indices") and hasattr(t, "dense_shape"):              # tf.SparseTensor
        idx, val, shape = t.indices, t.values, t.dense_shape
    elif hasattr(t, "is_sparse") and getattr(t, "is_sparse"):              # torch sparse COO ([2, nnz] indices)
        idx, val, shape = _np(t._indices()).T, t._values(), tuple(t.shape)
    elif hasattr(t, "tocoo"):                                              # scipy.sparse
        m = t.tocoo()
        idx, val, shape = np.stack([m.row, m.col], axis=1), m.data, m.shape
    else:
        raise TypeError("not a sparse tensor (tf.SparseTensor-like, torch sparse COO, scipy.sparse, or "
                        "(indices, values, dense_shape)): %r" % type(t))
    idx = np.ascontiguousarray(_np(idx), dtype=np.int64).reshape(-1, 2)
    val = np.ascontiguousarray(_np(val), dtype=np.float32).reshape(-1)
    shape = tuple(int(s) for s in _np(shape).reshape(-1))
    if len(shape) != 2 or val.shape[0] != idx.shape[0]:
        raise ValueError("sparse tensor must be 2-D with one value per index pair")
    return SparseTensor(idx, val, shape)


def compute_adj_indices(clauses: Sequence[Sequence[int]]):
    """data/dimac.py:14-18: [variable, clause] pairs of the positive and of the negative literals."""
    lits = np.fromiter((l for c in clauses for l in c), dtype=np.int64)
    cl = np.repeat(np.arange(len(clauses), dtype=np.int64), [len(c) for c in clauses])
    pos, neg = lits > 0, lits < 0
    return (np.stack([lits[pos] - 1, cl[pos]], axis=1), np.stack([-lits[neg] - 1, cl[neg]], axis=1))


def create_adj_matrices(batch) -> dict:
    """``SatSpecifics.create_adj_matrices`` (data/SatSpecifics.py:21-69) for a ``Batch`` (or a list of
    ``(n_vars, clauses)`` instances): the same dictionary keys, with ``SparseTensor`` triples.  The
    adjacency is ``concat([positive entries, negative entries + N])`` (:25-35), memberships are
    ``(graph, index)`` per clause / variable (:40-57), all values 1 (:76-80)."""
    if not isinstance(batch, Batch):
        batch = make_batch(list(batch))
    n, c, g = batch.n_vars, batch.n_clauses, batch.n_graphs
    lits = batch.clause_lits.astype(np.int64)
    cl = np.repeat(np.arange(c, dtype=np.int64), np.diff(batch.clause_rowptr))
    pos, neg = lits > 0, lits < 0
    idx_pos = np.stack([lits[pos] - 1, cl[pos]], axis=1)
    idx_neg = np.stack([-lits[neg] - 1, cl[neg]], axis=1)
    ones = lambda k: np.ones(k, dtype=np.float32)
    adj_idx = np.concatenate([idx_pos, idx_neg + np.array([[n, 0]], dtype=np.int64)], axis=0)
    cg = np.stack([batch.clause_graph_id.astype(np.int64), np.arange(c, dtype=np.int64)], axis=1)
    vg = np.stack([batch.var_graph_id.astype(np.int64), np.arange(n, dtype=np.int64)], axis=1)
    return {
        "adjacency_matrix_pos": SparseTensor(idx_pos, ones(len(idx_pos)), (n, c)),
        "adjacency_matrix_neg": SparseTensor(idx_neg, ones(len(idx_neg)), (n, c)),
        "adjacency_matrix": SparseTensor(adj_idx, ones(len(adj_idx)), (2 * n, c)),
        "clauses": batch.clauses() if hasattr(batch, "clauses") else None,
        "variables_in_graph": batch.variables_in_graph,
        "normal_clauses": [inst[1] for inst in batch.instances] if batch.instances else None,
        "clauses_graph_adj": SparseTensor(cg, ones(c), (g, c)),
        "variables_graph_adj": SparseTensor(vg, ones(n), (g, n)),
        "solutions": None,     # labels are unused on the sampling path (model/query_sat.py:476: labels=None)
    }
