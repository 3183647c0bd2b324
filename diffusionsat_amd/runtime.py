"""Python handles over the C-ABI objects (graph, weights, session).

PyTorch is used only for device memory and streams: every tensor handed to the library
is a contiguous CUDA(HIP) tensor whose ``data_ptr()`` crosses the boundary.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from .batch import Batch
from .weights import WeightSet, to_blob


def _ptr(t: Optional[torch.Tensor], dtype=None, numel=None, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU" % name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if numel is not None and t.numel() != numel:
        raise ValueError("%s has %d elements, expected %d" % (name, t.numel(), numel))
    return C.c_void_p(t.data_ptr())


def _stream(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _device(device) -> torch.device:
    """``cuda:i`` with an explicit index (the current device when none is given)."""
    d = torch.device("cuda" if device is None else device)
    if d.type != "cuda":
        raise ValueError("diffusionsat_amd runs on the GPU only (got device %r)" % (device,))
    return torch.device("cuda", torch.cuda.current_device() if d.index is None else d.index)


def _on_device(fn):
    """Run a method with the object's device current: the C library binds every object to the HIP
    device that was current at its creation and refuses calls made under another one."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        with torch.cuda.device(self.device):
            return fn(self, *a, **kw)
    return wrapper


class Arena:
    """Device + pinned-host bump arena (``dsat_arena``): graphs and sessions created in it cost no
    hipMalloc / hipFree and upload asynchronously on the current stream (include/dsat.h, "pooled
    objects").  ``reset()`` gives everything back once the objects created in it are closed."""

    def __init__(self, device_bytes: int, pinned_bytes: int, device=None):
        lib = _lib.load()
        self.device = _device(device)
        self.device_bytes, self.pinned_bytes = int(device_bytes), int(pinned_bytes)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.dsat_arena_create(self.device_bytes, self.pinned_bytes, C.byref(h)))
        self.handle = h

    def reset(self):
        _lib.check(_lib.load().dsat_arena_reset(self.handle))

    def usage(self):
        v = [C.c_int64(0) for _ in range(4)]
        _lib.check(_lib.load().dsat_arena_usage(self.handle, *[C.byref(x) for x in v]))
        return {"device_used": v[0].value, "pinned_used": v[1].value, "device_needed": v[2].value,
                "pinned_needed": v[3].value}

    def close(self):
        """Frees the arena; raises while graphs / sessions created in it are still open (their arrays live here)."""
        if getattr(self, "handle", None):
            with torch.cuda.device(self.device):
                _lib.check(_lib.load().dsat_arena_reset(self.handle))      # DSAT_E_STATE: objects still alive
                _lib.load().dsat_arena_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Graph:
    """Device-resident CSR clause-literal graph of one batch (``dsat_graph``).  With ``arena`` the
    device arrays come from that arena and the upload is asynchronous on the current stream."""

    def __init__(self, batch: Batch, device=None, arena: Optional[Arena] = None):
        lib = _lib.load()
        self.batch = batch
        self.arena = arena
        self.device = arena.device if (arena is not None and device is None) else _device(device)
        self._keep = [np.ascontiguousarray(a, dtype=np.int32) for a in
                      (batch.clause_rowptr, batch.clause_lits, batch.var_graph_id, batch.clause_graph_id)]
        h = C.c_void_p()
        args = [a.ctypes.data_as(C.c_void_p) if a.size else None for a in self._keep]
        with torch.cuda.device(self.device):
            if arena is None:
                _lib.check(lib.dsat_graph_create(args[0], args[1], args[2], args[3], batch.n_vars, batch.n_clauses,
                                                 batch.n_graphs, C.byref(h)))
            else:
                _lib.check(lib.dsat_graph_create_in(arena.handle, _stream(self.device), args[0], args[1], args[2],
                                                    args[3], batch.n_vars, batch.n_clauses, batch.n_graphs,
                                                    C.byref(h)))
        self.handle = h
        self.n_vars, self.n_clauses, self.n_graphs, self.n_edges = (
            batch.n_vars, batch.n_clauses, batch.n_graphs, batch.n_edges)

    @classmethod
    def from_coo(cls, adj_matrix, clauses_graph, variables_graph, device=None, arena: Optional[Arena] = None):
        """The device graph from the three sparse tensors ``QuerySAT.diffusion_step`` receives in the
        reference (model/query_sat.py:467-481; built by data/SatSpecifics.py:21-69), through
        ``dsat_graph_create_coo``.  Anything ``sparse.as_coo`` understands is accepted."""
        from . import sparse
        lib = _lib.load()
        adj, cg, vg = sparse.as_coo(adj_matrix), sparse.as_coo(clauses_graph), sparse.as_coo(variables_graph)
        if cg.dense_shape[0] != vg.dense_shape[0]:
            raise ValueError("clauses_graph has %d graphs, variables_graph %d" % (cg.dense_shape[0], vg.dense_shape[0]))
        if adj.dense_shape != (2 * vg.dense_shape[1], cg.dense_shape[1]):
            raise ValueError("adj_matrix is %s; the memberships describe %d variables and %d clauses"
                             % (adj.dense_shape, vg.dense_shape[1], cg.dense_shape[1]))
        self = cls.__new__(cls)
        self.arena = arena
        self.device = arena.device if (arena is not None and device is None) else _device(device)
        values = None if np.all(adj.values == 1.0) else adj.values
        self._keep = [adj.indices, values, cg.indices, vg.indices]
        p = lambda a: a.ctypes.data_as(C.c_void_p) if (a is not None and a.size) else None
        h = C.c_void_p()
        args = (p(adj.indices), p(values), adj.indices.shape[0], adj.dense_shape[0], adj.dense_shape[1],
                p(cg.indices), cg.indices.shape[0], p(vg.indices), vg.indices.shape[0], cg.dense_shape[0], C.byref(h))
        with torch.cuda.device(self.device):
            if arena is None:
                _lib.check(lib.dsat_graph_create_coo(*args))
            else:
                _lib.check(lib.dsat_graph_create_coo_in(arena.handle, _stream(self.device), *args))
        self.handle = h
        nv, nc, ne, ng = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32()
        _lib.check(lib.dsat_graph_info(h, C.byref(nv), C.byref(nc), C.byref(ne), C.byref(ng)))
        self.n_vars, self.n_clauses, self.n_graphs, self.n_edges = nv.value, nc.value, ng.value, ne.value
        self.batch = None
        return self

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().dsat_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Weights:
    """Packed device copy of a weight set (``dsat_weights``)."""

    def __init__(self, ws_or_blob, layout=None, device=None):
        lib = _lib.load()
        self.layout = layout or _lib.default_layout()
        if device is None and isinstance(ws_or_blob, torch.Tensor) and ws_or_blob.is_cuda:
            device = ws_or_blob.device
        self.device = _device(device)
        if isinstance(ws_or_blob, dict):
            blob = to_blob(ws_or_blob, self.layout)
        elif isinstance(ws_or_blob, torch.Tensor):
            blob = ws_or_blob
        else:
            blob = np.ascontiguousarray(ws_or_blob, dtype=np.float32)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            if isinstance(blob, torch.Tensor):
                if blob.dtype != torch.float32 or not blob.is_contiguous():
                    raise TypeError("weight blob tensor must be contiguous float32")
                _lib.check(lib.dsat_weights_load(C.c_void_p(blob.data_ptr()), blob.numel(), C.byref(self.layout),
                                                 C.byref(h)))
            else:
                _lib.check(lib.dsat_weights_load(blob.ctypes.data_as(C.c_void_p), blob.size, C.byref(self.layout),
                                                 C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().dsat_weights_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Session:
    """Workspace + launch sequence for one (graph, weights) pair (``dsat_session``)."""

    def __init__(self, graph: Graph, weights: Weights, max_rounds: int = 32, device=None,
                 arena: Optional[Arena] = None):
        lib = _lib.load()
        self.graph, self.weights = graph, weights
        self.arena = arena
        self.device = graph.device if device is None else _device(device)
        if not (self.device == graph.device == weights.device):
            raise ValueError("session on %s, graph on %s, weights on %s: all three must share one GPU"
                             % (self.device, graph.device, weights.device))
        self.max_rounds = max_rounds
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            if arena is None:
                _lib.check(lib.dsat_session_create(graph.handle, weights.handle, max_rounds, C.byref(h)))
            else:
                _lib.check(lib.dsat_session_create_in(arena.handle, _stream(self.device), graph.handle,
                                                      weights.handle, max_rounds, C.byref(h)))
        self.handle = h

    # -- the hot path ----------------------------------------------------------------
    TAP_FIELDS = (("query", "n"), ("literals", "2n"), ("clauses_loss4", "c"), ("clause_messages", "c"),
                  ("variables_loss_all", "c"), ("new_clause_value", "c"), ("variables_grad", "n"),
                  ("variables_loss", "2n"), ("new_variables", "n"), ("variables", "n"), ("clause_state", "c"))

    @_on_device
    def diffusion_step(self, noise_scale: float, noisy_num: torch.Tensor, normal=None, labels=None,
                       seed: int = 0, step_index: int = 0, rounds: int = 32, early_exit: bool = True,
                       want_logits: bool = False, want_round_logits: bool = False, denoised=None,
                       tap_round: Optional[int] = None):
        """``tap_round``: also return (key "tap") the intermediates of that message-passing round as the
        running kernels hold them (``dsat_session_set_tap``; the step then runs eagerly) -- the parity
        tests compare them with the oracle's trace."""
        lib = _lib.load()
        n, g = self.graph.n_vars, self.graph.n_graphs
        dev = self.device
        tap = None
        if tap_round is not None:
            rows = {"n": n, "2n": 2 * n, "c": self.graph.n_clauses}
            tap = {k: torch.zeros(rows[r], 128, dtype=torch.float32, device=dev) for k, r in self.TAP_FIELDS}
            if tap_round >= rounds - 1:
                tap.pop("clause_state")      # produced while the next round runs (include/dsat.h)
            t = _lib.Tap(round=int(tap_round), **{k: v.data_ptr() for k, v in tap.items()})
            _lib.check(lib.dsat_session_set_tap(self.handle, C.byref(t)))
        try:
            out = self._diffusion_step(lib, n, g, dev, noise_scale, noisy_num, normal, labels, seed, step_index,
                                       rounds, early_exit, want_logits, want_round_logits, denoised)
        finally:
            if tap is not None:
                _lib.check(lib.dsat_session_set_tap(self.handle, None))
        out["tap"] = tap
        return out

    def _diffusion_step(self, lib, n, g, dev, noise_scale, noisy_num, normal, labels, seed, step_index, rounds,
                        early_exit, want_logits, want_round_logits, denoised):
        pred = torch.empty(n, dtype=torch.float32, device=dev)
        logits = torch.empty(n, 8, dtype=torch.float32, device=dev) if want_logits else None
        rlog = torch.zeros(rounds, n, 8, dtype=torch.float32, device=dev) if want_round_logits else None
        best = torch.empty(g, dtype=torch.int32, device=dev)
        steps = torch.empty(1, dtype=torch.int32, device=dev)
        common = (_ptr(normal, torch.float32, rounds * n * 4, "normal"), _ptr(labels, torch.int32, n, "labels"),
                  int(seed), int(step_index), int(rounds), int(bool(early_exit)),
                  _ptr(logits), _ptr(rlog), _ptr(pred), _ptr(best), _ptr(steps), _stream(self.device))
        if denoised is None:
            _lib.check(lib.dsat_diffusion_step(
                self.handle, float(noise_scale), _ptr(noisy_num, torch.float32, 2 * n, "noisy_num"), *common))
        else:
            _lib.check(lib.dsat_diffusion_step_self(
                self.handle, float(noise_scale), _ptr(noisy_num, torch.float32, 2 * n, "noisy_num"),
                _ptr(denoised, torch.float32, n, "denoised_num"), *common))
        return {"prediction": pred, "logits": logits, "round_logits": rlog, "best_map": best,
                "steps_taken": steps}

    @_on_device
    def sample(self, n_steps: int = 32, rounds: int = 32, seed: int = 0, early_exit: bool = True,
               uniform=None, normal=None, labels=None, want_predictions: bool = False,
               want_solved_fraction: bool = True, self_conditioning: bool = False):
        lib = _lib.load()
        n, g = self.graph.n_vars, self.graph.n_graphs
        dev = self.device
        assign = torch.empty(n, dtype=torch.uint8, device=dev)
        solved = torch.empty(g, dtype=torch.uint8, device=dev)
        preds = torch.empty(n_steps, n, dtype=torch.float32, device=dev) if want_predictions else None
        frac = C.c_float(-1.0)
        fn = lib.dsat_sample_self if self_conditioning else lib.dsat_sample
        _lib.check(fn(
            self.handle, int(n_steps), int(rounds), int(seed), int(bool(early_exit)),
            _ptr(uniform, torch.float32, n_steps * n, "uniform"),
            _ptr(normal, torch.float32, n_steps * rounds * n * 4, "normal"),
            _ptr(labels, torch.int32, n_steps * n, "labels"),
            _ptr(assign), _ptr(solved), _ptr(preds),
            C.byref(frac) if want_solved_fraction else None, _stream(self.device)))
        return {"assignment": assign, "solved": solved, "predictions": preds,
                "solved_fraction": frac.value if want_solved_fraction else None}

    # -- single operators ----------------------------------------------------------------
    @_on_device
    def bench_mlp(self, mlp: int, rows: int, impl: int = 0, rb: int = 1, nwaves: int = 4, iters: int = 20) -> float:
        us = C.c_double(0.0)
        _lib.check(_lib.load().dsat_bench_mlp(self.handle, mlp, rows, impl, rb, nwaves, iters, C.byref(us), _stream(self.device)))
        return us.value

    @_on_device
    def op_mlp(self, mlp: int, x: torch.Tensor, impl: int = 0, rb: int = 1, nwaves: int = 4) -> torch.Tensor:
        lib = _lib.load()
        import ctypes
        i, o = ctypes.c_int32(), ctypes.c_int32()
        nl = self.weights.layout.n_layers[mlp]
        _lib.check(lib.dsat_layout_layer_dims(C.byref(self.weights.layout), mlp, 0, C.byref(i), C.byref(o)))
        in_dim = i.value
        _lib.check(lib.dsat_layout_layer_dims(C.byref(self.weights.layout), mlp, nl - 1, C.byref(i), C.byref(o)))
        out_dim = o.value
        rows = x.shape[0]
        y = torch.empty(rows, out_dim, dtype=torch.float32, device=self.device)
        _lib.check(lib.dsat_op_mlp_ex(self.handle, mlp, _ptr(x, torch.float32, rows * in_dim, "x"), rows, _ptr(y),
                                      impl, rb, nwaves, _stream(self.device)))
        return y

    @_on_device
    def op_clause_gather(self, q, lits):
        lib = _lib.load()
        n, c = self.graph.n_vars, self.graph.n_clauses
        closs = torch.empty(c, 128, dtype=torch.float32, device=self.device)
        cmsg = torch.empty(c, 128, dtype=torch.float32, device=self.device)
        _lib.check(lib.dsat_op_clause_gather(self.handle, _ptr(q, torch.float32, n * 128, "q"),
                                             _ptr(lits, torch.float32, 2 * n * 128, "lits"),
                                             _ptr(closs), _ptr(cmsg), _stream(self.device)))
        return closs, cmsg

    @_on_device
    def op_literal_gather(self, q, closs, vla):
        lib = _lib.load()
        n, c = self.graph.n_vars, self.graph.n_clauses
        grad = torch.empty(n, 128, dtype=torch.float32, device=self.device)
        vloss = torch.empty(2 * n, 128, dtype=torch.float32, device=self.device)
        _lib.check(lib.dsat_op_literal_gather(self.handle, _ptr(q, torch.float32, n * 128, "q"),
                                              _ptr(closs, torch.float32, c * 128, "closs"),
                                              _ptr(vla, torch.float32, c * 128, "vla"),
                                              _ptr(grad), _ptr(vloss), _stream(self.device)))
        return grad, vloss

    @_on_device
    def op_pair_norm(self, side: int, x):
        lib = _lib.load()
        rows = self.graph.n_vars if side == 0 else self.graph.n_clauses
        y = torch.empty(rows, 128, dtype=torch.float32, device=self.device)
        _lib.check(lib.dsat_op_pair_norm(self.handle, side, _ptr(x, torch.float32, rows * 128, "x"), _ptr(y),
                                         _stream(self.device)))
        return y

    @_on_device
    def op_is_batch_sat(self, logits) -> int:
        lib = _lib.load()
        out = C.c_int32(-1)
        _lib.check(lib.dsat_op_is_batch_sat(self.handle, _ptr(logits, torch.float32, self.graph.n_vars, "logits"),
                                            C.byref(out), _stream(self.device)))
        return out.value

    @_on_device
    def op_is_graph_sat(self, logits) -> torch.Tensor:
        """``is_graph_sat`` (utils/sat.py:165-180): uint8 [G], 1 = the graph's clauses are all satisfied."""
        out = torch.empty(self.graph.n_graphs, dtype=torch.uint8, device=self.device)
        _lib.check(_lib.load().dsat_op_is_graph_sat(self.handle, _ptr(logits, torch.float32, self.graph.n_vars, "logits"),
                                                    _ptr(out), _stream(self.device)))
        return out

    def packed_offsets(self) -> np.ndarray:
        """Byte offsets [G+1] of the graphs inside a packed assignment (``dsat_session_packed_bytes``)."""
        off = np.zeros(self.graph.n_graphs + 1, dtype=np.int32)
        _lib.load().dsat_session_packed_bytes(self.handle, off.ctypes.data_as(C.c_void_p))
        return off

    @_on_device
    def pack_assignment(self, assignment: torch.Tensor) -> torch.Tensor:
        """uint8 [N] assignment -> little-endian packed bits per graph (bit i <-> x_{i+1}), on the device."""
        lib = _lib.load()
        total = int(lib.dsat_session_packed_bytes(self.handle, None))
        out = torch.empty(total, dtype=torch.uint8, device=self.device)
        _lib.check(lib.dsat_op_pack_assignment(self.handle, _ptr(assignment, torch.uint8, self.graph.n_vars, "assignment"),
                                               _ptr(out), _stream(self.device)))
        return out

    @_on_device
    def op_rng(self, seed: int, step_index: int, rounds: int):
        lib = _lib.load()
        n = self.graph.n_vars
        normal = torch.empty(rounds, n, 4, dtype=torch.float32, device=self.device)
        labels = torch.empty(n, dtype=torch.int32, device=self.device)
        uniform = torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(lib.dsat_op_rng(self.handle, int(seed), int(step_index), int(rounds), _ptr(normal), _ptr(labels),
                                   _ptr(uniform), _stream(self.device)))
        return normal, labels, uniform

    # -- knobs ------------------------------------------------------------------------
    def set_graph_replay(self, enable: bool):
        _lib.check(_lib.load().dsat_session_set_graph_replay(self.handle, int(enable)))

    def set_supervised(self, supervised: bool):
        """``QuerySAT(supervised=...)``: True = logit map by the KL to the labels, False = by the graph's own
        unsupervised clause loss (model/query_sat.py:289-309; ``dsat_session_set_supervised``)."""
        _lib.check(_lib.load().dsat_session_set_supervised(self.handle, int(bool(supervised))))

    def set_retire(self, enable: bool):
        """Drop solved copies from a batch of identical copies between denoising steps (``dsat_session_set_retire``)."""
        _lib.check(_lib.load().dsat_session_set_retire(self.handle, int(enable)))

    def split_mask(self) -> int:
        """bit 0: the clause tile, bit 1: the variable-update tile run their MLP products as bf16 slices on
        the bf16 matrix pipe (``dsat_session_split_mask``)."""
        return int(_lib.load().dsat_session_split_mask(self.handle))

    def set_throughput(self, enable: bool):
        """Several batches in flight on this GPU: prefer CU-cycles to one batch's latency (``dsat_session_set_throughput``)."""
        _lib.check(_lib.load().dsat_session_set_throughput(self.handle, int(enable)))

    def set_profile(self, enable: bool):
        _lib.check(_lib.load().dsat_session_set_profile(self.handle, int(enable)))

    def profile(self):
        lib = _lib.load()
        cap = 32
        names = (C.c_char_p * cap)()
        us = (C.c_double * cap)()
        cnt = (C.c_int64 * cap)()
        n = lib.dsat_session_profile(self.handle, cap, names, us, cnt)
        return {names[i].decode(): {"avg_us": us[i], "launches": cnt[i]} for i in range(n)}

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().dsat_session_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
