"""Host-side mirror of the reference's model interface for the sampling path.

Same names, argument meaning and return dictionaries as ``model/query_sat.py``,
``model/mlp.py`` and ``layers/normalization.py`` of the reference, so code written against
``QuerySAT.diffusion_step`` / ``MLP`` / ``PairNorm`` reads the same; the arithmetic runs in
the HIP extension (``include/dsat.h``).  There is no CPU fallback: every call goes through
the C-ABI and raises if the extension or the GPU is missing.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

import ctypes as C

from . import _lib, runtime, sparse
from .batch import Batch
from .weights import WeightSet, glorot_weights, load_npz

MLP_INDEX = {"variables_query": 0, "lit_mlp": 1, "clause_mlp": 2, "update_gate": 3, "variables_output": 4,
             # Keras layer names used by the reference constructor (model/query_sat.py:117-122)
             "lit_query": 1, "clause_update": 2}


def leaky_relu(x, alpha=0.2):
    """``tf.nn.leaky_relu`` (default alpha 0.2): the hidden activation of every MLP of the model
    (model/mlp.py:14).  The kernels implement exactly this one; it is also the marker ``MLP`` compares its
    ``activation`` argument with."""
    return torch.where(x > 0, x, alpha * x)


class GraphInputs:
    """The per-batch inputs ``diffusion_step`` receives in the reference (``adj_matrix``,
    ``clauses_graph``, ``variables_graph`` built by data/SatSpecifics.py:21-69), held as the device CSR
    graph.  Built from a ``Batch`` (clause lists) or, ``from_coo``, from the three sparse tensors themselves."""

    def __init__(self, batch: Batch, device=None):
        self.batch = batch
        self.graph = runtime.Graph(batch, device=device)

    @classmethod
    def from_coo(cls, adj_matrix, clauses_graph, variables_graph, device=None) -> "GraphInputs":
        """``adj_matrix`` [2N x C], ``clauses_graph`` [G x C], ``variables_graph`` [G x N] as
        ``model/query_sat.py:467-481`` declares them; the device graph comes from ``dsat_graph_create_coo``
        and is, for the tensors of ``SatSpecifics.create_adj_matrices``, bit for bit the graph of the
        ``Batch`` route."""
        adj, cg, vg = sparse.as_coo(adj_matrix), sparse.as_coo(clauses_graph), sparse.as_coo(variables_graph)
        self = cls.__new__(cls)
        self.graph = runtime.Graph.from_coo(adj, cg, vg, device=device)
        self.batch = _batch_from_coo(adj, cg, vg)
        self.graph.batch = self.batch
        return self

    @property
    def n_vars(self):
        return self.graph.n_vars

    @property
    def n_graphs(self):
        return self.graph.n_graphs


def _batch_from_coo(adj, cg, vg) -> Batch:
    """Host bookkeeping copy (graph ids, sizes) of a batch given as sparse tensors; the device graph does
    not come from this (``dsat_graph_create_coo`` reads the tensors itself)."""
    n, c, g = vg.dense_shape[1], cg.dense_shape[1], cg.dense_shape[0]
    rep = np.maximum(adj.values.astype(np.int64), 1)
    lit_row, clause = np.repeat(adj.indices[:, 0], rep), np.repeat(adj.indices[:, 1], rep)
    order = np.argsort(clause, kind="stable")
    lits = np.where(lit_row < n, lit_row + 1, -(lit_row - n + 1))[order].astype(np.int32)
    rowptr = np.zeros(c + 1, dtype=np.int32)
    np.cumsum(np.bincount(clause, minlength=c), out=rowptr[1:])
    vgid = np.empty(n, dtype=np.int32)
    vgid[vg.indices[:, 1]] = vg.indices[:, 0]
    cgid = np.empty(c, dtype=np.int32)
    cgid[cg.indices[:, 1]] = cg.indices[:, 0]
    return Batch(n_vars=n, n_clauses=c, n_graphs=g, clause_rowptr=rowptr, clause_lits=lits, var_graph_id=vgid,
                 clause_graph_id=cgid, variables_in_graph=np.bincount(vgid, minlength=g).astype(np.int32),
                 clauses_in_graph=np.bincount(cgid, minlength=g).astype(np.int32), instances=[])


class Dense:
    """What ``MLP.dense_layers`` holds (``tf.keras.layers.Dense``: ``kernel`` [in, out], ``bias`` [out])."""

    def __init__(self, units: int, activation=None):
        self.units, self.activation = int(units), activation
        self.kernel: Optional[np.ndarray] = None
        self.bias: Optional[np.ndarray] = None


class MLP:
    """``MLP(layer_count, hidden_nmap, out_nmap, activation=tf.nn.leaky_relu, out_activation=None,
    out_bias=None, do_layer_norm=False, norm_axis=0, normalizer=None, dropout_rate=0.0, init_zero=False,
    **kwargs)(inputs, training=None)`` -- model/mlp.py:13-50, the reference's argument lists.

    ``layer_count - 1`` hidden ``Dense(hidden_nmap, activation)`` and a last ``Dense(out_nmap)``; weights are
    created at the first call from the input width, ``glorot_uniform`` kernels and zero biases like Keras
    (:24, :37-39), or set with ``set_weights`` (Keras order: kernel, bias per layer).  The arithmetic runs in
    the extension (``dsat_mlp_forward``; inputs and outputs are CUDA tensors).  The five MLPs of a ``QuerySAT``
    are these objects, bound to the model's weight set (``load_weights`` / checkpoints fill them).

    Supported: the configuration QuerySAT uses (:117-122) -- ``activation=leaky_relu``, no layer norm, no
    output activation; ``dropout_rate`` is accepted (identity at inference).  Anything else raises."""

    def __init__(self, layer_count, hidden_nmap, out_nmap, activation=leaky_relu, out_activation=None,
                 out_bias=None, do_layer_norm=False, norm_axis=0, normalizer=None, dropout_rate=0.0,
                 init_zero=False, **kwargs):
        if getattr(activation, "__name__", None) != "leaky_relu":
            raise NotImplementedError("the kernels implement leaky_relu(0.2) hidden activations (model/mlp.py:14)")
        if out_activation is not None or do_layer_norm or normalizer is not None:
            raise NotImplementedError("out_activation / layer norm are not used by QuerySAT (model/query_sat.py:117-122) "
                                      "and are not built")
        if int(layer_count) < 1:
            raise ValueError("layer_count must be positive")
        self.name = kwargs.pop("name", None)
        self._seed = int(kwargs.pop("seed", 0))
        self._device = kwargs.pop("device", None)
        self.out_bias, self.init_zero, self.dropout_rate = out_bias, bool(init_zero), float(dropout_rate)
        self.dense_layers = [Dense(hidden_nmap, activation) for _ in range(int(layer_count) - 1)] + [Dense(out_nmap)]
        self.normalizer = None
        self._model: Optional["QuerySAT"] = None       # bound: weights live in the model's weight set
        self._handle = None
        self._handle_key = None

    # -- weights (Keras surface) ----------------------------------------------------------------
    @property
    def built(self) -> bool:
        return self.dense_layers[0].kernel is not None

    def build(self, input_width: int):
        rng = np.random.default_rng(self._seed)
        fi = int(input_width)
        last = len(self.dense_layers) - 1
        for i, d in enumerate(self.dense_layers):
            lim = np.sqrt(6.0 / (fi + d.units))                       # glorot_uniform
            d.kernel = rng.uniform(-lim, lim, size=(fi, d.units)).astype(np.float32)
            d.bias = np.zeros(d.units, dtype=np.float32)
            if i == last:
                if self.init_zero:
                    d.kernel[:] = 0                                     # model/mlp.py:36-37
                if self.out_bias is not None:
                    d.bias[:] = self.out_bias                           # :34-35
            fi = d.units
        self._handle_key = None

    def get_weights(self):
        return [a for d in self.dense_layers for a in (d.kernel, d.bias)]

    def set_weights(self, arrays):
        arrays = list(arrays)
        if len(arrays) != 2 * len(self.dense_layers):
            raise ValueError("expected %d arrays (kernel, bias per layer)" % (2 * len(self.dense_layers)))
        fi = None
        for d, k, b in zip(self.dense_layers, arrays[0::2], arrays[1::2]):
            k, b = np.asarray(k, dtype=np.float32), np.asarray(b, dtype=np.float32)
            if k.ndim != 2 or k.shape[1] != d.units or b.shape != (d.units,) or (fi is not None and k.shape[0] != fi):
                raise ValueError("kernel %s / bias %s do not fit a Dense(%d)" % (k.shape, b.shape, d.units))
            d.kernel, d.bias, fi = k, b, d.units
        self._handle_key = None

    def _bind(self, model: "QuerySAT", attr: str):
        self._model, self._attr = model, attr
        self._sync_from_model()

    def _sync_from_model(self):
        for d, (k, b) in zip(self.dense_layers, self._model._weights_host[self._attr]):
            d.kernel, d.bias = k, b
        self._handle_key = None

    # -- the call ----------------------------------------------------------------------------------
    def _device_object(self, device):
        key = (str(device),) + tuple(id(a) for a in self.get_weights())
        if self._handle_key != key:
            self.close()
            lib = _lib.load()
            dims = np.asarray([self.dense_layers[0].kernel.shape[0]] + [d.units for d in self.dense_layers], dtype=np.int32)
            blob = np.concatenate([np.ascontiguousarray(a, dtype=np.float32).ravel() for a in self.get_weights()])
            h = C.c_void_p()
            with torch.cuda.device(device):
                _lib.check(lib.dsat_mlp_create(len(self.dense_layers), dims.ctypes.data_as(C.c_void_p),
                                               blob.ctypes.data_as(C.c_void_p), blob.size, C.byref(h)))
            self._handle, self._handle_key, self._keep = h, key, self.get_weights()
        return self._handle

    def __call__(self, inputs: torch.Tensor, training=None) -> torch.Tensor:
        if training:
            raise NotImplementedError("training is outside the hot path this package implements")
        if not (isinstance(inputs, torch.Tensor) and inputs.is_cuda and inputs.dtype == torch.float32 and inputs.dim() == 2):
            raise TypeError("MLP inputs must be a 2-D float32 CUDA tensor")
        if not self.built:
            self.build(inputs.shape[1])
        if inputs.shape[1] != self.dense_layers[0].kernel.shape[0]:
            raise ValueError("inputs are %d wide, the first Dense takes %d" % (inputs.shape[1], self.dense_layers[0].kernel.shape[0]))
        lib = _lib.load()
        x = inputs.contiguous()
        rows = x.shape[0]
        with torch.cuda.device(x.device):
            h = self._device_object(x.device)
            y = torch.empty(rows, self.dense_layers[-1].units, dtype=torch.float32, device=x.device)
            ws = torch.empty(int(lib.dsat_mlp_workspace_floats(h, rows)), dtype=torch.float32, device=x.device)
            _lib.check(lib.dsat_mlp_forward(h, C.c_void_p(x.data_ptr()), rows, C.c_void_p(y.data_ptr()),
                                            C.c_void_p(ws.data_ptr()), runtime._stream(x.device)))
        return y

    call = __call__

    def close(self):
        if getattr(self, "_handle", None):
            _lib.load().dsat_mlp_destroy(self._handle)
            self._handle = None
            self._handle_key = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PairNorm:
    """``PairNorm(epsilon=1e-6, subtract_mean=False, **kwargs)(inputs, graph=None)`` --
    layers/normalization.py:27-71 with the reference's argument lists.  ``graph`` is the graph-level
    adjacency [G x rows] the reference passes (QuerySAT hands in the one normalised by the graph sizes,
    model/query_sat.py:206-211, :265, :279): ``mean = graph @ inputs`` with the tensor's own values, subtracted
    row by row through ``graph.indices[:, 0]``; without a graph the mean over all rows.  Then every row is
    scaled by ``rsqrt(mean(x^2 over the features) + epsilon)``.  Runs in the extension (``dsat_pair_norm``)."""

    def __init__(self, epsilon=1e-6, subtract_mean=False, **kwargs):
        self.epsilon, self.subtract_mean = float(epsilon), bool(subtract_mean)
        self.bias = None
        self.name = kwargs.pop("name", None)

    def __call__(self, inputs: torch.Tensor, graph=None, **kwargs) -> torch.Tensor:
        if not (isinstance(inputs, torch.Tensor) and inputs.is_cuda and inputs.dtype == torch.float32 and inputs.dim() == 2):
            raise TypeError("PairNorm inputs must be a 2-D float32 CUDA tensor")
        lib = _lib.load()
        x = inputs.contiguous()
        rows, width = x.shape
        seg, weight, n_graphs = None, None, 0
        if graph is not None and self.subtract_mean:
            coo = sparse.as_coo(graph)
            if coo.dense_shape[1] != rows or coo.indices.shape[0] != rows or \
                    not np.array_equal(coo.indices[:, 1], np.arange(rows)):
                raise NotImplementedError("graph must hold one entry per input row, in row order "
                                          "(as data/SatSpecifics.py:40-57 builds the membership matrices)")
            gid = coo.indices[:, 0]
            if rows and np.any(np.diff(gid) < 0):
                raise NotImplementedError("graph.indices[:, 0] must be non-decreasing (graphs are contiguous row ranges)")
            n_graphs = int(coo.dense_shape[0])
            seg = np.zeros(n_graphs + 1, dtype=np.int32)
            np.cumsum(np.bincount(gid, minlength=n_graphs), out=seg[1:])
            weight = torch.from_numpy(np.ascontiguousarray(coo.values, dtype=np.float32)).to(x.device)
        with torch.cuda.device(x.device):
            y = torch.empty_like(x)
            ws = torch.empty(int(lib.dsat_pair_norm_workspace_floats(rows, width, n_graphs)), dtype=torch.float32,
                             device=x.device)
            _lib.check(lib.dsat_pair_norm(C.c_void_p(x.data_ptr()), rows, width,
                                          seg.ctypes.data_as(C.c_void_p) if seg is not None else None, n_graphs,
                                          C.c_void_p(weight.data_ptr()) if weight is not None else None,
                                          self.epsilon, int(self.subtract_mean), C.c_void_p(y.data_ptr()),
                                          C.c_void_p(ws.data_ptr()), runtime._stream(x.device)))
        return y

    call = __call__


class QuerySAT:
    """``QuerySAT(optimizer, feature_maps=128, msg_layers=3, vote_layers=3, train_rounds=32,
    test_rounds=64, query_maps=128, supervised=True, trial=None)`` (model/query_sat.py:86-89).

    ``optimizer`` is accepted and ignored (the sampler constructs one only to satisfy the
    reference constructor, satuniformity/DiffusionSampler.py:198-201).  Weights come from
    ``load_weights`` (``.npz`` with the checkpoint's variable paths, or a TF2 checkpoint
    directory through ``diffusionsat_amd.tf_checkpoint``) or are Glorot-initialised like Keras.
    """

    def __init__(self, optimizer=None, feature_maps=128, msg_layers=3, vote_layers=3, train_rounds=32,
                 test_rounds=64, query_maps=128, supervised=True, trial=None, device="cuda:0", seed=0, **kwargs):
        if trial is not None:
            raise NotImplementedError("Optuna trials belong to training (hyperopt.py); not on this path")
        if kwargs.pop("use_linear_loss", False):
            raise NotImplementedError("use_linear_loss (model/query_sat.py:97, :294-304) is off in the reference and not built")
        if not (1 <= feature_maps <= 128 and 1 <= query_maps <= 128):
            raise NotImplementedError("the kernels' rows are 128 floats wide: feature_maps and query_maps must be in 1..128")
        self.supervised = bool(supervised)
        self.optimizer = optimizer
        self.train_rounds, self.test_rounds = train_rounds, test_rounds
        self.feature_maps, self.query_maps, self.vote_layers = feature_maps, query_maps, vote_layers
        self.logit_maps = 8
        self.prediction_tries = 1
        self.device = torch.device(device)
        # the five MLPs and two PairNorms of the reference's constructor (model/query_sat.py:101-122): attribute name ->
        # (layer count, hidden width, output width, Keras layer name); the width factors are the reference's defaults
        # (update 1.8, output 1, query 1.2, clauses 1.6; lit_mlp 4 x query_maps wide, 2 x query_maps out)
        fm, qm = int(feature_maps), int(query_maps)
        shapes = {
            "update_gate": (3, int(fm * 1.8), fm, "update_gate"),
            "variables_output": (2, int(fm * 1), self.logit_maps, "variables_output"),
            "variables_query": (2, int(qm * 1.2), qm, "variables_query"),
            "clause_mlp": (2, int(fm * 1.6), fm + qm, "clause_update"),
            "lit_mlp": (int(msg_layers), qm * 4, qm * 2, "lit_query"),
        }
        self._mlps = {attr: MLP(n, hidden, out, name=keras_name, do_layer_norm=False)
                      for attr, (n, hidden, out, keras_name) in shapes.items()}
        for attr, mlp in self._mlps.items():
            setattr(self, attr, mlp)
        self.variables_norm, self.clauses_norm = PairNorm(subtract_mean=True), PairNorm(subtract_mean=True)
        # the same shape as a dsat_layout for the fused kernels
        self.layout = _lib.default_layout()
        self.layout.feature_maps, self.layout.query_maps = int(feature_maps), int(query_maps)
        for name, mlp in self._mlps.items():
            self.layout.n_layers[MLP_INDEX[name]] = len(mlp.dense_layers)
            self.layout.hidden[MLP_INDEX[name]] = mlp.dense_layers[0].units
        self._weights_host: WeightSet = glorot_weights(seed=seed, layout=self.layout)
        self._weights: Optional[runtime.Weights] = None
        self._sessions = {}
        self._coo_inputs = []          # [(adj_matrix object, GraphInputs)], most recent last
        self._seed = int(seed)
        self._calls = 0
        for name, mlp in self._mlps.items():
            mlp._bind(self, name)

    # -- weights ------------------------------------------------------------------------
    def set_weights(self, ws: WeightSet):
        self._weights_host = ws
        self._weights = None
        self._sessions.clear()
        for mlp in self._mlps.values():
            mlp._sync_from_model()

    def load_weights(self, path: str) -> bool:
        """``.npz`` (diffusionsat_amd.weights.save_npz) or a TF2 checkpoint directory/prefix.
        Returns False when nothing was found (the reference prints "Checkpoint not found!" and
        continues with fresh weights, satuniformity/DiffusionSampler.py:222-225)."""
        import os
        if path and os.path.isfile(path) and path.endswith(".npz"):
            self.set_weights(load_npz(path))
            return True
        from . import tf_checkpoint
        ws = tf_checkpoint.load_queriesat_weights(path, self.layout, base=self._weights_host)   # expect_partial()
        if ws is None:
            return False
        self.restore_report = dict(tf_checkpoint.last_restore_report)
        self.set_weights(ws)
        return True

    def device_weights(self) -> runtime.Weights:
        if self._weights is None:
            self._weights = runtime.Weights(self._weights_host, self.layout, device=self.device)
        return self._weights

    def graph_inputs(self, adj_matrix, clauses_graph=None, variables_graph=None) -> GraphInputs:
        """The device graph behind the first three arguments of ``diffusion_step`` / ``predict_step``: a
        ``GraphInputs`` as it is, or the three sparse tensors of the reference's signature
        (model/query_sat.py:467-481), converted once per ``adj_matrix`` object and kept for its later steps
        (the sampler passes the same ``step_data`` tensors T times, satuniformity/DiffusionSampler.py:105-117)."""
        if isinstance(adj_matrix, GraphInputs):
            return adj_matrix
        for obj, gi in self._coo_inputs:
            if obj is adj_matrix:
                return gi
        if clauses_graph is None or variables_graph is None:
            raise TypeError("diffusion_step(adj_matrix, clauses_graph, variables_graph, ...): pass the three sparse "
                            "tensors of the reference's signature, or a GraphInputs as adj_matrix")
        gi = GraphInputs.from_coo(adj_matrix, clauses_graph, variables_graph, device=self.device)
        self._coo_inputs.append((adj_matrix, gi))          # holding the object keeps its identity unique
        while len(self._coo_inputs) > 4:
            _, old = self._coo_inputs.pop(0)
            self._sessions.pop(id(old), None)
        return gi

    def _session(self, gi: GraphInputs) -> runtime.Session:
        key = id(gi)
        if key not in self._sessions:
            sess = runtime.Session(gi.graph, self.device_weights(), max_rounds=max(self.test_rounds, 1),
                                   device=self.device)
            sess.set_supervised(bool(self.supervised))       # :289-309: which loss picks a graph's logit map
            self._sessions[key] = (gi, sess)
        return self._sessions[key][1]

    # -- the tf.function entry points on the sampling path ---------------------------------
    def diffusion_step(self, adj_matrix, clauses_graph=None, variables_graph=None, solutions=None,
                       noise_scale: float = 1.0, noisy_num: torch.Tensor = None, *, normal=None, labels=None,
                       early_exit=True, denoised_num=None):
        """model/query_sat.py:467-481, the reference's argument list: ``adj_matrix`` [2N x C], ``clauses_graph``
        [G x C] and ``variables_graph`` [G x N] are the sparse tensors of data/SatSpecifics.py:21-69 (anything
        ``sparse.as_coo`` reads: ``tf.SparseTensor``-like, torch sparse COO, SciPy, ``(indices, values,
        dense_shape)``) -- or ``adj_matrix`` is a ``GraphInputs`` and the other two are left out; ``solutions``
        is unused at inference, as in the reference (:476: ``labels=None``).  ``normal`` / ``labels`` inject the
        random draws of :239 / :145; by default they come from the device Philox stream."""
        gi = self.graph_inputs(adj_matrix, clauses_graph, variables_graph)
        sess = self._session(gi)
        if not isinstance(noisy_num, torch.Tensor):
            noisy_num = torch.as_tensor(np.asarray(noisy_num, dtype=np.float32))
        noisy_num = noisy_num.to(device=self.device, dtype=torch.float32)
        self._calls += 1
        out = sess.diffusion_step(float(noise_scale), noisy_num.contiguous(), normal=normal, labels=labels,
                                  seed=self._seed, step_index=self._calls, rounds=self.test_rounds,
                                  early_exit=early_exit,
                                  denoised=None if denoised_num is None else denoised_num.reshape(-1).contiguous())
        return {"steps_taken": out["steps_taken"], "loss": torch.zeros((), device=self.device),
                "prediction": out["prediction"]}

    def diffusion_step_self(self, adj_matrix, clauses_graph=None, variables_graph=None,
                            solutions=None, noise_scale: float = 1.0, noisy_num: torch.Tensor = None,
                            denoised_num: torch.Tensor = None, **kw):
        """model/query_sat.py:483-498: ``denoised_num`` [N,1] is the previous step's
        sigmoid(prediction) (satuniformity/DiffusionSampler.py:50-59)."""
        return self.diffusion_step(adj_matrix, clauses_graph, variables_graph, solutions, noise_scale, noisy_num,
                                   denoised_num=denoised_num, **kw)

    def _predict_once(self, adj_matrix: GraphInputs):
        """``call(training=False)`` as predict_step makes it (model/query_sat.py:421, :133-184): noise_scale
        ~ U(0,1) (:144), random labels (:145), noisy_num = construct_training_input(labels, noise) (:76-82,
        :214); the host draws those three values, the rounds run on the device."""
        n = adj_matrix.n_vars
        rng = np.random.default_rng(self._seed + 7919 * (self._calls + 1))
        noise_scale = float(rng.random())
        labels = rng.integers(0, 2, n).astype(np.int32)
        one_hot = np.stack([1 - labels, labels], axis=1).astype(np.float32)   # tf.one_hot(labels, 2)
        t = np.float32(noise_scale) ** np.float32(0.5)
        num_at_t = one_hot * (1 - t) + t / 2
        rounded = np.floor(num_at_t[:, 0:1] + rng.random((n, 1)).astype(np.float32))
        noisy = np.concatenate([rounded, 1 - rounded], axis=1).astype(np.float32)
        return self.diffusion_step(adj_matrix, noise_scale=noise_scale,
                                   noisy_num=torch.from_numpy(noisy).to(self.device),
                                   labels=torch.from_numpy(labels).to(self.device))

    def predict_step(self, adj_matrix, clauses_graph=None, variables_graph=None, solutions=None):
        """model/query_sat.py:419-451.  ``prediction_tries`` == 1: one call.  Otherwise (:423-443) the model is
        run ``prediction_tries`` times; a graph keeps the prediction of the first try that satisfies all its
        clauses (``is_graph_sat``, utils/sat.py:165-180), graphs never solved end with zeros, as in the
        reference (``final_predictions`` starts at zero and only solved graphs are added)."""
        adj_matrix = self.graph_inputs(adj_matrix, clauses_graph, variables_graph)
        if self.prediction_tries == 1:
            return self._predict_once(adj_matrix)
        sess = self._session(adj_matrix)
        gid = torch.from_numpy(adj_matrix.batch.var_graph_id.astype(np.int64)).to(self.device)
        final = torch.zeros(adj_matrix.n_vars, dtype=torch.float32, device=self.device)
        solved = torch.zeros(adj_matrix.n_graphs, dtype=torch.float32, device=self.device)
        out = None
        for _ in range(int(self.prediction_tries)):
            out = self._predict_once(adj_matrix)
            sat = sess.op_is_graph_sat(out["prediction"].contiguous()).to(torch.float32)
            new = torch.clamp(sat - solved, 0, 1)                 # :433  graphs solved by this try for the first time
            final += out["prediction"] * new[gid]                # :435-436
            solved = solved + new                                 # :437
        return {"steps_taken": out["steps_taken"], "loss": out["loss"], "prediction": final}

    def get_config(self):
        return {"model": self.__class__.__name__, "feature_maps": self.feature_maps,
                "query_maps": self.query_maps, "train_rounds": self.train_rounds,
                "test_rounds": self.test_rounds, "mlp_layers": self.vote_layers}
