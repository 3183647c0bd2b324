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

from . import _lib, runtime
from .batch import Batch
from .weights import WeightSet, glorot_weights, load_npz

MLP_INDEX = {"variables_query": 0, "lit_mlp": 1, "clause_mlp": 2, "update_gate": 3, "variables_output": 4,
             # Keras layer names used by the reference constructor (model/query_sat.py:117-122)
             "lit_query": 1, "clause_update": 2}


class GraphInputs:
    """The per-batch inputs ``diffusion_step`` receives in the reference
    (``adj_matrix``, ``clauses_graph``, ``variables_graph`` built by
    data/SatSpecifics.py:21-69), held as the device CSR graph of one ``Batch``."""

    def __init__(self, batch: Batch, device=None):
        self.batch = batch
        self.graph = runtime.Graph(batch, device=device)

    @property
    def n_vars(self):
        return self.batch.n_vars

    @property
    def n_graphs(self):
        return self.batch.n_graphs


class MLP:
    """model/mlp.py:13-50 -- ``MLP(layer_count, hidden_nmap, out_nmap)(inputs)`` for one of the
    five MLPs of a QuerySAT model (weights are owned by the model)."""

    def __init__(self, model: "QuerySAT", name: str):
        self.model, self.name, self.index = model, name, MLP_INDEX[name]

    def __call__(self, inputs: torch.Tensor, graph_inputs: GraphInputs, training=None) -> torch.Tensor:
        if training:
            raise NotImplementedError("training is outside the hot path this package implements")
        return self.model._session(graph_inputs).op_mlp(self.index, inputs.contiguous())


class PairNorm:
    """layers/normalization.py:27-71 -- ``PairNorm(epsilon=1e-6, subtract_mean=True)(inputs, graph)``;
    ``side`` picks the graph membership (0 = variables_graph, 1 = clauses_graph)."""

    def __init__(self, model: "QuerySAT", epsilon=1e-6, subtract_mean=True):
        if epsilon != 1e-6 or not subtract_mean:
            raise NotImplementedError("the kernels implement the configuration QuerySAT uses: "
                                      "epsilon=1e-6, subtract_mean=True (model/query_sat.py:114-115)")
        self.model = model

    def __call__(self, inputs: torch.Tensor, graph_inputs: GraphInputs, side: int) -> torch.Tensor:
        return self.model._session(graph_inputs).op_pair_norm(side, inputs.contiguous())


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
        if not (1 <= feature_maps <= 128 and 1 <= query_maps <= 128):
            raise NotImplementedError("the kernels' rows are 128 floats wide: feature_maps and query_maps must be in 1..128")
        self.supervised = supervised
        self.train_rounds, self.test_rounds = train_rounds, test_rounds
        self.feature_maps, self.query_maps, self.vote_layers = feature_maps, query_maps, vote_layers
        self.logit_maps = 8
        self.prediction_tries = 1
        self.device = torch.device(device)
        self.layout = _lib.default_layout()
        self.layout.n_layers[1] = msg_layers          # lit_mlp depth (model/query_sat.py:122)
        # widths and the hidden sizes that follow from them (model/query_sat.py:101-122: query 1.2 x query_maps,
        # lit 4 x query_maps, clause 1.6 x, update 1.8 x, output 1 x feature_maps); 128 / 128 gives the defaults
        self.layout.feature_maps, self.layout.query_maps = int(feature_maps), int(query_maps)
        for m, h in enumerate((int(query_maps * 1.2), query_maps * 4, int(feature_maps * 1.6), int(feature_maps * 1.8),
                               int(feature_maps * 1))):
            self.layout.hidden[m] = h
        self._weights_host: WeightSet = glorot_weights(seed=seed, layout=self.layout)
        self._weights: Optional[runtime.Weights] = None
        self._sessions = {}
        self._seed = int(seed)
        self._calls = 0
        self.variables_norm = PairNorm(self)
        self.clauses_norm = PairNorm(self)
        self.update_gate = MLP(self, "update_gate")
        self.variables_output = MLP(self, "variables_output")
        self.variables_query = MLP(self, "variables_query")
        self.clause_mlp = MLP(self, "clause_mlp")
        self.lit_mlp = MLP(self, "lit_mlp")

    # -- weights ------------------------------------------------------------------------
    def set_weights(self, ws: WeightSet):
        self._weights_host = ws
        self._weights = None
        self._sessions.clear()

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

    def _session(self, gi: GraphInputs) -> runtime.Session:
        key = id(gi)
        if key not in self._sessions:
            self._sessions[key] = (gi, runtime.Session(gi.graph, self.device_weights(),
                                                       max_rounds=max(self.test_rounds, 1), device=self.device))
        return self._sessions[key][1]

    # -- the tf.function entry points on the sampling path ---------------------------------
    def diffusion_step(self, adj_matrix: GraphInputs, clauses_graph=None, variables_graph=None, solutions=None,
                       noise_scale: float = 1.0, noisy_num: torch.Tensor = None, *, normal=None, labels=None,
                       early_exit=True, denoised_num=None):
        """model/query_sat.py:467-481.  ``adj_matrix`` carries the batch (``GraphInputs``);
        ``clauses_graph`` / ``variables_graph`` / ``solutions`` are accepted for signature
        compatibility (the graph object already holds the memberships; solutions are unused at
        inference, as in the reference).  ``normal`` / ``labels`` inject the random draws of
        :239 / :145; by default they come from the device Philox stream."""
        sess = self._session(adj_matrix)
        self._calls += 1
        out = sess.diffusion_step(float(noise_scale), noisy_num.contiguous(), normal=normal, labels=labels,
                                  seed=self._seed, step_index=self._calls, rounds=self.test_rounds,
                                  early_exit=early_exit,
                                  denoised=None if denoised_num is None else denoised_num.reshape(-1).contiguous())
        return {"steps_taken": out["steps_taken"], "loss": torch.zeros((), device=self.device),
                "prediction": out["prediction"]}

    def diffusion_step_self(self, adj_matrix: GraphInputs, clauses_graph=None, variables_graph=None,
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

    def predict_step(self, adj_matrix: GraphInputs, clauses_graph=None, variables_graph=None, solutions=None):
        """model/query_sat.py:419-451.  ``prediction_tries`` == 1: one call.  Otherwise (:423-443) the model is
        run ``prediction_tries`` times; a graph keeps the prediction of the first try that satisfies all its
        clauses (``is_graph_sat``, utils/sat.py:165-180), graphs never solved end with zeros, as in the
        reference (``final_predictions`` starts at zero and only solved graphs are added)."""
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
