"""``DiffusionSampler(model_path, dimacs_filename).samples(n)`` -- the reference's outer surface
(satuniformity/DiffusionSampler.py:195-311, README.md:46-52) on the MI355X-native hot path.

What runs where
  * host (this file): DIMACS loading, replicate-instance batching, the ``dict[int -> count]``
    of distinct satisfying assignments, the 0.5 % abort rule;
  * device (``dsat_sample``): the whole T-step reverse diffusion of one batch -- randomized
    rounding, R message-passing rounds per step, per-graph SAT check, theoretic reverse step,
    freeze-on-first-solution -- with no host round trip inside the loop.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import runtime
from .batch import Batch, make_batch, replicate_instance
from .dimacs import DimacsFile, VariableAssignment
from .query_sat import GraphInputs, QuerySAT

# module constants of the reference (satuniformity/DiffusionSampler.py:17-21)
use_baseline_sampling = True
test_rounds = 32
diffusion_steps = 32
self_supervised = False
max_nodes_per_batch = 20000      # config.py:35


def diffusion(N: int, model: QuerySAT, graph_inputs: GraphInputs, seed: int = 0, early_exit: bool = True,
              uniform=None, normal=None, labels=None, want_predictions: bool = False):
    """``diffusion(N, model, dataset, step_data)`` (satuniformity/DiffusionSampler.py:78-191):
    returns (mean(cum_accuracy), predictions[N_vars] in {0,1}, var_correct[N_vars])."""
    sess = model._session(graph_inputs)
    out = sess.sample(N, model.test_rounds, seed=seed, early_exit=early_exit, uniform=uniform, normal=normal,
                      labels=labels, want_predictions=want_predictions,
                      self_conditioning=self_supervised)          # module switch, as in the reference (:21, :110)
    gid = torch.from_numpy(graph_inputs.batch.var_graph_id.astype(np.int64)).to(out["solved"].device)
    var_correct = out["solved"].to(torch.float32)[gid]            # :176-180
    res = (out["solved_fraction"], out["assignment"], var_correct)
    return res + (out["predictions"],) if want_predictions else res


class DiffusionSampler:
    """Same constructor and ``samples`` contract as the reference.

    ``model_path``: a TF2 checkpoint directory/prefix (satuniformity/DiffusionSampler.py:215-227),
    or an ``.npz`` written by ``diffusionsat_amd.weights.save_npz``; when nothing is found the
    model keeps its fresh Glorot weights, as the reference does ("Checkpoint not found!").
    ``test_solutions_multiplier_k`` etc. of the reference's dataset only size how many copies of
    the formula are generated; here copies are generated on demand until ``n_samples`` is met.
    """

    def __init__(self, model_path: Optional[str], dimacs_filename: str, device: str = "cuda:0", seed: int = 0,
                 rank: int = 0, world_size: int = 1):
        self.model = QuerySAT(optimizer=None, test_rounds=test_rounds, device=device, seed=seed)
        self.restored = bool(model_path) and self.model.load_weights(model_path)
        test_dimacs = DimacsFile(filename=dimacs_filename)
        test_dimacs.load()
        self.dimacs = test_dimacs
        self.n_vars = test_dimacs.number_of_vars()
        self.clauses = test_dimacs.clauses()
        if self.n_vars <= 0:
            raise ValueError("%s: no variables" % dimacs_filename)
        # a batch = as many copies of the formula as fit the node budget (data/dimac.py:267-293,
        # data/diffusion_sat_instances.py:80-94)
        self.copies_per_batch = replicate_instance(self.n_vars, self.clauses, copies=0,
                                                   max_nodes_per_batch=max_nodes_per_batch)
        self.batch: Batch = make_batch([(self.n_vars, self.clauses)] * self.copies_per_batch)
        self.inputs = GraphInputs(self.batch, device=device)
        self.seed, self.rank, self.world_size = int(seed), int(rank), int(world_size)
        self.success_rate = 0.0
        # True: re-check every copy on the host like the reference's loop (:283-307); False: trust the
        # device's per-copy SAT flag (tests/test_gpu_fullsize.py compares the two)
        self.host_recheck = False
        # independent batches kept in flight by samples(); 1 = strictly one after the other
        self.batches_in_flight = 2
        self._slots = []

    def _decode(self, assignment: np.ndarray, counts: Dict[int, int], still_needed: int):
        """Per-graph loop of ``samples`` (:283-307): int-encode (bit i <-> x_{i+1}), re-check."""
        total = sat = 0
        n = self.n_vars
        asgn = VariableAssignment(n_vars=n, clauses=self.clauses)
        for i in range(self.copies_per_batch):
            asgn.assign_all_from_bit_list(assignment[i * n:(i + 1) * n])
            total += 1
            if asgn.satisfiable():
                sat += 1
                key = int(asgn)
                counts[key] = counts.get(key, 0) + 1
                still_needed -= 1
                if still_needed == 0:
                    break
        return total, sat, still_needed

    def _decode_flags(self, assignment: np.ndarray, solved: np.ndarray, counts: Dict[int, int], still_needed: int):
        """Same result as ``_decode`` without the per-copy Python SAT check: the device already
        evaluated every copy's clauses on exactly these bits (``solved``: the per-graph flag of
        ``dsat_sample`` -- a copy is frozen at the step that satisfied it, so flag == satisfiable);
        the bits are packed little-endian, bit i <-> x_{i+1} (utils/VariableAssignment.py:63-69)."""
        n, g = self.n_vars, self.copies_per_batch
        packed = np.packbits(assignment[: g * n].reshape(g, n).astype(np.uint8), axis=1, bitorder="little")
        total = sat = 0
        for i in range(g):
            total += 1
            if solved[i]:
                sat += 1
                key = int.from_bytes(packed[i].tobytes(), "little")
                counts[key] = counts.get(key, 0) + 1
                still_needed -= 1
                if still_needed == 0:
                    break
        return total, sat, still_needed

    def _slots_for(self, k: int):
        """k independent sessions over the same device graph and weights, each with its own HIP
        stream: the batches of one ``samples`` call overlap on the GPU (a config-2-size batch
        leaves a quarter of the CUs idle in every kernel)."""
        while len(self._slots) < k:
            sess = runtime.Session(self.inputs.graph, self.model.device_weights(), max_rounds=test_rounds,
                                   device=self.model.device)
            self._slots.append((sess, torch.cuda.Stream(device=self.model.device)))
        return self._slots[:k]

    def samples(self, n_samples: int) -> Dict[int, int]:
        """:param n_samples: how many correct samples to generate
        :return: dict: solution as int => count   (satuniformity/DiffusionSampler.py:229-311)

        Batches are decoded in launch order with the reference's stopping rules, so the result does
        not depend on ``batches_in_flight`` (with more than one in flight the work of at most
        ``batches_in_flight - 1`` batches is discarded when the loop stops)."""
        counts: Dict[int, int] = {}
        total = sat = 0
        needed = int(n_samples)
        slots = self._slots_for(max(1, int(self.batches_in_flight)))
        pending = []          # (stream, output dict) in launch order
        it = 0

        def launch():
            nonlocal it
            sess, stream = slots[it % len(slots)]
            # every batch (and every rank) draws from its own Philox sub-stream
            seed = self.seed + 1000003 * (it * self.world_size + self.rank) + 1
            stream.wait_stream(torch.cuda.current_stream(self.model.device))
            with torch.cuda.stream(stream):
                out = sess.sample(diffusion_steps, test_rounds, seed=seed, early_exit=True,
                                  want_solved_fraction=False, self_conditioning=self_supervised)
            pending.append((stream, out))
            it += 1

        while needed > 0:
            if total > 0 and sat / total < 0.005:           # :261-263 too many unsat samples
                break
            while len(pending) < len(slots):
                launch()
            stream, out = pending.pop(0)
            stream.synchronize()
            assignment = out["assignment"].cpu().numpy()
            if self.host_recheck:
                t, s, needed = self._decode(assignment, counts, needed)
            else:
                t, s, needed = self._decode_flags(assignment, out["solved"].cpu().numpy() > 0, counts, needed)
            total += t
            sat += s
        for stream, _ in pending:                           # drain what was launched ahead
            stream.synchronize()
        self.success_rate = sat / total if total else 0.0
        return counts
