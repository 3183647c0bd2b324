"""``DiffusionSampler(model_path, dimacs_filename).samples(n)`` -- the reference's outer surface
(satuniformity/DiffusionSampler.py:195-311, README.md:46-52) on the MI355X-native hot path.

What runs where
  * host (this file): DIMACS loading, replicate-instance batching, the ``dict[int -> count]``
    of distinct satisfying assignments, the 0.5 % abort rule;
  * device (``dsat_sample``): the whole T-step reverse diffusion of one batch -- randomized
    rounding, R message-passing rounds per step, per-graph SAT check, theoretic reverse step,
    freeze-on-first-solution -- with no host round trip inside the loop.

Multi-GPU (one process per GPU, ``rank`` / ``world_size``): every rank produces its share of the
``n_samples`` (``distributed.split_sample_budget``) from its own Philox sub-streams and the per-rank
dictionaries are merged at the end (``distributed.merge_counts``); there is no exchange inside the
sampling loop.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import distributed, runtime
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


def epoch_batch_sizes(total_copies: int, fit: int, reference_drop_bug: bool = True) -> List[int]:
    """Copies per batch over one epoch of the reference's test dataset: ``total_copies`` identical
    formulas go through ``__batch_files`` (data/dimac.py:267-293), which closes a batch when the next
    copy would exceed the node budget -- and loses that copy (reproduced by default
    here because it decides how many copies the reference really runs per epoch)."""
    sizes, cur = [], 0
    for _ in range(int(total_copies)):
        if cur < fit or cur == 0:
            cur += 1
        else:
            sizes.append(cur)
            cur = 0 if reference_drop_bug else 1
    if cur:
        sizes.append(cur)
    return sizes


class DiffusionSampler:
    """Same constructor and ``samples`` contract as the reference.

    ``model_path``: a TF2 checkpoint directory/prefix (satuniformity/DiffusionSampler.py:215-227),
    or an ``.npz`` written by ``diffusionsat_amd.weights.save_npz``; when nothing is found the
    model keeps its fresh Glorot weights, as the reference does ("Checkpoint not found!").

    Batch composition.  The reference's dataset yields ``AllSolutions.count() * 10`` copies of the
    formula per epoch (data/diffusion_sat_instances.py:80-94, ``test_solutions_multiplier_k=10`` at
    satuniformity/DiffusionSampler.py:206-210), batched to the node budget, and repeats the epoch.
    The model counter behind ``AllSolutions`` (pyunigen / approxmc / a prebuilt binary) is outside this
    path, so the count is an input here: pass ``n_solutions`` to get the reference's epoch schedule
    (``epoch_batch_sizes``), or ``copies_per_batch`` to fix the batch size; with neither, every batch is
    filled to the node budget.  Early exit (``is_batch_sat``) is batch-global and the 0.5 % abort is
    evaluated at batch boundaries, so the batch composition decides how many rounds run and where
    ``samples()`` aborts: for a formula with few models the default differs from the reference's last
    (partial) batch of each epoch unless ``n_solutions`` is given -- parity unpinned either way.
    """

    def __init__(self, model_path: Optional[str], dimacs_filename: str, device: str = "cuda:0", seed: int = 0,
                 rank: int = 0, world_size: int = 1, n_solutions: Optional[int] = None,
                 test_solutions_multiplier_k: int = 10, copies_per_batch: Optional[int] = None,
                 retire_solved: bool = False):
        self.device = device
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
        fit = replicate_instance(self.n_vars, self.clauses, copies=0, max_nodes_per_batch=max_nodes_per_batch)
        if copies_per_batch is not None:
            if copies_per_batch < 1:
                raise ValueError("copies_per_batch must be positive")
            self.batch_sizes = [int(copies_per_batch)]
        elif n_solutions is not None:
            self.batch_sizes = epoch_batch_sizes(int(n_solutions) * int(test_solutions_multiplier_k), fit) or [fit]
        else:
            self.batch_sizes = [fit]
        self.copies_per_batch = self.batch_sizes[0]
        self.seed, self.rank, self.world_size = int(seed), int(rank), int(world_size)
        if not (0 <= self.rank < self.world_size):
            raise ValueError("rank %d outside world_size %d" % (self.rank, self.world_size))
        self.success_rate = 0.0
        # True: re-check every copy on the host like the reference's loop (:283-307); False: trust the
        # device's per-copy SAT flag (tests/test_gpu_fullsize.py compares the two)
        self.host_recheck = False
        # independent batches kept in flight by samples(); 1 = strictly one after the other
        self.batches_in_flight = 2
        # drop a copy from its batch once it has been solved (``dsat_session_set_retire``): the remaining
        # denoising steps run on fewer copies; results per copy as without it, except that the batch-global
        # early exit then refers to the copies still in the batch
        self.retire_solved = bool(retire_solved)
        # is_batch_sat break inside every denoising step (model/query_sat.py:330-338), as in the reference
        self.early_exit = True
        self._inputs: Dict[int, GraphInputs] = {}     # copies -> device graph
        self._slots: Dict[int, list] = {}             # copies -> [(session, stream), ...]

    # -- batches -------------------------------------------------------------------------------
    def _inputs_for(self, copies: int) -> GraphInputs:
        if copies not in self._inputs:
            batch: Batch = make_batch([(self.n_vars, self.clauses)] * copies)
            self._inputs[copies] = GraphInputs(batch, device=self.device)
        return self._inputs[copies]

    @property
    def inputs(self) -> GraphInputs:
        return self._inputs_for(self.copies_per_batch)

    @property
    def batch(self) -> Batch:
        return self.inputs.batch

    def _decode(self, packed: np.ndarray, copies: int, counts: Dict[int, int], still_needed: int):
        """Per-graph loop of ``samples`` (:283-307): int-encode (bit i <-> x_{i+1}), re-check."""
        total = sat = 0
        n = self.n_vars
        asgn = VariableAssignment(n_vars=n, clauses=self.clauses)
        bits = np.unpackbits(packed.reshape(copies, -1), axis=1, bitorder="little")[:, :n]
        for i in range(copies):
            asgn.assign_all_from_bit_list(bits[i])
            total += 1
            if asgn.satisfiable():
                sat += 1
                key = int(asgn)
                counts[key] = counts.get(key, 0) + 1
                still_needed -= 1
                if still_needed == 0:
                    break
        return total, sat, still_needed

    def _decode_flags(self, packed: np.ndarray, solved: np.ndarray, copies: int, counts: Dict[int, int],
                      still_needed: int):
        """Same result as ``_decode`` without the per-copy Python SAT check: the device already
        evaluated every copy's clauses on exactly these bits (``solved``: the per-graph flag of
        ``dsat_sample`` -- a copy is frozen at the step that satisfied it, so flag == satisfiable);
        the bits arrive packed by the device, little-endian, bit i <-> x_{i+1}
        (utils/VariableAssignment.py:63-69; ``dsat_op_pack_assignment``)."""
        g = copies
        packed = packed.reshape(g, -1)
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

    # -- device side (overridden by the CPU tests of the multi-rank path) -------------------------
    def _slots_for(self, k: int, copies: int):
        """k independent sessions over the same device graph and weights, each with its own HIP
        stream: the batches of one ``samples`` call overlap on the GPU (a config-2-size batch
        leaves a quarter of the CUs idle in every kernel)."""
        slots = self._slots.setdefault(copies, [])
        gi = self._inputs_for(copies)
        while len(slots) < k:
            sess = runtime.Session(gi.graph, self.model.device_weights(), max_rounds=test_rounds,
                                   device=self.model.device)
            if self.retire_solved:
                sess.set_retire(True)
            slots.append((sess, torch.cuda.Stream(device=self.model.device)))
        for sess, _ in slots[:k]:
            sess.set_throughput(k > 1)       # several batches in flight: CU-cycles count, not one batch's latency
        return slots[:k]

    def _launch(self, it: int, copies: int, seed: int):
        """Enqueue the T-step diffusion of batch number ``it``; returns a handle for ``_collect``."""
        k = max(1, int(self.batches_in_flight))
        sess, stream = self._slots_for(k, copies)[it % k]
        stream.wait_stream(torch.cuda.current_stream(self.model.device))
        with torch.cuda.stream(stream):
            out = sess.sample(diffusion_steps, test_rounds, seed=seed, early_exit=self.early_exit,
                              want_solved_fraction=False, self_conditioning=self_supervised)
            out["packed"] = sess.pack_assignment(out["assignment"])       # copies * ceil(n/8) bytes cross PCIe
        return stream, out

    def _collect(self, handle):
        """Wait for a launched batch: (packed bits uint8 [copies * ceil(n_vars/8)], solved bool [copies])."""
        stream, out = handle
        stream.synchronize()
        return out["packed"].cpu().numpy(), out["solved"].cpu().numpy() > 0

    # -- the reference surface ---------------------------------------------------------------------
    def samples(self, n_samples: int) -> Dict[int, int]:
        """:param n_samples: how many correct samples to generate
        :return: dict: solution as int => count   (satuniformity/DiffusionSampler.py:229-311)

        Batches are decoded in launch order with the reference's stopping rules, so the result does
        not depend on ``batches_in_flight`` (with more than one in flight the work of at most
        ``batches_in_flight - 1`` batches is discarded when the loop stops).  With ``world_size`` > 1
        this rank draws ``split_sample_budget(n_samples)`` of them and every rank returns the merged
        dictionary (``torch.distributed`` must be initialised by the launcher)."""
        if self.world_size > 1:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() == self.world_size):
                raise RuntimeError("DiffusionSampler(world_size=%d): torch.distributed is not initialised with "
                                   "that world size" % self.world_size)
            mine = distributed.split_sample_budget(n_samples, self.rank, self.world_size)
            local, total, sat = self._samples_local(mine)
            stats = distributed.merge_counts({-1: total, -2: sat})      # the same merge carries the two tallies
            self.success_rate = stats[-2] / stats[-1] if stats.get(-1) else 0.0
            return distributed.merge_counts(local)
        counts, total, sat = self._samples_local(int(n_samples))
        self.success_rate = sat / total if total else 0.0
        return counts

    def _samples_local(self, n_samples: int):
        counts: Dict[int, int] = {}
        total = sat = 0
        needed = int(n_samples)
        k = max(1, int(self.batches_in_flight))
        pending = []          # (copies, handle) in launch order
        it = 0
        while needed > 0:
            if total > 0 and sat / total < 0.005:           # :261-263 too many unsat samples
                break
            while len(pending) < k:
                copies = self.batch_sizes[it % len(self.batch_sizes)]
                # every batch (and every rank) draws from its own Philox sub-stream
                seed = self.seed + 1000003 * (it * self.world_size + self.rank) + 1
                pending.append((copies, self._launch(it, copies, seed)))
                it += 1
            copies, handle = pending.pop(0)
            packed, solved = self._collect(handle)
            if self.host_recheck:
                t, s, needed = self._decode(packed, copies, counts, needed)
            else:
                t, s, needed = self._decode_flags(packed, solved, copies, counts, needed)
            total += t
            sat += s
        for _, handle in pending:                           # drain what was launched ahead
            self._collect(handle)
        return counts, total, sat
