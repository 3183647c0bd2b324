"""Streaming sampler: ``dsat_sample`` over a stream of node-budget batches (BASELINE config 4:
instances -> ``batch.batch_by_node_budget`` (data/dimac.py:267-293) -> per-GPU shards
(``distributed.shard_instances``) -> this pipeline).

The reference feeds QuerySAT one batch after the other from a ``tf.data`` pipeline.  Here every
batch needs a device graph (both CSR orientations, degree weights), a session workspace and a
captured hipGraph of one denoising step; a config-2-size batch then runs ~0.1 s on the device.
``slots`` batches are kept in flight, each slot owning a HIP stream and a pooled arena
(``runtime.Arena``): while slot A's batch runs, the host builds and uploads slot B's batch, so the
per-batch host set-up is off the critical path, and two batches in flight also fill the CUs a
single 20k-node batch leaves idle.  Results are delivered in stream order.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, Iterable, List, Optional

import numpy as np
import torch

from . import _lib, runtime
from .batch import Batch


class _Slot:
    def __init__(self, device, arena: Optional[runtime.Arena]):
        self.stream = torch.cuda.Stream(device=device)
        self.arena = arena
        self.pending = None       # (index, batch, graph, session, out, ev0, ev1, host_ms)


class StreamSampler:
    """``run(batches)`` samples every batch for ``n_steps`` denoising steps of ``rounds`` rounds.

    pooled=True : graphs / sessions live in per-slot arenas (no hipMalloc / hipFree per batch);
    pooled=False: plain ``dsat_graph_create`` / ``dsat_session_create`` per batch (the A/B baseline
                  of bench.py --config 4)."""

    def __init__(self, weights: runtime.Weights, device=None, slots: int = 2, n_steps: int = 32, rounds: int = 32,
                 seed: int = 0, pooled: bool = True, graph_replay: bool = True, early_exit: bool = True,
                 arena_device_bytes: int = 256 << 20, arena_pinned_bytes: int = 16 << 20):
        self.weights = weights
        self.device = weights.device if device is None else runtime._device(device)
        self.n_steps, self.rounds, self.seed = int(n_steps), int(rounds), int(seed)
        self.pooled, self.graph_replay, self.early_exit = bool(pooled), bool(graph_replay), bool(early_exit)
        self._dev_bytes, self._pin_bytes = int(arena_device_bytes), int(arena_pinned_bytes)
        self.slots: List[_Slot] = []
        for _ in range(max(1, int(slots))):
            arena = runtime.Arena(self._dev_bytes, self._pin_bytes, self.device) if self.pooled else None
            self.slots.append(_Slot(self.device, arena))
        self.arena_growths = 0

    # -- one batch ---------------------------------------------------------------------------------
    def _enqueue(self, slot: _Slot, index: int, batch: Batch):
        t0 = time.perf_counter()
        while True:
            try:
                with torch.cuda.stream(slot.stream):
                    graph = runtime.Graph(batch, device=self.device, arena=slot.arena)
                    try:
                        sess = runtime.Session(graph, self.weights, max_rounds=self.rounds, device=self.device,
                                               arena=slot.arena)
                    except Exception:
                        graph.close()
                        raise
                    if len(self.slots) > 1:
                        sess.set_throughput(True)    # batches overlap on the GPU: CU-cycles count
                break
            except _lib.DsatError as ex:
                if slot.arena is None or "arena too small" not in str(ex):
                    raise
                # rare: a batch larger than any before -- drain this slot's stream and re-make its arena
                slot.stream.synchronize()
                need = slot.arena.usage()
                slot.arena.close()
                self._dev_bytes = max(self._dev_bytes, int(need["device_needed"] * 1.25) + (1 << 20))
                self._pin_bytes = max(self._pin_bytes, int(need["pinned_needed"] * 1.25) + (1 << 20))
                slot.arena = runtime.Arena(self._dev_bytes, self._pin_bytes, self.device)
                self.arena_growths += 1
        sess.set_graph_replay(self.graph_replay)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(slot.stream):
            ev0.record(slot.stream)
            out = sess.sample(self.n_steps, self.rounds, seed=self.seed + 1000003 * index + 1,
                              early_exit=self.early_exit, want_solved_fraction=False)
            ev1.record(slot.stream)
        host_ms = 1e3 * (time.perf_counter() - t0)
        slot.pending = (index, batch, graph, sess, out, ev0, ev1, host_ms)

    def _finish(self, slot: _Slot, stats: Dict[str, list], on_result):
        index, batch, graph, sess, out, ev0, ev1, host_ms = slot.pending
        slot.pending = None
        t0 = time.perf_counter()
        slot.stream.synchronize()
        wait_ms = 1e3 * (time.perf_counter() - t0)
        assignment = out["assignment"].cpu().numpy()
        solved = out["solved"].cpu().numpy() > 0
        stats["device_ms"].append(float(ev0.elapsed_time(ev1)))
        stats["host_setup_ms"].append(host_ms)
        stats["host_wait_ms"].append(wait_ms)
        stats["graphs"].append(batch.n_graphs)
        stats["nodes"].append(batch.n_nodes)
        stats["solved"].append(int(solved.sum()))
        sess.close()
        graph.close()
        if slot.arena is not None:
            slot.arena.reset()
        if on_result is not None:
            on_result(index, batch, assignment, solved)

    # -- the stream ----------------------------------------------------------------------------------
    def run(self, batches: Iterable[Batch], on_result: Optional[Callable] = None) -> Dict[str, object]:
        """Sample every batch; ``on_result(index, batch, assignment uint8 [N], solved bool [G])`` is
        called in stream order.  Returns per-batch timing lists and totals."""
        stats: Dict[str, list] = {k: [] for k in ("device_ms", "host_setup_ms", "host_wait_ms", "graphs", "nodes",
                                                   "solved")}
        k = len(self.slots)
        t0 = time.perf_counter()
        n = 0
        for i, batch in enumerate(batches):
            slot = self.slots[i % k]
            if slot.pending is not None:
                self._finish(slot, stats, on_result)
            self._enqueue(slot, i, batch)
            n += 1
        # drain in stream order: the oldest pending batch sits in slot (n mod k)
        for j in range(k):
            slot = self.slots[(n + j) % k]
            if slot.pending is not None:
                self._finish(slot, stats, on_result)
        wall = time.perf_counter() - t0
        out: Dict[str, object] = dict(stats)
        out.update({"batches": n, "wall_s": wall, "denoising_steps": n * self.n_steps,
                    "steps_per_s": n * self.n_steps / wall if wall > 0 else 0.0,
                    "instances": int(np.sum(stats["graphs"])) if n else 0,
                    "arena_growths": self.arena_growths})
        return out

    def close(self):
        for slot in self.slots:
            if slot.pending is not None:
                slot.stream.synchronize()
                _, _, graph, sess, *_ = slot.pending
                sess.close()
                graph.close()
                slot.pending = None
            if slot.arena is not None:
                slot.arena.close()
                slot.arena = None
