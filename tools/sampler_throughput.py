"""Throughput of DiffusionSampler.samples() with 1, 2, 3 batches in flight (run on the GPU box).
Random-init weights never solve a threshold instance, so the decoder is replaced by a counter
that lets exactly N_BATCHES batches through (the sampling work itself is unchanged)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionsat_amd import instances as I
from diffusionsat_amd.diffusion_sampler import DiffusionSampler
from diffusionsat_amd.dimacs import DimacsFile

N_BATCHES = 12
n, clauses, _ = I.threshold_3sat(100, seed=7)
DimacsFile("/tmp/f.cnf", clauses=clauses).store()
for k in (1, 2, 3):
    s = DiffusionSampler(None, "/tmp/f.cnf")
    s.batches_in_flight = k
    s.samples(1)                      # warm-up: graph capture, first import
    seen = [0]

    def count_only(packed, solved, copies, counts, need):
        seen[0] += 1
        return len(solved), len(solved), (0 if seen[0] >= N_BATCHES else need)

    s._decode_flags = count_only
    t0 = time.perf_counter()
    s.samples(10 ** 9)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("in flight %d: %d batches x 32 steps in %.3f s = %.1f denoising-steps/s" % (k, seen[0], el, seen[0] * 32 / el))
