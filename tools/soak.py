"""Soak: repeated sampler runs at several batch shapes; every repeat must reproduce the first
bit for bit (catches rare races / stale-state bugs that a single test run can miss)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionsat_amd import runtime, weights as W
from diffusionsat_amd.batch import make_batch
from diffusionsat_amd import instances as I

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
shapes = {
    "config2": I.config2_instances(31),
    "config3-64": I.config3_instances(64),
    "irregular": [I.irregular_cnf(1500, 9000, seed=5, zipf_a=1.1)[:2], I.irregular_cnf(700, 3000, seed=6)[:2]],
    "mixed-ksat": [I.sr_ksat(seed=s, min_vars=20, max_vars=60) for s in range(12)],
    "tiny": [(3, [[1, -2], [2, 3]])],
    # heavy literal rows and heavy clauses (split pieces + heavy_gather), level-2 PairNorm sums, select parts
    "heavy": [I.irregular_cnf(9000, 40000, seed=7, zipf_a=1.1)[:2]],
    "long-clauses": [(40, [[(i % 40 + 1) * (1 if (i * 7 + j) % 3 else -1) for i in range(j, j + 33 + 5 * j)] for j in range(12)])],
}
w = runtime.Weights(W.glorot_weights(0))
t_end = time.time() + budget
ref, runs = {}, 0
while time.time() < t_end:
    for name, insts in shapes.items():
        s = runtime.Session(runtime.Graph(make_batch(insts)), w, max_rounds=32)
        for replay in (True, False):
            s.set_graph_replay(replay)
            out = s.sample(3, 32, seed=11, early_exit=True, want_predictions=True, want_solved_fraction=False)
            torch.cuda.synchronize()
            key = (name,)
            sig = (out["assignment"].cpu().numpy().tobytes(), out["predictions"].cpu().numpy().tobytes())
            assert torch.isfinite(out["predictions"]).all(), name
            if key in ref:
                assert ref[key] == sig, "mismatch at %s (replay=%s, run %d)" % (name, replay, runs)
            else:
                ref[key] = sig
        runs += 1
    print("ok after %d session runs" % runs, flush=True)
print("soak passed: %d session runs, all reproduce their first result" % runs)
