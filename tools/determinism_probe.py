"""Run-to-run determinism probe at config-3 size (and config 2): the same sampler call N times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffusionsat_amd import runtime, weights as W
from diffusionsat_amd.batch import make_batch
from diffusionsat_amd.instances import config3_instances, config2_instances
for name, insts, T in (("cfg3", config3_instances(256, 100, seed=2), 16), ("cfg2", config2_instances(31), 32)):
    s = runtime.Session(runtime.Graph(make_batch(insts)), runtime.Weights(W.glorot_weights(0)), max_rounds=32)
    ref = None
    bad = 0
    for i in range(8):
        out = s.sample(T, 32, seed=11, early_exit=True, want_predictions=True, want_solved_fraction=False)
        torch.cuda.synchronize()
        p = out["predictions"].clone()
        if ref is None:
            ref = p
        elif not torch.equal(ref, p):
            bad += 1
            d = (ref - p).abs()
            first = int((d.amax(dim=1) > 0).nonzero()[0])
            print(name, "run", i, "differs: first step", first, "max", float(d.max()), "n", int((d > 0).sum()))
    print(name, "mismatching runs:", bad)
