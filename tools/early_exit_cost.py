import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from diffusionsat_amd import runtime, weights as W
from diffusionsat_amd.batch import make_batch
from diffusionsat_amd.instances import config2_instances
batch = make_batch(config2_instances(31))
s = runtime.Session(runtime.Graph(batch), runtime.Weights(W.glorot_weights(0)))
for ee in (True, False):
    s.sample(4, 32, seed=1, early_exit=ee, want_solved_fraction=False); torch.cuda.synchronize()
    t0 = time.perf_counter(); s.sample(64, 32, seed=2, early_exit=ee, want_solved_fraction=False); torch.cuda.synchronize()
    print("early_exit", ee, 64 / (time.perf_counter() - t0), "steps/s")
