"""Run one fused-MLP configuration a few times (for rocprofv3 counter passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from diffusionsat_amd import runtime
from diffusionsat_amd.batch import make_batch
from diffusionsat_amd.instances import config2_instances
from oracle import querysat_oracle as O
mlp, impl, rb, nw = [int(v) for v in sys.argv[1:5]]
batch = make_batch(config2_instances(31))
ws = O.make_weights(0, np.float32, bias_scale=0.05)
s = runtime.Session(runtime.Graph(batch), runtime.Weights(ws))
rows = batch.n_clauses if mlp == 2 else batch.n_vars
print(s.bench_mlp(mlp, rows, impl, rb, nw, iters=5))
