"""A/B of the MLP implementations at config-2 sizes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from diffusionsat_amd import runtime, weights as W, build
from diffusionsat_amd.batch import make_batch
from diffusionsat_amd.instances import config2_instances
from oracle import querysat_oracle as O
if not os.environ.get('DSAT_LIB'): build.build()
ONLY = [int(v) for v in os.environ.get('MLPS', '0,1,2,3,4').split(',')]
insts = config2_instances(int(os.environ.get("G", "31")))
batch = make_batch(insts)
ws = O.make_weights(0, np.float32, bias_scale=0.05)
s = runtime.Session(runtime.Graph(batch), runtime.Weights(ws))
flops = {0: 40545, 1: 463360, 2: 130560, 3: 202170, 4: 17408}
for mlp in ONLY:
    rows = batch.n_clauses if mlp == 2 else batch.n_vars
    name = O.MLP_ORDER[mlp]
    x = np.random.default_rng(mlp).standard_normal((rows, ws[name][0][0].shape[0])).astype(np.float32)
    ref = O.mlp(x.astype(np.float64), [(w.astype(np.float64), b.astype(np.float64)) for w, b in ws[name]])
    xd = torch.from_numpy(x).cuda()
    line = ["%-18s rows=%5d" % (name, rows)]
    for impl, rb, nw in [(0, 1, 4), (1, 1, 4), (1, 1, 8), (1, 2, 4), (1, 2, 8), (1, 4, 8)]:
        try:
            y = s.op_mlp(mlp, xd, impl, rb, nw).cpu().numpy()
        except Exception as ex:
            line.append("[%d,rb%d,w%d] n/a" % (impl, rb, nw))
            continue
        err = np.abs(y - ref).max()
        us = s.bench_mlp(mlp, rows, impl, rb, nw, iters=30)
        tf = 2 * flops[mlp] * rows / us / 1e6
        line.append("[%d,rb%d,w%d] %6.1fus %5.1fTF e=%.0e" % (impl, rb, nw, us, tf, err))
    print("  ".join(line), flush=True)
