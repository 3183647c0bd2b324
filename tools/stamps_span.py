"""Where a multi-wave launch's time goes: per XCD (blocks with equal blockIdx % 8 share a shader clock) the span from the
first stamp to the last, the sum of the blocks' own durations / CUs, and the distribution of block durations.
    python tools/stamps_span.py --config 5 [--kernel 1]      (kernel 0 = P1, 1 = P2, 2 = P3)"""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionsat_amd import _lib, runtime, weights as W
from diffusionsat_amd.batch import make_batch
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=5); ap.add_argument("--kernel", type=int, default=1)
a0 = ap.parse_args(); sys.argv = [sys.argv[0]]
args = argparse.Namespace(config=a0.config, graphs=31, copies=256, vars=100, irr_vars=10000, irr_clauses=60000, zipf=1.3)
inst, what = bench.make_workload(args, 0)
batch = make_batch(inst)
os.environ["DSAT_STAMPS"] = "1"
s = runtime.Session(runtime.Graph(batch), runtime.Weights(W.glorot_weights(0)), max_rounds=32)
os.environ.pop("DSAT_STAMPS")
s.set_graph_replay(False)
s.sample(1, 32, seed=1, early_exit=True, want_solved_fraction=False)
torch.cuda.synchronize()
buf = (C.c_int64 * (3 * 4096 * 8))()
_lib.check(_lib.load().dsat_session_read_stamps(s.handle, buf, 3 * 4096 * 8))
st = np.frombuffer(buf, dtype=np.int64).reshape(3, 4096, 8)[a0.kernel].astype(np.float64)
used = st[:, 7] > 0
idx = np.nonzero(used)[0]
last = np.max(st[:, :7], axis=1)
dur = last - st[:, 0]
ghz = np.median((dur[used]) / (st[used, 7] * 10.0))
print(what); print("kernel", a0.kernel, "blocks with stamps", used.sum(), "median clock %.2f GHz" % ghz)
for x in range(8):
    m = used & (np.arange(4096) % 8 == x)
    if not m.any(): continue
    span = last[m].max() - st[m, 0].min()
    print("XCD residue %d: %3d blocks, span %7.0f cycles = %5.1f us; sum of durations / 32 CUs %7.0f; block duration median %6.0f p90 %6.0f max %6.0f; first start spread %5.0f"
          % (x, m.sum(), span, span / (ghz * 1e3), dur[m].sum() / 32, np.median(dur[m]), np.percentile(dur[m], 90), dur[m].max(),
             np.percentile(st[m, 0], 10) - st[m, 0].min()))
