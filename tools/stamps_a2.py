"""Phase stamps of P1 on 32-row A tiles (role_a2), split by whether the tile runs the output / query pair itself:
    python tools/stamps_a2.py [--config 3]
Prints cycles (median / max): PairNorm, out chain, query chain (with the v1 slicing), lit_mlp, whole tile; role B beside it."""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionsat_amd import _lib, runtime, weights as W
from diffusionsat_amd.batch import make_batch
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=3); a0 = ap.parse_args()
sys.argv = [sys.argv[0]]
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
st = np.frombuffer(buf, dtype=np.int64).reshape(3, 4096, 8)[0]
n32 = (batch.n_vars + 31) // 32
A = st[:n32].astype(np.float64)
pair = (A[:, 2] - A[:, 1]) > 1000          # the out chain took time
def rep(name, T):
    if len(T) == 0: return
    f = lambda lo, hi: "%6.0f/%6.0f" % (np.median(T[:, hi] - T[:, lo]), (T[:, hi] - T[:, lo]).max())
    print("%-28s n=%4d  pairnorm %s  out %s  query+slice %s  lit_mlp %s  tile %s" % (name, len(T), f(0, 1), f(1, 2), f(2, 3), f(3, 4), f(0, 4)))
print(what, "mask", s.split_mask())
rep("A32 with the pair", A[pair]); rep("A32 without", A[~pair])
B = st[n32:]; B = B[B[:, 7] > 0].astype(np.float64); isb = B[:, 2] > 0
rep("role B (64 rows)", B[isb])
