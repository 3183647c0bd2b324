"""Phase breakdown of the fused round kernels from in-kernel clock stamps (DSAT_STAMPS=1)."""
import os, sys
os.environ["DSAT_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from diffusionsat_amd import runtime, _lib, weights as W
from diffusionsat_amd.batch import make_batch
from diffusionsat_amd.instances import config2_instances
batch = make_batch(config2_instances(31))
s = runtime.Session(runtime.Graph(batch), runtime.Weights(W.glorot_weights(0)))
s.set_graph_replay(False)
s.sample(2, 32, seed=1, early_exit=True, want_solved_fraction=False)
torch.cuda.synchronize()
buf = (C.c_int64 * (3 * 4096 * 8))()
_lib.check(_lib.load().dsat_session_read_stamps(s.handle, buf, 3 * 4096 * 8))
st = np.frombuffer(buf, dtype=np.int64).reshape(3, 4096, 8)
names = {0: ["pairnorm+aux", "out_mlp", "query_mlp", "lit_mlp"], 1: ["pairnorm", "gather", "clause_mlp", "colsums"],
         2: ["gather", "update_mlp", "colsums"]}
tiles = {0: (batch.n_vars + 15) // 16, 1: (batch.n_clauses + 63) // 64, 2: (batch.n_vars + 15) // 16}
for k in range(3):
    t = st[k, :tiles[k]]
    nph = len(names[k])
    d = np.diff(t[:, :nph + 1], axis=1).astype(np.float64)
    tot = (t[:, nph] - t[:, 0]).astype(np.float64)
    print("P%d: %d tiles; per-tile cycles median total %.0f (max %.0f)" % (k + 1, tiles[k], np.median(tot), tot.max()))
    for i, nm in enumerate(names[k]):
        print("    %-14s median %8.0f  max %8.0f cycles" % (nm, np.median(d[:, i]), d[:, i].max()))
    if k == 2:   # finer stamps inside the P3 gather (slots 4..6), thread 0's view
        for nm, lo, hi in (("  meta", 0, 4), ("  stage+rows", 4, 5), ("  neighbour loop", 5, 6), ("  finish", 6, 1)):
            dd = (t[:, hi] - t[:, lo]).astype(np.float64)
            print("    %-16s median %8.0f  max %8.0f cycles" % (nm, np.median(dd), dd.max()))
