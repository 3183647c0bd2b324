"""How many independent config-2 batches in flight does one MI355X want?  Each batch has its own session and HIP stream;
K denoising steps are dealt over them.  Prints steps/s for 1 .. 4 batches with the throughput planning on and off.
    python tools/inflight_probe.py [--steps 48]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionsat_amd import runtime, weights as W
from diffusionsat_amd.batch import make_batch
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=48); ap.add_argument("--config", type=int, default=2)
a0 = ap.parse_args()
args = argparse.Namespace(config=a0.config, graphs=31, copies=256, vars=100, irr_vars=10000, irr_clauses=60000, zipf=1.3)
wts = runtime.Weights(W.glorot_weights(0))
sess, streams = [], []
for b in range(4):
    bb = make_batch(bench.make_workload(args, 1000 * b)[0])
    s = runtime.Session(runtime.Graph(bb), wts, max_rounds=32)
    s.set_graph_replay(True)
    sess.append(s); streams.append(torch.cuda.Stream())
def run(n, k, seed):
    cur = torch.cuda.current_stream()
    for s, st in zip(sess[:n], streams[:n]):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            s.sample(k // n, 32, seed=seed, early_exit=True, want_solved_fraction=False)
    for st in streams[:n]:
        cur.wait_stream(st)
for thr in (0, 1):
    for s in sess: s.set_throughput(bool(thr))
    for n in (1, 2, 3, 4):
        k = a0.steps // 12 * 12
        run(n, 12, 1); torch.cuda.synchronize()
        best = 0.0
        for rep in range(3):
            t = time.perf_counter(); run(n, k, 2 + rep); torch.cuda.synchronize()
            best = max(best, k / (time.perf_counter() - t))
        print("throughput planning %d, %d in flight: %.1f steps/s" % (thr, n, best), flush=True)
