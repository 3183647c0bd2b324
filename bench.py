#!/usr/bin/env python3
"""Benchmark of the denoising-sampling hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config {2,3,4,5}] [--repeats R]

A "step" is one denoising step = one QuerySAT.diffusion_step of R=32 message-passing rounds over one
batch, run inside the device-side sampler loop (randomized rounding -> step -> SAT check -> reverse
step).  Inputs are resident in HBM before the timed region; the step runs with the reference's
semantics (early exit armed, per-round logits / map selection / is_batch_sat every round), which with
random-init weights never fires, so the work per step is fixed.

Workloads (BASELINE.json `configs`; --config, default 2 = the configuration the metric is quoted on):
  2  31 distinct solver-filtered threshold 3-SAT instances n=100 m=428 = 19 468 nodes       (T=32)
  3  one such instance x 256 copies = 160 768 nodes                                        (T=64)
  4  NeuroSAT-style SR(n) mixed-width k-SAT stream, n in [30,100]: instances -> node-budget batches
     -> per-GPU shards -> pooled sessions, host set-up of batch i+1 overlapped with batch i   (T=32)
  5  one irregular-degree CNF of 10 000 variables / 60 000 clauses (SHA-like stand-in)      (T=128)

N > 1: one process per GPU.  When the driver starts the ranks itself (torch.distributed.run sets
WORLD_SIZE) this process is one rank; when WORLD_SIZE is unset, `--gpus N` makes THIS process the
launcher: before anything touches a GPU it starts N fresh child processes (RANK/LOCAL_RANK/WORLD_SIZE/
MASTER_* set), waits for them, relays rank 0's single JSON line and exits non-zero if any child
failed.  Every rank owns its own batch (weak scaling), the weights are broadcast from rank 0 over RCCL
and there is no communication in the timed region.

The timed region (K steps between barrier + synchronize on both sides, MAX over ranks) is run
`--repeats` times (default 5); `value` comes from the median, the line carries min/max as well.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA = vector fp32 peak
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak
# Kernels that form an fp32 product from six bf16-slice products (csrc/split_bf16.h) run on the bf16 matrix pipe:
# dense bf16 MFMA peak (16x the fp32-input rate, same guide) / 6 = the fp32-equivalent ceiling of that form
PEAK_MFMA_BF16_TFLOPS = 16 * PEAK_MFMA_F32_TFLOPS
PEAK_SPLIT_F32EQ_TFLOPS = PEAK_MFMA_BF16_TFLOPS / 6

DEFAULT_STEPS = {2: 32, 3: 64, 4: 32, 5: 128}     # one full sample of the config's T


def gemm_flops_per_round(n_vars, n_clauses):
    """Algorithmic flops of the 12 Dense layers of one round."""
    return 1_446_966 * n_vars + 261_120 * n_clauses


def sparse_bytes_per_round(n_vars, n_clauses, n_edges, f=128):
    """Unique-traffic bytes of the four sparse products of one round."""
    return 4 * f * (7 * n_vars + 4 * n_clauses) + 16 * n_edges + 8 * (n_clauses + 1) + 8 * (2 * n_vars + 1)


# ------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n_ranks):
    """Launcher role: start `n_ranks` fresh python processes running this file as ranks.  Nothing here
    touches a GPU (no torch import even), so no process that has initialised HIP is ever re-executed."""
    env0 = dict(os.environ)
    env0["MASTER_ADDR"] = "127.0.0.1"
    env0["MASTER_PORT"] = str(_free_port())
    env0["WORLD_SIZE"] = str(n_ranks)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n_ranks):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    line = ""
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = ln
    if any(rcs) or not line:
        sys.stderr.write("bench.py launcher: rank exit codes %s%s\n" % (rcs, "" if line else "; no JSON line from rank 0"))
        return 1
    print(line)
    return 0


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(instances, ws, rounds, nodes, gpu_step=None):
    """CPU restatement (oracle/querysat_torch.py: PyTorch-CPU fp32, TensorFlow being unavailable) timed
    on this host as SURVEY.md §8d prescribes: 1 thread and all host cores, 1 warm-up then the median
    of 3; plus the fastest thread count of a short probe (sparse products of a 20k-node batch stop
    scaling long before all cores of a large host are used).  Bounded sample: the first `rs` of the
    step's `rounds` message-passing rounds (rounds cost the same), scaled to whole steps."""
    import torch
    from oracle import querysat_torch as QT
    tb = QT.TorchBatch(instances)
    tw = QT.to_torch_weights(ws)
    rng = np.random.default_rng(0)
    n = tb.n
    rs = max(1, min(rounds, int(round(8 * 19468 / max(nodes, 1)))))
    noisy = np.stack([np.ones(n), np.zeros(n)], 1).astype(np.float32)
    normal = rng.standard_normal((rs, n, 4)).astype(np.float32)
    labels = rng.integers(0, 2, n)
    ncpu = int(torch.get_num_threads())

    def steps_per_s(threads, reps=3, nr=rs):
        torch.set_num_threads(threads)
        with torch.no_grad():
            QT.diffusion_step(tw, tb, 0.5, noisy, normal[:nr], labels, nr)          # warm-up
            ts = []
            for _ in range(reps):
                c0 = time.perf_counter()
                QT.diffusion_step(tw, tb, 0.5, noisy, normal[:nr], labels, nr)
                ts.append(time.perf_counter() - c0)
        med = float(np.median(ts))
        return {"value": nr / rounds / med, "threads": threads, "seconds": [round(t, 4) for t in ts]}

    t_all = time.perf_counter()
    probe = {}
    for t in sorted({1, 4, 8, 16, 32, 64, ncpu}):
        if t <= ncpu:
            probe[t] = steps_per_s(t, reps=1, nr=min(2, rs))["value"]
    best_t = max(probe, key=probe.get)
    one, allc, best = steps_per_s(1), steps_per_s(ncpu), steps_per_s(best_t)
    check = None
    if gpu_step is not None:
        # the baseline leg is also the checker: the HIP path's logits after the same `rs` rounds on the same
        # injected draws against this restatement's (two fp32 computations with different summation orders)
        torch.set_num_threads(best_t)
        with torch.no_grad():
            ref = QT.diffusion_step(tw, tb, 0.5, noisy, normal, labels, rs)["logits"].numpy()
        got = gpu_step(0.5, noisy, normal, labels.astype(np.int32), rs)
        check = {"rounds": rs, "max_abs_logit_diff_gpu_vs_cpu_port": float(np.abs(got - ref).max()),
                 "max_abs_logit": float(np.abs(ref).max()), "tolerance_north_star": 1e-3}
    torch.set_num_threads(ncpu)
    return {"value": best["value"], "unit": "denoising-steps/s", "cores": best_t, "kind": "port", "parity_check": check,
            "one_thread": one, "all_cores": allc, "best_of_probe": best, "host_cores": ncpu,
            "sample": "the first %d of the %d message-passing rounds of one denoising step of the same batch "
                      "(scaled to whole steps), PyTorch-CPU fp32 restatement, early exit off; 1 warm-up then "
                      "median of 3 at 1 thread, at all %d cores and at the fastest thread count of a probe over "
                      "{1,4,8,16,32,64,all}; %.1f s in all" % (rs, rounds, ncpu, time.perf_counter() - t_all)}


# ------------------------------------------------------------------------------------------ PMC traffic files
def workload_key(args) -> str:
    """What a PMC traffic file must have been measured on to describe this run's launches."""
    if args.config == 2:
        return "cfg2:%dx%d:R%d" % (args.graphs, args.vars, args.rounds)
    if args.config == 3:
        return "cfg3:%dx%d:R%d" % (args.copies, args.vars, args.rounds)
    if args.config == 5:
        return "cfg5:%d/%d:zipf%.2f:R%d" % (args.irr_vars, args.irr_clauses, args.zipf, args.rounds)
    return "cfg%d" % args.config


def load_traffic(key: str):
    """({kernel class: HBM bytes per launch}, file name) from the newest profiles/r*_pmc_traffic*.json whose
    `csrc_hash` is this build's and whose `workload_key` is this run's (tools/pmc_traffic.py writes both);
    ({}, None) when there is none -- `roofline.traffic` is then null."""
    import glob
    from diffusionsat_amd import build
    h = build.source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get("csrc_hash") == h and d.get("workload_key") == key:
                return {k: v["hbm_bytes_per_launch"] for k, v in d["kernels"].items()}, os.path.basename(path)
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return {}, None


# ------------------------------------------------------------------------------------------ workloads
def make_workload(args, rank):
    """Instances of this rank's batch (configs 2, 3, 5) and the workload description."""
    from diffusionsat_amd.instances import config2_instances, config3_instances, irregular_cnf
    if args.config == 2:
        inst = config2_instances(args.graphs, args.vars, seed=2 + rank)
        what = ("BASELINE config 2: %d satisfiable threshold 3-SAT instances n=%d m=%d per GPU"
                % (len(inst), args.vars, len(inst[0][1])))
    elif args.config == 3:
        inst = config3_instances(args.copies, args.vars, seed=2 + rank)
        what = ("BASELINE config 3: one satisfiable threshold 3-SAT instance n=%d m=%d x %d copies per GPU"
                % (args.vars, len(inst[0][1]), len(inst)))
    elif args.config == 5:
        inst = [irregular_cnf(args.irr_vars, args.irr_clauses, seed=5 + rank, zipf_a=args.zipf)[:2]]
        what = ("BASELINE config 5: one irregular-degree CNF per GPU (SHA-1-preimage stand-in: clause lengths 2..8 "
                "with a heavy tail, power-law variable popularity zipf_a=%.2f), %d vars / %d clauses"
                % (args.zipf, args.irr_vars, args.irr_clauses))
    else:
        raise SystemExit("unknown --config %r" % args.config)
    return inst, what


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed denoising steps K (default: the config's T)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--repeats", type=int, default=None,
                    help="how many times the K-step timed region is run (default 5; 3 for the config-4 stream)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json workload")
    ap.add_argument("--rounds", type=int, default=32)
    ap.add_argument("--graphs", type=int, default=31, help="config 2: instances per batch (31 x n=100)")
    ap.add_argument("--copies", type=int, default=256, help="config 3: copies of the one instance")
    ap.add_argument("--vars", type=int, default=100)
    ap.add_argument("--irr-vars", type=int, default=10000, help="config 5")
    ap.add_argument("--irr-clauses", type=int, default=60000, help="config 5")
    ap.add_argument("--zipf", type=float, default=1.3, help="config 5: popularity exponent (1.1 = heavier tail)")
    ap.add_argument("--stream-instances", type=int, default=12500,
                    help="config 4: instances of the stream each rank processes (the full config: 100 000 over 8 GPUs = "
                         "12 500 per GPU, about 20 s of device time per pass)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures (fp32_mfma_only, throughput_mode): under rocprofv3 they would mix "
                         "other forms of the same kernels into the per-kernel averages")
    ap.add_argument("--eager", action="store_true", help="disable hipGraph replay")
    ap.add_argument("--batches-in-flight", type=int, default=1,
                    help="opt-in throughput mode: this many independent batches per GPU, each on its own HIP "
                         "stream (the K steps are dealt round-robin over them); the default, 1, is the contract's "
                         "one batch per GPU")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = DEFAULT_STEPS[args.config]
    if args.repeats is None:
        args.repeats = 3 if args.config == 4 else 5

    # ---- launcher role: before any GPU / torch.cuda call ------------------------------------------
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    # Rehearsal on a one-GPU box: DSAT_BENCH_BACKEND=gloo lets several ranks share cuda:0 (RCCL
    # refuses two ranks on one device); the driver's multi-GPU runs use the default, nccl = RCCL.
    backend = os.environ.get("DSAT_BENCH_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    if backend == "nccl" and world > ndev:
        raise SystemExit("--gpus %d but only %d GPU(s) visible" % (world, ndev))
    local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from diffusionsat_amd import build, distributed, runtime, weights as W
    from diffusionsat_amd.batch import make_batch
    build.build()

    # ---- weights: rank 0's blob to every rank (RCCL over xGMI; the only collective on the path) ------
    ws = W.glorot_weights(seed=0)
    blob = torch.from_numpy(W.to_blob(ws)).to(dev)
    if dist is not None:
        if rank != 0:
            blob.zero_()
        if backend == "nccl":
            distributed.broadcast_weight_blob(blob, src=0)
        else:                                  # gloo rehearsal: broadcast through the host
            hb = blob.cpu()
            distributed.broadcast_weight_blob(hb, src=0)
            blob.copy_(hb)
    wts = runtime.Weights(blob, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def gather_floats(x):
        """Every rank's value of a host float, on every rank."""
        if dist is None:
            return [float(x)]
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    if args.config == 4:
        from bench_stream import run_stream_bench
        result = run_stream_bench(args, rank, world, dev, wts, ws, barrier, gather_floats)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(result))
        return

    # ---- inputs: one batch per rank (weak scaling) -------------------------------------------------
    instances, what = make_workload(args, rank)
    batch = make_batch(instances)
    graph = runtime.Graph(batch, device=dev)
    sess = runtime.Session(graph, wts, max_rounds=args.rounds, device=dev)
    sess.set_graph_replay(not args.eager)
    nflight = max(1, args.batches_in_flight)
    extra = []   # (session, stream) of the other batches in flight; each has its own instances
    for b in range(1, nflight):
        a2 = argparse.Namespace(**vars(args))
        bb = make_batch(make_workload(a2, 1000 * b + rank)[0])
        sb = runtime.Session(runtime.Graph(bb, device=dev), wts, max_rounds=args.rounds, device=dev)
        sb.set_graph_replay(not args.eager)
        extra.append((sb, torch.cuda.Stream(device=dev)))
    if args.steps % nflight != 0:
        raise SystemExit("--steps must be a multiple of --batches-in-flight")

    def run(n_steps, seed):
        per = n_steps // nflight if nflight > 1 else n_steps
        cur = torch.cuda.current_stream()
        for sb, stream in extra:   # sample() only enqueues work: the batches overlap on the device
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                sb.sample(max(per, 1), args.rounds, seed=seed + 17, early_exit=True, want_solved_fraction=False)
        out = sess.sample(max(per, 1), args.rounds, seed=seed, early_exit=True, want_solved_fraction=False)
        for _, stream in extra:
            cur.wait_stream(stream)
        return out

    run(nflight, seed=0)       # untimed: first-use work (hipGraph capture, allocations) whatever W is
    if args.warmup > 0:
        run(args.warmup, seed=1)
    # ---- the timed region, `repeats` times: EXACTLY K steps between barrier + synchronize ------------
    elapsed_all, elapsed_mine = [], []
    out = None
    for rep in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        out = run(args.steps, seed=2 + rep)
        barrier()
        el = time.perf_counter() - t0
        elapsed_mine.append(el)
        elapsed_all.append(max(gather_floats(el)))                  # MAX over ranks
    elapsed = float(np.median(elapsed_all))
    per_rank = gather_floats(args.steps / float(np.median(elapsed_mine)))
    solved_fraction = float(out["solved"].float().mean().item())

    result = None
    if rank == 0:
        n, c, e = batch.n_vars, batch.n_clauses, batch.n_edges
        value = world * args.steps / elapsed
        ms = [1e3 * x / args.steps for x in elapsed_all]
        metric = "denoising-steps/sec, 20k-node 3-SAT batch" if args.config == 2 else \
                 "denoising-steps/sec, BASELINE config %d batch" % args.config
        result = {
            "metric": metric,
            "value": value, "unit": "denoising-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # f32 in, f32 out, f32 accumulation everywhere; where arithmetic_note says so, an MLP's f32 products are
            # formed exactly-to-f32-rounding from bf16 slices (same tolerances against the f64 oracle as the f32 MFMA)
            "arithmetic_note": ("clause_mlp" if sess.split_mask() & 1 else "") + (" update_gate" if sess.split_mask() & 2 else "")
                               + (" lit_mlp" if sess.split_mask() & (4 | 32) else "")
                               + (": fp32 products as six bf16-slice products on v_mfma_f32_16x16x32_bf16 (csrc/split_bf16.h); "
                                  "all else v_mfma_f32_16x16x4_f32" if sess.split_mask() else "v_mfma_f32_16x16x4_f32 everywhere"),
            # the metric's second half needs trained weights: the released checkpoints are not in the container
            # (GitHub Releases only, /root/reference/README.md:38-40), so with random-init weights it is UNMEASURED
            # (null), not zero; what the random-init run produced is kept beside it for the record
            "solved_fraction": None, "solved_fraction_random_init": solved_fraction,
            "repeats": {"n": len(ms), "ms_per_step_median": float(np.median(ms)), "ms_per_step_min": min(ms),
                        "ms_per_step_max": max(ms), "ms_per_step_all": [round(x, 5) for x in ms],
                        "value_min": world * args.steps / max(elapsed_all),
                        "value_max": world * args.steps / min(elapsed_all)},
            "per_rank_steps_per_s": [round(x, 3) for x in per_rank],
            "config": {"workload": "%s (N=%d vars, C=%d clauses, E=%d edges, %d nodes), R=%d rounds per denoising "
                                   "step, K=%d timed steps x %d repeats, random-init Glorot weights, device Philox "
                                   "noise, early exit armed (never fires)"
                                   % (what, n, c, e, batch.n_nodes, args.rounds, args.steps, len(ms)),
                       "baseline_config": args.config, "rounds": args.rounds, "nodes_per_gpu": batch.n_nodes,
                       "parallelism": "batch-sharded x%d" % world, "hip_graph_replay": not args.eager,
                       "batches_in_flight_per_gpu": nflight,
                       "max_clause_len": int(np.diff(batch.clause_rowptr).max()) if c else 0},
        }
        # ---- per-kernel device times, live (HIP events on the launch stream, eager) -------------
        if not args.no_profile:
            sess.set_profile(True)
            run(2 * nflight, seed=3)
            prof = sess.profile()
            sess.set_profile(False)
            steps_prof = 2
            all_us = sum(v["avg_us"] * v["launches"] for v in prof.values()) / steps_prof
            # algorithmic flops per launch of each fused round kernel (SURVEY.md §8d per-row MACs)
            # P1 runs R+1 times per step: rounds 0..R-1 carry query + lit, rounds 1..R carry the output MLP
            # of the previous round (the last launch is that tail alone) -> average over the R+1 launches
            r_ = args.rounds
            per_launch = {
                "round_p1_var_query_lit": 2.0 * n * (r_ * (40_545 + 463_360) + r_ * 17_408) / (r_ + 1),
                "round_p2_clause": 2.0 * c * 130_560,                               # clause MLP
                "round_p3_var_update": 2.0 * n * 202_170,                           # update gate
            }
            kern_name = {"round_p1_var_query_lit": "round_p1_kernel", "round_p2_clause": "round_p2_kernel<4>",
                         "round_p3_var_update": "round_p3_kernel"}
            # HBM bytes per launch from the PMC passes kept under profiles/ (rocprofv3 cannot run inside this
            # process).  A traffic file is tied to the kernels and the workload it was taken on: it carries the hash
            # of csrc/ (build.source_hash) and the workload key; anything else reads as null, never as a stale number
            traffic, traffic_src = load_traffic(workload_key(args))
            roofs = {}
            mask = sess.split_mask()
            if mask & 64:   # role D: P2's extra workgroups form the `variables` part of update_gate's first layer
                moved = 2.0 * n * 128 * 230          # 128 state columns x 230 hidden units per variable
                per_launch["round_p3_var_update"] -= moved
                per_launch["round_p2_clause"] += moved
            split = {"round_p2_clause": bool(mask & 1), "round_p3_var_update": bool(mask & 2)}
            for k, fl in per_launch.items():
                if k in prof and prof[k]["launches"] > 0:
                    us = prof[k]["avg_us"]
                    ach = fl / (us * 1e-6) / 1e12
                    sp = split.get(k, False)
                    peak = PEAK_SPLIT_F32EQ_TFLOPS if sp else PEAK_MFMA_F32_TFLOPS
                    a2 = k == "round_p1_var_query_lit" and bool(mask & (4 | 32))
                    if a2:   # lit_mlp as split products (32-row tiles: bit 2, 16-row tiles: bit 5), variables_query too when
                        # bit 3 is set, variables_output on the fp32 MFMA: the ceiling is the time-weighted mix of the two
                        # pipes' peaks (MACs per row: SURVEY.md §8d)
                        sp_macs = 463_360 + (40_545 if mask & 8 else 0)
                        f32_macs = 17_408 + (0 if mask & 8 else 40_545)
                        peak = (sp_macs + f32_macs) / (sp_macs / PEAK_SPLIT_F32EQ_TFLOPS + f32_macs / PEAK_MFMA_F32_TFLOPS)
                    roofs[k] = {"bound": "mfma", "kernel": kern_name[k].replace("_kernel<4>", "s_kernel").replace("p3_kernel", "p3s_kernel") if sp else kern_name[k],
                                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                                # the same algorithmic fp32 flops against the fp32-input MFMA peak (what a kernel that
                                # keeps v_mfma_f32_16x16x4_f32 is priced on; the round-1 fractions were of this peak)
                                "frac_of_fp32_mfma_peak": ach / PEAK_MFMA_F32_TFLOPS,
                                "pipe": ("v_mfma_f32_16x16x32_bf16, six bf16-slice products per fp32 product: achieved = "
                                         "algorithmic fp32 flops / time, peak = dense bf16 MFMA peak %.0f / 6"
                                         % PEAK_MFMA_BF16_TFLOPS) if sp else
                                        ("lit_mlp%s: v_mfma_f32_16x16x32_bf16 x6 per fp32 product on %d-row tiles; variables_output%s: "
                                         "v_mfma_f32_16x16x4_f32; peak = the two pipes' ceilings weighted by their flops"
                                         % (" and variables_query" if mask & 8 else "", 32 if mask & 4 else 16,
                                            "" if mask & 8 else " / variables_query")) if a2
                                        else "v_mfma_f32_16x16x4_f32",
                                "traffic": traffic.get(k), "traffic_source": traffic_src if k in traffic else None,
                                "avg_launch_us": us,
                                "launches_per_step": prof[k]["launches"] / steps_prof,
                                "algorithmic_flops_per_launch": fl,
                                "share_of_device_time": us * prof[k]["launches"] / steps_prof / all_us}
            if roofs:
                dom = max(roofs, key=lambda k: roofs[k]["share_of_device_time"])
                result["roofline"] = roofs[dom]
                result["roofline_kernels"] = roofs
            else:   # per-layer path (DSAT_FUSED=0)
                gemm = {k: v for k, v in prof.items() if k.startswith("gemm_") and v["launches"]}
                gemm_us = sum(v["avg_us"] * v["launches"] for v in gemm.values()) / steps_prof
                gemm_launches = sum(v["launches"] for v in gemm.values()) / steps_prof
                flops_step = gemm_flops_per_round(n, c) * args.rounds
                ach = flops_step / (gemm_us * 1e-6) / 1e12
                result["roofline"] = {
                    "bound": "mfma", "kernel": "gemm_f32_mfma_kernel", "achieved": ach, "peak": PEAK_MFMA_F32_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_F32_TFLOPS, "traffic": None,
                    "launches_per_step": gemm_launches, "avg_launch_us": gemm_us / gemm_launches,
                    "algorithmic_flops_per_launch": flops_step / gemm_launches,
                    "share_of_device_time": gemm_us / all_us}
            result["kernel_time_us_per_step"] = {k: v["avg_us"] * v["launches"] / steps_prof
                                                 for k, v in prof.items() if v["launches"]}
        if world == 1 and not args.no_profile and "roofline_kernels" in result:
            # The sparse half-rounds run inside P2 / P3: their duration comes from in-kernel clock stamps
            # (a diagnostic session, DSAT_STAMPS=1), converted with the clock the same kernels ran at
            # (stamped tile cycles / event duration).  Bytes: SURVEY.md §8d unique-traffic model.
            # tools/stamps.py writes the same stamps (and the per-phase table) under profiles/.
            try:
                import ctypes as C
                from diffusionsat_amd import _lib
                os.environ["DSAT_STAMPS"] = "1"
                s3 = runtime.Session(graph, wts, max_rounds=args.rounds, device=dev)
                os.environ.pop("DSAT_STAMPS", None)
                s3.set_graph_replay(False)
                s3.sample(1, args.rounds, seed=9, early_exit=True, want_solved_fraction=False)
                torch.cuda.synchronize()
                buf = (C.c_int64 * (3 * 4096 * 8))()
                _lib.check(_lib.load().dsat_session_read_stamps(s3.handle, buf, 3 * 4096 * 8))
                st = np.frombuffer(buf, dtype=np.int64).reshape(3, 4096, 8)
                t2 = st[1, :min((c + 63) // 64, 4096)].astype(np.float64)
                t3 = st[2, :min((n + 15) // 16, 4096)].astype(np.float64)
                g2, tot2 = np.median(t2[:, 2] - t2[:, 1]), np.median(t2[:, 4] - t2[:, 0])
                g3, tot3 = np.median(t3[:, 1] - t3[:, 0]), np.median(t3[:, 3] - t3[:, 0])
                rk = result["roofline_kernels"]
                # a kernel lasts (tiles / 256 CUs, rounded up) tile times: the gather phase's share of it
                w2 = -(-((c + 63) // 64) // 256)
                w3 = -(-((n + 15) // 16) // 256)
                us2 = g2 / tot2 * rk["round_p2_clause"]["avg_launch_us"]
                us3 = g3 / tot3 * rk["round_p3_var_update"]["avg_launch_us"]
                nbytes = sparse_bytes_per_round(n, c, e)
                ach = nbytes / ((us2 + us3) * 1e-6) / 1e9
                result["roofline_sparse"] = {
                    "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                    "algorithmic_bytes_per_round": nbytes, "gather_us_per_round": {"p2_literal_to_clause": us2,
                                                                                   "p3_clause_to_literal": us3},
                    "gather_cycles_per_tile": {"p2": float(g2), "p3": float(g3)},
                    "tile_waves": {"p2": w2, "p3": w3},
                    "note": "the four sparse products of a round (R3, R4, R7, R11) are the gather phases of the fused "
                            "P2 and P3 kernels; duration = that phase's share of the tile's stamped cycles x the "
                            "kernel's event duration; the rows are served by L2 / Infinity Cache (state written by the "
                            "previous kernel), so this is the sparse phases' rate against the HBM roofline, not HBM "
                            "traffic (see roofline.traffic for that)"}
                del s3
            except Exception as ex:      # diagnostics only: never fail the bench line
                result["roofline_sparse"] = {"error": str(ex)}
        if world == 1 and nflight == 1 and not args.no_profile and not args.no_secondary and sess.split_mask() != 0:
            # secondary figure, same run, same batch: every MLP on v_mfma_f32_16x16x4_f32 (no product is formed from
            # bf16 slices) -- the number for a reader who wants the fp32 matrix instruction only; never `value`
            keys = ("DSAT_P2_SPLIT", "DSAT_P3_SPLIT", "DSAT_P1_A2", "DSAT_P1_BSPLIT", "DSAT_P1_ASPLIT")
            old_env = {k: os.environ.get(k) for k in keys}
            os.environ.update({k: "0" for k in keys})
            try:
                sf = runtime.Session(runtime.Graph(batch, device=dev), wts, max_rounds=args.rounds, device=dev)
            finally:
                for k, v in old_env.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            sf.set_graph_replay(not args.eager)
            sf.sample(4, args.rounds, seed=7, early_exit=True, want_solved_fraction=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sf.sample(args.steps, args.rounds, seed=8, early_exit=True, want_solved_fraction=False)
            torch.cuda.synchronize()
            elf = time.perf_counter() - t1
            result["fp32_mfma_only"] = {"value": args.steps / elf, "unit": "denoising-steps/s", "ms_per_step": 1e3 * elf / args.steps,
                                        "split_mask": sf.split_mask()}
            del sf
        if world == 1 and nflight == 1 and not args.no_profile and not args.no_secondary and args.steps % 2 == 0 and args.config == 2:
            # secondary figure, same run: two independent config-2 batches in flight on two HIP streams
            # (what samples(n) can do when n needs more than one batch); never the headline `value`
            b2 = make_batch(make_workload(args, 1002 + rank)[0])
            s2 = runtime.Session(runtime.Graph(b2, device=dev), wts, max_rounds=args.rounds, device=dev)
            s2.set_graph_replay(not args.eager)
            extra.append((s2, torch.cuda.Stream(device=dev)))
            nflight = 2
            sess.set_throughput(True)    # several batches in flight: CU-cycles count, not one batch's latency
            s2.set_throughput(True)
            run(4, seed=5)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run(args.steps, seed=6)
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t1
            result["throughput_mode"] = {"batches_in_flight_per_gpu": 2, "value": args.steps / el2,
                                         "unit": "denoising-steps/s", "ms_per_step": 1e3 * el2 / args.steps}
            nflight = 1
            extra.clear()
            sess.set_throughput(False)
        if world == 1 and not args.no_cpu_baseline:
            def gpu_step(noise_scale, noisy, normal, labels, nr):
                d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                o = sess.diffusion_step(noise_scale, d(noisy), d(normal), d(labels), rounds=nr, early_exit=False,
                                        want_logits=True)
                return o["logits"].cpu().numpy()
            result["cpu_baseline"] = cpu_baseline(instances, ws, args.rounds, batch.n_nodes, gpu_step)
            result["cpu_baseline"]["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
