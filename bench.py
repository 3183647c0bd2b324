#!/usr/bin/env python3
"""Benchmark of the denoising-sampling hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one denoising step = one QuerySAT.diffusion_step of R=32 message-passing
rounds over one 20k-node batch (BASELINE config 2: 31 distinct solver-filtered threshold 3-SAT instances,
n=100, m=428 -> N=3100 vars, C=13268 clauses, 19468 nodes), run inside the device-side
sampler loop (randomized rounding -> step -> SAT check -> reverse step).  Inputs are
resident in HBM before the timed region; the step runs with the reference's semantics
(early exit armed, per-round logits / map selection / is_batch_sat every round), which
with random-init weights never fires, so the work per step is fixed.

For N>1 the driver launches one rank per GPU (torch.distributed.run); every rank owns
its own batch (weak scaling), weights are broadcast from rank 0 over RCCL, and there
is no communication in the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA = vector fp32 peak
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak


def gemm_flops_per_round(n_vars, n_clauses):
    """Algorithmic flops of the 12 Dense layers of one round."""
    return 1_446_966 * n_vars + 261_120 * n_clauses


def sparse_bytes_per_round(n_vars, n_clauses, n_edges, f=128):
    """Unique-traffic bytes of the four sparse products of one round."""
    return 4 * f * (7 * n_vars + 4 * n_clauses) + 16 * n_edges + 8 * (n_clauses + 1) + 8 * (2 * n_vars + 1)


def cpu_baseline(instances, ws, rounds, budget_s=20.0):
    """CPU restatement (oracle/querysat_torch.py), timed on this host's cores on a bounded
    sample of the same workload: whole denoising steps of the same batch."""
    import torch
    from oracle import querysat_torch as QT
    tb = QT.TorchBatch(instances)
    tw = QT.to_torch_weights(ws)
    rng = np.random.default_rng(0)
    n = tb.n
    noisy = np.stack([np.ones(n), np.zeros(n)], 1).astype(np.float32)
    normal = rng.standard_normal((rounds, n, 4)).astype(np.float32)
    labels = rng.integers(0, 2, n)
    with torch.no_grad():
        QT.diffusion_step(tw, tb, 0.5, noisy, normal[:2], labels, 2)      # warm-up: 2 rounds
        # be fair to the CPU: pick the thread count that runs this problem fastest (sparse products
        # of a 20k-node batch stop scaling long before all cores of a large host are used)
        ncpu = int(torch.get_num_threads())
        best_t, best_dt = ncpu, None
        for t in sorted({1, 4, 8, 16, 32, 64, ncpu}):
            if t > ncpu:
                continue
            torch.set_num_threads(t)
            c0 = time.perf_counter()
            QT.diffusion_step(tw, tb, 0.5, noisy, normal[:2], labels, 2)
            dt = time.perf_counter() - c0
            if best_dt is None or dt < best_dt:
                best_t, best_dt = t, dt
        torch.set_num_threads(best_t)
        done, t0 = 0, time.perf_counter()
        while True:
            QT.diffusion_step(tw, tb, 0.5, noisy, normal, labels, rounds)
            done += 1
            el = time.perf_counter() - t0
            if el > budget_s or done >= 48:
                break
    return {"value": done / el, "unit": "denoising-steps/s", "cores": int(torch.get_num_threads()),
            "kind": "port",
            "sample": "%d denoising steps (R=%d rounds) of the same config-2 batch, PyTorch-CPU fp32 restatement, "
                      "early exit off, %.1f s; thread count chosen as the fastest of {1,4,8,16,32,64,all=%d} on a "
                      "2-round probe" % (done, rounds, el, ncpu)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=32)
    ap.add_argument("--graphs", type=int, default=31, help="instances per batch (config 2: 31 x n=100)")
    ap.add_argument("--vars", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--eager", action="store_true", help="disable hipGraph replay")
    ap.add_argument("--batches-in-flight", type=int, default=1,
                    help="opt-in throughput mode: this many independent config-2 batches per GPU, each on its own "
                         "HIP stream (the K steps are dealt round-robin over them); the default, 1, is the contract's "
                         "one batch per GPU")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    # Rehearsal on a one-GPU box: DSAT_BENCH_BACKEND=gloo lets several ranks share cuda:0 (RCCL
    # refuses two ranks on one device); the driver's multi-GPU runs use the default, nccl = RCCL.
    backend = os.environ.get("DSAT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
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

    from diffusionsat_amd import build, runtime, weights as W
    from diffusionsat_amd.batch import make_batch
    from diffusionsat_amd.instances import config2_instances
    build.build()

    # ---- inputs: one config-2 batch per rank (weak scaling), weights from rank 0 -------------
    instances = config2_instances(args.graphs, args.vars, seed=2 + rank)
    batch = make_batch(instances)
    ws = W.glorot_weights(seed=0)
    blob = torch.from_numpy(W.to_blob(ws)).to(dev)
    if dist is not None:
        if rank != 0:
            blob.zero_()
        dist.broadcast(blob, src=0)            # RCCL over xGMI; the only collective on the path
    graph = runtime.Graph(batch)
    wts = runtime.Weights(blob)
    sess = runtime.Session(graph, wts, max_rounds=args.rounds, device=dev)
    sess.set_graph_replay(not args.eager)
    nflight = max(1, args.batches_in_flight)
    extra = []   # (session, stream) of the other batches in flight; each has its own instances
    for b in range(1, nflight):
        bb = make_batch(config2_instances(args.graphs, args.vars, seed=1000 * b + 2 + rank))
        sb = runtime.Session(runtime.Graph(bb), wts, max_rounds=args.rounds, device=dev)
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

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(nflight, seed=0)       # untimed: first-use work (hipGraph capture, allocations) whatever W is
    if args.warmup > 0:
        run(args.warmup, seed=1)
    barrier()
    t0 = time.perf_counter()
    out = run(args.steps, seed=2)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    solved_fraction = float(out["solved"].float().mean().item())

    result = None
    if rank == 0:
        n, c, e = batch.n_vars, batch.n_clauses, batch.n_edges
        value = world * args.steps / elapsed
        result = {
            "metric": "denoising-steps/sec, 20k-node 3-SAT batch",
            "value": value, "unit": "denoising-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "solved_fraction": solved_fraction,
            "config": {"workload": "BASELINE config 2: %d satisfiable threshold 3-SAT instances n=%d m=%d per GPU "
                                   "(N=%d vars, C=%d clauses, E=%d edges, %d nodes), R=%d rounds per denoising "
                                   "step, T=%d steps, random-init Glorot weights, device Philox noise, early exit "
                                   "armed (never fires)" % (batch.n_graphs, args.vars, len(instances[0][1]), n, c, e,
                                                            batch.n_nodes, args.rounds, args.steps),
                       "rounds": args.rounds, "nodes_per_gpu": batch.n_nodes, "parallelism": "batch-sharded x%d" % world,
                       "hip_graph_replay": not args.eager, "batches_in_flight_per_gpu": nflight},
        }
        # ---- per-kernel device times, live (HIP events on the launch stream, eager) -------------
        if not args.no_profile:
            sess.set_profile(True)
            run(2, seed=3)
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
            # HBM bytes per launch from the PMC passes kept under profiles/ (rocprofv3 cannot run inside
            # this process); null when the file is absent
            traffic = {}
            try:
                with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                    traffic = {k: v["hbm_bytes_per_launch"] for k, v in json.load(f)["kernels"].items()}
            except (OSError, KeyError, ValueError):
                pass
            roofs = {}
            for k, fl in per_launch.items():
                if k in prof and prof[k]["launches"] > 0:
                    us = prof[k]["avg_us"]
                    ach = fl / (us * 1e-6) / 1e12
                    roofs[k] = {"bound": "mfma", "kernel": kern_name[k], "achieved": ach,
                                "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_F32_TFLOPS,
                                "traffic": traffic.get(k) if (args.graphs, args.vars) == (31, 100) else None,
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
                t2 = st[1, :(c + 63) // 64].astype(np.float64)
                t3 = st[2, :(n + 15) // 16].astype(np.float64)
                g2, tot2 = np.median(t2[:, 2] - t2[:, 1]), np.median(t2[:, 4] - t2[:, 0])
                g3, tot3 = np.median(t3[:, 1] - t3[:, 0]), np.median(t3[:, 3] - t3[:, 0])
                rk = result["roofline_kernels"]
                us2 = g2 / tot2 * rk["round_p2_clause"]["avg_launch_us"]
                us3 = g3 / tot3 * rk["round_p3_var_update"]["avg_launch_us"]
                nbytes = sparse_bytes_per_round(n, c, e)
                ach = nbytes / ((us2 + us3) * 1e-6) / 1e9
                result["roofline_sparse"] = {
                    "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                    "algorithmic_bytes_per_round": nbytes, "gather_us_per_round": {"p2_literal_to_clause": us2,
                                                                                   "p3_clause_to_literal": us3},
                    "note": "the four sparse products of a round (R3, R4, R7, R11) are the gather phases of the fused "
                            "P2 and P3 kernels; duration = that phase's share of the tile's stamped cycles x the "
                            "kernel's event duration; the rows are served by L2 / Infinity Cache (state written by the "
                            "previous kernel), so this is the sparse phases' rate against the HBM roofline, not HBM "
                            "traffic (see roofline.traffic for that)"}
                del s3
            except Exception as ex:      # diagnostics only: never fail the bench line
                result["roofline_sparse"] = {"error": str(ex)}
        if world == 1 and nflight == 1 and not args.no_profile and args.steps % 2 == 0:
            # secondary figure, same run: two independent config-2 batches in flight on two HIP streams
            # (what samples(n) can do when n needs more than one batch); never the headline `value`
            b2 = make_batch(config2_instances(args.graphs, args.vars, seed=1002 + rank))
            s2 = runtime.Session(runtime.Graph(b2), wts, max_rounds=args.rounds, device=dev)
            s2.set_graph_replay(not args.eager)
            extra.append((s2, torch.cuda.Stream(device=dev)))
            nflight = 2
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
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(instances, ws, args.rounds)
            result["cpu_baseline"]["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
