"""bench.py --config 4: the NeuroSAT-style SR(n) stream (BASELINE.json configs[3]; data/k_sat.py).

Every rank takes its share of a 100 000-instance stream (`--stream-instances` of them; default = the full
config's 12 500 per GPU), batches it to the reference's 20 000-node budget
(`batch.batch_by_node_budget`, data/dimac.py:267-293) and samples every batch for T=32 denoising
steps through `diffusionsat_amd.stream.StreamSampler`: two pooled slots, the host set-up of batch
i+1 (CSR build, upload, session, hipGraph capture) overlapped with batch i on the device.

The timed region is the whole stream -- batches are handed over as host arrays, so graph upload
(about 1 MB per batch) is inside it; instance generation (a SAT solver per clause) and batching are
input preparation and stay outside.  One "step" is one denoising step of one batch.
"""
import time

import numpy as np

PEAK_MFMA_F32_TFLOPS = 157.3


def run_stream_bench(args, rank, world, dev, wts, ws, barrier, gather_floats):
    import torch
    from diffusionsat_amd import runtime
    from diffusionsat_amd.batch import batch_by_node_budget, make_batch
    from diffusionsat_amd.instances import sr_ksat_native
    from diffusionsat_amd.stream import StreamSampler
    import os
    from concurrent.futures import ThreadPoolExecutor

    T = args.steps                      # denoising steps per batch (the config's T = 32)
    n_inst = int(args.stream_instances)
    # ---- input preparation (untimed): this rank's instances of the global stream i = rank, rank+world, ...
    t_gen = time.perf_counter()
    base = 4 << 32
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        insts = list(ex.map(lambda j: sr_ksat_native(base + rank + j * world), range(n_inst)))
    gen_s = time.perf_counter() - t_gen
    t_b = time.perf_counter()
    groups = batch_by_node_budget(insts, 20000)
    batches = [make_batch(g) for g in groups]
    batch_s = time.perf_counter() - t_b
    nodes = [b.n_nodes for b in batches]

    def stream_once(pooled, slots, seed, collective=True):
        sync = barrier if collective else torch.cuda.synchronize      # rank 0's extra passes run alone
        smp = StreamSampler(wts, device=dev, slots=slots, n_steps=T, rounds=args.rounds, seed=seed, pooled=pooled,
                            graph_replay=not args.eager)
        try:
            smp.run(batches[:min(len(batches), 2 * slots)])          # untimed: arena sizing, first-use work
            sync()
            t0 = time.perf_counter()
            st = smp.run(batches)
            sync()
            st["elapsed"] = time.perf_counter() - t0
        finally:
            smp.close()
        return st

    runs = []
    for rep in range(max(1, args.repeats)):
        runs.append(stream_once(True, 2, seed=10 + rep))
    el_all = [max(gather_floats(r["elapsed"])) for r in runs]
    elapsed = float(np.median(el_all))
    steps_rank = len(batches) * T
    steps_all = sum(gather_floats(steps_rank))
    per_rank = gather_floats(steps_rank / float(np.median([r["elapsed"] for r in runs])))
    if rank != 0:
        return None
    mid = runs[int(np.argsort(el_all)[len(el_all) // 2])]
    value = steps_all / elapsed
    ms = [1e3 * e / steps_rank for e in el_all]
    result = {
        "metric": "denoising-steps/sec, SR(30-100) k-SAT stream in 20k-node batches",
        "value": value, "unit": "denoising-steps/s", "n_gpus": world, "steps": steps_rank, "warmup": 2 * 2 * T,
        "ms_per_step": 1e3 * elapsed / steps_rank, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "solved_fraction": None,      # unmeasured without trained weights (bench.py has the note)
        "solved_fraction_random_init": float(np.sum(mid["solved"]) / max(1, mid["instances"])),
        "repeats": {"n": len(ms), "ms_per_step_median": float(np.median(ms)), "ms_per_step_min": min(ms),
                    "ms_per_step_max": max(ms), "ms_per_step_all": [round(x, 5) for x in ms]},
        "per_rank_steps_per_s": [round(x, 3) for x in per_rank],
        "config": {
            "workload": "BASELINE config 4: NeuroSAT-style SR(n) k-SAT stream (n uniform in [30,100], clause width "
                        "(1 w.p. 0.3 else 2)+Geometric(0.4), solver-terminated, solutions multiplier 10; "
                        "data/k_sat.py), %d instances per GPU (the full config: 12 500) in %d node-budget batches of "
                        "<= 20 000 nodes (mean %.0f nodes, %.1f instances per batch), T=%d denoising steps x R=%d "
                        "rounds per batch, two pooled slots per GPU (arena-backed graph + session, host set-up of "
                        "the next batch overlapped with the running one), random-init Glorot weights, device Philox "
                        "noise, early exit armed; the timed region is the whole stream, graph upload included"
                        % (n_inst, len(batches), float(np.mean(nodes)), n_inst / max(1, len(batches)), T, args.rounds),
            "baseline_config": 4, "rounds": args.rounds, "batches_per_gpu": len(batches),
            "instances_per_gpu": n_inst, "parallelism": "stream-sharded x%d" % world,
            "hip_graph_replay": not args.eager, "batches_in_flight_per_gpu": 2},
        "stream": {
            "instances_per_s": world * n_inst / elapsed,
            "host_setup_ms_per_batch": {"median": float(np.median(mid["host_setup_ms"])),
                                        "max": float(np.max(mid["host_setup_ms"]))},
            "host_wait_ms_per_batch": {"median": float(np.median(mid["host_wait_ms"]))},
            "device_ms_per_batch": {"median": float(np.median(mid["device_ms"])),
                                    "note": "HIP-event span of one batch's T steps on its own stream; two batches "
                                            "overlap on the device, so the spans add up to more than the wall time"},
            "arena_growths": int(sum(r["arena_growths"] for r in runs)),
            "input_preparation_s": {"generate": gen_s, "batch": batch_s},
        },
    }
    if not args.no_profile:
        # A/B: the same stream without pooling (dsat_graph_create / dsat_session_create + hipFree per batch),
        # and strictly one batch at a time (host set-up exposed)
        # (on a bounded sample of the stream, its first 48 batches: the default run must stay within minutes)
        ab = {"batches": min(len(batches), 48)}
        full = batches
        batches = batches[:48]
        for name, pooled, slots in (("pooled_two_slots", True, 2), ("unpooled_two_slots", False, 2),
                                    ("pooled_one_slot", True, 1), ("unpooled_one_slot", False, 1)):
            r = stream_once(pooled, slots, seed=99, collective=False)
            ab[name] = {"value": len(batches) * T / r["elapsed"],
                        "host_setup_ms_median": float(np.median(r["host_setup_ms"]))}
        batches = full
        result["stream"]["ab"] = ab
        # roofline of the dominant kernel on the stream's median-size batch (eager profiling pass)
        b = batches[int(np.argsort(nodes)[len(nodes) // 2])]
        sess = runtime.Session(runtime.Graph(b, device=dev), wts, max_rounds=args.rounds, device=dev)
        sess.set_throughput(True)               # as the stream's own sessions (two slots in flight)
        sess.sample(2, args.rounds, seed=1, early_exit=True, want_solved_fraction=False)
        sess.set_profile(True)
        sess.sample(2, args.rounds, seed=2, early_exit=True, want_solved_fraction=False)
        prof = sess.profile()
        sess.set_profile(False)
        n, c, r_ = b.n_vars, b.n_clauses, args.rounds
        per_launch = {"round_p1_var_query_lit": 2.0 * n * (r_ * (40_545 + 463_360) + r_ * 17_408) / (r_ + 1),
                      "round_p2_clause": 2.0 * c * 130_560, "round_p3_var_update": 2.0 * n * 202_170}
        kern = {"round_p1_var_query_lit": "round_p1_kernel", "round_p2_clause": "round_p2_kernel<4>",
                "round_p3_var_update": "round_p3_kernel"}
        all_us = sum(v["avg_us"] * v["launches"] for v in prof.values())
        roofs = {}
        mask = sess.split_mask()   # kernels whose fp32 products are six bf16-slice products: priced on the bf16 pipe
        split = {"round_p2_clause": bool(mask & 1), "round_p3_var_update": bool(mask & 2)}
        for k, fl in per_launch.items():
            if prof.get(k, {}).get("launches", 0) > 0:
                us = prof[k]["avg_us"]
                ach = fl / (us * 1e-6) / 1e12
                sp = split.get(k, False)
                peak = 16 * PEAK_MFMA_F32_TFLOPS / 6 if sp else PEAK_MFMA_F32_TFLOPS
                if k == "round_p1_var_query_lit" and mask & 4:   # lit_mlp on the bf16 pipe, the pair on the fp32 MFMA
                    peak = (463_360 + 40_545 + 17_408) / (463_360 / (16 * PEAK_MFMA_F32_TFLOPS / 6) + (40_545 + 17_408) / PEAK_MFMA_F32_TFLOPS)
                roofs[k] = {"bound": "mfma", "kernel": {"round_p2_clause": "round_p2s_kernel", "round_p3_var_update": "round_p3s_kernel"}[k] if sp else kern[k],
                            "achieved": ach, "peak": peak,
                            "pipe": "v_mfma_f32_16x16x32_bf16 x6 per fp32 product (peak = bf16 dense / 6)" if sp else "v_mfma_f32_16x16x4_f32",
                            "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                            "avg_launch_us": us, "algorithmic_flops_per_launch": fl,
                            "share_of_device_time": us * prof[k]["launches"] / all_us,
                            "batch": {"n_vars": n, "n_clauses": c, "n_graphs": b.n_graphs}}
        if roofs:
            result["roofline"] = roofs[max(roofs, key=lambda k: roofs[k]["share_of_device_time"])]
            result["roofline_kernels"] = roofs
    if not args.no_cpu_baseline:
        from bench import cpu_baseline
        b = batches[int(np.argsort(nodes)[len(nodes) // 2])]
        cb = cpu_baseline([i.as_instance() for i in b.instances], ws, args.rounds, b.n_nodes)
        cb["sample"] = "median-size batch of the stream (%d nodes): " % b.n_nodes + cb["sample"]
        cb["gpu_over_cpu"] = value / cb["value"]
        result["cpu_baseline"] = cb
    return result
