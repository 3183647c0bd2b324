"""HBM bytes per launch of the fused round kernels from two rocprofv3 counter passes.

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o runc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o runc -- python bench.py ...        (separate pass: TCC slots)
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [bench.py workload flags, e.g. --config 3] > profiles/rNN_pmc_traffic_cfgN.json

The file records what it was measured on -- `csrc_hash` (diffusionsat_amd.build.source_hash(): the kernels) and
`workload_key` (bench.workload_key: the batch) -- and bench.py reports `roofline.traffic` from it only when both match
the run at hand (null otherwise: a traffic figure never outlives the build it was taken on).

Units and the gfx950 correction as MI355X_MICROARCH.md prescribes: both counters are in KiB;
FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B, so it is doubled."""
import csv, glob, json, os, statistics, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = {"round_p1_kernel": "round_p1_var_query_lit", "round_p2_kernel": "round_p2_clause",
         "round_p3_kernel": "round_p3_var_update", "round_p2s_kernel": "round_p2_clause",
         "round_p3s_kernel": "round_p3_var_update"}


def collect(d, counter):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from pmc_report import rows_of          # CSV or rocpd database, whichever the pass wrote
    vals = {}
    for row in rows_of(d):
        if row["Counter_Name"] != counter:
            continue
        for k, nm in NAMES.items():
            if k in row["Kernel_Name"]:
                # (median: the tail launch of P1 -- output MLP only -- would drag a mean of P1 down)
                vals.setdefault(nm, []).append(float(row["Counter_Value"]))
    return vals


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
def _keys():
    import argparse
    import bench
    from diffusionsat_amd import build
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2); ap.add_argument("--rounds", type=int, default=32)
    ap.add_argument("--graphs", type=int, default=31); ap.add_argument("--copies", type=int, default=256)
    ap.add_argument("--vars", type=int, default=100); ap.add_argument("--irr-vars", type=int, default=10000)
    ap.add_argument("--irr-clauses", type=int, default=60000); ap.add_argument("--zipf", type=float, default=1.3)
    a, _ = ap.parse_known_args(sys.argv[3:])
    return build.source_hash(), bench.workload_key(a)


csrc_hash, wkey = _keys()
out = {"csrc_hash": csrc_hash, "workload_key": wkey, "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 4 --warmup 1` "
                 "(workload_key), median per launch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced "
                 "reads on gfx950 (counter unit KiB)", "kernels": {}}
for nm in NAMES.values():
    if nm in fetch and nm in write:
        f, w = statistics.median(fetch[nm]), statistics.median(write[nm])
        out["kernels"][nm] = {"fetch_size_raw_kib": f, "write_size_raw_kib": w, "launches_seen": len(fetch[nm]),
                              "hbm_bytes_per_launch": (2 * f + w) * 1024}
print(json.dumps(out, indent=1))
