"""HBM bytes per launch of the fused round kernels from two rocprofv3 counter passes.

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o runc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o runc -- python bench.py ...        (separate pass: TCC slots)
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_pmc_traffic.json

Units and the gfx950 correction as MI355X_MICROARCH.md prescribes: both counters are in KiB;
FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B, so it is doubled."""
import csv, glob, json, os, statistics, sys

NAMES = {"round_p1_kernel": "round_p1_var_query_lit", "round_p2_kernel": "round_p2_clause",
         "round_p3_kernel": "round_p3_var_update"}


def collect(d, counter):
    vals = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                for k, nm in NAMES.items():
                    if k in row["Kernel_Name"]:
                        # the tail launch of P1 (output MLP only) would drag the median of P1 down
                        vals.setdefault(nm, []).append(float(row["Counter_Value"]))
    return vals


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 4 --warmup 1` "
                 "(config 2), median per launch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced "
                 "reads on gfx950 (counter unit KiB)", "kernels": {}}
for nm in NAMES.values():
    if nm in fetch and nm in write:
        f, w = statistics.median(fetch[nm]), statistics.median(write[nm])
        out["kernels"][nm] = {"fetch_size_raw_kib": f, "write_size_raw_kib": w, "launches_seen": len(fetch[nm]),
                              "hbm_bytes_per_launch": (2 * f + w) * 1024}
print(json.dumps(out, indent=1))
