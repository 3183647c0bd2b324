"""Per-kernel medians of rocprofv3 --pmc passes -> one JSON (profiles/rNN_pmc_counters.json).

    python tools/pmc_report.py gpurun_out/r02prof/pmc_* > profiles/r02_pmc_counters.json

Every pass is `rocprofv3 --pmc <counters> -d <dir> -o runc -- python bench.py --steps 4 --warmup 1 --repeats 1
--no-cpu-baseline --no-profile` (config 2).  Values are medians over the launches of a kernel (the median
ignores P1's two odd launches per step: round 0 without the clause-norm workgroups, and the tail).
Units as MI355X_MICROARCH.md states them: SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over the chip, SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD-pipe summed over the
chip; FETCH_SIZE / WRITE_SIZE are KiB and FETCH_SIZE tallies 128-B requests at 64 B (doubled here);
TCC_* are request counts summed over the channels.
"""
import csv
import glob
import json
import os
import statistics
import sys

KERNELS = {"round_p1_kernel": "round_p1", "round_p2_kernel": "round_p2", "round_p3_kernel": "round_p3",
           "round_p2s_kernel": "round_p2", "round_p3s_kernel": "round_p3",   # the split-product forms of the same tiles
           "heavy_gather_kernel": "heavy_row_gather", "segsum2_kernel": "pairnorm_segsum2",
           "select_sat_kernel": "select_sat", "aux_kernel": "aux"}


def rows_of(d):
    """Counter rows of one pass directory: rocprofv3's CSV output, or its default rocpd SQLite database."""
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                yield row
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        import sqlite3
        db = sqlite3.connect(f)
        q = ("select kernel_name, counter_name, value, vgpr_count, accum_vgpr_count, sgpr_count, lds_block_size, "
             "workgroup_size from counters_collection")
        for r in db.execute(q):
            yield {"Kernel_Name": r[0], "Counter_Name": r[1], "Counter_Value": r[2], "VGPR_Count": r[3],
                   "Accum_VGPR_Count": r[4], "SGPR_Count": r[5], "LDS_Block_Size": r[6], "Workgroup_Size": r[7]}
        db.close()


def kernel_stats(db_path):
    """`rocprofv3 --kernel-trace --stats` summary of a rocpd database as CSV text (Name, Calls, ...)."""
    import sqlite3
    db = sqlite3.connect(db_path)
    out = ['"Name","Calls","TotalDurationUs","AverageUs","Percentage"']
    for r in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        out.append('"%s",%d,%.3f,%.3f,%.4f' % r)
    return "\n".join(out) + "\n"


def main(dirs):
    if dirs and dirs[0] == "--stats":
        sys.stdout.write(kernel_stats(dirs[1]))
        return
    vals = {}
    for d in dirs:
        for row in rows_of(d):
            for k, nm in KERNELS.items():
                if k in row["Kernel_Name"]:
                    vals.setdefault(nm, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                    meta = vals[nm].setdefault("_meta", {})
                    meta["vgpr"] = int(row["VGPR_Count"]); meta["agpr"] = int(row["Accum_VGPR_Count"])
                    meta["sgpr"] = int(row["SGPR_Count"]); meta["lds_bytes"] = int(row["LDS_Block_Size"])
                    meta["workgroup"] = int(row["Workgroup_Size"])
    out = {"source": __doc__.strip().split("\n\n")[1].replace("\n", " "), "kernels": {}}
    for nm, cs in vals.items():
        k = {"registers": cs.pop("_meta", {})}
        for c, v in sorted(cs.items()):
            k[c] = {"median": statistics.median(v), "min": min(v), "max": max(v), "launches": len(v)}
        g = lambda c: k[c]["median"] if c in k else None
        d = {}
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
            d["l2_hit_rate"] = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            d["hbm_bytes_per_launch"] = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024
        if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") is not None:
            d["wave_time_waiting"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY") is not None:
            d["wave_time_issue_stalled"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
        if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_ANY") is not None:
            d["wave_time_issuing"] = g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")
        if g("SQ_LDS_IDX_ACTIVE") and g("SQ_LDS_BANK_CONFLICT") is not None:
            d["lds_bank_conflict_share"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
        if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            d["mfma_busy_cycles_over_sq_busy_quadcycles"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")
        k["derived"] = d
        out["kernels"][nm] = k
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
