"""Phase breakdown of the fused round kernels from in-kernel clock stamps (DSAT_STAMPS=1).

On the GPU box (writes the raw stamps + the table; both are committed under profiles/):
    python tools/stamps.py --config 2 --out profiles/r02_stamps_cfg2
Anywhere (re-derives the table from a committed dump, no GPU needed):
    python tools/stamps.py --from profiles/r02_stamps_cfg2.npz

The stamps are `s_memtime` (shader clock) values taken by thread 0 of every workgroup at the phase
boundaries of the LAST full launch of each kernel in a denoising step; slot 7 holds the 100 MHz
real-time ticks between the tile's first and last stamp, which gives the clock the kernel ran at.
Only a diagnostic session executes them (a session created with DSAT_STAMPS=1).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PHASES = {
    "round_p1": [("pairnorm+aux", 0, 1), ("out_mlp || query_mlp", 1, 3), ("lit_mlp", 3, 4)],
    "round_p2": [("state rows / pairnorm", 0, 1), ("gather (R3,R5,R7)", 1, 2), ("clause_mlp", 2, 3),
                 ("column sums", 3, 4)],
    "round_p3": [("gather (R4,R11)", 0, 1), ("  meta + list staging + own rows", 0, 5), ("  neighbour loop", 5, 6),
                 ("  closed form + tile write", 6, 1), ("update_mlp", 1, 2), ("column sums", 2, 3)],
}
LAST = {"round_p1": 4, "round_p2": 4, "round_p3": 3}
# MFMA issue floor of one tile: 16x16x4 MFMAs per SIMD x 32 cycles (fused_common.h blocking)
MFMA_FLOOR = {"round_p1": (4 * (9 * 32 + 32 * 32 + 32 * 16) + 4 * (8 * 8 + 8 * 1 + 9 * 10 + 10 * 8)) // 4 * 32,
              "round_p2": 16 * (24 * 13 + 13 * 16) // 4 * 32,
              "round_p3": 4 * (33 * 15 + 15 * 15 + 15 * 8) // 4 * 32}


def collect(config, rounds):
    os.environ["DSAT_STAMPS"] = "1"
    import ctypes as C
    import torch
    from diffusionsat_amd import runtime, _lib, weights as W
    from diffusionsat_amd.batch import make_batch
    from diffusionsat_amd.instances import config2_instances, config3_instances, irregular_cnf
    if config == 2:
        insts = config2_instances(31)
    elif config == 3:
        insts = config3_instances(256)
    else:
        insts = [irregular_cnf(10000, 60000, seed=5)[:2]]
    batch = make_batch(insts)
    s = runtime.Session(runtime.Graph(batch), runtime.Weights(W.glorot_weights(0)), max_rounds=rounds)
    s.set_graph_replay(False)
    for _ in range(3):     # bring the clocks to their loaded state
        s.sample(4, rounds, seed=1, early_exit=True, want_solved_fraction=False)
    torch.cuda.synchronize()
    buf = (C.c_int64 * (3 * 4096 * 8))()
    _lib.check(_lib.load().dsat_session_read_stamps(s.handle, buf, 3 * 4096 * 8))
    st = np.frombuffer(buf, dtype=np.int64).reshape(3, 4096, 8).copy()
    tiles = [min(4096, (batch.n_vars + 15) // 16), min(4096, (batch.n_clauses + 63) // 64),
             min(4096, (batch.n_vars + 15) // 16)]
    out = {"round_p1": st[0, :tiles[0]], "round_p2": st[1, :tiles[1]], "round_p3": st[2, :tiles[2]]}
    # the other workgroups of the same grids (role B / clause-norm blocks of P1, select/SAT blocks of P2):
    # whatever follows the tiles and carries stamps
    for name, k in (("round_p1_extra", 0), ("round_p2_extra", 1), ("round_p3_extra", 2)):
        rest = st[k, tiles[k]:]
        out[name] = rest[rest[:, 7] > 0]
    return out, {"config": config, "rounds": rounds, "n_vars": batch.n_vars, "n_clauses": batch.n_clauses,
                 "n_graphs": batch.n_graphs, "split_mask": s.split_mask()}


def table(stamps, meta):
    out = {"meta": meta, "kernels": {}}
    lines = []
    for k in ("round_p1", "round_p2", "round_p3"):
        t = stamps[k].astype(np.float64)
        tot = t[:, LAST[k]] - t[:, 0]
        rt = t[:, 7]                                   # 100 MHz ticks over the same span
        ok = rt > 0
        ghz = float(np.median(tot[ok] / (rt[ok] * 10.0))) if ok.any() else float("nan")
        floor = MFMA_FLOOR[k]
        mask = int(meta.get("split_mask", 0))
        # split-product tiles (csrc/split_bf16.h): six 16-cycle bf16 MFMAs per 32 k-values, busiest SIMD
        if k == "round_p1" and mask & 32:   # role_a1s: lit_mlp alone, two column tiles per wave in layers 0 and 1
            floor = (4 * 2 * 5 + 4 * 2 * 16 + 4 * 16) * 6 * 16
        if k == "round_p2" and mask & 1:
            floor = (13 * 12 + 16 * 7) * 6 * 16
        if k == "round_p3" and mask & 2:   # (role D of P2 takes 4 of layer 0's 17 k-steps: bit 6)
            floor = (4 * (13 if mask & 64 else 17) + 4 * 8 + 2 * 8) * 6 * 16
        kk = {"tiles": int(t.shape[0]), "cycles_per_tile_median": float(np.median(tot)),
              "cycles_per_tile_max": float(tot.max()), "clock_ghz_median": ghz,
              "us_per_tile_median": float(np.median(tot)) / (ghz * 1e3) if ghz == ghz else None,
              "mfma_issue_floor_cycles": floor, "mfma_issue_share": floor / float(np.median(tot)), "phases": {}}
        lines.append("%s: %d tiles; median %.0f cycles per tile (max %.0f) at %.2f GHz = %.1f us; MFMA issue floor "
                     "%d cycles = %.0f %%" % (k, t.shape[0], np.median(tot), tot.max(), ghz,
                                              np.median(tot) / (ghz * 1e3), floor, 100 * floor / np.median(tot)))
        phases = list(PHASES[k])
        if k == "round_p2" and (t[:, 5] > 0).all():    # the split-product tile stamps its layers
            phases[3:3] = [("  layer 0 (three segments)", 2, 5), ("  layer 1 (epilogue 0, barrier, k-loop)", 5, 6),
                           ("  outputs", 6, 3)]
        for name, lo, hi in phases:
            d = t[:, hi] - t[:, lo]
            kk["phases"][name.strip()] = {"median": float(np.median(d)), "max": float(d.max())}
            lines.append("    %-34s median %8.0f  max %8.0f cycles  (%4.1f %%)" % (
                name, np.median(d), d.max(), 100 * np.median(d) / np.median(tot)))
        out["kernels"][k] = kk
    for k, what in (("round_p1_extra", "P1 role B (64-row pair: pairnorm | out chain | query chain) and role C "
                                       "(clause-norm) workgroups"),
                    ("round_p2_extra", "P2 select/SAT workgroups"),
                    ("round_p3_extra", "P3 role C workgroups (clause PairNorm of this round, on the CUs the tiles leave idle)")):
        if k in stamps and len(stamps[k]):
            t = stamps[k].astype(np.float64)
            tot = t[:, 4] - t[:, 0]
            isb = t[:, 2] > 0            # role B stamps its chain boundaries, role C / select do not
            lines.append("%s: %d workgroups" % (what, len(tot)))
            for nm, m in (("with chain stamps", isb), ("without", ~isb)):
                if m.any():
                    lines.append("    %-18s n=%3d  median %8.0f  max %8.0f cycles" % (nm, m.sum(), np.median(tot[m]),
                                                                                      tot[m].max()))
                    out["kernels"][k + ("_b" if nm.startswith("with ") else "_c")] = {
                        "n": int(m.sum()), "cycles_median": float(np.median(tot[m])), "cycles_max": float(tot[m].max())}
            if isb.any():
                for nm, lo, hi in (("pairnorm(64 rows)", 0, 1), ("out chain", 1, 2), ("query chain", 2, 3)):
                    d = t[isb][:, hi] - t[isb][:, lo]
                    lines.append("      %-18s median %8.0f  max %8.0f" % (nm, np.median(d), d.max()))
    return out, "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 5])
    ap.add_argument("--rounds", type=int, default=32)
    ap.add_argument("--out", default=None, help="prefix: writes <prefix>.npz (raw stamps) and <prefix>.json (table)")
    ap.add_argument("--from", dest="src", default=None, help="re-derive the table from a committed .npz dump")
    a = ap.parse_args()
    if a.src:
        z = np.load(a.src, allow_pickle=False)
        stamps = {k: z[k] for k in z.files if k != "meta"}
        meta = json.loads(str(z["meta"]))
    else:
        stamps, meta = collect(a.config, a.rounds)
    res, text = table(stamps, meta)
    print(text)
    if a.out and not a.src:
        np.savez_compressed(a.out + ".npz", meta=json.dumps(meta), **stamps)
        with open(a.out + ".json", "w") as f:
            json.dump(res, f, indent=1)
        with open(a.out + ".txt", "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
