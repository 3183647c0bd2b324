"""Generate tests/golden/step_golden.npz: inputs and oracle outputs of one denoising step.

The reference's model path cannot run here (TensorFlow absent) and holds no golden vectors
(SURVEY.md §4), so this fixture is produced by the CPU oracle (oracle/querysat_oracle.py, fp64
and fp32) from seeded weights and injected noise.  It pins the oracle against accidental
change and gives the GPU tests a committed target; it does NOT pin the oracle against the
reference ("parity unpinned", oracle/__init__.py).

    python tests/golden/make_step_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import draws, make_all, oracle_weights  # noqa: E402
from oracle import querysat_oracle as O  # noqa: E402


def main():
    out = {}
    for kind, ns in (("3sat", 0.5), ("ragged", 1.0 / 32)):
        insts, batch, data = make_all(kind)
        ws = oracle_weights(5)
        rounds = 8
        d = draws(batch.n_vars, rounds, seed=11)
        noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, np.float32), d["uniform"])
        r32 = O.diffusion_step(ws, data, np.float32(ns), noisy, d["normal"], d["labels"], rounds,
                               early_exit=False, dt=np.float32, trace=True)
        w64 = O.cast_weights(ws, np.float64)
        r64 = O.diffusion_step(w64, data, np.float64(ns), noisy.astype(np.float64), d["normal"], d["labels"],
                               rounds, early_exit=False, dt=np.float64, trace=True)
        out[kind + "_noise_scale"] = np.float32(ns)
        out[kind + "_noisy"] = noisy
        out[kind + "_normal"] = d["normal"]
        out[kind + "_labels"] = d["labels"]
        out[kind + "_round_logits_f32"] = np.stack(r32["trace"]["logits"]).astype(np.float32)
        out[kind + "_round_logits_f64"] = np.stack(r64["trace"]["logits"]).astype(np.float64)
        out[kind + "_best_map"] = r32["best_map"].astype(np.int32)
        out[kind + "_prediction_f32"] = r32["prediction"].astype(np.float32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
