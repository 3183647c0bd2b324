"""Distance of the per-round logits to the fp64 oracle over a 32-round denoising step, for the three forms the
round runs in (split-product tiles, fp32-MFMA tiles, per-layer kernels).  GPU box:  python tests/tools/split_accuracy.py"""
import os, sys
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(_TESTS))
sys.path.insert(0, _TESTS)
import numpy as np
import torch
from oracle import querysat_oracle as O
from diffusionsat_amd import runtime as rt
from helpers import draws, make_all, oracle_weights

for kind in ("medium", "uf20"):
    insts, batch, data = make_all(kind)
    rounds = 32
    for wseed in (0, 7):
        ws32 = oracle_weights(wseed)
        ws64 = {k: [(w.astype(np.float64), b.astype(np.float64)) for w, b in v] for k, v in ws32.items()}   # the same weights, exactly
        d = draws(batch.n_vars, rounds, seed=5)
        noisy = O.randomized_rounding(np.full((batch.n_vars, 2), 0.5, np.float32), d["uniform"])
        ref64 = O.diffusion_step(ws64, data, np.float64(0.4), noisy.astype(np.float64),
                                 d["normal"].astype(np.float64), d["labels"], rounds, early_exit=False, dt=np.float64,
                                 trace=("logits",))["trace"]["logits"]
        ref32 = O.diffusion_step(ws32, data, np.float32(0.4), noisy, d["normal"], d["labels"], rounds, early_exit=False,
                                 dt=np.float32, trace=("logits",))["trace"]["logits"]
        scale = max(np.abs(ref64[r]).max() for r in range(rounds))
        line = ["%s weights %d  |logits| max %.2f; numpy fp32 oracle: %.2e" % (kind, wseed, scale, max(np.abs(ref32[r] - ref64[r]).max() for r in range(rounds)))]
        for name, env in (("split", {}), ("split+a2", {"DSAT_P1_A2": "1", "DSAT_P3_BM": "32"}), ("f32mfma", {"DSAT_P2_SPLIT": "0", "DSAT_P3_SPLIT": "0", "DSAT_P1_A2": "0", "DSAT_P1_BSPLIT": "0", "DSAT_P1_ASPLIT": "0"}), ("per_layer", {"DSAT_FUSED": "0"})):
            os.environ.update(env)
            try:
                s = rt.Session(rt.Graph(batch), rt.Weights(ws32), max_rounds=rounds)
            finally:
                for k in env: os.environ.pop(k, None)
            dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
            out = s.diffusion_step(0.4, dev(noisy), dev(d["normal"]), dev(d["labels"]), rounds=rounds, early_exit=False,
                                   want_round_logits=True)["round_logits"].cpu().numpy()
            err = [np.abs(out[r] - ref64[r]).max() for r in range(rounds)]
            line.append("%s: max %.2e (round 0 %.1e, round 31 %.1e)" % (name, max(err), err[0], err[31]))
        print("; ".join(line), flush=True)
