"""Shared builders for the parity tests: same seeded inputs for the oracle and the HIP path."""
import numpy as np

from oracle import querysat_oracle as O
from diffusionsat_amd.batch import make_batch
from diffusionsat_amd.instances import planted_ksat


def small_instances(kind="3sat", seed=0):
    if kind == "3sat":          # three different small 3-SAT formulas
        return [planted_ksat(12 + 3 * i, 40 + 7 * i, 3, seed=seed + i)[:2] for i in range(3)]
    if kind == "ragged":        # mixed clause widths, a unit clause, duplicate literals, an unused variable
        return [
            (6, [[1, -2, 3], [2], [-3, 4, 5, -6, 1], [1, 1, -4], [-5, 6]]),
            (4, [[1, 2], [-1, -2], [3, -3]]),            # variable 4 never occurs
            (9, planted_ksat(9, 30, 4, seed=seed + 5)[1]),
        ]
    if kind == "uf20":
        return [planted_ksat(20, 91, 3, seed=seed + 1)[:2]]
    if kind == "medium":        # 12 graphs of n=50: crosses several GEMM row tiles
        return [planted_ksat(50, 214, 3, seed=seed + 10 + i)[:2] for i in range(12)]
    raise ValueError(kind)


def oracle_weights(seed=0, bias_scale=0.05):
    return O.make_weights(seed, np.float32, bias_scale=bias_scale)


def draws(n_vars, rounds, steps=None, seed=0):
    rng = np.random.default_rng(seed)
    if steps is None:
        return {
            "normal": rng.standard_normal((rounds, n_vars, 4)).astype(np.float32),
            "labels": rng.integers(0, 2, n_vars).astype(np.int32),
            "uniform": rng.random((n_vars, 1)).astype(np.float32),
        }
    return {
        "normal": rng.standard_normal((steps, rounds, n_vars, 4)).astype(np.float32),
        "labels": rng.integers(0, 2, (steps, n_vars)).astype(np.int32),
        "uniform": rng.random((steps, n_vars, 1)).astype(np.float32),
    }


def make_all(kind="3sat", seed=0):
    insts = small_instances(kind, seed)
    batch = make_batch(insts)
    data = O.create_adj_matrices(insts)
    return insts, batch, data
