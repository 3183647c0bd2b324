"""Synthetic CNF instance generators for the benchmark/parity configurations.

The reference's generators need cnfgen / python-sat / a prebuilt ``cgen`` binary
(data/CNFGen.py, data/k_sat.py, data/SHAGen2019.py), none of which is available;
these produce inputs of the same shape with a *planted* satisfying assignment
instead of a solver filter (documented deviation).  All are
seeded and pure NumPy.
"""
from __future__ import annotations

import numpy as np


def threshold_clause_count(n_vars: int) -> int:
    """Number of clauses at the 3-SAT threshold, data/CNFGen.py:72-73."""
    return int(4.258 * n_vars + 58.26 * np.power(n_vars, -2 / 3.0))


def planted_ksat(n_vars: int, n_clauses: int, k: int = 3, seed: int = 0):
    """Uniform k-SAT with distinct variables per clause, each clause satisfied by a
    hidden assignment.  Returns (n_vars, clauses, planted_bits)."""
    rng = np.random.default_rng(seed)
    planted = rng.integers(0, 2, size=n_vars).astype(bool)
    clauses = []
    while len(clauses) < n_clauses:
        vs = rng.choice(n_vars, size=k, replace=False)
        signs = rng.integers(0, 2, size=k).astype(bool)  # True = positive literal
        if not np.any(signs == planted[vs]):
            continue
        clauses.append([int(v + 1) if s else -int(v + 1) for v, s in zip(vs, signs)])
    return n_vars, clauses, planted.astype(np.int32)


def hard_3sat(n_vars: int = 100, seed: int = 0):
    """BASELINE config 2/3 instance: n=100, m=428 (ratio 4.28)."""
    return planted_ksat(n_vars, threshold_clause_count(n_vars), 3, seed)


def uf20_91(seed: int = 1):
    """Stand-in for SATLIB uf20-91 (BASELINE config 1): n=20, m=91."""
    return planted_ksat(20, 91, 3, seed)


def neurosat_like_ksat(seed: int = 0, min_vars: int = 30, max_vars: int = 100,
                       p_k_2: float = 0.3, p_geo: float = 0.4, ratio: float = 4.0):
    """Mixed-width clauses as in data/k_sat.py:39-47: k = (1 w.p. p_k_2 else 2) +
    Geometric(p_geo), capped at n.  Planted instead of solver-terminated."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(min_vars, max_vars + 1))
    planted = rng.integers(0, 2, size=n).astype(bool)
    m = int(ratio * n)
    clauses = []
    while len(clauses) < m:
        k_base = 1 if rng.random() < p_k_2 else 2
        k = min(n, k_base + int(rng.geometric(p_geo)))
        vs = rng.choice(n, size=k, replace=False)
        signs = rng.integers(0, 2, size=k).astype(bool)
        if not np.any(signs == planted[vs]):
            continue
        clauses.append([int(v + 1) if s else -int(v + 1) for v, s in zip(vs, signs)])
    return n, clauses, planted.astype(np.int32)


def irregular_cnf(n_vars: int = 10000, n_clauses: int = 60000, seed: int = 5,
                  max_len: int = 8, zipf_a: float = 1.3):
    """Irregular-degree stress input (stand-in for SHA-1 preimage CNFs, BASELINE
    config 5): clause lengths 2..max_len with a heavy tail, variable popularity
    power-law.  Planted."""
    rng = np.random.default_rng(seed)
    planted = rng.integers(0, 2, size=n_vars).astype(bool)
    w = 1.0 / np.power(np.arange(1, n_vars + 1), 1.0 / zipf_a)
    w = w[rng.permutation(n_vars)]
    w /= w.sum()
    lens = np.clip(2 + rng.geometric(0.45, size=n_clauses) - 1, 2, max_len)
    clauses = []
    for L in lens:
        while True:
            vs = np.unique(rng.choice(n_vars, size=int(L), replace=True, p=w))
            signs = rng.integers(0, 2, size=len(vs)).astype(bool)
            if np.any(signs == planted[vs]):
                break
        clauses.append([int(v + 1) if s else -int(v + 1) for v, s in zip(vs, signs)])
    return n_vars, clauses, planted.astype(np.int32)


def config2_instances(n_graphs: int = 31, n_vars: int = 100, seed: int = 2):
    """31 distinct hard 3-SAT instances = 19 468 nodes (BASELINE config 2)."""
    return [hard_3sat(n_vars, seed=seed * 1000 + i)[:2] for i in range(n_graphs)]


def config3_instances(n_copies: int = 256, n_vars: int = 100, seed: int = 2):
    """One config-2 instance replicated (BASELINE config 3)."""
    inst = hard_3sat(n_vars, seed=seed * 1000)[:2]
    return [inst] * n_copies
