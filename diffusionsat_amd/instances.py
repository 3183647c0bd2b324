"""Synthetic CNF instance generators for the benchmark/parity configurations.

The reference's generators need cnfgen / python-sat / a prebuilt ``cgen`` binary
(data/CNFGen.py, data/k_sat.py, data/SHAGen2019.py), none of which is available.
Two families live here, all seeded: *planted* stand-ins (pure NumPy: same shapes, a hidden
satisfying assignment instead of a solver filter) and, at the end of the file, the
reference's own distributions for threshold 3-SAT and NeuroSAT-style k-SAT, filtered by
the extension's host-side CDCL solver (``dsat_host_solve``) in place of PySAT.  The SHA
generator's ``cgen`` binary has no source in the reference: ``irregular_cnf`` stays a
shape stand-in.
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


def config2_instances(n_graphs: int = 31, n_vars: int = 100, seed: int = 2, planted: bool = False):
    """31 distinct hard 3-SAT instances = 19 468 nodes (BASELINE config 2), drawn like the
    reference's SAT_3 dataset (solver-filtered threshold 3-SAT, ``threshold_3sat``);
    ``planted=True`` gives the planted stand-ins instead (no solver needed)."""
    gen = hard_3sat if planted else threshold_3sat
    return [gen(n_vars, seed=seed * 1000 + i)[:2] for i in range(n_graphs)]


def config3_instances(n_copies: int = 256, n_vars: int = 100, seed: int = 2, planted: bool = False):
    """One config-2 instance replicated (BASELINE config 3)."""
    inst = (hard_3sat if planted else threshold_3sat)(n_vars, seed=seed * 1000)[:2]
    return [inst] * n_copies


# ---------------------------------------------------------------------- solver-filtered generators
# The reference's own distributions: same clause draws as the planted stand-ins
# above, but filtered by a SAT solver as the reference does -- here the host-side CDCL solver of
# the extension (include/dsat.h: dsat_host_solve), since PySAT is not available.

def solve(n_vars: int, clauses, conflict_limit: int = 0):
    """SAT check of a CNF on the host.  Returns (True, model bits[n_vars]) / (False, None) /
    (None, None) when ``conflict_limit`` (> 0) was reached."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    lens = np.fromiter((len(c) for c in clauses), dtype=np.int64, count=len(clauses))
    rowptr = np.zeros(len(clauses) + 1, dtype=np.int32)
    np.cumsum(lens, out=rowptr[1:])
    lits = np.fromiter((l for c in clauses for l in c), dtype=np.int32, count=int(rowptr[-1]))
    model = np.zeros(max(n_vars, 1), dtype=np.uint8)
    res = C.c_int32(-2)
    _lib.check(lib.dsat_host_solve(int(n_vars), len(clauses), rowptr.ctypes.data_as(C.c_void_p),
                                   lits.ctypes.data_as(C.c_void_p), int(conflict_limit), C.byref(res),
                                   model.ctypes.data_as(C.c_void_p)))
    if res.value == 1:
        return True, model[:n_vars].astype(np.int32)
    return (False, None) if res.value == 0 else (None, None)


def random_kcnf(n_vars: int, n_clauses: int, k: int, rng):
    """cnfgen.RandomKCNF(k, n, m) as data/CNFGen.py:47 uses it: m distinct clauses, each over k
    distinct variables with independent random signs."""
    seen, clauses = set(), []
    while len(clauses) < n_clauses:
        vs = np.sort(rng.choice(n_vars, size=k, replace=False))
        signs = rng.integers(0, 2, size=k)
        c = tuple(int(v + 1) if s else -int(v + 1) for v, s in zip(vs, signs))
        if c in seen:
            continue
        seen.add(c)
        clauses.append(list(c))
    return clauses


def threshold_3sat(n_vars: int = 100, seed: int = 0):
    """One draw of SAT_3_Instances._generator (data/CNFGen.py:40-58) for a fixed n: random 3-SAT
    with m = int(4.258 n + 58.26 n^(-2/3)) clauses, redrawn until the solver finds it satisfiable.
    Returns (n_vars, clauses, solution bits)."""
    rng = np.random.default_rng(seed)
    m = threshold_clause_count(n_vars)
    while True:
        clauses = random_kcnf(n_vars, m, 3, rng)
        sat, model = solve(n_vars, clauses)
        if sat:
            return n_vars, clauses, model


def sr_ksat(seed: int = 0, min_vars: int = 30, max_vars: int = 100, p_k_2: float = 0.3, p_geo: float = 0.4,
            solutions_multiplier: float = 1.0):
    """One draw of the NeuroSAT-style generator (data/k_sat.py:36-93): clauses of width
    k = (1 w.p. p_k_2 else 2) + Geometric(p_geo) are added while the formula stays satisfiable; the
    clause that makes it unsatisfiable gets its first literal negated (which makes the formula
    satisfiable again, NeuroSAT's SR(n) pair), duplicates/subsumed clauses are removed, and with
    ``solutions_multiplier`` > 1 some clauses are dropped to leave about that many times more models
    (:66-84).  Returns (n_vars, clauses)."""
    import math
    from .dimacs import DimacsFile
    rng = np.random.default_rng(seed)
    n = int(rng.integers(min_vars, max_vars + 1))
    clauses = []
    while True:
        k_base = 1 if rng.random() < p_k_2 else 2
        k = k_base + int(rng.geometric(p_geo))
        vs = rng.choice(n, size=min(n, k), replace=False)
        c = [int(v + 1) if rng.random() < 0.5 else -int(v + 1) for v in vs]
        sat, _ = solve(n, clauses + [c])
        if sat:
            clauses.append(c)
        else:
            clauses.append([-c[0]] + c[1:])
            break
    df = DimacsFile(clauses=clauses)
    df.reduce_clauses()                      # also removes subsumed clauses (k_sat.py:100-103)
    clauses = df.clauses()
    if solutions_multiplier > 1:
        m = len(clauses)
        x = pow(2.0, n / m)
        d = int(max(0, min(round(math.log(solutions_multiplier, x), 0), m - 1)))
        drop = set(int(i) for i in rng.choice(m, size=d, replace=False))
        clauses = [c for i, c in enumerate(clauses) if i not in drop]
    return n, clauses


def sr_ksat_native(seed: int = 0, min_vars: int = 30, max_vars: int = 100, p_k_2: float = 0.3, p_geo: float = 0.4,
                   solutions_multiplier: float = 10.0):
    """``sr_ksat`` generated by the extension itself (``dsat_host_gen_sr``: incremental CDCL instead of one
    solver call per clause from Python; its own random stream, so not draw-for-draw ``sr_ksat``).
    ``solutions_multiplier`` defaults to the reference's 10 (data/k_sat.py:21).  Returns a
    ``CompactInstance`` (flat arrays)."""
    import ctypes as C
    from . import _lib
    from .batch import CompactInstance
    lib = _lib.load()
    cap_c, cap_l = 64 * max_vars + 64, 64 * max_vars * 8
    while True:
        rowptr = np.zeros(cap_c, dtype=np.int32)
        lits = np.zeros(cap_l, dtype=np.int32)
        n, m = C.c_int32(0), C.c_int32(0)
        rc = lib.dsat_host_gen_sr(int(seed), int(min_vars), int(max_vars), float(p_k_2), float(p_geo),
                                  float(solutions_multiplier), C.byref(n), C.byref(m),
                                  rowptr.ctypes.data_as(C.c_void_p), cap_c, lits.ctypes.data_as(C.c_void_p), cap_l)
        if rc == 0:
            break
        if b"too small" not in lib.dsat_last_error():
            _lib.check(rc)
        cap_c, cap_l = cap_c * 4, cap_l * 4
    m = m.value
    return CompactInstance(n.value, rowptr[:m + 1].copy(), lits[:rowptr[m]].copy())


def sr_stream(count: int, seed: int = 4, threads: int = 0, **kw):
    """``count`` SR(n) instances of the config-4 stream (seeds seed*2^32 + i), generated on a thread pool
    (the ctypes call releases the GIL)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or min(16, os.cpu_count() or 1)
    base = int(seed) << 32
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(lambda i: sr_ksat_native(base + i, **kw), range(int(count))))
