"""Host-side CDCL solver (dsat_host_solve) and the solver-filtered generators that replace the
reference's PySAT-based ones (data/CNFGen.py:40-58, data/k_sat.py:36-93).  CPU only."""
import itertools

import numpy as np
import pytest

from diffusionsat_amd import instances as I
from oracle import querysat_oracle as O


def _brute(n, clauses):
    return [bits for bits in itertools.product([0, 1], repeat=n) if O.satisfiable(clauses, bits)]


def test_solver_agrees_with_brute_force_on_random_small_formulas():
    rng = np.random.default_rng(0)
    n_sat = 0
    for _ in range(400):
        n = int(rng.integers(1, 11))
        clauses = []
        for _ in range(int(rng.integers(0, 60))):
            vs = rng.choice(n, size=min(n, int(rng.integers(1, 4))), replace=False)
            clauses.append([int(v + 1) if rng.random() < 0.5 else -int(v + 1) for v in vs])
        sat, model = I.solve(n, clauses)
        assert sat == bool(_brute(n, clauses))
        if sat:
            n_sat += 1
            assert O.satisfiable(clauses, model)
    assert 50 < n_sat < 350        # both answers really occur


def test_solver_known_answers():
    # the 14-model formula of the reference's own tests (utils/test_AllSolutions.py:7,16)
    clauses = [[-1, 2], [1, -2], [-3, 4, 5]]
    assert len(_brute(5, clauses)) == 14 and I.solve(5, clauses)[0] is True
    # blocking every model one by one enumerates exactly 14 before UNSAT
    found, cl = 0, [list(c) for c in clauses]
    while True:
        sat, m = I.solve(5, cl)
        if not sat:
            break
        found += 1
        cl.append([-(i + 1) if b else (i + 1) for i, b in enumerate(m)])
    assert found == 14
    # pigeonhole 7 -> 6: unsatisfiable, needs real conflict analysis
    P, H = 7, 6
    php = [[p * H + h + 1 for h in range(H)] for p in range(P)]
    php += [[-(p1 * H + h + 1), -(p2 * H + h + 1)] for h in range(H) for p1 in range(P) for p2 in range(p1 + 1, P)]
    assert I.solve(P * H, php)[0] is False
    # edge cases: no clauses, an empty clause, duplicate / tautological literals, conflict limit
    assert I.solve(3, [])[0] is True
    assert I.solve(3, [[1], []])[0] is False
    assert I.solve(2, [[1, 1, -2], [2, -2], [-1]])[0] is True
    assert I.solve(P * H, php, conflict_limit=3)[0] is None
    with pytest.raises(Exception):
        I.solve(2, [[3]])


def test_threshold_3sat_is_the_references_distribution():
    """data/CNFGen.py:40-58: m = int(4.258 n + 58.26 n^(-2/3)) distinct 3-clauses over distinct
    variables, satisfiable (the returned solution satisfies it)."""
    for seed, n in ((0, 20), (1, 50), (2, 100)):
        nv, clauses, sol = I.threshold_3sat(n, seed=seed)
        assert nv == n and len(clauses) == I.threshold_clause_count(n)
        assert all(len(c) == 3 and len({abs(l) for l in c}) == 3 for c in clauses)
        assert len({tuple(c) for c in clauses}) == len(clauses)
        assert O.satisfiable(clauses, sol)
    assert I.threshold_clause_count(100) == 428
    a, b = I.threshold_3sat(30, seed=5), I.threshold_3sat(30, seed=5)
    assert a[1] == b[1]                                  # seeded
    c2 = I.config2_instances()
    assert len(c2) == 31 and sum(2 * n + len(c) for n, c in c2) == 19468


def test_sr_ksat_follows_the_neurosat_construction():
    """data/k_sat.py:36-93: satisfiable, and one literal flip away from an unsatisfiable core."""
    for seed in range(4):
        n, clauses = I.sr_ksat(seed=seed, min_vars=10, max_vars=25)
        assert 10 <= n <= 25 and len(clauses) > n
        assert all(1 <= len(c) <= n and len({abs(l) for l in c}) == len(c) for c in clauses)
        assert I.solve(n, clauses)[0] is True
    n, few = I.sr_ksat(seed=1, min_vars=12, max_vars=12)
    n2, more = I.sr_ksat(seed=1, min_vars=12, max_vars=12, solutions_multiplier=20)
    assert n2 == n and len(more) < len(few)
    assert len(_brute(n, more)) >= len(_brute(n, few)) >= 1


def test_native_sr_generator_properties():
    """dsat_host_gen_sr (the reference's data/k_sat.py:36-93 generator, natively): deterministic per seed,
    every instance satisfiable, clauses reduced (no duplicates / subsumed clauses, literals sorted, order by
    (length, lexicographic) as dimacs.reduce_clauses), sizes and widths in the distribution's range, and --
    without clause removal -- negating the first literal of one clause back gives an UNSAT formula, i.e.
    the instance sits right at the SAT/UNSAT boundary as NeuroSAT's SR(n) pairs do."""
    from diffusionsat_amd.dimacs import DimacsFile
    a, b = I.sr_ksat_native(11), I.sr_ksat_native(11)
    assert a.n_vars == b.n_vars and np.array_equal(a.rowptr, b.rowptr) and np.array_equal(a.lits, b.lits)
    assert not np.array_equal(a.lits, I.sr_ksat_native(12).lits)
    ratios, widths = [], []
    for seed in range(40):
        x = I.sr_ksat_native(1000 + seed, solutions_multiplier=1.0)
        n, cl = x.as_instance()
        assert 30 <= n <= 100 and all(1 <= abs(l) <= n for c in cl for l in c)
        assert I.solve(n, cl)[0] is True
        df = DimacsFile(clauses=[list(c) for c in cl])
        df.reduce_clauses()
        assert df.clauses() == cl                                   # already reduced, same canonical order
        # exactly one clause, with its first literal negated back, makes the formula unsatisfiable
        flips = sum(I.solve(n, cl[:i] + [[-c[0]] + c[1:]] + cl[i + 1:])[0] is False
                    for i, c in enumerate(cl) if len(c) >= 1)
        assert flips >= 1
        ratios.append(len(cl) / n)
        widths.extend(len(c) for c in cl)
    assert 3.5 < np.mean(ratios) < 7.0 and 3.5 < np.mean(widths) < 5.0 and min(widths) >= 1
    # the reference's default multiplier (10) removes a few clauses: still satisfiable, fewer clauses
    x1, x10 = I.sr_ksat_native(5, solutions_multiplier=1.0), I.sr_ksat_native(5, solutions_multiplier=10.0)
    assert x10.n_vars == x1.n_vars and 0 < x1.n_clauses - x10.n_clauses < 40
    assert I.solve(*x10.as_instance())[0] is True
    # same distribution as the Python restatement (different random stream): clause/variable ratio
    py = [I.sr_ksat(seed=s) for s in range(12)]
    assert abs(np.mean([len(c) / n for n, c in py]) - np.mean(ratios)) < 1.0


def test_compact_batches_equal_list_batches():
    from diffusionsat_amd.batch import CompactInstance, batch_by_node_budget, make_batch
    xs = [I.sr_ksat_native(50 + i, min_vars=5, max_vars=20) for i in range(12)]
    b1, b2 = make_batch(xs), make_batch([x.as_instance() for x in xs])
    for f in ("clause_rowptr", "clause_lits", "var_graph_id", "clause_graph_id", "variables_in_graph",
              "clauses_in_graph"):
        assert np.array_equal(getattr(b1, f), getattr(b2, f)), f
    assert (b1.n_vars, b1.n_clauses, b1.n_graphs) == (b2.n_vars, b2.n_clauses, b2.n_graphs)
    g1 = batch_by_node_budget(xs, 300)
    g2 = batch_by_node_budget([x.as_instance() for x in xs], 300)
    assert [[i.n_vars for i in g] for g in g1] == [[n for n, _ in g] for g in g2]
    rt = CompactInstance.from_instance(xs[0].as_instance())
    assert np.array_equal(rt.lits, xs[0].lits) and np.array_equal(rt.rowptr, xs[0].rowptr)
    with pytest.raises(ValueError):
        make_batch([CompactInstance(2, [0, 1], [3])])
