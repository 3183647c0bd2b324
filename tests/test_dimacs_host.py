"""Host logic vs fixtures generated from the reference's own numpy-only classes
(tests/golden/dimacs_golden.json, made by tests/golden/make_dimacs_golden.py from
utils/DimacsFile.py and utils/VariableAssignment.py) and the reference's known-answer tests."""
import json
import os

import numpy as np
import pytest

from diffusionsat_amd import batch as B
from diffusionsat_amd.dimacs import DimacsFile, VariableAssignment
from diffusionsat_amd import instances as I

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dimacs_golden.json")))


def test_parse_matches_reference():
    assert sum("raises" in c for c in GOLD["parse"]) >= 5
    for case in GOLD["parse"]:
        if "raises" in case:                        # texts the reference's parser refuses: the same error class
            with pytest.raises(ValueError):
                DimacsFile().load_from_string(case["text"])
            assert case["raises"] == "ValueError"
            continue
        df = DimacsFile().load_from_string(case["text"])
        assert df.number_of_vars() == case["n_vars"]
        assert df.clauses() == case["clauses"]
        assert {str(k): v for k, v in df.b_values.items()} == case["b_values"]
        assert str(df) == case["str"]


def test_reduce_clauses_matches_reference():
    for case in GOLD["reduce"]:
        df = DimacsFile(clauses=[list(c) for c in case["clauses"]])
        df.reduce_clauses()
        assert df.clauses() == case["reduced_sorted"]
        assert sorted(map(tuple, df.clauses())) == sorted(map(tuple, case["reduced_raw"]))
        assert [len(c) for c in df.clauses()] == sorted(len(c) for c in case["reduced_raw"])


def test_reference_known_answers():
    # utils/test_DimacsFile.py:5-23
    for cl, want in (([[1, 2], [1, 2], [3]], [[3], [1, 2]]), ([[1, 2, 3], [1, 2], [3]], [[3], [1, 2]]),
                     ([[1, 2, -3], [1, -2], [1]], [[1]])):
        df = DimacsFile(clauses=cl)
        df.reduce_clauses()
        assert df.clauses() == want
    # utils/test_AllSolutions.py:7,16: 14 models over 5 variables
    ka = GOLD["known_answer"]["all_solutions"]
    a = VariableAssignment(n_vars=ka["n_vars"], clauses=ka["clauses"])
    n = 0
    for m in range(1 << ka["n_vars"]):
        a.assign_all_from_int(m)
        n += a.satisfiable()
    assert n == ka["model_count"]


def test_assignment_encoding_matches_reference():
    for case in GOLD["assign"]:
        a = VariableAssignment(n_vars=case["n_vars"], clauses=case["clauses"])
        a.assign_all_from_bit_list(case["bits"])
        assert str(int(a)) == case["int"]
        assert str(a) == case["str"]
        assert a.satisfiable() == case["satisfiable"]
        assert a.as_int_list() == case["int_list"]
        b = VariableAssignment(n_vars=case["n_vars"], clauses=case["clauses"])
        b.assign_all_from_int(int(case["int"]))
        assert [int(v) for v in b.values()] == case["roundtrip_bits"]


def test_dimacs_store_load_roundtrip(tmp_path):
    df = DimacsFile(str(tmp_path / "a.cnf"), clauses=[[1, -2, 3], [-1], [2, 3]])
    df.store("a comment")
    back = DimacsFile(str(tmp_path / "a.cnf")).load()
    assert back.clauses() == df.clauses() and back.number_of_vars() == 3


def test_make_batch_layout():
    insts = [(3, [[1, -2], [3]]), (2, [[-1, 2], [1]]), (4, [])]
    b = B.make_batch(insts)
    assert (b.n_vars, b.n_clauses, b.n_graphs, b.n_edges) == (9, 4, 3, 6)
    assert b.clause_rowptr.tolist() == [0, 2, 3, 5, 6]
    assert b.clause_lits.tolist() == [1, -2, 3, -4, 5, 4]      # variables of graph 1 are offset by 3
    assert b.var_graph_id.tolist() == [0, 0, 0, 1, 1, 2, 2, 2, 2]
    assert b.clause_graph_id.tolist() == [0, 0, 1, 1]
    assert b.n_nodes == 2 * 9 + 4
    with pytest.raises(ValueError):
        B.make_batch([(2, [[3]])])
    with pytest.raises(ValueError):
        B.make_batch([(2, [[0]])])


def test_node_budget_batching_and_reference_drop_bug():
    insts = [(10, [[1]] * 30)] * 5          # 50 nodes each
    fixed = B.batch_by_node_budget(insts, max_nodes_per_batch=120)
    assert [len(b) for b in fixed] == [2, 2, 1]
    ref = B.batch_by_node_budget(insts, max_nodes_per_batch=120, reference_drop_bug=True)
    assert [len(b) for b in ref] == [2, 2]            # data/dimac.py:280-287 loses the overflowing formula
    big = B.batch_by_node_budget([(100, [[1]] * 1000)], max_nodes_per_batch=120)
    assert [len(b) for b in big] == [1]               # an oversized formula still forms a batch alone (:281)
    assert B.replicate_instance(100, [[1]] * 428, 0, 20000) == 31
    assert B.replicate_instance(100, [[1]] * 428, 10, 20000) == 10


def test_generators_are_satisfied_by_their_planted_assignment():
    assert I.threshold_clause_count(100) == 428        # data/CNFGen.py:72-73
    for gen in (lambda: I.hard_3sat(40, seed=3), I.uf20_91, lambda: I.neurosat_like_ksat(seed=2),
                lambda: I.irregular_cnf(300, 1500, seed=1)):
        n, clauses, planted = gen()
        a = VariableAssignment(n_vars=n, clauses=clauses)
        a.assign_all_from_bit_list(planted)
        assert a.satisfiable()
        assert all(1 <= abs(l) <= n for c in clauses for l in c)
    c2 = I.config2_instances()
    b = B.make_batch(c2)
    assert (b.n_vars, b.n_clauses, b.n_edges, b.n_graphs, b.n_nodes) == (3100, 13268, 39804, 31, 19468)
