"""Generate tests/golden/dimacs_golden.json from the reference's own numpy-only
classes (utils/DimacsFile.py, utils/VariableAssignment.py), which ARE importable
in the build container.  Run once in the container where /root/reference exists:

    python tests/golden/make_dimacs_golden.py

The JSON holds inputs and the reference's outputs only (data, no source).  It
pins the build's DIMACS parser, clause reduction, bit-list <-> int encoding and
``satisfiable()``.  Also holds the reference's own known-answer
tests: utils/test_DimacsFile.py:5-23 and the 14-model formula of
utils/test_AllSolutions.py:7,16.
"""
import importlib.util
import json
import os
import random
import sys

REF = "/root/reference/utils"


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    DimacsFile = _load("DimacsFile").DimacsFile
    VariableAssignment = _load("VariableAssignment").VariableAssignment
    rnd = random.Random(1234)
    out = {"parse": [], "reduce": [], "assign": [], "known_answer": {}}

    texts = [
        "p cnf 3 2\n1 -2 0\n2 3 0",
        "c a comment\np cnf 5 3\n1 -5 4 0\n-1 5 3 4 0\n-3 -4 0\n",
        "c sol 1 0 1\np cnf 3 2\n1 2 0\n-- awkward\n3 0\nv 1 -2 3 0\n",
        "p cnf 2 1\n 1   2   0   \n\n-7 0\n",
        "p cnf 4 2\n1 2 0 3 4\n-1 -2 0\n",
        # corner cases of the line rules (round 3: the parser was re-derived from these behaviours)
        "p cnf 12\n1 -2 0\n",                                  # header without a clause count
        "  c indented comment\n\tp cnf 6 1\t\n\t1\t-6\t0\n",   # tabs and leading blanks
        "junk before p cnf 9 1 trailing words\n3 -4 0\n",       # "p cnf" not at the start of its line
        "p cnf 3 1\nv1 -2 0 3\nV 3\nx 5\n1 2 3 0\n",              # v-lines (glued first number, zero inside), other letters
        "p cnf 3 1\nvalues 2\n1 2 3 0\n",                        # a word that starts with v
        "p cnf 2 3\n0\n1 0\n--2 0\n-- 1 2\n2\n",                  # empty clause, "--" lines, no terminating zero
        "p cnf 2 1\n- 1 0\n",                                    # a lone minus sign
        "c only comments\nc p cnf in a comment 5 5\n",            # a comment line that contains "p cnf"
        "c p cnf 5 5\n1 0\n",                                     # ... with numbers right behind it
        "p cnf 7 1 \n%\n0\n",                                    # SATLIB trailer lines
        "p cnf 3 2\r\n1 -3 0\r\n2 0\r\n",                       # CRLF
    ]
    for _ in range(6):
        n = rnd.randint(3, 12)
        m = rnd.randint(1, 20)
        lines = ["p cnf %d %d" % (n, m)]
        for _ in range(m):
            k = rnd.randint(1, 5)
            lits = [rnd.choice([-1, 1]) * rnd.randint(1, n) for _ in range(k)]
            lines.append(" ".join(map(str, lits)) + " 0")
        texts.append("\n".join(lines))
    for t in texts:
        df = DimacsFile()
        try:
            df.load_from_string(t)
        except Exception as e:                      # a text the reference refuses: the class of the error is the fixture
            out["parse"].append({"text": t, "raises": type(e).__name__})
            continue
        out["parse"].append({"text": t, "n_vars": df.number_of_vars(), "clauses": df.clauses(),
                             "b_values": {str(k): v for k, v in df.b_values.items()},
                             "str": str(df)})

    reduce_inputs = [
        [[1, 2], [1, 2], [3]],            # utils/test_DimacsFile.py:5-10
        [[1, 2, 3], [1, 2], [3]],         # :13-17
        [[1, 2, -3], [1, -2], [1]],       # :20-23
    ]
    for _ in range(8):
        n = rnd.randint(3, 8)
        cl = [[rnd.choice([-1, 1]) * v for v in rnd.sample(range(1, n + 1), rnd.randint(1, min(4, n)))]
              for _ in range(rnd.randint(2, 14))]
        reduce_inputs.append(cl)
    for cl in reduce_inputs:
        df = DimacsFile(clauses=[list(c) for c in cl])
        df.reduce_clauses()
        res = df.clauses()
        # order among equal-length clauses follows Python set iteration in the reference;
        # store the canonical (len, lexicographic) ordering alongside the raw result
        out["reduce"].append({"clauses": cl, "reduced_raw": res,
                              "reduced_sorted": sorted(res, key=lambda c: (len(c), c))})

    for _ in range(12):
        n = rnd.randint(1, 70)
        bits = [rnd.randint(0, 1) for _ in range(n)]
        cl = [[rnd.choice([-1, 1]) * v for v in rnd.sample(range(1, n + 1), rnd.randint(1, min(3, n)))]
              for _ in range(rnd.randint(1, 12))]
        a = VariableAssignment(n_vars=n, clauses=cl)
        a.assign_all_from_bit_list(bits)
        b = VariableAssignment(n_vars=n, clauses=cl)
        b.assign_all_from_int(int(a))
        out["assign"].append({"n_vars": n, "bits": bits, "clauses": cl, "int": str(int(a)),
                              "str": str(a), "satisfiable": bool(a.satisfiable()),
                              "int_list": a.as_int_list(), "roundtrip_bits": [int(v) for v in b.values()]})

    # utils/test_AllSolutions.py:7,16 -- this formula over 5 variables has 14 models
    out["known_answer"]["all_solutions"] = {"clauses": [[-1, 2], [1, -2], [-3, 4, 5]], "n_vars": 5,
                                            "model_count": 14}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dimacs_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
