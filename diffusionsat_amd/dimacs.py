"""DIMACS CNF front-end and assignment helpers (host side, pure Python/NumPy).

Mirrors the behaviour of the reference's two numpy-only helper classes so the
sampler façade returns the same dictionary the reference does:

  * ``DimacsFile``          -- utils/DimacsFile.py:5-128 (parse, store, reduce_clauses)
  * ``VariableAssignment``  -- utils/VariableAssignment.py:6-112 (bit list <-> int,
    ``satisfiable``)

Their outputs are pinned by fixtures generated from the reference classes
themselves (``tests/golden/dimacs_golden.json``).
"""
from __future__ import annotations

import itertools
from typing import Iterable, List, Sequence


class DimacsFile:
    """CNF container that reads DIMACS text the way the reference's class does (utils/DimacsFile.py:39-85).

    The line rules below are written down from observed behaviour: ``tests/golden/dimacs_golden.json`` holds
    what the reference class itself returns -- or which error it raises -- for ordinary files and for every
    corner named here (``tests/golden/make_dimacs_golden.py``).

    * blank lines are skipped; surrounding blanks, tabs and ``\r`` never matter;
    * a line that contains ``p cnf`` is the header.  The variable count is read at a fixed place, right
      behind the line's first five characters, and ends at the next space -- so a header that does not
      start its line, or a comment that mentions ``p cnf``, is a ``ValueError``, and a header without a
      clause count loses its last digit (``p cnf 12`` reads 1);
    * other lines that start with a letter are comments, except ``v`` lines: every non-zero number on
      them assigns a variable (``b_values``), the first may be glued to the ``v``;
    * lines that start with ``--`` are skipped;
    * anything else is a clause: the numbers up to the first ``0`` (what follows it is not looked at; a
      missing ``0`` is fine, a lone ``0`` is the empty clause).  SATLIB's ``%`` trailer is a ``ValueError``;
    * ``n_vars`` grows to cover every literal seen.
    """

    HEADER = "p cnf"

    def __init__(self, filename: str = "file.cnf", n_vars: int = 0, clauses=None):
        clauses = [list(map(int, c)) for c in clauses] if clauses is not None else []
        if n_vars == 0 and clauses:
            n_vars = max(abs(l) for c in clauses for l in c)
        self.filename = filename
        self.n_vars = int(n_vars)
        self.i_clauses: List[List[int]] = clauses
        self.b_values = {}
        self.comments: List[str] = []

    # -- loading -------------------------------------------------------------
    def load(self):
        with open(self.filename, "r") as f:
            self.load_from_lines(f.readlines())
        return self

    def load_from_string(self, text: str):
        self.load_from_lines(text.split("\n"))
        return self

    @classmethod
    def _kind(cls, line: str) -> str:
        if not line:
            return "blank"
        if cls.HEADER in line:
            return "header"
        if line[0].isalpha():
            return "values" if line[0] == "v" else "comment"
        return "comment" if line[:2] == "--" else "clause"

    @classmethod
    def _header_count(cls, line: str) -> int:
        field = line[len(cls.HEADER):].lstrip()
        return int(field[:field.find(" ")])          # no further space: find() is -1 and the last character goes

    def load_from_lines(self, lines: Iterable[str]):
        self.n_vars, self.i_clauses = 0, []
        for line in map(str.strip, lines):
            kind = self._kind(line)
            if kind == "header":
                self.n_vars = self._header_count(line)
            elif kind == "values":
                self.b_values.update((abs(v), v > 0) for v in map(int, line[1:].split()) if v)
            elif kind == "clause":
                # lazily: nothing behind the terminating zero is parsed
                self.add_clause(list(itertools.takewhile(bool, map(int, line.split()))))

    # -- accessors -----------------------------------------------------------
    def number_of_vars(self) -> int:
        return self.n_vars

    def number_of_clauses(self) -> int:
        return len(self.i_clauses)

    def clauses(self) -> List[List[int]]:
        return self.i_clauses

    def add_comment(self, comment: str):
        self.comments.append(comment)

    def add_clause(self, clause: Sequence[int]):
        self.add_clauses([clause])

    def add_clauses(self, clauses):
        for c in clauses:
            c = list(c)
            for lit in c:
                self.n_vars = max(self.n_vars, abs(lit))
            self.i_clauses.append(c)

    def reduce_clauses(self):
        """Dedup + subsumption, shortest clauses first (utils/DimacsFile.py:110-128).

        The reference dedups through a Python ``set`` and then stable-sorts by
        length, so the relative order of equal-length survivors follows set
        iteration order; here ties are ordered lexicographically, which is a
        deterministic choice among the orders the reference can produce.
        """
        uniq = sorted({tuple(sorted(c)) for c in self.i_clauses}, key=lambda c: (len(c), c))
        kept: List[tuple] = []
        for c in uniq:
            cs = set(c)
            if any(set(k) <= cs for k in kept):
                continue
            kept.append(c)
        self.i_clauses = [list(c) for c in kept]

    def is_satisfiable(self) -> bool:
        for i in range(1, self.n_vars + 1):
            if i not in self.b_values:
                raise Exception("Not all variables have values. Variable " + str(i) + " does not.")
        for c in self.i_clauses:
            if not any((l > 0) == self.b_values[abs(l)] for l in c):
                return False
        return True

    def __str__(self) -> str:
        out = ["p cnf %d %d" % (self.n_vars, len(self.i_clauses))]
        for c in self.i_clauses:
            out.append("".join("%d " % l for l in c) + "0")
        return "\n".join(out) + "\n"

    def store(self, *comments: str):
        with open(self.filename, "w") as f:
            for c in list(self.comments) + list(comments):
                f.write("c " + c + "\n")
            f.write(str(self))


class VariableAssignment:
    """utils/VariableAssignment.py: bit 0 of the int encodes x1 (``__int__``, :63-69)."""

    def __init__(self, n_vars: int = 0, clauses=None):
        clauses = [list(map(int, c)) for c in clauses] if clauses is not None else []
        if n_vars == 0 and clauses:
            n_vars = max(abs(l) for c in clauses for l in c)
        self.x = [False] * int(n_vars)
        self.clauses = clauses

    def assign(self, i: int, value: bool):
        self.x[i] = bool(value)

    def assign_all(self, x):
        self.x = [bool(b) for b in x]

    def assign_all_from_int_list(self, x):
        for i in x:
            self.assign(abs(i) - 1, i > 0)

    def assign_all_from_bit_list(self, x):
        for bit_no, bit in enumerate(x):
            self.assign(bit_no, int(bit) == 1)

    def assign_all_from_int(self, i: int):
        for bit_no in range(len(self.x)):
            self.assign(bit_no, (i >> bit_no) & 1 != 0)

    def __int__(self) -> int:
        res = 0
        for bit_no, b in enumerate(self.x):
            if b:
                res |= 1 << bit_no
        return res

    def __str__(self) -> str:
        return "".join("1" if b else "0" for b in self.x)

    def satisfiable(self) -> bool:
        for c in self.clauses:
            if not any((lit > 0) == self.x[abs(lit) - 1] for lit in c):
                return False
        return True

    def as_int_list(self):
        return [(i + 1) if b else -(i + 1) for i, b in enumerate(self.x)]

    def value(self, i: int) -> bool:
        return self.x[i]

    def values(self):
        return self.x
