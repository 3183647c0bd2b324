"""Host-side batching of CNF instances into one block-diagonal clause-literal graph.

Produces the flat arrays the C-ABI's ``dsat_graph_create`` consumes (include/dsat.h):
clause CSR over *signed, batch-numbered DIMACS literals* plus per-variable and
per-clause graph ids.  Semantics follow the reference's batch construction:

  * variable offsets per formula           -- data/dimac.py:166-171, 239-241
  * literal rows (positive 0..N-1, negative N..2N-1) -- data/SatSpecifics.py:25-35
  * graph membership by repetition         -- data/SatSpecifics.py:40-57
  * node budget ``2*n_vars + n_clauses``   -- data/dimac.py:172-174
  * greedy batching sorted by node count   -- data/dimac.py:176-186, 267-293
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

Instance = Tuple[int, Sequence[Sequence[int]]]


class CompactInstance:
    """One formula as flat arrays (clause CSR over signed DIMACS literals): what the native generators
    emit and what the streaming pipeline batches without touching Python lists."""
    __slots__ = ("n_vars", "rowptr", "lits")

    def __init__(self, n_vars: int, rowptr, lits):
        self.n_vars = int(n_vars)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        self.lits = np.ascontiguousarray(lits, dtype=np.int32)

    @property
    def n_clauses(self) -> int:
        return int(self.rowptr.shape[0]) - 1

    def clauses(self) -> List[List[int]]:
        r, l = self.rowptr, self.lits
        return [[int(x) for x in l[r[c]:r[c + 1]]] for c in range(self.n_clauses)]

    def as_instance(self) -> Instance:
        return self.n_vars, self.clauses()

    @staticmethod
    def from_instance(inst) -> "CompactInstance":
        if isinstance(inst, CompactInstance):
            return inst
        n, clauses = inst[0], inst[1]
        lens = np.fromiter((len(c) for c in clauses), dtype=np.int64, count=len(clauses))
        rowptr = np.zeros(len(clauses) + 1, dtype=np.int32)
        np.cumsum(lens, out=rowptr[1:])
        lits = np.fromiter((l for c in clauses for l in c), dtype=np.int32, count=int(rowptr[-1]))
        return CompactInstance(n, rowptr, lits)


def instance_size(inst) -> Tuple[int, int]:
    """(n_vars, n_clauses) of either instance representation."""
    if isinstance(inst, CompactInstance):
        return inst.n_vars, inst.n_clauses
    return int(inst[0]), len(inst[1])


@dataclass
class Batch:
    n_vars: int                 # N, total variables
    n_clauses: int              # C
    n_graphs: int               # G
    clause_rowptr: np.ndarray   # int32 [C+1]
    clause_lits: np.ndarray     # int32 [E], signed DIMACS literal in batch numbering (1-based)
    var_graph_id: np.ndarray    # int32 [N], non-decreasing
    clause_graph_id: np.ndarray  # int32 [C], non-decreasing
    variables_in_graph: np.ndarray  # int32 [G]
    clauses_in_graph: np.ndarray    # int32 [G]
    instances: List[Instance]

    @property
    def n_edges(self) -> int:
        return int(self.clause_lits.shape[0])

    @property
    def n_nodes(self) -> int:
        return 2 * self.n_vars + self.n_clauses


def sat_node_count(n_vars: int, n_clauses: int) -> int:
    """data/dimac.py:172-174."""
    return 2 * n_vars + n_clauses


def make_batch_compact(instances: Sequence[CompactInstance]) -> Batch:
    """``make_batch`` for flat-array instances, vectorised (the streaming pipeline's batcher)."""
    g = len(instances)
    nv = np.fromiter((i.n_vars for i in instances), dtype=np.int64, count=g)
    nc = np.fromiter((i.n_clauses for i in instances), dtype=np.int64, count=g)
    ne = np.fromiter((i.lits.shape[0] for i in instances), dtype=np.int64, count=g)
    voff = np.concatenate([[0], np.cumsum(nv)[:-1]]) if g else np.zeros(0, np.int64)
    eoff = np.concatenate([[0], np.cumsum(ne)[:-1]]) if g else np.zeros(0, np.int64)
    lits = np.concatenate([i.lits for i in instances]).astype(np.int64) if g else np.zeros(0, np.int64)
    for i in instances:
        if i.lits.size and (np.any(i.lits == 0) or int(np.abs(i.lits).max()) > i.n_vars):
            raise ValueError("literal out of range for a %d-variable formula" % i.n_vars)
    shift = np.repeat(voff, ne)
    lits = np.where(lits > 0, lits + shift, lits - shift).astype(np.int32)
    rowptr = np.concatenate([[0]] + [i.rowptr[1:].astype(np.int64) + o for i, o in zip(instances, eoff)]) \
        if g else np.zeros(1, np.int64)
    gids = np.arange(g, dtype=np.int32)
    return Batch(
        n_vars=int(nv.sum()), n_clauses=int(nc.sum()), n_graphs=g,
        clause_rowptr=rowptr.astype(np.int32), clause_lits=lits,
        var_graph_id=np.repeat(gids, nv), clause_graph_id=np.repeat(gids, nc),
        variables_in_graph=nv.astype(np.int32), clauses_in_graph=nc.astype(np.int32),
        instances=list(instances),
    )


def make_batch(instances: Sequence[Instance]) -> Batch:
    """Concatenate ``[(n_vars, clauses), ...]`` into one batch (no node-budget check)."""
    if instances and all(isinstance(i, CompactInstance) for i in instances):
        return make_batch_compact(instances)
    instances = [i.as_instance() if isinstance(i, CompactInstance) else i for i in instances]
    rowptr = [0]
    lits: List[int] = []
    vgid: List[np.ndarray] = []
    cgid: List[np.ndarray] = []
    vcount, ccount = [], []
    offset = 0
    for g, (n, clauses) in enumerate(instances):
        n = int(n)
        for c in clauses:
            for lit in c:
                lit = int(lit)
                if lit == 0 or abs(lit) > n:
                    raise ValueError("literal %d out of range for a %d-variable formula" % (lit, n))
                lits.append(lit + offset if lit > 0 else lit - offset)
            rowptr.append(len(lits))
        vgid.append(np.full(n, g, dtype=np.int32))
        cgid.append(np.full(len(clauses), g, dtype=np.int32))
        vcount.append(n)
        ccount.append(len(clauses))
        offset += n
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.int32)
    return Batch(
        n_vars=offset, n_clauses=len(rowptr) - 1, n_graphs=len(instances),
        clause_rowptr=np.asarray(rowptr, dtype=np.int32),
        clause_lits=np.asarray(lits, dtype=np.int32),
        var_graph_id=cat(vgid).astype(np.int32), clause_graph_id=cat(cgid).astype(np.int32),
        variables_in_graph=np.asarray(vcount, dtype=np.int32),
        clauses_in_graph=np.asarray(ccount, dtype=np.int32),
        instances=list(instances),
    )


def batch_by_node_budget(instances: Sequence[Instance], max_nodes_per_batch: int = 20000,
                         reference_drop_bug: bool = False) -> List[List[Instance]]:
    """Greedy node-budget batching of formulas sorted by node count.

    data/dimac.py:176-186 (sort by ``2n+m``) and :267-293 (``__batch_files``).  A
    formula larger than the budget still forms a batch alone (``nodes_in_batch==0``
    clause, :281).  The reference loses the formula that overflows a batch (it is
    appended to neither the closed nor the new batch, :280-287); that is a bug, so
    by default the overflowing formula opens the next batch.
    ``reference_drop_bug=True`` reproduces the reference's behaviour exactly.
    """
    order = sorted(range(len(instances)), key=lambda i: sat_node_count(*instance_size(instances[i])))
    batches: List[List[Instance]] = []
    cur: List[Instance] = []
    nodes = 0
    for i in order:
        inst = instances[i]
        cnt = sat_node_count(*instance_size(inst))
        if cnt + nodes <= max_nodes_per_batch or nodes == 0:
            cur.append(inst)
            nodes += cnt
        else:
            batches.append(cur)
            if reference_drop_bug:
                cur, nodes = [], 0
            else:
                cur, nodes = [inst], cnt
    if cur:
        batches.append(cur)
    return batches


def replicate_instance(n_vars: int, clauses, copies: int, max_nodes_per_batch: int = 20000):
    """Batches of identical copies of ONE formula, as the sampler's dataset yields them
    (data/diffusion_sat_instances.py:80-94: the test generator repeats the formula).
    Returns the number of copies per batch that fits the node budget (at least 1)."""
    per = sat_node_count(int(n_vars), len(clauses))
    fit = max(1, max_nodes_per_batch // per)
    return min(fit, copies) if copies > 0 else fit
