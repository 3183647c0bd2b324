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


def make_batch(instances: Sequence[Instance]) -> Batch:
    """Concatenate ``[(n_vars, clauses), ...]`` into one batch (no node-budget check)."""
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
    order = sorted(range(len(instances)),
                   key=lambda i: sat_node_count(int(instances[i][0]), len(instances[i][1])))
    batches: List[List[Instance]] = []
    cur: List[Instance] = []
    nodes = 0
    for i in order:
        inst = instances[i]
        cnt = sat_node_count(int(inst[0]), len(inst[1]))
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
