"""Second, independently written CPU restatement (PyTorch-CPU, fp32) of the round.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED.

Purpose: (1) cross-check ``querysat_oracle`` -- this file shares no code with it:
sparse products are edge-list ``index_add_`` gathers instead of scipy CSR, the
inner gradient (model/query_sat.py:227-246) is taken with ``torch.autograd``
instead of the closed form, PairNorm uses per-graph ``index_add_`` means;
(2) the ``cpu_baseline`` leg of ``bench.py`` ("CPU restatement, TF unavailable",
SURVEY.md §8d) -- multi-threaded MKL GEMMs + vectorised gathers.
"""
from __future__ import annotations

import numpy as np
import torch

F = 128


class TorchBatch:
    """Edge lists of the batched bipartite graph (data/SatSpecifics.py:21-69)."""

    def __init__(self, instances):
        lit_idx, cl_idx = [], []
        vg, cgid = [], []
        off, c = 0, 0
        for g, (n, clauses) in enumerate(instances):
            for cl in clauses:
                for lit in cl:
                    v = abs(lit) - 1 + off
                    lit_idx.append((v, lit > 0))
                    cl_idx.append(c)
                c += 1
                cgid.append(g)
            vg += [g] * n
            off += n
        self.n, self.c, self.g = off, c, len(instances)
        v = torch.tensor([a for a, _ in lit_idx], dtype=torch.long)
        pos = torch.tensor([b for _, b in lit_idx], dtype=torch.bool)
        self.edge_lit = torch.where(pos, v, v + self.n)       # literal row per edge
        self.edge_cl = torch.tensor(cl_idx, dtype=torch.long)
        self.var_gid = torch.tensor(vg, dtype=torch.long)
        self.cl_gid = torch.tensor(cgid, dtype=torch.long)
        ones = torch.ones(len(cl_idx))
        self.lit_deg = torch.zeros(2 * self.n).index_add_(0, self.edge_lit, ones).unsqueeze(1)
        self.cl_deg = torch.zeros(self.c).index_add_(0, self.edge_cl, ones).unsqueeze(1)
        self.var_cnt = torch.zeros(self.g).index_add_(0, self.var_gid, torch.ones(self.n))
        self.cl_cnt = torch.zeros(self.g).index_add_(0, self.cl_gid, torch.ones(self.c))

    def lit_to_clause(self, x):   # A^T @ x
        return torch.zeros(self.c, x.shape[1]).index_add_(0, self.edge_cl, x[self.edge_lit])

    def clause_to_lit(self, x):   # A @ x
        return torch.zeros(2 * self.n, x.shape[1]).index_add_(0, self.edge_lit, x[self.edge_cl])


def _mlp(x, layers):
    for i, (w, b) in enumerate(layers):
        x = torch.addmm(b, x, w)
        if i + 1 < len(layers):
            x = torch.nn.functional.leaky_relu(x, 0.2)
    return x


def _pair_norm(x, gid, cnt, g):
    mean = torch.zeros(g, x.shape[1]).index_add_(0, gid, x) / cnt.unsqueeze(1)
    x = x - mean[gid]
    return x * torch.rsqrt(x.square().mean(dim=1, keepdim=True) + 1e-6)


def _kl(a, b):
    return a * (a.log() - b.log()) + (1 - a) * (torch.log1p(-a) - torch.log1p(-b))


def to_torch_weights(weights):
    return {k: [(torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)),
                 torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32))) for w, b in v]
            for k, v in weights.items()}


def diffusion_step(tw, tb: TorchBatch, noise_scale, noisy_num, normal, labels, rounds,
                   use_autograd=False, all_logits=False):
    """One denoising step, early exit OFF (fixed work); see SURVEY.md §2.3 R1-R16."""
    n, c, g = tb.n, tb.c, tb.g
    variables = torch.ones(n, F)
    clause_state = torch.ones(c, F)
    deg_w = torch.rsqrt(tb.lit_deg.clamp(min=1))
    vdeg_w = 4 * torch.rsqrt((tb.lit_deg[:n] + tb.lit_deg[n:]).clamp(min=1))
    rev_w = torch.rsqrt(tb.cl_deg.clamp(min=1))
    noisy_num = torch.as_tensor(noisy_num, dtype=torch.float32)
    nl = torch.cat([noisy_num, torch.full((n, 1), float(noise_scale)), torch.zeros(n, 2)], dim=1)
    labels = torch.as_tensor(labels).float().unsqueeze(1).expand(n, 8)
    normal = torch.as_tensor(normal, dtype=torch.float32)
    t = float(noise_scale) ** 0.5
    tl = min(t + 0.01, 1.0)
    a = labels * (1 - tl) + tl / 2
    norm = _kl(torch.tensor(tl / 2), torch.tensor(0.5)) + 1e-4
    logits_all = []
    logits = None
    for r in range(rounds):
        v1 = torch.cat([variables, normal[r], nl], dim=1)
        if use_autograd:
            q = _mlp(v1, tw["variables_query"]).detach().requires_grad_(True)
            closs = torch.exp(-tb.lit_to_clause(torch.nn.functional.softplus(torch.cat([q, -q], 0))))
            (grad,) = torch.autograd.grad(closs.sum(), q)
            q, closs = q.detach(), closs.detach()
            grad = grad * vdeg_w
        else:
            q = _mlp(v1, tw["variables_query"])
            closs = torch.exp(-tb.lit_to_clause(torch.nn.functional.softplus(torch.cat([q, -q], 0))))
            m = tb.clause_to_lit(closs)
            grad = (-torch.sigmoid(q) * m[:n] + torch.sigmoid(-q) * m[n:]) * vdeg_w
        vm = _mlp(v1, tw["lit_mlp"])
        lits = torch.cat([vm[:, :F], vm[:, F:]], 0)
        cmsg = tb.lit_to_clause(lits) * rev_w
        cd = _mlp(torch.cat([clause_state, cmsg, closs * 4], 1), tw["clause_mlp"])
        clause_state = _pair_norm(cd[:, F:], tb.cl_gid, tb.cl_cnt, g) * 0.25 + 0.1 * clause_state
        vl = tb.clause_to_lit(cd[:, :F]) * deg_w
        nv = _mlp(torch.cat([grad, v1, vl[:n], vl[n:]], 1), tw["update_gate"])
        variables = _pair_norm(nv, tb.var_gid, tb.var_cnt, g) * 0.25 + 0.1 * variables
        logits = _mlp(variables, tw["variables_output"])
        if all_logits:
            logits_all.append(logits.clone())
    b = torch.sigmoid(logits) * (1 - t) + t / 2
    pvl = _kl(a, b) / norm
    pgl = torch.zeros(g, 8).index_add_(0, tb.var_gid, pvl) / tb.var_cnt.unsqueeze(1)
    best = pgl.argmin(dim=1)
    pred = logits.gather(1, best[tb.var_gid].unsqueeze(1)).squeeze(1)
    return {"prediction": pred, "logits": logits, "best_map": best, "logits_all": logits_all,
            "per_graph_loss": pgl}
