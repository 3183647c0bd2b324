// Host-side CDCL SAT solver for instance preparation (no device code in this file).
//
// The reference filters its generated formulas with a solver from PySAT: threshold 3-SAT keeps
// drawing until the formula is satisfiable (data/CNFGen.py:44-58, Glucose via
// satsolvers/Default.py), the NeuroSAT-style k-SAT stream adds clauses until the solver reports
// UNSAT (data/k_sat.py:44-60, pysat Solver()).  PySAT is not available here and would not be the
// product's dependency anyway; formulas on this path have at most a few hundred variables, for
// which a small conflict-driven solver (two watched literals, first-UIP learning, activity
// order with phase saving, geometric restarts) answers in well under a millisecond to a few ms.
// Only SAT/UNSAT (and one model) matter to the generators, so nothing here has to match
// Glucose's search order.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "dsat_internal.h"

namespace {

struct Cdcl {
    // literal encoding: 2*v for +x_{v+1}, 2*v+1 for -x_{v+1}
    int n = 0;
    std::vector<std::vector<int>> cls;          // clause database (originals + learnt)
    std::vector<std::vector<int>> watch;        // watch[lit] = clauses watching lit
    std::vector<int8_t> val;                    // per variable: -1 unassigned, 0 false, 1 true
    std::vector<int8_t> phase;
    std::vector<int> level, reason, trail, lim;
    std::vector<double> act;
    std::vector<char> seen;
    double inc = 1.0;
    size_t qhead = 0;
    bool ok = true;
    int64_t conflicts = 0;

    explicit Cdcl(int nv) : n(nv), watch(2 * nv), val(nv, -1), phase(nv, 0), level(nv, 0), reason(nv, -1),
                            act(nv, 0.0), seen(nv, 0) {}

    int lit_val(int l) const { const int8_t v = val[l >> 1]; return v < 0 ? -1 : (v ^ (l & 1)); }
    void enqueue(int l, int why) {
        const int v = l >> 1;
        val[v] = (int8_t)((l & 1) ^ 1);
        level[v] = (int)lim.size();
        reason[v] = why;
        trail.push_back(l);
    }
    void attach(int ci) { watch[cls[ci][0]].push_back(ci); watch[cls[ci][1]].push_back(ci); }

    // returns false when the formula became trivially unsatisfiable
    bool add(std::vector<int> c) {
        if (!ok) return false;
        std::sort(c.begin(), c.end());
        c.erase(std::unique(c.begin(), c.end()), c.end());
        for (size_t i = 0; i + 1 < c.size(); ++i)
            if ((c[i] ^ 1) == c[i + 1]) return true;   // tautology
        // (called at decision level 0 only) drop false literals, detect satisfied clauses
        std::vector<int> d;
        for (int l : c) {
            const int lv = lit_val(l);
            if (lv == 1) return true;
            if (lv < 0) d.push_back(l);
        }
        if (d.empty()) { ok = false; return false; }
        if (d.size() == 1) {
            enqueue(d[0], -1);
            if (propagate() >= 0) ok = false;
            return ok;
        }
        cls.push_back(d);
        attach((int)cls.size() - 1);
        return true;
    }

    int propagate() {   // returns the index of a conflicting clause or -1
        while (qhead < trail.size()) {
            const int fl = trail[qhead++] ^ 1;   // this literal just became false
            std::vector<int>& ws = watch[fl];
            size_t keep = 0;
            for (size_t i = 0; i < ws.size(); ++i) {
                const int ci = ws[i];
                std::vector<int>& c = cls[ci];
                if (c[0] == fl) std::swap(c[0], c[1]);
                if (lit_val(c[0]) == 1) { ws[keep++] = ci; continue; }
                bool moved = false;
                for (size_t k = 2; k < c.size(); ++k)
                    if (lit_val(c[k]) != 0) {
                        std::swap(c[1], c[k]);
                        watch[c[1]].push_back(ci);
                        moved = true;
                        break;
                    }
                if (moved) continue;
                ws[keep++] = ci;
                if (lit_val(c[0]) == 0) {   // conflict: keep the remaining watches, stop
                    for (size_t r = i + 1; r < ws.size(); ++r) ws[keep++] = ws[r];
                    ws.resize(keep);
                    qhead = trail.size();
                    return ci;
                }
                enqueue(c[0], ci);
            }
            ws.resize(keep);
        }
        return -1;
    }

    void bump(int v) {
        if ((act[v] += inc) > 1e100) {
            for (double& a : act) a *= 1e-100;
            inc *= 1e-100;
        }
    }

    void backtrack(int lvl) {
        if ((int)lim.size() <= lvl) return;
        for (size_t i = trail.size(); i-- > (size_t)lim[lvl];) {
            const int v = trail[i] >> 1;
            phase[v] = val[v];
            val[v] = -1;
        }
        trail.resize(lim[lvl]);
        lim.resize(lvl);
        qhead = trail.size();
    }

    // first-UIP learning; returns the backtrack level, learnt[0] is the asserting literal
    int analyze(int confl, std::vector<int>& learnt) {
        learnt.assign(1, 0);
        int open = 0, p = -1;
        size_t idx = trail.size();
        const int cur = (int)lim.size();
        do {
            const std::vector<int>& c = cls[confl];
            for (size_t k = (p < 0 ? 0 : 1); k < c.size(); ++k) {
                const int v = c[k] >> 1;
                if (!seen[v] && level[v] > 0) {
                    seen[v] = 1;
                    bump(v);
                    if (level[v] >= cur) ++open; else learnt.push_back(c[k]);
                }
            }
            while (!seen[trail[--idx] >> 1]) {}
            p = trail[idx];
            confl = reason[p >> 1];
            seen[p >> 1] = 0;
            --open;
            if (confl >= 0 && open > 0) {   // the reason clause must have p in front
                std::vector<int>& r = cls[confl];
                if (r[0] != p) for (size_t k = 1; k < r.size(); ++k) if (r[k] == p) { std::swap(r[0], r[k]); break; }
            }
        } while (open > 0);
        learnt[0] = p ^ 1;
        int bt = 0;
        size_t at = 1;
        for (size_t k = 1; k < learnt.size(); ++k) {
            seen[learnt[k] >> 1] = 0;
            if (level[learnt[k] >> 1] > bt) { bt = level[learnt[k] >> 1]; at = k; }
        }
        if (learnt.size() > 1) std::swap(learnt[1], learnt[at]);
        return bt;
    }

    // 1 = SAT, 0 = UNSAT, -1 = conflict limit reached
    int solve(int64_t conflict_limit) {
        if (!ok) return 0;
        if (propagate() >= 0) { ok = false; return 0; }
        int64_t restart_at = conflicts + 100;   // `conflicts` accumulates over incremental calls
        std::vector<int> learnt;
        for (;;) {
            const int confl = propagate();
            if (confl >= 0) {
                ++conflicts;
                if (lim.empty()) { ok = false; return 0; }
                const int bt = analyze(confl, learnt);
                backtrack(bt);
                if (learnt.size() == 1) {
                    enqueue(learnt[0], -1);
                } else {
                    cls.push_back(learnt);
                    attach((int)cls.size() - 1);
                    enqueue(learnt[0], (int)cls.size() - 1);
                }
                inc *= 1.0 / 0.95;
                if (conflict_limit > 0 && conflicts >= conflict_limit) { backtrack(0); return -1; }
                if (conflicts >= restart_at) { restart_at += restart_at / 2 + 100; backtrack(0); }
                continue;
            }
            int best = -1;
            for (int v = 0; v < n; ++v)
                if (val[v] < 0 && (best < 0 || act[v] > act[best])) best = v;
            if (best < 0) return 1;
            lim.push_back((int)trail.size());
            enqueue(2 * best + (phase[best] ? 0 : 1), -1);
        }
    }
};

}  // namespace

extern "C" int dsat_host_solve(int32_t n_vars, int32_t n_clauses, const int32_t* clause_rowptr,
                               const int32_t* clause_lits, int64_t conflict_limit, int32_t* result_out,
                               uint8_t* model_out) {
    using namespace dsat;
    DSAT_REQUIRE(result_out != nullptr, DSAT_E_INVALID, "dsat_host_solve: result_out is required");
    DSAT_REQUIRE(n_vars >= 0 && n_clauses >= 0, DSAT_E_INVALID, "dsat_host_solve: negative size");
    DSAT_REQUIRE(n_clauses == 0 || (clause_rowptr != nullptr && clause_lits != nullptr), DSAT_E_INVALID,
                 "dsat_host_solve: null clause arrays");
    Cdcl s(n_vars);
    for (int c = 0; c < n_clauses; ++c) {
        const int e0 = clause_rowptr[c], e1 = clause_rowptr[c + 1];
        DSAT_REQUIRE(e0 >= 0 && e1 >= e0, DSAT_E_INVALID, "dsat_host_solve: clause_rowptr must be non-decreasing");
        std::vector<int> lits;
        for (int e = e0; e < e1; ++e) {
            const int l = clause_lits[e];
            DSAT_REQUIRE(l != 0 && l >= -n_vars && l <= n_vars, DSAT_E_INVALID,
                         "dsat_host_solve: literal out of range");
            lits.push_back(l > 0 ? 2 * (l - 1) : 2 * (-l - 1) + 1);
        }
        if (lits.empty()) { s.ok = false; break; }   // the empty clause
        if (!s.add(lits)) break;
    }
    const int r = s.solve(conflict_limit);
    *result_out = r;
    if (r == 1 && model_out != nullptr)
        for (int v = 0; v < n_vars; ++v) model_out[v] = s.val[v] == 1 ? 1 : 0;
    return DSAT_OK;
}

// ---------------------------------------------------------------------------------------------
// NeuroSAT-style SR(n) instance generator (data/k_sat.py:36-93), natively: the reference adds
// random clauses of width k = (1 w.p. p_k_2 else 2) + Geometric(p_geo) to an incremental PySAT
// solver until it reports UNSAT, negates the first literal of the last clause (which makes the
// formula satisfiable again), removes duplicate and subsumed clauses (utils/DimacsFile.py:110-128)
// and, for a desired multiplier of the number of solutions > 1, drops round(log_x(multiplier))
// random clauses with x = 2^(n/m) (:66-84).  A 100 000-instance stream (BASELINE config 4) needs
// this at native speed; the random stream is this file's own (splitmix64), not Python's.
namespace {

struct SplitMix {
    uint64_t s;
    explicit SplitMix(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   // [0,1)
    int below(int n) { return (int)(uniform() * n); }
    int geometric(double p) {   // trials up to and including the first success, support {1,2,...}
        const double u = uniform();
        const double k = std::ceil(std::log1p(-u) / std::log1p(-p));
        return k < 1.0 ? 1 : (k > 1e6 ? 1000000 : (int)k);
    }
};

// Dedup + subsumption with the order of diffusionsat_amd/dimacs.py::reduce_clauses: literals of a
// clause ascending (signed), clauses by (length, lexicographic); a clause that contains an earlier
// kept clause is dropped.
void reduce_clauses(std::vector<std::vector<int>>& cls) {
    for (auto& c : cls) {
        std::sort(c.begin(), c.end());
        c.erase(std::unique(c.begin(), c.end()), c.end());   // reduce_clauses goes through set(tuple(sorted))
    }
    std::sort(cls.begin(), cls.end(), [](const std::vector<int>& a, const std::vector<int>& b) {
        if (a.size() != b.size()) return a.size() < b.size();
        return a < b;
    });
    cls.erase(std::unique(cls.begin(), cls.end()), cls.end());
    std::vector<std::vector<int>> kept;
    for (const auto& c : cls) {
        bool sub = false;
        for (const auto& k : kept) {
            if (k.size() > c.size()) break;   // kept is ordered by length
            if (std::includes(c.begin(), c.end(), k.begin(), k.end())) { sub = true; break; }
        }
        if (!sub) kept.push_back(c);
    }
    cls.swap(kept);
}

}  // namespace

extern "C" int dsat_host_gen_sr(uint64_t seed, int32_t min_vars, int32_t max_vars, double p_k_2, double p_geo,
                                double solutions_multiplier, int32_t* n_vars_out, int32_t* n_clauses_out,
                                int32_t* clause_rowptr_out, int32_t rowptr_cap, int32_t* clause_lits_out,
                                int32_t lits_cap) {
    using namespace dsat;
    DSAT_REQUIRE(n_vars_out && n_clauses_out && clause_rowptr_out && clause_lits_out, DSAT_E_INVALID,
                 "dsat_host_gen_sr: null output");
    DSAT_REQUIRE(min_vars >= 1 && max_vars >= min_vars && max_vars <= 100000, DSAT_E_INVALID,
                 "dsat_host_gen_sr: need 1 <= min_vars <= max_vars");
    DSAT_REQUIRE(p_k_2 >= 0.0 && p_k_2 <= 1.0 && p_geo > 0.0 && p_geo <= 1.0, DSAT_E_INVALID,
                 "dsat_host_gen_sr: probabilities out of range");
    SplitMix rng(seed * 0x2545F4914F6CDD1Dull + 0x1234567ull);
    const int n = min_vars + rng.below(max_vars - min_vars + 1);       // random.randint(min, max)
    Cdcl solver(n);
    std::vector<std::vector<int>> clauses;   // signed DIMACS literals
    std::vector<int> perm(n);
    std::vector<int> last;
    for (;;) {
        const int k_base = rng.uniform() < p_k_2 ? 1 : 2;
        const int k = std::min(n, k_base + (p_geo >= 1.0 ? 1 : rng.geometric(p_geo)));
        for (int i = 0; i < n; ++i) perm[i] = i;
        std::vector<int> c(k), enc(k);
        for (int i = 0; i < k; ++i) {   // np.random.choice(n, k, replace=False): partial Fisher-Yates
            const int j = i + rng.below(n - i);
            std::swap(perm[i], perm[j]);
            const int v = perm[i];
            const bool pos = rng.uniform() < 0.5;
            c[i] = pos ? v + 1 : -(v + 1);
            enc[i] = pos ? 2 * v : 2 * v + 1;
        }
        solver.backtrack(0);
        const bool alive = solver.add(enc);
        const int r = alive ? solver.solve(0) : 0;
        if (r == 1) {
            clauses.push_back(c);
        } else {
            last = c;
            break;
        }
    }
    last[0] = -last[0];          // the SAT member of NeuroSAT's SR(n) pair (k_sat.py:57-64)
    clauses.push_back(last);
    reduce_clauses(clauses);
    if (solutions_multiplier > 1.0 && !clauses.empty()) {
        const int m = (int)clauses.size();
        const double x = std::pow(2.0, (double)n / m);
        double d = std::nearbyint(std::log(solutions_multiplier) / std::log(x));   // Python round(): half to even
        d = std::max(0.0, std::min(d, (double)(m - 1)));
        std::vector<int> idx(m);
        for (int i = 0; i < m; ++i) idx[i] = i;
        std::vector<char> drop(m, 0);
        for (int i = 0; i < (int)d; ++i) {   // random.sample(range(m), d)
            const int j = i + rng.below(m - i);
            std::swap(idx[i], idx[j]);
            drop[idx[i]] = 1;
        }
        std::vector<std::vector<int>> kept;
        for (int i = 0; i < m; ++i)
            if (!drop[i]) kept.push_back(clauses[i]);
        clauses.swap(kept);
    }
    int64_t total = 0;
    for (const auto& c : clauses) total += (int64_t)c.size();
    *n_vars_out = n;
    *n_clauses_out = (int32_t)clauses.size();
    DSAT_REQUIRE((int64_t)clauses.size() + 1 <= rowptr_cap && total <= lits_cap, DSAT_E_INVALID,
                 "dsat_host_gen_sr: output buffers too small (need " + std::to_string(clauses.size() + 1) +
                     " row pointers and " + std::to_string(total) + " literals)");
    clause_rowptr_out[0] = 0;
    int32_t e = 0;
    for (size_t ci = 0; ci < clauses.size(); ++ci) {
        for (int l : clauses[ci]) clause_lits_out[e++] = l;
        clause_rowptr_out[ci + 1] = e;
    }
    return DSAT_OK;
}
