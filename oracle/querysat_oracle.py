"""NumPy/SciPy restatement of the reference's denoising-sampling path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED for the
model path (no golden vectors exist in the reference; TensorFlow is absent).

Every function cites the reference lines it follows (paths relative to
``/root/reference``).  All randomness the reference draws from TF's RNG is an
explicit argument here:

  * ``normal[R, N, 4]``  -- ``tf.random.normal([n_vars, 4])`` per round
    (model/query_sat.py:239)
  * ``labels[N]``        -- ``tf.random.uniform([n_vars], 0, 2, int32)``
    (model/query_sat.py:145)
  * ``uniform[N, 1]``    -- ``tf.random.uniform`` in ``randomized_rounding_tf``
    (model/query_sat.py:55-60)

The sparse products mirror ``tf.sparse.sparse_dense_matmul`` on COO tensors
built exactly as ``data/SatSpecifics.py:21-69`` builds them (duplicate
coordinates add up, as they do in TF).  ``dt`` selects float32 (the reference's
dtype) or float64 (a higher-precision cross-check).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.special import expit

T_POWER = 0.5  # model/query_sat.py:13

# MLP constructor arithmetic, model/query_sat.py:101-122 (trial=None defaults)
FEATURE_MAPS = 128
QUERY_MAPS = 128
LOGIT_MAPS = 8


def mlp_dims(feature_maps=FEATURE_MAPS, query_maps=QUERY_MAPS, logit_maps=LOGIT_MAPS,
             msg_layers=3):
    """Layer widths of the five MLPs, in -> ... -> out.

    model/query_sat.py:101-122 with trial=None: update_layers=3, output_layers=2,
    query_layers=2, clauses_layers=2, scales 1.8 / 1 / 1.6 / 1.2.  Input widths
    follow from the concats at :239 (v1), :258 (clause_unit), :277 (unit).
    """
    v1 = feature_maps + 4 + 5  # variables, normal[.,4], noisy_labels[.,5]
    upd_h = int(feature_maps * 1.8)
    out_h = int(feature_maps * 1)
    cl_h = int(feature_maps * 1.6)
    q_h = int(query_maps * 1.2)
    return {
        "variables_query": [v1] + [q_h] * (2 - 1) + [query_maps],
        "lit_mlp": [v1] + [query_maps * 4] * (msg_layers - 1) + [query_maps * 2],
        "clause_mlp": [feature_maps + 2 * query_maps] + [cl_h] * (2 - 1) + [feature_maps + query_maps],
        "update_gate": [query_maps + v1 + 2 * query_maps] + [upd_h] * (3 - 1) + [feature_maps],
        "variables_output": [feature_maps] + [out_h] * (2 - 1) + [logit_maps],
    }


MLP_ORDER = ("variables_query", "lit_mlp", "clause_mlp", "update_gate", "variables_output")


def make_weights(seed=0, dt=np.float32, bias_scale=0.0, dims=None):
    """Glorot-uniform kernels (Keras ``Dense`` default) and zero / small biases.

    model/mlp.py:24,37-39: ``Dense(...)`` with default ``glorot_uniform`` kernel
    and ``zeros`` bias.  ``bias_scale>0`` draws non-zero biases so tests also
    exercise the bias path a trained checkpoint would.
    Returns {mlp_name: [(kernel[in,out], bias[out]), ...]}.
    """
    rng = np.random.default_rng(seed)
    dims = dims or mlp_dims()
    out = {}
    for name in MLP_ORDER:
        d = dims[name]
        layers = []
        for fi, fo in zip(d[:-1], d[1:]):
            lim = np.sqrt(6.0 / (fi + fo))
            w = rng.uniform(-lim, lim, size=(fi, fo)).astype(dt)
            b = (rng.standard_normal(fo) * bias_scale).astype(dt)
            layers.append((w, b))
        out[name] = layers
    return out


def cast_weights(weights, dt):
    return {k: [(w.astype(dt), b.astype(dt)) for (w, b) in v] for k, v in weights.items()}


# --------------------------------------------------------------------------- graph

def compute_adj_indices(clauses):
    """data/dimac.py:14-18."""
    pos = [[v - 1, idx] for idx, c in enumerate(clauses) for v in c if v > 0]
    neg = [[abs(v) - 1, idx] for idx, c in enumerate(clauses) for v in c if v < 0]
    return pos, neg


def shift_clause(clauses, offset):
    """data/dimac.py:166-171."""
    return [[(x + offset) if x > 0 else (x - offset) for x in c] for c in clauses]


def create_adj_matrices(instances, dt=np.float32):
    """Batch ``instances`` = [(n_vars, clauses), ...] into the reference's tensors.

    data/dimac.py:213-260 (``prepare_example``: variable offsets, batched clauses)
    and data/SatSpecifics.py:21-69 (``create_adj_matrices``).  Returns scipy CSR
    matrices: ``adj`` [2N x C] (rows 0..N-1 positive, N..2N-1 negative literals),
    ``clauses_graph`` [G x C], ``variables_graph`` [G x N], all values 1.
    """
    batched = []
    variable_count, clauses_in_formula = [], []
    offset = 0
    for n_vars, clauses in instances:
        batched.extend(shift_clause(clauses, offset))
        variable_count.append(int(n_vars))
        clauses_in_formula.append(len(clauses))
        offset += int(n_vars)
    n = int(sum(variable_count))
    c = int(sum(clauses_in_formula))
    g = len(instances)
    pos, neg = compute_adj_indices(batched)
    pos = np.asarray(pos, dtype=np.int64).reshape(-1, 2)
    neg = np.asarray(neg, dtype=np.int64).reshape(-1, 2)
    neg = neg + np.array([[n, 0]], dtype=np.int64)  # SatSpecifics.py:25-30
    idx = np.concatenate([pos, neg], axis=0)
    adj = sp.coo_matrix((np.ones(len(idx), dtype=dt), (idx[:, 0], idx[:, 1])), shape=(2 * n, c)).tocsr()
    vmask = np.repeat(np.arange(g), variable_count)
    cmask = np.repeat(np.arange(g), clauses_in_formula)
    cg = sp.coo_matrix((np.ones(c, dtype=dt), (cmask, np.arange(c))), shape=(g, c)).tocsr()
    vg = sp.coo_matrix((np.ones(n, dtype=dt), (vmask, np.arange(n))), shape=(g, n)).tocsr()
    return {
        "adjacency_matrix": adj,
        "clauses_graph_adj": cg,
        "variables_graph_adj": vg,
        "variables_in_graph": np.asarray(variable_count),
        "clauses_in_formula": np.asarray(clauses_in_formula),
        "normal_clauses": [cl for _, cl in instances],
        "clauses": batched,
        "variables_mask": vmask,
        "clauses_mask": cmask,
    }


# --------------------------------------------------------------------------- elementwise

def softplus(x):
    """``tf.nn.softplus`` = log(1 + exp(x)), evaluated stably."""
    return np.logaddexp(x.dtype.type(0), x)


def sigmoid(x):
    return expit(x)


def leaky_relu(x, alpha=0.2):
    """``tf.nn.leaky_relu`` default alpha=0.2 (model/mlp.py:14)."""
    return np.where(x > 0, x, x * x.dtype.type(alpha))


def mlp(x, layers):
    """model/mlp.py:22-25,37-50: hidden ``Dense(activation=leaky_relu)``, last linear."""
    cur = x
    last = len(layers) - 1
    for i, (w, b) in enumerate(layers):
        cur = cur @ w + b
        if i < last:
            cur = leaky_relu(cur)
    return cur


def pair_norm(x, graph_norm, mask, eps=1e-6):
    """layers/normalization.py:43-71 with subtract_mean=True.

    mean = graph_norm @ x  ([G,F]);  x -= mean[mask];  per-ROW variance over the
    feature axis (the reference's stated deviation from PairNorm, :63-69).
    """
    dt = x.dtype.type
    mean = graph_norm @ x
    x = x - mean[mask]
    variance = np.mean(np.square(x), axis=1, keepdims=True, dtype=x.dtype)
    return x * (dt(1) / np.sqrt(variance + dt(eps)))


def softplus_loss_adj(q, cl_adj):
    """loss/sat.py:125-137: exp(-(A^T @ softplus([q; -q])))."""
    literals = softplus(np.concatenate([q, -q], axis=0))
    return np.exp(-(cl_adj @ literals))


def softplus_mixed_loss_adj(variable_predictions, cl_adj, eps=1e-8):
    """loss/sat.py:92-100: clauses_val * -(log(1 - clauses_val + eps) - log(1 + eps)) per clause and logit map;
    every operation in the dtype of the predictions, as the reference's float32 graph does (1 + 1e-8 is 1 there)."""
    dt = variable_predictions.dtype.type
    val = softplus_loss_adj(variable_predictions, cl_adj)
    log_clauses = -(np.log((dt(1) - val) + dt(eps)) - np.log(dt(1) + dt(eps)))
    return val * log_clauses


def is_batch_sat(predictions, cl_adj):
    """utils/sat.py:118-124.  np.round is half-to-even like tf.round."""
    dt = predictions.dtype.type
    variables = np.round(sigmoid(predictions))
    literals = np.concatenate([variables, dt(1) - variables], axis=0)
    clauses_sat = np.clip(cl_adj @ literals, 0, 1)
    return clauses_sat.min() if clauses_sat.size else dt(1)


def is_graph_sat(predictions, adj, clauses_graph):
    """utils/sat.py:165-180: per graph, 1 when every clause is satisfied by round(sigmoid(predictions))."""
    dt = predictions.dtype.type
    variables = np.round(sigmoid(predictions)).reshape(-1, 1)
    literals = np.concatenate([variables, dt(1) - variables], axis=0)
    clauses_sat = np.clip(adj.T @ literals, 0, 1)
    in_g = clauses_graph @ clauses_sat
    total = np.asarray(clauses_graph.sum(axis=1)).reshape(-1, 1)
    return np.clip(in_g + 1 - total, 0, 1)[:, 0]


def distribution_at_time(x, t):
    """model/query_sat.py:66-68."""
    return x * (1 - t) + t / 2


def bernoulli_kl(a, b):
    """KL(Bernoulli(a) || Bernoulli(b)) for probabilities in (0, 1).

    Third-party: tensorflow_probability~=0.12.1 ``Bernoulli.kl_divergence``
    (not vendored in the reference; call site model/query_sat.py:45-50).  The
    published definition is a*(log a - log b) + (1-a)*(log(1-a) - log(1-b)).
    PARITY UNPINNED: no reference test pins this.
    """
    return a * (np.log(a) - np.log(b)) + (1 - a) * (np.log1p(-a) - np.log1p(-b))


def train_loss(labels, logits, t, label_smoothing=0.01):
    """model/query_sat.py:40-53."""
    dt = logits.dtype.type
    t = dt(np.power(dt(t), dt(T_POWER)))
    tl = dt(min(t + dt(label_smoothing), dt(1)))
    labels_at_t = distribution_at_time(labels, tl)
    probs_at_t = distribution_at_time(sigmoid(logits), t)
    loss = bernoulli_kl(labels_at_t, probs_at_t)
    n1 = dt(distribution_at_time(dt(0.0), tl))
    n2 = dt(distribution_at_time(dt(0.0), dt(1.0)))
    norm = bernoulli_kl(n1, n2)
    return loss / (norm + dt(1e-4))


# --------------------------------------------------------------------------- model

def query_sat_loop(weights, data, noise_scale, noisy_num, normal, labels, rounds,
                   early_exit=True, dt=np.float32, denoised_num=None, trace=False, supervised=True):
    """``QuerySAT.call`` + ``QuerySAT.loop`` at inference.

    model/query_sat.py:133-184 (state init, labels) and :186-373 (prologue and
    rounds).  ``normal`` is [rounds, N, 4]; ``labels`` is int [N] in {0,1}.
    Returns dict: ``prediction`` [N] (:371-373, :480), ``steps_taken`` (last
    executed round index, the value of the autograph loop variable),
    ``logits`` [N,8] of the last executed round, ``best_map`` [G]; with
    ``trace`` also the per-round lists used by the parity tests.
    """
    f = dt
    adj = data["adjacency_matrix"].astype(dt)           # [2N x C]
    cl_adj = adj.T.tocsr()                               # :188
    vg = data["variables_graph_adj"].astype(dt)
    cg = data["clauses_graph_adj"].astype(dt)
    n2, n_clauses = adj.shape
    n_vars = n2 // 2
    n_graphs = vg.shape[0]
    vmask = np.asarray(data["variables_mask"])
    cmask = np.asarray(data["clauses_mask"])
    w = weights

    # feature_maps / query_maps as the weights have them (model/query_sat.py:86-89; 128 / 128 in the released model)
    feature_maps = w["variables_output"][0][0].shape[0]
    query_maps = w["variables_query"][-1][0].shape[1]
    clause_state = np.ones((n_clauses, feature_maps), dtype=dt)   # :141
    variables = np.ones((n_vars, feature_maps), dtype=dt)          # :148

    lit_degree = np.asarray(adj.sum(axis=1), dtype=dt).reshape(n2, 1)           # :193
    degree_weight = f(1) / np.sqrt(np.maximum(lit_degree, f(1)))                 # :194
    var_degree_weight = f(4) / np.sqrt(np.maximum(lit_degree[:n_vars] + lit_degree[n_vars:], f(1)))  # :195
    rev_lit_degree = np.asarray(cl_adj.sum(axis=1), dtype=dt).reshape(n_clauses, 1)  # :196
    rev_degree_weight = f(1) / np.sqrt(np.maximum(rev_lit_degree, f(1)))             # :197

    vg_norm = sp.diags((f(1) / np.asarray(vg.sum(axis=1), dtype=dt).ravel()).astype(dt)) @ vg   # :206
    cg_norm = sp.diags((f(1) / np.asarray(cg.sum(axis=1), dtype=dt).ravel()).astype(dt)) @ cg   # :211
    vg_norm = vg_norm.tocsr().astype(dt)
    cg_norm = cg_norm.tocsr().astype(dt)

    noisy_labels = np.concatenate(
        [np.asarray(noisy_num, dtype=dt), np.zeros((n_vars, 1), dtype=dt) + f(noise_scale)], axis=-1)  # :216, :70-74
    if denoised_num is None:
        den = np.zeros((n_vars, 2), dtype=dt)                        # :217
    else:
        den = np.concatenate([denoised_num, 1 - denoised_num], axis=-1).astype(dt)  # :218
    noisy_labels = np.concatenate([noisy_labels, den], axis=-1)     # :219  -> [N,5]

    labels_f = np.asarray(labels).astype(dt).reshape(n_vars, 1)
    smoothed = np.tile(labels_f, [1, LOGIT_MAPS])                   # :289-290

    last_logits = np.zeros((n_vars, LOGIT_MAPS), dtype=dt)          # :192
    best_map = np.zeros(n_graphs, dtype=np.int64)
    step = -1
    tr = {k: [] for k in ("logits", "query", "clause_state", "variables", "out_logits",
                          "best_map", "per_graph_loss", "variables_grad", "is_sat",
                          "literals", "clauses_loss4", "clause_messages", "variables_loss_all",
                          "new_clause_value", "variables_loss", "new_variables")}
    keep = set(tr) if trace is True else (set(trace) if trace else set())   # trace=("logits", ...) keeps a subset

    def rec(key, value):
        if key in keep:
            tr[key].append(np.array(value, copy=True))
    for step in range(rounds):                                       # :225
        v1 = np.concatenate([variables, np.asarray(normal[step], dtype=dt), noisy_labels], axis=-1)  # :239
        query = mlp(v1, w["variables_query"])                        # :240
        clauses_loss = softplus_loss_adj(query, cl_adj)              # :241
        # :242-246, closed form of d(sum clauses_loss)/d query
        m = adj @ clauses_loss                                       # [2N, Q]
        variables_grad = (-sigmoid(query) * m[:n_vars] + sigmoid(-query) * m[n_vars:]) * var_degree_weight
        clauses_loss = clauses_loss * f(4)                           # :248

        var_msg = mlp(v1, w["lit_mlp"])                              # :252
        lit1, lit2 = var_msg[:, :query_maps], var_msg[:, query_maps:]  # :253
        literals = np.concatenate([lit1, lit2], axis=0)             # :254
        clause_messages = (cl_adj @ literals) * rev_degree_weight   # :255-256
        clause_unit = np.concatenate([clause_state, clause_messages, clauses_loss], axis=-1)  # :258
        clause_data = mlp(clause_unit, w["clause_mlp"])             # :261

        variables_loss_all = clause_data[:, :query_maps]            # :263
        new_clause_value = clause_data[:, query_maps:]              # :264
        if trace:   # pre-normalisation values (the tensors the fused kernels' debug taps expose)
            rec("literals", literals); rec("clauses_loss4", clauses_loss)
            rec("clause_messages", clause_messages)
            rec("variables_loss_all", variables_loss_all)
            rec("new_clause_value", new_clause_value)
        new_clause_value = pair_norm(new_clause_value, cg_norm, cmask) * f(0.25)  # :265
        clause_state = new_clause_value + f(0.1) * clause_state     # :266

        variables_loss = (adj @ variables_loss_all) * degree_weight  # :269-270
        vl_pos, vl_neg = variables_loss[:n_vars], variables_loss[n_vars:]  # :273

        unit = np.concatenate([variables_grad, v1, vl_pos, vl_neg], axis=-1)  # :277
        new_variables = mlp(unit, w["update_gate"])                 # :278
        if trace:
            rec("variables_loss", variables_loss); rec("new_variables", new_variables)
        new_variables = pair_norm(new_variables, vg_norm, vmask) * f(0.25)  # :279
        variables = new_variables + f(0.1) * variables              # :280

        logits = mlp(variables, w["variables_output"])              # :283
        if supervised:
            per_var_loss = train_loss(smoothed, logits, noise_scale)    # :291
            per_graph_loss = vg_norm @ per_var_loss                     # :292   [G, 8]
        else:   # QuerySAT(supervised=False), use_linear_loss = False (:97): :306-309
            per_clause_loss = softplus_mixed_loss_adj(logits, cl_adj)   # [C, 8]
            per_graph_loss = cg @ per_clause_loss                       # the un-normalised clauses_graph: a sum
            per_graph_loss = np.sqrt(per_graph_loss + f(1e-6)) - np.sqrt(f(1e-6))
        best_map = np.argmin(per_graph_loss, axis=-1)               # :317 (first minimum on ties)
        best_per_var = best_map[vmask]                               # :318-320
        out_logits = logits[np.arange(n_vars), best_per_var].reshape(n_vars, 1)  # :328-329
        is_sat = is_batch_sat(out_logits, cl_adj)                   # :330
        last_logits = logits                                         # :337 / :339
        if trace:
            rec("logits", logits); rec("query", query)
            rec("clause_state", clause_state); rec("variables", variables)
            rec("out_logits", out_logits[:, 0]); rec("best_map", best_map)
            rec("per_graph_loss", per_graph_loss); rec("variables_grad", variables_grad)
            if "is_sat" in keep:
                tr["is_sat"].append(float(is_sat))
        if early_exit and is_sat == 1:                               # :331-338
            break
        # :347-348  x*0.2 + x*0.8 (forward value only)
        variables = variables * f(0.2) + variables * f(0.8)
        clause_state = clause_state * f(0.2) + clause_state * f(0.8)

    best_per_var = best_map[vmask] if n_vars else np.zeros(0, dtype=np.int64)
    out_logits = last_logits[np.arange(n_vars), best_per_var]       # :371
    res = {"prediction": out_logits, "steps_taken": step, "logits": last_logits, "best_map": best_map}
    if trace:
        res["trace"] = tr
    return res


def diffusion_step(weights, data, noise_scale, noisy_num, normal, labels, rounds=32,
                   early_exit=True, dt=np.float32, trace=False, denoised_num=None, supervised=True):
    """model/query_sat.py:467-481 (``QuerySAT.diffusion_step``): labels=None at the
    call site means random labels (:145); they are the explicit ``labels`` here."""
    # denoised_num [N,1]: QuerySAT.diffusion_step_self (model/query_sat.py:483-498)
    return query_sat_loop(weights, data, noise_scale, noisy_num, normal, labels, rounds,
                          early_exit=early_exit, dt=dt, trace=trace, denoised_num=denoised_num, supervised=supervised)


# --------------------------------------------------------------------------- sampler

def randomized_rounding(x, uniform):
    """model/query_sat.py:55-60 with the uniform noise [N,1] injected."""
    dt = x.dtype.type
    x0 = x[..., 0:1]
    rounded = np.floor(x0 + uniform.astype(x.dtype))
    return np.concatenate([rounded, dt(1) - rounded], axis=-1)


def reverse_distribution_step_theoretic(x, x0, t, t_increment):
    """satuniformity/DiffusionSampler.py:29-37."""
    dt = x.dtype.type
    t1 = dt(np.power(dt(t), dt(T_POWER)))
    t2 = dt(np.power(dt(max(0.0, t - t_increment)), dt(T_POWER)))
    x_new = distribution_at_time(x0, t1)
    alpha_t = dt((dt(1) - t1) / (dt(1) - t2))
    x_unnormed = distribution_at_time(x, dt(1) - alpha_t) * x_new
    return x_unnormed / (np.sum(x_unnormed, axis=-1, keepdims=True) + dt(1e-8))


def sat_accuracy(prediction_logits, data):
    """metrics/sat_metrics.py:60-85 (``SATAccuracyTF.accuracy``), ``total_acc`` only:
    per graph, are all its clauses satisfied by round(sigmoid(logits))."""
    adj = data["adjacency_matrix"]
    pred = np.round(sigmoid(prediction_logits)).astype(np.int64)
    literals = np.concatenate([pred, 1 - pred], axis=0).reshape(-1, 1)
    adj_i = adj.astype(np.int64)
    clauses = np.clip(adj_i.T @ literals, 0, 1)
    cg = data["clauses_graph_adj"].astype(np.int64)
    sat_per_graph = np.asarray(cg @ clauses).ravel()
    return sat_per_graph == np.asarray(cg.sum(axis=1)).ravel()


def satisfiable(clauses, bits):
    """utils/VariableAssignment.py:79-90."""
    for c in clauses:
        ok = False
        for lit in c:
            if (lit > 0) == bool(bits[abs(lit) - 1]):
                ok = True
                break
        if not ok:
            return False
    return True


def assignment_to_int(bits):
    """utils/VariableAssignment.py:63-69 (bit i <-> x_{i+1})."""
    res = 0
    for i, b in enumerate(bits):
        if int(b) == 1:
            res |= 1 << i
    return res


def diffusion(weights, data, n_steps, normal, labels, uniform, rounds=32, early_exit=True,
              dt=np.float32, trace=False, self_supervised=False):
    """satuniformity/DiffusionSampler.py:78-191 (baseline sampling, not self-supervised).

    ``normal`` [T, R, N, 4], ``labels`` [T, N], ``uniform`` [T, N, 1].
    Returns (mean(cum_accuracy), predictions[N] in {0,1}, per-step trace).
    """
    f = dt
    n_vars = int(np.sum(data["variables_in_graph"]))
    n_graphs = len(data["variables_in_graph"])
    x = np.zeros((n_vars, 2), dtype=dt) + f(0.5)                 # :86
    fixed_step = np.full(n_vars, -1, dtype=np.int64)            # :87
    x_fixed = np.zeros(n_vars, dtype=dt)                         # :88
    cum_accuracy = np.zeros(n_graphs)                            # :89
    predictions = None
    steps = []
    for t in range(n_steps):                                     # :105
        noise_scale = f(1 - t / n_steps)                         # :106
        x_noisy = randomized_rounding(x, np.asarray(uniform[t]))  # :107
        x = x_noisy                                              # :108-109 (use_baseline_sampling)
        den = predictions[:, None] if (self_supervised and predictions is not None) else None   # :110-113, :58
        out = diffusion_step(weights, data, noise_scale, x_noisy, normal[t], labels[t], rounds,
                             early_exit=early_exit, dt=dt, denoised_num=den)   # :115-117 -> :40-63
        logits = out["prediction"]
        predictions = sigmoid(logits)                            # :61
        total_accuracy = sat_accuracy(logits, data)              # :119-124
        x = reverse_distribution_step_theoretic(
            x, np.stack([f(1) - predictions, predictions], axis=1), noise_scale, 1 / n_steps)  # :127-129
        cum_accuracy = np.maximum(cum_accuracy, total_accuracy)  # :130
        xx = np.round(predictions)                               # :154
        shift = 0
        for cur_clauses, cur_n in zip(data["normal_clauses"], data["variables_in_graph"]):  # :157
            cur_n = int(cur_n)
            if fixed_step[shift] >= 0:                           # :158-161
                shift += cur_n
                continue
            vals = xx[shift:shift + cur_n]
            if satisfiable(cur_clauses, [bool(b) for b in vals]):  # :162-169
                x_fixed[shift:shift + cur_n] = vals
                fixed_step[shift:shift + cur_n] = t
            shift += cur_n
        if trace:
            steps.append({"x_noisy": x_noisy.copy(), "logits": logits.copy(), "x": x.copy(),
                          "total_accuracy": total_accuracy.copy(), "steps_taken": out["steps_taken"]})
    final = np.round(predictions)                                # :181
    final = np.where(fixed_step >= 0, x_fixed, final)            # :182-184
    return float(np.mean(cum_accuracy)), final, steps


def samples_from_predictions(predictions, n_graphs, n_vars, clauses, counts=None):
    """Body of the per-graph loop of ``DiffusionSampler.samples``
    (satuniformity/DiffusionSampler.py:283-307): int-encode each graph's bits,
    keep the satisfying ones.  Returns (dict int->count, n_sat)."""
    counts = {} if counts is None else counts
    n_sat = 0
    for i in range(n_graphs):
        bits = [int(b) for b in predictions[i * n_vars:(i + 1) * n_vars]]
        if satisfiable(clauses, bits):
            key = assignment_to_int(bits)
            counts[key] = counts.get(key, 0) + 1
            n_sat += 1
    return counts, n_sat
