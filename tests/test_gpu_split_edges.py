"""The split-product chain (split_bf16.h: fp32 products from three bf16 slices per operand) as a single operator,
``dsat_op_mlp_ex(impl=2)`` -- the same ``split_chain`` the round kernels' P3 and role-B tiles run -- on ordinary
inputs and at the ends of the fp32 range: values next to FLT_MAX, denormals, infinities and NaN, in the inputs and
in the weights.  The class of every result must be that of ``x @ W`` on the fp32 MFMA (``impl=1``): finite stays
finite to fp32 rounding, non-finite inputs poison their own row and nothing else.

One band is different, and pinned here: an ACTIVATION with |x| above bf16's largest finite value
(3.3895e38 .. FLT_MAX = 3.4028e38, the top 0.4 % of the last binade) rounds to infinity when it is sliced, so its
row comes out non-finite where the fp32 MFMA would still return a finite number.  Weights have no such band: they
are sliced on the host at load time, exactly, up to FLT_MAX (dsat_api.hip, ``bf16_rne_host``).
"""
import numpy as np
import pytest

from helpers import make_all, oracle_weights
from oracle import querysat_oracle as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

FLT_MAX = np.float32(np.finfo(np.float32).max)
BF16_MAX = np.float32(3.3895313892515355e38)        # 0x7F7F0000


@pytest.fixture(scope="module")
def rt():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    from diffusionsat_amd import runtime
    return runtime


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _ref64(x, layers):
    with np.errstate(all="ignore"):
        return O.mlp(x.astype(np.float64), [(w.astype(np.float64), b.astype(np.float64)) for w, b in layers])


def _scale64(x, layers):
    """sum_k |x_k w_kn| pushed through the layers: the magnitude fp32 rounding errors are relative to."""
    with np.errstate(all="ignore"):
        cur = np.abs(x.astype(np.float64))
        for w, b in layers:
            cur = cur @ np.abs(w.astype(np.float64)) + np.abs(b.astype(np.float64))
        return cur


@pytest.mark.parametrize("mlp", [0, 2, 3, 4])
@pytest.mark.parametrize("rb", [1, 2, 4])
def test_op_mlp_ex_split_chain(rt, mlp, rb):
    """Every MLP the split chain is built for (all but the 512-wide lit_mlp), every tile height."""
    from diffusionsat_amd import _lib
    _, batch, _ = make_all("medium")
    ws = oracle_weights(3)
    s = rt.Session(rt.Graph(batch), rt.Weights(ws), max_rounds=2)
    name = O.MLP_ORDER[mlp]
    rows = batch.n_clauses if mlp == 2 else batch.n_vars
    x = np.random.default_rng(20 + mlp).standard_normal((rows, ws[name][0][0].shape[0])).astype(np.float32)
    try:
        y = s.op_mlp(mlp, _dev(x), impl=2, rb=rb).cpu().numpy()
    except _lib.DsatError as e:
        # the plane images of a 32- / 64-row tile of the 384- and 528-wide inputs exceed the CU's 160 KiB: refused,
        # not run; 16-row tiles always fit
        assert "LDS" in str(e) and ((mlp == 3 and rb >= 2) or (mlp == 2 and rb == 4)), (mlp, rb, str(e))
        return
    ref, scale = _ref64(x, ws[name]), _scale64(x, ws[name])
    assert np.all(np.isfinite(y))
    assert (np.abs(y - ref) / scale).max() < 5e-7                    # fp32-level: 1.5e-7 of sum|xw| per layer
    y1 = s.op_mlp(mlp, _dev(x), impl=1, rb=1, nwaves=4).cpu().numpy()
    assert (np.abs(y1 - ref) / scale).max() < 5e-7                   # the fp32-MFMA chain, for comparison
    with pytest.raises(_lib.DsatError):
        s.op_mlp(1, _dev(np.zeros((4, 137), np.float32)), impl=2, rb=1)     # lit_mlp: 32 column tiles


def _edge_session(rt, ws):
    _, batch, _ = make_all("medium")
    return rt.Session(rt.Graph(batch), rt.Weights(ws), max_rounds=2), batch


def test_split_products_at_the_ends_of_the_input_range(rt):
    ws = oracle_weights(3, bias_scale=0.0)
    mlp, name = 0, "variables_query"                                   # 137 -> 153 -> 128
    ws[name] = [ws[name][0], (ws[name][1][0] * np.float32(0.1), ws[name][1][1])]   # 153 terms of ~5e37 must not overflow
    s, batch = _edge_session(rt, ws)
    k = ws[name][0][0].shape[0]
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((96, k)) * 0.5).astype(np.float32)
    tiny = np.float32(1e-40)                                           # denormal
    x[1, :] = 0; x[1, 3] = np.float32(0.99) * FLT_MAX                  # next to the end of the range (below bf16's)
    x[2, :] = 0; x[2, 7] = -np.float32(0.99) * FLT_MAX
    x[3, :] = tiny * rng.integers(1, 9, k)                             # only denormals
    x[4, ::3] = tiny                                                   # denormals among ordinary values
    x[5, :] = np.float32(2.0) ** -120 * rng.standard_normal(k)         # normal, but the low slices are denormal
    x[6, 5] = np.inf
    x[7, 9] = -np.inf
    x[8, 11] = np.nan
    x[9, :] = 0; x[9, 3] = FLT_MAX                                     # the band above bf16's largest finite value
    x[10, :] = 0; x[10, 3] = BF16_MAX                                  # ... and its lower end: still exact
    y1 = s.op_mlp(mlp, _dev(x), impl=1, rb=1, nwaves=4).cpu().numpy()  # v_mfma_f32_16x16x4_f32
    y2 = s.op_mlp(mlp, _dev(x), impl=2, rb=1).cpu().numpy()            # six bf16-slice products
    ref, scale = _ref64(x, ws[name]), _scale64(x, ws[name])
    poisoned = [6, 7, 8]
    for r in range(x.shape[0]):
        if r in poisoned:                                              # inf / NaN propagate, as in x @ W
            assert not np.isfinite(y1[r]).any() and not np.isfinite(y2[r]).any(), r
            continue
        if r == 9:                                                     # the documented band (module docstring)
            assert np.isfinite(y1[r]).all() and not np.isfinite(y2[r]).any()
            continue
        assert np.isfinite(y1[r]).all() and np.isfinite(y2[r]).all(), r
        # fp32-level error relative to sum|xw|; the matrix pipes may flush denormal operands, which moves a result by
        # at most (number of terms) x 1.2e-38 x |w| -- the absolute term
        tol = 1e-6 * scale[r] + 1e-36
        assert (np.abs(y2[r] - ref[r]) <= tol).all(), (r, np.abs(y2[r] - ref[r]).max(), scale[r].max())
        assert (np.abs(y1[r] - ref[r]) <= tol).all(), (r, np.abs(y1[r] - ref[r]).max(), scale[r].max())
    # rows are independent: the ordinary rows next to the poisoned ones are untouched
    assert (np.abs(y2[11:] - ref[11:]) / scale[11:]).max() < 5e-7


def test_split_products_at_the_ends_of_the_weight_range(rt):
    """Weights are sliced on the host: a weight next to (or at) FLT_MAX keeps all 24 bits; denormal weights may be
    flushed by the pipe (absolute effect < 1e-36).  Non-finite weights are refused at load."""
    from diffusionsat_amd import _lib
    ws = oracle_weights(3, bias_scale=0.0)
    name = "variables_output"                                          # 128 -> 128 -> 8
    w0 = ws[name][0][0].copy()
    w1 = ws[name][1][0].copy()
    w0[:, 0] = 0; w0[5, 0] = FLT_MAX                                   # hidden unit 0 = FLT_MAX * x[5]
    w0[:, 1] = 0; w0[6, 1] = -np.float32(0.999) * FLT_MAX
    w0[:, 2] = np.float32(1e-40)                                       # a column of denormal weights
    w0[:, 3] *= np.float32(2.0) ** -110
    w1[0:2, :] *= np.float32(1e-3)                                     # keep the second layer finite
    ws[name] = [(w0, ws[name][0][1]), (w1, ws[name][1][1])]
    s, batch = _edge_session(rt, ws)
    rng = np.random.default_rng(6)
    x = (rng.standard_normal((64, 128)) * 0.5).astype(np.float32)
    x[:, 5] = np.float32(1e-3) * rng.standard_normal(64)               # FLT_MAX * 1e-3: finite
    x[:, 6] = np.float32(1e-3) * rng.standard_normal(64)
    y1 = s.op_mlp(4, _dev(x), impl=1, rb=1, nwaves=4).cpu().numpy()
    y2 = s.op_mlp(4, _dev(x), impl=2, rb=1).cpu().numpy()
    ref, scale = _ref64(x, ws[name]), _scale64(x, ws[name])
    assert np.isfinite(ref).all() and np.abs(ref).max() > 1e30         # the big weights really reach the output
    assert np.isfinite(y1).all() and np.isfinite(y2).all()
    assert (np.abs(y2 - ref) <= 1e-6 * scale + 1e-36).all(), (np.abs(y2 - ref) / scale).max()
    assert (np.abs(y1 - ref) <= 1e-6 * scale + 1e-36).all()
    bad = oracle_weights(3)
    bad[name][0][0][0, 0] = np.inf
    with pytest.raises(_lib.DsatError, match="non-finite"):
        rt.Weights(bad)
