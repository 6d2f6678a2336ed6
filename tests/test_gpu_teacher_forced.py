"""-m gpu: teacher-forced per-layer backward parity.

The whole-step tests (test_gpu_segmenter.py) can only say that the HIP step's gradients are in the same error class as a float32 CPU
evaluation: 30+ layers of kinks (leaky-ReLU, dropout, max-pool ties, the 0.005 clip) amplify single roundings, so end-to-end gradient
differences of 1e-3..4e-2 of max|g| appear between ANY two float32 evaluation orders.  That band could hide a real 1e-3 defect in one
backward kernel.  Here every conv(-BN-shortcut-activation) unit of the segmenter is checked in isolation: the oracle runs the full
graph once, each unit's saved input / shortcut / filter / BN parameters and ITS OWN upstream gradient are handed to the product's
autograd unit (functional.ConvBNActFn / Conv2dDropFn -> pnp_conv2d_fwd, pnp_bn_*, pnp_conv2d_dgrad, pnp_conv2d_wgrad), and the unit's
dx, dw, dgamma, dbeta, dshortcut must match the oracle's to 1e-4 of max|ref| — north_star's gradient bar, per kernel and per layer.

  * B=2 against the float64 oracle (every unit);
  * B=16 (BASELINE batch: other tiles / reduction splits are planned) against the float32 oracle (slow).
"""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import nets
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu
COST = {"cross_flag": True, "miu_cross": 1.0, "dice_flag": True, "miu_dice": 1.0, "regularizer": 1e-4}
TOL = 1e-4


def _blob_labels(rng, B):
    yy, xx = np.mgrid[0:256, 0:256]
    lab = np.zeros((B, 256, 256), np.float32)
    for b in range(B):
        for c in range(1, 5):
            cy, cx = rng.integers(40, 216, 2)
            ry, rx = rng.integers(12, 40, 2)
            lab[b][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1] = c
    return lab


def _state(seed):
    """He-scaled filters, non-trivial BN affine parameters"""
    rng = np.random.default_rng(seed)
    sd = {}
    for k, s in nets.segmenter_variable_shapes().items():
        if "Variable" in k:
            sd[k] = (rng.standard_normal(s) * np.sqrt(2.0 / (s[0] * s[1] * s[2])) * 0.9).astype(np.float32)
        elif k.endswith("gamma"):
            sd[k] = (1.0 + 0.1 * rng.standard_normal(s)).astype(np.float32)
        elif k.endswith("beta"):
            sd[k] = (0.1 * rng.standard_normal(s)).astype(np.float32)
        elif k.endswith("moving_variance"):
            sd[k] = np.ones(s, np.float32)
        else:
            sd[k] = np.zeros(s, np.float32)
    return sd


def _rel(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def _run_units(dev, B, dtype, keep_prob, seed):
    F, K = pkg("functional"), pkg("kernels")
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dtype)
    y = torch.from_numpy(T.label_decomp(5, _blob_labels(rng, B))).to(dtype)
    sd = _state(seed + 1)
    V = nets.make_variables(sd, dtype=dtype)
    units = []
    logits = nets.segmenter_forward(V, x, keep_prob, True, True, seed=3, units=units)
    cost, reg, _, _ = nets.segmenter_cost(V, logits, y)
    cost.backward()                         # d cost: the L2 term only touches the filters directly (added back below)
    worst = {}
    rows = []
    for u in units:
        w64, xin, sc, out = V[u["w"]], u["x"], u["shortcut"], u["out"]
        dout = out.grad
        assert dout is not None, u["w"]
        xd = xin.detach().float().to(dev).requires_grad_(xin.requires_grad)
        wd = w64.detach().float().to(dev).requires_grad_(True)
        R = wd.shape[0]
        xin_d, padding = xd, u["padding"]
        if padding == "SYMMETRIC":                      # layers._prepad: tf.pad SYMMETRIC materialised, then a VALID convolution
            xin_d, padding = F.SymPadFn.apply(xd, R // 2), "VALID"
        g = K.conv_geom(tuple(xin_d.shape), tuple(wd.shape), 1, u["dil"], padding)
        if u["bn"] is None:
            od = F.Conv2dDropFn.apply(xin_d, wd, g, float(u["keep"]), 3, u["sid"])
            leaves = {"dw": (wd, w64.grad)}
        else:
            gam = V[u["bn"] + "/gamma"].detach().float().to(dev).requires_grad_(True)
            bet = V[u["bn"] + "/beta"].detach().float().to(dev).requires_grad_(True)
            C = gam.numel()
            mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            scd = sc.detach().float().to(dev).requires_grad_(True) if sc is not None else None
            od = F.ConvBNActFn.apply(xin_d, wd, gam, bet, mm, mv, scd, g, float(u["keep"]), 3, u["sid"], True, 0.2 if u["act"] else -1.0)
            leaves = {"dw": (wd, w64.grad), "dgamma": (gam, V[u["bn"] + "/gamma"].grad), "dbeta": (bet, V[u["bn"] + "/beta"].grad)}
            if sc is not None:
                leaves["dshortcut"] = (scd, sc.grad)
        e_out = _rel(od, out)
        od.backward(dout.detach().float().to(dev))
        if xin.requires_grad:
            leaves["dx"] = (xd, xin.grad)
        errs = {"out": e_out}
        for name, (leaf, ref) in leaves.items():
            assert ref is not None and leaf.grad is not None, (u["w"], name)
            errs[name] = _rel(leaf.grad, ref)
        rows.append((u["w"], u["bn"], errs))
        for k, e in errs.items():
            if e > worst.get(k, (0, ""))[0]:
                worst[k] = (e, u["w"])
        del xd, wd, od
    for k, (e, w) in sorted(worst.items()):
        print("teacher-forced B=%d %s: worst %-9s %.3e at %s" % (B, str(dtype).split(".")[-1], k, e, w))
    bad = [(w, b, {k: "%.2e" % e for k, e in errs.items() if e >= TOL}) for w, b, errs in rows if any(e >= TOL for e in errs.values())]
    assert not bad, "units beyond %.0e: %s" % (TOL, bad[:6])
    return rows


def test_units_backward_teacher_forced_vs_float64(dev):
    rows = _run_units(dev, 2, torch.float64, 0.75, seed=11)
    assert len(rows) == 33                      # every convolution of the segmenter is a unit


@pytest.mark.slow
def test_units_backward_teacher_forced_B16_vs_float32(dev):
    """BASELINE batch: 128x128 tiles, 7-way filter-gradient splits, data-gradient reduction splits — the kernel variants B=2 never plans"""
    _run_units(dev, 16, torch.float32, 0.75, seed=12)
