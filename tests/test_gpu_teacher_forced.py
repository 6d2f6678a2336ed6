"""-m gpu: teacher-forced per-layer, per-kernel parity of the segmenter's forward and backward.

The whole-step tests (test_gpu_segmenter.py) can only say that the HIP step's gradients are in the same error class as a float32 CPU
evaluation (1e-3..4e-2 of max|g| for ANY two float32 evaluation orders): a band that could hide a real 1e-3 defect in one backward
kernel.  This test removes the amplification instead of tolerating it.  The oracle runs the full graph once; then EVERY kernel of
EVERY conv(-dropout-BN-shortcut-activation) unit is run on the oracle's own tensors —

    pnp_conv2d_fwd      (x, w)                                   -> conv output after dropout
    pnp_bn_stats        (conv output)                            -> batch mean / variance
    pnp_bn_apply        (conv output, stats, gamma, beta, shortcut)   -> unit output
    pnp_bn_bwd          (upstream gradient, unit output, conv output, stats, gamma) -> d conv accumulator, dgamma, dbeta, dshortcut
    pnp_conv2d_dgrad    (d conv accumulator, w)                  -> dx
    pnp_conv2d_wgrad    (x, d conv accumulator)                  -> dw

— i.e. each kernel receives ITS OWN saved inputs and ITS OWN upstream gradient from the oracle, and must reproduce the oracle's output
of that same step to 1e-4 of max|ref| (north_star's gradient bar; the kernels in fact land at 1e-6..1e-5).

Why the inputs must be forced one kernel at a time, and where the whole-step band comes from: the unit's output passes a leaky-ReLU,
so the backward pass multiplies the upstream gradient by 1 or 0.2 depending on the SIGN of the forward output.  Two float32 forward
evaluations differ by ~1e-6 of max|out|; among the 10^6..10^7 activations of a layer a handful sit closer to zero than that and flip
their slope, which changes the gradient AT THOSE PIXELS by 80 % — 1e-3..2e-2 of max|g| in the max norm, per layer, for any float32
implementation (measured here: feeding the unit the HIP path's own forward output instead of the oracle's reproduces exactly that
band, with every kernel individually at 1e-6).  Dropout masks and max-pool arg-maxes are hash- / data-defined and identical on both
sides; the slope flips are the only amplifier, and with the oracle's output forced they are gone.

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
SEED = 3


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


from parity_util import batch_moments64, rel as _rel  # noqa: E402  (float64, on the device the product's result lives on)


def _run_units(dev, B, dtype, keep_prob, seed):
    K = pkg("kernels")
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dtype)
    y = torch.from_numpy(T.label_decomp(5, _blob_labels(rng, B))).to(dtype)
    V = nets.make_variables(_state(seed + 1), dtype=dtype)
    units = []
    logits = nets.segmenter_forward(V, x, keep_prob, True, True, seed=SEED, units=units)
    cost, _, _, _ = nets.segmenter_cost(V, logits, y)
    cost.backward()                         # d cost: every filter is used by exactly one unit, so w.grad is that unit's dw
    f32 = lambda t: t.detach().float().contiguous().to(dev)
    rows, worst = [], {}
    flips = 0
    for u in units:
        w64, xin, sc, out, yc, yd = V[u["w"]], u["x"], u["shortcut"], u["out"], u["conv"], u["dropped"]
        keep, sid = float(u["keep"]), int(u["sid"])
        xd, wd = f32(xin), f32(w64)
        R = wd.shape[0]
        xin_d, padding = xd, u["padding"]
        if padding == "SYMMETRIC":          # layers._prepad: tf.pad SYMMETRIC materialised, then a VALID convolution
            xin_d, padding = K.sympad_fwd(xd, R // 2), "VALID"
        g = K.conv_geom(tuple(xin_d.shape), tuple(wd.shape), 1, u["dil"], padding)
        errs = {}
        # ---- forward kernels on the oracle's inputs
        conv_hip = K.conv2d_fwd(xin_d, wd, g, keep, SEED, sid)
        errs["conv"] = _rel(conv_hip, yd)
        dyd = yd.grad                                   # gradient w.r.t. the conv output AFTER dropout (= BN input)
        dyc = yc.grad                                   # ... w.r.t. the conv accumulator (dropout mask applied)
        if u["bn"] is not None:
            gam, bet = f32(V[u["bn"] + "/gamma"]), f32(V[u["bn"] + "/beta"])
            xc = f32(yd)
            mean_h, var_h = K.bn_stats(xc)
            P = yd.numel() // yd.shape[-1]
            m64, v64 = batch_moments64(yd, dev)
            errs["mean"], errs["var"] = _rel(mean_h, m64) if float(m64.abs().max()) > 1e-3 else 0.0, _rel(var_h, v64)
            mean_d, var_d = m64.float().to(dev), v64.float().to(dev)
            scd = f32(sc) if sc is not None else None
            alpha = 0.2 if u["act"] else -1.0
            out_hip = K.bn_apply(xc, mean_d, var_d, gam, bet, scd, 1e-3, alpha)
            errs["out"] = _rel(out_hip, out)
            flips += int(((out_hip > 0) != (f32(out) > 0)).sum())
            # ---- backward of BN (+activation, +shortcut, +dropout mask) on the oracle's tensors
            need_sc = sc.shape[-1] if (sc is not None and sc.requires_grad) else 0
            dxc, dgamma, dbeta, dsc = K.bn_bwd(f32(out.grad), f32(out), xc, mean_d, var_d, gam, need_sc, 1e-3, alpha, True, keep, SEED, sid)
            errs["dconv"] = _rel(dxc, dyc)
            errs["dgamma"], errs["dbeta"] = _rel(dgamma, V[u["bn"] + "/gamma"].grad), _rel(dbeta, V[u["bn"] + "/beta"].grad)
            if need_sc:
                errs["dshortcut"] = _rel(dsc, sc.grad)
        else:
            errs["dconv"] = _rel(K.dropout(f32(dyd), keep, SEED, sid) if keep < 1.0 else f32(dyd), dyc)
        # ---- conv backward kernels on the oracle's d(conv accumulator)
        dyc_d = f32(dyc)
        errs["dw"] = _rel(K.conv2d_wgrad(xin_d, dyc_d, g), w64.grad)
        if xin.requires_grad:
            dx = K.conv2d_dgrad(dyc_d, wd, g)
            if u["padding"] == "SYMMETRIC":
                dx = K.sympad_bwd(dx, R // 2)
            errs["dx"] = _rel(dx, xin.grad)
        rows.append((u["w"], u["bn"], errs))
        for k_, e in errs.items():
            if e > worst.get(k_, (-1.0, ""))[0]:
                worst[k_] = (e, u["w"])
    tag = "B=%d vs %s oracle" % (B, str(dtype).split(".")[-1])
    for k_, (e, w) in sorted(worst.items()):
        print("teacher-forced %s: worst %-9s %.3e at %s" % (tag, k_, e, w))
    print("teacher-forced %s: %d units; activations whose sign differs between the HIP forward and the oracle's: %d" % (tag, len(rows), flips))
    bad = [(w, b, {k_: "%.2e" % e for k_, e in errs.items() if e >= TOL}) for w, b, errs in rows if any(e >= TOL for e in errs.values())]
    assert not bad, "kernels beyond %.0e: %s" % (TOL, bad[:6])
    return rows


def test_every_kernel_of_every_unit_teacher_forced_vs_float64(dev):
    rows = _run_units(dev, 2, torch.float64, 0.75, seed=11)
    assert len(rows) == 33                      # every convolution of the segmenter is a unit


@pytest.mark.slow
def test_every_kernel_of_every_unit_teacher_forced_B16_vs_float32(dev):
    """BASELINE batch: 128x128 tiles, 7-way filter-gradient splits, data-gradient reduction splits — the kernel variants B=2 never plans"""
    _run_units(dev, 16, torch.float32, 0.75, seed=12)


def test_slope_flips_explain_the_whole_unit_band(dev):
    """the claim in the header, measured: run one unit through the product's autograd unit (functional.ConvBNActFn: HIP forward output
    decides the leaky-ReLU slopes) and through the forced kernels; the first may differ from the oracle by the flip band, and does so
    ONLY at activations whose forward value is within float32 noise of zero"""
    F, K = pkg("functional"), pkg("kernels")
    rng = np.random.default_rng(5)
    N, H, C = 4, 32, 256
    x = rng.standard_normal((N, H, H, C)).astype(np.float32)
    w = (rng.standard_normal((3, 3, C, C)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    dout = rng.standard_normal((N, H, H, C)).astype(np.float32)
    xo = torch.from_numpy(x).double().requires_grad_(True)
    wo = torch.from_numpy(w).double().requires_grad_(True)
    g64, b64 = torch.ones(C, dtype=torch.float64, requires_grad=True), torch.zeros(C, dtype=torch.float64, requires_grad=True)
    yc = T.dropout(T.conv2d(xo, wo), 0.75, 9, 4)
    out = T.leaky_relu(T.batch_norm(yc, g64, b64, torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64), True))
    out.backward(torch.from_numpy(dout).double())
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    wd = torch.from_numpy(w).to(dev).requires_grad_(True)
    gd, bd = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
    geo = K.conv_geom(x.shape, w.shape, 1, 1, "SAME")
    od = F.ConvBNActFn.apply(xd, wd, gd, bd, torch.zeros(C, device=dev), torch.ones(C, device=dev), None, geo, 0.75, 9, 4, True, 0.2)
    od.backward(torch.from_numpy(dout).to(dev))
    flipped = (od.detach().cpu() > 0) != (out.detach() > 0)
    nflip = int(flipped.sum())
    near_zero = float(out.detach().abs()[flipped].max()) if nflip else 0.0
    e_dx = _rel(xd.grad, xo.grad)
    print("own-forward unit: %d slope flips of %d activations (largest |out| among them %.2e), dx error %.2e, dw error %.2e" % (
        nflip, flipped.numel(), near_zero, e_dx, _rel(wd.grad, wo.grad)))
    assert near_zero < 1e-5 * float(out.detach().abs().max())          # flips only where the forward value is float32 noise
    if nflip == 0:
        assert e_dx < TOL
