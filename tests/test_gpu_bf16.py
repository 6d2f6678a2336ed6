"""-m gpu: BASELINE configs[4] — bf16 MFMA convolution tiles (csrc/conv_bf16.hip), fp32 accumulation / master weights / BN.

Arithmetic under test: both operands of the contraction rounded to bfloat16 (nearest-even), products summed in float32.  The oracle
states exactly that with oracle.tf_ops.round_bf16 on the operands of a float32 (or float64) convolution; a bf16 x bf16 product is
exact in float32, so per-op agreement is float32-summation-order tight (1e-5 of max|ref|), NOT bf16-loose.  Whole-network tolerance:
the budget measured on the CPU in tests/test_bf16_budget.py (logits within 1.1e-2 of max|logit|, argmax agreement >= 99.5 %,
label-map Dice >= 0.994 against the fp32 path)."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import nets
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu
COST = {"cross_flag": True, "miu_cross": 1.0, "dice_flag": True, "miu_dice": 1.0, "regularizer": 1e-4}

# (N, H, W, C, K, k, stride, dil, padding): every class of layer that runs on the bf16 kernels
CASES = [
    (2, 32, 32, 512, 512, 3, 1, 1, "SAME"),       # group_7..9: 128x128 tiles at B=16, 128x64 here
    (2, 32, 32, 512, 512, 3, 1, 2, "SAME"),       # group_8 dilated
    (2, 34, 34, 512, 2560, 3, 1, 1, "VALID"),     # group_10 after the SYMMETRIC pre-pad
    (2, 64, 64, 64, 64, 3, 1, 1, "SAME"),         # group_3
    (2, 128, 128, 32, 32, 3, 1, 1, "SAME"),       # group_2: C = 32, one channel group
    (2, 32, 32, 128, 256, 3, 1, 1, "SAME"),       # inc_dim first convs
    (4, 128, 128, 64, 64, 3, 2, 1, "SAME"),       # critic cls_1_3: stride 2 (forward + stride-phase data gradient)
    (4, 64, 64, 128, 128, 5, 2, 1, "SAME"),       # critic cls_2_3: 5x5 stride 2
    (4, 16, 16, 512, 512, 5, 4, 1, "SAME"),       # critic cls_5_3: 5x5 stride 4, reduction-split forward
    (3, 37, 41, 96, 72, 3, 1, 1, "SAME"),         # ragged: M, K not multiples of the tile
]


def _rel(a, b):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_bf16_fwd_dgrad_wgrad_vs_rounded_oracle(dev, case):
    K, L = pkg("kernels"), pkg("_lib")
    N, H, W, C, Kf, k, stride, dil, padding = case
    rng = np.random.default_rng(sum(case[:7]))
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    w = (rng.standard_normal((k, k, C, Kf)) * np.sqrt(2.0 / (k * k * C))).astype(np.float32)
    g = K.conv_geom(x.shape, w.shape, stride, dil, padding, dtype=L.DTYPE_BF16)
    dy = rng.standard_normal((N, g.OH, g.OW, Kf)).astype(np.float32)
    xd, wd, dyd = (torch.from_numpy(a).to(dev) for a in (x, w, dy))
    y = K.conv2d_fwd(xd, wd, g)
    dx = K.conv2d_dgrad(dyd, wd, g)
    dw = K.conv2d_wgrad(xd, dyd, g)
    # oracle: float64 convolution of the bf16-rounded operands (each gradient rounds ITS two operands)
    r = lambda a: T.round_bf16(torch.from_numpy(a)).double()
    xr, wr, dyr = r(x), r(w), r(dy)
    yo = T.conv2d(xr, wr, stride, dil, padding)
    xg = xr.clone().requires_grad_(True)
    T.conv2d(xg, wr, stride, dil, padding).backward(dyr)
    wg = wr.clone().requires_grad_(True)
    T.conv2d(xr, wg, stride, dil, padding).backward(dyr)
    errs = {"y": _rel(y, yo), "dx": _rel(dx, xg.grad), "dw": _rel(dw, wg.grad)}
    # the fp32 path on the same data, for scale: bf16 rounding of the operands moves results by ~2^-9 relative
    g32 = K.conv_geom(x.shape, w.shape, stride, dil, padding, dtype=L.DTYPE_F32)
    moved = _rel(K.conv2d_fwd(xd, wd, g32), yo)
    print("bf16 conv %s: vs rounded-operand oracle %s ; fp32 kernel vs the same oracle %.2e" % (case, {k_: "%.2e" % e for k_, e in errs.items()}, moved))
    wg_bf16 = stride == 1 and W >= 32        # strided / narrow filter gradients stay on the fp32 kernel (header of conv_bf16.hip)
    assert errs["y"] < 2e-5 and errs["dx"] < 2e-5, errs
    assert errs["dw"] < (2e-5 if wg_bf16 else 1e-2), errs
    assert moved > 1e-4                       # the bf16 path really rounds (a silent fp32 fallback would agree with fp64 to 1e-6)


def _blob_labels(rng, B):
    yy, xx = np.mgrid[0:256, 0:256]
    lab = np.zeros((B, 256, 256), np.float32)
    for b in range(B):
        for c in range(1, 5):
            cy, cx = rng.integers(40, 216, 2)
            ry, rx = rng.integers(12, 40, 2)
            lab[b][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1] = c
    return lab


def test_segmenter_bf16_forward_within_budget_and_trains(dev):
    ss, F = pkg("source_segmenter"), pkg("functional")
    B = 2
    rng = np.random.default_rng(8)
    x = rng.standard_normal((B, 256, 256, 3)).astype(np.float32)
    y = T.label_decomp(5, _blob_labels(rng, B))
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)

    def build(dtype):
        F.set_conv_dtype(dtype)
        net = ss.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, cost_kwargs=dict(COST), seed=3)
        sd = net.store.state_dict()
        for k, a in sd.items():
            if "/Variable" in k:
                sd[k] = (a * (np.sqrt(2.0 / (a.shape[0] * a.shape[1] * a.shape[2])) / 0.01 * 0.9)).astype(np.float32)
        net.store.load_state_dict(sd)
        return net, sd
    try:
        net16, sd = build("bf16")
        with torch.no_grad():
            l16 = net16.forward(xd, keep_prob=1.0, main_bn=True, adapt_bn=True).cpu()
        tr = ss.Trainer(net16, None, None, num_cls=5, batch_size=B, optimizer="adam", opt_kwargs={"learning_rate": 1e-3})
        tr.opt = tr._get_optimizer(10)
        losses16 = [float(tr.train_step(xd, yd, 0.75, i)) for i in range(4)]
        assert net16.store.arena.dtype == torch.float32          # fp32 master weights
    finally:
        F.set_conv_dtype("f32")
    net32, _ = build("f32")
    with torch.no_grad():
        l32 = net32.forward(xd, keep_prob=1.0, main_bn=True, adapt_bn=True).cpu()
    tr32 = ss.Trainer(net32, None, None, num_cls=5, batch_size=B, optimizer="adam", opt_kwargs={"learning_rate": 1e-3})
    tr32.opt = tr32._get_optimizer(10)
    losses32 = [float(tr32.train_step(xd, yd, 0.75, i)) for i in range(4)]
    # (a) the same arithmetic restated on the CPU: operands rounded to bf16 on the layers the product runs on its bf16 kernels
    V = nets.make_variables(sd, requires_grad=False)
    with torch.no_grad():
        lo = nets.segmenter_forward(V, torch.from_numpy(x), 1.0, True, True, operand_round=T.round_bf16,
                                    round_if=lambda s: s[2] % 32 == 0 and s[3] % 4 == 0)
    e_oracle = _rel(l16, lo)
    # (b) the budget against the fp32 path
    e32 = _rel(l16, l32)
    a16, a32 = l16.argmax(-1), l32.argmax(-1)
    agree = float((a16 == a32).float().mean())
    dice = []
    for c in range(5):
        p, q = (a16 == c), (a32 == c)
        dice.append(2.0 * float((p & q).sum()) / (float(p.sum()) + float(q.sum()) + 1e-7))
    print("bf16 segmenter: logits vs rounded-operand oracle %.3e, vs fp32 path %.3e; argmax agreement %.5f; label-map Dice %s" % (
        e_oracle, e32, agree, ["%.4f" % d for d in dice]))
    print("bf16 losses %s | fp32 losses %s" % (["%.5f" % v for v in losses16], ["%.5f" % v for v in losses32]))
    assert e_oracle < 5e-3                      # same roundings up to bf16 ulp flips of operands that differ in the last fp32 bits
    assert 1e-4 < e32 < 1.1e-2 and agree >= 0.995 and min(dice) >= 0.994
    assert all(np.isfinite(v) for v in losses16) and abs(losses16[-1] - losses32[-1]) < 0.05 * abs(losses32[-1]) + 1e-3
    assert losses16[-1] < losses16[0]
