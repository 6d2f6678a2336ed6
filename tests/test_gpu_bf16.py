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
    # which kernel symbols run is OBSERVED (pnp_prof_*: the library records every convolution launch by name), not inferred from errors
    def ran(fn, cls):
        L.prof_enable(cls)
        out = fn()
        torch.cuda.synchronize()
        L.prof_enable(0)
        return out, [r["name"] for r in L.prof_summary()]
    L.prof_summary()                                       # drop records of earlier tests
    y, k_fwd = ran(lambda: K.conv2d_fwd(xd, wd, g), L.PROF_CONV_FWD)
    dx, k_dg = ran(lambda: K.conv2d_dgrad(dyd, wd, g), L.PROF_CONV_DGRAD)
    dw, k_wg = ran(lambda: K.conv2d_wgrad(xd, dyd, g), L.PROF_CONV_WGRAD)
    on_bf16 = lambda names: bool(names) and all("bf16" in n for n in names)
    # oracle: float64 convolution of the bf16-rounded operands (each gradient rounds ITS two operands)
    r = lambda a: T.round_bf16(torch.from_numpy(a)).double()
    xr, wr, dyr = r(x), r(w), r(dy)
    yo = T.conv2d(xr, wr, stride, dil, padding)
    xg = xr.clone().requires_grad_(True)
    T.conv2d(xg, wr, stride, dil, padding).backward(dyr)
    wg = wr.clone().requires_grad_(True)
    T.conv2d(xr, wg, stride, dil, padding).backward(dyr)
    errs = {"y": _rel(y, yo), "dx": _rel(dx, xg.grad), "dw": _rel(dw, wg.grad)}
    # a gradient that ran on a float32 kernel (conv_bf16.hip's header says which do: e.g. all stride phases of a small-map data gradient
    # in one conv_dgrad_phases_kernel launch) is held to float64 of the UNROUNDED operands instead — at the same 2e-5, not a loose bar
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    w64 = torch.from_numpy(w).double().requires_grad_(True)
    T.conv2d(x64, w64, stride, dil, padding).backward(torch.from_numpy(dy).double())
    if not on_bf16(k_dg):
        errs["dx"] = _rel(dx, x64.grad)
    if not on_bf16(k_wg):
        errs["dw"] = _rel(dw, w64.grad)
    print("   kernels: fwd %s | dgrad %s | wgrad %s" % (sorted(set(k_fwd)), sorted(set(k_dg)), sorted(set(k_wg))))
    # the fp32 path on the same data, for scale: bf16 rounding of the operands moves results by ~2^-9 relative
    g32 = K.conv_geom(x.shape, w.shape, stride, dil, padding, dtype=L.DTYPE_F32)
    moved = _rel(K.conv2d_fwd(xd, wd, g32), yo)
    print("bf16 conv %s: vs rounded-operand oracle %s ; fp32 kernel vs the same oracle %.2e" % (case, {k_: "%.2e" % e for k_, e in errs.items()}, moved))
    # which of the three run on the bf16 kernels (header of conv_bf16.hip): forward needs C % 32 == 0, the data gradient (a convolution
    # whose input channels are the K filters) K % 32 == 0, the filter gradient stride 1 and rows of >= 32 pixels; the others stay fp32
    # ... and 16 / 32 input channels with 32 / 64 filters on >= 8192 pixels go to conv_small.hip's fp32 16x16x4 tiles (faster there)
    small_wg = C in (16, 32) and Kf in (32, 64) and k == 3 and stride == 1 and dil == 1 and N * H * W >= 8192
    wg_bf16 = stride == 1 and W >= 32 and not small_wg
    dg_bf16 = Kf % 32 == 0
    assert on_bf16(k_fwd), k_fwd                                    # the forward of every CASE is a bf16-tile layer
    assert errs["y"] < 2e-5 and errs["dx"] < 2e-5 and errs["dw"] < 2e-5, errs        # whichever arithmetic ran, it is exact to its oracle
    assert on_bf16(k_wg) == wg_bf16, (k_wg, wg_bf16)
    if stride == 1:
        assert on_bf16(k_dg) == dg_bf16, (k_dg, dg_bf16)
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
    """whole network: (a) inference-mode forward of the bf16 path against the fp32 path, held to the deviation the CPU oracle predicts
    for operand rounding on the same weights (two bf16 evaluations of a 33-layer network decorrelate — an operand that differs in
    its last fp32 bits rounds to the other bf16 neighbour — so the rounded oracle predicts the SIZE of the deviation, not its sign);
    (b) a few training steps with bf16 convolutions track the fp32 run"""
    ss, F = pkg("source_segmenter"), pkg("functional")
    B = 2
    rng = np.random.default_rng(8)
    x = rng.standard_normal((B, 256, 256, 3)).astype(np.float32)
    y = T.label_decomp(5, _blob_labels(rng, B))
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    state = {}
    for k, s_ in nets.segmenter_variable_shapes().items():
        if "Variable" in k:
            state[k] = (rng.standard_normal(s_) * np.sqrt(2.0 / (s_[0] * s_[1] * s_[2])) * 0.9).astype(np.float32)
        elif k.endswith("moving_mean"):
            state[k] = (0.05 * rng.standard_normal(s_)).astype(np.float32)
        elif k.endswith(("gamma", "moving_variance")):
            state[k] = (1.0 + 0.1 * rng.random(s_)).astype(np.float32)
        else:
            state[k] = np.zeros(s_, np.float32)

    def run(dtype):
        F.set_conv_dtype(dtype)
        try:
            net = ss.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, cost_kwargs=dict(COST), seed=3)
            net.store.load_state_dict(state)
            with torch.no_grad():
                lg = net.forward(xd, keep_prob=1.0, main_bn=False, adapt_bn=False).cpu()
            tr = ss.Trainer(net, None, None, num_cls=5, batch_size=B, optimizer="adam", opt_kwargs={"learning_rate": 1e-3})
            tr.opt = tr._get_optimizer(10)
            losses = [float(tr.train_step(xd, yd, 0.75, i)) for i in range(4)]
            assert net.store.arena.dtype == torch.float32          # fp32 master weights in either mode
            return lg, losses
        finally:
            F.set_conv_dtype("f32")
    l16, losses16 = run("bf16")
    l32, losses32 = run("f32")
    V = nets.make_variables(state, requires_grad=False)
    with torch.no_grad():
        lo32 = nets.segmenter_forward(V, torch.from_numpy(x), 1.0, False, False)
        lo16 = nets.segmenter_forward(V, torch.from_numpy(x), 1.0, False, False, operand_round=T.round_bf16,
                                      round_if=lambda s_: s_[2] % 32 == 0 and s_[3] % 4 == 0)

    def stats(a, b):
        la, lb = a.argmax(-1), b.argmax(-1)
        dice = [2.0 * float(((la == c) & (lb == c)).sum()) / (float((la == c).sum()) + float((lb == c).sum()) + 1e-7) for c in range(5)
                if int((lb == c).sum()) > 0]
        return _rel(a, b), float((la == lb).float().mean()), min(dice)
    e_hip, agree_hip, dice_hip = stats(l16, l32)
    e_cpu, agree_cpu, dice_cpu = stats(lo16, lo32)
    print("bf16 vs fp32, whole segmenter (inference BN): HIP logits %.3e argmax agreement %.5f min Dice %.4f | CPU oracle prediction "
          "%.3e %.5f %.4f | fp32 paths HIP vs CPU %.2e" % (e_hip, agree_hip, dice_hip, e_cpu, agree_cpu, dice_cpu, _rel(l32, lo32)))
    print("bf16 losses %s | fp32 losses %s" % (["%.5f" % v for v in losses16], ["%.5f" % v for v in losses32]))
    assert _rel(l32, lo32) < 1e-4
    assert 1e-4 < e_hip < 2.5 * e_cpu and e_hip < 0.1                           # tests/test_bf16_budget.py's bounds, and the oracle's size
    assert agree_hip > min(0.97, agree_cpu - 0.01) and dice_hip > min(0.95, dice_cpu - 0.02)
    # the bf16 run tracks the fp32 run step by step (same data, same dropout masks, same Adam): within 2 % after four updates
    assert all(np.isfinite(v) for v in losses16)
    assert all(abs(a_ - b_) < 0.02 * abs(b_) + 1e-3 for a_, b_ in zip(losses16, losses32)), (losses16, losses32)


def test_joint_step_bf16_within_budget(dev):
    """BASELINE configs[4] is the JOINT segmenter+GAN step in bf16: one discriminator step and one generator step of the adaptation graph
    with bf16 MFMA operands against the same steps in float32 (same state, same dropout masks).  Operand rounding moves a 50-layer
    graph's outputs by ~1e-2 relative (tests/test_bf16_budget.py measures the segmenter's share on the CPU); gradients keep their
    direction.  Also checked: the bf16 kernels really ran (observed through pnp_prof_*), master weights stay float32."""
    adv, F, L = pkg("adversarial"), pkg("functional"), pkg("_lib")
    from test_gpu_adversarial import COST as GCOST, NETCFG, he_state
    B = 2
    rng = np.random.default_rng(0)
    mr = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dev)
    ct = torch.from_numpy((rng.standard_normal((B, 256, 256, 3)) * 1.2 + 0.1).astype(np.float32)).to(dev)
    net0 = adv.Full_DRN(channels=3, n_class=5, batch_size=B, cost_kwargs=dict(GCOST), network_config=dict(NETCFG), device=dev, seed=1)
    sd = he_state(net0, 7)
    del net0

    def run(dtype):
        F.set_conv_dtype(dtype)
        try:
            net = adv.Full_DRN(channels=3, n_class=5, batch_size=B, cost_kwargs=dict(GCOST), network_config=dict(NETCFG), device=dev, seed=1)
            net.store.load_state_dict(sd)
            L.prof_summary()
            L.prof_enable(L.PROF_CONV_FWD | L.PROF_CONV_DGRAD | L.PROF_CONV_WGRAD)
            dl = float(net.dis_loss_and_grads(mr, ct, 0.75, drop_seed=11))
            sc_d = net.miu_dis * sum(float(v.abs().mean()) * (net.lambda_mask_loss if "mask" in k else 1.0) for k, v in net.critic_scores.items())
            g_dis = {v.name: v.tensor.grad.detach().cpu().clone() for v in net.store.trainable() if "cls" in v.name}
            lg = net.ct_logits.cpu().clone()
            net.store.load_state_dict(sd)
            gl = float(net.gen_loss_and_grads(ct, 0.75, drop_seed=12))
            sc_g = net.miu_gen * sum(float(v.abs().mean()) * (net.lambda_mask_loss if "mask" in k else 1.0) for k, v in net.critic_scores.items())
            g_gen = {v.name: v.tensor.grad.detach().cpu().clone() for v in net.store.trainable() if v.name.startswith("adapt_")}
            torch.cuda.synchronize()
            L.prof_enable(0)
            names = [r_["name"] for r_ in L.prof_summary() for _ in range(r_["launches"])]
            assert net.store.arena.dtype == torch.float32
            return dl, gl, lg, g_dis, g_gen, names, sc_d, sc_g
        finally:
            L.prof_enable(0)
            F.set_conv_dtype("f32")
    d16, g16, lg16, gd16, gg16, names16, _, _ = run("bf16")
    d32, g32, lg32, gd32, gg32, names32, sc_d, sc_g = run("f32")
    share = sum("bf16" in n for n in names16) / float(len(names16))
    assert not any("bf16" in n for n in names32) and share > 0.5, share

    def cos(a, b):
        a, b = a.double().reshape(-1), b.double().reshape(-1)
        return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
    c_dis = np.array([cos(gd16[k], gd32[k]) for k in gd32 if float(gd32[k].abs().max()) > 0])
    c_gen = np.array([cos(gg16[k], gg32[k]) for k in gg32 if float(gg32[k].abs().max()) > 0])
    e_lg = _rel(lg16, lg32)
    print("joint step bf16 vs fp32 (B=2): dis loss %.6e vs %.6e (scale of its terms %.2e), gen loss %.6e vs %.6e (scale %.2e), CT logits %.3e; "
          "gradient cosine dis median %.5f min %.5f, gen median %.5f min %.5f; %.0f %% of %d conv launches on bf16 kernels" % (
              d16, d32, sc_d, g16, g32, sc_g, e_lg, np.median(c_dis), c_dis.min(), np.median(c_gen), c_gen.min(), 100 * share, len(names16)))
    assert np.isfinite([d16, g16]).all()
    # bars = ~3x the deviations MEASURED on the GPU (round 3, gpurun_out/r3f): CT logits 6.6e-3, losses 0.55 % / 0.37 %, gradient cosine
    # dis median 0.979 (min 0.950), gen median 0.952 (min 0.905) — operand rounding (2^-9 per operand) through ~50 layers of leaky-ReLU /
    # dropout kinks turns the gradient by a few degrees; it does not change what it points at
    assert 1e-4 < e_lg < 2e-2                                     # the segmenter's budget (tests/test_bf16_budget.py: 1.1e-2 typical)
    # the WGAN losses are DIFFERENCES of mean critic scores (adversarial.py:455-459) over B = 2 samples: a cancelling quantity, so the bar
    # is taken against the size of the terms, not of their difference.  Measured over 5 input seeds (profiles/r04_bf16_loss_noise.txt): a
    # single critic score moves by up to +-10 % of the mean |score| under bf16 operand rounding, with random sign, for the resident AND
    # the staged-rounding kernels alike; the 8-score discriminator loss by up to 1.6 % of the terms' scale, the 2-4-score generator loss
    # by up to 3.1 %.  Bars: 4 % / 8 %.
    assert abs(d16 - d32) < 0.04 * sc_d + 1e-6 and abs(g16 - g32) < 0.08 * sc_g + 1e-6, (d16, d32, sc_d, g16, g32, sc_g)
    assert np.median(c_dis) > 0.94 and c_dis.min() > 0.85 and np.median(c_gen) > 0.86 and c_gen.min() > 0.75
