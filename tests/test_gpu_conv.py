"""-m gpu: the MFMA implicit-GEMM convolution (fwd / dgrad / wgrad) through the C-ABI vs the CPU oracle.
Tolerance: fp32, max-abs error <= 2e-5 * sum_k |a_k*b_k| bound, reported relative to max|ref| (north_star: 1e-4 rel)."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _rel(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# (N,H,W,C,K,R,stride,dil,padding)  — every conv shape class on the reference path, spatially reduced
CASES = [
    (2, 16, 16, 3, 16, 3, 1, 1, "SAME"),        # g1 conv1_1: C=3 scalar loader, K=16 tile
    (2, 16, 16, 16, 16, 3, 1, 1, "SAME"),       # wr1_x: C=16 (vec4 generic), K<32
    (2, 16, 16, 16, 32, 3, 1, 1, "SAME"),       # wr2_1
    (2, 12, 12, 32, 64, 3, 1, 1, "SAME"),       # wr3_1 (odd-ish spatial, M not multiple of 128)
    (2, 8, 8, 64, 128, 3, 1, 1, "SAME"),        # wr4_1
    (1, 8, 8, 128, 256, 3, 1, 1, "SAME"),       # wr5_1
    (1, 8, 8, 256, 512, 3, 1, 1, "SAME"),       # wr7_1
    (1, 8, 8, 512, 512, 3, 1, 2, "SAME"),       # g8 dilated
    (1, 6, 6, 512, 2560, 3, 1, 1, "SYMMETRIC"), # g10 (reduced spatial)
    (2, 24, 24, 40, 5, 5, 1, 1, "SYMMETRIC"),   # output conv 5x5, C=40, K=5 (scalar B path)
    (2, 16, 16, 64, 64, 3, 2, 1, "SAME"),       # critic k3 s2 (asymmetric SAME pad 0,1)
    (2, 16, 16, 128, 128, 5, 2, 1, "SAME"),     # critic k5 s2 (pad 1,2)
    (2, 16, 16, 512, 512, 5, 4, 1, "SAME"),     # critic k5 s4 (pad 0,1)
    (2, 4, 4, 512, 512, 3, 2, 1, "SYMMETRIC"),  # cls_6
    (2, 16, 16, 5, 16, 3, 2, 1, "SAME"),        # mask critic C=5
    (2, 1, 1, 2048, 1, 1, 1, 1, "VALID"),       # FC as 1x1 conv (adversarial.py:395-397)
    (3, 9, 7, 20, 24, 3, 1, 1, "SAME"),         # ragged everything
    (2, 15, 13, 32, 32, 3, 2, 1, "SAME"),       # stride phases with odd extents (phase grids of different sizes)
    (2, 9, 9, 32, 64, 3, 2, 1, "VALID"),        # stride 2 VALID: bottom/right input rows no output reads (dx rows of zeros)
    (1, 14, 14, 32, 64, 5, 3, 1, "SAME"),       # stride 3, 5x5: phases with 2 and 1 taps per axis
    (2, 8, 8, 32, 32, 1, 2, 1, "SAME"),         # filter smaller than the stride: zero-upsampled fallback
    (2, 10, 10, 20, 24, 3, 2, 1, "SAME"),       # stride phases on the generic (C % 32 != 0) loaders
    (2, 64, 64, 3, 16, 3, 1, 1, "SAME"),        # >= 8192 pixels, K <= 16: direct (vector-ALU) filter gradient, 27 pairs x 9 pixel groups
    (2, 64, 64, 16, 16, 3, 1, 1, "SAME"),       # direct wgrad, 144 pairs, one group
    (2, 68, 68, 40, 5, 5, 1, 1, "VALID"),       # direct wgrad, 1000 pairs = 4 per thread, K = 5 (the pre-padded logits conv)
    (2, 128, 128, 5, 16, 3, 2, 1, "SAME"),      # direct wgrad with stride 2 (mask critic m_cls_1)
    (2, 64, 64, 8, 7, 3, 1, 1, "SAME"),         # direct wgrad, K = 7 on the 8-wide instance
    (2, 64, 64, 4, 12, 3, 1, 2, "SAME"),        # direct wgrad, K = 12 on the 16-wide instance, dilation 2
    (2, 64, 64, 32, 12, 3, 1, 2, "SAME"),       # direct narrow-output forward, K = 12 on the 16-wide instance, dilation 2
    (2, 64, 64, 16, 32, 3, 1, 1, "SAME"),       # data gradient with 16 INPUT channels: narrow-output kernel over dy (32 channels)
    (2, 70, 66, 16, 16, 3, 1, 1, "SAME"),       # narrow kernels with ragged 8x32 output patches
    (2, 33, 45, 64, 96, 3, 1, 1, "SAME"),       # ring filter gradient (rows >= 32 pixels), ragged rows / tiles, K not a tile multiple
    (2, 67, 65, 32, 48, 3, 2, 1, "SAME"),       # ring filter gradient walking a STRIDED output with odd extents (OW = 33)
    (1, 70, 70, 32, 64, 3, 2, 1, "VALID"),      # same, VALID: the last input row / column is never read (OW = 34)
    (2, 40, 72, 64, 64, 5, 2, 1, "SAME"),       # 5x5 stride 2 on the ring kernel (OW = 36), phases of the data gradient in one launch
    (2, 64, 64, 32, 64, 3, 1, 1, "SAME"),       # 16x16x4-MFMA filter gradient with 4 filter groups (group_3's first conv)
    (1, 96, 100, 32, 32, 3, 1, 1, "SAME"),      # same, ragged tiles (100 = 3 x 32 + 4), 2 filter groups; data gradient on conv_n16? no: K = 32
    (2, 70, 66, 3, 16, 3, 1, 1, "SAME"),        # the first layer's MFMA kernels with ragged tiles
    (2, 66, 70, 32, 16, 3, 1, 1, "VALID"),      # 32 -> 16 channels, VALID (the data gradient of a 16 -> 32 layer has this shape)
]


def _mk(case, seed=0):
    N, H, W, C, K, R, stride, dil, padding = case
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    w = (rng.standard_normal((R, R, C, K)) / np.sqrt(R * R * C)).astype(np.float32)
    return x, w, stride, dil, padding


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_conv_fwd_dgrad_wgrad(dev, case):
    K = pkg("kernels")
    x, w, stride, dil, padding = _mk(case)
    g = K.conv_geom(x.shape, w.shape, stride, dil, padding)
    xt = torch.from_numpy(x).requires_grad_(True)
    wt = torch.from_numpy(w).requires_grad_(True)
    yo = T.conv2d(xt, wt, stride, dil, padding)
    assert tuple(yo.shape) == (g.N, g.OH, g.OW, g.K)
    dy = torch.from_numpy(np.random.default_rng(1).standard_normal(tuple(yo.shape)).astype(np.float32))
    yo.backward(dy)

    xd, wd, dyd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), dy.to(dev)
    y = K.conv2d_fwd(xd, wd, g)
    yn = K.conv2d_fwd(xd, wd, g, naive=True)
    dx = K.conv2d_dgrad(dyd, wd, g)
    dw = K.conv2d_wgrad(xd, dyd, g)
    torch.cuda.synchronize()
    errs = dict(naive=_rel(yn, yo), y=_rel(y, yo), dx=_rel(dx, xt.grad), dw=_rel(dw, wt.grad))
    print(case, errs)
    assert errs["naive"] < TOL, errs
    assert errs["y"] < TOL, errs
    assert errs["dx"] < TOL, errs
    assert errs["dw"] < TOL, errs
    # "add into" variants (gradient sinks / residual shortcut): dx = dgrad + residual, dw_slot += wgrad — against the plain results above
    res = torch.from_numpy(np.random.default_rng(2).standard_normal(tuple(dx.shape)).astype(np.float32)).to(dev)
    dx2 = K.conv2d_dgrad(dyd, wd, g, residual=res)
    assert _rel(dx2, dx + res) < 2e-6, ("dgrad_add", case, _rel(dx2, dx + res))
    slot0 = torch.from_numpy(np.random.default_rng(3).standard_normal(tuple(dw.shape)).astype(np.float32)).to(dev)
    slot = slot0.clone()
    assert K.conv2d_wgrad(xd, dyd, g, into=slot) is slot
    assert _rel(slot, slot0 + dw) < 2e-6, ("wgrad_acc", case, _rel(slot, slot0 + dw))
    K.conv2d_wgrad(xd, dyd, g, into=slot)            # a second use of the same filters (shared critic weights)
    assert _rel(slot, slot0 + 2 * dw) < 4e-6


def test_odd_channel_group_count_keeps_the_two_stage_kernel(dev):
    """conv_taps3_kernel walks channel groups in pairs for odd tap counts: a layer with an ODD number of 32-channel groups (C = 96)
    must stay on conv_taps_kernel (host predicate in launch_taps; nothing on the device would catch a wrong routing) — observed through
    the profiler's symbol names, and held to the oracle like every other shape"""
    K, L = pkg("kernels"), pkg("_lib")
    rng = np.random.default_rng(4)
    for C, want in ((96, "conv_taps_kernel<"), (64, "conv_taps3_kernel<")):
        x = rng.standard_normal((16, 32, 32, C)).astype(np.float32)          # 16 384 pixels: no reduction split (a split would pair groups per split)
        w = (rng.standard_normal((3, 3, C, 64)) * 0.05).astype(np.float32)
        g = K.conv_geom(x.shape, w.shape, 1, 1, "SAME")
        xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
        L.prof_summary()
        L.prof_enable(L.PROF_CONV_FWD)
        y = K.conv2d_fwd(xd, wd, g)
        torch.cuda.synchronize()
        L.prof_enable(0)
        names = [r["name"] for r in L.prof_summary()]
        assert names and all(n.startswith(want) for n in names), (C, names)
        assert _rel(y, T.conv2d(torch.from_numpy(x), torch.from_numpy(w), 1, 1, "SAME")) < TOL


def test_conv_transpose_detecting(dev):
    """A = identity-like input with ASYMMETRIC weights: catches row/col swaps in the MFMA C/D mapping"""
    K = pkg("kernels")
    N, H, W, C, Kc = 1, 16, 16, 64, 96
    x = np.zeros((N, H, W, C), np.float32)
    for i in range(H * W):
        x[0, i // W, i % W, i % C] = 1.0 + (i % 7)
    w = np.zeros((1, 1, C, Kc), np.float32)
    for c in range(C):
        for k in range(Kc):
            w[0, 0, c, k] = c * 0.01 + k * 1.0
    g = K.conv_geom(x.shape, w.shape, 1, 1, "SAME")
    y = K.conv2d_fwd(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), g).cpu().numpy()
    ref = np.einsum("nhwc,ck->nhwk", x.astype(np.float64), w[0, 0].astype(np.float64))
    assert np.abs(y - ref).max() < 1e-3 * np.abs(ref).max()


@pytest.mark.parametrize("case", [(2, 16, 16, 32, 64, 3, 1, 1, "SAME"),
                                  (2, 8, 8, 512, 512, 3, 1, 1, "SAME"),      # 8 tiles: reduction split 18 ways, dropout in the summing kernel
                                  (2, 16, 16, 512, 512, 5, 4, 1, "SAME"),    # critic k5 s4 -> 4x4 map
                                  (2, 68, 68, 40, 5, 5, 1, 1, "VALID")])     # logits conv on the direct narrow-output kernel
def test_dropout_epilogue_matches_oracle_mask(dev, case):
    K = pkg("kernels")
    x, w, stride, dil, padding = _mk(case)
    g = K.conv_geom(x.shape, w.shape, stride, dil, padding)
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    y0 = K.conv2d_fwd(xd, wd, g).cpu()
    y1 = K.conv2d_fwd(xd, wd, g, keep_prob=0.75, seed=1234567, stream_id=5).cpu()
    mask = T.dropout_mask(tuple(y0.shape), 0.75, 1234567, 5)
    ref = y0 * torch.from_numpy(mask) * np.float32(1.0 / np.float32(0.75))
    assert torch.equal(y1 != 0, (torch.from_numpy(mask) != 0) & (y0 != 0))
    assert _rel(y1, ref) < 1e-6
    assert abs(mask.mean() - 0.75) < (0.01 if mask.size > 100000 else 0.03)
    # the standalone dropout kernel (used in backward) draws the same stream
    y2 = K.dropout(xd.new_tensor(y0.numpy()), 0.75, 1234567, 5).cpu()
    assert _rel(y2, ref) < 1e-6


@pytest.mark.parametrize("shape", [(16, 32, 32, 512, 512, 3, 1, "SAME"), (16, 256, 256, 16, 16, 3, 1, "SAME"),
                                   (16, 256, 256, 64, 64, 3, 1, "SAME"), (4, 32, 32, 512, 2560, 3, 1, "SYMMETRIC")])
def test_conv_full_size_vs_naive_kernel(dev, shape):
    """BASELINE sizes: MFMA kernel vs the one-thread-per-output fmaf kernel on device (the CPU oracle is too slow here)
    + linearity property conv(a*x1 + x2) == a*conv(x1) + conv(x2)."""
    K = pkg("kernels")
    N, H, W, C, Kc, R, dil, padding = shape
    gen = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn((N, H, W, C), generator=gen).to(dev)
    x2 = torch.randn((N, H, W, C), generator=gen).to(dev)
    w = (torch.randn((R, R, C, Kc), generator=gen) / np.sqrt(R * R * C)).to(dev)
    g = K.conv_geom(tuple(x.shape), tuple(w.shape), 1, dil, padding)
    y = K.conv2d_fwd(x, w, g)
    yn = K.conv2d_fwd(x, w, g, naive=True)
    assert _rel(y, yn) < 2e-5
    y2 = K.conv2d_fwd(x2, w, g)
    ylin = K.conv2d_fwd(x * 0.5 + x2, w, g)
    assert _rel(ylin, y * 0.5 + y2) < 2e-5


@pytest.mark.parametrize("shape", [
    (16, 32, 32, 512, 512, 3, 1, 1, "SAME"), (16, 32, 32, 512, 512, 3, 1, 2, "SAME"), (4, 32, 32, 512, 2560, 3, 1, 1, "SYMMETRIC"),
    (16, 256, 256, 3, 16, 3, 1, 1, "SAME"), (16, 256, 256, 16, 16, 3, 1, 1, "SAME"), (16, 260, 260, 40, 5, 5, 1, 1, "VALID"),
    (16, 128, 128, 16, 32, 3, 1, 1, "SAME"), (16, 128, 128, 32, 32, 3, 1, 1, "SAME"), (16, 64, 64, 32, 64, 3, 1, 1, "SAME"),
    (16, 256, 256, 40, 5, 5, 1, 1, "SYMMETRIC"), (16, 256, 256, 64, 64, 3, 2, 1, "SAME"),
    (16, 256, 256, 64, 64, 3, 1, 1, "SAME"), (16, 256, 256, 32, 64, 3, 1, 1, "SAME"),      # cls_1 (8192 tiles of the three-stage kernel)
    (5, 250, 250, 64, 64, 3, 1, 1, "SAME"),                                                   # ... ragged last tile, extents not powers of two
    (16, 128, 128, 128, 128, 5, 2, 1, "SAME"), (16, 64, 64, 256, 256, 3, 2, 1, "SAME"), (16, 16, 16, 512, 512, 5, 4, 1, "SAME"),
    (16, 4, 4, 512, 512, 3, 2, 1, "SYMMETRIC")])
def test_conv_full_size_adjoint_identities(dev, shape):
    """BASELINE sizes (B=16), size-independent property: conv is bilinear, so <conv(x,w), dy> = <x, dgrad(dy,w)> = <w, wgrad(x,dy)>.
    One identity ties the three kernels (incl. the stride-phase data gradients and the reduction splits) together on every layer
    shape of the hot path; inner products accumulated in float64 on the device."""
    K = pkg("kernels")
    N, H, W, C, Kc, R, stride, dil, padding = shape
    gen = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((N, H, W, C), generator=gen).to(dev)
    w = (torch.randn((R, R, C, Kc), generator=gen) / np.sqrt(R * R * C)).to(dev)
    g = K.conv_geom(tuple(x.shape), tuple(w.shape), stride, dil, padding)
    dy = torch.randn((N, g.OH, g.OW, Kc), generator=gen).to(dev)
    y = K.conv2d_fwd(x, w, g)
    dx = K.conv2d_dgrad(dy, w, g)
    dw = K.conv2d_wgrad(x, dy, g)
    assert dx.shape == x.shape and dw.shape == w.shape

    def dot(a, b):
        return float((a.double() * b.double()).sum())

    a, b, c = dot(y, dy), dot(x, dx), dot(w, dw)
    scale = float(y.double().norm() * dy.double().norm())       # |<y,dy>| <= |y||dy|: the natural error scale
    assert abs(a - b) < 2e-6 * scale and abs(a - c) < 2e-6 * scale, (a, b, c, scale)
    # the "add into" entry points at the same sizes (reduction splits accumulate in the summing kernel, un-split tiles in the epilogue)
    slot = torch.ones_like(dw)
    K.conv2d_wgrad(x, dy, g, into=slot)
    assert _rel(slot, dw + 1.0) < 2e-6
    res = torch.randn(tuple(dx.shape), generator=gen).to(dev)
    assert _rel(K.conv2d_dgrad(dy, w, g, residual=res), dx + res) < 2e-6
