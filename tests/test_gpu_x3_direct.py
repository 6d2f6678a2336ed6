"""-m gpu: the narrow 3x3 layers of the Winograd route as DIRECT split-bf16 convolutions (csrc/conv_x3_direct.hip, kernels.x3_direct):
forward and data gradient against the float64 convolution of the same float32 operands (oracle.tf_ops.conv2d / autograd), and every
epilogue feature (dropout mask stream, residual add, BN statistics partials -> mean / variance) against the direct fp32-MFMA kernels.
Each case checks which kernel symbols ran: a silent hand-back to the Winograd kernels cannot pass.

Bar: 5e-6 of max|ref| (measured 6e-7 .. 9e-7: every product is exact, one fp32 accumulation chain of 9 C terms; the direct fp32 kernel
lands at 1e-6 .. 3e-6, the F(4x4) routes at 1e-6 .. 8e-6 on such layers)."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

# (N, H, W, C, K, padding): output extents are multiples of 16, 32 / 64 input channels, 32 / 64 / 128 filters
CASES = [
    (2, 64, 64, 64, 64, "SAME"),          # critic cls_1's 64 -> 64 (at 256^2 in the model)
    (3, 32, 48, 64, 128, "SAME"),         # cls_2's 64 -> 128: two filter blocks per tile; its data gradient is 128 -> 64 (not taken: 128 input channels)
    (2, 32, 32, 32, 64, "SAME"),          # one channel half per tile (the patch buffer alternates with the item)
    (1, 34, 50, 64, 64, "VALID"),         # a mirror-padded input run as VALID (padding 0): the forward is taken (32 x 48 outputs), its data
                                          # gradient (34 x 50 outputs) is not
    (1, 32, 48, 64, 64, "VALID"),         # ... and the other way round: a data gradient with padding 2
    (5, 16, 16, 32, 128, "SAME"),         # fewer items than workgroups on most of the chip: 10 items
    (2, 32, 48, 64, 32, "SAME"),          # 32 filters: 32-filter blocks (the data gradient of a 32 -> 64 layer is this shape)
]
BAR = 5e-6


def _rel(a, b):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def _ran(L, fn, cls):
    L.prof_summary()
    L.prof_enable(cls)
    out = fn()
    torch.cuda.synchronize()
    L.prof_enable(0)
    return out, sorted(r["name"] for r in L.prof_summary())


@pytest.fixture
def route():
    """the Winograd planner takes every layer it can (mode 2), so that the narrow test layers are the route's; restores what was in force"""
    K = pkg("kernels")
    prev = (K.wino_mode(2), K.wino_tile(4), K.x3_direct(-1))
    yield K
    K.wino_mode(prev[0]); K.wino_tile(prev[1]); K.x3_direct(prev[2])


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_direct_split_bf16_fwd_dgrad_vs_float64(dev, route, case):
    K, L = route, pkg("_lib")
    N, H, W, C, Kf, padding = case
    rng = np.random.default_rng(sum(case[:5]))
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    w = (rng.standard_normal((3, 3, C, Kf)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    g = K.conv_geom(x.shape, w.shape, 1, 1, padding)
    dy = rng.standard_normal((N, g.OH, g.OW, Kf)).astype(np.float32)
    res = rng.standard_normal(x.shape).astype(np.float32)
    xd, wd, dyd, resd = (torch.from_numpy(a).to(dev) for a in (x, w, dy, res))
    xg = torch.from_numpy(x).double().requires_grad_(True)
    yo = T.conv2d(xg, torch.from_numpy(w).double(), 1, 1, padding)
    yo.backward(torch.from_numpy(dy).double())
    assert K.wino_chosen(g, 0) == 4

    K.x3_direct(0)
    y0, names0 = _ran(L, lambda: K.conv2d_fwd(xd, wd, g), L.PROF_CONV_FWD)
    assert not any("x3_direct" in n or "x3d" in n for n in names0), names0
    dx0 = K.conv2d_dgrad(dyd, wd, g)
    K.x3_direct(2)          # (2: wherever the shapes allow — mode 1 leaves launches that cannot fill the chip to the old route)
    y1, names1 = _ran(L, lambda: K.conv2d_fwd(xd, wd, g), L.PROF_CONV_FWD)
    taken_f = g.OH % 16 == 0 and g.OW % 16 == 0
    if taken_f:
        assert names1 == sorted(["x3d_filter_kernel<false>", "conv_x3_direct_kernel<%d, %d, 0>" % (C // 32, 32 if Kf == 32 else 64)]), names1
    else:
        assert not any("x3_direct" in n for n in names1), names1
    dx1, names2 = _ran(L, lambda: K.conv2d_dgrad(dyd, wd, g), L.PROF_CONV_DGRAD)
    # (the data gradient is a convolution of dy: Kf input channels, C filters, H x W outputs — taken when THOSE fit)
    taken_d = Kf in (32, 64) and C in (32, 64, 128) and H % 16 == 0 and W % 16 == 0
    if taken_d:
        assert names2 == sorted(["x3d_filter_kernel<true>", "conv_x3_direct_kernel<%d, %d, 1>" % (Kf // 32, 32 if C == 32 else 64)]), names2
    else:
        assert not any("x3_direct" in n for n in names2), names2
    dxr = K.conv2d_dgrad(dyd, wd, g, residual=resd)
    errs = {"y": _rel(y1, yo), "dx": _rel(dx1, xg.grad), "dx+res": _rel(dxr, xg.grad + torch.from_numpy(res).double()),
            "y vs winograd": _rel(y1, y0), "dx vs winograd": _rel(dx1, dx0)}
    print("x3 direct %s: %s (data gradient on the route: %s)" % (case, {k: "%.2e" % v for k, v in errs.items()}, taken_d))
    assert (not taken_f or errs["y"] < BAR) and (not taken_d or (errs["dx"] < BAR and errs["dx+res"] < BAR)), errs
    assert taken_f or taken_d
    assert errs["y vs winograd"] < 3e-5 and errs["dx vs winograd"] < 3e-5, errs


def test_direct_split_bf16_epilogues_equal_the_direct_kernels(dev, route):
    """dropout (the same counter hash on the flat output index: identical zero pattern), BN statistics partials -> mean / variance / moving
    averages, the residual add of a data gradient: against conv_epilogue of the fp32-MFMA kernels (PNP_WINOGRAD off)"""
    K, L = route, pkg("_lib")
    rng = np.random.default_rng(11)
    for (N, H, C, Kf) in ((2, 64, 64, 64), (2, 32, 32, 128), (3, 16, 64, 128)):
        x = torch.from_numpy(rng.standard_normal((N, H, H, C)).astype(np.float32)).to(dev)
        w = torch.from_numpy((rng.standard_normal((3, 3, C, Kf)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)).to(dev)
        g = K.conv_geom(tuple(x.shape), tuple(w.shape), 1, 1, "SAME")
        shift = torch.from_numpy((rng.standard_normal(Kf) * 0.1).astype(np.float32)).to(dev)
        out = {}
        for which in ("direct", "x3"):
            K.wino_mode(0 if which == "direct" else 2)
            K.x3_direct(0 if which == "direct" else 2)
            yd, names = _ran(L, lambda: K.conv2d_fwd(x, w, g, keep_prob=0.75, seed=99, stream_id=3), L.PROF_CONV_FWD)
            assert any("x3_direct" in n for n in names) == (which == "x3"), names
            mm, mv = torch.zeros(Kf, device=dev), torch.ones(Kf, device=dev)
            nparts = K.conv_stats_parts(g)
            if nparts > 0:
                ys, parts = K.conv2d_fwd_stats(x, w, g, shift, keep_prob=0.75, seed=99, stream_id=3)
                mean, var = K.bn_stats_finish(parts, shift, N * H * H, mm, mv)
            else:
                ys = yd
                mean, var = K.bn_stats(yd)
            out[which] = (yd, ys, mean, var, nparts)
        (yd0, ys0, m0, v0, _), (yd1, ys1, m1, v1, np1) = out["direct"], out["x3"]
        assert np1 == N * H * H // 64                                                    # one partial per consumer wave (64 pixels)
        assert torch.equal(yd0 == 0, yd1 == 0), "dropout masks differ"
        assert 0.2 < float((yd1 == 0).float().mean()) < 0.3
        assert torch.equal(ys1, yd1)                                                    # the statistics launch leaves the same output
        yd64 = yd1.double().reshape(-1, Kf)
        errs = {"drop": _rel(yd1, yd0), "mean": _rel(m1, m0), "var": _rel(v1, v0),
                "mean vs f64": float((m1.double() - yd64.mean(0)).abs().max() / yd64.std()), "var vs f64": _rel(v1, yd64.var(0, unbiased=False))}
        print("x3 direct epilogues (%d, %d, %d->%d): %s" % (N, H, C, Kf, {k: "%.2e" % v for k, v in errs.items()}))
        assert max(errs.values()) < 1e-5, errs


def test_direct_split_bf16_planner(dev, route):
    """mode 1 takes a layer only when its launch fills the chip (>= 256 (tile, filter block) items); layers the Winograd planner leaves
    alone (32 input channels at the default policy) are taken from the direct kernels' entry; bf16 geometries never"""
    K, L = route, pkg("_lib")
    K.wino_mode(1)
    rng = np.random.default_rng(3)
    for (N, H, C, Kf, want) in ((16, 64, 32, 64, True), (2, 32, 32, 64, False), (4, 128, 32, 64, True)):
        x = torch.from_numpy(rng.standard_normal((N, H, H, C)).astype(np.float32)).to(dev)
        w = torch.from_numpy((rng.standard_normal((3, 3, C, Kf)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)).to(dev)
        g = K.conv_geom(tuple(x.shape), tuple(w.shape), 1, 1, "SAME")
        assert K.wino_chosen(g, 0) == 0                      # (the default policy does not route 32-channel layers)
        K.x3_direct(0)
        y0 = K.conv2d_fwd(x, w, g)
        K.x3_direct(1)
        y1, names = _ran(L, lambda: K.conv2d_fwd(x, w, g), L.PROF_CONV_FWD)
        assert any("x3_direct" in n for n in names) == want, (N, H, names)
        assert _rel(y1, y0) < 1e-5
        gb = K.conv_geom(tuple(x.shape), tuple(w.shape), 1, 1, "SAME", dtype=L.DTYPE_BF16)
        _, names_b = _ran(L, lambda: K.conv2d_fwd(x, w, gb), L.PROF_CONV_FWD)
        assert not any("x3_direct" in n for n in names_b), names_b
