"""-m gpu: the bf16-RESIDENT convolutions (csrc/conv_bf16r.hip; BASELINE configs[4]) — operands stored as bf16 in HBM, staged by LDS-DMA,
filter gradient fragments by ds_read_b64_tr_b16.

Arithmetic under test is the one tests/test_gpu_bf16.py states for the staged-rounding kernels: both operands of every product rounded
to bfloat16 (nearest-even), products summed in float32 — so the oracle is a float64 convolution of the ROUNDED operands and the bar is
float32-summation-order tight (2e-5 of max|ref|), not bf16-loose.  The bf16 side outputs (`yh`, `dxh`) are held to the float32 output
of the same launch rounded once (bit-exact)."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

# (N, H, W, C, K, k, stride, dil, padding) — every tile / row-width / filter-shape class of conv_bf16r.hip
CASES = [
    (4, 32, 32, 512, 512, 3, 1, 1, "SAME"),       # group_7..9 (128x64 tiles at this batch; 256x128 at B=16)
    (4, 32, 32, 512, 512, 3, 1, 2, "SAME"),       # group_8 dilated
    (4, 34, 34, 512, 2560, 3, 1, 1, "VALID"),     # group_10 after the SYMMETRIC pre-pad
    (2, 64, 64, 64, 64, 3, 1, 1, "SAME"),         # group_3: 64-channel rows (128-byte LDS rows in the filter gradient)
    (2, 64, 64, 64, 128, 3, 1, 1, "SAME"),        # 64 x 128 filter-gradient tile
    (2, 64, 64, 128, 64, 3, 1, 1, "SAME"),        # 128 x 64 filter-gradient tile
    (2, 128, 128, 32, 64, 3, 1, 1, "SAME"),       # C = 32: 64-byte rows (BKC = 32), forward only
    (4, 128, 128, 64, 64, 3, 2, 1, "SAME"),       # critic cls_1_3: stride 2 forward, stride-phase data gradient, strided filter gradient
    (4, 64, 64, 128, 128, 5, 2, 1, "SAME"),       # critic cls_2_3: 5x5 stride 2
    (3, 37, 41, 96, 128, 3, 1, 1, "SAME"),        # ragged M, C = 96 (three 32-groups), non-power-of-two map: forward only
]


def _rel(a, b):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def _ran(L, fn, cls):
    L.prof_summary()
    L.prof_enable(cls)
    out = fn()
    torch.cuda.synchronize()
    L.prof_enable(0)
    return out, [r["name"] for r in L.prof_summary()]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_resident_fwd_dgrad_wgrad_vs_rounded_oracle(dev, case):
    K, L = pkg("kernels"), pkg("_lib")
    N, H, W, C, Kf, k, stride, dil, padding = case
    rng = np.random.default_rng(sum(case[:7]))
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    w = (rng.standard_normal((k, k, C, Kf)) * np.sqrt(2.0 / (k * k * C))).astype(np.float32)
    g = K.conv_geom(x.shape, w.shape, stride, dil, padding, dtype=L.DTYPE_BF16)
    dy = rng.standard_normal((N, g.OH, g.OW, Kf)).astype(np.float32)
    xd, wd, dyd = (torch.from_numpy(a).to(dev) for a in (x, w, dy))
    xh, dyh = K.cast_bf16(xd), K.cast_bf16(dyd)
    w_io, w_oi = K.filter_bf16(wd)
    # the casts themselves: round-to-nearest-even, and the two filter layouts
    rb = lambda a: T.round_bf16(torch.from_numpy(a))
    assert torch.equal(xh.float().cpu(), rb(x)) and torch.equal(dyh.float().cpu(), rb(dy))
    assert torch.equal(w_io.float().cpu().reshape(k, k, C, Kf), rb(w))
    assert torch.equal(w_oi.float().cpu().reshape(k, k, Kf, C), rb(w).permute(0, 1, 3, 2))
    xr, wr, dyr = rb(x).double(), rb(w).double(), rb(dy).double()
    assert K.bf16r_served(g, 0)
    (y, yh, _), names = _ran(L, lambda: K.conv2d_fwd_bf16r(xh, w_oi, g, want_h=True), L.PROF_CONV_FWD)
    assert names and all("conv_bf16r_kernel" in n for n in names), names
    yo = T.conv2d(xr, wr, stride, dil, padding)
    errs = {"y": _rel(y, yo)}
    assert torch.equal(yh.float(), y.bfloat16().float())          # the bf16 copy is the float32 output rounded once (torch cast: test only)
    if K.bf16r_served(g, 1):
        xg = xr.clone().requires_grad_(True)
        T.conv2d(xg, wr, stride, dil, padding).backward(dyr)
        res = torch.from_numpy(rng.standard_normal(x.shape).astype(np.float32)).to(dev)
        # (a strided data gradient is one resident launch per stride phase, its rows scattered: no bf16 copy, no residual)
        (dx, dxh), names = _ran(L, lambda: K.conv2d_dgrad_bf16r(dyh, w_io, g, want_h=stride == 1), L.PROF_CONV_DGRAD)
        assert names and all("conv_bf16r_kernel" in n for n in names), names
        assert len(names) == (1 if stride == 1 else stride * stride)
        errs["dx"] = _rel(dx, xg.grad)
        if stride == 1:
            assert torch.equal(dxh.float(), dx.bfloat16().float())
            dx2, _ = K.conv2d_dgrad_bf16r(dyh, w_io, g, residual=res)        # + the gradient arriving over a residual shortcut
            errs["dx+res"] = _rel(dx2, xg.grad + res.cpu().double())
    else:
        assert C % 64 != 0
    if K.bf16r_served(g, 2):
        wg = wr.clone().requires_grad_(True)
        T.conv2d(xr, wg, stride, dil, padding).backward(dyr)
        dw, names = _ran(L, lambda: K.conv2d_wgrad_bf16r(xh, dyh, g), L.PROF_CONV_WGRAD)
        assert names and all("conv_wgrad_bf16r_kernel" in n for n in names), names
        errs["dw"] = _rel(dw, wg.grad)
        acc = torch.from_numpy(rng.standard_normal(w.shape).astype(np.float32)).to(dev)
        acc0 = acc.cpu().double()
        K.conv2d_wgrad_bf16r(xh, dyh, g, into=acc)                        # "add into" (a slot of the gradient arena)
        errs["dw+="] = _rel(acc, acc0 + wg.grad)
    else:
        assert C % 64 != 0 or (g.OW & (g.OW - 1)) != 0
    print("resident bf16 conv %s: %s" % (case, {k_: "%.2e" % e for k_, e in errs.items()}))
    assert all(e < 2e-5 for e in errs.values()), errs


def test_resident_forward_epilogues_statistics_dropout_fused_bn(dev):
    """the resident forward shares conv_epilogue with the fp32 kernels: dropout mask stream, BN-statistics partials (its OWN tile
    geometry: pnp_conv2d_fwd_bf16r_stats_parts) and the fused inference BN + shortcut + leaky-ReLU — held to the separate kernels"""
    K, L = pkg("kernels"), pkg("_lib")
    rng = np.random.default_rng(5)
    N, H, C, Kf = 4, 32, 128, 256
    x = torch.from_numpy(rng.standard_normal((N, H, H, C)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal((3, 3, C, Kf)) * 0.03).astype(np.float32)).to(dev)
    g = K.conv_geom(tuple(x.shape), tuple(w.shape), 1, 1, "SAME", dtype=L.DTYPE_BF16)
    xh = K.cast_bf16(x)
    w_oi = K.filter_bf16(w)[1]
    y0 = K.conv2d_fwd_bf16r(xh, w_oi, g)[0]
    # dropout: the counter-hash stream of pnp_dropout on the flat output index
    yd = K.conv2d_fwd_bf16r(xh, w_oi, g, keep_prob=0.75, seed=7, stream_id=3)[0]
    assert torch.equal(yd, K.dropout(y0, 0.75, 7, 3))
    # statistics partials -> (mean, var) == a reduction pass over the output
    shift = torch.from_numpy((0.1 * rng.standard_normal(Kf)).astype(np.float32)).to(dev)
    y1, _, parts = K.conv2d_fwd_bf16r(xh, w_oi, g, keep_prob=0.75, seed=7, stream_id=3, stat_shift=shift, want_stats=True)
    assert torch.equal(y1, yd) and parts[1] > 0
    mean, var = K.bn_stats_finish(parts, shift, N * H * H)
    m2, v2 = K.bn_stats(yd)
    assert _rel(mean, m2) < 1e-5 and _rel(var, v2) < 1e-5
    # fused inference BN + zero-padded shortcut + leaky-ReLU, with the bf16 copy of the RESULT
    gamma, beta = (torch.from_numpy(a.astype(np.float32)).to(dev) for a in (1 + 0.1 * rng.standard_normal(Kf), 0.1 * rng.standard_normal(Kf)))
    sc = torch.from_numpy(rng.standard_normal((N, H, H, C)).astype(np.float32)).to(dev)
    ss = K.bn_fold(gamma, beta, mean, var, 1e-3)
    out, outh, _ = K.conv2d_fwd_bf16r(xh, w_oi, g, want_h=True, bn=(ss, sc, 0.2))
    ref = K.bn_apply(y0, mean, var, gamma, beta, sc, 1e-3, 0.2)
    assert _rel(out, ref) < 2e-6
    assert torch.equal(outh.float(), out.bfloat16().float())


def test_bf16_side_outputs_of_the_elementwise_producers(dev):
    """pnp_bn_apply_h / pnp_bn_bwd_acc_h / pnp_bn_bwd_apply_h / pnp_dropout_h: float32 results unchanged bit for bit, the bf16 copy is
    that result rounded once"""
    K = pkg("kernels")
    rng = np.random.default_rng(9)
    P, C = 4 * 32 * 32, 128
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev)
    x, dout, sc = t(P, C), t(P, C), t(P, 64)
    mean, var = K.bn_stats(x)
    gamma, beta = 1 + 0.1 * t(C), 0.1 * t(C)
    y0 = K.bn_apply(x, mean, var, gamma, beta, sc, 1e-3, 0.2)
    y1 = K.bn_apply(x, mean, var, gamma, beta, sc, 1e-3, 0.2, want_h=True)
    assert torch.equal(y0, y1) and torch.equal(K.bf16_of(y1).float(), y0.bfloat16().float()) and getattr(y0, "_pnp_h", None) is None
    for training in (True, False):
        a = K.bn_bwd(dout, y0, x, mean, var, gamma, 64, 1e-3, 0.2, training, 0.75, 3, 1)
        b = K.bn_bwd(dout, y0, x, mean, var, gamma, 64, 1e-3, 0.2, training, 0.75, 3, 1, want_h=True)
        assert all(torch.equal(u, v) for u, v in zip(a, b))
        assert torch.equal(b[0]._pnp_h[0].float(), a[0].bfloat16().float())
    d0 = K.dropout(x, 0.75, 5, 2)
    d1 = K.dropout(x, 0.75, 5, 2, want_h=True)
    assert torch.equal(d0, d1) and torch.equal(d1._pnp_h[0].float(), d0.bfloat16().float())
    # on-demand cast: counted, cached on the tensor object, invalidated by an in-place write
    n0 = K.CAST_COUNT[0]
    h = K.bf16_of(x)
    assert K.bf16_of(x) is h and K.CAST_COUNT[0] == n0 + 1
    x.mul_(2.0)
    assert K.bf16_of(x) is not h and K.CAST_COUNT[0] == n0 + 2


def test_training_graph_runs_on_the_resident_kernels(dev):
    """set_conv_dtype('bf16'): a residual block's forward + backward through the autograd glue launches conv_bf16r / conv_wgrad_bf16r
    symbols (observed, pnp_prof_*), the bf16 copies come from the producers (no on-demand cast after the block input), filter shadows
    and the result is the staged-rounding path's (same arithmetic: operands rounded once, float32 sums).  The two evaluations differ in
    float32 summation order, which flips a few bf16 roundings / ReLU signs downstream, so whole tensors are compared by direction
    (1 - cosine; measured 6e-8 / 1e-4 / 1e-4 on MI355X) — the per-kernel bars are the 2e-5 of the tests above."""
    K, L, F, layers, variables = pkg("kernels"), pkg("_lib"), pkg("functional"), pkg("layers"), pkg("variables")
    rng = np.random.default_rng(2)
    x0 = torch.from_numpy(rng.standard_normal((4, 32, 32, 128)).astype(np.float32)).to(dev)

    def run(resident):
        F.set_conv_dtype("bf16")
        old = K.bf16r
        if not resident:
            K.bf16r = lambda g, kind: False
        try:
            store = variables.VariableStore(dev, seed=4)
            with store.as_default():
                def graph(xin):
                    store.begin_trace(drop_seed=11)
                    with store.name_scope("group_a"):
                        w = [layers.weight_variable(s_, stddev=0.03) for s_ in ([3, 3, 128, 128], [3, 3, 128, 128], [3, 3, 128, 256],
                                                                               [3, 3, 256, 256])]
                        h = layers.residual_block(xin, w[0], w[1], 0.75, is_train=True)
                        return layers.residual_block(h, w[2], w[3], 0.75, inc_dim=True, is_train=True)
                graph(x0)
                store.finalize()
                xin = x0.clone().requires_grad_(True)
                K.CAST_COUNT[0] = 0
                L.prof_summary()
                L.prof_enable(L.PROF_CONV_FWD | L.PROF_CONV_DGRAD | L.PROF_CONV_WGRAD)
                store.zero_grad()
                out = graph(xin)
                out.backward(torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(out.shape)).astype(np.float32)).to(dev))
                torch.cuda.synchronize()
                L.prof_enable(0)
                names = [r["name"] for r in L.prof_summary()]
                return out.detach().cpu(), xin.grad.cpu(), store.grad_arena.detach().cpu().clone(), names, K.CAST_COUNT[0], store
        finally:
            K.bf16r = old
            F.set_conv_dtype("f32")
    o1, dx1, ga1, names1, casts1, store = run(True)
    o0, dx0, ga0, names0, casts0, _ = run(False)
    assert any("conv_bf16r_kernel" in n for n in names1) and any("conv_wgrad_bf16r_kernel" in n for n in names1), names1
    assert not any("bf16r" in n for n in names0) and all("bf16" in n for n in names0), names0
    assert not any("conv_taps_bf16_kernel" in n or "conv_wgrad_bf16_kernel" in n for n in names1), names1
    # one cast for the block input (produced outside the graph); every other bf16 operand came from its producer's side output
    assert casts1 <= 1 and casts0 == 0, (casts1, casts0)
    ncos = lambda a, b: 1.0 - float((a.double().reshape(-1) * b.double().reshape(-1)).sum() / (a.double().norm() * b.double().norm()))
    d = (ncos(o1, o0), ncos(dx1, dx0), ncos(ga1, ga0))
    print("resident vs staged-rounding path (1 - cosine): out %.2e dx %.2e parameter gradients %.2e; casts %d; symbols %s" % (
        d + (casts1, sorted(set(names1)))))
    assert d[0] < 1e-5 and d[1] < 2e-3 and d[2] < 2e-3, d
