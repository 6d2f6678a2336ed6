"""-m gpu: batch-norm (+residual/leaky-ReLU/dropout-mask), max-pool, PS, sympad backward, critic input — C-ABI vs oracle."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("shape,Cs,alpha", [((2, 16, 16, 16), 0, 0.2), ((2, 16, 16, 32), 16, 0.2), ((2, 8, 8, 64), 64, 0.2),
                                            ((3, 5, 7, 512), 256, 0.2), ((2, 8, 8, 128), 0, -1.0), ((2, 6, 6, 40), 0, 0.2),
                                            ((2, 4, 4, 6), 0, 0.0)])
@pytest.mark.parametrize("training", [True, False])
def test_bn_unit_fwd_bwd(dev, shape, Cs, alpha, training):
    K = pkg("kernels")
    rng = np.random.default_rng(0)
    C = shape[-1]
    x = (rng.standard_normal(shape) * 1.7 + 0.6).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(C)).astype(np.float32)
    mm = (0.3 * rng.standard_normal(C)).astype(np.float32)
    mv = (1 + 0.2 * rng.random(C)).astype(np.float32)
    sc = rng.standard_normal(shape[:-1] + (Cs,)).astype(np.float32) if Cs else None
    dout = rng.standard_normal(shape).astype(np.float32)
    keep, seed, sid = 0.75, 99, 3

    # oracle: xc = dropout(xa) where xa is the conv accumulator; we test the BN unit given xc, and the masked dxa
    xa = torch.from_numpy(x).requires_grad_(True)
    mask = torch.from_numpy(T.dropout_mask(shape, keep, seed, sid))
    xc = xa * mask * np.float32(1.0 / np.float32(keep))
    g_t = torch.from_numpy(gamma).requires_grad_(True)
    b_t = torch.from_numpy(beta).requires_grad_(True)
    mm_t, mv_t = torch.from_numpy(mm.copy()), torch.from_numpy(mv.copy())
    z = T.batch_norm(xc, g_t, b_t, mm_t, mv_t, training)
    sc_t = None
    if sc is not None:
        sc_t = torch.from_numpy(sc).requires_grad_(True)
        z = z + (T.pad_channels(sc_t, (C - Cs) // 2) if Cs != C else sc_t)
    out_o = T.leaky_relu(z, alpha) if alpha >= 0 else z
    out_o.backward(torch.from_numpy(dout))

    xcd = xc.detach().to(dev).contiguous()
    gd, bd = torch.from_numpy(gamma).to(dev), torch.from_numpy(beta).to(dev)
    mmd, mvd = torch.from_numpy(mm.copy()).to(dev), torch.from_numpy(mv.copy()).to(dev)
    P = xcd.numel() // C
    if training:
        mean, var = K.bn_stats(xcd)
        K.bn_update_moving(mmd, mvd, mean, var, P, 0.9)
        assert _rel(mmd, mm_t) < 1e-5 and _rel(mvd, mv_t) < 1e-5
    else:
        mean, var = mmd.clone(), mvd.clone()
    scd = torch.from_numpy(sc).to(dev) if sc is not None else None
    out = K.bn_apply(xcd, mean, var, gd, bd, scd, 1e-3, alpha)
    assert _rel(out, out_o) < 2e-5
    dxa, dgamma, dbeta, dsc = K.bn_bwd(torch.from_numpy(dout).to(dev), out, xcd, mean, var, gd, Cs, 1e-3, alpha, training, keep,
                                       seed, sid)
    assert _rel(dxa, xa.grad) < 1e-4
    assert _rel(dgamma, g_t.grad) < 1e-4
    assert _rel(dbeta, b_t.grad) < 1e-4
    if sc is not None:
        assert _rel(dsc, sc_t.grad) < 1e-5
    if sc is None and alpha >= 0:
        # `out` not handed in: the kernels recompute the activation's sign from the BN input (what functional.ConvBNActFn does for
        # units without a shortcut) — bit for bit the backward that reads the saved output, scalar (C % 4 != 0) and vector kernels alike
        dxr, dgr, dbr, _ = K.bn_bwd(torch.from_numpy(dout).to(dev), None, xcd, mean, var, gd, 0, 1e-3, alpha, training, keep, seed, sid,
                                    beta=bd)
        assert torch.equal(dxr, dxa) and torch.equal(dgr, dgamma) and torch.equal(dbr, dbeta)
        with pytest.raises(pkg("_lib").PnpError):
            K.bn_bwd(torch.from_numpy(dout).to(dev), None, xcd, mean, var, gd, 0, 1e-3, alpha, training, keep, seed, sid)      # no beta
    # pnp_bn_bwd_acc: the same sums also added into caller-owned slots, twice (a BN layer shared by two passes)
    sg, sb = torch.full((C,), 2.0, device=dev), torch.full((C,), -1.0, device=dev)
    for rep in (1, 2):
        dxa2, dg2, db2, _ = K.bn_bwd(torch.from_numpy(dout).to(dev), out, xcd, mean, var, gd, Cs, 1e-3, alpha, training, keep, seed, sid,
                                     into=(sg, sb))
        assert torch.equal(dxa2, dxa) and torch.equal(dg2, dgamma) and torch.equal(db2, dbeta)
        assert _rel(sg, 2.0 + rep * dgamma) < 1e-6 and _rel(sb, -1.0 + rep * dbeta) < 1e-6


def test_bn_stats_large_mean_is_stable(dev):
    K = pkg("kernels")
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((4, 64, 64, 32)) * 0.01 + 100.0).astype(np.float32)
    mean, var = K.bn_stats(torch.from_numpy(x).to(dev))
    x64 = x.astype(np.float64).reshape(-1, 32)
    assert np.abs(mean.cpu().numpy() - x64.mean(0)).max() < 1e-4
    assert np.abs(var.cpu().numpy() - x64.var(0)).max() < 1e-2 * x64.var(0).max()


@pytest.mark.parametrize("P,C", [(16384, 64), (65536, 64), (300000, 32), (262144, 5), (1048576, 16), (40000, 512)],
                         ids=lambda v: str(v))
def test_bn_reductions_with_long_partial_lists(dev, P, C):
    """pnp_bn_stats / pnp_bn_bwd_reduce at the row counts of the 64^2 ... 256^2 layers: 512+ workgroup partials are compacted in
    place (double sums as float high/low pairs) before the final combine; both against float64 sums of the same tensors"""
    K = pkg("kernels")
    rng = np.random.default_rng(P % 1000 + C)
    x = torch.from_numpy((rng.standard_normal((P, C)) * 1.3 + 0.4).astype(np.float32)).to(dev)
    dout = torch.from_numpy(rng.standard_normal((P, C)).astype(np.float32)).to(dev)
    x4, d4 = x.view(1, 1, P, C), dout.view(1, 1, P, C)
    mean, var = K.bn_stats(x4)
    x64 = x.double()
    m64, v64 = x64.mean(0), x64.var(0, unbiased=False)
    assert float((mean.double() - m64).abs().max()) < 1e-6 and float((var.double() - v64).abs().max()) < 1e-5
    gamma = torch.ones(C, device=dev)
    out = K.bn_apply(x4, mean, var, gamma, torch.zeros(C, device=dev), None, 1e-3, 0.2)
    dx, dg, db, _ = K.bn_bwd(d4, out, x4, mean, var, gamma, 0, 1e-3, 0.2, True, 1.0, 0, 0)
    g = torch.where(out.view(P, C) > 0, dout, dout * 0.2).double()
    xh = (x64 - mean.double()) / torch.sqrt(var.double() + 1e-3)
    db64, dg64 = g.sum(0), (g * xh).sum(0)
    assert float((db.double() - db64).abs().max() / db64.abs().max()) < 2e-6
    assert float((dg.double() - dg64).abs().max() / dg64.abs().max()) < 2e-6
    assert bool(torch.isfinite(dx).all())


def test_one_launch_combine_rearms_its_tickets(dev):
    """round 5 (opt-in, PNP_BN_ONE_LAUNCH=1 — off by default: slower, profiles/r05_bn_one_launch_ab.txt): the column reductions sum their
    partial rows inside the reduction launch (last workgroup of a slab, then last slab; tickets in a library-owned counter buffer that
    every launch leaves zeroed).  In a process of its own with the switch on (the library reads it once): 640 launches of mixed shapes —
    more than the buffer has slots — must each finish (outputs pre-poisoned with NaN) with bit-identical results per shape and the
    float64 statistics: a counter left armed would make a later launch on its slot finish early (a wrong sum) or never (NaN)"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PNP_BN_ONE_LAUNCH="1")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "bn_ticket_worker.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "TICKETS OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_bn_statistics_of_an_overflowing_tensor_stay_infinite_not_nan(dev):
    """a partial sum beyond float range must come out as inf (like the single-launch combine), never as inf - inf = NaN"""
    K = pkg("kernels")
    x = torch.full((1, 1, 65536, 32), 1e25, device=dev)
    x[0, 0, 0, :] = 0.0                 # the shift (first row): every other row contributes d = 1e25, d*d = inf in float32
    mean, var = K.bn_stats(x)
    assert bool(torch.isfinite(mean).all()) and abs(float(mean[0]) / 1e25 - 1.0) < 1e-3
    assert bool(torch.isinf(var).all()) and not bool(torch.isnan(var).any())


@pytest.mark.parametrize("shape", [(2, 16, 16, 16), (2, 8, 12, 32), (1, 4, 4, 6)])
def test_maxpool(dev, shape):
    K = pkg("kernels")
    rng = np.random.default_rng(0)
    x = rng.integers(-3, 4, size=shape).astype(np.float32)    # many ties: exercises first-max routing
    dy = rng.standard_normal((shape[0], shape[1] // 2, shape[2] // 2, shape[3])).astype(np.float32)
    xt = torch.from_numpy(x).requires_grad_(True)
    yo = T.max_pool2(xt)
    yo.backward(torch.from_numpy(dy))
    xd = torch.from_numpy(x).to(dev)
    assert torch.equal(K.maxpool2_fwd(xd).cpu(), yo.detach())
    assert torch.equal(K.maxpool2_bwd(xd, torch.from_numpy(dy).to(dev)).cpu(), xt.grad)


@pytest.mark.parametrize("N,A,B,r,nc", [(2, 4, 4, 8, 40), (3, 2, 5, 8, 2), (2, 3, 3, 2, 4), (2, 4, 32, 8, 4), (1, 2, 6, 8, 8), (1, 2, 2, 8, 72),
                                        (16, 32, 32, 8, 40)])
def test_ps(dev, N, A, B, r, nc):
    K = pkg("kernels")
    x = np.arange(N * A * B * nc * r * r, dtype=np.float32).reshape(N, A, B, nc * r * r)
    y = K.ps_fwd(torch.from_numpy(x).to(dev), r, nc).cpu()
    yo = T.PS(torch.from_numpy(x), r, nc)
    assert torch.equal(y, yo)    # pure data movement: bit exact
    back = K.ps_bwd(y.to(dev), r, nc).cpu()
    assert torch.equal(back, torch.from_numpy(x))


def test_sympad_bwd(dev):
    K = pkg("kernels")
    rng = np.random.default_rng(0)
    for (N, H, W, C, p) in [(2, 6, 5, 8, 1), (1, 7, 7, 4, 2), (1, 2, 2, 3, 1)]:
        x = torch.from_numpy(rng.standard_normal((N, H, W, C)).astype(np.float32)).requires_grad_(True)
        xp = T.pad_symmetric(x, p, p)
        assert torch.equal(K.sympad_fwd(x.detach().to(dev), p).cpu(), xp.detach())      # forward: pure data movement, bit exact
        g = rng.standard_normal(tuple(xp.shape)).astype(np.float32)
        xp.backward(torch.from_numpy(g))
        dx = K.sympad_bwd(torch.from_numpy(g).to(dev), p).cpu()
        assert _rel(dx, x.grad) < 1e-6


def test_critic_input(dev):
    K = pkg("kernels")
    rng = np.random.default_rng(0)
    N, H, W = 2, 8, 8
    a = rng.standard_normal((N, H, W, 2)).astype(np.float32)
    b = rng.standard_normal((N, H, W, 4)).astype(np.float32)
    c = rng.standard_normal((N, H, W, 8)).astype(np.float32)
    d = rng.standard_normal((N, H, W, 8)).astype(np.float32)
    lg = rng.standard_normal((N, H, W, 5)).astype(np.float32)
    ts = [torch.from_numpy(t).requires_grad_(True) for t in (a, b, c, d, lg)]
    am = torch.argmax(ts[4].detach(), dim=-1, keepdim=True).float()
    ref = torch.cat([ts[0].repeat(1, 1, 1, 3), ts[1], ts[2], ts[3], ts[4], am], dim=3)
    out = K.critic_input_fwd(*(torch.from_numpy(t).to(dev) for t in (a,)), 3, *(torch.from_numpy(t).to(dev) for t in (b, c, d, lg)))
    assert torch.equal(out.cpu(), ref.detach())
    g = rng.standard_normal(tuple(ref.shape)).astype(np.float32)
    ref.backward(torch.from_numpy(g))
    grads = K.critic_input_bwd(torch.from_numpy(g).to(dev), tuple(tuple(t.shape) for t in ts), 3)
    for got, t in zip(grads, ts):
        assert _rel(got, t.grad) < 1e-6


@pytest.mark.parametrize("case", [(2, 16, 16, 32, 64, 3, 1, 1, True, 0.75, 0.2), (2, 16, 16, 64, 64, 3, 1, 2, True, 1.0, 0.2),
                                  (2, 16, 16, 20, 24, 3, 2, 1, False, 0.75, -1.0), (1, 8, 8, 512, 512, 3, 1, 1, True, 1.0, 0.2)])
def test_fused_conv_bn_inference_equals_separate_kernels(dev, case):
    """conv -> dropout -> inference BN -> (+ channel-padded shortcut) -> leaky-ReLU in ONE kernel (pnp_conv2d_fwd_bn)
    against the same chain on separate kernels (pnp_conv2d_fwd, pnp_bn_apply), forward and the gradients reaching x, w, shortcut."""
    N, H, W, C, Kc, R, stride, dil, with_sc, keep, alpha = case
    F, K = pkg("functional"), pkg("kernels")
    rng = np.random.default_rng(5)
    g = K.conv_geom((N, H, W, C), (R, R, C, Kc), stride, dil, "SAME")
    mk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev)
    x0, w0 = mk(N, H, W, C), mk(R, R, C, Kc) * float(1.0 / np.sqrt(R * R * C))
    gamma, beta, mm = 1.0 + 0.1 * mk(Kc), 0.1 * mk(Kc), 0.2 * mk(Kc)
    mv = 0.5 + torch.from_numpy(rng.random(Kc).astype(np.float32)).to(dev)
    sc0 = mk(N, g.OH, g.OW, Kc // 2 if Kc % 4 == 0 else Kc) if with_sc else None       # Kc/2 channels: zero-padded Kc/4 each side
    dout = mk(N, g.OH, g.OW, Kc)
    res = {}
    for fused in (True, False):
        F.FUSE_BN_INFER = fused
        x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        sc = sc0.clone().requires_grad_(True) if with_sc else None
        mm1, mv1 = mm.clone(), mv.clone()
        out = F.ConvBNActFn.apply(x, w, gamma, beta, mm1, mv1, sc, g, keep, 77, 3, False, alpha)
        out.backward(dout)
        assert torch.equal(mm1, mm) and torch.equal(mv1, mv)                        # inference mode: moving statistics untouched
        res[fused] = (out.detach(), x.grad, w.grad, sc.grad if with_sc else None)
    F.FUSE_BN_INFER = True
    names = ("out", "dx", "dw", "dshortcut")
    for a, b, nme in zip(res[True], res[False], names):
        if a is None:
            continue
        err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        assert err < 1e-5, (nme, err)      # fused = one fp32 FMA per element more/less; the small-map case also differs in the reduction split
    # with trainable BN parameters the fused route must not be taken (their gradients need the pre-BN tensor)
    gam = gamma.clone().requires_grad_(True)
    out = F.ConvBNActFn.apply(x0, w0, gam, beta, mm.clone(), mv.clone(), None, g, 1.0, 0, 0, False, alpha)
    out.backward(dout)
    assert gam.grad is not None and float(gam.grad.abs().max()) > 0


@pytest.mark.parametrize("case", [(16, 32, 32, 256, 256, 3, 1, 1, 0.75), (8, 64, 64, 64, 64, 3, 1, 1, 1.0), (13, 37, 41, 96, 72, 3, 1, 1, 0.75),
                                  (8, 128, 128, 64, 64, 3, 2, 1, 0.75), (16, 32, 32, 512, 512, 3, 1, 2, 0.75), (2, 256, 256, 32, 64, 3, 1, 1, 0.75),
                                  (5, 120, 100, 32, 64, 3, 1, 1, 0.75),        # 938 partials: compacted in 7 slabs, the last one 170 long
                                  (4, 256, 256, 64, 64, 3, 1, 1, 0.75)],       # cls_1 at 256^2: 4096 partials
                         ids=lambda c: "x".join(str(v) for v in c))
def test_bn_statistics_from_the_conv_epilogue(dev, case):
    """pnp_conv2d_fwd_stats + pnp_bn_stats_finish == pnp_conv2d_fwd + pnp_bn_stats (+ pnp_bn_update_moving): same output tensor bit
    for bit, statistics to float32 round-off, with a non-trivial shift (the moving mean) and ragged tiles"""
    K = pkg("kernels")
    N, H, W, C, Kf, k, stride, dil, keep = case
    rng = np.random.default_rng(sum(case[:6]))
    x = torch.from_numpy(rng.standard_normal((N, H, W, C)).astype(np.float32) + 0.3).to(dev)
    w = torch.from_numpy((rng.standard_normal((k, k, C, Kf)) * np.sqrt(2.0 / (k * k * C)) + 0.01).astype(np.float32)).to(dev)
    g = K.conv_geom(tuple(x.shape), tuple(w.shape), stride, dil, "SAME")
    assert K.conv_stats_parts(g) > 0
    mm = torch.from_numpy((0.2 * rng.standard_normal(Kf)).astype(np.float32)).to(dev)
    mv = torch.from_numpy((1.0 + 0.1 * rng.random(Kf)).astype(np.float32)).to(dev)
    mm2, mv2 = mm.clone(), mv.clone()
    y_ref = K.conv2d_fwd(x, w, g, keep, 5, 2)
    mean_ref, var_ref = K.bn_stats_update(y_ref, mm2, mv2, 0.9)
    y, parts = K.conv2d_fwd_stats(x, w, g, mm, keep, 5, 2)
    P = y.numel() // Kf
    mean, var = K.bn_stats_finish(parts, mm, P, mm, mv, 0.9)
    assert torch.equal(y, y_ref)
    y64 = y_ref.double().reshape(P, Kf)
    m64, v64 = y64.mean(0), y64.var(0, unbiased=False)
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    print("epilogue stats %s: mean %.2e var %.2e (two-pass kernel: %.2e %.2e)" % (case, rel(mean, m64), rel(var, v64), rel(mean_ref, m64), rel(var_ref, v64)))
    assert rel(mean, m64) < 2e-6 and rel(var, v64) < 1e-5
    assert rel(mm, mm2.double()) < 1e-6 and rel(mv, mv2.double()) < 1e-5


def test_no_epilogue_statistics_where_the_forward_is_not_on_the_mfma_tiles(dev):
    K = pkg("kernels")
    assert K.conv_stats_parts(K.conv_geom((16, 256, 256, 16), (3, 3, 16, 16), 1, 1, "SAME")) == 0       # narrow-output vector-ALU kernel
    assert K.conv_stats_parts(K.conv_geom((16, 16, 16, 512), (5, 5, 512, 512), 4, 1, "SAME")) == 0      # reduction-split tiny layer
    assert K.conv_stats_parts(K.conv_geom((2, 64, 64, 64), (3, 3, 64, 64), 1, 1, "SAME")) == 0          # few tiles at a small batch: split too
