"""-m gpu: the Winograd F(2x2, 3x3) and F(4x4, 3x3) routes of the wide stride-1 3x3 convolutions (csrc/conv_wino.hip): forward, data
gradient and filter gradient (the transposed algorithm), every test once per output tile (kernels.wino_tile 2 / 4).

Oracle: the float64 convolution (oracle.tf_ops.conv2d / autograd) of the same float32 operands; bar 2e-5 of max|ref| (the direct fp32
kernels land at 1e-7..5e-6 on the same cases; the Winograd transforms add ~1 ulp in front of and behind the contraction).  Every case
runs BOTH routes through the same entry points (kernels.wino_mode 0 / 2) and checks which kernel symbols ran, so a silent fall-back to
the direct kernels cannot pass as Winograd.  Epilogue features (dropout mask stream, residual add, BN statistics partials, fused inference
BN + channel-padded shortcut + leaky-ReLU) are held to the direct route's results: same mask bits, same statistics to fp32 rounding."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

# (N, H, W, C, K, dil, padding)
CASES = [
    (4, 32, 32, 512, 512, 1, "SAME"),        # group_7..9
    (4, 32, 32, 512, 512, 2, "SAME"),        # group_8: atrous rate 2 = four dense sub-images
    (2, 34, 34, 512, 2560, 1, "VALID"),      # group_10 on its mirror-padded input (K = 2560: three channel slices of the output transform)
    (4, 32, 32, 256, 512, 1, "SAME"),        # group_7's first layer
    (2, 64, 64, 256, 256, 1, "SAME"),        # critic cls_3
    (1, 31, 33, 64, 96, 1, "SAME"),          # odd extents: ragged 2x2 tiles, T and K not multiples of the GEMM tile (K = 96, 24 channel quads)
    (1, 30, 34, 32, 160, 2, "SAME"),         # dilation 2 on odd sub-image extents (15 x 17), K = 160: 40 channel quads
    (3, 12, 20, 96, 64, 1, "VALID"),         # VALID (padding 0; its data gradient: padding 2), C = 96 (three 32-groups)
    (1, 16, 16, 32, 1056, 1, "SAME"),        # K / 4 = 264: two channel slices, the second 8 quads wide
]
BAR = 2e-5


def _rel(a, b):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def _ran(L, fn, cls):
    L.prof_summary()
    L.prof_enable(cls)
    out = fn()
    torch.cuda.synchronize()
    L.prof_enable(0)
    # (the split-bf16 GEMM has a 128-row and a 256-row tile instance, chosen by the launch's shape: one name here)
    return out, [r["name"].replace("wino_gemm_x3_kernel<256,", "wino_gemm_x3_kernel<128,") for r in L.prof_summary()]


@pytest.fixture(params=[(2, 0), (4, 0), (2, 1), (4, 1)], ids=["F2x2", "F4x4", "F2x2-x3", "F4x4-x3"])
def wino(request):
    """yields a setter of the route policy with the output tile (wino.tile) and the GEMM arithmetic (wino.x3: 0 = fp32 matrix pipe, 1 =
    split-bf16 operands, csrc/conv_wino_x3.hip) of this run in force; restores the modes in force before the test"""
    K = pkg("kernels")
    tile, x3 = request.param
    prev, prev_w, prev_t, prev_x = K.wino_mode(-1), K.wino_wgrad_mode(-1), K.wino_tile(tile), K.wino_x3(2 if x3 else 0)        # (2: wherever the shapes allow)

    def setter(mode):
        return K.wino_mode(mode)
    setter.tile = tile
    setter.x3 = x3
    yield setter
    K.wino_mode(prev)
    K.wino_wgrad_mode(prev_w)
    K.wino_tile(prev_t)
    K.wino_x3(prev_x)


def _route_names(tile, x3, trans, bn, sym):
    """kernel symbols of one forward / data-gradient launch of the route"""
    b = "true" if x3 else "false"
    gemm = ("wino_gemm_x3_kernel<128, %d, %d>" if x3 else "wino_gemm_kernel<128, %d, 2, 2, %d>") % (bn, sym)
    return sorted(["wino_filter_kernel<%s, %d, %s>" % ("true" if trans else "false", tile, b), gemm, "wino_in_kernel<%d, %s>" % (tile, b),
                   "wino_out_kernel<%d>" % tile])


def _where(err, K_):
    """where a mismatch sits (for the log of a failing run): output-tile position, border vs interior, channel block"""
    e = err.abs().cpu().double()
    N, H, W, Kc = e.shape
    pos = [[float(e[:, p::2, q::2].max()) for q in (0, 1)] for p in (0, 1)]
    border = float(torch.cat([e[:, :2].flatten(), e[:, -2:].flatten(), e[:, :, :2].flatten(), e[:, :, -2:].flatten()]).max())
    inner = float(e[:, 2:-2, 2:-2].max()) if H > 4 and W > 4 else 0.0
    blocks = [float(e[..., i:i + 32].max()) for i in range(0, min(Kc, 256), 32)]
    return {"by tile position": pos, "border": border, "interior": inner, "per 32 channels": blocks}


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_winograd_fwd_dgrad_vs_float64_and_direct(dev, wino, case):
    K, L = pkg("kernels"), pkg("_lib")
    N, H, W, C, Kf, dil, padding = case
    rng = np.random.default_rng(sum(case[:6]))
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    w = (rng.standard_normal((3, 3, C, Kf)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    g = K.conv_geom(x.shape, w.shape, 1, dil, padding)
    dy = rng.standard_normal((N, g.OH, g.OW, Kf)).astype(np.float32)
    res = rng.standard_normal(x.shape).astype(np.float32)
    xd, wd, dyd, resd = (torch.from_numpy(a).to(dev) for a in (x, w, dy, res))
    xg = torch.from_numpy(x).double().requires_grad_(True)
    yo = T.conv2d(xg, torch.from_numpy(w).double(), 1, dil, padding)
    yo.backward(torch.from_numpy(dy).double())

    wino(0)
    assert not K.wino_chosen(g, 0) and not K.wino_chosen(g, 1)
    y0, names0 = _ran(L, lambda: K.conv2d_fwd(xd, wd, g), L.PROF_CONV_FWD)
    dx0 = K.conv2d_dgrad(dyd, wd, g)
    assert not any("wino" in n for n in names0), names0
    wino(2)
    # (on the fp32 matrix pipe F(4x4) does not take reductions over more than 1 024 channels: group_10's data gradient runs on F(2x2)
    # whatever the mode; the split-bf16 GEMM accumulates in 96-channel chunks and has no such cap)
    # (the split-bf16 GEMM takes reductions over a multiple of 64 channels; the others keep the fp32 pipe)
    x3f, x3d = bool(wino.x3 and C % 64 == 0), bool(wino.x3 and Kf % 64 == 0)
    tile_d = 2 if (wino.tile == 4 and Kf > 1024 and not x3d) else wino.tile
    assert K.wino_chosen(g, 0) == wino.tile and K.wino_chosen(g, 1) == tile_d
    y1, names1 = _ran(L, lambda: K.conv2d_fwd(xd, wd, g), L.PROF_CONV_FWD)
    kb = 0 if wino.tile == 2 else 2          # wino_gemm[_x3]_kernel<.., 0 / 1>: F(2x2) forward / data gradient, <.., 2 / 3>: F(4x4)
    # (128 x 64 GEMM tiles where the GEMM has <= 64 columns: the forward's filters, the data gradient's channels)
    assert sorted(names1) == _route_names(wino.tile, x3f, False, 64 if Kf <= 64 else 128, kb), names1
    dx1, names2 = _ran(L, lambda: K.conv2d_dgrad(dyd, wd, g), L.PROF_CONV_DGRAD)
    kbd = 0 if tile_d == 2 else 2
    assert sorted(names2) == _route_names(tile_d, x3d, True, 64 if C <= 64 else 128, kbd + 1), names2
    dxr = K.conv2d_dgrad(dyd, wd, g, residual=resd)
    # the filter gradient on the route (its own switch): plain and added into a buffer that already holds a contribution
    wg = torch.from_numpy(w).double().requires_grad_(True)
    T.conv2d(torch.from_numpy(x).double(), wg, 1, dil, padding).backward(torch.from_numpy(dy).double())
    K.wino_wgrad_mode(0)
    assert not K.wino_chosen(g, 2)
    dw0 = K.conv2d_wgrad(xd, dyd, g)
    K.wino_wgrad_mode(2)
    assert K.wino_chosen(g, 2) == wino.tile
    dw1, names3 = _ran(L, lambda: K.conv2d_wgrad(xd, dyd, g), L.PROF_CONV_WGRAD)
    # (+ wino_splitsum_kernel where a narrow layer's reduction is split many ways)
    if wino.x3:        # split-bf16 GEMMs (sym 4 / 5 = F(2x2) / F(4x4)) on operands transposed by the transforms (the reduction runs over the tiles)
        want3 = sorted(["wino_dy_t_kernel<%d>" % wino.tile, "wino_in_t_kernel<%d>" % wino.tile, "wino_wgrad_out_kernel<%d>" % wino.tile,
                        "wino_gemm_x3_kernel<128, %d, %d>" % (64 if Kf <= 64 else 128, 4 if wino.tile == 2 else 5)])
    else:
        want3 = [n % wino.tile for n in ("wino_dy_kernel<%d>", "wino_in_kernel<%d, false>", "wino_wgrad_gemm_kernel<128, 128, 2, 2, %d>", "wino_wgrad_out_kernel<%d>")]
    assert sorted(n for n in names3 if n != "wino_splitsum_kernel") == want3, names3
    held = torch.from_numpy(rng.standard_normal(w.shape).astype(np.float32)).to(dev)
    dwa = K.conv2d_wgrad(xd, dyd, g, into=held.clone())
    errs = {"y direct": _rel(y0, yo), "y wino": _rel(y1, yo), "dx direct": _rel(dx0, xg.grad), "dx wino": _rel(dx1, xg.grad),
            "dx+res wino": _rel(dxr, xg.grad + torch.from_numpy(res).double()), "dw direct": _rel(dw0, wg.grad), "dw wino": _rel(dw1, wg.grad),
            "dw+held wino": _rel(dwa, wg.grad + held.cpu().double())}
    print("wino F(%dx%d)%s %s: %s" % (wino.tile, wino.tile, " x3" if wino.x3 else "", case, {k: "%.2e" % v for k, v in errs.items()}))
    if errs["y wino"] > BAR:
        print("  forward mismatch:", _where(y1.cpu().double() - yo.detach(), Kf))
    if errs["dx wino"] > BAR:
        print("  data-gradient mismatch:", _where(dx1.cpu().double() - xg.grad, C))
    assert max(errs.values()) < BAR, errs


def test_winograd_epilogues_equal_the_direct_route(dev, wino):
    """dropout (the same mask stream: identical zero pattern), BN statistics partials -> mean / variance / moving averages, fused
    inference BN + channel-padded shortcut + leaky-ReLU: the Winograd route's output transform against conv_epilogue of the direct kernels"""
    K, L = pkg("kernels"), pkg("_lib")
    rng = np.random.default_rng(5)
    for (N, H, C, Kf, dil, Cs) in ((2, 32, 256, 256, 1, 128), (1, 32, 64, 160, 2, 160), (2, 16, 32, 1056, 1, 1000)):
        x = torch.from_numpy(rng.standard_normal((N, H, H, C)).astype(np.float32)).to(dev)
        w = torch.from_numpy((rng.standard_normal((3, 3, C, Kf)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)).to(dev)
        g = K.conv_geom(tuple(x.shape), tuple(w.shape), 1, dil, "SAME")
        shift = torch.from_numpy((rng.standard_normal(Kf) * 0.1).astype(np.float32)).to(dev)
        sc = torch.from_numpy(rng.standard_normal((N, H, H, Cs)).astype(np.float32)).to(dev)
        ss = torch.from_numpy(np.stack([rng.uniform(0.5, 1.5, Kf), rng.standard_normal(Kf)]).astype(np.float32)).to(dev)
        out = {}
        for mode in (0, 2):
            wino(mode)
            assert K.wino_chosen(g, 0) == (wino.tile if mode == 2 else 0)
            yd = K.conv2d_fwd(x, w, g, keep_prob=0.75, seed=1234, stream_id=7)
            mm, mv = torch.zeros(Kf, device=dev), torch.ones(Kf, device=dev)
            nparts = K.conv_stats_parts(g)
            if nparts > 0:
                ys, parts = K.conv2d_fwd_stats(x, w, g, shift, keep_prob=0.75, seed=1234, stream_id=7)
                mean, var = K.bn_stats_finish(parts, shift, N * H * H, mm, mv)
            else:                                   # (the direct route splits the reduction of a layer with few tiles: no epilogue statistics)
                ys = yd
                mean, var = K.bn_stats(yd)
            yb = K.conv2d_fwd_bn(x, w, g, ss, shortcut=sc, alpha=0.2, keep_prob=0.75, seed=1234, stream_id=7)
            out[mode] = (yd, ys, mean, var, yb)
        (yd0, ys0, m0, v0, yb0), (yd1, ys1, m1, v1, yb1) = out[0], out[2]
        assert torch.equal(yd0 == 0, yd1 == 0), "dropout masks differ"               # the counter hash on the flat output index
        assert 0.2 < float((yd1 == 0).float().mean()) < 0.3
        assert torch.equal(ys1, yd1)                                                    # the statistics launch leaves the same output
        errs = {"drop": _rel(yd1, yd0), "mean": _rel(m1, m0), "var": _rel(v1, v0), "fused bn": _rel(yb1, yb0)}
        # the statistics against float64 moments of the route's own output
        yd64 = yd1.double().reshape(-1, Kf)
        errs["mean vs f64"] = float((m1.double() - yd64.mean(0)).abs().max() / yd64.std())
        errs["var vs f64"] = _rel(v1, yd64.var(0, unbiased=False))
        print("wino F(%dx%d) epilogues (%d, %d, %d->%d, dil %d): %s" % (wino.tile, wino.tile, N, H, C, Kf, dil, {k: "%.2e" % v for k, v in errs.items()}))
        assert max(errs.values()) < BAR, errs


def test_segmenter_step_on_the_winograd_route(dev, wino):
    """one source-segmenter train step (source_segmenter.py:484-489) with the planner's own choice (mode 1) against the direct kernels
    (mode 0): the wide layers really take the route (wino_gemm_kernel in both passes, wino_wgrad_gemm_kernel for the filter gradients), loss and every gradient agree to fp32 rounding
    through 33 convolutions (cosine over the whole gradient arena, loss to 1e-5)"""
    ss, L, K = pkg("source_segmenter"), pkg("_lib"), pkg("kernels")
    B = 4
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dev)
    lab = rng.integers(0, 5, (B, 256, 256))
    y = torch.from_numpy(np.eye(5, dtype=np.float32)[lab]).to(dev)
    res = {}
    for mode in (0, 1):
        wino(mode)
        K.wino_wgrad_mode(mode)
        net = ss.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, cost_kwargs={"cross_flag": True, "miu_cross": 1.0, "dice_flag": True, "miu_dice": 1.0, "regularizer": 1e-4}, seed=0)
        tr = ss.Trainer(net, None, None, num_cls=5, batch_size=B, optimizer="adam", opt_kwargs={"learning_rate": 1e-3})
        tr.opt = tr._get_optimizer(10)
        L.prof_summary()
        L.prof_enable(L.PROF_CONV_FWD | L.PROF_CONV_DGRAD | L.PROF_CONV_WGRAD)
        loss = float(tr.train_step(x, y, 0.75, 11))
        torch.cuda.synchronize()
        L.prof_enable(0)
        names = [r["name"].replace("wino_gemm_x3_kernel<256,", "wino_gemm_x3_kernel<128,") for r in L.prof_summary()]
        res[mode] = (loss, net.store.grad_arena.clone(), net.store.arena.clone(), names)
    (l0, g0, w0, n0), (l1, g1, w1, n1) = res[0], res[1]
    assert not any("wino" in n for n in n0)
    assert any(n.startswith("wino_gemm_x3_kernel<128, 128, %d>" % (4 if wino.tile == 2 else 5) if wino.x3 else "wino_wgrad_gemm_kernel") for n in n1), sorted(set(n1))
    kb = 0 if wino.tile == 2 else 2
    gk = "wino_gemm_x3_kernel<128, 128, %d>" if wino.x3 else "wino_gemm_kernel<128, 128, 2, 2, %d>"
    assert any(n.startswith(gk % kb) for n in n1) and any(n.startswith(gk % (kb + 1)) for n in n1), sorted(set(n1))
    cos = float(torch.nn.functional.cosine_similarity(g0.double().flatten(), g1.double().flatten(), dim=0))
    cw = float(torch.nn.functional.cosine_similarity(w0.double().flatten(), w1.double().flatten(), dim=0))
    print("segmenter step, Winograd F(%dx%d)%s route vs direct: loss %.7f vs %.7f, gradient cosine %.8f, weights-after-Adam cosine %.8f" % (wino.tile, wino.tile, " x3" if wino.x3 else "", l1, l0, cos, cw))
    assert abs(l1 - l0) < 1e-5 * max(1.0, abs(l0)) and cos > 0.99999 and cw > 0.999999


def test_transformed_filter_cache(dev, wino):
    """pnp_conv2d_wino_filter_bind / pnp_weights_changed: a store-owned filter is transformed once per weight change, not once per launch —
    same bits as the un-cached launch; a write reported by range, a write by a torch op (version counter) and a new tensor object on the same
    address each invalidate the entry; an unrelated range does not; ad-hoc filters are never cached"""
    K, L = pkg("kernels"), pkg("_lib")
    rng = np.random.default_rng(17)
    N, H, C, Kf = 2, 16, 64, 96
    x = torch.from_numpy(rng.standard_normal((N, H, H, C)).astype(np.float32)).to(dev)
    dy = torch.from_numpy(rng.standard_normal((N, H, H, Kf)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal((3, 3, C, Kf)) * 0.05).astype(np.float32)).to(dev)
    g = K.conv_geom(tuple(x.shape), tuple(w.shape), 1, 1, "SAME")
    wino(2)
    K.wino_u_cache_clear()
    K.wino_u_cache_stats(reset=True)
    y_ref, dx_ref = K.conv2d_fwd(x, w, g), K.conv2d_dgrad(dy, w, g)                 # ad-hoc filter: no entry
    assert K.wino_u_cache_stats() == (0, 0)
    w._pnp_var = True                                                                # what VariableStore.finalize does for its filters
    for i in range(3):
        assert torch.equal(K.conv2d_fwd(x, w, g), y_ref) and torch.equal(K.conv2d_dgrad(dy, w, g), dx_ref)
    assert K.wino_u_cache_stats(reset=True) == (4, 2), "one fill per pass, then hits"
    lo = w.data_ptr()
    K.weights_changed([(lo + 4 * w.numel(), lo + 8 * w.numel())])                   # somebody else's weights
    assert torch.equal(K.conv2d_fwd(x, w, g), y_ref) and K.wino_u_cache_stats(reset=True) == (1, 0)
    K.axpby(w, w, 1.0, 1.0)                                                          # w <- 2 w through a raw kernel: reported by range
    K.weights_changed([(lo, lo + 4 * w.numel())])
    y2 = K.conv2d_fwd(x, w, g)
    assert K.wino_u_cache_stats(reset=True) == (0, 1) and _rel(y2, 2 * y_ref.double()) < 1e-6
    w.mul_(0.5)                                                                      # a torch op: the version counter moves
    assert torch.equal(K.conv2d_fwd(x, w, g), y_ref) and K.wino_u_cache_stats(reset=True) == (0, 1)
    assert _rel(K.conv2d_dgrad(dy, w, g), dx_ref.double()) == 0.0                    # (the data-gradient entry was dropped by the range above)
    K.wino_u_cache_stats(reset=True)
    K.weights_changed()                                                              # everything
    assert torch.equal(K.conv2d_fwd(x, w, g), y_ref) and K.wino_u_cache_stats(reset=True)[1] == 1
    # a binding goes when its filter tensor does: an ad-hoc filter that lands on the recycled address (the caching allocator hands the
    # block straight back) is transformed afresh — never served the dead filter's U
    addr = w.data_ptr()
    del w
    assert not any(k[0] == addr for k in K._u_cache)
    w3 = torch.from_numpy((rng.standard_normal((3, 3, C, Kf)) * 0.05).astype(np.float32)).to(dev)
    y3 = K.conv2d_fwd(x, w3, g)
    wino(0)
    y3d = K.conv2d_fwd(x, w3, g)
    wino(2)
    print("recycled address: %s; new filter on the route vs direct %.2e" % (w3.data_ptr() == addr, _rel(y3, y3d)))
    assert K.wino_u_cache_stats(reset=True) == (0, 0) and _rel(y3, y3d) < BAR
    K.wino_u_cache_clear()


def test_joint_steps_with_and_without_the_filter_cache(dev, wino):
    """one discriminator + one generator update (adversarial.py:839-882) with the transformed-filter cache on and off: identical losses and
    weights (the cache only skips launches), and with it on the frozen layers' transforms are skipped from the second pass on"""
    adv, K = pkg("adversarial"), pkg("kernels")
    if wino.tile == 2:
        pytest.skip("one tile is enough here")
    B = 2
    rng = np.random.default_rng(4)
    mr = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dev)
    ct = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dev)
    wino(2)
    K.wino_wgrad_mode(2)
    out = {}
    prev = K.U_CACHE
    try:
        for on in (False, True):
            K.U_CACHE = on
            K.wino_u_cache_clear()
            K.wino_u_cache_stats(reset=True)
            net = adv.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, seed=0,
                               cost_kwargs={"regularizer": 1e-4, "gan_regularizer": 1e-4, "miu_gen": 0.002, "miu_dis": 0.002, "lambda_mask_loss": 0.3},
                               network_config={"mr_front_trainable": False, "joint_trainable": False, "ct_front_trainable": True, "cls_trainable": True,
                                               "m_cls_trainable": True})
            # He-scaled filters and non-trivial moving statistics (tests/test_gpu_adversarial.py::he_state — what every other whole-step test
            # starts from): the reference's stddev-.01 init with un-calibrated statistics overflows within two updates, and a comparison
            # of two overflowed networks proves little
            from test_gpu_adversarial import he_state
            net.store.load_state_dict(he_state(net, 9))
            K.weights_changed()
            tr = adv.Trainer(net, None, None, None, None, num_cls=5, batch_size=B, opt_kwargs={"learning_rate": 3e-4},
                             train_config={"dis_sub_iter": 1, "gen_sub_iter": 1})
            tr._get_optimizer()
            losses = []
            for i in range(2):
                losses.append(float(tr.dis_step(mr, ct, 0.75, 2 * i + 1)))
                losses.append(float(tr.gen_step(ct, 0.75, 2 * i + 2)))
            torch.cuda.synchronize()
            out[on] = (losses, net.store.arena.clone(), K.wino_u_cache_stats(reset=True))
    finally:
        K.U_CACHE = prev
        K.wino_u_cache_clear()
    (l0, a0, s0), (l1, a1, s1) = out[False], out[True]
    print("joint steps: filter-transform cache off %s / on %s (hits, fills); losses %s" % (s0, s1, l1))
    assert s0 == (0, 0) and s1[0] > s1[1] > 0
    # bit for bit, on finite numbers
    assert all(np.isfinite(v) for v in l0) and bool(torch.isfinite(a0).all()) and bool(torch.isfinite(a1).all())
    assert l0 == l1 and bool((a0 == a1).all())


@pytest.mark.parametrize("split", ["0", "1"], ids=["persistent", "persistent+tail-split"])
def test_persistent_gemm_and_its_tail_split(dev, split):
    """round 5: with more tiles than workgroup slots the route's GEMM launch is persistent (the software pipeline runs across tile
    boundaries); PNP_WINO_TAILSPLIT=1 (read once by the library: a process of its own) additionally cuts every XCD's tail tiles into
    pieces of the reduction whose partial products the output transform sums.  BASELINE-batch layers, forward and data gradient against
    the direct kernels, and the workspace sizes of both plans (tests/wino_split_worker.py)"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PNP_WINO_TAILSPLIT=split)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "wino_split_worker.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:])
    assert r.returncode == 0 and "SPLIT WORKER OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
