"""-m gpu: whole-network parity of the segmenter training step (BASELINE config 2 at reduced batch) vs the CPU oracle:
logits, bit-exact argmax label map, loss, every gradient, post-Adam weights and BN moving statistics.

Tolerances.  Per-op kernels meet 1e-4 relative (tests/test_gpu_conv.py ...).  Through the 33-conv network with
training-mode BN, two DIFFERENT fp32 evaluation orders legitimately diverge by more than 1e-4 at the earliest layers,
so the gradient check is adjudicated by the float64 oracle: the HIP result must be as close to float64 as the float32
CPU oracle is (same round-off class), and never worse than 2e-2 of the gradient's max magnitude."""
import numpy as np
import pytest
import torch

from conftest import pkg
from parity_util import assert_argmax_exact
from oracle import nets
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

COST = {"cross_flag": True, "miu_cross": 1.0, "dice_flag": True, "miu_dice": 1.0, "regularizer": 1e-4}


def _blob_labels(rng, B):
    yy, xx = np.mgrid[0:256, 0:256]
    lab = np.zeros((B, 256, 256), np.int64)
    for b in range(B):
        for c in range(1, 5):
            cy, cx = rng.integers(40, 216, 2)
            ry, rx = rng.integers(12, 40, 2)
            lab[b][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1] = c
    return lab


def _rel(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def he_scaled(sd):
    """init stddev .01 gives ~0 logits; rescale conv weights to ~He so that logits have realistic margins"""
    out = {}
    for k, a in sd.items():
        if "/Variable" in k:
            out[k] = (a * (np.sqrt(2.0 / (a.shape[0] * a.shape[1] * a.shape[2])) / 0.01 * 0.9)).astype(np.float32)
        else:
            out[k] = a
    return out


def _cos(a, b):
    a = torch.as_tensor(a).double().reshape(-1)
    b = torch.as_tensor(b).double().reshape(-1)
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


# logit_scale 0.05: softmax probabilities stay far from the 0.005 clip of the weighted cross-entropy, the loss is smooth and the
#   whole backward chain can be held to a tight tolerance against float64.
# logit_scale 1.0 : realistic margins; pixels sitting on the clip threshold flip between ANY two fp32 evaluation orders (the
#   float32 CPU oracle itself is then 1e-2 away from float64), so the check is statistical: same error class as cpu-fp32.
@pytest.mark.parametrize("keep_prob,logit_scale", [(1.0, 0.05), (0.75, 0.05), (0.75, 1.0)])
def test_segmenter_train_step_parity(dev, keep_prob, logit_scale):
    ss = pkg("source_segmenter")
    B = 2
    rng = np.random.default_rng(0)
    x = rng.standard_normal((B, 256, 256, 3)).astype(np.float32)
    y = T.label_decomp(5, _blob_labels(rng, B))
    net = ss.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, cost_kwargs=dict(COST), seed=3)
    sd = he_scaled(net.store.state_dict())
    sd["output/Variable"] = (sd["output/Variable"] * logit_scale).astype(np.float32)
    net.store.load_state_dict(sd)
    tr = ss.Trainer(net, None, None, num_cls=5, batch_size=B, optimizer="adam", opt_kwargs={"learning_rate": 1e-3})
    tr.opt = tr._get_optimizer(10)

    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    loss = tr.train_step(xd, yd, keep_prob, step=0)       # drop seed = step+1 = 1
    logits = net.logits.detach().cpu()
    # the product applies the L2 term reg_coeff*mult*w inside the optimiser kernel; add it back to compare d(cost+reg)/dw
    grads = {v.name: v.tensor.grad.detach().cpu().clone() + COST["regularizer"] * nets.l2_multiplicity(v.name) * torch.from_numpy(sd[v.name])
             for v in net.store.trainable()}
    after = net.store.state_dict()

    V32 = nets.make_variables(sd)
    cost32, g32, logits32 = nets.segmenter_train_step(V32, {}, torch.from_numpy(x), torch.from_numpy(y), keep_prob, seed=1, lr=1e-3, t=1)
    V64 = nets.make_variables(sd, dtype=torch.float64)
    cost64, g64, logits64 = nets.segmenter_train_step(V64, {}, torch.from_numpy(x).double(), torch.from_numpy(y).double(), keep_prob,
                                                      seed=1, lr=1e-3, t=1)

    e_hip, e_cpu = _rel(logits, logits64), _rel(logits32, logits64)
    print("logits rel err vs fp64: hip %.3e  cpu-fp32 %.3e ; hip vs cpu-fp32 %.3e" % (e_hip, e_cpu, _rel(logits, logits32)))
    print("loss hip %.7f cpu32 %.7f fp64 %.7f" % (float(loss), float(cost32), float(cost64)))
    assert e_hip < 1e-4
    assert abs(float(loss) - float(cost64)) < 1e-4 * max(1.0, abs(float(cost64)))

    # argmax label map must be bit exact; a mismatch is tolerated only where the fp64 top-2 margin is itself within fp32 noise
    assert_argmax_exact(logits, logits64, "HIP label map vs float64 oracle")
    print("cpu-fp32 oracle mismatches vs fp64: %d" % int((logits32.argmax(-1) != logits64.argmax(-1)).sum()))

    rows = []
    for k, g in g64.items():
        rows.append((k, _rel(grads[k], g), _rel(g32[k], g), _cos(grads[k], g)))
    eh_all = np.array([r[1] for r in rows])
    ec_all = np.array([r[2] for r in rows])
    print("gradient error vs fp64 over %d variables: hip median %.3e max %.3e | cpu-fp32 median %.3e max %.3e | min cosine %.8f" % (
        len(rows), np.median(eh_all), eh_all.max(), np.median(ec_all), ec_all.max(), min(r[3] for r in rows)))
    rows.sort(key=lambda r: -r[1])
    for r in rows[:5]:
        print("   %-28s hip %.3e cpu %.3e cos %.8f" % r)
    # same round-off class as the float32 CPU oracle (both measured against float64), and the same direction
    assert np.median(eh_all) < 3.0 * np.median(ec_all) + 1e-4
    assert eh_all.max() < max(3.0 * ec_all.max(), 1e-3)
    assert min(r[3] for r in rows) > 0.9999

    # BN moving statistics, and the Adam update given the HIP path's own gradients (isolates optimiser + L2 + arena plumbing)
    upd_worst = 0.0
    for k, v in V64.items():
        if k.endswith("moving_mean") or k.endswith("moving_variance"):
            assert _rel(after[k], v.detach()) < 1e-4, k
            continue
        w = torch.from_numpy(sd[k].copy())
        T.adam_update(w, grads[k], torch.zeros_like(w), torch.zeros_like(w), 1e-3, 1)
        upd_worst = max(upd_worst, float((w - torch.from_numpy(after[k])).abs().max() / 1e-3))
    print("worst post-step weight error (fraction of lr): %.3e" % upd_worst)
    assert upd_worst < 1e-3


def test_segmenter_eval_outputs(dev):
    """the monitoring fetches of the reference graph: predicter, compact_pred, dice_eval, regularizer_loss"""
    ss = pkg("source_segmenter")
    B = 2
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, 256, 256, 3)).astype(np.float32)
    y = T.label_decomp(5, _blob_labels(rng, B))
    net = ss.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, cost_kwargs=dict(COST), seed=4)
    sd = he_scaled(net.store.state_dict())
    net.store.load_state_dict(sd)
    net.evaluate(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), keep_prob=1.0, main_bn=False, adapt_bn=False,
                 want_confusion=True)
    V = nets.make_variables(sd, requires_grad=False)
    with torch.no_grad():
        lo = nets.segmenter_forward(V, torch.from_numpy(x), 1.0, False, False)
        cost, reg, wl, dl = nets.segmenter_cost(V, lo, torch.from_numpy(y))
        po = T.pixel_wise_softmax_2(lo)
        de, _ = T.dice_eval(T.argmax_lowest(po), torch.from_numpy(y), 5)
    assert _rel(net.logits, lo) < 1e-4
    assert abs(float(net.cost) - float(cost)) < 1e-4
    assert abs(float(net.regularizer_loss) - float(reg)) < 1e-5 * float(reg)
    assert abs(float(net.dice_eval) - float(de)) < 1e-4
    assert net.confusion_matrix.sum() == B * 256 * 256
    # inference-mode BN must not move the moving statistics
    after = net.store.state_dict()
    for k in sd:
        if k.endswith("moving_mean") or k.endswith("moving_variance"):
            assert np.array_equal(after[k], sd[k]), k


@pytest.mark.slow
def test_segmenter_train_step_B16_vs_float32_oracle(dev):
    """BASELINE configs[1] at its own batch: at B=16 the planners pick 128x128 tiles, 7-way filter-gradient splits and reduction-split
    data gradients that no B=2 test reaches; the whole step is held to the float32 CPU oracle here (float64 at this size costs
    minutes; the per-kernel 1e-4 pins at B=16 live in test_gpu_teacher_forced.py)"""
    ss = pkg("source_segmenter")
    B = 16
    rng = np.random.default_rng(40)
    x = rng.standard_normal((B, 256, 256, 3)).astype(np.float32)
    y = T.label_decomp(5, _blob_labels(rng, B))
    net = ss.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, cost_kwargs=dict(COST), seed=5)
    sd = he_scaled(net.store.state_dict())
    net.store.load_state_dict(sd)
    tr = ss.Trainer(net, None, None, num_cls=5, batch_size=B, optimizer="adam", opt_kwargs={"learning_rate": 1e-3})
    tr.opt = tr._get_optimizer(10)
    loss = tr.train_step(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), 0.75, step=0)
    logits = net.logits.detach().cpu()
    grads = {v.name: v.tensor.grad.detach().cpu().clone() + COST["regularizer"] * nets.l2_multiplicity(v.name) * torch.from_numpy(sd[v.name])
             for v in net.store.trainable()}
    after = net.store.state_dict()
    V32 = nets.make_variables(sd)
    cost32, g32, logits32 = nets.segmenter_train_step(V32, {}, torch.from_numpy(x), torch.from_numpy(y), 0.75, seed=1, lr=1e-3, t=1)
    e = _rel(logits, logits32)
    print("B=16 logits hip vs cpu-fp32 %.3e ; loss hip %.7f cpu %.7f" % (e, float(loss), float(cost32)))
    assert e < 1e-4 and abs(float(loss) - float(cost32)) < 1e-5 * max(1.0, abs(float(cost32)))
    # label map: identical wherever the oracle's own top-2 margin is above the float32 noise of two evaluations
    top2 = torch.topk(logits32, 2, dim=-1).values
    far = (top2[..., 0] - top2[..., 1]) > 3e-5 * float(logits32.abs().max())
    mism = logits.argmax(-1) != logits32.argmax(-1)
    print("B=16 argmax mismatches %d of %d, away from near-ties %d" % (int(mism.sum()), mism.numel(), int((mism & far).sum())))
    assert int((mism & far).sum()) == 0
    cs = {k: _cos(grads[k], g32[k]) for k in g32}
    er = np.array([_rel(grads[k], g32[k]) for k in g32])
    print("B=16 gradients hip vs cpu-fp32 over %d variables: median %.3e max %.3e, min cosine %.8f (%s)" % (
        len(er), np.median(er), er.max(), min(cs.values()), min(cs, key=cs.get)))
    # fp32 against fp32 through the whole backward chain: both sides carry the slope-flip noise that the B=2 tests measure against
    # float64.  The bar follows the MEASUREMENT (profiles/r03_pytest_gpu.log, MI355X round 3): median 4.2e-3, max 3.5e-2 of max|g|, min
    # cosine 0.999983 — i.e. round 1's original bar (0.9999 / 1e-2) holds at this batch; round 2's last commit had widened it to
    # 0.9997 / 2e-2 without a run.  The maximum is bounded too: a defect in one B=16-only path (128x128 tiles, 7-way filter-
    # gradient splits, split data gradients, two-stage BN compaction) would show up in a single variable first.  The per-kernel 1e-4
    # pins at this batch are tests/test_gpu_teacher_forced.py.
    assert min(cs.values()) > 0.9999 and np.median(er) < 1e-2 and er.max() < 0.1
    for k, v in V32.items():
        if k.endswith("moving_mean") or k.endswith("moving_variance"):
            assert _rel(after[k], v.detach()) < 1e-4, k


def test_gradient_sinks_and_shortcut_links_equal_the_autograd_engine(dev):
    """The parameter gradients the kernels add straight into the arena (gradsink) and the shortcut gradient the data-gradient kernel
    adds (functional.ResLink) must equal what the autograd engine's own AccumulateGrad / AddN produce: same step, both ways, and no
    `aten::add` launches left on the fast path."""
    ss, F, gs = pkg("source_segmenter"), pkg("functional"), pkg("gradsink")
    rng = np.random.default_rng(7)
    B = 2
    x = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dev)
    y = torch.from_numpy(np.eye(5, dtype=np.float32)[_blob_labels(rng, B)]).to(dev)
    net = ss.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, cost_kwargs=dict(COST), seed=3)
    sd = he_scaled(net.store.state_dict())
    arenas = {}
    for mode in ("engine", "sinks"):
        gs.ENABLED = F.RES_LINK = (mode == "sinks")
        try:
            net.store.load_state_dict(sd)
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
                net.loss_and_grads(x, y, 0.75, drop_seed=11)
            adds = sum(e.count for e in prof.key_averages() if e.key in ("aten::add", "aten::add_"))
            arenas[mode] = (net.store.grad_arena.clone(), adds)
        finally:
            gs.ENABLED = F.RES_LINK = True
    (g0, adds0), (g1, adds1) = arenas["engine"], arenas["sinks"]
    print("engine: %d aten::add launches, sinks: %d; max |diff| / max |g| = %.3e" % (adds0, adds1, _rel(g1, g0)))
    assert adds0 > 100 and adds1 <= 2
    assert _rel(g1, g0) < 2e-6
    worst = max(_rel(g1[v.offset:v.offset + v.numel], g0[v.offset:v.offset + v.numel]) for v in net.store.trainable())
    assert worst < 1e-4, worst
