"""-m gpu: segmentation loss fwd/bwd, softmax+argmax, dice_eval, Adam/RMSProp/Momentum/clip/L2 — C-ABI vs oracle."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _labels(rng, shape, ncls):
    lab = rng.integers(0, ncls, size=shape)
    lab[rng.random(shape) < 0.6] = 0
    return lab


@pytest.mark.parametrize("shape,scale", [((2, 32, 32), 1.0), ((3, 17, 9), 4.0), ((1, 64, 64), 0.1)])
def test_seg_loss_fwd_bwd(dev, shape, scale):
    K = pkg("kernels")
    rng = np.random.default_rng(0)
    ncls = 5
    z = (rng.standard_normal(shape + (ncls,)) * scale).astype(np.float32)   # scale 4 drives p below the 0.005 clip
    y = T.label_decomp(ncls, _labels(rng, shape, ncls))
    zt = torch.from_numpy(z).requires_grad_(True)
    yt = torch.from_numpy(y)
    wl, dl = T.softmax_weighted_loss(zt, yt), T.dice_loss(zt, yt)
    (1.0 * wl + 1.0 * dl).backward()
    zd, yd = torch.from_numpy(z).to(dev), torch.from_numpy(y).to(dev)
    out, ws = K.seg_loss_fwd(zd, yd, 1.0, 1.0)
    o = out.cpu().numpy()
    assert abs(o[1] - float(wl)) < 1e-5 * max(1, abs(float(wl)))
    assert abs(o[2] - float(dl)) < 1e-5
    assert abs(o[0] - float(wl + dl)) < 2e-5
    dz = K.seg_loss_bwd(zd, yd, ws, 1.0, 1.0, 1.0)
    assert _rel(dz, zt.grad) < 1e-4
    dz_half = K.seg_loss_bwd(zd, yd, ws, 1.0, 1.0, 0.5)
    assert _rel(dz_half, zt.grad * 0.5) < 1e-4


def test_softmax_argmax_and_dice_eval(dev):
    K = pkg("kernels")
    rng = np.random.default_rng(0)
    z = rng.standard_normal((2, 32, 32, 5)).astype(np.float32) * 3
    z[0, 0, 0] = [1.0, 1.0, 0.5, 1.0, 0.0]          # exact tie -> lowest index
    y = T.label_decomp(5, _labels(rng, (2, 32, 32), 5))
    prob, label = K.softmax_argmax(torch.from_numpy(z).to(dev))
    po = T.pixel_wise_softmax_2(torch.from_numpy(z))
    assert _rel(prob, po) < 1e-6
    lo = T.argmax_lowest(po)
    assert int(label[0, 0, 0]) == 0
    assert float((label.cpu() == lo).float().mean()) == 1.0
    out = K.dice_eval(label, torch.from_numpy(y).to(dev)).cpu().numpy()
    dm, arr = T.dice_eval(lo, torch.from_numpy(y), 5)
    assert abs(out[0] - float(dm)) < 1e-6
    assert np.abs(out[1:] - np.array([float(a) for a in arr])).max() < 1e-6
    # Dice of identical masks == 1 for every present class
    out_same = K.dice_eval(torch.from_numpy(np.argmax(y, -1)).to(dev), torch.from_numpy(y).to(dev)).cpu().numpy()
    assert np.abs(out_same[1:] - 1.0).max() < 1e-6


def test_optimizers(dev):
    K = pkg("kernels")
    rng = np.random.default_rng(0)
    n = 5000   # 5 chunks, last one ragged
    w0 = rng.standard_normal(n).astype(np.float32) * 0.05
    l2 = np.array([1e-4, 2e-4, 0.0, 1e-4, 1e-4], np.float32)
    mask = np.array([1, 1, 1, 0, 1], np.uint8)
    l2e = np.repeat(l2, 1024)[:n]
    me = np.repeat(mask, 1024)[:n].astype(bool)
    l2d, md = torch.from_numpy(l2).to(dev), torch.from_numpy(mask).to(dev)

    # Adam, 3 steps
    w = torch.from_numpy(w0.copy()).to(dev); m = torch.zeros_like(w); v = torch.zeros_like(w)
    wo = torch.from_numpy(w0.copy()); mo = torch.zeros_like(wo); vo = torch.zeros_like(wo)
    for t in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        K.adam_step(w, torch.from_numpy(g).to(dev), m, v, l2d, md, 1e-3, 0.9, 0.999, 1e-8, t)
        ge = torch.from_numpy(g) + torch.from_numpy(l2e) * wo
        wn, mn, vn = wo.clone(), mo.clone(), vo.clone()
        T.adam_update(wn, ge, mn, vn, 1e-3, t)
        sel = torch.from_numpy(me)
        wo = torch.where(sel, wn, wo); mo = torch.where(sel, mn, mo); vo = torch.where(sel, vn, vo)
    assert _rel(w, wo) < 1e-6 and _rel(m, mo) < 1e-6 and _rel(v, vo) < 1e-6

    # RMSProp (ms init 1.0) then clip
    w = torch.from_numpy(w0.copy()).to(dev); ms = torch.ones_like(w)
    wo = torch.from_numpy(w0.copy()); mso = torch.ones_like(wo)
    g = rng.standard_normal(n).astype(np.float32)
    K.rmsprop_step(w, torch.from_numpy(g).to(dev), ms, None, None, 3e-4, 0.9, 1e-10)
    T.rmsprop_update(wo, torch.from_numpy(g), mso, 3e-4)
    assert _rel(w, wo) < 1e-6 and _rel(ms, mso) < 1e-6
    w_before = w.cpu().clone()
    K.clip(w, md, -0.03, 0.03)
    wc = torch.where(torch.from_numpy(me), torch.clamp(w_before, -0.03, 0.03), w_before)
    assert torch.equal(w.cpu(), wc)
    assert float(wc[torch.from_numpy(me)].abs().max()) <= 0.03 + 1e-9 and float(w_before.abs().max()) > 0.03

    # Momentum
    w = torch.from_numpy(w0.copy()).to(dev); acc = torch.zeros_like(w)
    wo = torch.from_numpy(w0.copy()); acco = torch.zeros_like(wo)
    for _ in range(2):
        g = rng.standard_normal(n).astype(np.float32)
        K.momentum_step(w, torch.from_numpy(g).to(dev), acc, None, None, 0.2, 0.2)
        T.momentum_update(wo, torch.from_numpy(g), acco, 0.2, 0.2)
    assert _rel(w, wo) < 1e-6

    # L2 loss
    val = float(K.l2_loss(torch.from_numpy(w0).to(dev), l2d).cpu())
    ref = float((l2e.astype(np.float64) * w0.astype(np.float64) ** 2).sum() / 2)
    assert abs(val - ref) < 1e-5 * abs(ref)

    # axpby
    a = torch.from_numpy(w0.copy()).to(dev); b = torch.ones_like(a)
    K.axpby(a, b, 2.0, 0.5)
    assert _rel(b, torch.from_numpy(w0) * 2 + 0.5) < 1e-6


def test_label_decomp_confusion_matrix_and_bn_moments(dev):
    """the label / monitoring helpers of the step path as kernels (lib.py:75-92, source_segmenter.py:83-85) and the SyncBN moment
    conversion, against plain numpy"""
    K, lib = pkg("kernels"), pkg("lib")
    rng = np.random.default_rng(3)
    lab = rng.integers(0, 7, size=(3, 64, 48)).astype(np.float32)          # labels 5, 6 >= num_cls: all-zero rows
    oh = K.label_decomp(torch.from_numpy(lab).to(dev), 5)
    ref = lib._label_decomp(5, lab)
    assert np.array_equal(oh.cpu().numpy(), ref)
    pred = torch.from_numpy(rng.integers(0, 5, size=lab.shape)).to(dev)
    cy, cm = lib.compact_and_confusion(oh, pred)
    cy_ref = ref.argmax(-1)                                                # lowest index on ties: all-zero rows -> class 0
    assert np.array_equal(cy.cpu().numpy(), cy_ref)
    cm_ref = np.zeros((5, 5), np.int64)
    np.add.at(cm_ref, (cy_ref.ravel(), pred.cpu().numpy().ravel()), 1)
    assert np.array_equal(cm, cm_ref) and cm.sum() == lab.size
    cy2, none = lib.compact_and_confusion(oh, None)
    assert none is None and torch.equal(cy2, cy)
    # SyncBN moments: two "replicas" with known statistics -> statistics of the concatenation
    a, b = rng.standard_normal((1000, 8)) * 2 + 1, rng.standard_normal((1000, 8)) * 0.5 - 3
    f = lambda t: torch.from_numpy(t.astype(np.float32)).to(dev)
    mom = K.bn_moments(f(a.mean(0)), f(a.var(0))) + K.bn_moments(f(b.mean(0)), f(b.var(0)))
    mean, var = K.bn_from_moments(mom, 2)
    both = np.concatenate([a, b])
    assert np.allclose(mean.cpu().numpy(), both.mean(0), rtol=1e-6, atol=1e-6) and np.allclose(var.cpu().numpy(), both.var(0), rtol=1e-5)
