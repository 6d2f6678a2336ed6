"""-m gpu: the HIP kernels against the hand-written TensorFlow-documentation known answers (tests/tfdoc_kats.py).  These do not pass
through oracle/tf_ops.py: they hold libpnp_hip.so directly to TF's published semantics (tf.pad SYMMETRIC, the SAME rule, fused
batch norm incl. the Bessel-corrected moving variance, the tie rules of the gradients, Adam / RMSProp / l2_loss / argmax)."""
import numpy as np
import pytest
import torch

import tfdoc_kats as KAT
from conftest import pkg

pytestmark = pytest.mark.gpu


def _d(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_hip_sympad_doc_example(dev):
    K = pkg("kernels")
    x = _d(KAT.PAD_SYM_IN.reshape(1, 2, 3, 1), dev)
    # pnp_sympad pads H and W by the same amount; the doc example pads H by 1 and W by 2: check the W pattern with p = 2 on a
    # tensor tall enough, and the H pattern with p = 1
    xp2 = K.sympad_fwd(torch.cat([x, x], 1), 2).cpu().numpy()[0, 2:4, :, 0]       # rows 0-1 of the (doubled) image, padded by 2 in W
    assert np.array_equal(xp2, KAT.PAD_SYM_OUT[1:3])
    xp1 = K.sympad_fwd(x, 1).cpu().numpy()[0, :, :, 0]                            # p = 1: [[1 1 2 3 3],[1 1 2 3 3],[4 4 5 6 6],[4 4 5 6 6]]
    assert np.array_equal(xp1, KAT.PAD_SYM_OUT[:, 1:6])
    # the mirror folded into the convolution's gather (PNP_PAD_SYMMETRIC) reads the same samples: 3x3 box filter == sums of the padded image
    g = K.conv_geom((1, 2, 3, 1), (3, 3, 1, 1), 1, 1, "SYMMETRIC")
    y = K.conv2d_fwd(x, _d(np.ones((3, 3, 1, 1), np.float32), dev), g).cpu().numpy()[0, :, :, 0]
    P = KAT.PAD_SYM_OUT[:, 1:6]
    want = np.array([[P[i:i + 3, j:j + 3].sum() for j in range(3)] for i in range(2)], np.float32)
    assert np.array_equal(y, want)


def test_hip_same_padding_numeric_instance(dev):
    K = pkg("kernels")
    g = K.conv_geom(KAT.SAME_NUMERIC_X.shape, KAT.SAME_NUMERIC_W.shape, 2, 1, "SAME")
    y = K.conv2d_fwd(_d(KAT.SAME_NUMERIC_X, dev), _d(KAT.SAME_NUMERIC_W, dev), g)
    assert np.array_equal(y.cpu().numpy(), KAT.SAME_NUMERIC_Y)
    # the same asymmetry on the MFMA path (C % 32 == 0): channel 0 carries the instance, the other 31 channels are zero
    x = np.zeros((1, 1, 4, 32), np.float32)
    x[..., 0] = KAT.SAME_NUMERIC_X[..., 0]
    w = np.zeros((1, 3, 32, 32), np.float32)
    w[:, :, 0, 5] = KAT.SAME_NUMERIC_W[:, :, 0, 0]
    g = K.conv_geom(x.shape, w.shape, 2, 1, "SAME")
    y = K.conv2d_fwd(_d(x, dev), _d(w, dev), g).cpu().numpy()
    assert np.array_equal(y[0, 0, :, 5], KAT.SAME_NUMERIC_Y[0, 0, :, 0]) and float(np.abs(y).sum()) == 364.0
    # data gradient of the strided instance: dx = W^T dy with the same (0, 1) padding; dy = (1, 1) -> dx = [1, 10, 101, 10]
    dy = np.ones((1, 1, 2, 32), np.float32) * (np.arange(32) == 5)
    dx = K.conv2d_dgrad(_d(dy.astype(np.float32), dev), _d(w, dev), g).cpu().numpy()
    assert np.array_equal(dx[0, 0, :, 0], np.array([1, 10, 101, 10], np.float32))


def test_hip_fused_batch_norm_doc_semantics(dev):
    K = pkg("kernels")
    x = _d(KAT.BN_X, dev)
    mean, var = K.bn_stats(x)
    assert abs(float(mean) - 2.5) < 1e-6 and abs(float(var) - 1.25) < 1e-6
    mm, mv = torch.zeros(1, device=dev), torch.ones(1, device=dev)
    K.bn_update_moving(mm, mv, mean, var, 4, 0.9)
    assert abs(float(mm) - KAT.BN_MOVING_MEAN) < 1e-7 and abs(float(mv) - KAT.BN_MOVING_VAR) < 1e-6
    g, b = torch.ones(1, device=dev), torch.zeros(1, device=dev)
    y = K.bn_apply(x, mean, var, g, b, None, 1e-3, -1.0)
    assert np.allclose(y.cpu().numpy(), KAT.BN_Y, rtol=2e-6, atol=0)
    yi = K.bn_apply(x, mm, mv, g, b, None, 1e-3, -1.0)
    assert np.allclose(yi.cpu().numpy(), KAT.BN_Y_INFER, rtol=2e-6, atol=0)


def test_hip_gradient_tie_rules(dev):
    K = pkg("kernels")
    # leaky-ReLU fused behind BN: identity BN (mean 0, var 1 - eps, gamma 1) so that out = lrelu(x) and dx = dout * slope(x)
    x = _d(KAT.LRELU_X.reshape(3, 1, 1, 1), dev)
    mean, var = torch.zeros(1, device=dev), torch.full((1,), 1.0 - 1e-3, device=dev)
    g, b = torch.ones(1, device=dev), torch.zeros(1, device=dev)
    out = K.bn_apply(x, mean, var, g, b, None, 1e-3, 0.2)
    assert np.allclose(out.cpu().numpy().ravel(), KAT.LRELU_Y, rtol=1e-6)
    dx, _ = K.bn_bwd_apply(torch.ones_like(x), out, x, mean, var, g, None, 3, 0, 1e-3, 0.2, False)
    assert np.allclose(dx.cpu().numpy().ravel(), KAT.LRELU_GRAD, rtol=1e-6)
    # max-pool ties: first maximal element in row-major order
    dxp = K.maxpool2_bwd(_d(KAT.POOL_X, dev), torch.ones((1, 1, 1, 1), device=dev))
    assert np.array_equal(dxp.cpu().numpy(), KAT.POOL_DX)
    # clip(p, .005, 1) inside the weighted cross-entropy: the xent gradient vanishes for p < .005 and survives at p >= .005.
    # Five classes, two pixels; pixel 0 is labelled class 0 with p(class 0) = .001 / .5, pixel 1 (class 1) keeps two classes populated;
    # Dice term off
    import math
    for p0, alive in ((0.001, False), (0.5, True)):
        z = np.zeros((2, 5), np.float32)
        z[0, 0] = math.log(4.0 * p0 / (1.0 - p0))             # exp(z) / (exp(z) + 4) = p0
        yv = np.zeros((2, 5), np.float32)
        yv[0, 0] = yv[1, 1] = 1
        zl, yl = _d(z.reshape(1, 1, 2, 5), dev), _d(yv.reshape(1, 1, 2, 5), dev)
        out, ws = K.seg_loss_fwd(zl, yl, 1.0, 0.0)
        dz = K.seg_loss_bwd(zl, yl, ws, 1.0, 0.0).cpu().numpy().reshape(2, 5)
        assert (float(np.abs(dz[0]).max()) > 1e-3) == alive, (p0, dz)


def test_hip_optimizer_doc_formulas(dev):
    K = pkg("kernels")
    n = 1024
    w = torch.full((n,), KAT.ADAM_W0, device=dev)
    g = torch.full((n,), KAT.ADAM_G, device=dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    K.adam_step(w, g, m, v, None, None, KAT.ADAM_LR, 0.9, 0.999, 1e-8, 1)
    assert abs(float(w[0]) - KAT.ADAM_W1) < 1.2e-7 and float(w.std()) == 0.0
    K.adam_step(w, g, m, v, None, None, KAT.ADAM_LR, 0.9, 0.999, 1e-8, 2)
    assert abs(float(w[0]) - KAT.ADAM_W2) < 2e-7
    w = torch.full((n,), KAT.RMS_W0, device=dev)
    ms = torch.ones(n, device=dev)
    K.rmsprop_step(w, torch.full((n,), KAT.RMS_G, device=dev), ms, None, None, KAT.RMS_LR)
    assert abs(float(ms[0]) - KAT.RMS_MS1) < 2e-7 and abs(float(w[0]) - KAT.RMS_W1) < 1.2e-7
    t = torch.zeros(n, device=dev)
    t[:3] = _d(KAT.L2_T, dev)
    assert float(K.l2_loss(t, torch.ones(1, device=dev))) == KAT.L2_OUT


def test_hip_softmax_and_argmax(dev):
    K = pkg("kernels")
    z = np.zeros((1, 1, 2, 5), np.float32)
    z[0, 0] = KAT.ARGMAX_Z
    prob, lab = K.softmax_argmax(_d(z, dev))
    assert np.array_equal(lab.cpu().numpy().ravel(), KAT.ARGMAX_OUT)          # lowest index on ties
    z2 = _d(KAT.SOFTMAX_Z.reshape(1, 1, 1, 2), dev)
    p2, _ = K.softmax_argmax(z2)
    assert np.allclose(p2.cpu().numpy().ravel(), KAT.SOFTMAX_P, rtol=1e-6)
