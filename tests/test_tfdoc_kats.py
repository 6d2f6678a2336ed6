"""CPU: the oracle (oracle/tf_ops.py) and the product's host-side geometry against the hand-written TensorFlow-documentation known
answers of tests/tfdoc_kats.py — the pin of the TF-1.4 arithmetic that does not go through the oracle itself."""
import numpy as np
import torch

import tfdoc_kats as KAT
from conftest import pkg
from oracle import tf_ops as T


def _nhwc(a2d):
    return torch.from_numpy(a2d).reshape(1, a2d.shape[0], a2d.shape[1], 1)


def test_pad_symmetric_doc_example():
    out = T.pad_symmetric(_nhwc(KAT.PAD_SYM_IN), *KAT.PAD_SYM_PADDINGS)
    assert np.array_equal(out.reshape(4, 7).numpy(), KAT.PAD_SYM_OUT)
    assert T.sym_index(3, 2) == [1, 0, 0, 1, 2, 2, 1]


def test_same_padding_rule_oracle_and_product():
    K = pkg("kernels")
    for (n, k, s, d), want in KAT.SAME_CASES:
        assert T.same_pad(n, k, s, d) == want, (n, k, s, d)
        assert K.same_pad(n, k, s, d) == want, (n, k, s, d)
        g = K.conv_geom((1, n, n, 4), (k, k, 4, 4), s, d, "SAME")          # what the HIP kernels are told
        assert (g.OH, g.OW, g.pad_t, g.pad_l) == (want[0], want[0], want[1], want[1])


def test_same_padding_numeric_instance():
    y = T.conv2d(torch.from_numpy(KAT.SAME_NUMERIC_X), torch.from_numpy(KAT.SAME_NUMERIC_W), stride=2, padding="SAME")
    assert np.array_equal(y.numpy(), KAT.SAME_NUMERIC_Y)


def test_fused_batch_norm_doc_semantics():
    g, b = torch.ones(1), torch.zeros(1)
    mm, mv = torch.zeros(1), torch.ones(1)
    y = T.batch_norm(torch.from_numpy(KAT.BN_X), g, b, mm, mv, True)
    assert np.allclose(y.numpy(), KAT.BN_Y, rtol=1e-6, atol=0)
    assert abs(float(mm) - KAT.BN_MOVING_MEAN) < 1e-7 and abs(float(mv) - KAT.BN_MOVING_VAR) < 1e-6
    yi = T.batch_norm(torch.from_numpy(KAT.BN_X), g, b, mm, mv, False)
    assert np.allclose(yi.numpy(), KAT.BN_Y_INFER, rtol=1e-6, atol=0)
    assert abs(float(mm) - KAT.BN_MOVING_MEAN) < 1e-7            # inference never moves the statistics


def test_gradient_masks():
    t = torch.from_numpy(KAT.CLIP_T).requires_grad_(True)
    torch.clamp(t, 0.005, 1).sum().backward()                   # the form softmax_weighted_loss uses
    assert np.array_equal(t.grad.numpy(), KAT.CLIP_GRAD)
    x = torch.from_numpy(KAT.LRELU_X).requires_grad_(True)
    y = T.leaky_relu(x)
    y.sum().backward()
    assert np.allclose(y.detach().numpy(), KAT.LRELU_Y) and np.allclose(x.grad.numpy(), KAT.LRELU_GRAD)
    p = torch.from_numpy(KAT.POOL_X).requires_grad_(True)
    T.max_pool2(p).sum().backward()
    assert np.array_equal(p.grad.numpy(), KAT.POOL_DX)


def test_optimizer_doc_formulas():
    w, m, v = torch.tensor([KAT.ADAM_W0], dtype=torch.float64), torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)
    g = torch.tensor([KAT.ADAM_G], dtype=torch.float64)
    T.adam_update(w, g, m, v, KAT.ADAM_LR, 1)
    assert abs(float(w) - KAT.ADAM_W1) < 1e-15
    T.adam_update(w, g, m, v, KAT.ADAM_LR, 2)
    assert abs(float(w) - KAT.ADAM_W2) < 1e-15
    w, ms = torch.tensor([KAT.RMS_W0], dtype=torch.float64), torch.ones(1, dtype=torch.float64)
    T.rmsprop_update(w, torch.tensor([KAT.RMS_G], dtype=torch.float64), ms, KAT.RMS_LR)
    assert abs(float(ms) - KAT.RMS_MS1) < 1e-15 and abs(float(w) - KAT.RMS_W1) < 1e-15
    assert float(T.l2_loss(torch.from_numpy(KAT.L2_T))) == KAT.L2_OUT


def test_softmax_and_argmax():
    p = T.pixel_wise_softmax_2(torch.from_numpy(KAT.SOFTMAX_Z))
    assert np.allclose(p.numpy(), KAT.SOFTMAX_P, rtol=1e-6)
    assert np.array_equal(T.argmax_lowest(torch.from_numpy(KAT.ARGMAX_Z)).numpy(), KAT.ARGMAX_OUT)
