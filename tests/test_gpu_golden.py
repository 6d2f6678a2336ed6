"""-m gpu: the HIP path against the committed golden fixtures (reference graph executed by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tf_ops as T
from test_golden_oracle import golden_segmenter_state

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_hip_segmenter_forward_vs_reference_graph_golden(dev):
    z = np.load(os.path.join(HERE, "golden", "golden.npz"))
    meta = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    ss = pkg("source_segmenter")
    net = ss.Full_DRN(channels=3, n_class=5, batch_size=2, device=dev, seed=meta["seg_seed"],
                      cost_kwargs={"cross_flag": True, "miu_cross": 1.0, "dice_flag": True, "miu_dice": 1.0, "regularizer": 1e-4})
    net.store.load_state_dict(golden_segmenter_state(meta))
    x = np.random.default_rng(21).standard_normal((2, 256, 256, 3)).astype(np.float32)
    y = T.label_decomp(5, z["seg_label"].astype(np.float32))
    net.evaluate(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), keep_prob=1.0, main_bn=True, adapt_bn=True)
    lg = net.logits.cpu().numpy()
    err = np.abs(lg[:, ::8, ::8, :] - z["seg_logits_sub"]).max() / np.abs(z["seg_logits_sub"]).max()
    mism = int((net.compact_pred.cpu().numpy() != z["seg_argmax"]).sum())
    print("golden logits rel err %.3e, argmax mismatches %d / %d" % (err, mism, z["seg_argmax"].size))
    assert err < 1e-4
    assert mism <= 2          # bit-exact label map up to fp32 near-ties (the fp64 adjudication lives in test_gpu_segmenter.py)
    s = meta["seg_scalars"]
    assert abs(float(net.cost) - s["cost"]) < 1e-4
    assert abs(float(net.regularizer_loss) - s["reg"]) < 1e-5 * s["reg"]
    assert abs(float(net.dice_eval) - s["dice_eval"]) < 1e-4


def test_hip_ps_vs_reference_op_sequence_golden(dev):
    z = np.load(os.path.join(HERE, "golden", "golden.npz"))
    meta = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    K = pkg("kernels")
    for tag in ("ps_a", "ps_b", "ps_c"):
        B, a, b, r, nc = meta[tag]
        x = np.arange(B * a * b * nc * r * r, dtype=np.float32).reshape(B, a, b, nc * r * r)
        y = K.ps_fwd(torch.from_numpy(x).to(dev), r, nc).cpu().numpy()
        assert np.array_equal(y, z[tag + "_out"]), tag
