"""-m gpu: the HIP path against the committed golden fixtures (reference graph executed by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import nets
from oracle import tf_ops as T
from parity_util import assert_argmax_exact
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
    # bit-exact label map, adjudicated in float64: the fixture holds the float32 oracle's label map, so (a) the float64 oracle must
    # explain every pixel where the HIP path differs from it, and (b) likewise every pixel where the committed map differs
    V64 = nets.make_variables(golden_segmenter_state(meta), dtype=torch.float64, requires_grad=False)
    with torch.no_grad():
        l64 = nets.segmenter_forward(V64, torch.from_numpy(x).double(), 1.0, True, True)
    assert_argmax_exact(net.logits, l64, "HIP label map vs float64 oracle")
    gold = torch.from_numpy(z["seg_argmax"])
    top2 = torch.topk(l64, 2, dim=-1).values
    far = (top2[..., 0] - top2[..., 1]) > 1e-5 * float(l64.abs().max())
    assert bool((gold[far] == l64.argmax(-1)[far]).all()), "committed golden label map disagrees with the float64 oracle away from ties"
    assert bool((net.compact_pred.cpu()[far] == gold[far]).all())
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


def test_hip_adaptation_graph_vs_reference_builders_golden(dev):
    """the HIP path of the whole adaptation graph (both fronts, shared half, feature critic, mask critic, WGAN losses) against the
    fixtures produced by adversarial.py's own builders (keep_prob 1 everywhere, critic BN on batch statistics)"""
    from test_golden_oracle import _name_init
    z = np.load(os.path.join(HERE, "golden", "golden.npz"))
    meta = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    adv = pkg("adversarial")
    ck = meta["adv_cost_kwargs"]
    net = adv.Full_DRN(channels=3, n_class=5, batch_size=2, device=dev, cost_kwargs=dict(ck),
                       network_config={"mr_front_trainable": False, "joint_trainable": False, "ct_front_trainable": True,
                                       "cls_trainable": True, "m_cls_trainable": True})
    sd = net.store.state_dict()
    for k in sd:
        if sd[k].ndim >= 2:
            sd[k] = _name_init(k, sd[k].shape)
    net.store.load_state_dict(sd)
    rng = np.random.default_rng(33)
    mr = torch.from_numpy(rng.standard_normal((2, 256, 256, 3)).astype(np.float32)).to(dev)
    ct = torch.from_numpy((rng.standard_normal((2, 256, 256, 3)) * 1.2 + 0.1).astype(np.float32)).to(dev)
    with torch.no_grad():
        o = net._graph(mr, ct, 1.0, mr_front_bn=False, joint_bn=False, ct_front_bn=True, critic_keep=1.0)
    for tag in ("ct_cls", "mr_cls", "ct_mask", "mr_mask"):
        ref = z["adv_" + tag]
        got = o[tag].cpu().numpy()
        assert got.shape == ref.shape and np.abs(got - ref).max() < 5e-4 * max(1.0, np.abs(ref).max()), (tag, got.ravel(), ref.ravel())
    for br in ("ct", "mr"):
        ref = z["adv_%s_logits_sub" % br]
        assert np.abs(o[br + "_logits"].cpu().numpy()[:, ::16, ::16, :] - ref).max() < 1e-4 * np.abs(ref).max()
    s = meta["adv_scalars"]
    dis = -ck["miu_dis"] * float((o["mr_cls"] - o["ct_cls"]).mean()) - ck["lambda_mask_loss"] * ck["miu_dis"] * float((o["mr_mask"] - o["ct_mask"]).mean())
    gen = -ck["miu_gen"] * float(o["ct_cls"].mean()) - ck["lambda_mask_loss"] * ck["miu_gen"] * float(o["ct_mask"].mean())
    assert abs(dis - s["dis_loss"]) < 1e-6 + 5e-4 * abs(s["dis_loss"]) and abs(gen - s["gen_loss"]) < 1e-6 + 5e-4 * abs(s["gen_loss"])
