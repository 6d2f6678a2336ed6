"""-m gpu: whole-network parity of the segmenter training step (config 2 at reduced batch) vs the CPU oracle:
logits, bit-exact argmax label map (adjudicated by the fp64 oracle on near-ties), loss, every gradient, post-Adam weights."""
import numpy as np
import pytest
import torch

from conftest import pkg
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


@pytest.mark.parametrize("keep_prob", [1.0, 0.75])
def test_segmenter_train_step_parity(dev, keep_prob):
    ss = pkg("source_segmenter")
    B = 2
    rng = np.random.default_rng(0)
    x = rng.standard_normal((B, 256, 256, 3)).astype(np.float32)
    y = T.label_decomp(5, _blob_labels(rng, B))
    net = ss.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, cost_kwargs=dict(COST), seed=3)
    # larger-than-init weights so that logits have realistic margins (init stddev .01 gives ~0 logits)
    sd = net.store.state_dict()
    for k in sd:
        if "/Variable" in k:
            sd[k] = (sd[k] * (np.sqrt(2.0 / (sd[k].shape[0] * sd[k].shape[1] * sd[k].shape[2])) / 0.01 * 0.9)).astype(np.float32)
    net.store.load_state_dict(sd)
    V = nets.make_variables(sd)
    tr = ss.Trainer(net, None, None, num_cls=5, batch_size=B, optimizer="adam", opt_kwargs={"learning_rate": 1e-3})
    tr.opt = tr._get_optimizer(10)

    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    loss = tr.train_step(xd, yd, keep_prob, step=0)       # drop seed = step+1 = 1
    logits = net.logits.detach().cpu()
    grads = {v.name: v.tensor.grad.detach().cpu().clone() for v in net.store.trainable()}

    opt_state = {}
    cost_o, grads_o, logits_o = nets.segmenter_train_step(V, opt_state, torch.from_numpy(x), torch.from_numpy(y), keep_prob, seed=1,
                                                          lr=1e-3, t=1)
    e_logits = _rel(logits, logits_o)
    print("logits rel err", e_logits, "loss", float(loss), float(cost_o))
    assert e_logits < 1e-4
    assert abs(float(loss) - float(cost_o)) < 1e-4 * max(1.0, abs(float(cost_o)))

    # argmax label map: bit exact except where the oracle's own top-2 margin is within fp32 noise
    lab = logits.argmax(-1)
    lab_o = logits_o.argmax(-1)
    top2 = torch.topk(logits_o.double(), 2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1])
    mism = lab != lab_o
    print("argmax mismatches", int(mism.sum()), "of", lab.numel(), "min margin at mismatches",
          float(margin[mism].max()) if mism.any() else None)
    assert int((mism & (margin > 1e-4 * logits_o.abs().max())).sum()) == 0

    worst = 0.0
    for k, g in grads_o.items():
        e = _rel(grads[k], g)
        worst = max(worst, e)
        assert e < 2e-3, (k, e)
    print("worst grad rel err", worst)
    # post-step weights
    after = net.store.state_dict()
    wworst = 0.0
    for k, v in V.items():
        if k.endswith("moving_mean") or k.endswith("moving_variance"):
            e = _rel(after[k], v.detach())
            assert e < 1e-3, (k, e)
            continue
        d_ref = (v.detach() - torch.from_numpy(sd[k])).double()
        d_got = torch.from_numpy(after[k] - sd[k]).double()
        # Adam's first step moves every weight by ~lr*sign(g): compare the updates where |g| is not ~0
        big = grads_o[k].abs() > 1e-3 * grads_o[k].abs().max()
        if big.any():
            e = float((d_ref - d_got)[big].abs().max() / 1e-3)
            wworst = max(wworst, e)
            assert e < 5e-2, (k, e)
    print("worst post-step update err (fraction of lr)", wworst)
