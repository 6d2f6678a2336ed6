"""-m gpu: Trainer.test_eval (volume inference) on small synthetic NIfTI volumes for both trainers: the reassembled
label volume equals the per-slice hard predictions of the same network in inference mode, the confusion matrix equals the one
recomputed on the host from those predictions (zero-filled slots included), critic BN statistics are not touched."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu

COST = {"regularizer": 1e-4, "gan_regularizer": 1e-4, "miu_gen": 0.002, "miu_dis": 0.002, "lambda_mask_loss": 0.3}
NETCFG = {"mr_front_trainable": False, "joint_trainable": False, "ct_front_trainable": True, "cls_trainable": True, "m_cls_trainable": True}


def _he(net, seed):
    rng = np.random.default_rng(seed)
    sd = net.store.state_dict()
    for k, a in sd.items():
        if "Variable" in k:
            sd[k] = (rng.standard_normal(a.shape) * np.sqrt(2.0 / np.prod(a.shape[:-1])) * 0.9).astype(np.float32)
        elif k.endswith("moving_mean"):
            sd[k] = (0.05 * rng.standard_normal(a.shape)).astype(np.float32)
        elif k.endswith("moving_variance"):
            sd[k] = (1.0 + 0.2 * rng.random(a.shape)).astype(np.float32)
    net.store.load_state_dict(sd)


def _volume(tmp_path, depth, seed):
    L = pkg("lib")
    rng = np.random.default_rng(seed)
    raw = rng.standard_normal((256, 256, depth)).astype(np.float32)
    lab = np.zeros((256, 256, depth), np.float32)
    for k in range(depth):
        for c in range(1, 5):
            cy, cx = rng.integers(40, 216, 2)
            lab[cy - 20:cy + 20, cx - 15:cx + 15, k] = c
    aff = np.diag([1.25, 1.25, 2.0, 1.0])
    return (L.write_nii(raw, "img_%d.nii.gz" % seed, str(tmp_path), affine=aff), L.write_nii(lab, "lab_%d.nii.gz" % seed, str(tmp_path), affine=aff),
            raw, lab)


def _expected(predict_one, raw, lab, B, frames_batches):
    """host recomputation: hard predictions slice by slice, confusion matrix with the zero-filled slots"""
    fr, fy = np.flip(np.flip(raw, 0), 1), np.flip(np.flip(lab, 0), 1)
    vol_pred = np.zeros(fy.shape)
    cm = np.zeros((5, 5))
    for chunk in frames_batches:
        x = np.zeros((B, 256, 256, 3), np.float32)
        y = np.zeros((B, 256, 256), np.float32)
        for i, jj in enumerate(chunk):
            x[i], y[i] = fr[..., jj - 1:jj + 2], fy[..., jj]
        p = predict_one(x)
        for i, jj in enumerate(chunk):
            vol_pred[..., jj] = p[i]
        np.add.at(cm, (y.astype(int).ravel(), p.astype(int).ravel()), 1)
    return vol_pred, cm


def test_segmenter_test_eval(dev, tmp_path):
    ss, L = pkg("source_segmenter"), pkg("lib")
    B = 2
    net = ss.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, cost_kwargs={"cross_flag": True, "miu_cross": 1.0, "dice_flag": True, "miu_dice": 1.0, "regularizer": 1e-4}, seed=3)
    _he(net, 5)
    img, lab, raw, laby = _volume(tmp_path, 6, 0)
    tr = ss.Trainer(net, None, None, num_cls=5, batch_size=B, test_nii_list=[img], test_label_list=[lab], optimizer="adam",
                    opt_kwargs={"learning_rate": 1e-3})
    before = {k: v.copy() for k, v in net.store.state_dict().items()}
    out = str(tmp_path / "out")
    dice, second = tr.test_eval(None, out, flip_correction=True, save_result=True)
    after = net.store.state_dict()
    assert all(np.array_equal(before[k], after[k]) for k in before)         # inference: no variable (incl. BN moving stats) moves

    def predict_one(x):
        with torch.no_grad():
            lg = net.forward(torch.from_numpy(x).to(dev), 1.0, main_bn=False, adapt_bn=False)
        # the graph's compact_pred is the argmax of the softmax OUTPUT (source_segmenter.py:80-81): logits closer than one fp32 ulp of
        # their probabilities tie to the lower class, so the check goes through the same op (its parity: test_gpu_loss_optim.py)
        return pkg("kernels").softmax_argmax(lg.contiguous())[1].cpu().numpy()

    vol_pred, cm = _expected(predict_one, raw, laby, B, [[1, 2], [3, 4], []])
    saved = L.read_nii_object(os.path.join(out, "test_pred", "dense_pred_img_0.nii.gz"))
    assert np.array_equal(saved.get_data(), vol_pred) and np.allclose(saved.get_affine(), np.diag([1.25, 1.25, 2.0, 1.0]))
    gth = L.read_nii_image(os.path.join(out, "test_pred", "gth_dense_pred_img_0.nii.gz"))
    assert np.array_equal(gth, np.flip(np.flip(laby, 0), 1))
    assert np.allclose(dice, L._dice(cm)) and second.shape == (1, 2)
    assert cm.sum() == 3 * B * 256 * 256
    # test_choose_model (source_segmenter.py:666-675): restore a checkpoint, then the same evaluation
    ck = net.save(str(tmp_path / "ck.npz"))
    d2, _ = tr.test_choose_model(ck, str(tmp_path / "out2"))
    assert np.allclose(d2, dice)


def test_adaptation_test_eval(dev, tmp_path):
    adv, L = pkg("adversarial"), pkg("lib")
    B = 2
    net = adv.Full_DRN(channels=3, n_class=5, batch_size=B, cost_kwargs=dict(COST), network_config=dict(NETCFG), device=dev, seed=1)
    _he(net, 7)
    vols = [_volume(tmp_path, 5, s) for s in (1, 2)]
    tr = adv.Trainer(net, None, None, None, None, num_cls=5, batch_size=B, test_nii_list=[v[0] for v in vols],
                     test_label_list=[v[1] for v in vols], opt_kwargs={"learning_rate": 3e-4})
    before = {k: v.copy() for k, v in net.store.state_dict().items()}
    out = str(tmp_path / "out")
    np.random.seed(11)
    dice, second = tr.test_eval(None, out)
    after = net.store.state_dict()
    assert all(np.array_equal(before[k], after[k]) for k in before)         # critics pruned: their BN moving statistics stay put

    def predict_one(x):
        p, _ = net.predict_ct(torch.from_numpy(x).to(dev), torch.zeros((B, 256, 256, 5), device=dev))
        return p.cpu().numpy()

    np.random.seed(11)                                                       # replay the shuffles of the two volumes
    dices, total = [], np.zeros((5, 5))
    for img, lab, raw, laby in vols:
        frames = [1, 2, 3]
        np.random.shuffle(frames)
        _, cm = _expected(predict_one, raw, laby, B, [frames[0:2], frames[2:4]])
        dices.append(L._dice(cm))
        total += cm
    assert np.allclose(dice, np.mean(dices, axis=0)) and second.shape == (1, 2)
    assert np.array_equal(np.loadtxt(os.path.join(out, "cm.csv")), total)
    assert total.sum() == 2 * 2 * B * 256 * 256
    ck = net.save(str(tmp_path / "ck.npz"))
    np.random.seed(11)
    d2, _ = tr.test_model(ck, str(tmp_path / "out2"))                       # adversarial.py:1097-1108
    assert np.allclose(d2, dice)
