"""not gpu: the adaptation model's graph construction on CPU (symbolic pass): TF variable names against the lists the reference
authors recorded (lists/half_zip_*_vars, lists/pred_bn_list via tests/golden/golden.json), variable grouping, optimiser tables."""
import json
import os

import numpy as np
import pytest

from conftest import pkg

HERE = os.path.dirname(os.path.abspath(__file__))
COST = {"regularizer": 1e-4, "gan_regularizer": 1e-4, "miu_gen": 0.002, "miu_dis": 0.002, "lambda_mask_loss": 0.3}
NETCFG = {"mr_front_trainable": False, "joint_trainable": False, "ct_front_trainable": True, "cls_trainable": True, "m_cls_trainable": True}


@pytest.fixture(scope="module")
def net():
    adv = pkg("adversarial")
    return adv.Full_DRN(channels=3, n_class=5, batch_size=2, cost_kwargs=dict(COST), network_config=dict(NETCFG), device="cpu")


def test_variable_names_match_reference_lists(net):
    meta = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    names = list(net.store.vars.keys())
    strip = lambda l: [n.split(":")[0] for n in l]
    mri, ct = strip(meta["list_half_zip_mri_vars"]), strip(meta["list_half_zip_ct_vars"])
    assert len(mri) == len(ct) == 101
    for n in mri + ct:
        assert n in net.store.vars, n
    # positional correspondence used by the MR -> CT weight copy (adversarial.py:706-717): same shapes, same order of creation
    for m, c in zip(mri, ct):
        assert net.store.vars[m].shape == net.store.vars[c].shape, (m, c)
    ours_mr = [n for n in names if n.startswith("group_") and int(n.split("/")[0].split("_")[1]) <= 6]
    ours_ct = [n for n in names if n.startswith("adapt_")]
    assert sorted(ours_mr) == sorted(mri) and sorted(ours_ct) == sorted(ct)
    # BN scopes of the MR path + shared half: pred_bn_list (120 names, recorded without the group_k/ prefix)
    pred = strip(meta["list_pred_bn_list"])
    ours_pred = [n.split("/", 1)[1] for n in names if "/pred_" in n]
    assert sorted(ours_pred) == sorted(pred)
    assert len([n for n in names if n.endswith("/gamma")]) == 30 + 20 + 16 + 8      # segmenter + adapt + cls + mask critic BN layers


def test_variable_groups_and_parameter_counts(net):
    cnt = lambda vs: sum(v.numel for v in vs if "Variable" in v.name)
    assert cnt(net.adapt_vars) == 5087664                                       # early layers 5.09 M
    assert cnt([v for v in net.cls_vars if v.name.startswith("cls_scope")]) == 21729280
    assert cnt([v for v in net.cls_vars if v.name.startswith("mask_cls_scope")]) == 1098448
    assert cnt(net.mri_seg_vars) == 39302456
    assert all(not v.trainable for v in net.mri_seg_vars)                       # mr_front / joint frozen
    assert all(v.trainable for v in net.adapt_vars if not v.name.endswith(("moving_mean", "moving_variance")))
    # weight lists: critics appended once per builder call (CT, MR) -> every name twice; joint_weights stays empty (reference bug)
    wl = net._weight_lists
    assert all(wl["cls_weights"].count(n) == 2 for n in set(wl["cls_weights"])) and len(set(wl["cls_weights"])) == 17
    assert all(wl["m_cls_weights"].count(n) == 2 for n in set(wl["m_cls_weights"])) and len(set(wl["m_cls_weights"])) == 9
    assert all(wl["ct_front_weights"].count(n) == 1 for n in wl["ct_front_weights"]) and len(wl["ct_front_weights"]) == 21
    assert wl["joint_weights"] == []


def test_optimiser_tables(net):
    adv = pkg("adversarial")
    tr = adv.Trainer(net, None, None, None, None, num_cls=5, batch_size=2, opt_kwargs={"learning_rate": 3e-4},
                     train_config={"dis_sub_iter": 20, "gen_sub_iter": 1})
    dis, gen = tr._get_optimizer()
    st = net.store
    chunk = lambda name: st.vars[name].offset // 1024
    l2d, l2g = dis.l2.numpy(), gen.l2.numpy()
    assert np.isclose(l2d[chunk("cls_scope/cls_1/Variable")], 1e-4 * 0.002 * 2 / 20)
    assert np.isclose(l2d[chunk("mask_cls_scope/mask_cls_1/Variable")], 1e-4 * 0.002 * 2 * 0.3 / 20)
    assert l2d[chunk("adapt_1/Variable")] == 0 and l2d[chunk("cls_scope/cls_1/cls_1_1/gamma")] == 0
    assert np.isclose(l2g[chunk("adapt_3/Variable_2")], 1e-4 * 0.002) and l2g[chunk("cls_scope/cls_1/Variable")] == 0
    md, mg, mc = dis.mask.numpy(), gen.mask.numpy(), tr.clip_mask.numpy()
    assert md[chunk("cls_scope/cls_out/Variable")] == 1 and md[chunk("adapt_1/Variable")] == 0
    assert mg[chunk("adapt_1/adapt_1_1/gamma")] == 1 and mg[chunk("mask_cls_scope/m_cls_out/Variable")] == 0
    assert mc[chunk("cls_scope/cls_out/Variable")] == 1 and mc[chunk("mask_cls_scope/mask_cls_4/Variable")] == 1
    assert mc[chunk("cls_scope/cls_1/cls_1_1/gamma")] == 0                      # BN variables are not clipped
    assert float(dis.ms.min()) == 1.0                                           # RMSProp slot initialised to ONE


def test_pretrain_phase_freezes_adaptation_module():
    adv = pkg("adversarial")
    cfg = dict(NETCFG, ct_front_trainable=False)
    n = adv.Full_DRN(channels=3, n_class=5, batch_size=2, cost_kwargs=dict(COST, lambda_mask_loss=0), network_config=cfg, device="cpu")
    assert all(not v.trainable for v in n.adapt_vars)
    assert n.lambda_mask_loss == 0.0
    assert {v.name.split("/")[0] for v in n.store.trainable()} == {"cls_scope", "mask_cls_scope"}


def test_gan_training_schedule_with_stub_steps(tmp_path):
    """adversarial.py:831-946 on the CPU with the two step functions stubbed out: no update at step 0, discriminator before generator,
    sub-iteration growth every `iter_upd_interval` steps, a training + a validation monitoring batch every display_step, learning-rate
    decay with every periodic checkpoint, one fresh batch per (sub-)step."""
    import numpy as np
    import torch
    adv = pkg("adversarial")

    class Src(object):
        def __init__(self, tag):
            self.tag, self.n = tag, 0

        def next_batch(self, B):
            self.n += 1
            b = np.zeros((B, 4, 4, 4), np.float32)
            b[..., 0] = self.n
            return b, ["%s%d" % (self.tag, self.n)] * B

    class Opt(object):
        lr = 1.0

        def state_dict(self):
            return {"ms": np.ones(4, np.float32), "lr": np.float64(self.lr)}

    class Net(object):
        device = torch.device("cpu")
        n_class = 5
        saved = 0

        def evaluate(self, ct, ct_y, mr, mr_y, keep_prob=1.0, detail=False):
            events.append(("eval", detail))
            self.confusion_matrix = np.eye(5)
            return 0.5, 0.5

        def save(self, path):
            Net.saved += 1
            return path

    events = []
    srcs = {k: Src(k) for k in ("mr_t", "mr_v", "ct_t", "ct_v")}
    tr = adv.Trainer(Net(), srcs["mr_t"], srcs["mr_v"], srcs["ct_t"], srcs["ct_v"], num_cls=5, batch_size=2, opt_kwargs={"learning_rate": 1.0},
                     train_config={"dis_interval": 1, "gen_interval": 2, "dis_sub_iter": 2, "gen_sub_iter": 1, "dis_sub_iter_inc": 1,
                                   "gen_sub_iter_inc": 0, "iter_upd_interval": 4, "checkpoint_space": 3, "lr_decay_factor": 0.5})
    tr.dis_optimizer, tr.gen_optimizer = Opt(), Opt()
    tr.dis_step = lambda mr, ct, dropout, seed: events.append(("dis", float(mr[0, 0, 0, 0]), float(ct[0, 0, 0, 0]), seed))
    tr.gen_step = lambda ct, dropout, seed: events.append(("gen", float(ct[0, 0, 0, 0]), seed))
    adv.verbose = False
    try:
        tr.train(str(tmp_path / "o"), restore=False, training_iters=7, epochs=1, display_step=5)
    finally:
        adv.verbose = True
    per_step, cur = [], []
    # rebuild the per-step grouping from the order of events: evals close a display step
    kinds = [e[0] + ("*" if e[0] == "eval" and e[1] else "") for e in events]
    # step 0: only the two monitoring batches; steps 1..6: dis x2 (x3 from step 5 on: growth at step 4 applies afterwards), gen on even steps
    expect = (["eval", "eval*"]                                  # step 0
              + ["dis", "dis"]                                     # 1
              + ["dis", "dis", "gen"]                              # 2
              + ["dis", "dis"]                                     # 3
              + ["dis", "dis", "gen"]                              # 4 (sub-iterations grow AFTER this step's updates)
              + ["dis", "dis", "dis", "eval", "eval*"]            # 5 (display step)
              + ["dis", "dis", "dis", "gen"])                      # 6
    assert kinds == expect, kinds
    seeds = [e[-1] for e in events if e[0] in ("dis", "gen")]
    assert seeds == list(range(seeds[0], seeds[0] + len(seeds)))          # one dropout stream per update
    ct_used = [e[2] for e in events if e[0] == "dis"] + [e[1] for e in events if e[0] == "gen"]
    assert len(set(ct_used)) == len(ct_used)                              # every update sees a fresh CT batch
    assert tr.dis_optimizer.lr == 0.25 and tr.gen_optimizer.lr == 0.25    # checkpoints at steps 3 and 6
    assert Net.saved == 3                                                 # steps 3, 6 and the final one
    # the scalar log (metrics.py): one gan_step row per iteration with the number of updates it ran, eval rows at display steps
    import json
    rows = [json.loads(l) for l in open(str(tmp_path / "o" / "metrics.jsonl"))]
    gs = [r for r in rows if r["kind"] == "gan_step"]
    assert [r["step"] for r in gs] == list(range(7)) and [r["dis_updates"] for r in gs] == [0, 2, 2, 2, 2, 3, 3]
    assert [r["gen_updates"] for r in gs] == [0, 0, 1, 0, 1, 0, 1] and all(r["host_time_s"] >= 0 for r in gs)
    ev = [r for r in rows if r["kind"].endswith("_eval")]
    assert [(r["kind"], r["step"]) for r in ev] == [("train_eval", 0), ("val_eval", 0), ("train_eval", 5), ("val_eval", 5)]
    assert all(r["ct_dice"] == 0.5 and r["dis_loss"] is None for r in ev)


def test_baseline_hand_off_follows_the_reference_lists():
    """train_gan.py --phase pre-train (adversarial.py:706-765) with the reference's OWN lists: every BatchNorm_k of the source segmenter
    lands on the pred_*/adapt-era name at the same list position (group_<n>/ prefix rebuilt from the name like the reference does), conv
    filters go across by name, and the MR early layers are copied onto the CT adaptation module pairwise (half_zip_mri_vars ->
    half_zip_ct_vars)."""
    adv, ss = pkg("adversarial"), pkg("source_segmenter")
    meta = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    old_bn, new_bn = meta["list_old_bn_list"], meta["list_pred_bn_list"]
    mr_vars, ct_vars = meta["list_half_zip_mri_vars"], meta["list_half_zip_ct_vars"]
    assert len(old_bn) == len(new_bn) == 120 and len(mr_vars) == len(ct_vars)
    seg = ss.Full_DRN(channels=3, n_class=5, batch_size=2, device="cpu", cost_kwargs={"regularizer": 1e-4})
    rng = np.random.default_rng(0)
    seg_state = {k: rng.standard_normal(v.shape).astype(np.float32) for k, v in seg.store.state_dict().items()}
    net = adv.Full_DRN(channels=3, n_class=5, batch_size=2, cost_kwargs=dict(COST), network_config=dict(NETCFG), device="cpu")
    before = net.store.state_dict()
    net.load_baseline(seg_state, old_bn, new_bn, ct_vars, mr_vars)
    after = net.store.state_dict()
    strip = lambda n: n.split(":")[0]
    for o, n in zip(old_bn, new_bn):
        n = strip(n)
        tgt = "group_" + n.split("_")[1] + "/" + n                     # adversarial.py:753-755
        assert tgt in after, tgt
        assert np.array_equal(after[tgt], seg_state[strip(o)]), (o, tgt)
    for k in seg_state:
        if "/Variable" in k:
            assert np.array_equal(after[k], seg_state[k]), k               # same TF name in both graphs
    for m, a in zip(mr_vars, ct_vars):
        assert np.array_equal(after[strip(a)], after[strip(m)]), (m, a)
    untouched = [k for k in after if "cls" in k]
    assert untouched and all(np.array_equal(after[k], before[k]) for k in untouched)        # critics keep their initialisation
    with pytest.raises(KeyError):
        net.load_baseline(seg_state, ["BatchNorm_999/beta:0"], ["pred_7_1/beta:0"], ct_vars, mr_vars)
