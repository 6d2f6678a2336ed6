"""-m gpu: the reference's entry points end to end on synthetic tfrecords (BASELINE config 1 plumbing, on the HIP path):
train_segmenter.py -> checkpoint -> train_gan.py --phase pre-train (baseline hand-off: BN rename + MR->CT copy) -> --phase train-gan."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def test_three_phase_chain(dev, tmp_path):
    ts, tg = pkg("train_segmenter"), pkg("train_gan")
    out1 = str(tmp_path / "seg")
    tr = ts.main(["--synthetic", "6", "--batch-size", "2", "--iters", "3", "--epochs", "1", "--output", out1])
    assert len(tr.step_times) == 3 and np.isfinite(tr.loss_dict["train"][1]) and np.isfinite(tr.loss_dict["val"][2])
    ck = os.path.join(out1, "checkpoint.npz")
    assert os.path.exists(ck)
    seg_state = tr.net.store.state_dict()

    out2 = str(tmp_path / "gan")
    t2 = tg.main("pre-train", ["--synthetic", "4", "--batch-size", "2", "--iters", "3", "--epochs", "1", "--output", out2, "--baseline", ck])
    st = t2.net.store.state_dict()
    # hand-off: conv weights by name, BatchNorm_k -> pred_* (positional), MR early layers copied onto adapt_*
    assert np.array_equal(st["group_7/Variable_2"], seg_state["group_7/Variable_2"])
    assert np.array_equal(st["group_1/pred_1_1_1/moving_variance"], seg_state["BatchNorm/moving_variance"])
    assert np.array_equal(st["group_9/pred_9_2/gamma"], seg_state["BatchNorm_29/gamma"])
    assert np.array_equal(st["adapt_3/Variable_1"], seg_state["group_3/Variable_1"])
    assert np.array_equal(st["adapt_1/adapt_1_1/beta"], seg_state["BatchNorm/beta"])
    assert t2.global_step == 2                              # steps 1,2 run the critic (step 0 is skipped like the reference)
    assert float(np.abs(st["cls_scope/cls_1/Variable"]).max()) <= 0.03 + 1e-9     # clipped
    assert np.isfinite(float(t2.net.dis_loss))

    t3 = tg.main("train-gan", ["--synthetic", "4", "--batch-size", "2", "--iters", "2", "--epochs", "1", "--output", out2])
    # step 1: 20 critic sub-iterations + 1 generator update, on top of the 2 steps of the pre-train phase: global_step and the critics'
    # RMSProp slots travel with the checkpoint (tf.train.Saver restores them by name; clear_rms is False in this phase)
    assert t3.global_step == 23
    assert np.isfinite(float(t3.net.ct_gen_loss)) and np.isfinite(float(t3.net.dis_loss))
    moved = t3.net.store.state_dict()
    assert not np.array_equal(moved["adapt_1/Variable"], st["adapt_1/Variable"])   # the adaptation module trains in this phase
    assert np.array_equal(moved["group_7/Variable"], st["group_7/Variable"])       # the shared segmenter half never does


def test_entry_point_dtype_flag_selects_the_bf16_kernels(dev, tmp_path):
    """--dtype bf16 (BASELINE configs[4]) at the entry point: the convolution launches of the training loop are observed on the bf16
    symbols (pnp_prof_*), the run is finite, and a later run without the flag is back on the fp32 kernels (the flag is per main())."""
    ts, L, K = pkg("train_segmenter"), pkg("_lib"), pkg("kernels")
    L.prof_summary()
    L.prof_enable(L.PROF_CONV_FWD)
    try:
        tr = ts.main(["--synthetic", "4", "--batch-size", "2", "--iters", "2", "--epochs", "1", "--output", str(tmp_path / "b"), "--dtype", "bf16"])
        torch.cuda.synchronize()
        names = [r["name"] for r in L.prof_summary()]
        assert K.CONV_DTYPE == L.DTYPE_BF16
        assert any("bf16" in n for n in names), sorted(set(names))[:8]
        assert np.isfinite(tr.loss_dict["train"][1])
        ts.main(["--synthetic", "4", "--batch-size", "2", "--iters", "1", "--epochs", "1", "--output", str(tmp_path / "f")])
        torch.cuda.synchronize()
        names = [r["name"] for r in L.prof_summary()]
        assert K.CONV_DTYPE == L.DTYPE_F32 and names and not any("bf16" in n for n in names), sorted(set(names))[:8]
    finally:
        L.prof_enable(0)
        pkg("functional").set_conv_dtype("f32")
