"""Entry point mirroring the reference's train_gan.py (config dicts and --phase switch, train_gan.py:27-157).
  --phase pre-train : warm up the feature critic (dis only, 1 sub-iteration, lambda_mask_loss = 0, CT front frozen)
  --phase train-gan : joint training (20 dis : 1 gen, lambda_mask_loss = rate, dis_sub_iter += 1 every 300 steps)
  --phase fine-tune : continue from a breakpoint (the reference's `training_config` NameError at train_gan.py:121 is fixed)
Extra flags: --synthetic N, --batch-size, --iters, --epochs, --output, --baseline (source-segmenter .npz for the pre-train hand-off).
"""
import argparse
import datetime
import logging
import os

import numpy as np

from . import adversarial as drn
from .lib import _read_lists
from .parallel import GradReducer, barrier, enable_sync_stats, init_distributed

logging.basicConfig(level=logging.INFO)
rate = 0.3
date = datetime.datetime.now().strftime('%m%d')

cost_kwargs = {"regularizer": 1e-4, "gan_regularizer": 1e-4, "miu_gen": 0.002, "miu_dis": 0.002, "lambda_mask_loss": None}
opt_kwargs = {"learning_rate": 3e-4}
network_config = {"mr_front_trainable": False, "joint_trainable": False, "ct_front_trainable": None, "cls_trainable": True,
                  "m_cls_trainable": True, "restore_skip_kwd": ["Adam", "RMS", "cls"]}
train_config = {"restore_from_baseline": None, "copy_main": None, "clear_rms": None, "lr_update": None, "dis_interval": 1, "gen_interval": 1,
                "dis_sub_iter": 20, "gen_sub_iter": 1, "tag": "gan-" + str(rate) + "_" + date, "iter_upd_interval": 300, "dis_sub_iter_inc": 1,
                "gen_sub_iter_inc": 0, "lr_decay_factor": 0.98, "checkpoint_space": 100, "training_iters": 200, "epochs": 600}


def configure(phase):
    """train_gan.py:85-126"""
    ck, nc, tc = dict(cost_kwargs), dict(network_config), dict(train_config)
    if phase == 'pre-train':
        nc["ct_front_trainable"] = False
        tc.update(restore_from_baseline=True, copy_main=True, clear_rms=True, lr_update=True, gen_interval=0, dis_sub_iter=1,
                  dis_sub_iter_inc=0, checkpoint_space=2000, training_iters=201, epochs=100)
        ck["lambda_mask_loss"] = 0
    elif phase == 'train-gan':
        nc["ct_front_trainable"] = True
        tc.update(restore_from_baseline=False, copy_main=False, clear_rms=False, lr_update=True, tag=tc["tag"] + "-gan")
        ck["lambda_mask_loss"] = rate
    elif phase == 'fine-tune':
        nc["ct_front_trainable"] = True
        tc.update(restore_from_baseline=False, copy_main=False, clear_rms=False, lr_update=False, gen_interval=1, dis_sub_iter=30,
                  tag=tc["tag"] + "-fine_tune")
        ck["lambda_mask_loss"] = rate
    else:
        raise Exception("Please set a training phase!")
    return ck, nc, tc


def main(phase, argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--phase", default=phase)
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--batch-size", type=int, default=6)
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--output", default="./tmp_exps/mr2ct" + date)
    ap.add_argument("--baseline", default=None)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--sync-stats", action="store_true", help="data parallel only: all-reduce BN statistics and loss normalisers "
                    "(exactly the single-GPU step on the concatenated batch; ~2 tiny collectives per BN layer per pass)")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32", help="arithmetic of the convolution operands: f32 = the reference's "
                    "(default); bf16 = BASELINE configs[4]: bf16 MFMA operands, fp32 accumulation / master weights / BN")
    args = ap.parse_args(argv)
    from .functional import set_conv_dtype
    set_conv_dtype(args.dtype)
    ck, nc, tc = configure(args.phase)
    num_cls, batch_size = 5, args.batch_size
    output_path = args.output
    rank, local, world = init_distributed()     # >1 only under torch.distributed.run: data-parallel, --batch-size slices per rank
    if args.sync_stats:
        enable_sync_stats()
    if os.environ.get("PNP_SAME_DEVICE"):       # test mode: several gloo ranks on one GPU (tests/test_gpu_dp.py)
        local = 0
    device = "cuda:%d" % local if (world > 1 and args.device == "cuda") else args.device
    os.makedirs(output_path, exist_ok=True)
    if args.synthetic:
        from .synthetic import write_dataset
        sets = (("syn_mr_train", args.synthetic, 0, "mr"), ("syn_ct_train", args.synthetic, 1, "ct"),
                ("syn_mr_val", batch_size, 100, "mr"), ("syn_ct_val", batch_size, 101, "ct"))
        if rank == 0:
            for folder, n, seed, prefix in sets:
                write_dataset(os.path.join(output_path, folder), n, seed=seed, prefix=prefix)
        barrier()
        mr_train, ct_train, mr_val, ct_val = [_read_lists(os.path.join(output_path, folder, prefix + "_list")) for folder, _, _, prefix in sets]
    else:
        mr_train, mr_val = _read_lists("./lists/mr_train_list"), _read_lists("./lists/mr_val_list")
        ct_train, ct_val = _read_lists("./lists/ct_train_list"), _read_lists("./lists/ct_val_list")
        if not mr_train or not ct_train:
            raise SystemExit("no ./lists/*_train_list found (use --synthetic N)")
    adapt_var_list, mr_var_list = _read_lists("./lists/half_zip_ct_vars"), _read_lists("./lists/half_zip_mri_vars")
    old_bn_list, new_bn_list = _read_lists("./lists/old_bn_list"), _read_lists("./lists/pred_bn_list")

    net = drn.Full_DRN(channels=3, batch_size=batch_size, n_class=num_cls, cost_kwargs=ck, network_config=nc, device=device, world_size=world)
    print("Network has been built ...")
    if tc["restore_from_baseline"] and not args.baseline:
        # the reference cannot get past this point either: _load_batch_norm_weights / restore raise without a baseline checkpoint
        # (adversarial.py:706-765, 790-801); warming a critic up against a randomly initialised segmenter is never what was asked for
        raise SystemExit("--phase %s starts from the source segmenter: pass --baseline <train_segmenter output>/checkpoint.npz" % args.phase)
    if tc["restore_from_baseline"]:
        with np.load(args.baseline) as z:
            seg = {k.replace("|", "/"): z[k] for k in z.files}
        net.load_baseline(seg, old_bn_list, new_bn_list, adapt_var_list, mr_var_list)
        print("initializing from baseline model!")
    trainer = drn.Trainer(net, mr_train, mr_val, ct_train, ct_val, adapt_var_list=adapt_var_list, mr_var_list=mr_var_list,
                          old_bn_list=old_bn_list, new_bn_list=new_bn_list, num_cls=num_cls, batch_size=batch_size, opt_kwargs=dict(opt_kwargs),
                          train_config=tc, reducer=GradReducer(net.store) if world > 1 else None, shard=(rank, world) if world > 1 else None)
    print("Now start training...")
    trainer.train(output_path=output_path, restored_path=output_path, restore=not tc["restore_from_baseline"],
                  training_iters=args.iters or tc["training_iters"], epochs=args.epochs or tc["epochs"])
    return trainer


if __name__ == "__main__":
    import sys
    ph = "train-gan"
    for i, a in enumerate(sys.argv):
        if a == "--phase" and i + 1 < len(sys.argv):
            ph = sys.argv[i + 1]
    main(phase=ph)
