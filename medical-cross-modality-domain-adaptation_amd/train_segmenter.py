"""Entry point mirroring the reference's train_segmenter.py (same hard-coded config dicts, train_segmenter.py:22-80): trains the
source segmenter.  Extra flags (defaults keep the reference behaviour): --synthetic N writes N synthetic tfrecords and trains on
them, --batch-size, --iters, --epochs, --output.  Launched under `python -m torch.distributed.run --nproc-per-node N` it trains data-parallel:
one process per GPU over RCCL, --batch-size slices PER RANK, file lists sharded by rank, rank 0 writes the checkpoint.
  python -m "medical-cross-modality-domain-adaptation_amd.train_segmenter" --synthetic 8 --batch-size 4 --iters 2 --epochs 1
"""
import argparse
import logging
import os

from . import source_segmenter as drn
from .lib import _read_lists
from .parallel import GradReducer, barrier, enable_sync_stats, init_distributed

logging.basicConfig(level=logging.INFO)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--batch-size", type=int, default=10)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--epochs", type=int, default=5000)
    ap.add_argument("--output", default="./tmp_exps/mr_baseline")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--sync-stats", action="store_true", help="data parallel only: all-reduce BN statistics and loss normalisers "
                    "(exactly the single-GPU step on the concatenated batch; ~2 tiny collectives per BN layer per pass)")
    ap.add_argument("--restore", action="store_true")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32", help="arithmetic of the convolution operands: f32 = the reference's "
                    "(default); bf16 = BASELINE configs[4]: bf16 MFMA operands, fp32 accumulation / master weights / BN")
    args = ap.parse_args(argv)
    from .functional import set_conv_dtype
    set_conv_dtype(args.dtype)

    train_fid, val_fid = "./lists/mr_train_list", "./lists/mr_val_list"
    output_path = args.output
    num_cls = 5
    batch_size = args.batch_size
    training_iters, epochs = args.iters, args.epochs
    checkpoint_space = 1500
    optimizer = 'adam'
    cost_kwargs = {"cross_flag": True, "miu_cross": 1.0, "dice_flag": True, "miu_dice": 1.0, "regularizer": 1e-4}
    opt_kwargs = {"learning_rate": 1e-3}
    rank, local, world = init_distributed()
    if args.sync_stats:
        enable_sync_stats()
    if os.environ.get("PNP_SAME_DEVICE"):       # test mode: several gloo ranks on one GPU (tests/test_gpu_dp.py)
        local = 0
    device = "cuda:%d" % local if (world > 1 and args.device == "cuda") else args.device
    os.makedirs(output_path, exist_ok=True)

    if args.synthetic:
        from .synthetic import write_dataset
        # next to (not inside) output_path: Trainer.train(restore=False) clears output_path like the reference (source_segmenter.py:416-418)
        data_root = output_path.rstrip("/") + "_data"
        if rank == 0:
            write_dataset(os.path.join(data_root, "synthetic_train"), args.synthetic, seed=0)
            write_dataset(os.path.join(data_root, "synthetic_val"), max(batch_size, args.synthetic // 4), seed=100)
        barrier()
        train_list = _read_lists(os.path.join(data_root, "synthetic_train", "slice_list"))
        val_list = _read_lists(os.path.join(data_root, "synthetic_val", "slice_list"))
    else:
        train_list, val_list = _read_lists(train_fid), _read_lists(val_fid)
        if not train_list:
            raise SystemExit("no training list at %s (use --synthetic N)" % train_fid)

    net = drn.Full_DRN(channels=3, batch_size=batch_size, n_class=num_cls, cost_kwargs=cost_kwargs, device=device, world_size=world)
    print("Network has been built!")
    trainer = drn.Trainer(net, train_list=train_list, val_list=val_list, num_cls=num_cls, batch_size=batch_size, opt_kwargs=opt_kwargs,
                          checkpoint_space=checkpoint_space, optimizer=optimizer, lr_update_flag=False,
                          reducer=GradReducer(net.store) if world > 1 else None, shard=(rank, world) if world > 1 else None)
    print("Now start training...")
    trainer.train(output_path=output_path, training_iters=training_iters, epochs=epochs, restore=args.restore, restored_path=output_path)
    return trainer


if __name__ == "__main__":
    main()
