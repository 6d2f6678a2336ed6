"""worker for tests/test_gpu_dp.py: 2 ranks.  Default: both on cuda:0 over gloo (RCCL refuses duplicate devices on a 1-GPU box).
PNP_DP_NATIVE=1 (boxes with >= 2 GPUs): one GPU per rank, native RCCL through pnp_comm_* — the production transport.
Checks that the backward-overlapped bucketed all-reduce equals a plain all-reduce of the locally computed gradients."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "medical-cross-modality-domain-adaptation_amd"
par = importlib.import_module(PKG + ".parallel")
ss = importlib.import_module(PKG + ".source_segmenter")

NATIVE = os.environ.get("PNP_DP_NATIVE") == "1"
rank, local, world = par.init_distributed(None if NATIVE else "gloo")
if NATIVE:
    assert par.native_comm() is not None and "rccl-native" in par.transport(), par.transport()
torch.cuda.set_device(local if NATIVE else 0)
dev = torch.device("cuda", local if NATIVE else 0)
B = 2
COST = {"cross_flag": True, "miu_cross": 1.0, "dice_flag": True, "miu_dice": 1.0, "regularizer": 1e-4}
net = ss.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, cost_kwargs=dict(COST), seed=0, world_size=world)
sd = net.store.state_dict()
for k in sd:
    if "/Variable" in k:
        s = sd[k].shape
        sd[k] = (sd[k] * (np.sqrt(2.0 / (s[0] * s[1] * s[2])) / 0.01)).astype(np.float32)
net.store.load_state_dict(sd)
sd0 = net.store.state_dict()      # incl. the BN moving averages: a training-mode forward moves them, and the epilogue statistics are
                                  # accumulated around the moving mean, so "same gradients" means "from the same state"
rng = np.random.default_rng(10 + rank)
x = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dev)
lab = rng.integers(0, 5, size=(B, 256, 256))
y = torch.from_numpy(np.eye(5, dtype=np.float32)[lab]).to(dev)

# reference: local gradients, then one plain all-reduce
net.store.load_state_dict(sd0)
net.loss_and_grads(x, y, 0.75, drop_seed=5 + rank)
ref = net.store.grad_arena.clone()
if NATIVE:
    par.native_comm().allreduce_(ref)
else:
    dist.all_reduce(ref, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()

red = par.GradReducer(net.store, bucket_bytes=16 << 20, overlap=True)
assert red.overlap and len(red.buckets) >= 5 and (red.native is not None) == NATIVE
for it in range(2):                       # twice: the hook counters must re-arm
    net.store.load_state_dict(sd0)
    net.loss_and_grads(x, y, 0.75, drop_seed=5 + rank)
    red.allreduce()
    torch.cuda.synchronize()
    got = net.store.grad_arena
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 1e-6, (it, err)
    logs = [None] * world
    dist.all_gather_object(logs, list(red.launch_log))
    assert all(l == logs[0] for l in logs) and logs[0] == sorted(logs[0]) and len(logs[0]) == len(red.buckets), logs   # same collective sequence
# gradient scale: loss gradient carries 1/world (the sum over ranks is the average)
assert net.world_size == world
dist.barrier()
print("rank %d ok (buckets %d)" % (rank, len(red.buckets)))
par.shutdown()
