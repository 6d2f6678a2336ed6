"""shared by tests/dp_sync_worker.py and tests/test_gpu_dp.py: the same weights and the same 4-slice batch on both sides"""
import numpy as np
import torch

COST = {"cross_flag": True, "miu_cross": 1.0, "dice_flag": True, "miu_dice": 1.0, "regularizer": 1e-4}


def scaled_state(net):
    sd = net.store.state_dict()
    for k in sd:
        if "/Variable" in k:
            s = sd[k].shape
            sd[k] = (sd[k] * (np.sqrt(2.0 / (s[0] * s[1] * s[2])) / 0.01)).astype(np.float32)
    sd["output/Variable"] = (sd["output/Variable"] * 0.05).astype(np.float32)    # smooth loss (far from the 0.005 clip)
    return sd


def make_batch(B):
    rng = np.random.default_rng(42)
    x = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32))
    lab = np.zeros((B, 256, 256), np.int64)
    for b in range(B):                      # different class proportions per slice: per-replica normalisers would differ
        for c in range(1, 5):
            cy, cx = rng.integers(40, 216, 2)
            r = 8 + 10 * ((b + c) % 4)
            lab[b, cy - r:cy + r, cx - r:cx + r] = c
    y = torch.from_numpy(np.eye(5, dtype=np.float32)[lab])
    return x, y
