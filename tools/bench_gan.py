"""Throughput of the GAN steps at B=16 on one GPU (BASELINE configs 3/4): discriminator step (B MR + B CT), generator step (B CT),
joint step = 1 dis + clip + 1 gen, counted as B slices per step."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
adv = importlib.import_module("medical-cross-modality-domain-adaptation_amd.adversarial")

COST = {"regularizer": 1e-4, "gan_regularizer": 1e-4, "miu_gen": 0.002, "miu_dis": 0.002, "lambda_mask_loss": 0.3}
NETCFG = {"mr_front_trainable": False, "joint_trainable": False, "ct_front_trainable": True, "cls_trainable": True, "m_cls_trainable": True}


def main():
    B = int(os.environ.get("B", 16))
    steps = int(os.environ.get("STEPS", 5))
    dev = torch.device("cuda:0")
    net = adv.Full_DRN(channels=3, n_class=5, batch_size=B, cost_kwargs=dict(COST), network_config=dict(NETCFG), device=dev)
    rng = np.random.default_rng(0)
    sd = net.store.state_dict()
    for k, a in sd.items():
        if "Variable" in k:
            sd[k] = (rng.standard_normal(a.shape) * np.sqrt(2.0 / np.prod(a.shape[:-1]))).astype(np.float32)
    net.store.load_state_dict(sd)
    tr = adv.Trainer(net, None, None, None, None, num_cls=5, batch_size=B, opt_kwargs={"learning_rate": 3e-4},
                     train_config={"dis_sub_iter": 1, "gen_sub_iter": 1})
    tr._get_optimizer()
    mr = torch.randn((B, 256, 256, 3), device=dev)
    ct = torch.randn((B, 256, 256, 3), device=dev)
    res = {}
    for name, fn in (("dis_step", lambda s: tr.dis_step(mr, ct, 0.75, s)), ("gen_step", lambda s: tr.gen_step(ct, 0.75, s)),
                     ("joint_step", lambda s: (tr.dis_step(mr, ct, 0.75, s), tr.gen_step(ct, 0.75, s + 1000)))):
        for w in range(2):
            fn(w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            fn(10 + s)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        res[name] = {"ms_per_step": ms, "slices_per_s": B / (ms * 1e-3)}
    res["loss_finite"] = bool(np.isfinite(float(net.dis_loss)) and np.isfinite(float(net.ct_gen_loss)))
    res["B"] = B
    print(json.dumps(res))


if __name__ == "__main__":
    main()
