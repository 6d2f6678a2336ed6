"""round 6: where does the HOST time of a joint step go (cProfile over 6 eager bf16 steps at B = 16; the bf16 step is host-bound)"""
import cProfile
import importlib
import os
import pstats
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")
PKG = "medical-cross-modality-domain-adaptation_amd"
adv = importlib.import_module(PKG + ".adversarial")
Fn = importlib.import_module(PKG + ".functional")
dtype = os.environ.get("DTYPE", "bf16")
if dtype == "bf16":
    Fn.set_conv_dtype("bf16")
B = int(os.environ.get("B", 16))
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
net = adv.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, seed=0, cost_kwargs=dict(bench.GAN_COST), network_config=dict(bench.GAN_NETCFG))
net.store.load_state_dict(bench.he_state(net.store.state_dict()))
tr = adv.Trainer(net, None, None, None, None, num_cls=5, batch_size=B, opt_kwargs={"learning_rate": 3e-4}, train_config={"dis_sub_iter": 1, "gen_sub_iter": 1})
tr._get_optimizer()
x = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dev)
ct = torch.from_numpy((rng.standard_normal((B, 256, 256, 3)) * 1.2 + 0.1).astype(np.float32)).to(dev)


def step(i):
    tr.dis_step(x, ct, 0.75, 2 * i + 1)
    return tr.gen_step(ct, 0.75, 2 * i + 2)


for i in range(3):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(3, 9):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
