"""is the bf16-vs-fp32 deviation of the WGAN losses at B = 2 noise or bias?  several input seeds, resident on / off"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
P = "medical-cross-modality-domain-adaptation_amd."
K, L, F, adv = (importlib.import_module(P + m) for m in ("kernels", "_lib", "functional", "adversarial"))
from test_gpu_adversarial import COST as GCOST, NETCFG, he_state
dev = torch.device("cuda:0")
B = 2
net0 = adv.Full_DRN(channels=3, n_class=5, batch_size=B, cost_kwargs=dict(GCOST), network_config=dict(NETCFG), device=dev, seed=1)
sd = he_state(net0, 7)
del net0
def run(dtype, resident, mr, ct):
    F.set_conv_dtype(dtype)
    old = K.bf16r
    if not resident:
        K.bf16r = lambda g, kind: False
    try:
        net = adv.Full_DRN(channels=3, n_class=5, batch_size=B, cost_kwargs=dict(GCOST), network_config=dict(NETCFG), device=dev, seed=1)
        net.store.load_state_dict(sd)
        dl = float(net.dis_loss_and_grads(mr, ct, 0.75, drop_seed=11))
        sc = {k: v.cpu().numpy().ravel() for k, v in net.critic_scores.items()}
        return dl, sc
    finally:
        K.bf16r = old
        F.set_conv_dtype("f32")
for seed in range(5):
    rng = np.random.default_rng(seed)
    mr = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dev)
    ct = torch.from_numpy((rng.standard_normal((B, 256, 256, 3)) * 1.2 + 0.1).astype(np.float32)).to(dev)
    d32, s32 = run("f32", False, mr, ct)
    dr, sr = run("bf16", True, mr, ct)
    ds, ss_ = run("bf16", False, mr, ct)
    scale = 0.002 * sum(np.abs(v).mean() for v in s32.values())
    print("seed %d: loss f32 %.5e resident %+.2f%% staged %+.2f%% of the terms' scale | per-score deviation / |score|: resident %s | staged %s" % (
        seed, d32, 100 * (dr - d32) / scale, 100 * (ds - d32) / scale,
        " ".join("%+.3f" % x for k in s32 for x in (sr[k] - s32[k]) / np.abs(s32[k]).mean()),
        " ".join("%+.3f" % x for k in s32 for x in (ss_[k] - s32[k]) / np.abs(s32[k]).mean())), flush=True)
