#!/usr/bin/env python
"""bench.py — BASELINE.json metric: training slices/sec (256x256x3) of the PnP-AdaNet segmenter train step on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): source segmenter fwd + bwd + Adam, B=16 slices per GPU, fp32, dropout keep 0.75,
BN in training mode — exactly `sess.run(optimizer)` of source_segmenter.py:484-489, on synthetic N(0,1) slices with blob label
maps already resident in HBM.  Weak scaling: every rank processes its own 16 slices; gradients are all-reduced over RCCL.
One JSON line is printed by rank 0, carrying `roofline` (dominant kernel: the 3x3 fp32-MFMA forward convolution, timed live
with HIP events around each of its launches inside the timed region) and `cpu_baseline` (the CPU oracle on the host cores).

`--workload gan` measures BASELINE configs[3] on the same contract instead: the joint step (1 discriminator update on B MR + B CT
slices, weight clip, 1 generator update on B CT slices; adversarial.py:839-882), B slices counted per step; no cpu_baseline there.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "medical-cross-modality-domain-adaptation_amd"

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
COST = {"cross_flag": True, "miu_cross": 1.0, "dice_flag": True, "miu_dice": 1.0, "regularizer": 1e-4}


def blob_labels(rng, B):
    yy, xx = np.mgrid[0:256, 0:256]
    lab = np.zeros((B, 256, 256), np.int64)
    for b in range(B):
        for c in range(1, 5):
            cy, cx = rng.integers(40, 216, 2)
            ry, rx = rng.integers(12, 40, 2)
            lab[b][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1] = c
    return lab


def one_hot(lab, ncls=5):
    out = np.zeros(lab.shape + (ncls,), np.float32)
    for i in range(ncls):
        out[..., i][lab == i] = 1
    return out


class ConvFwdProbe(object):
    """HIP-event timing of every pnp_conv2d_fwd launch (forward 3x3 convolutions) inside the timed region."""

    def __init__(self, K):
        self.K = K
        self.orig = K.conv2d_fwd
        self.records = []      # (flops, bytes, tile_class, ev0, ev1)
        self.enabled = False

    def install(self):
        probe = self

        def wrapped(x, w, g, keep_prob=1.0, seed=0, stream_id=0, out=None, naive=False):
            if not probe.enabled:
                return probe.orig(x, w, g, keep_prob, seed, stream_id, out, naive)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            y = probe.orig(x, w, g, keep_prob, seed, stream_id, out, naive)
            e1.record()
            flops = 2.0 * g.N * g.OH * g.OW * g.R * g.S * g.C * g.K
            nbytes = 4.0 * (g.N * g.H * g.W * g.C + g.N * g.OH * g.OW * g.K + g.R * g.S * g.C * g.K)
            # launches served by the kernel symbol conv_taps_kernel<128,128,2,2,0,3> (csrc/conv_igemm.hip::launch_fwd_tile):
            # 3x3, stride 1, zero/VALID padding, C % 32 == 0, K % 4 == 0, >= 384 tiles of 128x128
            big = (g.R == 3 and g.S == 3 and g.stride == 1 and g.pad_mode == 0 and g.K > 64 and g.K % 4 == 0 and g.C % 32 == 0
                   and (-(-g.N * g.OH * g.OW // 128) * -(-g.K // 128) >= 384))
            probe.records.append((flops, nbytes, big, e0, e1))
            return y
        self.K.conv2d_fwd = wrapped

    def summary(self):
        tot = {True: [0.0, 0.0, 0.0, 0], False: [0.0, 0.0, 0.0, 0]}
        for flops, nbytes, big, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            t = tot[big]
            t[0] += flops
            t[1] += nbytes
            t[2] += ms
            t[3] += 1
        return tot


def pmc_traffic_bytes(kernel):
    """HBM-side bytes per launch of `kernel` from the committed PMC summary (collected offline with rocprofv3 --pmc in separate
    passes, corrected as MI355X_MICROARCH.md prescribes); None when the summary is not there."""
    p = os.path.join(ROOT, "profiles", "r01_pmc_counters.json")
    try:
        k = json.load(open(p))["kernels"][kernel]
        return (k["hbm_read_MB_per_launch_corrected_x2"] + k["hbm_write_MB_per_launch"]) * 1e6
    except Exception:
        return None


def cpu_baseline(seconds_budget=30.0):
    """The CPU oracle (a port: TF-1.4 cannot run here) timed on this host's cores on a bounded sample of the same workload:
    ONE segmenter train step (fwd+bwd+Adam) at B=2 slices, all cores."""
    from oracle import nets
    # torch's default intra-op thread count honours the cgroup / affinity mask of the box (os.cpu_count() does not:
    # forcing 256 threads onto a restricted mask made this sample 50x slower)
    ncores = torch.get_num_threads()
    rng = np.random.default_rng(0)
    Bc = 2
    x = torch.from_numpy(rng.standard_normal((Bc, 256, 256, 3)).astype(np.float32))
    y = torch.from_numpy(one_hot(blob_labels(rng, Bc)))
    shapes = nets.segmenter_variable_shapes()
    state = {}
    for k, s in shapes.items():
        if "Variable" in k:
            state[k] = (rng.standard_normal(s) * np.sqrt(2.0 / (s[0] * s[1] * s[2]))).astype(np.float32)
        elif k.endswith("gamma") or k.endswith("moving_variance"):
            state[k] = np.ones(s, np.float32)
        else:
            state[k] = np.zeros(s, np.float32)
    V = nets.make_variables(state)
    opt = {}
    t0 = time.time()
    nets.segmenter_train_step(V, opt, x, y, 0.75, seed=1, lr=1e-3, t=1)
    t1 = time.time()
    steps = 1
    el = t1 - t0
    if el < seconds_budget / 3:       # fast host: take a second, warm sample
        t0 = time.time()
        nets.segmenter_train_step(V, opt, x, y, 0.75, seed=2, lr=1e-3, t=2)
        el = time.time() - t0
        steps = 2
    return {"value": Bc / el, "unit": "slices/s", "cores": ncores, "kind": "port",
            "sample": "oracle.nets.segmenter_train_step (torch-CPU fp32 restatement of source_segmenter.py:484-489), B=%d, "
                      "%d step(s), last one timed: %.2f s" % (Bc, steps, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="slices per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--workload", choices=["segmenter", "gan"], default="segmenter",
                    help="segmenter: BASELINE configs[1] (the default, the headline line); gan: configs[3] joint step = 1 dis + clip + 1 gen")
    args = ap.parse_args()

    par = importlib.import_module(PKG + ".parallel")
    if os.environ.get("PNP_SAME_DEVICE"):        # test mode: every rank on GPU 0 (with PNP_DIST_BACKEND=gloo); never used for results
        os.environ["LOCAL_RANK_ORIG"] = os.environ.get("LOCAL_RANK", "0")
    rank, local, world = par.init_distributed()
    assert world == max(args.gpus, 1) or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if os.environ.get("PNP_SAME_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    ss = importlib.import_module(PKG + ".source_segmenter")
    K = importlib.import_module(PKG + ".kernels")
    B = args.batch
    rng = np.random.default_rng(100 + rank)
    x = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dev)

    def he_scale(net):
        # He-scaled weights (the reference's stddev=.01 init gives vanishing activations after 30 layers; either is "random init")
        sd = net.store.state_dict()
        wr = np.random.default_rng(7)
        for k in sd:
            if "Variable" in k:
                s = sd[k].shape
                if len(s) == 4 and "cls" not in k:      # segmenter conv filters: rescale the truncated-normal(0.01) init
                    sd[k] = (sd[k] * (np.sqrt(2.0 / (s[0] * s[1] * s[2])) / 0.01)).astype(np.float32)
                else:                                   # critic convs / FC (stddev 0.1 shared variables): fresh He-normal draw
                    sd[k] = (wr.standard_normal(s) * np.sqrt(2.0 / np.prod(s[:-1]))).astype(np.float32)
        net.store.load_state_dict(sd)

    if args.workload == "segmenter":
        net = ss.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, cost_kwargs=dict(COST), seed=0, world_size=world)
        he_scale(net)
        reducer = par.GradReducer(net.store, overlap=not args.no_overlap) if world > 1 else None
        tr = ss.Trainer(net, None, None, num_cls=5, batch_size=B, optimizer="adam", opt_kwargs={"learning_rate": 1e-3}, reducer=reducer)
        tr.opt = tr._get_optimizer(10)
        y = torch.from_numpy(one_hot(blob_labels(rng, B))).to(dev)

        def train_step(i):
            return tr.train_step(x, y, 0.75, i * world + rank)
        metric = "training slices/sec (256x256x3, B=16 per GPU) segmenter train step (fwd+bwd+Adam)"
        workload = "BASELINE configs[1]: source segmenter fwd+bwd+Adam, B=%d/GPU, 256x256x3, fp32, dropout .75, BN train" % B
    else:
        adv = importlib.import_module(PKG + ".adversarial")
        net = adv.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, seed=0, world_size=world,
                           cost_kwargs={"regularizer": 1e-4, "gan_regularizer": 1e-4, "miu_gen": 0.002, "miu_dis": 0.002, "lambda_mask_loss": 0.3},
                           network_config={"mr_front_trainable": False, "joint_trainable": False, "ct_front_trainable": True,
                                           "cls_trainable": True, "m_cls_trainable": True})
        he_scale(net)
        reducer = par.GradReducer(net.store, overlap=not args.no_overlap) if world > 1 else None
        tr = adv.Trainer(net, None, None, None, None, num_cls=5, batch_size=B, opt_kwargs={"learning_rate": 3e-4},
                         train_config={"dis_sub_iter": 1, "gen_sub_iter": 1}, reducer=reducer)
        tr._get_optimizer()
        ct = torch.from_numpy((rng.standard_normal((B, 256, 256, 3)) * 1.2 + 0.1).astype(np.float32)).to(dev)

        def train_step(i):
            tr.dis_step(x, ct, 0.75, 2 * (i * world + rank) + 1)
            return tr.gen_step(ct, 0.75, 2 * (i * world + rank) + 2)
        metric = "training slices/sec (256x256x3, B=16 per GPU) joint segmenter+GAN step (1 dis update on B MR + B CT, clip, 1 gen update on B CT)"
        workload = "BASELINE configs[3]: train_gan.py --phase train-gan joint step, B=%d/GPU of each domain, fp32, dropout .75, mask critic on" % B

    probe = ConvFwdProbe(K)
    if not args.no_probe:
        probe.install()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    step = 0
    for _ in range(args.warmup):
        train_step(step)
        step += 1
    barrier()
    probe.enabled = not args.no_probe
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(step)
        step += 1
    barrier()
    el = time.perf_counter() - t0
    probe.enabled = False
    if world > 1:
        tmax = torch.tensor([el], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        el = float(tmax.item())
    lossv = float(loss)
    assert np.isfinite(lossv), "training diverged: loss=%r" % lossv

    if rank == 0:
        res = {
            "metric": metric,
            "value": world * B * args.steps / el, "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch": B * world, "parallelism": "dp%d" % world, "final_loss": lossv},
        }
        if not args.no_probe:
            tot = probe.summary()
            fl, by, ms, n = tot[True]
            if n:
                ach = fl / (ms * 1e-3) / 1e12
                res["roofline"] = {"bound": "mfma", "kernel": "conv_taps_kernel<128,128,2,2,0,3,3> (forward 3x3 convs on the 128x128 fp32-MFMA tile: 256->512, 512->512 (+dilated), 512->2560)",
                                   "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                                   "traffic": pmc_traffic_bytes("conv_taps_kernel<128, 128, 2, 2, 0, 3, 3>"), "launches": n, "avg_launch_ms": ms / n,
                                   "algorithmic_gflop_per_launch": fl / n / 1e9, "algorithmic_mbytes_per_launch": by / n / 1e6,
                                   "traffic_note": "HBM-side bytes per launch of this kernel symbol from the committed rocprofv3 PMC passes "
                                                   "(profiles/r01_pmc_counters.json: 2*FETCH_SIZE + WRITE_SIZE, separate passes); null if absent"}
            fl2, by2, ms2, n2 = tot[False]
            if n2:
                res["roofline_small_convs"] = {"launches": n2, "avg_launch_ms": ms2 / n2, "achieved_tflops": fl2 / (ms2 * 1e-3) / 1e12,
                                               "algorithmic_GBps": by2 / (ms2 * 1e-3) / 1e9}
        if world == 1 and not args.no_cpu_baseline and args.workload == "segmenter":     # the oracle sample is the segmenter step
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
