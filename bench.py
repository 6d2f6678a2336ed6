#!/usr/bin/env python
"""bench.py — BASELINE.json metric: training slices/sec (256x256x3, B=16) of PnP-AdaNet's segmenter+GAN step on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

Headline workload (default, BASELINE.json configs[3]): the JOINT step of `train_gan.py --phase train-gan` — one discriminator
update on B MR + B CT slices (both critics, RMSProp, weight clip; adversarial.py:839-862) followed by one generator update on the
B CT slices (adversarial.py:866-882) — B=16 slices of each domain per GPU, fp32, dropout keep 0.75, synthetic N(0,1) slices already
resident in HBM; B slices are counted per step.  Weak scaling: every rank processes its own 16+16 slices, gradients are
all-reduced (RCCL).  Rank 0 prints, as the LAST line of stdout, ONE compact JSON record (< 2 KB: the driver parses the tail of
stdout; round 3's 20 KB line arrived unparsed) carrying
  * `roofline`          the kernel symbol with the largest share of the timed region's convolution time, timed live with HIP
                        events recorded by libpnp_hip.so around each of its launches (pnp_prof_*), against the fp32-MFMA peak;
  * `roofline_all_mfma_convs`  the aggregate over every MFMA convolution symbol of the step;
  * `segmenter_step`    BASELINE configs[1] (source segmenter fwd + bwd + Adam, source_segmenter.py:484-489) timed the same way;
  * `cpu_baseline`      the CPU oracle's joint step (oracle/nets_adv.py, torch-CPU fp32) on this host's cores AT THE GPU LINE'S BATCH
                        (B = 16: 1 warm-up + 2 timed steps), with the B = 2 figure as a second field.
The per-symbol table (`roofline_kernels`: forward, data gradient, filter gradient of every convolution symbol) goes to a side file,
`bench_kernels_<workload>_<dtype>.json` under gpurun_out/ (on a GPU box) or profiles/, and its path is named in the record.
`--workload segmenter` makes configs[1] the headline line instead (same contract); `--dtype bf16` runs configs[4]'s arithmetic (a
separate line, never the headline).  The joint workload starts from BN moving statistics calibrated on the synthetic batches (40
untimed training-mode forwards) — the phase itself starts from a trained baseline checkpoint; with un-calibrated statistics the
frozen-BN segmenter of the GAN steps is un-normalised and its random-filter activations overflow (see make_joint).
"""
import argparse
import glob
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
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (~6.3 achievable)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA
COST = {"cross_flag": True, "miu_cross": 1.0, "dice_flag": True, "miu_dice": 1.0, "regularizer": 1e-4}
GAN_COST = {"regularizer": 1e-4, "gan_regularizer": 1e-4, "miu_gen": 0.002, "miu_dis": 0.002, "lambda_mask_loss": 0.3}
GAN_NETCFG = {"mr_front_trainable": False, "joint_trainable": False, "ct_front_trainable": True, "cls_trainable": True,
              "m_cls_trainable": True}


# algorithmic convolution flops (2*N*OH*OW*R*S*C*K, forward + data gradient + filter gradient) per slice of one step
ALG_GFLOP_PER_SLICE = {"joint": 399.8 + 256.2, "segmenter": 248.95}


def is_x3(name):
    """the split-bf16 GEMMs of the Winograd route (csrc/conv_wino_x3.hip): bf16 matrix pipe, six plane products per fp32 multiply-add"""
    return name.startswith("wino_gemm_x3_kernel") or name.startswith("conv_x3_direct_kernel")


def wino_alg_factor(name):
    """a Winograd GEMM row counts the flops it EXECUTES; the convolution's algorithmic flops are 9 M^2 / (M + 2)^2 times the fp32
    multiply-adds of the transform domain (F(2x2, 3x3): 2.25, F(4x4, 3x3): 4; exact for full tiles).  Symbols: wino_gemm_kernel<.., 0 / 1>
    F(2x2), <.., 2 / 3> F(4x4); wino_wgrad_gemm_kernel<.., TILE>; wino_gemm_x3_kernel<.., 0 / 1 / 4> F(2x2), <.., 2 / 3 / 5> F(4x4) — the x3
    kernel EXECUTES six bf16 MFMA products per fp32 multiply-add, so its factor is a sixth of the fp32 kernels'."""
    if name.startswith("conv_x3_direct_kernel"):        # a direct convolution: every fp32 multiply-add of the layer, six bf16 products each
        return 1.0 / 6.0
    if is_x3(name):
        return (4.0 if name.rstrip(">").split(",")[-1].strip() in ("2", "3", "5") else 2.25) / 6.0
    if name.startswith("wino_gemm_kernel"):
        return 4.0 if name.rstrip(">").split(",")[-1].strip() in ("2", "3") else 2.25
    if name.startswith("wino_wgrad_gemm_kernel"):
        return 4.0 if name.rstrip(">").split(",")[-1].strip() == "4" else 2.25
    return 1.0


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


def he_state(sd, seed=7):
    """He-scaled weights (the reference's stddev=.01 / .1 inits give vanishing / exploding activations after 30 layers; either is
    "random init").  Every filter is rescaled to std sqrt(2 / fan_in) from its OWN empirical std: the reference mixes
    weight_variable(stddev=.01) and sharable_weight_variable(stddev=.1) (layers.py:47-55), and rescaling the latter as if it were the
    former put 1e35 into the logits of the joint workload."""
    wr = np.random.default_rng(seed)
    for k in sd:
        if "Variable" in k or np.ndim(sd[k]) == 4:
            s = sd[k].shape
            if len(s) == 4 and "cls" not in k:      # segmenter conv filters (truncated-normal draws of the store's own seed)
                sd[k] = (sd[k] * (np.sqrt(2.0 / (s[0] * s[1] * s[2])) / max(float(np.std(sd[k])), 1e-12))).astype(np.float32)
            else:                                   # critic convs / FC: fresh He-normal draw
                sd[k] = (wr.standard_normal(s) * np.sqrt(2.0 / np.prod(s[:-1]))).astype(np.float32)
    return sd


def pmc_traffic_bytes(kernel):
    """HBM-side bytes per launch of `kernel` from the newest committed PMC summary (collected offline with rocprofv3 --pmc in
    separate passes, corrected as MI355X_MICROARCH.md prescribes: 2 x FETCH_SIZE + WRITE_SIZE); None when there is none."""
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_counters.json")), reverse=True):
        try:
            k = json.load(open(p))["kernels"][kernel]
            return (k["hbm_read_MB_per_launch_corrected_x2"] + k["hbm_write_MB_per_launch"]) * 1e6, os.path.basename(p)
        except Exception:
            continue
    return None, None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(workload, Bc=16, timed_steps=3, warm=1):
    """The CPU oracle (a port: TF-1.4 cannot run here) on this host's cores, same step definition as the GPU line (SURVEY.md §8(d):
    "same synthetic batch, same step definition"): Bc slices per domain, `warm` warm-up + `timed_steps` timed steps, median.  The default
    is the GPU line's own batch, B = 16, 1 + 2 steps from main() (~3.5 min of host time for the joint step; a third timed step moved the
    median by < 3 %); main() adds a B = 2 sample as a second field (the smallest batch the reference's PS accepts, ops.py:7)."""
    # torch's default intra-op thread count honours the cgroup / affinity mask of the box (os.cpu_count() does not:
    # forcing 256 threads onto a restricted mask made this sample 50x slower)
    ncores = torch.get_num_threads()
    rng = np.random.default_rng(0)
    times = []
    if workload == "segmenter":
        from oracle import nets
        x = torch.from_numpy(rng.standard_normal((Bc, 256, 256, 3)).astype(np.float32))
        y = torch.from_numpy(one_hot(blob_labels(rng, Bc)))
        state = {}
        for k, s in nets.segmenter_variable_shapes().items():
            if "Variable" in k:
                state[k] = (rng.standard_normal(s) * np.sqrt(2.0 / (s[0] * s[1] * s[2]))).astype(np.float32)
            elif k.endswith("gamma") or k.endswith("moving_variance"):
                state[k] = np.ones(s, np.float32)
            else:
                state[k] = np.zeros(s, np.float32)
        V = nets.make_variables(state)
        opt = {}
        for i in range(warm + timed_steps):
            t0 = time.time()
            nets.segmenter_train_step(V, opt, x, y, 0.75, seed=1 + i, lr=1e-3, t=1 + i)
            times.append(time.time() - t0)
        what = "oracle.nets.segmenter_train_step (torch-CPU fp32 port of source_segmenter.py:484-489)"
    else:
        from oracle import nets_adv
        adv = importlib.import_module(PKG + ".adversarial")
        # variable names / shapes from the product's symbolic build pass (meta tensors: no kernel runs, no GPU touched)
        net = adv.Full_DRN(channels=3, n_class=5, batch_size=Bc, device="cpu", seed=0, cost_kwargs=dict(GAN_COST),
                           network_config=dict(GAN_NETCFG))
        sd = he_state(net.store.state_dict())
        V = {k: torch.from_numpy(np.array(a)) for k, a in sd.items()}
        mr = torch.from_numpy(rng.standard_normal((Bc, 256, 256, 3)).astype(np.float32))
        ct = torch.from_numpy((rng.standard_normal((Bc, 256, 256, 3)) * 1.2 + 0.1).astype(np.float32))
        ms_d, ms_g = {}, {}
        for i in range(warm + timed_steps):
            t0 = time.time()
            nets_adv.joint_train_step(V, ms_d, ms_g, mr, ct, 0.75, seed=1 + 2 * i)
            times.append(time.time() - t0)
        what = "oracle.nets_adv.joint_train_step (torch-CPU fp32 port of adversarial.py:839-882)"
    med = float(np.median(times[warm:]))
    return {"value": Bc / med, "unit": "slices/s", "cores": ncores, "cpu_model": cpu_model(), "kind": "port", "batch": Bc,
            "s_per_step": med,
            # `sample`: what was timed, short enough for the driver's line; `sample_detail` (side file only): the port and every step time
            "sample": "CPU oracle %s step, B=%d, %d warm-up + %d timed steps, median %.1f s/step" % (workload, Bc, warm, timed_steps, med),
            "sample_detail": "%s, B=%d per domain, %d warm-up + %d timed steps, median %.2f s/step (all: %s)"
                             % (what, Bc, warm, timed_steps, med, " ".join("%.1f" % t for t in times))}


def _r(v, nd=4):
    """numbers of the printed record at `nd` significant digits (the full-precision values are in the side file)"""
    if isinstance(v, float):
        return float("%.*g" % (nd, v)) if np.isfinite(v) else None
    if isinstance(v, dict):
        return {k: _r(x, nd) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, nd) for x in v]
    return v


MAX_LINE = 1700      # the driver keeps a ~2000-character tail of stdout and parses the last line out of it


def compact_record(res):
    """The ONE line the driver parses: every contract field + `roofline` (dominant kernel) + `roofline_all_mfma_convs` + the secondary
    workload + `cpu_baseline`, in < MAX_LINE characters whatever the number of kernel symbols (the per-symbol table lives in the side
    file named by `kernels_file`).  Fields are dropped from the least important end if a record would still be too long."""
    out = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                               "vs_baseline", "dtype", "data") if k in res}
    cfg = dict(res.get("config", {}))
    comm = cfg.get("comm")
    if comm:       # N > 1: what was all-reduced and how much of it the step waited for (launch orders / bucket sizes: side file)
        c = {k: comm.get(k) for k in ("transport", "rccl_version", "overlap", "buckets", "allreduce_MB_per_step") if k in comm}
        for ph in ("dis_step", "gen_step"):
            if ph in comm:
                c[ph + "_exposed_ms"] = comm[ph].get("exposed_ms")
        if "exposed_ms" in comm:
            c["exposed_ms"] = comm["exposed_ms"]
        cfg["comm"] = c
    out["config"] = cfg
    if "roofline" in res:
        r = res["roofline"]
        out["roofline"] = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                 "launches", "avg_launch_ms", "share_of_conv_time", "flops_counted", "gflop_per_launch",
                                                 "achieved_algorithmic", "achieved_fp32_equivalent", "algorithmic_mbytes_per_launch") if k in r}
    for k in ("roofline_all_mfma_convs", "step_algorithmic", "segmenter_step", "joint_step", "fp32_mfma_step", "bf16_step"):
        if k in res:
            out[k] = {a: b for a, b in res[k].items() if a not in ("workload", "unit", "steps", "warmup", "peak", "probed_steps")}
    if "cpu_baseline" in res:
        out["cpu_baseline"] = {k: v for k, v in res["cpu_baseline"].items() if not k.startswith("sample_")}
    if "kernels_file" in res:
        out["kernels_file"] = res["kernels_file"]
    out = _r(out)
    for drop in (("cpu_baseline", "cpu_model"), ("roofline", "algorithmic_mbytes_per_launch"), ("roofline", "traffic_source"), ("kernels_file",),
                 ("roofline_all_mfma_convs", "launches_per_step"), ("bf16_step", "final_loss"), ("segmenter_step", "final_loss"), ("config", "comm"),
                 ("cpu_baseline", "sample")):
        if len(json.dumps(out)) < MAX_LINE:
            break
        d = out
        for k in drop[:-1]:
            d = d.get(k, {})
        d.pop(drop[-1], None)
    return out


def side_file_path(workload, dtype, world):
    """where the per-symbol table goes: gpurun_out/ when it exists (merged back from a GPU box), else profiles/"""
    d = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else os.path.join(ROOT, "profiles")
    return os.path.join(d, "bench_kernels_%s_%s%s.json" % (workload, dtype, "" if world == 1 else "_n%d" % world))


PROBE_STEPS = 2      # timed steps whose convolution launches carry HIP events (see timed_loop)


def timed_loop(step_fn, warmup, steps, world, dev, prof=None):
    """W untimed + K timed steps bracketed by barrier + synchronize; returns (seconds = max over ranks, last loss).
    prof: the per-kernel HIP events are recorded during the FIRST PROBE_STEPS steps of the timed region only — an event pair around
    each of the ~560 convolution launches of a joint step costs ~9 ms per step (measured: 117.3 vs 108.4 ms), which would otherwise
    be charged to `value`; with 2 probed steps out of K the perturbation of the reported time is below 1 % at the driver's K = 20."""
    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    k = 0
    for _ in range(warmup):
        step_fn(k)
        k += 1
    barrier()
    if prof is not None:
        prof(True)
    t0 = time.perf_counter()
    for i in range(steps):
        if prof is not None and i == PROBE_STEPS:
            prof(False)
        loss = step_fn(k)
        k += 1
    barrier()
    el = time.perf_counter() - t0
    if prof is not None:
        prof(False)
    if world > 1:
        el = importlib.import_module(PKG + ".parallel").all_max_scalar(el, dev)
    lossv = float(loss)
    assert np.isfinite(lossv), "training diverged: loss=%r" % lossv
    return el, lossv


def roofline_records(rows, peak):
    """per kernel symbol: achieved TFLOP/s of ALGORITHMIC flops / summed launch duration (live HIP events), fraction of peak"""
    out = []
    tot_ms = sum(r["ms"] for r in rows) or 1.0
    for r in sorted(rows, key=lambda r: -r["ms"]):
        if r["ms"] <= 0 or not r["launches"]:
            continue
        traffic, src = pmc_traffic_bytes(r["name"])
        if r["flops"] <= 0:          # the transform kernels of the Winograd route (csrc/conv_wino.hip): no contraction, HBM-bound
            gbs = r["bytes"] / (r["ms"] * 1e-3) / 1e9
            out.append({"kernel": r["name"], "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                        "traffic": traffic, "traffic_source": src, "launches": r["launches"], "avg_launch_ms": r["ms"] / r["launches"],
                        "share_of_conv_time": r["ms"] / tot_ms, "flops_counted": "none", "gflop_per_launch": 0.0,
                        "algorithmic_mbytes_per_launch": r["bytes"] / r["launches"] / 1e6})
            continue
        ach = r["flops"] / (r["ms"] * 1e-3) / 1e12
        wf = wino_alg_factor(r["name"])
        x3 = is_x3(r["name"])
        kpeak = PEAK_BF16_MFMA_TFLOPS if x3 else peak            # the x3 GEMMs run on the bf16 matrix pipe: THEIR roof is the dense bf16 peak
        out.append({"kernel": r["name"], "bound": "mfma", "achieved": ach, "peak": kpeak, "unit": "TFLOP/s", "frac": ach / kpeak,
                    "traffic": traffic, "traffic_source": src, "launches": r["launches"], "avg_launch_ms": r["ms"] / r["launches"],
                    "share_of_conv_time": r["ms"] / tot_ms,
                    # direct kernels: executed = algorithmic (2*N*OH*OW*R*S*C*K); Winograd GEMMs execute 2.25x / 4x fewer fp32
                    # multiply-adds; the split-bf16 GEMMs execute six bf16 MFMA products for each of those
                    "flops_counted": ("executed bf16 MFMA flops (6 plane products per fp32 multiply-add)" if x3 else "executed") if wf != 1.0 else "algorithmic",
                    **({"achieved_fp32_equivalent": ach / 6.0} if x3 else {}),
                    "gflop_per_launch": r["flops"] / r["launches"] / 1e9,
                    "achieved_algorithmic": ach * wf,
                    "algorithmic_mbytes_per_launch": r["bytes"] / r["launches"] / 1e6})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="slices per GPU (of each domain for the joint step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=16, help="slices per domain of the cpu_baseline sample (default: the GPU line's B = 16)")
    ap.add_argument("--cpu-steps", type=int, default=2, help="timed steps of the cpu_baseline sample (after --cpu-warmup)")
    ap.add_argument("--cpu-warmup", type=int, default=1)
    ap.add_argument("--cpu-small-batch", type=int, default=2, help="second, small cpu_baseline sample (0: none)")
    ap.add_argument("--no-probe", action="store_true", help="no per-kernel HIP events (no roofline objects)")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-sub", action="store_true", help="skip the secondary workload's sub-record")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16: BASELINE configs[4] arithmetic (bf16 MFMA conv operands, fp32 accumulation / master weights / BN); a separate "
                         "line, never the headline")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="step capture (step_capture.py): every un-probed step of the timed region is ONE hipGraph launch, like the "
                         "reference's one sess.run per step; auto = on for N = 1 with a per-GPU batch <= 4 (where the eager host side is the bound), "
                         "off otherwise")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): --batch slices PER GPU, the job grows with N; strong: --batch is the GLOBAL batch, every rank "
                         "takes batch / N slices (the reference's B = 16 spread over the node)")
    ap.add_argument("--workload", choices=["joint", "gan", "segmenter"], default="joint",
                    help="joint (= gan): BASELINE configs[3], the headline; segmenter: configs[1]")
    args = ap.parse_args()
    if args.workload == "gan":
        args.workload = "joint"

    par = importlib.import_module(PKG + ".parallel")
    if os.environ.get("PNP_SAME_DEVICE"):        # test mode: every rank on GPU 0 (with PNP_DIST_BACKEND=gloo); never used for results
        os.environ["LOCAL_RANK_ORIG"] = os.environ.get("LOCAL_RANK", "0")
    rank, local, world = par.init_distributed()
    assert world == max(args.gpus, 1) or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if os.environ.get("PNP_SAME_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    L = importlib.import_module(PKG + "._lib")
    K = importlib.import_module(PKG + ".kernels")
    if args.dtype == "bf16":
        importlib.import_module(PKG + ".functional").set_conv_dtype("bf16")
    B = args.batch
    if args.scaling == "strong":
        assert args.batch % world == 0 and args.batch // world >= 2, "--scaling strong: --batch must split into >= 2 slices per rank (PS, ops.py:7)"
        B = args.batch // world
    rng = np.random.default_rng(100 + rank)
    x = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dev)
    peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_FP32_MFMA_TFLOPS

    reducers = {}
    # auto: capture only where the eager host side is the bound (small per-GPU batches; profiles/r04_batch_sweep.json) — at B = 16 a
    # captured step is no faster and loses the side-stream overlap of the filter gradients (hipGraph replay serialised the branch)
    use_graph = args.graph == "on" or (args.graph == "auto" and world == 1 and args.batch <= 4)
    assert not (use_graph and world > 1), "--graph on: step capture is single-GPU (the bucketed all-reduce runs on a side stream)"
    probing = {"on": False}

    captured = {"ok": None}

    def try_capture(fn):
        """--graph auto never costs the run: a recording that fails leaves the eager step in place (and says so in the record)"""
        try:
            fn()
            captured["ok"] = captured["ok"] is not False
        except Exception as e:
            if args.graph == "on":
                raise
            captured["ok"] = False
            sys.stderr.write("step capture failed, running eagerly: %r\n" % (e,))

    def eager_while_probing(tr, fn):
        """the steps whose convolution launches carry HIP events run eagerly (events are recorded by the host around each launch)"""
        def step(i):
            if probing["on"] and getattr(tr, "_cap", None) is not None:
                cap, tr._cap = tr._cap, None
                try:
                    return fn(i)
                finally:
                    tr._cap = cap
            return fn(i)
        return step

    def make_segmenter():
        ss = importlib.import_module(PKG + ".source_segmenter")
        net = ss.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, cost_kwargs=dict(COST), seed=0, world_size=world)
        net.store.load_state_dict(he_state(net.store.state_dict()))
        reducer = par.GradReducer(net.store, overlap=not args.no_overlap) if world > 1 else None
        reducers["segmenter"] = reducer
        tr = ss.Trainer(net, None, None, num_cls=5, batch_size=B, optimizer="adam", opt_kwargs={"learning_rate": 1e-3}, reducer=reducer)
        tr.opt = tr._get_optimizer(10)
        y = torch.from_numpy(one_hot(blob_labels(rng, B))).to(dev)
        if use_graph:
            try_capture(lambda: tr.capture_step(x, y, 0.75))
        return eager_while_probing(tr, lambda i: tr.train_step(x, y, 0.75, i * world + rank))

    def make_joint():
        adv = importlib.import_module(PKG + ".adversarial")
        net = adv.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, seed=0, world_size=world, cost_kwargs=dict(GAN_COST),
                           network_config=dict(GAN_NETCFG))
        net.store.load_state_dict(he_state(net.store.state_dict()))
        reducer = par.GradReducer(net.store, overlap=not args.no_overlap) if world > 1 else None
        reducers["joint"] = reducer
        tr = adv.Trainer(net, None, None, None, None, num_cls=5, batch_size=B, opt_kwargs={"learning_rate": 3e-4},
                         train_config={"dis_sub_iter": 1, "gen_sub_iter": 1}, reducer=reducer)
        tr._get_optimizer()
        ct = torch.from_numpy((rng.standard_normal((B, 256, 256, 3)) * 1.2 + 0.1).astype(np.float32)).to(dev)
        # The phase starts from a trained segmenter whose BN moving statistics match its activations (train_gan.py restores the baseline
        # checkpoint); the GAN steps run the segmenter's BN in inference mode.  Random filters with the initial (0, 1) statistics would
        # leave those layers un-normalised (activations grow by sqrt(2) per residual block), so the statistics are calibrated first:
        # training-mode forwards of both domains, untimed, until the moving averages (decay .9) have converged to the batch statistics.
        with torch.no_grad():
            for i in range(40):
                net._graph(x, ct, 1.0, mr_front_bn=True, joint_bn=True, ct_front_bn=True, critics=False, drop_seed=1000 + i)

        if use_graph:
            try_capture(lambda: tr.capture_steps(x, ct, 0.75))

        def step(i):
            tr.dis_step(x, ct, 0.75, 2 * (i * world + rank) + 1)
            if reducer is not None:
                step.comm_dis = (reducer.bytes_step, list(reducer.launch_log), reducer.exposed_time_ms() if reducer.measure_exposed else None)
            return tr.gen_step(ct, 0.75, 2 * (i * world + rank) + 2)
        inner = eager_while_probing(tr, step)

        def outer(i):
            r = inner(i)
            if hasattr(step, "comm_dis"):
                outer.comm_dis = step.comm_dis
            return r
        return outer

    names = {
        "joint": ("training slices/sec (256x256x3, B=%d/GPU) joint segmenter+GAN step (1 dis update on B MR + B CT, clip, 1 gen update on B CT)" % B,
                  "BASELINE configs[3]: train_gan.py --phase train-gan joint step, B=%d/GPU per domain, %s, dropout .75, mask critic on" % (B, args.dtype)),
        "segmenter": ("training slices/sec (256x256x3, B=%d/GPU) segmenter train step (fwd+bwd+Adam)" % B,
                      "BASELINE configs[1]: source segmenter fwd+bwd+Adam, B=%d/GPU, 256x256x3, %s, dropout .75, BN train" % (B, args.dtype)),
    }
    makers = {"joint": make_joint, "segmenter": make_segmenter}
    other = "segmenter" if args.workload == "joint" else "joint"

    Fn = importlib.import_module(PKG + ".functional")
    overlap_default = Fn.WGRAD_STREAM

    def prof(on):
        if not args.no_probe:
            probing["on"] = bool(on)
            # a probed step times each convolution launch on its own: no filter gradient next to a data gradient on another stream
            # (two kernels sharing the chip would each be charged the other's time)
            Fn.WGRAD_STREAM = overlap_default and not on
            L.prof_enable((L.PROF_CONV_FWD | L.PROF_CONV_DGRAD | L.PROF_CONV_WGRAD) if on else 0)

    step_fn = makers[args.workload]()
    el, lossv = timed_loop(step_fn, args.warmup, args.steps, world, dev, prof)
    rows = [] if args.no_probe else L.prof_summary()
    comm = None
    if world > 1:
        # what the all-reduce moved and how much of it the step had to wait for: ONE extra, untimed step with an event pair around the
        # compute stream's wait on the side stream (GradReducer.measure_exposed)
        red = reducers[args.workload]
        red.measure_exposed = True
        step_fn(args.warmup + args.steps)
        torch.cuda.synchronize()
        nc = par.native_comm()
        comm = {"transport": par.transport(), "rccl_version": nc.version if nc is not None else None, "overlap": bool(red.overlap),
                "buckets": len(red.buckets), "bucket_MB": [round((e - s_) * 4 / 1e6, 1) for s_, e in red.buckets],
                "distinct_bucket_sets": len(red.sets_seen)}
        if args.workload == "joint":
            db, dlog, dexp = getattr(step_fn, "comm_dis", (None, None, None))
            comm.update({"dis_step": {"allreduce_MB": db / 1e6 if db is not None else None, "launch_order": dlog, "exposed_ms": dexp},
                         "gen_step": {"allreduce_MB": red.bytes_step / 1e6, "launch_order": list(red.launch_log),
                                      "exposed_ms": red.exposed_time_ms()}})
            comm["allreduce_MB_per_step"] = ((db or 0) + red.bytes_step) / 1e6
        else:
            comm.update({"allreduce_MB_per_step": red.bytes_step / 1e6, "launch_order": list(red.launch_log),
                         "exposed_ms": red.exposed_time_ms()})
        red.measure_exposed = False
    del step_fn
    sub = None
    if not args.no_sub:
        sub_steps, sub_warm = max(5, min(args.steps, 10)), 2
        el2, loss2 = timed_loop(makers[other](), sub_warm, sub_steps, world, dev, None)
        sub = {"workload": names[other][1], "value": world * B * sub_steps / el2, "unit": "slices/s", "ms_per_step": 1e3 * el2 / sub_steps,
               "steps": sub_steps, "warmup": sub_warm, "final_loss": loss2}

    # The headline's Winograd GEMMs run on split-bf16 operands (csrc/conv_wino_x3.hip: every fp32 value as three bf16 planes, six products,
    # fp32 accumulation in 64-channel chunks — fp32 results, measured CLOSER to float64 than the fp32 matrix pipe's: DESIGN.md §2).  The
    # same step with every GEMM on the native fp32 MFMA pipe (PNP_WINOGRAD_X3=0 PNP_X3_DIRECT=0, round 5's arithmetic) stays selectable and is reported
    # beside the headline: 2 warm-up + 8 timed steps
    sub_fp32 = None
    if not args.no_sub and args.dtype == "f32" and world == 1 and K.wino_x3(-1) != 0:
        prev_x3 = K.wino_x3(0)
        prev_x3d = K.x3_direct(0)        # (and the narrow layers back on the fp32-pipe routes: nothing of this sub-run touches the bf16 pipe)
        try:
            K.weights_changed()
            nf = 8
            el4, loss4 = timed_loop(makers[args.workload](), 2, nf, world, dev, None)
            sub_fp32 = {"workload": "the headline step with every convolution on the fp32 matrix pipe (PNP_WINOGRAD_X3=0 PNP_X3_DIRECT=0)", "value": world * B * nf / el4,
                        "unit": "slices/s", "ms_per_step": 1e3 * el4 / nf, "steps": nf, "warmup": 2, "final_loss": loss4}
        finally:
            K.wino_x3(prev_x3)
            K.x3_direct(prev_x3d)
            K.weights_changed()

    # BASELINE configs[4] arithmetic (bf16 MFMA operands, fp32 accumulation / master weights / BN) on the headline workload, as a
    # sub-record of the driver's own line: same step definition, 2 warm-up + 10 timed steps
    sub_bf16 = None
    if not args.no_sub and args.dtype == "f32" and args.workload == "joint" and world == 1:      # (N = 1 only: the scaling runs stay what they were)
        Fn.set_conv_dtype("bf16")
        try:
            nb = 10
            el3, loss3 = timed_loop(make_joint(), 2, nb, world, dev, None)
            sub_bf16 = {"workload": "BASELINE configs[4] arithmetic on the joint step: bf16 MFMA conv operands (resident bf16 activations / filter shadows), "
                                    "fp32 accumulation, master weights and BN; B=%d/GPU" % B,
                        "value": world * B * nb / el3, "unit": "slices/s", "ms_per_step": 1e3 * el3 / nb, "steps": nb, "warmup": 2, "final_loss": loss3}
        finally:
            Fn.set_conv_dtype("f32")

    if rank == 0:
        res = {
            "metric": names[args.workload][0],
            "value": world * B * args.steps / el, "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            # (N = 1: no GradReducer is built — the driver's SCALE N=1 run and its BENCH run execute the same code path)
            "config": {"workload": names[args.workload][1], "global_batch": B * world, "per_gpu_batch": B, "parallelism": "dp%d" % world, "step_capture": bool(use_graph and captured["ok"]),
                       "final_loss": lossv, "comm": comm},
        }
        if rows:
            recs = roofline_records(rows, peak)
            if recs:
                top = dict(recs[0])
                res["roofline"] = top
                res["roofline_note"] = ("roofline = the kernel symbol with the largest share of the timed region's convolution time; `achieved` = sum of "
                                        "algorithmic FLOP (2*N*OH*OW*R*S*C*K) / sum of launch durations, HIP events recorded by libpnp_hip.so on the "
                                        "launch stream around every launch of the symbol in the first %d steps of the timed region; `traffic` = HBM-side "
                                        "bytes per launch from the committed rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, separate passes; "
                                        "`traffic_source` names the file), null if absent.  Symbols of the Winograd route (csrc/conv_wino.hip): `wino_gemm_kernel` / "
                                        "`wino_wgrad_gemm_kernel` are priced at the flops they EXECUTE (`flops_counted`: 2*P^2*T*C*K, T = tiles of MxM outputs, "
                                        "P = M + 2: 2.25x (F(2x2)) / 4x (F(4x4)) fewer than the convolution's 2*N*OH*OW*9*C*K — `achieved_algorithmic` is the "
                                        "same launch priced at SURVEY 8(d)'s convolution flops); `wino_gemm_x3_kernel` (csrc/conv_wino_x3.hip: the same GEMMs on split-bf16 operands) is priced "
                                        "at the bf16 MFMA flops it executes — six plane products per fp32 multiply-add — against the dense bf16 peak, `achieved_fp32_equivalent` = a sixth "
                                        "of that; the transform kernels appear as HBM-bound rows (bytes / "
                                        "duration against 8 TB/s).  `step_algorithmic`: SURVEY 8(d) flops of the whole step / the step's wall time" % PROBE_STEPS)
                res["roofline_kernels"] = recs
                fl, ms = sum(r["flops"] / (6.0 if is_x3(r["name"]) else 1.0) for r in rows), sum(r["ms"] for r in rows)      # (fp32-equivalent executed flops)
                # `achieved`: flops the kernels EXECUTE / all convolution kernel time (Winograd transforms included: time, no flops);
                # `achieved_algorithmic`: SURVEY.md §8(d) convolution flops of the step / the same time — comparable across rounds whatever
                # the route (it exceeds the fp32 MFMA peak when the Winograd routes skip enough multiplications)
                alg = ALG_GFLOP_PER_SLICE[args.workload] * B * min(PROBE_STEPS, args.steps) * 1e9
                res["roofline_all_mfma_convs"] = {"achieved": fl / (ms * 1e-3) / 1e12, "peak": peak, "frac": fl / (ms * 1e-3) / 1e12 / peak,
                                                  "achieved_algorithmic": alg / (ms * 1e-3) / 1e12,
                                                  "unit": "TFLOP/s", "ms_per_step": ms / min(PROBE_STEPS, args.steps),
                                                  "launches_per_step": sum(r["launches"] for r in rows) / min(PROBE_STEPS, args.steps),
                                                  "probed_steps": min(PROBE_STEPS, args.steps)}
        tfl = ALG_GFLOP_PER_SLICE[args.workload] * B * 1e-3                      # TFLOP per step per GPU
        res["step_algorithmic"] = {"tflop_per_step": tfl, "achieved": tfl / (el / args.steps), "frac_of_mfma_peak": tfl / (el / args.steps) / peak}
        if sub is not None:
            res["segmenter_step" if other == "segmenter" else "joint_step"] = sub
        if sub_fp32 is not None:
            res["fp32_mfma_step"] = sub_fp32
        if sub_bf16 is not None:
            res["bf16_step"] = sub_bf16
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.workload, args.cpu_batch, args.cpu_steps, args.cpu_warmup)
            if args.cpu_small_batch and args.cpu_small_batch != args.cpu_batch:
                small = cpu_baseline(args.workload, args.cpu_small_batch, 2, 1)
                cb["value_B%d" % args.cpu_small_batch] = small["value"]
                cb["sample_B%d" % args.cpu_small_batch] = small["sample_detail"]
            res["cpu_baseline"] = cb
        # what the host side of the library holds on to between steps (workspaces per slot and stream, transformed filters, bf16 shadows) and
        # what torch's allocator has reserved at the end of the run: side file only
        res["device_memory"] = dict(K.memory_report(), torch_reserved=int(torch.cuda.memory_reserved(dev)), torch_peak_allocated=int(torch.cuda.max_memory_allocated(dev)))
        # the per-symbol table and the full-precision record: side file (and stderr); the LAST stdout line is the compact record
        try:
            path = side_file_path(args.workload, args.dtype, world)
            with open(path, "w") as f:
                json.dump(res, f, indent=1)
            res["kernels_file"] = os.path.relpath(path, ROOT)
        except OSError:
            pass
        sys.stderr.write(json.dumps(res) + "\n")
        sys.stderr.flush()
        line = json.dumps(compact_record(res))
        assert len(line) < MAX_LINE and "\n" not in line, len(line)
        sys.stdout.flush()
        print(line, flush=True)
    if world > 1:
        par.shutdown()


if __name__ == "__main__":
    main()
