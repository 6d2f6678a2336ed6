"""Throughput of the train_gan.py --phase train-gan ENTRY POINT itself (adversarial.py:831-946) — the headline phase end to end:
synthetic tfrecords on disk -> four SliceQueues / DeviceFeeders (CT / MR, train / val) -> the reference's schedule of 20 discriminator
updates + 1 generator update per outer iteration (dis_sub_iter = 20, gen_sub_iter = 1, train_gan.py:57-60) -> monitoring forwards on a
training and a validation batch every 5th iteration -> metrics.jsonl.  The phase starts the way the reference's does: from a source
segmenter (train_segmenter.py, a few iterations: calibrated BN moving statistics) handed over through --phase pre-train.

Prints one line to hold against bench.py's resident-input numbers:  20 x dis_step + 1 x gen_step of tools/bench_gan.py vs the measured
outer iteration."""
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
PKG = "medical-cross-modality-domain-adaptation_amd"
B = int(os.environ.get("B", 16))
OUT = os.environ.get("OUT", "/tmp/e2e_gan")
ITERS = int(os.environ.get("ITERS", 7))          # outer iterations of train-gan (iteration 0 only monitors)

ts = importlib.import_module(PKG + ".train_segmenter")
tg = importlib.import_module(PKG + ".train_gan")
t0 = time.time()
ts.main(["--synthetic", "32", "--batch-size", str(B), "--iters", "12", "--epochs", "1", "--output", OUT + "/seg"])
tg.main("pre-train", ["--phase", "pre-train", "--synthetic", "32", "--batch-size", str(B), "--iters", "6", "--epochs", "1", "--output", OUT + "/gan",
                      "--baseline", OUT + "/seg/checkpoint.npz"])
t_setup = time.time() - t0
torch.cuda.synchronize()
if os.path.exists(OUT + "/gan/metrics.jsonl"):
    os.remove(OUT + "/gan/metrics.jsonl")       # the pre-train phase logged into the same folder
t1 = time.time()
tr = tg.main("train-gan", ["--phase", "train-gan", "--synthetic", "64", "--batch-size", str(B), "--iters", str(ITERS), "--epochs", "1",
                            "--output", OUT + "/gan"])
torch.cuda.synchronize()
wall = time.time() - t1
rows = [json.loads(l) for l in open(OUT + "/gan/metrics.jsonl")]
full = [r for r in rows if r["kind"] == "gan_step" and r.get("dis_updates", 0) == 20]
# The host only QUEUES a step; the device drains at the monitoring forwards (every 5th iteration) and at the end.  Steady state = wall
# clock between the log lines of the first and the last full iteration (they are written after the iteration's monitoring, if any).
span = full[-1]["t"] - full[0]["t"] if len(full) > 1 else float("nan")
per_iter = span / (len(full) - 1) if len(full) > 1 else float("nan")
evals = [r for r in rows if r["kind"] in ("train_eval", "val_eval")]
print("E2E train-gan B=%d: %d outer iterations (20 dis + 1 gen each) in %.2f s wall incl. start-up; steady state %.3f s per outer iteration "
      "= %.1f slices/s counted per generator update, %.1f slices/s counted per update of either kind (21 updates x %d slices); "
      "monitoring forwards logged: %d; setup (segmenter + pre-train phases) %.1f s" % (
          B, len(full), wall, per_iter, B / per_iter, 21 * B / per_iter, B, len(evals), t_setup))
print("E2E last eval:", evals[-1] if evals else None)
