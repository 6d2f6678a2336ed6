"""Throughput of the train_segmenter.py entry point itself (synthetic tfrecords on disk -> SliceQueue -> DeviceFeeder -> train step,
monitoring forwards every 5th step): the number to hold against bench.py's resident-input slices/s."""
import importlib, time, sys, os
sys.path.insert(0, os.getcwd())
ts = importlib.import_module("medical-cross-modality-domain-adaptation_amd.train_segmenter")
import torch
t0 = time.time()
tr = ts.main(["--synthetic", "64", "--batch-size", "16", "--iters", "40", "--epochs", "1", "--output", "/tmp/e2e_out"])
torch.cuda.synchronize()
st = tr.step_times
import numpy as np
st = np.array(st)
print("E2E steps", len(st), "median step s", np.median(st[5:]), "mean", st[5:].mean(), "slices/s (mean, incl. monitor fwd)", 16 / st[5:].mean())
