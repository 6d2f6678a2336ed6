"""Scalar log of a training run as JSON lines (the reference writes TensorBoard summaries through tf.summary.FileWriter,
source_segmenter.py:383-427 / adversarial.py:658-704; TensorBoard itself is out of scope, the scalars are not).

One object per line in <output_path>/metrics.jsonl: {"t": seconds since the log was opened, "kind": ..., ...scalars}.  Rank 0 only
under data parallelism; every line is flushed, so a killed run keeps what it logged.  Values must already be host numbers: the
trainers fetch a loss only where the reference's loop synchronises anyway (display steps), never once per step.
"""
import json
import os
import time


class ScalarLog(object):
    def __init__(self, output_path, rank=0, name="metrics.jsonl"):
        self.f = None
        self.t0 = time.time()
        if rank == 0 and output_path is not None:
            os.makedirs(output_path, exist_ok=True)
            self.f = open(os.path.join(output_path, name), "a", buffering=1)

    def write(self, kind, **scalars):
        if self.f is None:
            return
        rec = {"t": round(time.time() - self.t0, 4), "kind": kind}
        for k, v in scalars.items():
            rec[k] = v if isinstance(v, (int, str, bool, list, type(None))) else float(v)
        self.f.write(json.dumps(rec) + "\n")

    def close(self):
        if self.f is not None:
            self.f.close()
            self.f = None
