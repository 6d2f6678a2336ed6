"""PS (phase shift) fwd/bwd timing at BASELINE sizes: g10 output (nc=40) and the critic feature inputs (nc=2,4,8), B=16."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("medical-cross-modality-domain-adaptation_amd.kernels")


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for nc in (40, 8, 4, 2):
    x = torch.randn((16, 32, 32, nc * 64), device="cuda")
    y = K.ps_fwd(x, 8, nc)
    tf, tb = timeit(lambda: K.ps_fwd(x, 8, nc)), timeit(lambda: K.ps_bwd(y, 8, nc))
    mb = 2 * x.numel() * 4 / 1e6
    print("PS nc=%2d: fwd %.3f ms (%.0f GB/s)  bwd %.3f ms (%.0f GB/s)" % (nc, tf, mb / tf, tb, mb / tb))
