"""Reads the per-stage shader-clock trace of a -DPNP_TRACE=1 build (csrc/conv_igemm.hip: PNP_TRACE_MARK) for one forward launch of a
layer and prints, over the traced workgroups: set-up, prologue (first loads -> main loop), cycles per stage (median / p10 / p90), epilogue.
  PNP_LIB=.../libpnp_hip_trace.so python tools/experiments/stage_trace.py narrow|wide|long64"""
import ctypes, importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
K = importlib.import_module("medical-cross-modality-domain-adaptation_amd.kernels")
L = importlib.import_module("medical-cross-modality-domain-adaptation_amd._lib")
dev = torch.device("cuda:0")
CASES = {"narrow": ((16, 256, 256, 64), (3, 3, 64, 64), True),       # cls_1: 8192 tiles of 128x64, 18 stages
         "long64": ((16, 32, 32, 256), (3, 3, 256, 256), True),       # 512 tiles of 128x64, 72 stages
         "wide": ((16, 32, 32, 512), (3, 3, 512, 512), False),       # 512 tiles of 128x128, 144 stages
         "wide36": ((16, 128, 128, 128), (3, 3, 128, 128), False)}   # cls_2: 4096 tiles of 128x128, 36 stages
for which in sys.argv[1:] or ["narrow", "long64", "wide", "wide36"]:
    xs, ws, t3 = CASES[which]
    x = torch.randn(xs, device=dev); w = torch.randn(ws, device=dev) * 0.03
    g = K.conv_geom(xs, ws, 1, 1, "SAME")
    for _ in range(3):
        y = K.conv2d_fwd(x, w, g)
    torch.cuda.synchronize()
    buf = np.zeros((512, 192), np.uint64)
    L.load()
    raw = ctypes.CDLL(L.LIB_PATH)                  # same handle as the loaded library: reads its g_trace
    raw.pnp_debug_trace_read.restype = ctypes.c_int
    raw.pnp_debug_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert raw.pnp_debug_trace_read(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
    S = ws[0] * ws[1] * (xs[3] // 32)
    base = 2 if t3 else 3                         # slot of "end of stage k" is base + k
    t = buf.astype(np.int64)
    ok = t[:, 0] > 0
    t = t[ok]
    setup = t[:, 1] - t[:, 0]
    prol = t[:, 2] - t[:, 1]
    ends = t[:, base + 1: base + 1 + min(S, 180)]
    starts = np.concatenate([t[:, 2:3], ends[:, :-1]], axis=1)
    per = (ends - starts)
    ep0 = t[:, base + 1 + S] if base + 2 + S < 192 else None
    line = "%-7s %d workgroups traced, %d stages: set-up %5.0f clk, prologue %6.0f, stage median %6.0f (p10 %6.0f, p90 %6.0f; first %6.0f, last %6.0f)" % (
        which, t.shape[0], S, np.median(setup), np.median(prol), np.median(per), np.percentile(per, 10), np.percentile(per, 90),
        np.median(per[:, 0]), np.median(per[:, -1]))
    if ep0 is not None:
        ep = t[:, base + 2 + S] - ep0
        tot = t[:, base + 2 + S] - t[:, 0]
        line += ", epilogue %6.0f, whole workgroup %7.0f clk (main loop %.0f %%)" % (np.median(ep), np.median(tot), 100 * np.median((ep0 - t[:, 2]) / tot))
    print(line)
    t0, t1 = t[:, 0], t[:, base + 2 + S] if base + 2 + S < 192 else t[:, 2]
    print("        first instruction of the traced workgroups spread over %7.0f clk (p50 - min %6.0f); last instruction spread %7.0f; first start -> last end %8.0f clk"
          % (t0.max() - t0.min(), np.median(t0) - t0.min(), t1.max() - t1.min(), t1.max() - t0.min()))
