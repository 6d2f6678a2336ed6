"""Per-layer timing of the conv kernels at BASELINE sizes (B=16): fwd / dgrad / wgrad TFLOP/s vs the 157.3 TF fp32 MFMA peak."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("medical-cross-modality-domain-adaptation_amd.kernels")
L = importlib.import_module("medical-cross-modality-domain-adaptation_amd._lib")

DTYPE = os.environ.get("DTYPE", "f32")          # bf16: bf16 MFMA operands (csrc/conv_bf16.hip) where the layer is on those kernels
PEAK = 2500.0 if DTYPE == "bf16" else 157.3
B = int(os.environ.get("B", 16))
if os.environ.get("WINO") is not None:          # route of the wide stride-1 3x3 layers (csrc/conv_wino.hip): 0 direct, 1 planner, 2 wherever eligible
    K.wino_mode(int(os.environ["WINO"]))
if os.environ.get("WINO_WGRAD") is not None:    # the same for the filter gradient
    K.wino_wgrad_mode(int(os.environ["WINO_WGRAD"]))
if os.environ.get("TILE") is not None:          # largest output tile of the route: 2 = F(2x2, 3x3) only, 4 = F(4x4, 3x3) first
    K.wino_tile(int(os.environ["TILE"]))
if os.environ.get("X3") is not None:            # arithmetic of the route's GEMMs: 0 fp32 matrix pipe, 1 split-bf16 operands (csrc/conv_wino_x3.hip)
    K.wino_x3(int(os.environ["X3"]))
SKIP_WGRAD = bool(os.environ.get("SKIP_WGRAD"))
PROF = bool(os.environ.get("PROF"))             # after each layer's line: one more pass of each kind with the library's per-launch HIP events, per kernel symbol
LAYERS = [  # name, H, C, K, R, dil, padding, count in segmenter fwd
    ("g1 3->16", 256, 3, 16, 3, 1, "SAME", 1),
    ("g1 16->16", 256, 16, 16, 3, 1, "SAME", 2),
    ("g2 16->32", 128, 16, 32, 3, 1, "SAME", 1),
    ("g2 32->32", 128, 32, 32, 3, 1, "SAME", 1),
    ("g3 32->64", 64, 32, 64, 3, 1, "SAME", 1),
    ("g3 64->64", 64, 64, 64, 3, 1, "SAME", 3),
    ("g4 64->128", 32, 64, 128, 3, 1, "SAME", 1),
    ("g4 128->128", 32, 128, 128, 3, 1, "SAME", 3),
    ("g5 128->256", 32, 128, 256, 3, 1, "SAME", 1),
    ("g5/6 256->256", 32, 256, 256, 3, 1, "SAME", 7),
    ("g7 256->512", 32, 256, 512, 3, 1, "SAME", 1),
    ("g7/9 512->512", 32, 512, 512, 3, 1, "SAME", 5),
    ("g8 512->512 d2", 32, 512, 512, 3, 2, "SAME", 4),
    ("g10 512->2560", 32, 512, 2560, 3, 1, "SYMMETRIC", 1),
    ("out 40->5 k5", 256, 40, 5, 5, 1, "SYMMETRIC", 1),
    ("cls1 32->64", 256, 32, 64, 3, 1, "SAME", 0),
    ("cls1 64->64", 256, 64, 64, 3, 1, "SAME", 0),
    ("cls2 64->128", 128, 64, 128, 3, 1, "SAME", 0),
    ("cls2 128->128", 128, 128, 128, 3, 1, "SAME", 0),
    ("cls3 128->256", 64, 128, 256, 3, 1, "SAME", 0),
    ("cls3 256->256", 64, 256, 256, 3, 1, "SAME", 0),
    ("cls5 512->512@16", 16, 512, 512, 3, 1, "SAME", 0),
    # strided critic convolutions (adversarial.py:342-391); 9th field = stride
    ("cls k3s2 64@256", 256, 64, 64, 3, 1, "SAME", 0, 2),
    ("cls k5s2 128@128", 128, 128, 128, 5, 1, "SAME", 0, 2),
    ("cls k3s2 256@64", 64, 256, 256, 3, 1, "SAME", 0, 2),
    ("cls k3s2 512@32", 32, 512, 512, 3, 1, "SAME", 0, 2),
    ("cls k5s4 512@16", 16, 512, 512, 5, 1, "SAME", 0, 4),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    totflop = 0.0
    print("%-18s %9s | %8s %6s | %8s %6s | %8s %6s" % ("layer", "GFLOP", "fwd ms", "TF/s", "dgrad ms", "TF/s", "wgrad ms", "TF/s"))
    only = os.environ.get("ONLY")
    for name, H, C, Kc, R, dil, padding, cnt, *rest in LAYERS:
        stride = rest[0] if rest else 1
        if only and not any(o in name for o in only.split(",")):
            continue
        x = torch.randn((B, H, H, C), device=dev)
        w = torch.randn((R, R, C, Kc), device=dev) * 0.05
        if padding == "SYMMETRIC":      # the model path: mirror-pad once (pnp_sympad_fwd), then a VALID convolution
            x = K.sympad_fwd(x, R // 2)
            padding = "VALID"
        g = K.conv_geom(tuple(x.shape), tuple(w.shape), stride, dil, padding, dtype=L.DTYPE_BF16 if DTYPE == "bf16" else L.DTYPE_F32)
        dy = torch.randn((B, g.OH, g.OW, Kc), device=dev)
        flop = 2.0 * B * g.OH * g.OW * R * R * C * Kc
        tf = timeit(lambda: K.conv2d_fwd(x, w, g))
        td = timeit(lambda: K.conv2d_dgrad(dy, w, g))
        tw = float('nan') if SKIP_WGRAD else timeit(lambda: K.conv2d_wgrad(x, dy, g))
        print("%-18s %9.2f | %8.3f %6.1f | %8.3f %6.1f | %8.3f %6.1f%s" % (name, flop / 1e9, tf, flop / tf / 1e9, td, flop / td / 1e9, tw,
                                                                     flop / tw / 1e9, "  [winograd fwd/dgrad/wgrad: %d/%d/%d]" % (K.wino_chosen(g, 0), K.wino_chosen(g, 1), K.wino_chosen(g, 2))
                                                                     if (K.wino_chosen(g, 0) or K.wino_chosen(g, 1) or K.wino_chosen(g, 2)) else ""))
        if PROF:
            for kind, cls, fn in (("fwd", L.PROF_CONV_FWD, lambda: K.conv2d_fwd(x, w, g)), ("dgrad", L.PROF_CONV_DGRAD, lambda: K.conv2d_dgrad(dy, w, g)),
                                  ("wgrad", L.PROF_CONV_WGRAD, lambda: K.conv2d_wgrad(x, dy, g))):
                L.prof_summary()
                L.prof_enable(cls)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                L.prof_enable(0)
                for r in L.prof_summary():
                    us = 1e3 * r["ms"] / max(r["launches"], 1)
                    print("      %-5s %-46s %8.1f us  %s" % (kind, r["name"], us, ("%6.1f TF/s executed" % (r["flops"] / r["launches"] / us / 1e6)) if r["flops"] > 0
                                                              else ("%6.2f TB/s" % (r["bytes"] / r["launches"] / us / 1e6))))
        tot["fwd"] += tf * cnt
        tot["dgrad"] += td * cnt
        tot["wgrad"] += tw * cnt
        totflop += flop * cnt
    print("segmenter conv totals (ms): fwd %.2f dgrad %.2f wgrad %.2f ; fwd GFLOP %.1f -> %.1f TF/s (%.1f%% of %.1f)" % (
        tot["fwd"], tot["dgrad"], tot["wgrad"], totflop / 1e9, totflop / tot["fwd"] / 1e9, 100 * totflop / tot["fwd"] / 1e9 / PEAK, PEAK))


if __name__ == "__main__":
    main()
