"""bf16-resident convolutions (csrc/conv_bf16r.hip) layer by layer at B=16: correctness against the fp32-MFMA kernels on bf16-ROUNDED
operands (products of bf16 values are exact in fp32, so the two differ by fp32 summation order only) and TF/s against the 2.5 PF bf16
MFMA peak, next to the staged-rounding kernels of rounds 2-3 (csrc/conv_bf16.hip).   PNP_BF16R_TILE=0/1/2 forces a tile."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module("medical-cross-modality-domain-adaptation_amd.kernels")
L = importlib.import_module("medical-cross-modality-domain-adaptation_amd._lib")
from bench_conv import LAYERS, timeit      # noqa: E402

B = int(os.environ.get("B", 16))
PEAK = 2500.0


def main():
    dev = torch.device("cuda:0")
    only = os.environ.get("ONLY")
    check = os.environ.get("CHECK", "1") != "0"
    print("%-18s %8s | %8s %7s %8s %9s | %8s %7s %8s %9s | %8s %7s %8s %9s" % ("layer", "GFLOP", "fwd ms", "TF/s", "old ms", "max err", "dgrad ms",
                                                                             "TF/s", "old ms", "max err", "wgrad ms", "TF/s", "old ms", "max err"))
    tot = {"f": 0.0, "d": 0.0, "w": 0.0, "fo": 0.0, "do": 0.0, "wo": 0.0}
    totflop = 0.0
    for name, H, C, Kc, R, dil, padding, cnt, *rest in LAYERS:
        stride = rest[0] if rest else 1
        if only and not any(o in name for o in only.split(",")):
            continue
        x = torch.randn((B, H, H, C), device=dev)
        w = torch.randn((R, R, C, Kc), device=dev) * 0.05
        if padding == "SYMMETRIC":
            x = K.sympad_fwd(x, R // 2)
            padding = "VALID"
        g = K.conv_geom(tuple(x.shape), tuple(w.shape), stride, dil, padding, dtype=L.DTYPE_BF16)
        gf = K.conv_geom(tuple(x.shape), tuple(w.shape), stride, dil, padding, dtype=L.DTYPE_F32)
        flop = 2.0 * B * g.OH * g.OW * R * R * C * Kc
        line = "%-18s %8.2f |" % (name, flop / 1e9)
        dy = torch.randn((B, g.OH, g.OW, Kc), device=dev)
        xh, dyh = K.cast_bf16(x), K.cast_bf16(dy)
        w_io, w_oi = K.filter_bf16(w)
        xr, wr, dyr = xh.float(), w.bfloat16().float(), dyh.float()      # the rounded operands as fp32 tensors (test tool: torch casts)
        for kind in (0, 1, 2):
            if not K.bf16r_served(g, kind):
                line += " %8s %7s %8s %9s |" % ("-", "-", "-", "-")
                continue
            if kind == 0:
                fn = lambda: K.conv2d_fwd_bf16r(xh, w_oi, g, want_h=True)
                old = lambda: K.conv2d_fwd(x, w, g)
                ref = lambda: K.conv2d_fwd(xr, wr, gf)
            elif kind == 1:
                fn = lambda: K.conv2d_dgrad_bf16r(dyh, w_io, g, want_h=stride == 1)
                old = lambda: K.conv2d_dgrad(dy, w, g)
                ref = lambda: K.conv2d_dgrad(dyr, wr, gf)
            else:
                fn = lambda: (K.conv2d_wgrad_bf16r(xh, dyh, g), None)
                old = lambda: K.conv2d_wgrad(x, dy, g)
                ref = lambda: K.conv2d_wgrad(xr, dyr, gf)
            err = float("nan")
            if check:
                out = fn()
                r = ref()
                err = float((out[0] - r).abs().max() / r.abs().max())
                assert err < 2e-5, (name, kind, err)
                if out[1] is not None:
                    errh = float((out[1].float() - r).abs().max() / r.abs().max())
                    assert errh < 6e-3, (name, kind, errh)       # bf16 copy: 2^-8 relative of the largest value
                if kind == 2:                                    # "add into": twice the gradient
                    acc = r.clone()
                    K.conv2d_wgrad_bf16r(xh, dyh, g, into=acc)
                    assert float((acc - 2 * r).abs().max() / r.abs().max()) < 4e-5, name
            t = timeit(fn, 10)
            to = timeit(old, 5)
            line += " %8.3f %7.1f %8.3f %9.2e |" % (t, flop / t / 1e9, to, err)
            tot["fdw"[kind]] += t * cnt
            tot["fdw"[kind] + "o"] += to * cnt
        if cnt and K.bf16r_served(g, 0):
            totflop += flop * cnt
        print(line, flush=True)
    if tot["f"] > 0:
        print("segmenter layers served by the resident kernels: fwd %.2f ms (staged-rounding kernels %.2f), dgrad %.2f (%.2f), wgrad %.2f (%.2f); fwd "
              "%.1f TF/s = %.1f%% of %.0f" % (tot["f"], tot["fo"], tot["d"], tot["do"], tot["w"], tot["wo"], totflop / tot["f"] / 1e9,
                                              100 * totflop / tot["f"] / 1e9 / PEAK, PEAK))


if __name__ == "__main__":
    main()
