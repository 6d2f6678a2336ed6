"""worker of tests/test_gpu_wino.py::test_persistent_gemm_and_its_tail_split (run with PNP_WINO_TAILSPLIT=0 / 1): the BASELINE-batch layers
whose tile counts make the route's GEMM launch persistent (more tiles than workgroup slots) — and, with the switch on, cut an XCD's tail
tiles into pieces of the reduction — forward and data gradient against the direct kernels (bar 2e-5 of max|ref|: both sides are fp32) and
the workspace the C-ABI asks for"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "medical-cross-modality-domain-adaptation_amd"


def main():
    split = os.environ.get("PNP_WINO_TAILSPLIT") == "1"
    K, L = importlib.import_module(PKG + ".kernels"), importlib.import_module(PKG + "._lib")
    lib = L.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    K.wino_mode(2)
    K.wino_tile(4)
    K.wino_x3(0)                # the persistent launch and its tail split belong to the fp32-pipe GEMM (wino_gemm_kernel)
    worst = 0.0
    # (N, H, C, K, dilation, padding, pieces per tail tile when the split is on: tiles / 8 per XCD = whole rounds of 64 + R, s R <= 64)
    for (N, H, C, Kf, dil, pad, s_want) in ((16, 32, 512, 512, 1, "SAME", 4), (16, 32, 256, 256, 1, "SAME", 4), (8, 32, 512, 512, 2, "SAME", 8),
                                            (4, 64, 256, 256, 1, "SAME", 4), (6, 34, 512, 2560, 1, "VALID", 4), (5, 30, 512, 544, 1, "SAME", 1)):      # (the last: 540 tiles, not a multiple of 8 XCDs: never split)
        x = rng.standard_normal((N, H, H, C)).astype(np.float32)
        w = (rng.standard_normal((3, 3, C, Kf)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
        g = K.conv_geom(x.shape, w.shape, 1, dil, pad)
        dy = rng.standard_normal((N, g.OH, g.OW, Kf)).astype(np.float32)
        xd, wd, dyd = (torch.from_numpy(a).to(dev) for a in (x, w, dy))
        assert K.wino_chosen(g, 0) == 4
        T = N * dil * dil * (-(-(g.OH // dil) // 4)) * (-(-(g.OW // dil) // 4))
        ntiles = 36 * (-(-T // 128)) * (-(-Kf // 128))
        assert ntiles > 512, ntiles                                   # persistent launch
        base = 36 * 4 * (C * Kf + T * C + T * Kf)
        ws = int(lib.pnp_conv2d_fwd_workspace_bytes(ctypes.byref(g)))
        R = ntiles // 8 - (ntiles // 8 // 64) * 64
        want = base + (8 * R * (s_want - 1) * 128 * 128 * 4 if (split and ntiles % 8 == 0 and R) else 0)
        assert ws == want, (N, H, C, Kf, ws - base, want - base)
        y, dx = K.conv2d_fwd(xd, wd, g), K.conv2d_dgrad(dyd, wd, g)
        K.wino_mode(0)
        y0, dx0 = K.conv2d_fwd(xd, wd, g), K.conv2d_dgrad(dyd, wd, g)
        K.wino_mode(2)
        ey = float((y - y0).abs().max() / y0.abs().max())
        ed = float((dx - dx0).abs().max() / dx0.abs().max())
        print("persistent GEMM%s %s: %d tiles, y vs direct %.2e, dx vs direct %.2e" % (" + tail split" if split else "", (N, H, C, Kf, dil, pad), ntiles, ey, ed))
        worst = max(worst, ey, ed)
    assert worst < 2e-5, worst
    print("SPLIT WORKER OK")


if __name__ == "__main__":
    main()
