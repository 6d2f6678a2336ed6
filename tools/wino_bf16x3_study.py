"""Tolerance study for split-bf16 operands in the Winograd F(4x4, 3x3) GEMMs: CPU only, torch-CPU.

gfx950 has no tf32 and its fp32 MFMA runs at 1/16 of the bf16 rate.  A float32 value is the exact sum of three bf16 values
(hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid): 3 x 8 significand bits), so the product of two float32 values is the sum of
9 bf16 x bf16 products, each exact in fp32:
    x9: all of them                                      (the 24-bit product, exactly)
    x6: hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi     (drops terms <= 2^-27 of the product)
    x3: hi.hi, hi.mid, mid.hi                            (drops terms ~ 2^-18)
What the study measures: max|y - y64| / max|y64| of the whole F(4x4, 3x3) layer against the float64 convolution (the figure of
tools/wino_f43_study.py / profiles/r05_wino_f43_tolerance.txt) when ONLY the contraction M[pos] = V[pos] x U[pos] changes:
  * `fp32 seq`      one fp32 rounding per multiply-add, channels in order (the model of v_mfma_f32_32x32x2_f32's accumulation chain)
  * `fp32 MKL`      torch.matmul (blocked: r05's table)
  * `xN blk16`      per 16-channel block and per kept term the 16 products are summed exactly and added to the fp32 accumulator with one
                    rounding (the optimistic model of v_mfma_f32_32x32x16_bf16: one rounding per instruction)
  * `xN seq`        one rounding per bf16 product (the pessimistic model)
  * `... /flush64`  two-level accumulation: the MFMA accumulator is added into a second fp32 register set and cleared every 64 channels
                    (the error of a length-C chain ~ sqrt(C) eps |sum| becomes ~ (sqrt(64) sqrt(64 / C) + sqrt(C / 64)) eps |sum|)
The hardware's own summation order inside one MFMA is not documented: the GPU tests (tests/test_gpu_wino.py) hold the kernel to float64
directly; this table says what to expect and which variant can meet "<= 1.1x native fp32 F(4x4)".

  python tools/wino_bf16x3_study.py > profiles/r06_wino_bf16x3_tolerance.txt
"""
import argparse
import os
import sys
from fractions import Fraction

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from wino_f43_study import toom_cook  # noqa: E402

TERMS = {
    9: [(i, j) for i in range(3) for j in range(3)],
    6: [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)],
    3: [(0, 0), (0, 1), (1, 0)],
}


def split3(v):
    hi = v.bfloat16().float()
    r = v - hi
    mid = r.bfloat16().float()
    lo = (r - mid).bfloat16().float()
    return [hi, mid, lo]


def gemm_fp32_seq(V, U):
    """[P][T][C] x [P][C][K], one rounding per fused multiply-add"""
    acc = torch.zeros(V.shape[0], V.shape[1], U.shape[2], dtype=torch.float32)
    Vd, Ud = V.double(), U.double()
    for c in range(V.shape[2]):
        acc = (acc.double() + Vd[:, :, c:c + 1] * Ud[:, c:c + 1, :]).float()
    return acc


def gemm_split(V, U, nterm, blk, flush=0, small_first=True):
    """blk = 16: one rounding per (term, 16-channel block); blk = 1: one rounding per product.  flush > 0: second-level accumulator"""
    Vp = [p.double() for p in split3(V)]
    Up = [p.double() for p in split3(U)]
    terms = TERMS[nterm]
    if small_first:
        terms = sorted(terms, key=lambda ij: -(ij[0] + ij[1]))          # the kernel issues the small terms first
    C = V.shape[2]
    acc = torch.zeros(V.shape[0], V.shape[1], U.shape[2], dtype=torch.float32)
    total = torch.zeros_like(acc)
    for c0 in range(0, C, blk):
        for (i, j) in terms:
            part = torch.matmul(Vp[i][:, :, c0:c0 + blk], Up[j][:, c0:c0 + blk, :])          # exact in float64
            acc = (acc.double() + part).float()
        if flush and (c0 + blk) % flush == 0:
            total = total + acc
            acc = torch.zeros_like(acc)
    return total + acc if flush else acc


def layer_error(x, w, mats, gemm, N, th, tw):
    AT, G, BT = (torch.from_numpy(a).float() for a in mats)
    C, K = w.shape[2], w.shape[3]
    n = 6
    xp = F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1))
    pt = xp.unfold(2, n, 4).unfold(3, n, 4)
    d = pt.permute(4, 5, 0, 2, 3, 1).reshape(n, n, -1, C)
    V = torch.einsum("iptc,qp->iqtc", torch.einsum("ip,pjtc->ijtc", BT, d), BT).reshape(36, -1, C)
    U = torch.einsum("isck,js->ijck", torch.einsum("ir,rsck->isck", G, w), G).reshape(36, C, K)
    Mm = gemm(V, U).reshape(n, n, -1, K)
    o = torch.einsum("pjtk,qj->pqtk", torch.einsum("pi,ijtk->pjtk", AT, Mm), AT)
    return o.reshape(4, 4, N, th, tw, K).permute(2, 3, 0, 4, 1, 5).reshape(N, th * 4, tw * 4, K)


LAYERS = [
    ("512->512 @32^2 (g7-g9)", 1, 32, 32, 512, 128),
    ("256->256 @32^2 (g6)", 1, 32, 32, 256, 128),
    ("2560->512 @32^2 (g10 data gradient)", 1, 16, 16, 2560, 64),
    ("128->128 @32^2", 1, 32, 32, 128, 128),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    mats = toom_cook((0, 1, -1, Fraction(1, 2), -2), 4)
    print("# F(4x4, 3x3) on the points (0, +-1, 1/2, -2, inf): max|y - y64| / max|y64| against the float64 convolution, by the arithmetic of")
    print("# the 36 GEMMs M[pos] = V[pos] x U[pos] (transforms in float32 in every row; K reduced to 64-128 filters: the figure does not depend on it)")
    variants = [
        ("fp32 seq (model of the fp32 MFMA chain)", lambda V, U: gemm_fp32_seq(V, U)),
        ("fp32 MKL (r05's table)", lambda V, U: torch.matmul(V, U)),
        ("fp64 GEMM (transforms' own error)", lambda V, U: torch.matmul(V.double(), U.double()).float()),
        ("bf16 x9 blk16", lambda V, U: gemm_split(V, U, 9, 16)),
        ("bf16 x6 blk16", lambda V, U: gemm_split(V, U, 6, 16)),
        ("bf16 x6 blk16 /flush64", lambda V, U: gemm_split(V, U, 6, 16, 64)),
        ("bf16 x6 blk16 /flush128", lambda V, U: gemm_split(V, U, 6, 16, 128)),
        ("bf16 x6 seq", lambda V, U: gemm_split(V, U, 6, 1)),
        ("bf16 x6 seq /flush64", lambda V, U: gemm_split(V, U, 6, 1, 64)),
        ("bf16 x3 blk16", lambda V, U: gemm_split(V, U, 3, 16)),
    ]
    if a.quick:
        variants = [v for v in variants if "seq" not in v[0] or "fp32" in v[0]]
    for lname, N, H, W, C, K in (LAYERS[:1] if a.quick else LAYERS):
        g = torch.Generator().manual_seed(C + K + H)
        x = F.leaky_relu(torch.randn(N, H, W, C, generator=g), 0.2)
        w = torch.randn(3, 3, C, K, generator=g) * (2.0 / (9 * C)) ** 0.5
        y64 = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
        print(f"\n## {lname}   N={N} K={K}")
        for vname, fn in variants:
            y = layer_error(x, w, mats, fn, N, H // 4, W // 4)
            e = float((y.double() - y64).abs().max() / y64.abs().max())
            r = float(((y.double() - y64) ** 2).mean().sqrt() / (y64 ** 2).mean().sqrt())
            print(f"{vname:44s} max {e:9.2e}   rms {r:9.2e}", flush=True)


if __name__ == "__main__":
    main()
