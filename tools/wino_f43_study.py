"""Tolerance study for a Winograd F(4x4, 3x3) route: CPU only, numpy / torch-CPU.

For every layer class the F(2x2, 3x3) route serves, restate the minimal filtering algorithm F(m x m, 3x3) for m = 2 and m = 4 with all
intermediates (V, U, M, y) rounded to float32 and the contraction as an fp32 GEMM, and report max|y - y64| / max|y64| against the float64
convolution of the same float32 operands, forward and filter gradient, next to the direct fp32 convolution.  Transform matrices are
generated from the interpolation points (Toom-Cook: A^T = E_m^T, G = E_r scaled, B^T = V^-T scaled), so several point sets can be compared;
the classic set (0, +-1, +-2) reproduces Lavin & Gray's matrices.

  python tools/wino_f43_study.py [--quick] > profiles/r05_wino_f43_tolerance.txt
"""
import argparse
import itertools
from fractions import Fraction

import numpy as np
import torch
import torch.nn.functional as F


def toom_cook(points, m, r=3):
    """transform matrices of F(m, r) for the finite interpolation points + infinity.  Returns (AT [m x n], G [n x r], BT [n x n]) as float64,
    with Lavin's scaling: G row j divided by N_j = prod_{l != j} (p_j - p_l), B^T row j multiplied by it."""
    n = m + r - 1
    assert len(points) == n - 1
    p = [Fraction(q) for q in points]
    V = [[q ** k for k in range(n)] for q in p] + [[Fraction(0)] * (n - 1) + [Fraction(1)]]
    # exact inverse by Gauss-Jordan over the rationals
    M = [row[:] + [Fraction(int(i == j)) for j in range(n)] for i, row in enumerate(V)]
    for c in range(n):
        piv = next(i for i in range(c, n) if M[i][c] != 0)
        M[c], M[piv] = M[piv], M[c]
        M[c] = [v / M[c][c] for v in M[c]]
        for i in range(n):
            if i != c and M[i][c] != 0:
                M[i] = [a - M[i][c] * b for a, b in zip(M[i], M[c])]
    Vinv = [row[n:] for row in M]
    BT = [[Vinv[k][j] for k in range(n)] for j in range(n)]               # V^-T
    G = [[q ** k for k in range(r)] for q in p] + [[Fraction(0)] * (r - 1) + [Fraction(1)]]
    AT = [[(q ** i) for q in p] + [Fraction(int(i == m - 1))] for i in range(m)]
    for j in range(n - 1):
        Nj = Fraction(1)
        for l in range(n - 1):
            if l != j:
                Nj *= p[j] - p[l]
        BT[j] = [v * Nj for v in BT[j]]
        G[j] = [v / Nj for v in G[j]]
    f = lambda A: np.array([[float(v) for v in row] for row in A], np.float64)
    return f(AT), f(G), f(BT)


def check_exact(AT, G, BT, m):
    rng = np.random.default_rng(0)
    g, d = rng.standard_normal(3), rng.standard_normal(m + 2)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(3)) for i in range(m)])
    assert np.abs(y - ref).max() < 1e-9, (y, ref)


def wino_conv(x, w, m, mats, dt=torch.float32):
    """x [N][H][W][C], w [3][3][C][K] float32 tensors; SAME, stride 1; H, W multiples of m.  Every intermediate rounded to dt."""
    AT, G, BT = (torch.from_numpy(a).to(dt) for a in mats)
    N, H, W, C = x.shape
    K = w.shape[3]
    n = m + 2
    xp = F.pad(x.to(dt).permute(0, 3, 1, 2), (1, 1, 1, 1))              # [N][C][H+2][W+2]
    pt = xp.unfold(2, n, m).unfold(3, n, m)                               # [N][C][th][tw][n][n]
    th, tw = pt.shape[2], pt.shape[3]
    d = pt.permute(4, 5, 0, 2, 3, 1).reshape(n, n, -1, C)                 # [n][n][T][C]
    r = torch.einsum("ip,pjtc->ijtc", BT, d)                              # rows, then columns (the kernels' order)
    V = torch.einsum("ipjc,qp->iqjc", r.permute(0, 1, 2, 3), BT) if False else torch.einsum("iptc,qp->iqtc", r, BT)
    g1 = torch.einsum("ir,rsck->isck", G, w.to(dt))
    U = torch.einsum("isck,js->ijck", g1, G)
    Mm = torch.matmul(V.reshape(n * n, -1, C), U.reshape(n * n, C, K)).reshape(n, n, -1, K)
    r2 = torch.einsum("pi,ijtk->pjtk", AT, Mm)
    o = torch.einsum("pjtk,qj->pqtk", r2, AT)                             # [m][m][T][K]
    y = o.reshape(m, m, N, th, tw, K).permute(2, 3, 0, 4, 1, 5).reshape(N, th * m, tw * m, K)
    return y


def wino_wgrad(x, dy, m, mats, dt=torch.float32):
    """dW = G^T [ sum_t (B^T d B) (.) (A dy A^T) ] G, intermediates in dt"""
    AT, G, BT = (torch.from_numpy(a).to(dt) for a in mats)
    N, H, W, C = x.shape
    K = dy.shape[3]
    n = m + 2
    xp = F.pad(x.to(dt).permute(0, 3, 1, 2), (1, 1, 1, 1))
    pt = xp.unfold(2, n, m).unfold(3, n, m)
    th, tw = pt.shape[2], pt.shape[3]
    d = pt.permute(4, 5, 0, 2, 3, 1).reshape(n, n, -1, C)
    V = torch.einsum("iptc,qp->iqtc", torch.einsum("ip,pjtc->ijtc", BT, d), BT)
    yt = dy.to(dt).reshape(N, th, m, tw, m, K).permute(2, 4, 0, 1, 3, 5).reshape(m, m, -1, K)
    Y = torch.einsum("pjtk,pi->ijtk", torch.einsum("pqtk,qj->pjtk", yt, AT), AT)     # A y A^T  [n][n][T][K]
    S = torch.matmul(V.reshape(n * n, -1, C).transpose(1, 2), Y.reshape(n * n, -1, K)).reshape(n, n, C, K)
    return torch.einsum("isck,ir->rsck", torch.einsum("ijck,js->isck", S, G), G)


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


POINT_SETS = {
    "F(2,3) 0,+-1": (2, (0, 1, -1)),
    "F(4,3) 0,+-1,+-2 (Lavin)": (4, (0, 1, -1, 2, -2)),
    "F(4,3) 0,+-1,+-1/2": (4, (0, 1, -1, Fraction(1, 2), Fraction(-1, 2))),
    "F(4,3) 0,+-1,1/2,-2": (4, (0, 1, -1, Fraction(1, 2), -2)),
    "F(4,3) 0,+-1/2,+-2": (4, (0, Fraction(1, 2), Fraction(-1, 2), 2, -2)),
    "F(4,3) 0,+-1,2,-1/2": (4, (0, 1, -1, 2, Fraction(-1, 2))),
}

# (N, H, W, C, K): the layer classes on the F(2,3) route at B = 16 (N reduced where the figure does not depend on it)
LAYERS = [
    ("512->512 @32^2 (g7-g9)", 8, 32, 32, 512, 512),
    ("256->512 @32^2 (g7 first)", 8, 32, 32, 256, 512),
    ("256->256 @32^2 (g6)", 8, 32, 32, 256, 256),
    ("512->2560 @32^2 (g10, SAME here)", 2, 32, 32, 512, 2560),
    ("256->256 @64^2 (critic cls_3)", 2, 64, 64, 256, 256),
    ("512->512 @16^2 (critic)", 8, 16, 16, 512, 512),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    mats = {}
    for name, (m, pts) in POINT_SETS.items():
        mats[name] = (m, toom_cook(pts, m))
        check_exact(*mats[name][1], m)
    print("# Winograd F(m x m, 3x3) in float32 against the float64 convolution of the same operands: max|err| / max|ref|")
    print("# x = leaky_relu(N(0,1), 0.2)-like activations, w = N(0, 2/(9C)), dy = N(0,1); every intermediate (V, U, M, y) rounded to float32")
    layers = LAYERS[:2] if a.quick else LAYERS
    for lname, N, H, W, C, K in layers:
        g = torch.Generator().manual_seed(C + K + H)
        x = F.leaky_relu(torch.randn(N, H, W, C, generator=g), 0.2)
        w = torch.randn(3, 3, C, K, generator=g) * (2.0 / (9 * C)) ** 0.5
        dy = torch.randn(N, H, W, K, generator=g)
        x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
        w64 = w.double().permute(3, 2, 0, 1).requires_grad_(True)
        y64 = F.conv2d(x64, w64, padding=1)
        y64.backward(dy.double().permute(0, 3, 1, 2))
        yref = y64.detach().permute(0, 2, 3, 1)
        dwref = w64.grad.permute(2, 3, 1, 0)
        dxref = x64.grad.permute(0, 2, 3, 1)
        xt = x.permute(0, 3, 1, 2).requires_grad_(True)
        wt = w.permute(3, 2, 0, 1).requires_grad_(True)
        yd = F.conv2d(xt, wt, padding=1)
        yd.backward(dy.permute(0, 3, 1, 2))
        print(f"\n## {lname}   N={N}")
        print(f"{'algorithm':34s} {'forward':>10s} {'data grad':>10s} {'filter grad':>11s}")
        print(f"{'direct fp32 (oneDNN)':34s} {rel(yd.detach().permute(0, 2, 3, 1), yref):10.2e} {rel(xt.grad.permute(0, 2, 3, 1), dxref):10.2e} "
              f"{rel(wt.grad.permute(2, 3, 1, 0), dwref):11.2e}")
        wflip = torch.flip(w, (0, 1)).transpose(2, 3).contiguous()
        for name, (m, mm) in mats.items():
            ef = rel(wino_conv(x, w, m, mm), yref)
            ed = rel(wino_conv(dy, wflip, m, mm), dxref)
            ew = rel(wino_wgrad(x, dy, m, mm), dwref)
            print(f"{name:34s} {ef:10.2e} {ed:10.2e} {ew:11.2e}")


if __name__ == "__main__":
    main()
