"""worker of tests/test_gpu_elementwise.py::test_one_launch_combine_rearms_its_tickets (run with PNP_BN_ONE_LAUNCH=1)"""
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
    assert os.environ.get("PNP_BN_ONE_LAUNCH") == "1"
    K, L = importlib.import_module(PKG + ".kernels"), importlib.import_module(PKG + "._lib")
    lib = L.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    shapes = [(40000, 512), (16384, 64), (300, 32), (65536, 128)]            # 625 / 256 / 5 / 1024 partial rows: 5 / 2 / 1 / 8 slabs
    data, first = {}, {}
    for (P, C) in shapes:
        x = torch.from_numpy((rng.standard_normal((P, C)) + 0.3).astype(np.float32)).to(dev)
        d = torch.from_numpy(rng.standard_normal((P, C)).astype(np.float32)).to(dev)
        data[(P, C)] = (x, d, torch.ones(C, device=dev), torch.zeros(C, device=dev))
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    L.prof_summary()
    L.prof_enable(0)
    for it in range(320):          # 640 launches: more than the 512 ticket slots
        P, C = shapes[it % len(shapes)]
        x, d, gamma, beta = data[(P, C)]
        mean = torch.full((C,), float("nan"), device=dev)
        var = torch.full((C,), float("nan"), device=dev)
        L.check(lib.pnp_bn_stats(K._p(x), K._p(mean), K._p(var), P, C, ctypes.c_void_p(ws.data_ptr()), ws.numel(), K._stream()), "pnp_bn_stats")
        dg = torch.full((C,), float("nan"), device=dev)
        db = torch.full((C,), float("nan"), device=dev)
        L.check(lib.pnp_bn_bwd_reduce(K._p(d), None, K._p(x), K._p(mean), K._p(var), K._p(gamma), K._p(beta), K._p(dg), K._p(db), P, C, 1e-3, 0.2,
                                      ctypes.c_void_p(ws.data_ptr()), ws.numel(), K._stream()), "pnp_bn_bwd_reduce")
        got = torch.stack([mean, var, dg, db])
        assert bool(torch.isfinite(got).all()), (it, P, C)
        if (P, C) in first:
            assert torch.equal(got, first[(P, C)]), (it, P, C)
        else:
            first[(P, C)] = got
            x64 = x.double()
            assert float((mean.double() - x64.mean(0)).abs().max()) < 1e-6 and float((var.double() - x64.var(0, unbiased=False)).abs().max()) < 1e-5
            g = torch.where(x64 * 1.0 / torch.sqrt(x64.var(0, unbiased=False) + 1e-3) - x64.mean(0) / torch.sqrt(x64.var(0, unbiased=False) + 1e-3) > 0,
                            d.double(), d.double() * 0.2)
            xh = (x64 - mean.double()) / torch.sqrt(var.double() + 1e-3)
            assert float((db.double() - g.sum(0)).abs().max() / g.sum(0).abs().max()) < 1e-4
            assert float((dg.double() - (g * xh).sum(0)).abs().max() / (g * xh).sum(0).abs().max()) < 1e-4
    print("TICKETS OK")


if __name__ == "__main__":
    main()
