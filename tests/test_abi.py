"""not gpu: the C-ABI library builds for gfx950, loads, and exports exactly the symbols include/pnp_hip.h declares;
the product path fails loudly (no CPU fallback) when handed CPU tensors."""
import os
import re

import pytest
import torch

from conftest import ROOT, pkg


def header_symbols():
    src = open(os.path.join(ROOT, "include", "pnp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pnp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol(built):
    lib = built._lib.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "libpnp_hip.so does not export %s" % s
    assert sorted(built._lib.PROTOTYPES.keys()) == syms          # ctypes prototypes cover the header exactly
    assert lib.pnp_abi_version() == built._lib.ABI_VERSION == 4


def test_workspace_queries_and_error_channel(built):
    import ctypes
    L = built._lib
    lib = L.load()
    K = pkg("kernels")
    g = K.conv_geom((16, 32, 32, 512), (3, 3, 512, 2560), 1, 1, "SYMMETRIC")
    assert (g.OH, g.OW, g.pad_t, g.pad_mode) == (32, 32, 1, L.PAD_SYMMETRIC)
    need = lib.pnp_conv2d_dgrad_workspace_bytes(ctypes.byref(g))
    assert need >= 3 * 3 * 512 * 2560 * 4 + 16 * 34 * 34 * 512 * 4
    assert lib.pnp_bn_workspace_bytes(16 * 256 * 256, 16) > 0
    # bad geometry is rejected before any launch, with a message
    bad = K.conv_geom((1, 8, 8, 4), (3, 3, 4, 4), 1, 1, "SAME")
    bad.K = 0
    rc = lib.pnp_conv2d_fwd(None, None, None, ctypes.byref(bad), 1.0, 0, 0, None)
    assert rc == -1 and b"non-positive" in lib.pnp_last_error()


def test_no_cpu_fallback(built):
    K = pkg("kernels")
    x = torch.zeros((1, 8, 8, 4))
    w = torch.zeros((3, 3, 4, 4))
    g = K.conv_geom(tuple(x.shape), tuple(w.shape))
    with pytest.raises(built._lib.PnpError):
        K.conv2d_fwd(x, w, g)          # CPU tensors: must raise, never compute on the host


def test_product_never_imports_oracle():
    pdir = os.path.join(ROOT, "medical-cross-modality-domain-adaptation_amd")
    for dp, _, files in os.walk(pdir):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                assert "oracle" not in open(os.path.join(dp, f)).read().replace("oracle/dropout.py", "").replace("oracle/tf_ops.py", ""), f


def test_torch_is_loaded_before_the_library():
    """a fresh interpreter that imports ONLY the binding must end up with torch's HIP runtime mapped before libpnp_hip.so: loaded the
    other way round, the system libamdhip64 becomes a second runtime in the process and every launch reports 'no ROCm-capable device'"""
    import subprocess
    import sys
    from conftest import PKG, ROOT
    code = ("import sys, importlib; sys.path.insert(0, %r); L = importlib.import_module(%r + '._lib'); "
            "assert 'torch' in sys.modules; L.load(); "
            "hip = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}); print(hip); "
            "assert len(hip) == 1, hip") % (ROOT, PKG)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr


def test_host_planners_through_the_workspace_queries(built):
    """the reduction-split / stride-phase / direct-kernel planners are host code: their decisions show in the workspace sizes
    (no GPU needed).  B=16 shapes of the hot path."""
    import ctypes
    lib = built._lib.load()
    K = pkg("kernels")
    prev, prev_w = K.wino_mode(0), K.wino_wgrad_mode(0)     # the planners of the DIRECT kernels (the Winograd route's own: test_winograd_route_planner_on_the_host)
    try:
        _direct_planners(lib, K, ctypes)
    finally:
        K.wino_mode(prev)
        K.wino_wgrad_mode(prev_w)


def _direct_planners(lib, K, ctypes):
    def geo(N, H, C, Kc, R, stride=1, dil=1, padding="SAME"):
        return K.conv_geom((N, H, H, C), (R, R, C, Kc), stride, dil, padding)

    fwd = lambda g: lib.pnp_conv2d_fwd_workspace_bytes(ctypes.byref(g))
    dgr = lambda g: lib.pnp_conv2d_dgrad_workspace_bytes(ctypes.byref(g))
    wgr = lambda g: lib.pnp_conv2d_wgrad_workspace_bytes(ctypes.byref(g))
    big = geo(16, 32, 512, 512, 3)
    filt = lambda g: ((g.R * g.S * g.C * g.K * 4 + 255) // 256) * 256
    assert fwd(big) == 0                                     # 512 tiles: a full dispatch round, never split
    assert dgr(big) == filt(big)                             # only the flipped filters
    assert wgr(big) == 7 * 3 * 3 * 512 * 512 * 4             # 144 tiles x 7 splits = 1.97 rounds
    g4 = geo(16, 32, 128, 128, 3)
    assert fwd(g4) == 0 and dgr(g4) == filt(g4)              # 256 tiles = half a round: splitting loses (measured)
    tiny = geo(16, 16, 512, 512, 5, stride=4)                # critic cls_5: 4x4 maps, 8 tiles
    out_elems = 16 * 4 * 4 * 512
    assert fwd(tiny) % (out_elems * 4) == 0 and 8 <= fwd(tiny) // (out_elems * 4) <= 32
    # strided data gradient by stride phases: filters + the largest phase's partials, far less than the zero-upsampled formulation
    k5s2 = geo(16, 128, 128, 128, 5, stride=2)
    assert filt(k5s2) <= dgr(k5s2) < filt(k5s2) + 2 * 16 * 128 * 128 * 128 * 4
    # g10's data gradient on the pre-padded input: 580 tiles = 1.13 rounds -> 7 splits (7.93 rounds)
    g10 = K.conv_geom((16, 34, 34, 512), (3, 3, 512, 2560), 1, 1, "VALID")
    assert dgr(g10) == ((3 * 3 * 512 * 2560 * 4 + 255) // 256) * 256 + 7 * 16 * 34 * 34 * 512 * 4
    # 16 -> 16 channels at 256^2: the 16x16x4-MFMA filter gradient (conv_small.hip), one partial per workgroup, 1024 workgroups
    g1 = geo(16, 256, 16, 16, 3)
    assert wgr(g1) == 1024 * 3 * 3 * 16 * 16 * 4
    g1a = geo(16, 256, 3, 16, 3)          # the first layer (3 input channels) likewise
    assert wgr(g1a) == 1024 * 3 * 3 * 3 * 16 * 4
    # (the direct vector-ALU filter gradient keeps the other K <= 16 layers: one partial per workgroup, 2048 slabs — `logits` below)
    logits = K.conv_geom((16, 260, 260, 40), (5, 5, 40, 5), 1, 1, "VALID")
    assert wgr(logits) == 2048 * 5 * 5 * 40 * 5 * 4
    # PNP_DTYPE_BF16 permits bf16 operands; the 16-channel layers keep conv_small.hip's fp32 plan (round 3, run 14), wide layers do not
    L = pkg("_lib")
    g1b = K.conv_geom((16, 256, 256, 16), (3, 3, 16, 16), 1, 1, "SAME", dtype=L.DTYPE_BF16)
    assert wgr(g1b) == wgr(g1)
    g2b = K.conv_geom((16, 128, 128, 32), (3, 3, 32, 32), 1, 1, "SAME", dtype=L.DTYPE_BF16)
    g2f = geo(16, 128, 32, 32, 3)
    assert wgr(g2b) == wgr(g2f) and wgr(g2f) % (3 * 3 * 32 * 32 * 4) == 0
    assert lib.pnp_conv2d_dgrad_workspace_bytes(None) == 0 and lib.pnp_conv2d_fwd_workspace_bytes(None) == 0


def test_documents_quote_the_current_abi_size(built):
    n = len(built._lib.PROTOTYPES)
    for doc in ("README.md", "DESIGN.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        assert ("%d entry points" % n in text) or ("%d `extern \"C\"` entry points" % n in text) or ("all %d symbols" % n in text), doc


def test_every_environment_switch_is_documented():
    """INTEGRATION.md §E lists every PNP_* variable the library (getenv) and the Python host side (os.environ) read — an experiment switch
    that silently changes which kernel runs must not exist only in the source."""
    import glob
    pk = os.path.join(ROOT, "medical-cross-modality-domain-adaptation_amd")
    names = set()
    for f in glob.glob(os.path.join(pk, "csrc", "*.hip")) + glob.glob(os.path.join(pk, "csrc", "*.h")):
        names |= set(re.findall(r'(?:getenv|env_int|env_dbl)\("(PNP_[A-Z0-9_]+)"', open(f).read()))
    for f in glob.glob(os.path.join(pk, "*.py")):
        names |= set(re.findall(r'environ(?:\.get)?[\(\[]\s*"(PNP_[A-Z0-9_]+)"', open(f).read()))
    assert len(names) > 25, names
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = doc[doc.index("## E. Environment switches"):]
    missing = sorted(n for n in names if "`%s`" % n not in section)
    assert not missing, missing


def test_resident_bf16_planners_on_the_host(built):
    import ctypes
    """which geometries the bf16-resident kernels serve, their statistics-partials count and the filter gradient's workspace are host
    functions of the C-ABI (pnp_conv2d_bf16r_served / _fwd_bf16r_stats_parts / _wgrad_bf16r_workspace_bytes): pinned here without a GPU"""
    K, L = built.kernels, built._lib
    lib = L.load()
    geo = lambda N, H, C, Kf, k, stride=1, pad="SAME", dt=L.DTYPE_BF16: K.conv_geom((N, H, H, C), (k, k, C, Kf), stride, 1, pad, dtype=dt)
    srv = lambda g: tuple(int(lib.pnp_conv2d_bf16r_served(ctypes.byref(g), k)) for k in (0, 1, 2))
    assert srv(geo(16, 32, 512, 512, 3)) == (1, 1, 1)                       # group_7..9
    assert srv(geo(16, 34, 512, 2560, 3, pad="VALID")) == (1, 1, 1)         # group_10 after the mirror pre-pad
    assert srv(geo(16, 256, 32, 64, 3)) == (1, 0, 0)                        # C = 32: forward only (64-byte rows); K = 32 outputs / C % 64
    assert srv(geo(16, 256, 64, 64, 3, stride=2)) == (1, 1, 1)              # critic k3 s2: stride-phase data gradient, strided filter gradient
    assert srv(geo(16, 128, 128, 128, 5, stride=2)) == (1, 1, 1)            # k5 s2: phases (3,3) (3,2) (2,3) (2,2)
    assert srv(geo(16, 16, 512, 512, 5, stride=4)) == (0, 0, 0)             # 4x4 output maps: a handful of tiles, stays on the split kernels
    assert srv(geo(2, 32, 512, 512, 3)) == (0, 0, 0)                        # 2 048 pixels: below the 4 096-pixel floor
    assert srv(geo(16, 256, 40, 5, 5, pad="VALID")) == (0, 0, 0)            # the logits convolution
    assert srv(geo(16, 256, 16, 16, 3)) == (0, 0, 0)                        # 16-channel layers: conv_small.hip's fp32 tiles
    assert srv(geo(3, 37, 96, 128, 3)) [2] == 0                             # non-power-of-two map: no resident filter gradient
    # statistics partials: pixel tiles x wave rows of the tile the planner picks (256 x 128: 4 wave rows; 128-row tiles: 2)
    parts = lambda g: int(lib.pnp_conv2d_fwd_bf16r_stats_parts(ctypes.byref(g)))
    assert parts(geo(16, 32, 512, 512, 3)) == (16 * 32 * 32 // 256) * 4          # 256 tiles of 256 x 128
    assert parts(geo(16, 32, 256, 256, 3)) == (16 * 32 * 32 // 128) * 2          # 256 tiles of 128 x 128 would be one per CU: 128 x 64
    assert parts(geo(16, 256, 16, 16, 3)) == 0
    # filter gradient workspace = split count x |dW| x 4, with the split from the cost model (1 <= split <= 64; 0 bytes when un-split)
    ws = lambda g: int(lib.pnp_conv2d_wgrad_bf16r_workspace_bytes(ctypes.byref(g)))
    for g in (geo(16, 32, 512, 512, 3), geo(16, 256, 64, 64, 3), geo(16, 32, 256, 256, 3)):
        nout = g.R * g.S * g.C * g.K * 4
        assert ws(g) % nout == 0 and 0 <= ws(g) // nout <= 64
    assert ws(geo(16, 256, 64, 64, 3)) // (9 * 64 * 64 * 4) >= 16                # 9 tiles over 16 384 chunks: split deep
    assert ws(geo(16, 256, 16, 16, 3)) == 0 and lib.pnp_conv2d_wgrad_bf16r_workspace_bytes(None) == 0


def test_winograd_route_planner_on_the_host(built):
    """which layers the Winograd F(2x2, 3x3) route takes (csrc/conv_wino.hip), its workspace (transformed filter + input + product) and
    its statistics partial rows are host functions of the C-ABI: pinned here without a GPU, for every policy mode"""
    import ctypes
    import importlib
    K, L = importlib.import_module(built.__name__ + ".kernels"), built._lib
    lib = L.load()
    geo = lambda N, H, C, Kf, k=3, stride=1, dil=1, pad="SAME", dt=L.DTYPE_F32: K.conv_geom((N, H, H, C), (k, k, C, Kf), stride, dil, pad, dtype=dt)
    ch = lambda g: (bool(K.wino_chosen(g, 0)), bool(K.wino_chosen(g, 1)))
    prev, prev_t = K.wino_mode(-1), K.wino_tile(2)          # F(2x2, 3x3) alone first; F(4x4, 3x3): the next test
    prev_x3 = K.wino_x3(0)                                  # (the fp32-pipe plan; the split-bf16 GEMM's: test_winograd_x3_plan_on_the_host)
    try:
        K.wino_mode(0)
        assert ch(geo(16, 32, 512, 512)) == (False, False)
        g = geo(16, 32, 512, 512)
        direct_ws, direct_parts = int(lib.pnp_conv2d_fwd_workspace_bytes(ctypes.byref(g))), K.conv_stats_parts(g)
        assert direct_ws == 0 and direct_parts == int(lib.pnp_conv2d_fwd_stats_parts(ctypes.byref(g))) > 0
        K.wino_mode(1)
        assert ch(geo(16, 32, 512, 512)) == (True, True)                        # group_7..9
        assert ch(geo(16, 32, 512, 512, dil=2)) == (True, True)                 # group_8 (atrous rate 2)
        assert ch(geo(16, 32, 256, 256)) == (True, True) and ch(geo(16, 32, 256, 512)) == (True, True)
        assert ch(geo(16, 34, 512, 2560, pad="VALID")) == (True, True)          # group_10 on its mirror-padded input; dgrad: padding 2
        assert ch(geo(16, 32, 128, 256)) == (True, True)                        # C K / (C + K) = 85: break-even, measured
        assert ch(geo(16, 32, 128, 128)) == (False, False)                      # 64: the transforms cost more than they save
        assert ch(geo(16, 256, 64, 64)) == (False, False)
        assert ch(geo(16, 64, 256, 256, stride=2)) == (False, False)            # strided: direct / stride-phase kernels
        assert ch(geo(16, 128, 128, 128, k=5, stride=2)) == (False, False)
        assert ch(geo(16, 32, 512, 512, dt=L.DTYPE_BF16)) == (False, False)     # bf16 operands: conv_bf16 / conv_bf16r
        assert ch(geo(1, 8, 512, 512)) == (False, False)                        # 16 tiles: below the 512-tile floor
        assert ch(geo(16, 33, 512, 512, dil=2)) == (False, False)               # odd extent with dilation 2: no phase decomposition
        K.wino_mode(2)
        assert ch(geo(16, 32, 128, 128)) == (True, True) and ch(geo(1, 8, 64, 32)) == (True, True)
        assert ch(geo(16, 256, 16, 16)) == (False, False) and ch(geo(16, 256, 32, 16)) == (False, False)     # K = 16: conv_small / narrow kernels
        assert ch(geo(16, 256, 3, 32)) == (False, False)                        # C % 32
        # workspace = 16 x (C K + T C + T K) floats (each block 256-byte aligned), T = N d^2 ceil(OHs/2) ceil(OWs/2)
        K.wino_mode(1)
        for g, T in ((geo(16, 32, 512, 512), 16 * 16 * 16), (geo(16, 32, 512, 512, dil=2), 16 * 4 * 8 * 8), (geo(16, 34, 512, 2560, pad="VALID"), 16 * 16 * 16)):
            want = 16 * 4 * (g.C * g.K + T * g.C + T * g.K)
            assert int(lib.pnp_conv2d_fwd_workspace_bytes(ctypes.byref(g))) == want, (g.C, g.K)
        g10 = geo(16, 34, 512, 2560, pad="VALID")
        assert int(lib.pnp_conv2d_dgrad_workspace_bytes(ctypes.byref(g10))) == 16 * 4 * (2560 * 512 + 16 * 17 * 17 * (2560 + 512))
        # statistics partial rows = tile slabs of the output transform (~1024 per channel slice, a multiple of the tiles per pass)
        assert K.conv_stats_parts(geo(16, 32, 512, 512)) == 4096 // 4 and K.conv_stats_parts(geo(16, 34, 512, 2560, pad="VALID")) == 4096 // 4
        assert K.conv_stats_parts(geo(16, 32, 128, 128)) == int(lib.pnp_conv2d_fwd_stats_parts(ctypes.byref(geo(16, 32, 128, 128))))
        # the filter gradient has its own switch; workspace = V + Y + split partials of the 16 [C x K] products
        prev_w = K.wino_wgrad_mode(0)
        try:
            g = geo(16, 32, 512, 512)
            direct = int(lib.pnp_conv2d_wgrad_workspace_bytes(ctypes.byref(g)))
            assert not K.wino_chosen(g, 2) and direct % (9 * 512 * 512 * 4) == 0
            K.wino_wgrad_mode(1)
            assert K.wino_chosen(g, 2) and K.wino_chosen(geo(16, 32, 256, 256), 2) and not K.wino_chosen(geo(16, 32, 128, 256), 2)      # break-even 256 -> 256
            assert not K.wino_chosen(geo(16, 32, 128, 128), 2) and not K.wino_chosen(geo(16, 64, 256, 256, stride=2), 2)
            T = 16 * 16 * 16                    # 256 tiles of 128 x 128 x 16 points: split in two to fill a dispatch round
            assert int(lib.pnp_conv2d_wgrad_workspace_bytes(ctypes.byref(g))) == 16 * 4 * (T * 512 + T * 512 + 2 * 512 * 512)
            K.wino_mode(0)
            assert not K.wino_chosen(g, 2)              # PNP_WINOGRAD=0 is the master switch of the route
            K.wino_mode(1)
            g10 = geo(16, 34, 512, 2560, pad="VALID")       # 1280 tiles: un-split
            assert int(lib.pnp_conv2d_wgrad_workspace_bytes(ctypes.byref(g10))) == 16 * 4 * (T * 512 + T * 2560 + 512 * 2560)
        finally:
            K.wino_wgrad_mode(prev_w)
    finally:
        K.wino_mode(prev)
        K.wino_tile(prev_t)
        K.wino_x3(prev_x3)
    assert K.wino_mode(-1) == prev and K.wino_tile(-1) == prev_t


def test_winograd_f4_planner_on_the_host(built):
    """round 5: the route's second output tile, F(4x4, 3x3) (36 transform points, T = N d^2 ceil(OHs/4) ceil(OWs/4) tiles): which layers
    take it (pnp_conv2d_wino_chosen returns the tile edge), the fall-back to F(2x2) and to the direct kernels, workspace and partial rows"""
    import ctypes
    import importlib
    K, L = importlib.import_module(built.__name__ + ".kernels"), built._lib
    lib = L.load()
    geo = lambda N, H, C, Kf, k=3, stride=1, dil=1, pad="SAME", dt=L.DTYPE_F32: K.conv_geom((N, H, H, C), (k, k, C, Kf), stride, dil, pad, dtype=dt)
    ch = lambda g: (K.wino_chosen(g, 0), K.wino_chosen(g, 1), K.wino_chosen(g, 2))
    prev, prev_w, prev_t, prev_x3 = K.wino_mode(1), K.wino_wgrad_mode(1), K.wino_tile(4), K.wino_x3(0)
    try:
        assert K.wino_tile(-1) == 4
        assert ch(geo(16, 32, 512, 512)) == (4, 4, 4) and ch(geo(16, 32, 512, 512, dil=2)) == (4, 4, 4)
        assert ch(geo(16, 34, 512, 2560, pad="VALID")) == (4, 2, 4)             # group_10: its data gradient reduces over 2 560 channels -> F(2x2) (rounding)
        assert ch(geo(16, 32, 256, 256)) == (4, 4, 4)
        assert ch(geo(16, 32, 128, 256)) == (4, 4, 4)
        assert ch(geo(16, 32, 128, 128)) == (0, 0, 4)                           # C K / (C + K) = 64 >= 60, but 36 x 8 x 1 = 288 workgroups of 4 stages: only the filter gradient pays
        assert ch(geo(16, 128, 128, 128)) == (4, 4, 4)                          # the critics' 128 -> 128 at 128^2: 4 608 workgroups
        assert ch(geo(16, 128, 64, 128)) == (4, 4, 4)                           # 42.7, >= 32 on a large map: 4 608 workgroups, 16 384 tiles
        assert ch(geo(16, 32, 64, 128)) == (0, 0, 4)                            # the same layer at 32^2 (288 workgroups): only its filter gradient
        assert ch(geo(16, 256, 64, 64)) == (4, 4, 0)                            # cls1 64 -> 64 @256^2: 128 x 64 GEMM tiles; 65 536 tiles: filter gradient direct
        assert ch(geo(16, 64, 64, 64)) == (0, 0, 4) and ch(geo(16, 256, 32, 64)) == (0, 0, 0)        # 64 -> 64 @64^2; 32 -> 64 (21.3)
        assert ch(geo(16, 64, 256, 256, stride=2)) == (0, 0, 0)
        assert ch(geo(16, 32, 512, 512, dt=L.DTYPE_BF16)) == (0, 0, 0)
        assert ch(geo(2, 32, 512, 512)) == (2, 2, 4)                            # B = 2 per GPU: 36 x 1 x 4 = 144 workgroups do not cover the chip -> F(2x2) (16 x 4 x 4 = 256); filter gradient F(4x4)
        assert ch(geo(4, 32, 512, 512)) == (4, 4, 4) and ch(geo(2, 34, 512, 2560, pad="VALID")) == (4, 2, 4) and ch(geo(2, 32, 256, 256)) == (0, 0, 4)
        assert ch(geo(1, 32, 512, 512)) == (0, 0, 0)                            # 64 tiles of 4x4, 256 of 2x2: below both floors
        assert ch(geo(1, 45, 512, 512))[0] == 4 and ch(geo(1, 44, 512, 512))[0] == 0       # ragged: 12 x 12 = 144 tiles of 4x4 / 11 x 11 = 121 (484 of 2x2)
        K.wino_tile(2)
        assert ch(geo(16, 32, 512, 512)) == (2, 2, 2) and ch(geo(16, 32, 128, 128)) == (0, 0, 0)
        K.wino_tile(4)
        # workspace = 36 x (C K + T C + T K) floats, T = tiles of 4x4 outputs (+ the partial tiles of the GEMM's tail split when
        # PNP_WINO_TAILSPLIT=1: tests/wino_split_worker.py)
        for g, T in ((geo(16, 32, 512, 512), 16 * 8 * 8), (geo(16, 32, 512, 512, dil=2), 16 * 4 * 4 * 4), (geo(16, 34, 512, 2560, pad="VALID"), 16 * 8 * 8)):
            assert int(lib.pnp_conv2d_fwd_workspace_bytes(ctypes.byref(g))) == 36 * 4 * (g.C * g.K + T * g.C + T * g.K), (g.C, g.K)
        g10 = geo(16, 34, 512, 2560, pad="VALID")
        assert int(lib.pnp_conv2d_dgrad_workspace_bytes(ctypes.byref(g10))) == 16 * 4 * (2560 * 512 + 16 * 17 * 17 * (2560 + 512))    # F(2x2): 34 x 34 outputs, 17 x 17 tiles
        assert K.conv_stats_parts(geo(16, 32, 512, 512)) == 1024 // 2 and K.conv_stats_parts(g10) == 1024
        # filter gradient: V + Y + the split partials of the 36 [C x K] products; 16 x 36 = 576 workgroups = 2.25 per CU un-split: the cost
        # model splits the 32-stage reduction in two (4.5 per CU); g10 (80 x 36 workgroups) stays un-split
        g, T = geo(16, 32, 512, 512), 16 * 8 * 8
        assert int(lib.pnp_conv2d_wgrad_workspace_bytes(ctypes.byref(g))) == 36 * 4 * (T * 512 + T * 512 + 2 * 512 * 512)
        assert int(lib.pnp_conv2d_wgrad_workspace_bytes(ctypes.byref(g10))) == 36 * 4 * (T * 512 + T * 2560 + 512 * 2560)
        K.wino_mode(2)
        K.wino_wgrad_mode(2)
        assert ch(geo(1, 8, 64, 32)) == (4, 4, 4) and ch(geo(16, 256, 32, 16)) == (0, 0, 0)
    finally:
        K.wino_mode(prev)
        K.wino_wgrad_mode(prev_w)
        K.wino_tile(prev_t)
        K.wino_x3(prev_x3)


def test_winograd_x3_plan_on_the_host(built):
    """round 6: the split-bf16 GEMMs of the route (csrc/conv_wino_x3.hip; pnp_conv2d_wino_x3 / PNP_WINOGRAD_X3).  Mode 1 (the default) gives
    them the reductions over >= 256 channels, mode 2 every reduction over a multiple of 64 channels, mode 0 none; their operands are three
    bf16 planes (6 bytes per transformed value instead of 4) and the product M stays fp32; chunked accumulation lifts the 1 024-channel cap
    of F(4x4) on the fp32 pipe (group_10's data gradient takes F(4x4))"""
    import ctypes
    import importlib
    K, L = importlib.import_module(built.__name__ + ".kernels"), built._lib
    lib = L.load()
    geo = lambda N, H, C, Kf, dil=1, pad="SAME": K.conv_geom((N, H, H, C), (3, 3, C, Kf), 1, dil, pad)
    al = lambda b: (b + 255) // 256 * 256
    ws = lambda g: int(lib.pnp_conv2d_fwd_workspace_bytes(ctypes.byref(g)))
    f32 = lambda g, T: al(36 * 4 * g.C * g.K) + al(36 * 4 * T * g.C) + al(36 * 4 * T * g.K)
    x3 = lambda g, T: al(36 * 6 * g.C * g.K) + al(36 * 6 * T * g.C) + al(36 * 4 * T * g.K)
    prev, prev_t, prev_x3 = K.wino_mode(1), K.wino_tile(4), K.wino_x3(1)
    try:
        assert prev_x3 == 1, "the split-bf16 GEMM is the default arithmetic of the route"
        T = 16 * 8 * 8
        g512, g256, g128, g10 = geo(16, 32, 512, 512), geo(16, 32, 256, 512), geo(16, 128, 128, 128), geo(16, 34, 512, 2560, pad="VALID")
        assert ws(g512) == x3(g512, T) and ws(g256) == x3(g256, T) and ws(g10) == x3(g10, T)
        assert ws(g128) == f32(g128, 16 * 32 * 32)                               # 128-channel reduction: stays on the fp32 pipe under mode 1
        assert K.wino_chosen(g10, 1) == 4                                        # data gradient: reduction over 2 560 channels, chunked
        Td = 16 * 9 * 9                                                          # its output is the 34 x 34 mirror-padded map: 9 x 9 tiles of 4 x 4
        assert int(lib.pnp_conv2d_dgrad_workspace_bytes(ctypes.byref(g10))) == al(36 * 6 * 2560 * 512) + al(36 * 6 * Td * 2560) + al(36 * 4 * Td * 512)
        assert int(lib.pnp_conv2d_wino_filter_bytes(512, 512)) == 36 * 6 * 512 * 512          # an entry of the transformed-filter cache serves either format
        K.wino_x3(2)
        assert ws(g128) == x3(g128, 16 * 32 * 32)
        g96 = geo(4, 32, 96, 128)
        K.wino_mode(2)
        assert ws(g96) == f32(g96, 4 * 8 * 8)                                     # 96 channels: no multiple of 64 -> fp32 pipe in every mode
        K.wino_mode(1)
        K.wino_x3(0)
        assert ws(g512) == f32(g512, T) and K.wino_chosen(g10, 1) == 2
    finally:
        K.wino_mode(prev)
        K.wino_tile(prev_t)
        K.wino_x3(prev_x3)


@pytest.mark.parametrize("m", [2, 4])
def test_winograd_eligibility_and_tile_count_agree_with_the_restatement(built, m):
    """random geometries: the C planner's eligibility (mode 2 = wherever the geometry allows) is exactly the set of conditions the numpy
    restatement needs (stride 1, 3x3, zero padding 0 / dil / 2 dil on both axes, extents divisible by the dilation, C % 32, K % 4, K >= 32,
    fp32), and its workspace is 16 x (C K + T C + T K) floats with the restatement's tile count T"""
    import ctypes
    import importlib
    import numpy as np
    from oracle import tf_ops as T_
    K, L = importlib.import_module(built.__name__ + ".kernels"), built._lib
    lib = L.load()
    rng = np.random.default_rng(21)
    prev, prev_t, prev_x3 = K.wino_mode(2), K.wino_tile(m), K.wino_x3(0)
    np_ = (m + 2) ** 2
    try:
        seen = {True: 0, False: 0}
        for _ in range(400):
            N, H, W = int(rng.integers(1, 5)), int(rng.integers(2, 40)), int(rng.integers(2, 40))
            C, Kf = int(rng.choice([3, 16, 32, 64, 96, 100])), int(rng.choice([5, 16, 32, 36, 64, 130]))
            k, stride, dil = int(rng.choice([3, 3, 3, 5])), int(rng.choice([1, 1, 1, 2])), int(rng.choice([1, 1, 2, 3]))
            pad = int(rng.choice([0, dil, 2 * dil, 1]))
            g = L.ConvGeom()
            g.N, g.H, g.W, g.C, g.K, g.R, g.S = N, H, W, C, Kf, k, k
            g.stride, g.dil, g.pad_t, g.pad_l, g.pad_mode, g.dtype = stride, dil, pad, pad, L.PAD_ZERO, L.DTYPE_F32
            eff = (k - 1) * dil + 1
            if H + 2 * pad < eff or W + 2 * pad < eff:
                continue
            g.OH, g.OW = (H + 2 * pad - eff) // stride + 1, (W + 2 * pad - eff) // stride + 1
            want = (k == 3 and stride == 1 and dil <= 2 and pad % dil == 0 and pad <= 2 * dil and H % dil == 0 and W % dil == 0
                    and g.OH % dil == 0 and g.OW % dil == 0 and C % 32 == 0 and Kf % 4 == 0 and Kf >= 32)
            got = K.wino_chosen(g, 0)
            assert got == (m if want else 0), (N, H, W, C, Kf, k, stride, dil, pad, got, want)
            seen[want] += 1
            if want:
                Tn = T_.wino_tiles(N, g.OH, g.OW, dil, m)[4]
                al = lambda b: (b + 255) // 256 * 256
                assert int(lib.pnp_conv2d_fwd_workspace_bytes(ctypes.byref(g))) == al(4 * np_ * C * Kf) + al(4 * np_ * Tn * C) + al(4 * np_ * Tn * Kf)
        assert seen[True] >= 10 and seen[False] >= 100, seen
    finally:
        K.wino_mode(prev)
        K.wino_tile(prev_t)
        K.wino_x3(prev_x3)


def test_tail_split_plan_through_the_workspace_query(built):
    """PNP_WINO_TAILSPLIT=1 (off by default; read once by the library, hence a process of its own): a persistent GEMM launch of the
    Winograd route deals tiles / 8 to each XCD = whole rounds of 64 workgroups + R tail tiles, cuts every tail tile into the largest
    s in {8, 4, 2} pieces with s R <= 64, s | stages, >= 2 stages per piece, and asks for 8 R (s - 1) dense 128 x 128 partial tiles on top
    of the route's workspace; launches of <= 512 tiles or with a tile count that is no multiple of 8 are never split"""
    import subprocess
    import sys
    code = r'''
import ctypes, importlib, sys
sys.path.insert(0, %r)
K = importlib.import_module("medical-cross-modality-domain-adaptation_amd.kernels")
L = importlib.import_module("medical-cross-modality-domain-adaptation_amd._lib")
lib = L.load()
K.wino_mode(2); K.wino_tile(4); K.wino_x3(0)          # (the tail split belongs to the fp32-pipe GEMM)
tile = 128 * 128 * 4
for (N, H, C, Kf, pad, extra) in ((16, 32, 512, 512, "SAME", 8 * 16 * 3 * tile),     # 1152 tiles: 144 per XCD = 2 rounds + 16, 16 stages -> 4 pieces
                                  (16, 32, 256, 256, "SAME", 8 * 8 * 3 * tile),      # 576: 72 = 1 round + 8, 8 stages -> 4 pieces of 2
                                  (8, 32, 512, 512, "SAME", 8 * 8 * 7 * tile),       # 576, 16 stages -> 8 pieces of 2
                                  (16, 34, 512, 2560, "VALID", 8 * 16 * 3 * tile),   # 5760: 720 = 11 rounds + 16
                                  (16, 128, 128, 128, "SAME", 0),                    # 4608: 576 = 9 whole rounds
                                  (4, 32, 512, 512, "SAME", 0),                      # 288 tiles: not persistent
                                  (5, 30, 512, 544, "SAME", 0)):                     # 540 tiles: no multiple of 8
    g = K.conv_geom((N, H, H, C), (3, 3, C, Kf), 1, 1, pad)
    T = N * (-(-g.OH // 4)) * (-(-g.OW // 4))
    base = 36 * 4 * (C * Kf + T * C + T * Kf)
    got = int(lib.pnp_conv2d_fwd_workspace_bytes(ctypes.byref(g))) - base
    assert got == extra, ((N, H, C, Kf), got, extra)
print("PLAN OK")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PNP_WINO_TAILSPLIT="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PLAN OK" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
