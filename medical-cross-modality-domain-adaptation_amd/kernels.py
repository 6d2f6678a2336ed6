"""Raw (non-autograd) Python entry points over the C-ABI: torch tensors in, torch tensors out.

torch is used here only as the device-memory container and for the current HIP stream; every
arithmetic result comes from libpnp_hip.so.  Tensors are NHWC float32, filters HWIO — the
reference's layouts (layers.py).  Each function names the reference op it stands for.
"""
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import ConvGeom, PAD_SYMMETRIC, PAD_ZERO, check

_ws_cache = {}


# the current stream's raw handle straight from the C extension: torch.cuda.current_stream() walks is_available() -> os.environ.get on every
# call — 36 % of a joint step's host time at ~700 launches per step (cProfile, tools/experiments/r6_hostprof.py), and the bf16 step is
# host-bound
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream_ptr():
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _stream():
    return ctypes.c_void_p(_stream_ptr())


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.PnpError("pnp kernels need CUDA/HIP tensors (got a CPU tensor) — there is no CPU fallback")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.PnpError("pnp kernels need contiguous float32 tensors, got %s contiguous=%s" % (t.dtype, t.is_contiguous()))
    return ctypes.c_void_p(t.data_ptr())


def _ph(t):
    """device pointer of a contiguous bfloat16 tensor (the bf16-resident convolutions' operands)"""
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous():
        raise _lib.PnpError("expected a contiguous CUDA bfloat16 tensor, got %s %s contiguous=%s" % (t.device, t.dtype, t.is_contiguous()))
    return ctypes.c_void_p(t.data_ptr())


_ws_retired = []     # buffers replaced by a larger one while a recorded step exists: its hipGraph holds their ADDRESSES


def workspace(nbytes, device, slot="main"):
    """Grow-only scratch buffer per (device, slot, stream). All kernels are stream-ordered on the current stream.
    A buffer that has to grow is normally dropped (the allocator recycles it in stream order).  Not while a CapturedStep is alive or a
    stream is recording: a hipGraph replays the ADDRESS it recorded, so the replaced buffer is kept (`_ws_retired`, emptied when the last
    recorded step dies).  Round 6: the generator step's warm-up grew the filter-gradient side stream's buffer AFTER the discriminator
    step had been recorded with the smaller one — the bf16 joint step at B = 16 faulted on replay ("write access to a read-only page")."""
    key = (device.index if isinstance(device, torch.device) else str(device), slot, _stream_ptr())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and (PINNED[0] > 0 or torch.cuda.is_current_stream_capturing()):
            _ws_retired.append(buf)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def same_pad(in_size, k, stride, dil=1):
    """TF 'SAME': out = ceil(in/stride); pad_total = max((out-1)*stride + (k-1)*dil + 1 - in, 0); before = total//2."""
    out = -(-in_size // stride)
    eff = (k - 1) * dil + 1
    total = max((out - 1) * stride + eff - in_size, 0)
    return out, total // 2, total - total // 2


CONV_DTYPE = _lib.DTYPE_F32      # arithmetic type new geometries ask for (functional.set_conv_dtype)


def conv_geom(x_shape, w_shape, stride=1, dil=1, padding="SAME", dtype=None):
    """Geometry of layers.conv2d / dilate_conv2d (layers.py:64-74, 84-93)."""
    N, H, W, C = x_shape
    R, S, Cw, K = w_shape
    if Cw != C:
        raise ValueError("conv2d: input has %d channels, filter expects %d" % (C, Cw))
    g = ConvGeom()
    g.N, g.H, g.W, g.C, g.K, g.R, g.S = N, H, W, C, K, R, S
    g.stride, g.dil = stride, dil
    g.dtype = CONV_DTYPE if dtype is None else dtype
    if padding == "SAME":
        g.OH, g.pad_t, _ = same_pad(H, R, stride, dil)
        g.OW, g.pad_l, _ = same_pad(W, S, stride, dil)
        g.pad_mode = PAD_ZERO
    elif padding == "SYMMETRIC":
        # tf.pad(x, floor(k/2), 'SYMMETRIC') then VALID conv (layers.py:19-24)
        ph, pw = R // 2, S // 2
        g.pad_t, g.pad_l = ph, pw
        g.OH = (H + 2 * ph - ((R - 1) * dil + 1)) // stride + 1
        g.OW = (W + 2 * pw - ((S - 1) * dil + 1)) // stride + 1
        g.pad_mode = PAD_SYMMETRIC
    elif padding == "VALID":
        g.pad_t = g.pad_l = 0
        g.OH = (H - ((R - 1) * dil + 1)) // stride + 1
        g.OW = (W - ((S - 1) * dil + 1)) // stride + 1
        g.pad_mode = PAD_ZERO
    else:
        raise ValueError("unknown padding %r" % (padding,))
    return g


def conv2d_fwd(x, w, g, keep_prob=1.0, seed=0, stream_id=0, out=None, naive=False):
    lib = _lib.load()
    y = out if out is not None else torch.empty((g.N, g.OH, g.OW, g.K), dtype=torch.float32, device=x.device)
    if naive:
        check(lib.pnp_conv2d_fwd_naive(_p(x), _p(w), _p(y), ctypes.byref(g), _stream()), "pnp_conv2d_fwd_naive")
    else:
        nbytes = lib.pnp_conv2d_fwd_workspace_bytes(ctypes.byref(g))      # > 0 only for layers with too few output tiles
        ws = workspace(nbytes, x.device) if nbytes else None
        _wino_u(w, g, 0)
        check(lib.pnp_conv2d_fwd_ws(_p(x), _p(w), _p(y), ctypes.byref(g), float(keep_prob), int(seed), int(stream_id),
                                    ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, ws.numel() if ws is not None else 0,
                                    _stream()), "pnp_conv2d_fwd")
    return y


def conv_stats_parts(g):
    """number of BN-statistics partial rows the forward of `g` can leave behind (0: not available, use bn_stats)"""
    return int(_lib.load().pnp_conv2d_fwd_stats_ws_parts(ctypes.byref(g)))


def wino_mode(mode=-1):
    """route policy of the wide stride-1 3x3 convolutions (csrc/conv_wino.hip): 0 direct kernels only, 1 Winograd F(2x2, 3x3) where the
    cost model says it pays, 2 wherever the geometry allows; returns the previous mode (mode < 0: read only)"""
    return int(_lib.load().pnp_conv2d_wino_mode(int(mode)))


def wino_wgrad_mode(mode=-1):
    """the same switch for the filter gradient (pnp_conv2d_wgrad*); returns the previous mode"""
    return int(_lib.load().pnp_conv2d_wino_wgrad_mode(int(mode)))


def wino_tile(tile=-1):
    """largest output tile of the route: 2 = F(2x2, 3x3) only, 4 = F(4x4, 3x3) where its planner takes the layer (36 multiplications per
    4x4 outputs instead of 64; points (0, +-1, 1/2, -2)); returns the previous value (tile < 2: read only)"""
    return int(_lib.load().pnp_conv2d_wino_tile(int(tile)))


def wino_x3(mode=-1):
    """arithmetic of the route's forward / data-gradient GEMMs: 0 = fp32 matrix pipe, 1 = split-bf16 operands (three bf16 planes per value,
    six MFMA products, fp32 accumulation in 64-channel chunks: csrc/conv_wino_x3.hip) where it pays (reductions over >= 256 channels), 2 =
    wherever the shapes allow; returns the previous mode (mode < 0: read only)"""
    return int(_lib.load().pnp_conv2d_wino_x3(int(mode)))


def x3_direct(mode=-1):
    """the narrow 3x3 layers of the Winograd route (32 / 64 input channels, <= 128 filters) as direct split-bf16 convolutions
    (csrc/conv_x3_direct.hip): 0 off, 1 on; returns the previous mode (mode < 0: read only)"""
    return int(_lib.load().pnp_conv2d_x3_direct(int(mode)))


def wino_chosen(g, kind=0):
    """pnp_conv2d_fwd* (kind 0) / pnp_conv2d_dgrad* (kind 1) / pnp_conv2d_wgrad* (kind 2) of this layer (g = the forward geometry): 0 = the
    direct kernels, else the output tile edge of the Winograd route (2 or 4) — truthy exactly when the layer is on the route"""
    return int(_lib.load().pnp_conv2d_wino_chosen(ctypes.byref(g), int(kind)))


# ---- transformed-filter cache of the Winograd route (pnp_conv2d_wino_filter_bind) ------------------------------------------------------
# U = G g G^T changes only when the filter does.  Filters OWNED BY A VariableStore (variables.py marks their tensors `_pnp_var`) get one
# buffer per pass, lent to the library; the library skips wino_filter_kernel while the entry is valid.  Validity: the library drops an
# entry when weights_changed() names its address range (optimisers, clip, host-side loads); here an entry is tied to the tensor OBJECT it
# was made for (weakref: a new tensor at a recycled address starts over) and to its torch version counter (an in-place torch op on the
# filter or on the arena it is a view of).  Ad-hoc filters (tests, tools) are never cached.  PNP_WINOGRAD_UCACHE=0: off.
U_CACHE = os.environ.get("PNP_WINOGRAD_UCACHE", "1") != "0"
_u_cache = {}


def _u_drop(key, ref=None):
    """withdraw the binding of `key` (its filter tensor died, or an ad-hoc filter now sits on its address)"""
    ent = _u_cache.get(key)
    if ent is None or (ref is not None and ent[0] is not ref):
        return
    del _u_cache[key]
    if ent[1] is not None:
        try:
            _lib.load().pnp_conv2d_wino_filter_bind(ctypes.c_void_p(key[0]), int(key[1]), None, 0)
        except Exception:          # interpreter shutdown: the library may be gone already
            pass


def _wino_u(w, g, kind):
    if not U_CACHE:
        return
    key = (w.data_ptr(), kind)
    ent = _u_cache.get(key)
    if not getattr(w, "_pnp_var", False):
        if ent is not None:                  # an ad-hoc filter on the address of a cached one: the library must not serve the old U
            _u_drop(key)
        return
    if ent is not None and ent[0]() is w:
        if ent[2] != w._version:             # written by a torch op since the entry was filled
            if ent[1] is not None:
                _lib.load().pnp_weights_changed(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(w.data_ptr() + 4 * w.numel()))
            ent[2] = w._version
        if ent[1] is not None or not (g.R == 3 and g.S == 3 and wino_chosen(g, kind)):
            return
        # no buffer was bound when this filter was first seen (a small-batch forward the planner declined, a policy change): the route
        # takes it NOW — bind one instead of leaving the cache silently off for the lifetime of the tensor
    lib = _lib.load()
    if ent is not None and not (ent[0]() is w and ent[1] is None):
        _u_drop(key)
    U = None
    if g.R == 3 and g.S == 3 and wino_chosen(g, kind):          # (decided once per filter and pass: a later policy change runs un-cached)
        nbytes = int(lib.pnp_conv2d_wino_filter_bytes(int(w.shape[2]), int(w.shape[3])))
        if nbytes:
            U = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
            check(lib.pnp_conv2d_wino_filter_bind(ctypes.c_void_p(w.data_ptr()), int(kind), ctypes.c_void_p(U.data_ptr()), nbytes),
                  "pnp_conv2d_wino_filter_bind")
    # the binding goes when the tensor object does (CPython frees it, and with it the arena it may keep alive, deterministically): nothing
    # allocated later on the same address can meet a valid entry
    _u_cache[key] = [weakref.ref(w, lambda r, k=key: _u_drop(k, r)), U, w._version]


def wino_u_cache_clear():
    """withdraw every transformed-filter buffer (tests; a store that goes away)"""
    if _u_cache:
        _lib.load().pnp_conv2d_wino_filter_bind(None, 0, None, 0)
        _u_cache.clear()


def memory_report():
    """bytes of device memory the host side of the library holds on to between steps (nothing reported it):
    grow-only workspaces per (slot, stream), workspaces retired under a recorded step, transformed filters of the Winograd route,
    bf16 filter shadows"""
    nb = lambda t: 0 if t is None else int(t.numel()) * int(t.element_size())
    rep = {"workspaces": sum(nb(b) for b in _ws_cache.values()), "workspace_buffers": len(_ws_cache),
           "workspaces_retired": sum(nb(b) for b in _ws_retired), "recorded_steps_alive": int(PINNED[0]),
           "winograd_filters": sum(nb(e[1]) for e in _u_cache.values()), "winograd_filter_entries": len(_u_cache),
           "bf16_filter_shadows": sum(nb(e[1]) + nb(e[2]) for e in _filter_cache.values()), "bf16_filter_entries": len(_filter_cache)}
    rep["total"] = rep["workspaces"] + rep["workspaces_retired"] + rep["winograd_filters"] + rep["bf16_filter_shadows"]
    return rep


def wino_u_cache_stats(reset=False):
    """(filter transforms skipped, transforms run into a cache entry) since the last reset"""
    h, f = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.load().pnp_conv2d_wino_filter_stats(ctypes.byref(h), ctypes.byref(f), 1 if reset else 0)
    return int(h.value), int(f.value)


def _fwd_ws(g, device):
    nbytes = _lib.load().pnp_conv2d_fwd_workspace_bytes(ctypes.byref(g))
    if not nbytes:
        return None, 0
    ws = workspace(nbytes, device)
    return ctypes.c_void_p(ws.data_ptr()), ws.numel()


def conv2d_fwd_stats(x, w, g, shift, keep_prob=1.0, seed=0, stream_id=0):
    """forward conv (+dropout) that also writes the batch-norm statistics partials of its output -> (y, parts [nparts, 2, K])"""
    lib = _lib.load()
    nparts = conv_stats_parts(g)
    y = torch.empty((g.N, g.OH, g.OW, g.K), dtype=torch.float32, device=x.device)
    parts = workspace(nparts * 2 * g.K * 4, x.device, slot="stats")
    wsp, wsn = _fwd_ws(g, x.device)
    _wino_u(w, g, 0)
    check(lib.pnp_conv2d_fwd_stats_ws(_p(x), _p(w), _p(y), ctypes.byref(g), float(keep_prob), int(seed), int(stream_id), _p(shift),
                                      ctypes.c_void_p(parts.data_ptr()), parts.numel(), wsp, wsn, _stream()), "pnp_conv2d_fwd_stats_ws")
    return y, (parts, nparts)


def bn_stats_finish(parts_n, shift, P, mm=None, mv=None, decay=0.9):
    """(mean, biased var) from conv2d_fwd_stats' partials; with mm / mv also the moving-average update"""
    parts, nparts = parts_n
    C = shift.numel()
    mean = torch.empty(C, dtype=torch.float32, device=shift.device)
    var = torch.empty(C, dtype=torch.float32, device=shift.device)
    check(_lib.load().pnp_bn_stats_finish(ctypes.c_void_p(parts.data_ptr()), int(nparts), _p(shift), _p(mean), _p(var), _p(mm), _p(mv), int(P), C,
                                          float(decay), _stream()), "pnp_bn_stats_finish")
    return mean, var


def bn_fold(gamma, beta, mean, var, eps=1e-3):
    """inference-mode BN as per-channel (scale, shift)"""
    lib = _lib.load()
    C = gamma.numel()
    ss = torch.empty((2, C), dtype=torch.float32, device=gamma.device)
    check(lib.pnp_bn_fold(_p(gamma), _p(beta), _p(mean), _p(var), _p(ss[0]), _p(ss[1]), C, float(eps), _stream()), "pnp_bn_fold")
    return ss


def conv2d_fwd_bn(x, w, g, scale_shift, shortcut=None, alpha=0.2, keep_prob=1.0, seed=0, stream_id=0):
    """conv -> dropout -> inference BN -> (+ shortcut) -> leaky-ReLU in one launch (pnp_conv2d_fwd_bn)"""
    lib = _lib.load()
    y = torch.empty((g.N, g.OH, g.OW, g.K), dtype=torch.float32, device=x.device)
    cs = shortcut.shape[-1] if shortcut is not None else 0
    wsp, wsn = _fwd_ws(g, x.device)
    _wino_u(w, g, 0)
    check(lib.pnp_conv2d_fwd_bn_ws(_p(x), _p(w), _p(y), ctypes.byref(g), float(keep_prob), int(seed), int(stream_id), _p(scale_shift[0]),
                                   _p(scale_shift[1]), _p(shortcut), cs, float(alpha), wsp, wsn, _stream()), "pnp_conv2d_fwd_bn_ws")
    return y


# ---- bf16-resident convolutions (BASELINE configs[4]; csrc/conv_bf16r.hip) ------------------------------------------------------------
def cast_bf16(x):
    """bf16 copy (round-to-nearest-even) of a float32 tensor: pnp_cast_bf16 — for tensors whose producer has no bf16 output"""
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(_lib.load().pnp_cast_bf16(_p(x), _ph(y), x.numel(), _stream()), "pnp_cast_bf16")
    return y


def filter_bf16(w, want_io=True, want_oi=True):
    """bf16 shadows of the fp32 master filter [R,S,C,K]: (w_io [R*S,C,K], w_oi [R*S,K,C])"""
    R, S, C, Kc = w.shape
    w_io = torch.empty((R * S, C, Kc), dtype=torch.bfloat16, device=w.device) if want_io else None
    w_oi = torch.empty((R * S, Kc, C), dtype=torch.bfloat16, device=w.device) if want_oi else None
    check(_lib.load().pnp_filter_bf16(_p(w), _ph(w_io), _ph(w_oi), R, S, C, Kc, _stream()), "pnp_filter_bf16")
    return w_io, w_oi


# bf16 copies ("shadows") of tensors the resident convolutions read.  An activation's shadow is written by its producer (yh / dxh side
# outputs) and travels as an attribute of the tensor OBJECT; a tensor that arrives without one (re-wrapped by a view, produced by a
# kernel without a bf16 output) is cast on demand.  Filter shadows are cached per filter and weight epoch: every kernel that writes
# weights (optimisers, clip) and every host-side load bumps the epoch.
_WEIGHT_EPOCH = [0]
_CHANGES = []              # (epoch, ((lo, hi), ...)) of the range-limited weight writes since _CHANGE_FLOOR, oldest first
_CHANGE_FLOOR = [0]        # a write of unknown extent happened at this epoch: everything older is stale
_filter_cache = {}
CAST_COUNT = [0]          # on-demand casts since the last reset (tests / profiling: how many producers still lack a bf16 output)


PINNED = [0]              # live CapturedStep objects: their hipGraphs hold the ADDRESSES of filter shadows — nothing may be freed under them


def weights_changed(ranges=None):
    """weights were (or are queued to be) written.  ranges: [(first byte address, end address), ...] of what changed — the library drops
    the transformed filters (Winograd route) of exactly those; None: everything.  The bf16 shadows follow the same ranges (round 6: one
    global epoch re-built the shadows of every FROZEN filter after every optimiser step — 102 filter_bf16_kernel launches per joint step)."""
    _WEIGHT_EPOCH[0] += 1
    if ranges is None or len(_CHANGES) >= 64:
        _CHANGES.clear()
        _CHANGE_FLOOR[0] = _WEIGHT_EPOCH[0]         # everything filled before this epoch is stale
    else:
        _CHANGES.append((_WEIGHT_EPOCH[0], tuple((int(lo), int(hi)) for lo, hi in ranges)))
    if len(_filter_cache) > 1024 and not PINNED[0]:      # shadows of filters of stores that no longer exist (test suites): start over
        _filter_cache.clear()
    if _u_cache:
        lib = _lib.load()
        if ranges is None and len(_u_cache) > 4096 and not PINNED[0]:    # buffers of stores that no longer exist (test suites): start over
            wino_u_cache_clear()
        elif ranges is None:
            lib.pnp_weights_changed(None, None)
        else:
            for lo, hi in ranges:
                lib.pnp_weights_changed(ctypes.c_void_p(lo), ctypes.c_void_p(hi))


class HalfOnly(object):
    """A tensor that exists ONLY as its bf16 copy (the gradient w.r.t. a convolution's accumulator when both of its consumers — data
    and filter gradient — are resident kernels: the float32 copy would be written and never read)."""
    __slots__ = ("h",)

    def __init__(self, h):
        self.h = h

    @property
    def shape(self):
        return self.h.shape


def bf16_of(t):
    if isinstance(t, HalfOnly):
        return t.h
    h = getattr(t, "_pnp_h", None)
    # (while a hipGraph is being recorded a cached shadow would leave the cast OUT of the graph: a replay on new input data would then
    # read the warm-up's copy — cast inside the recording instead)
    if h is not None and h[1] == t._version and h[0].shape == t.shape and not torch.cuda.is_current_stream_capturing():
        return h[0]
    CAST_COUNT[0] += 1
    hh = cast_bf16(t)
    set_bf16(t, hh)
    return hh


def set_bf16(t, h):
    if h is not None:
        t._pnp_h = (h, t._version)
    return t


def _changed_since(epoch, lo, hi):
    """has a reported weight write touched [lo, hi) after `epoch`"""
    if epoch < _CHANGE_FLOOR[0]:
        return True
    for ep, rs in reversed(_CHANGES):
        if ep <= epoch:
            break
        for a, b in rs:
            if a < hi and lo < b:
                return True
    return False


def filter_shadows(w):
    """(w_io, w_oi) of an fp32 filter [R,S,C,K], refreshed when a reported weight write has touched the filter since they were made"""
    key = (w.data_ptr(), tuple(w.shape))
    ent = _filter_cache.get(key)
    # valid: same weight epoch (kernels that write weights), same torch version counter (an in-place torch op on the filter or the arena
    # it is a view of), same owner object (a new tensor on a recycled address is another filter)
    owner = w._base if w._base is not None else w
    if ent is None or ent[3] != w._version or ent[4]() is not owner or (ent[0] != _WEIGHT_EPOCH[0] and _changed_since(ent[0], key[0], key[0] + 4 * w.numel())):
        if ent is None:
            w_io, w_oi = filter_bf16(w)
        else:                   # refresh in place: same buffers, no allocation in the steady state
            w_io, w_oi = ent[1], ent[2]
            R, S, C, Kc = w.shape
            check(_lib.load().pnp_filter_bf16(_p(w), _ph(w_io), _ph(w_oi), R, S, C, Kc, _stream()), "pnp_filter_bf16")
        ent = (_WEIGHT_EPOCH[0], w_io, w_oi, w._version, weakref.ref(owner))
        _filter_cache[key] = ent
    return ent[1], ent[2]


_served_cache = {}


def bf16r(g, kind):
    """does the bf16-RESIDENT kernel family serve (geometry, kind) — only asked for geometries whose dtype is PNP_DTYPE_BF16"""
    if g.dtype != _lib.DTYPE_BF16:
        return False
    key = (g.key(), kind)
    v = _served_cache.get(key)
    if v is None:
        v = _served_cache[key] = bf16r_served(g, kind)
    return v


def bf16r_served(g, kind):
    """kind 0 forward / 1 data gradient: do the resident kernels serve this geometry"""
    return bool(_lib.load().pnp_conv2d_bf16r_served(ctypes.byref(g), int(kind)))


def conv2d_fwd_bf16r(xh, w_oi, g, keep_prob=1.0, seed=0, stream_id=0, want_h=False, stat_shift=None, want_stats=False, bn=None):
    """resident forward.  Returns (y, yh or None, parts or None).  want_stats: BN statistics partials (-> bn_stats_finish);
    bn = (scale_shift [2,K], shortcut or None, alpha): fused inference-mode BN + shortcut + leaky-ReLU"""
    lib = _lib.load()
    y = torch.empty((g.N, g.OH, g.OW, g.K), dtype=torch.float32, device=xh.device)
    yh = torch.empty((g.N, g.OH, g.OW, g.K), dtype=torch.bfloat16, device=xh.device) if want_h else None
    parts, nparts, pbytes = None, 0, 0
    if want_stats:
        nparts = int(lib.pnp_conv2d_fwd_bf16r_stats_parts(ctypes.byref(g)))
        parts = workspace(nparts * 2 * g.K * 4, xh.device, slot="stats")
        pbytes = parts.numel()
    ss, sc, alpha = (bn[0], bn[1], bn[2]) if bn is not None else (None, None, -1.0)
    check(lib.pnp_conv2d_fwd_bf16r(_ph(xh), _ph(w_oi), _p(y), _ph(yh), ctypes.byref(g), float(keep_prob), int(seed), int(stream_id),
                                   _p(stat_shift), ctypes.c_void_p(parts.data_ptr()) if parts is not None else None, pbytes,
                                   _p(ss[0]) if ss is not None else None, _p(ss[1]) if ss is not None else None, _p(sc),
                                   sc.shape[-1] if sc is not None else 0, float(alpha), _stream()), "pnp_conv2d_fwd_bf16r")
    return y, yh, ((parts, nparts) if want_stats else None)


def conv2d_dgrad_bf16r(dyh, w_io, g, residual=None, want_h=False):
    lib = _lib.load()
    dx = torch.empty((g.N, g.H, g.W, g.C), dtype=torch.float32, device=dyh.device)
    dxh = torch.empty((g.N, g.H, g.W, g.C), dtype=torch.bfloat16, device=dyh.device) if want_h else None
    check(lib.pnp_conv2d_dgrad_bf16r(_ph(dyh), _ph(w_io), _p(residual), _p(dx), _ph(dxh), ctypes.byref(g), _stream()),
          "pnp_conv2d_dgrad_bf16r")
    return dx, dxh


def conv2d_wgrad_bf16r(xh, dyh, g, into=None):
    """filter gradient from the bf16 copies of x and dy; `into`: dw is ADDED to it (a slot of the gradient arena)"""
    lib = _lib.load()
    nbytes = lib.pnp_conv2d_wgrad_bf16r_workspace_bytes(ctypes.byref(g))
    ws = workspace(nbytes, xh.device)
    dw = into if into is not None else torch.empty((g.R, g.S, g.C, g.K), dtype=torch.float32, device=xh.device)
    if tuple(dw.shape) != (g.R, g.S, g.C, g.K):
        raise ValueError("conv2d_wgrad_bf16r: `into` %s is not the filter shape %s" % (tuple(dw.shape), (g.R, g.S, g.C, g.K)))
    check(lib.pnp_conv2d_wgrad_bf16r(_ph(xh), _ph(dyh), _p(dw), 1 if into is not None else 0, ctypes.byref(g),
                                     ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()), "pnp_conv2d_wgrad_bf16r")
    return dw


def conv2d_dgrad(dy, w, g, residual=None):
    """gradient w.r.t. the conv input; with `residual` ([N,H,W,C]) dx = gradient + residual in the same launch (pnp_conv2d_dgrad_add)"""
    lib = _lib.load()
    dx = torch.empty((g.N, g.H, g.W, g.C), dtype=torch.float32, device=dy.device)
    nbytes = lib.pnp_conv2d_dgrad_workspace_bytes(ctypes.byref(g))
    ws = workspace(nbytes, dy.device)
    _wino_u(w, g, 1)
    if residual is not None:
        if tuple(residual.shape) != tuple(dx.shape):
            raise ValueError("conv2d_dgrad: residual %s does not match dx %s" % (tuple(residual.shape), tuple(dx.shape)))
        check(lib.pnp_conv2d_dgrad_add(_p(dy), _p(w), _p(residual), _p(dx), ctypes.byref(g), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                       _stream()), "pnp_conv2d_dgrad_add")
        return dx
    check(lib.pnp_conv2d_dgrad(_p(dy), _p(w), _p(dx), ctypes.byref(g), ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
          "pnp_conv2d_dgrad")
    return dx


def conv2d_wgrad(x, dy, g, into=None):
    """filter gradient; `into` ([R,S,C,K], e.g. the variable's slot of the gradient arena): dw is ADDED to it (pnp_conv2d_wgrad_acc)
    and `into` is returned"""
    lib = _lib.load()
    nbytes = lib.pnp_conv2d_wgrad_workspace_bytes(ctypes.byref(g))
    ws = workspace(nbytes, x.device)
    if into is not None:
        if tuple(into.shape) != (g.R, g.S, g.C, g.K):
            raise ValueError("conv2d_wgrad: `into` %s is not the filter shape %s" % (tuple(into.shape), (g.R, g.S, g.C, g.K)))
        check(lib.pnp_conv2d_wgrad_acc(_p(x), _p(dy), _p(into), ctypes.byref(g), ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
              "pnp_conv2d_wgrad_acc")
        return into
    dw = torch.empty((g.R, g.S, g.C, g.K), dtype=torch.float32, device=x.device)
    check(lib.pnp_conv2d_wgrad(_p(x), _p(dy), _p(dw), ctypes.byref(g), ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
          "pnp_conv2d_wgrad")
    return dw


def dropout(x, keep_prob, seed, stream_id, want_h=False):
    """want_h: also the bf16 copy of the result (attached as its shadow)"""
    lib = _lib.load()
    y = torch.empty_like(x)
    yh = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_h else None
    check(lib.pnp_dropout_h(_p(x), _p(y), _ph(yh), x.numel(), float(keep_prob), int(seed), int(stream_id), _stream()), "pnp_dropout")
    return set_bf16(y, yh)


def bn_stats(x2d_like):
    """per-channel mean / biased variance over all leading dims (layers.py:100, training path)."""
    lib = _lib.load()
    C = x2d_like.shape[-1]
    P = x2d_like.numel() // C
    mean = torch.empty(C, dtype=torch.float32, device=x2d_like.device)
    var = torch.empty(C, dtype=torch.float32, device=x2d_like.device)
    ws = workspace(lib.pnp_bn_workspace_bytes(P, C), x2d_like.device)
    check(lib.pnp_bn_stats(_p(x2d_like), _p(mean), _p(var), P, C, ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
          "pnp_bn_stats")
    return mean, var


def bn_stats_update(x2d_like, mm, mv, decay=0.9):
    """bn_stats + bn_update_moving in one pass (per-replica statistics)"""
    lib = _lib.load()
    C = x2d_like.shape[-1]
    P = x2d_like.numel() // C
    mean = torch.empty(C, dtype=torch.float32, device=x2d_like.device)
    var = torch.empty(C, dtype=torch.float32, device=x2d_like.device)
    ws = workspace(lib.pnp_bn_workspace_bytes(P, C), x2d_like.device)
    check(lib.pnp_bn_stats_update(_p(x2d_like), _p(mean), _p(var), _p(mm), _p(mv), P, C, float(decay), ctypes.c_void_p(ws.data_ptr()),
                                  ws.numel(), _stream()), "pnp_bn_stats_update")
    return mean, var


def bn_update_moving(mm, mv, mean, var, P, decay=0.9):
    lib = _lib.load()
    check(lib.pnp_bn_update_moving(_p(mm), _p(mv), _p(mean), _p(var), P, mm.numel(), float(decay), _stream()),
          "pnp_bn_update_moving")


def bn_apply(x, mean, var, gamma, beta, shortcut=None, eps=1e-3, alpha=0.2, want_h=False):
    lib = _lib.load()
    C = x.shape[-1]
    P = x.numel() // C
    y = torch.empty_like(x)
    yh = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_h else None
    Cs = shortcut.shape[-1] if shortcut is not None else 0
    check(lib.pnp_bn_apply_h(_p(x), _p(mean), _p(var), _p(gamma), _p(beta), _p(shortcut), Cs, _p(y), _ph(yh), P, C, float(eps),
                             float(alpha), _stream()), "pnp_bn_apply")
    return set_bf16(y, yh)


def bn_bwd(dout, out, x, mean, var, gamma, shortcut_channels=0, eps=1e-3, alpha=0.2, training=True, keep_prob=1.0, seed=0,
           stream_id=0, into=None, beta=None, want_h=False, only_h=False):
    """`into` = (dgamma_slot, dbeta_slot): the parameter gradients are also ADDED to these [C] buffers (pnp_bn_bwd_acc).
    out=None (with `beta`, units without a shortcut): the kernels recompute the activation's sign from x instead of reading `out`"""
    lib = _lib.load()
    C = x.shape[-1]
    P = x.numel() // C
    dx = torch.empty_like(x) if not only_h else None
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    dsc = None
    if shortcut_channels:
        dsc = torch.empty(x.shape[:-1] + (shortcut_channels,), dtype=torch.float32, device=x.device)
    ws = workspace(lib.pnp_bn_workspace_bytes(P, C), x.device)
    dxh = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if (want_h or only_h) else None
    a0, a1 = (into[0], into[1]) if into is not None else (None, None)
    check(lib.pnp_bn_bwd_acc_h(_p(dout), _p(out), _p(x), _p(mean), _p(var), _p(gamma), _p(beta), _p(dx), _ph(dxh), _p(dgamma), _p(dbeta), _p(a0),
                               _p(a1), _p(dsc), shortcut_channels, P, C, float(eps), float(alpha), 1 if training else 0,
                               float(keep_prob), int(seed), int(stream_id), ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
          "pnp_bn_bwd_acc" if into is not None else "pnp_bn_bwd")
    return (HalfOnly(dxh) if only_h else set_bf16(dx, dxh)), dgamma, dbeta, dsc


def bn_bwd_reduce(dout, out, x, mean, var, eps=1e-3, alpha=0.2, gamma=None, beta=None):
    """local sums (dgamma = sum dz*xhat, dbeta = sum dz) — first half of bn_bwd, for SyncBN"""
    lib = _lib.load()
    C = x.shape[-1]
    P = x.numel() // C
    sums = torch.empty((2, C), dtype=torch.float32, device=x.device)      # [dgamma, dbeta]
    ws = workspace(lib.pnp_bn_workspace_bytes(P, C), x.device)
    check(lib.pnp_bn_bwd_reduce(_p(dout), _p(out), _p(x), _p(mean), _p(var), _p(gamma), _p(beta), _p(sums[0]), _p(sums[1]), P, C, float(eps), float(alpha),
                                ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()), "pnp_bn_bwd_reduce")
    return sums


def bn_bwd_apply(dout, out, x, mean, var, gamma, sums, P_norm, shortcut_channels=0, eps=1e-3, alpha=0.2, training=True, keep_prob=1.0,
                 seed=0, stream_id=0, beta=None, want_h=False, only_h=False):
    """second half of bn_bwd with the (possibly all-reduced) sums and the row count behind them"""
    lib = _lib.load()
    C = x.shape[-1]
    P = x.numel() // C
    dx = torch.empty_like(x) if not only_h else None
    dsc = None
    if shortcut_channels:
        dsc = torch.empty(x.shape[:-1] + (shortcut_channels,), dtype=torch.float32, device=x.device)
    dg, db = (sums[0], sums[1]) if sums is not None else (None, None)      # not read in inference mode
    dxh = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if (want_h or only_h) else None
    check(lib.pnp_bn_bwd_apply_h(_p(dout), _p(out), _p(x), _p(mean), _p(var), _p(gamma), _p(beta), _p(dg), _p(db), _p(dx), _ph(dxh), _p(dsc),
                                 shortcut_channels, P, int(P_norm), C, float(eps), float(alpha), 1 if training else 0, float(keep_prob),
                                 int(seed), int(stream_id), _stream()), "pnp_bn_bwd_apply")
    return (HalfOnly(dxh) if only_h else set_bf16(dx, dxh)), dsc


def maxpool2_fwd(x):
    lib = _lib.load()
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    check(lib.pnp_maxpool2_fwd(_p(x), _p(y), N, H, W, C, _stream()), "pnp_maxpool2_fwd")
    return y


def maxpool2_bwd(x, dy):
    lib = _lib.load()
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    check(lib.pnp_maxpool2_bwd(_p(x), _p(dy), _p(dx), N, H, W, C, _stream()), "pnp_maxpool2_bwd")
    return dx


def ps_fwd(x, r, nc):
    lib = _lib.load()
    N, A, B, Cin = x.shape
    if Cin != nc * r * r:
        raise ValueError("PS: %d channels != n_channel*r*r = %d" % (Cin, nc * r * r))
    y = torch.empty((N, A * r, B * r, nc), dtype=torch.float32, device=x.device)
    check(lib.pnp_ps_fwd(_p(x), _p(y), N, A, B, r, nc, _stream()), "pnp_ps_fwd")
    return y


def ps_bwd(dy, r, nc):
    lib = _lib.load()
    N, Ar, Br, _ = dy.shape
    A, B = Ar // r, Br // r
    dx = torch.empty((N, A, B, nc * r * r), dtype=torch.float32, device=dy.device)
    check(lib.pnp_ps_bwd(_p(dy), _p(dx), N, A, B, r, nc, _stream()), "pnp_ps_bwd")
    return dx


def sympad_fwd(x, p):
    """tf.pad(x, p, 'SYMMETRIC') in H and W (layers.py:23,72,91)"""
    lib = _lib.load()
    N, H, W, C = x.shape
    xp = torch.empty((N, H + 2 * p, W + 2 * p, C), dtype=torch.float32, device=x.device)
    check(lib.pnp_sympad_fwd(_p(x), _p(xp), N, H, W, C, p, _stream()), "pnp_sympad_fwd")
    return xp


def sympad_bwd(dxp, p):
    lib = _lib.load()
    N, Hp, Wp, C = dxp.shape
    dx = torch.empty((N, Hp - 2 * p, Wp - 2 * p, C), dtype=torch.float32, device=dxp.device)
    check(lib.pnp_sympad_bwd(_p(dxp), _p(dx), N, Hp - 2 * p, Wp - 2 * p, C, p, _stream()), "pnp_sympad_bwd")
    return dx


def seg_loss_fwd(logits, y, miu_cross=1.0, miu_dice=1.0):
    """returns (out[3] = total, xent, dice ; workspace tensor to hand to seg_loss_bwd)"""
    lib = _lib.load()
    ncls = logits.shape[-1]
    P = logits.numel() // ncls
    out = torch.empty(3, dtype=torch.float32, device=logits.device)
    ws = torch.empty(lib.pnp_seg_loss_workspace_bytes(P, ncls), dtype=torch.uint8, device=logits.device)
    check(lib.pnp_seg_loss_fwd(_p(logits), _p(y), _p(out), P, ncls, float(miu_cross), float(miu_dice),
                               ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()), "pnp_seg_loss_fwd")
    return out, ws


def seg_loss_bwd(logits, y, ws, miu_cross=1.0, miu_dice=1.0, gscale=1.0, P_norm=None):
    """P_norm: pixel count behind the sums in `ws` when they were all-reduced over the replicas (default: this tensor's)"""
    lib = _lib.load()
    ncls = logits.shape[-1]
    P = logits.numel() // ncls
    dl = torch.empty_like(logits)
    check(lib.pnp_seg_loss_bwd_norm(_p(logits), _p(y), _p(dl), P, int(P_norm or P), ncls, float(miu_cross), float(miu_dice),
                                    float(gscale), ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()), "pnp_seg_loss_bwd_norm")
    return dl


def seg_loss_sums(ws):
    """the 32 double sums at the head of a seg_loss workspace (per class: n_i, sum p*y, sum p*p, cross-entropy terms), as a view"""
    return ws[:256].view(torch.float64)


def softmax_argmax(logits, want_prob=True):
    """layers.pixel_wise_softmax_2 + tf.argmax(.,3) (layers.py:134-138; source_segmenter.py:80-81)"""
    lib = _lib.load()
    ncls = logits.shape[-1]
    P = logits.numel() // ncls
    prob = torch.empty_like(logits) if want_prob else None
    label = torch.empty(logits.shape[:-1], dtype=torch.int64, device=logits.device)
    check(lib.pnp_softmax_argmax(_p(logits), _p(prob), ctypes.c_void_p(label.data_ptr()), P, ncls, _stream()), "pnp_softmax_argmax")
    return prob, label


def dice_eval(label, y):
    """lib._dice_eval (lib.py:96-110): returns tensor [1+ncls] = (mean, per-class...)"""
    lib = _lib.load()
    ncls = y.shape[-1]
    P = y.numel() // ncls
    out = torch.empty(1 + ncls, dtype=torch.float32, device=y.device)
    ws = workspace(1024 * 24 * 4, y.device)
    check(lib.pnp_dice_eval(ctypes.c_void_p(label.data_ptr()), _p(y), _p(out), P, ncls, ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                            _stream()), "pnp_dice_eval")
    return out


def label_decomp(label, ncls):
    """lib._label_decomp (lib.py:75-92) on the device: integer-valued float label map [...] -> one-hot float32 [..., ncls]"""
    lab = label if label.dtype == torch.float32 else label.to(torch.float32)
    lab = lab.contiguous()
    out = torch.empty(tuple(lab.shape) + (ncls,), dtype=torch.float32, device=lab.device)
    check(_lib.load().pnp_label_decomp(_p(lab), _p(out), lab.numel(), int(ncls), _stream()), "pnp_label_decomp")
    return out


def confusion_matrix(y, pred=None, want_compact=True):
    """(compact_y = argmax of the one-hot labels, confusion matrix [ncls, ncls] int64 rows = truth) — source_segmenter.py:83-85"""
    ncls = y.shape[-1]
    P = y.numel() // ncls
    cy = torch.empty(y.shape[:-1], dtype=torch.int64, device=y.device) if want_compact else None
    cm = torch.empty((ncls, ncls), dtype=torch.int64, device=y.device) if pred is not None else None
    if pred is not None and (pred.dtype != torch.int64 or not pred.is_contiguous()):
        raise _lib.PnpError("confusion_matrix: pred must be a contiguous int64 label map")
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    check(_lib.load().pnp_confusion_matrix(_p(y), vp(pred), vp(cy), vp(cm), P, ncls, _stream()), "pnp_confusion_matrix")
    return cy, cm


def bn_moments(mean, var):
    C = mean.numel()
    mom = torch.empty(2 * C, dtype=torch.float64, device=mean.device)
    check(_lib.load().pnp_bn_moments(_p(mean), _p(var), ctypes.c_void_p(mom.data_ptr()), C, _stream()), "pnp_bn_moments")
    return mom


def bn_from_moments(mom, world):
    C = mom.numel() // 2
    mean = torch.empty(C, dtype=torch.float32, device=mom.device)
    var = torch.empty(C, dtype=torch.float32, device=mom.device)
    check(_lib.load().pnp_bn_from_moments(ctypes.c_void_p(mom.data_ptr()), int(world), _p(mean), _p(var), C, _stream()), "pnp_bn_from_moments")
    return mean, var


def _bumps_weights(fn):
    """a kernel that writes the weight arena: the filters' bf16 shadows and transformed filters are stale afterwards.
    ranges=[(lo, hi), ...]: byte-address ranges of the weights this call writes (an optimiser over a var_list: the chunks its mask selects)"""
    def wrapped(*a, **k):
        weights_changed(k.pop("ranges", None))
        return fn(*a, **k)
    wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
    return wrapped


def _u8p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@_bumps_weights
def adam_step(w, g, m, v, chunk_l2, chunk_mask, lr, beta1, beta2, eps, t):
    check(_lib.load().pnp_adam_step(_p(w), _p(g), _p(m), _p(v), w.numel(), _p(chunk_l2), _u8p(chunk_mask), float(lr), float(beta1),
                                    float(beta2), float(eps), int(t), _stream()), "pnp_adam_step")


@_bumps_weights
def rmsprop_step(w, g, ms, chunk_l2, chunk_mask, lr, decay=0.9, eps=1e-10):
    check(_lib.load().pnp_rmsprop_step(_p(w), _p(g), _p(ms), w.numel(), _p(chunk_l2), _u8p(chunk_mask), float(lr), float(decay),
                                       float(eps), _stream()), "pnp_rmsprop_step")


@_bumps_weights
def momentum_step(w, g, acc, chunk_l2, chunk_mask, lr, momentum):
    check(_lib.load().pnp_momentum_step(_p(w), _p(g), _p(acc), w.numel(), _p(chunk_l2), _u8p(chunk_mask), float(lr),
                                        float(momentum), _stream()), "pnp_momentum_step")


@_bumps_weights
def clip(w, chunk_mask, lo, hi):
    check(_lib.load().pnp_clip(_p(w), w.numel(), _u8p(chunk_mask), float(lo), float(hi), _stream()), "pnp_clip")


def l2_loss(w, chunk_l2):
    lib = _lib.load()
    out = torch.empty(1, dtype=torch.float32, device=w.device)
    ws = workspace(lib.pnp_reduce_workspace_bytes(w.numel()), w.device)
    check(lib.pnp_l2_loss(_p(w), w.numel(), _p(chunk_l2), _p(out), ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
          "pnp_l2_loss")
    return out


def axpby(x, y, a, b):
    check(_lib.load().pnp_axpby(_p(x), _p(y), x.numel(), float(a), float(b), _stream()), "pnp_axpby")
    return y


def add(x, y):
    """x + y into a new tensor (pnp_add)"""
    out = torch.empty_like(x)
    check(_lib.load().pnp_add(_p(x), _p(y), _p(out), x.numel(), _stream()), "pnp_add")
    return out


def fill_(t, value):
    check(_lib.load().pnp_fill(_p(t), t.numel(), float(value), _stream()), "pnp_fill")
    return t


def critic_input_fwd(a, tile_a, b, c, d, logits):
    lib = _lib.load()
    P = logits.numel() // logits.shape[-1]
    Ca, Cb, Cc, Cd, ncls = a.shape[-1], b.shape[-1], c.shape[-1], d.shape[-1], logits.shape[-1]
    Ct = Ca * tile_a + Cb + Cc + Cd + ncls + 1
    out = torch.empty(logits.shape[:-1] + (Ct,), dtype=torch.float32, device=logits.device)
    check(lib.pnp_critic_input_fwd(_p(a), Ca, tile_a, _p(b), Cb, _p(c), Cc, _p(d), Cd, _p(logits), ncls, _p(out), P, _stream()),
          "pnp_critic_input_fwd")
    return out


def critic_input_bwd(dout, shapes, tile_a, need=(True,) * 5):
    lib = _lib.load()
    (sa, sb, sc, sd, sl) = shapes
    P = dout.numel() // dout.shape[-1]
    mk = lambda s, n: torch.empty(s, dtype=torch.float32, device=dout.device) if n else None
    da, db, dc, dd, dl = mk(sa, need[0]), mk(sb, need[1]), mk(sc, need[2]), mk(sd, need[3]), mk(sl, need[4])
    check(lib.pnp_critic_input_bwd(_p(dout), _p(da), sa[-1], tile_a, _p(db), sb[-1], _p(dc), sc[-1], _p(dd), sd[-1], _p(dl), sl[-1],
                                   P, _stream()), "pnp_critic_input_bwd")
    return da, db, dc, dd, dl


def wgan_loss(ct_cls, mr_cls, ct_mask, mr_mask, coefs):
    """adversarial.py:455-459: sum_i coef_i * mean(logits_i) over the non-None critic outputs -> 1-element tensor"""
    ref = next(t for t in (ct_cls, mr_cls, ct_mask, mr_mask) if t is not None)
    out = torch.empty(1, dtype=torch.float32, device=ref.device)
    check(_lib.load().pnp_wgan_loss(_p(ct_cls), _p(mr_cls), _p(ct_mask), _p(mr_mask), ref.numel(), float(coefs[0]), float(coefs[1]),
                                    float(coefs[2]), float(coefs[3]), _p(out), _stream()), "pnp_wgan_loss")
    return out


def filled(shape, value, device):
    t = torch.empty(shape, dtype=torch.float32, device=device)
    check(_lib.load().pnp_fill(_p(t), t.numel(), float(value), _stream()), "pnp_fill")
    return t
