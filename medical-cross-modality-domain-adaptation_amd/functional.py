"""torch.autograd glue over the HIP kernels (kernels.py).  torch supplies the tape; every forward and
backward value is produced by libpnp_hip.so.

Fused units (what TF-1.4 lowers layers.py's chains to, restated as one forward + one backward each):
  Conv2dDropFn    : conv2d / atrous_conv2d -> dropout                      (layers.py:64-74, 84-93)
  ConvBNActFn     : conv -> dropout -> batch_norm [-> + shortcut] [-> leaky_relu]
                    (layers.py:9-45 and the tails of residual_block / DR_block, 145-189)
  MaxPool2Fn, PSFn, SegLossFn, CriticInputFn

Sums that TF's autodiff emits as AddN are done by the kernels, not by the engine: parameter gradients go straight into the
variable's slot of the gradient arena (gradsink.py -> pnp_conv2d_wgrad_acc / pnp_bn_bwd_acc), the gradient arriving at a block input
over the residual shortcut is added by the block's first data-gradient kernel (ResLink -> pnp_conv2d_dgrad_add).
"""
import os

import torch
from torch.autograd import Function

from . import _lib
from . import gradsink
from . import kernels as K
from . import parallel as par

BN_EPS = 1e-3      # tf.contrib.layers.batch_norm default epsilon (layers.py:100 does not override it)
BN_DECAY = 0.90    # layers.py:100
LEAK = 0.2         # tf.nn.leaky_relu default alpha (layers.py:12,35,166,187)
# inference-mode conv -> dropout -> BN -> shortcut -> activation in one kernel (pnp_conv2d_fwd_bn) whenever the BN parameters take no
# gradient (monitoring forwards, frozen-BN forwards of the GAN steps, volume inference); False: separate kernels
FUSE_BN_INFER = os.environ.get("PNP_FUSE_BN_INFER", "1") != "0"
# training-mode BN statistics from the convolution's epilogue (pnp_conv2d_fwd_stats) instead of a reduction pass over its output
FUSE_BN_STATS = os.environ.get("PNP_FUSE_BN_STATS", "1") != "0"
# backward of training-mode BN + activation WITHOUT a shortcut: the sign of the activation is recomputed from the BN input (bit for bit
# the value the forward activated) instead of read back from the saved output — one activation-sized read less in the reduction and in
# the apply kernel, and the unit's output is not kept for its own backward pass
BN_RECOMPUTE_SIGN = os.environ.get("PNP_BN_RECOMPUTE_SIGN", "1") != "0"


def sync_now():
    """Synchronised statistics only inside training steps (a tape is being recorded): monitoring / evaluation forwards run under
    no_grad and may be rank-asymmetric (rank-0-only test_eval) — a collective there would deadlock the job.  Evaluated by the CALLER
    of Function.apply (inside Function.forward grad mode is always off) and passed in as `sync`."""
    return par.sync_world() > 1 and torch.is_grad_enabled()


# Inference-mode BN saves its statistics for the backward pass BY REFERENCE (a device copy of two [C] vectors per layer cost ~200
# copy launches per GAN step).  The only writer of moving statistics is pnp_bn_update_moving below (a training-mode forward of the same
# layer); it bumps this per-buffer counter, and a backward that finds its saved statistics bumped refuses to run.
_STAT_VERSION = {}


def forget_stat_versions(lo, hi):
    """a new VariableStore packed its BN statistics into [lo, hi): counters left behind by a dead store whose arena sat at the same
    addresses must not be inherited (called by VariableStore.finalize — nothing of the new store has a backward pass pending)"""
    for k in [k for k in _STAT_VERSION if lo <= k < hi]:
        del _STAT_VERSION[k]


def _stat_version(t):
    return _STAT_VERSION.get(t.data_ptr(), 0)


def _bump_stat_version(*ts):
    for t in ts:
        _STAT_VERSION[t.data_ptr()] = _STAT_VERSION.get(t.data_ptr(), 0) + 1


def _frozen_stats(ctx, moving_mean, moving_var):
    ctx.stat_versions = (_stat_version(moving_mean), _stat_version(moving_var))
    return moving_mean, moving_var


def _check_frozen_stats(ctx, mean, var):
    sv = getattr(ctx, "stat_versions", None)
    if sv is not None and sv != (_stat_version(mean), _stat_version(var)):
        raise RuntimeError("BN moving statistics were updated between an inference-mode forward and its backward pass "
                           "(a training-mode forward of the same layer ran in between)")


def _contig(t):
    return t if t.is_contiguous() else t.contiguous()


# bf16 mode: tensors whose only readers are resident convolutions exist only as bf16 (no float32 copy is written)
BF16_ONLY_H = os.environ.get("PNP_BF16_ONLY_H", "1") != "0"
# the gradient that reaches a block input through the shortcut is added by the data-gradient kernel of the block's first convolution
# (pnp_conv2d_dgrad_add) instead of by the autograd engine's elementwise add
RES_LINK = os.environ.get("PNP_RES_LINK", "1") != "0"


class ResLink(object):
    """Ties the two conv-BN units of a residual / DR block that consume the SAME block input x (layers.residual_block, DR_block):
    the tail unit (x = its shortcut) parks the shortcut gradient here, the head unit (x = its conv input) adds it to its data
    gradient.  The tail's backward always runs before the head's (the head's output feeds the tail)."""
    __slots__ = ("dsc",)

    def __init__(self):
        self.dsc = None

    def take(self):
        d, self.dsc = self.dsc, None
        return d


# Filter gradients on a SIDE stream: a filter gradient is off the critical path of the backward pass (nothing but the optimiser reads
# it), so it runs next to the data gradient of the same layer and fills the partially empty dispatch rounds either kernel leaves on its
# own (round 4, within-run A/B at B = 16: fp32 joint 166.9 -> 168.5 slices/s, segmenter 459 -> 472, bf16 joint 436.6 -> 452.3).  Only
# inside `with wgrad_overlap():` (the step functions wrap their backward pass in it and join on exit — code that drives autograd by hand
# keeps everything on its own stream), only for gradients that go straight into the arena.  Under data parallelism too (round 6): a
# bucket's all-reduce is fenced behind BOTH the compute stream and this side stream (parallel.GradReducer._launch asks
# wgrad_side_stream()), so a gradient announced by its sink right after its kernel was queued here is complete before RCCL reads it.
# PNP_WGRAD_STREAM=0: off.
WGRAD_STREAM = os.environ.get("PNP_WGRAD_STREAM", "1") != "0"
DP_WGRAD_STREAM = os.environ.get("PNP_WGRAD_STREAM_DP", "1") != "0"      # (0: round 5's behaviour — compute stream only while bucket hooks are set)
_wgrad_side = {}
_overlap = [0]


class wgrad_overlap(object):
    """backward passes recorded inside run their filter gradients on the side stream; the compute stream joins it on exit"""

    def __enter__(self):
        _overlap[0] += 1
        return self

    def __exit__(self, *exc):
        _overlap[0] -= 1
        join_wgrad_stream()
        return False


def _side_stream():
    dev = torch.cuda.current_device()
    s = _wgrad_side.get(dev)
    if s is None:
        s = _wgrad_side[dev] = torch.cuda.Stream()
    return s


def wgrad_side_stream():
    """the side stream filter gradients of this device have been queued on so far (None: none yet / experiment off)"""
    if not (WGRAD_STREAM and _wgrad_side and torch.cuda.is_available()):
        return None
    return _wgrad_side.get(torch.cuda.current_device())


def join_wgrad_stream():
    """the compute stream waits for the filter gradients queued on the side stream (no-op when the experiment is off)"""
    if WGRAD_STREAM and _wgrad_side and torch.cuda.is_available():
        s = _wgrad_side.get(torch.cuda.current_device())
        if s is not None:
            torch.cuda.current_stream().wait_stream(s)


def _wgrad(ctx, x, dy, sink):
    """filter gradient of a conv call site: into the variable's arena slot when it has one (returns None to the engine).
    bf16-resident path (configs[4]): from the bf16 copies of x (kept by the forward) and dy."""
    g = ctx.geom
    xh = getattr(ctx, "xh", None)
    res = xh is not None and K.bf16r(g, 2)
    if sink is not None and sink.grad() is not None:
        into = sink.grad().view(g.R, g.S, g.C, g.K)
        side = _side_stream() if (WGRAD_STREAM and _overlap[0] > 0 and (DP_WGRAD_STREAM or not gradsink.has_ready_hooks())) else None
        if side is not None:
            dyh = K.bf16_of(dy) if res else None
            side.wait_stream(torch.cuda.current_stream())        # dy (and x) are complete on the compute stream up to here
            with torch.cuda.stream(side):
                if res:
                    K.conv2d_wgrad_bf16r(xh, dyh, g, into=into)
                else:
                    K.conv2d_wgrad(x, dy, g, into=into)
            for t in ((xh, dyh) if res else (x, dy)):             # the allocator may not hand these blocks out before the side kernel ran
                t.record_stream(side)
        elif res:
            K.conv2d_wgrad_bf16r(xh, K.bf16_of(dy), g, into=into)
        else:
            K.conv2d_wgrad(x, dy, g, into=into)
        gradsink.done(sink)
        return None
    return K.conv2d_wgrad_bf16r(xh, K.bf16_of(dy), g) if res else K.conv2d_wgrad(x, dy, g)


def _dgrad(ctx, dy, w, residual=None):
    """data gradient of a conv call site (bf16-resident where the geometry is served: filter shadow [tap][C][K], no flip launch)"""
    g = ctx.geom
    if K.bf16r(g, 1) and (g.stride == 1 or residual is None):       # (strided: one resident launch per stride phase, no residual input)
        return K.conv2d_dgrad_bf16r(K.bf16_of(dy), K.filter_shadows(w)[0], g, residual=residual)[0]
    if isinstance(dy, K.HalfOnly):
        raise RuntimeError("data gradient off the resident kernels got a bf16-only upstream gradient")
    return K.conv2d_dgrad(dy, w, g, residual=residual)


def _bwd_wants_h(ctx):
    """does a resident backward kernel of this call site read the bf16 copy of the upstream gradient"""
    g = ctx.geom
    return (ctx.needs_input_grad[0] and K.bf16r(g, 1)) or (ctx.needs_input_grad[1] and getattr(ctx, "xh", None) is not None and K.bf16r(g, 2))


def _bwd_only_h(ctx):
    """EVERY consumer of the gradient w.r.t. the conv accumulator is a resident kernel: the BN backward writes only the bf16 copy"""
    g = ctx.geom
    dg_ok = (not ctx.needs_input_grad[0]) or (K.bf16r(g, 1) and (g.stride == 1 or getattr(ctx, "link", None) is None))
    wg_ok = (not ctx.needs_input_grad[1]) or (getattr(ctx, "xh", None) is not None and K.bf16r(g, 2))
    return _bwd_wants_h(ctx) and dg_ok and wg_ok and BF16_ONLY_H


def _bn_bwd(ctx, dout, out, xc, mean, var, gamma, need_sc, keep, seed, sid, beta=None, want_h=False, only_h=False):
    """backward of BN (+activation, +dropout mask of the conv in front).  With synchronised statistics the per-channel sums are
    all-reduced between the reduction and the apply kernel; the PARAMETER gradients stay local (the GradReducer sums them).
    Returns (dxc, dgamma, dbeta, dsc); dgamma / dbeta are None when they went straight into the variables' gradient slots."""
    sinks = getattr(ctx, "bn_sinks", None)
    slots = (sinks[0].grad(), sinks[1].grad()) if sinks is not None else None
    if slots is not None and (slots[0] is None or slots[1] is None):
        slots = None
    if ctx.is_train and ctx.P_norm != xc.numel() // xc.shape[-1]:
        sums = K.bn_bwd_reduce(dout, out, xc, mean, var, BN_EPS, ctx.alpha, gamma, beta)
        gsums = par.all_sum_(sums.clone())
        dxc, dsc = K.bn_bwd_apply(dout, out, xc, mean, var, gamma, gsums, ctx.P_norm, need_sc, BN_EPS, ctx.alpha, True, keep, seed, sid,
                                  beta=beta, want_h=want_h, only_h=only_h)
        dgamma, dbeta = sums[0], sums[1]
        if slots is not None:                   # (opt-in SyncBN path: two [C]-sized adds)
            K.axpby(dgamma, slots[0], 1.0, 1.0)
            K.axpby(dbeta, slots[1], 1.0, 1.0)
    else:
        dxc, dgamma, dbeta, dsc = K.bn_bwd(dout, out, xc, mean, var, gamma, need_sc, BN_EPS, ctx.alpha, ctx.is_train, keep, seed, sid,
                                           into=slots, beta=beta, want_h=want_h, only_h=only_h)
    if slots is not None:
        gradsink.done(sinks[0])
        gradsink.done(sinks[1])
        return dxc, None, None, dsc
    return dxc, dgamma, dbeta, dsc


def _bn_sinks(ctx, gamma, beta, ig, ib, taped=True):
    """forward: record the uses of gamma / beta when both take a gradient and both have a slot (and a tape is being recorded)"""
    ctx.bn_sinks = None
    if taped and ctx.needs_input_grad[ig] and ctx.needs_input_grad[ib]:
        sg, sb = gradsink.lookup(gamma), gradsink.lookup(beta)
        if sg is not None and sb is not None:
            sg.pending += 1
            sb.pending += 1
            ctx.bn_sinks = (sg, sb)


class Conv2dDropFn(Function):
    @staticmethod
    def forward(ctx, x, w, geom, keep_prob, seed, stream_id, taped=True):
        """taped: torch.is_grad_enabled() AT THE CALL SITE (gradsink.py: inside forward it is always off)"""
        x = _contig(x)
        w_ = _contig(w)
        ctx.xh = None
        if K.bf16r(geom, 0):          # bf16-resident operands (configs[4])
            xh = K.bf16_of(x)
            y = K.conv2d_fwd_bf16r(xh, K.filter_shadows(w_)[1], geom, keep_prob, seed, stream_id)[0]
            if taped and ctx.needs_input_grad[1] and K.bf16r(geom, 2):
                ctx.xh = xh
        else:
            y = K.conv2d_fwd(x, w_, geom, keep_prob, seed, stream_id)
        ctx.save_for_backward(x, w_)
        ctx.geom, ctx.keep, ctx.seed, ctx.sid = geom, keep_prob, seed, stream_id
        ctx.w_sink = gradsink.use(w_, taped) if ctx.needs_input_grad[1] else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _contig(dy)
        if ctx.keep < 1.0:
            dy = K.dropout(dy, ctx.keep, ctx.seed, ctx.sid, want_h=_bwd_wants_h(ctx))
        dx = _dgrad(ctx, dy, w) if ctx.needs_input_grad[0] else None
        dw = _wgrad(ctx, x, dy, ctx.w_sink) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None, None, None


class ConvBNActFn(Function):
    """y = act( BN( dropout( conv(x,w) ) ) + pad_channels(shortcut) )"""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, moving_mean, moving_var, shortcut, geom, keep_prob, seed, stream_id, is_train, alpha, sync=False,
                link=None, taped=True):
        x = _contig(x)
        w_ = _contig(w)
        sc = _contig(shortcut) if shortcut is not None else None
        ctx.geom, ctx.keep, ctx.seed, ctx.sid = geom, keep_prob, seed, stream_id
        ctx.is_train, ctx.alpha = is_train, alpha
        ctx.sc_channels = sc.shape[-1] if sc is not None else 0
        ctx.link = link if RES_LINK else None
        ctx.w_sink = gradsink.use(w_, taped) if ctx.needs_input_grad[1] else None
        _bn_sinks(ctx, gamma, beta, 2, 3, taped)
        # (no tape: no backward pass can follow, whatever the BN parameters' requires_grad says — monitoring forwards of a TRAINABLE net)
        ctx.fused = (not is_train) and FUSE_BN_INFER and (not taped or not (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]))
        # bf16-resident operands (configs[4]): the bf16 copy of x comes from its producer (or one cast), the filter from its shadow
        res = K.bf16r(geom, 0)
        ctx.xh = None
        h_mode = geom.dtype == _lib.DTYPE_BF16          # producers leave a bf16 copy of their output next to the float32 one
        if res:
            xh = K.bf16_of(x)
            w_oi = K.filter_shadows(w_)[1]
            if taped and ctx.needs_input_grad[1] and K.bf16r(geom, 2):
                ctx.xh = xh
        if ctx.fused:
            mean, var = _frozen_stats(ctx, moving_mean, moving_var)
            ss = K.bn_fold(gamma, beta, mean, var, BN_EPS)
            if res:
                out, outh, _ = K.conv2d_fwd_bf16r(xh, w_oi, geom, keep_prob, seed, stream_id, want_h=True, bn=(ss, sc, alpha))
                K.set_bf16(out, outh)
            else:
                out = K.conv2d_fwd_bn(x, w_, geom, ss, sc, alpha, keep_prob, seed, stream_id)
            ctx.P_norm = out.numel() // out.shape[-1]
            ctx.save_for_backward(x, w_, out, out, mean, var, gamma)     # the pre-BN tensor is never read in inference mode
            return out
        P = geom.N * geom.OH * geom.OW
        ctx.P_norm = P
        # training mode on the MFMA kernels: the convolution's epilogue leaves the statistics partials (no second pass over xc)
        parts = None
        if res:
            xc, _, parts = K.conv2d_fwd_bf16r(xh, w_oi, geom, keep_prob, seed, stream_id, stat_shift=moving_mean,
                                              want_stats=is_train and FUSE_BN_STATS)
        elif is_train and FUSE_BN_STATS and K.conv_stats_parts(geom) > 0:
            xc, parts = K.conv2d_fwd_stats(x, w_, geom, moving_mean, keep_prob, seed, stream_id)
        else:
            xc = K.conv2d_fwd(x, w_, geom, keep_prob, seed, stream_id)
        if is_train:
            if sync:                  # opt-in SyncBN: statistics of the batch concatenated over the replicas
                local = K.bn_stats_finish(parts, moving_mean, P) if parts is not None else K.bn_stats(xc)
                mean, var = par.sync_bn_stats(*local)
                ctx.P_norm = P * par.sync_world()
                K.bn_update_moving(moving_mean, moving_var, mean, var, ctx.P_norm, BN_DECAY)
            elif parts is not None:
                mean, var = K.bn_stats_finish(parts, moving_mean, P, moving_mean, moving_var, BN_DECAY)
            else:
                mean, var = K.bn_stats_update(xc, moving_mean, moving_var, BN_DECAY)
            _bump_stat_version(moving_mean, moving_var)
        else:
            mean, var = _frozen_stats(ctx, moving_mean, moving_var)
        out = K.bn_apply(xc, mean, var, gamma, beta, sc, BN_EPS, alpha, want_h=h_mode)
        # no shortcut: the backward kernels recompute the activation's sign from xc, gamma, beta — `out` is not kept for this unit
        ctx.resign = BN_RECOMPUTE_SIGN and sc is None and alpha >= 0.0
        # contract of the recomputed sign: gamma / beta / statistics are NOT written between this forward and its backward.  The
        # optimiser and clip kernels write the arena in place without bumping tensor versions — the weight epoch (bumped by every
        # weight-writing kernel wrapper) stands in for them and is checked in backward
        ctx.wepoch = K._WEIGHT_EPOCH[0]
        ctx.save_for_backward(x, w_, xc, beta if ctx.resign else out, mean, var, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, xc, out, mean, var, gamma = ctx.saved_tensors
        beta = None
        if getattr(ctx, "resign", False):
            if ctx.wepoch != K._WEIGHT_EPOCH[0]:
                raise RuntimeError("parameters were updated (optimiser / clip / load) between a forward pass and its backward pass: the "
                                   "BN backward recomputes the activation's sign from gamma / beta and would use the NEW values "
                                   "(set PNP_BN_RECOMPUTE_SIGN=0 for flows that step between forward and backward)")
            out, beta = None, out
        _check_frozen_stats(ctx, mean, var)
        dout = _contig(dout)
        need_sc = ctx.sc_channels if ctx.needs_input_grad[6] else 0
        want_h = _bwd_wants_h(ctx)          # a resident backward kernel reads dxc as bf16: written by the same BN-backward launch
        only_h = _bwd_only_h(ctx)           # ... and nobody reads it as float32
        if ctx.fused:
            dxc, dsc = K.bn_bwd_apply(dout, out, out, mean, var, gamma, None, ctx.P_norm, need_sc, BN_EPS, ctx.alpha, False, ctx.keep,
                                      ctx.seed, ctx.sid, want_h=want_h, only_h=only_h)
            dgamma = dbeta = None
        else:
            dxc, dgamma, dbeta, dsc = _bn_bwd(ctx, dout, out, xc, mean, var, gamma, need_sc, ctx.keep, ctx.seed, ctx.sid, beta, want_h, only_h)
        res = None
        if ctx.link is not None:
            if ctx.sc_channels:             # tail of a block: the head's data-gradient kernel adds the shortcut gradient
                ctx.link.dsc, dsc = dsc, None
            else:                           # head of a block
                res = ctx.link.take()
        if ctx.needs_input_grad[0]:
            dx = _dgrad(ctx, dxc, w, residual=res)
        else:
            dx = None
        dw = _wgrad(ctx, x, dxc, ctx.w_sink) if ctx.needs_input_grad[1] else None
        return (dx, dw, dgamma if ctx.needs_input_grad[2] else None, dbeta if ctx.needs_input_grad[3] else None, None, None,
                dsc, None, None, None, None, None, None, None, None, None)


class BNActFn(Function):
    """batch_norm alone (layers.batch_norm, layers.py:95-100), optional activation; no conv in front."""

    @staticmethod
    def forward(ctx, xc, gamma, beta, moving_mean, moving_var, is_train, alpha, sync=False, taped=True):
        xc = _contig(xc)
        P = xc.numel() // xc.shape[-1]
        ctx.P_norm = P
        _bn_sinks(ctx, gamma, beta, 1, 2, taped)
        if is_train:
            if sync:                  # opt-in SyncBN: statistics of the batch concatenated over the replicas
                mean, var = par.sync_bn_stats(*K.bn_stats(xc))
                ctx.P_norm = P * par.sync_world()
                K.bn_update_moving(moving_mean, moving_var, mean, var, ctx.P_norm, BN_DECAY)
            else:
                mean, var = K.bn_stats_update(xc, moving_mean, moving_var, BN_DECAY)
            _bump_stat_version(moving_mean, moving_var)
        else:
            mean, var = _frozen_stats(ctx, moving_mean, moving_var)
        out = K.bn_apply(xc, mean, var, gamma, beta, None, BN_EPS, alpha)
        ctx.save_for_backward(xc, out, mean, var, gamma)
        ctx.is_train, ctx.alpha = is_train, alpha
        return out

    @staticmethod
    def backward(ctx, dout):
        xc, out, mean, var, gamma = ctx.saved_tensors
        _check_frozen_stats(ctx, mean, var)
        dxc, dgamma, dbeta, _ = _bn_bwd(ctx, _contig(dout), out, xc, mean, var, gamma, 0, 1.0, 0, 0)
        return dxc, dgamma, dbeta, None, None, None, None, None, None


class MaxPool2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _contig(x)
        ctx.save_for_backward(x)
        return K.maxpool2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.maxpool2_bwd(x, _contig(dy))


class SymPadFn(Function):
    """tf.pad(x, k//2, 'SYMMETRIC') (layers.py:19-24): materialised once so that the convolution behind it runs as a VALID
    convolution on the tap-unrolled MFMA kernel (and its wgrad / dgrad on plain zero-pad geometry)."""

    @staticmethod
    def forward(ctx, x, p):
        ctx.p = p
        return K.sympad_fwd(_contig(x), p)

    @staticmethod
    def backward(ctx, dxp):
        return K.sympad_bwd(_contig(dxp), ctx.p), None


class PSFn(Function):
    @staticmethod
    def forward(ctx, x, r, nc):
        ctx.r, ctx.nc = r, nc
        return K.ps_fwd(_contig(x), r, nc)

    @staticmethod
    def backward(ctx, dy):
        return K.ps_bwd(_contig(dy), ctx.r, ctx.nc), None, None


class SegLossFn(Function):
    """returns tensor [3] = (miu_cross*xent + miu_dice*dice, xent, dice).

    Element 0 must be the ROOT of the backward pass (the reference does minimize(cost + reg),
    source_segmenter.py:378): the upstream gradient is taken to be `gscale` (1, or 1/world_size under
    data parallelism) and is applied inside the HIP kernel — no torch arithmetic on the gradient path."""

    @staticmethod
    def forward(ctx, logits, y, miu_cross, miu_dice, gscale, sync=False):
        logits = _contig(logits)
        y = _contig(y)
        out, ws = K.seg_loss_fwd(logits, y, miu_cross, miu_dice)
        ctx.P_norm = None
        w = par.sync_world() if sync else 1
        if w > 1:       # opt-in batch-global normalisers: class counts and Dice sums of all replicas, mean over all pixels;
            par.all_sum_(K.seg_loss_sums(ws))      # `out` stays this replica's value (logging only)
            ctx.P_norm = w * (logits.numel() // logits.shape[-1])
            gscale = gscale * w                     # the replicas' gradients are summed, not averaged, in this mode
        ctx.save_for_backward(logits, y, ws)
        ctx.mc, ctx.md, ctx.gscale = miu_cross, miu_dice, gscale
        return out

    @staticmethod
    def backward(ctx, dout):
        logits, y, ws = ctx.saved_tensors
        g = K.seg_loss_bwd(logits, y, ws, ctx.mc, ctx.md, ctx.gscale, ctx.P_norm)
        return g, None, None, None, None, None


class CriticInputFn(Function):
    @staticmethod
    def forward(ctx, a, b, c, d, logits, tile_a):
        a, b, c, d, logits = map(_contig, (a, b, c, d, logits))
        ctx.shapes = tuple(tuple(t.shape) for t in (a, b, c, d, logits))
        ctx.tile_a = tile_a
        return K.critic_input_fwd(a, tile_a, b, c, d, logits)

    @staticmethod
    def backward(ctx, dout):
        need = tuple(ctx.needs_input_grad[:5])
        da, db, dc, dd, dl = K.critic_input_bwd(_contig(dout), ctx.shapes, ctx.tile_a, need)
        return da, db, dc, dd, dl, None


class FanOutFn(Function):
    """One tensor consumed by TWO branches of the graph (adversarial.py:91-119: conv4_2 / conv6_2 / block7_2 / conv9_2 / the logits feed
    the segmenter's next layer AND a critic).  TF's autodiff sums the two gradients with AddN; left to torch's engine that sum is an
    `aten::add` launch — here it is pnp_add, so that every arithmetic result of a step comes from libpnp_hip.so."""

    @staticmethod
    def forward(ctx, x):
        # an edge nobody differentiates through hands None to backward (default: a full-size zero fill by the engine plus a pnp_add —
        # create_first_half always fans out conv4_2, also when no critic consumes it)
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None or g2 is None:
            return g2 if g1 is None else g1
        return K.add(_contig(g1), _contig(g2))


def fan_out(x):
    """(x, x) as two autograd edges whose gradients are summed by pnp_add; plain (x, x) where no tape is recorded / on meta tensors"""
    if x.is_meta or not (torch.is_grad_enabled() and x.requires_grad):
        return x, x
    a, b = FanOutFn.apply(x)
    h = getattr(x, "_pnp_h", None)
    if h is not None and h[1] == x._version:          # the views are the same values: they inherit the bf16 copy
        K.set_bf16(a, h[0])
        K.set_bf16(b, h[0])
    return a, b


class WganLossFn(Function):
    """scalar = sum_i coef_i * mean(critic_logits_i)  (adversarial.py:455-459).  Must be the ROOT of backward: the upstream
    gradient is `gscale` (1/world_size under data parallelism); each input receives the constant coef_i * gscale / B."""

    @staticmethod
    def forward(ctx, ct_cls, mr_cls, ct_mask, mr_mask, coefs, gscale):
        ts = [None if t is None else _contig(t) for t in (ct_cls, mr_cls, ct_mask, mr_mask)]
        ctx.shapes = [None if t is None else tuple(t.shape) for t in ts]
        ctx.coefs, ctx.gscale = tuple(coefs), gscale
        ctx.dev = next(t for t in ts if t is not None).device
        return K.wgan_loss(ts[0], ts[1], ts[2], ts[3], coefs)

    @staticmethod
    def backward(ctx, dout):
        grads = []
        for i, shp in enumerate(ctx.shapes):
            if shp is None or not ctx.needs_input_grad[i]:
                grads.append(None)
            else:
                n = 1
                for d in shp:
                    n *= d
                grads.append(K.filled(shp, ctx.coefs[i] * ctx.gscale / n, ctx.dev))
        return grads[0], grads[1], grads[2], grads[3], None, None


# ---- arithmetic type of the convolution operands (BASELINE configs[4]) ----------------------------------------------------------
CONV_DTYPE = "f32"


def set_conv_dtype(name):
    """'f32' (the reference's arithmetic; default) or 'bf16' (bf16 MFMA operands, fp32 accumulation, fp32 master weights)"""
    global CONV_DTYPE
    if name not in ("f32", "bf16"):
        raise ValueError("conv dtype must be 'f32' or 'bf16', got %r" % (name,))
    CONV_DTYPE = name
    K.CONV_DTYPE = _lib.DTYPE_BF16 if name == "bf16" else _lib.DTYPE_F32
