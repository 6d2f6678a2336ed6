"""Drop-in for the reference's layers.py (same function names, argument order and defaults), executing
eagerly on MI355X through libpnp_hip.so instead of building TF-1.4 graph ops.

Differences that are inherent to leaving TF graph mode (documented in INTEGRATION.md):
  * tensors are torch CUDA tensors (NHWC float32); `W` is a variable tensor from `weight_variable`
  * `keep_prob` / `is_train` are plain Python values, not placeholders
  * variables live in the active `variables.VariableStore` (the stand-in for the TF default graph)
Each function cites the reference lines it replaces.
"""
from math import floor  # noqa: F401  (the reference imports it; kept for parity of the module surface)

import numpy as np
import torch

from . import kernels as K
from .functional import BNActFn, Conv2dDropFn, ConvBNActFn, LEAK, MaxPool2Fn, ResLink, SymPadFn, sync_now
from .variables import current_store, truncated_normal


def _stride_of(strides):
    if strides[0] != 1 or strides[3] != 1 or strides[1] != strides[2]:
        raise ValueError("only [1,s,s,1] strides are supported (the reference uses nothing else): %r" % (strides,))
    return int(strides[1])


def _bn_vars(scope, C, trainable):
    """tf.contrib.layers.batch_norm variable set: <scope>/{beta,gamma,moving_mean,moving_variance}; scope=None ->
    'BatchNorm', 'BatchNorm_1', ... (TF default_name uniquification inside the current variable scope)."""
    st = current_store()
    base = st.scoped(scope) if scope is not None else st.unique("BatchNorm", var_scope=True)
    beta = st.get(base + "/beta", (C,), 0.0, trainable, "bn")
    gamma = st.get(base + "/gamma", (C,), 1.0, trainable, "bn")
    mm = st.get(base + "/moving_mean", (C,), 0.0, False, "bn_stat")
    mv = st.get(base + "/moving_variance", (C,), 1.0, False, "bn_stat")
    return gamma.tensor, beta.tensor, mm.tensor, mv.tensor


def _drop_ids(keep_prob):
    st = current_store()
    sid = st.next_drop_stream()          # one stream id per conv call site, consumed even when keep_prob == 1
    return float(keep_prob), st.drop_seed, sid


# ---- layers.py:64-74 ---------------------------------------------------------------------------------
def _meta_out(g):
    return torch.empty((g.N, g.OH, g.OW, g.K), device="meta")


def _prepad(x, W, padding):
    """'SYMMETRIC' = tf.pad(x, floor(k/2), 'SYMMETRIC') + VALID conv (layers.py:19-24): do exactly that — one mirror-pad kernel,
    then a VALID convolution (the MFMA kernels also accept the mirror folded into their gather; pre-padding is faster)."""
    if padding == 'SYMMETRIC' and not x.is_meta and W.shape[0] == W.shape[1]:
        return SymPadFn.apply(x, int(W.shape[0]) // 2), 'VALID'
    return x, padding


def conv2d(x, W, keep_prob_, strides=[1, 1, 1, 1], padding='SAME'):
    x, padding = _prepad(x, W, padding)
    g = K.conv_geom(tuple(x.shape), tuple(W.shape), _stride_of(strides), 1, padding)
    keep, seed, sid = _drop_ids(keep_prob_)
    if x.is_meta:           # symbolic build pass (graph construction): shapes and variables only
        return _meta_out(g)
    return Conv2dDropFn.apply(x, W, g, keep, seed, sid, torch.is_grad_enabled())


# ---- layers.py:84-93 ---------------------------------------------------------------------------------
def dilate_conv2d(x, W, keep_prob_, rate=2, padding='SAME'):
    x, padding = _prepad(x, W, padding)
    g = K.conv_geom(tuple(x.shape), tuple(W.shape), 1, int(rate), padding)
    keep, seed, sid = _drop_ids(keep_prob_)
    if x.is_meta:
        return _meta_out(g)
    return Conv2dDropFn.apply(x, W, g, keep, seed, sid, torch.is_grad_enabled())


def _conv_bn(x, W, keep_prob, padding, stride, dil, is_train, scope, bn_trainable, alpha, shortcut=None, link=None):
    x, padding = _prepad(x, W, padding)
    g = K.conv_geom(tuple(x.shape), tuple(W.shape), stride, dil, padding)
    gamma, beta, mm, mv = _bn_vars(scope, g.K, bn_trainable)
    keep, seed, sid = _drop_ids(keep_prob)
    if x.is_meta:
        return _meta_out(g)
    return ConvBNActFn.apply(x, W, gamma, beta, mm, mv, shortcut, g, keep, seed, sid, bool(is_train), float(alpha), sync_now(), link,
                             torch.is_grad_enabled())


# ---- layers.py:16-27 ---------------------------------------------------------------------------------
def conv_bn_2d(x, W, keep_prob, padding='SAME', strides=[1, 1, 1, 1], is_train=True, scope=None, bn_trainable=True):
    return _conv_bn(x, W, keep_prob, padding, _stride_of(strides), 1, is_train, scope, bn_trainable, -1.0)


# ---- layers.py:9-14 ----------------------------------------------------------------------------------
def conv_bn_relu2d(x, W, keep_prob, padding='SAME', strides=[1, 1, 1, 1], is_train=True, scope=None, bn_trainable=True,
                   leak=False):
    return _conv_bn(x, W, keep_prob, padding, _stride_of(strides), 1, is_train, scope, bn_trainable, LEAK if leak is True else 0.0)


# ---- layers.py:39-45 ---------------------------------------------------------------------------------
def dilate_conv_bn(x, W, keep_prob, padding='SAME', rate=2, is_train=True, scope=None, bn_trainable=True):
    return _conv_bn(x, W, keep_prob, padding, 1, int(rate), is_train, scope, bn_trainable, -1.0)


# ---- layers.py:29-37 ---------------------------------------------------------------------------------
def dilate_conv_bn_relu2d(x, W, keep_prob, padding='SAME', rate=2, is_train=True, scope=None, bn_trainable=True, leak=False):
    return _conv_bn(x, W, keep_prob, padding, 1, int(rate), is_train, scope, bn_trainable, LEAK if leak is True else 0.0)


# ---- layers.py:77-82 (dead code in the reference; exported for surface completeness) -------------------
def conv_relu2d(x, W, keep_prob, padding='SAME', strides=[1, 1, 1, 1], leak=False):
    raise NotImplementedError("conv_relu2d has no call site in the reference (layers.py:77-82) and is not on the hot path")


# ---- layers.py:95-100 --------------------------------------------------------------------------------
def batch_norm(x, is_training=True, scope=None, trainable=True):
    gamma, beta, mm, mv = _bn_vars(scope, x.shape[-1], trainable)
    if x.is_meta:
        return torch.empty(tuple(x.shape), device="meta")
    return BNActFn.apply(x, gamma, beta, mm, mv, bool(is_training), -1.0, sync_now(), torch.is_grad_enabled())


# ---- layers.py:102-103 -------------------------------------------------------------------------------
def max_pool2d(x, n):
    if n != 2:
        raise ValueError("max_pool2d: only n=2 is on the reference path (source_segmenter.py:97,106,118)")
    if x.is_meta:
        return torch.empty((x.shape[0], x.shape[1] // 2, x.shape[2] // 2, x.shape[3]), device="meta")
    return MaxPool2Fn.apply(x)


# ---- layers.py:47-55 ---------------------------------------------------------------------------------
def weight_variable(shape, stddev=0.01, trainable=True):
    """tf.Variable(tf.truncated_normal(shape, stddev)) named '<name_scope>/Variable[_k]'."""
    st = current_store()
    name = st.unique("Variable")
    return st.get(name, shape, lambda rng, s: truncated_normal(rng, s, stddev), trainable, "weight").tensor


def sharable_weight_variable(shape, stddev=0.1, trainable=True, name="IhaveNoName"):
    """tf.get_variable(name, shape, truncated_normal_initializer(stddev)): shared by scope + name."""
    st = current_store()
    return st.get(st.scoped(name), shape, lambda rng, s: truncated_normal(rng, s, stddev), trainable, "weight").tensor


# ---- helpers without a call site in the reference: exported so that `from layers import *` finds every name ----
def weight_variable_deconv(shape, stddev=0.1):
    """layers.py:57-58: tf.Variable(tf.truncated_normal(shape, stddev)) — a plain filter variable, no arithmetic involved"""
    return weight_variable(shape, stddev=stddev)


def bias_variable(shape):
    """layers.py:60-62: tf.Variable(tf.constant(0.1, shape)) named like any other tf.Variable"""
    st = current_store()
    return st.get(st.unique("Variable"), shape, lambda rng, s: np.full(s, 0.1, np.float32), True, "weight").tensor


def _dead(name, where):
    def f(*a, **k):
        raise NotImplementedError("%s has no call site in the reference (%s) and is not on the hot path" % (name, where))
    f.__name__ = name
    return f


avg_pool2d = _dead("avg_pool2d", "layers.py:105-106")
crop_and_concat = _dead("crop_and_concat", "layers.py:108-115")
pixel_wise_softmax = _dead("pixel_wise_softmax", "layers.py:129-132")
cross_entropy = _dead("cross_entropy", "layers.py:140-141")


# ---- layers.py:117-127 -------------------------------------------------------------------------------
def simple_concat2d(x1, x2):
    """concatenation on the channel axis without offset check"""
    if tuple(x1.shape[:-1]) != tuple(x2.shape[:-1]):
        print("x1_shape: %s" % str(list(x1.shape)))
        print("x2_shape: %s" % str(list(x2.shape)))
        raise ValueError("Cannot concatenate tensors with different shape, igonoring feature map depth")
    return torch.cat([x1, x2], 3)   # pure data movement; the critic input uses the fused pnp_critic_input kernel instead


# ---- layers.py:134-138 -------------------------------------------------------------------------------
def pixel_wise_softmax_2(output_map):
    prob, _ = K.softmax_argmax(output_map.detach().contiguous(), want_prob=True)
    return prob


# ---- layers.py:145-166 -------------------------------------------------------------------------------
def residual_block(x, w1, w2, keep_prob, inc_dim=False, is_train=True, scope=None, bn_trainable=True, leak=False,
                   padding='SAME'):
    """conv-BN-act -> conv-BN -> (+ x, channel zero-padded C/2 each side when inc_dim) -> act.
    The shortcut add, the channel padding and the activation are fused into the second BN kernel."""
    if scope is None:
        s1 = s2 = None
    else:
        s1, s2 = scope + "_1", scope + "_2"
    alpha = LEAK if leak is True else 0.0
    # x feeds the first conv AND the shortcut: one fused gradient add (functional.ResLink).  Not with SYMMETRIC padding: the first conv
    # then reads a mirror-padded copy of x, whose gradient has another shape.
    link = ResLink() if (padding == 'SAME' and not x.is_meta) else None
    inner = _conv_bn(x, w1, keep_prob, padding, 1, 1, is_train, s1, bn_trainable, alpha, link=link)
    C = x.shape[-1]
    Cout = w2.shape[-1]
    if inc_dim is True:
        if Cout != C + 2 * (C // 2):
            raise ValueError("residual_block(inc_dim): shortcut %d + 2*%d != %d output channels" % (C, C // 2, Cout))
    elif Cout != C:
        raise ValueError("residual_block: shortcut has %d channels, block output %d" % (C, Cout))
    return _conv_bn(inner, w2, keep_prob, padding, 1, 1, is_train, s2, bn_trainable, alpha, shortcut=x, link=link)


# ---- layers.py:168-189 -------------------------------------------------------------------------------
def DR_block(x, w1, w2, rate, keep_prob, inc_dim=False, is_train=True, bn_trainable=True, scope=None, leak=False):
    if scope is None:
        s1 = s2 = None
    else:
        s1, s2 = scope + "_1", scope + "_2"
    alpha = LEAK if leak is True else 0.0
    link = None if x.is_meta else ResLink()
    inner = _conv_bn(x, w1, keep_prob, 'SAME', 1, int(rate), is_train, s1, bn_trainable, alpha, link=link)
    C = x.shape[-1]
    Cout = w2.shape[-1]
    if (inc_dim is True and Cout != C + 2 * (C // 2)) or (inc_dim is not True and Cout != C):
        raise ValueError("DR_block: shortcut/output channel mismatch (%d -> %d, inc_dim=%s)" % (C, Cout, inc_dim))
    return _conv_bn(inner, w2, keep_prob, 'SAME', 1, int(rate), is_train, s2, bn_trainable, alpha, shortcut=x, link=link)
