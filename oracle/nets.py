"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/tf_ops.py header; PARITY UNPINNED for TF arithmetic).

Table-driven CPU restatement of the reference's graphs, built from oracle.tf_ops:
  * segmenter   : Full_DRN.create_network + _get_cost          (source_segmenter.py:88-273)
  * train step  : AdamOptimizer.minimize(cost + reg)             (source_segmenter.py:357-381, 484-489)
Variables are a plain dict {TF variable name: torch CPU tensor}; the same names as the product's VariableStore so
that both sides can be loaded with identical values.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import tf_ops as T

# (group, [blocks]) ; block = (kind, cin, cout) ; kinds: conv (no BN), rb (residual), drb (dilated residual), cbr
SEGMENTER_SPEC = [
    ("group_1", [("conv", 3, 16), ("rb", 16, 16)], True),
    ("group_2", [("rb", 16, 32)], True),
    ("group_3", [("rb", 32, 64), ("rb", 64, 64)], True),
    ("group_4", [("rb", 64, 128), ("rb", 128, 128)], False),
    ("group_5", [("rb", 128, 256), ("rb", 256, 256)], False),
    ("group_6", [("rb", 256, 256), ("rb", 256, 256)], False),
    ("group_7", [("rb", 256, 512), ("rb", 512, 512)], False),
    ("group_8", [("drb", 512, 512), ("drb", 512, 512)], False),
    ("group_9", [("cbr", 512, 512), ("cbr", 512, 512)], False),
]
ADAPT_GROUPS = ("group_1", "group_2", "group_3", "group_4")   # is_train = adapt_bn ; the rest = main_bn


class _Namer(object):
    def __init__(self):
        self.cnt = {}

    def weight(self, group):
        k = self.cnt.get(group, 0)
        self.cnt[group] = k + 1
        return group + "/Variable" + ("" if k == 0 else "_%d" % k)

    def bn(self):
        k = self.cnt.get("__bn__", 0)
        self.cnt["__bn__"] = k + 1
        return "BatchNorm" + ("" if k == 0 else "_%d" % k)


def segmenter_variable_shapes(n_class=5, channels=3):
    """OrderedDict name -> shape in TF creation order (weights of a block are created before its BN variables)."""
    out = OrderedDict()
    nm = _Namer()

    def bn(c):
        b = nm.bn()
        for leaf in ("beta", "gamma", "moving_mean", "moving_variance"):
            out[b + "/" + leaf] = (c,)

    for group, blocks, _ in SEGMENTER_SPEC:
        for kind, cin, cout in blocks:
            if kind == "conv":
                out[nm.weight(group)] = (3, 3, channels, cout)
            elif kind in ("rb", "drb"):
                out[nm.weight(group)] = (3, 3, cin, cout)
                out[nm.weight(group)] = (3, 3, cout, cout)
                bn(cout)
                bn(cout)
            else:
                out[nm.weight(group)] = (3, 3, cin, cout)
                bn(cout)
    out[nm.weight("group_10")] = (3, 3, 512, 64 * n_class * 8)
    out[nm.weight("output")] = (5, 5, n_class * 8, n_class)
    return out


def l2_multiplicity(name):
    """conv_weights list of source_segmenter.py (wr4_4 appended twice, wr4_3 never: lines 132-135)"""
    if name == "group_4/Variable_2":
        return 0
    if name == "group_4/Variable_3":
        return 2
    return 1 if "/Variable" in name else 0


def segmenter_forward(V, x, keep_prob=1.0, main_bn=True, adapt_bn=True, seed=0, n_class=5, taps=None, operand_round=None, units=None,
                      round_if=None):
    """logits of Full_DRN.create_network.  V: dict of torch tensors (moving stats are updated in place when training).
    operand_round (e.g. tf_ops.round_bf16): applied to both operands of every convolution, accumulation stays float32 — the
    arithmetic of a bf16-MFMA mixed-precision path (BASELINE config 5), used to budget its tolerance on the CPU.
    round_if(filter_shape) -> bool restricts the rounding to the layers a given implementation runs on its bf16 kernels.
    units: a list that receives one record per conv(-BN-shortcut-activation) unit — its input, filter, BN names, shortcut, output
    and dropout stream id, each activation a distinct autograd node with retain_grad() — so that after backward() a test can feed
    every unit's OWN upstream gradient and saved input to the kernels under test (teacher-forced per-layer backward check)."""
    nm = _Namer()
    sid = [0]

    def tap(t):
        """a distinct node for `t` whose .grad is the gradient flowing into THIS use of it (a tensor feeding a conv and a shortcut
        otherwise accumulates both)"""
        if units is None or not t.requires_grad:
            return t
        u = t + 0
        u.retain_grad()
        return u

    def unit(x, wname, group_train, dil=1, padding="SAME", keep=keep_prob, bn=True, shortcut=None, act=True):
        w = V[wname]
        xin, sc = tap(x), (tap(shortcut) if shortcut is not None else None)
        rnd = operand_round is not None and (round_if is None or round_if(tuple(w.shape)))
        xo, wo = (operand_round(xin), operand_round(w)) if rnd else (xin, w)
        s = sid[0]
        sid[0] += 1
        yc = T.conv2d(xo, wo, 1, dil, padding)
        yd = y = T.dropout(yc, keep, seed, s)
        b = None
        if bn:
            b = nm.bn()
            y = T.batch_norm(y, V[b + "/gamma"], V[b + "/beta"], V[b + "/moving_mean"], V[b + "/moving_variance"], group_train)
        if sc is not None:
            cin, cout = sc.shape[-1], y.shape[-1]
            y = (T.pad_channels(sc, cin // 2) if cout != cin else sc) + y
        if act:
            y = T.leaky_relu(y)
        if units is not None:
            for t_ in (y, yc, yd):
                if t_.requires_grad:
                    t_.retain_grad()
            units.append({"w": wname, "bn": b, "x": xin, "shortcut": sc, "out": y, "conv": yc, "dropped": yd, "dil": dil, "padding": padding,
                          "keep": keep, "sid": s, "act": act, "is_train": group_train})
        return y

    h = x
    for group, blocks, pool in SEGMENTER_SPEC:
        is_train = adapt_bn if group in ADAPT_GROUPS else main_bn
        for kind, cin, cout in blocks:
            if kind == "conv":
                h = unit(h, nm.weight(group), is_train, bn=False, act=False)
            elif kind in ("rb", "drb"):
                dil = 2 if kind == "drb" else 1
                w1, w2 = nm.weight(group), nm.weight(group)
                inner = unit(h, w1, is_train, dil)
                h = unit(inner, w2, is_train, dil, shortcut=h)
            else:
                h = unit(h, nm.weight(group), is_train)
            if taps is not None:
                taps.append((group, kind, h))
        if pool:
            h = T.max_pool2(h)
    h = unit(h, nm.weight("group_10"), True, padding="SYMMETRIC", bn=False, act=False)
    h = T.PS(h, 8, n_class * 8)
    logits = unit(h, nm.weight("output"), True, padding="SYMMETRIC", keep=1.0, bn=False, act=False)
    return logits


def segmenter_cost(V, logits, y, miu_cross=1.0, miu_dice=1.0, reg_coeff=1e-4):
    """Full_DRN._get_cost (source_segmenter.py:211-239): (cost, regularizer, weighted_loss, dice_loss)"""
    wl = T.softmax_weighted_loss(logits, y)
    dl = T.dice_loss(logits, y)
    cost = miu_cross * wl + miu_dice * dl
    reg = 0
    for name, t in V.items():
        m = l2_multiplicity(name)
        if m:
            reg = reg + m * T.l2_loss(t)
    return cost, reg_coeff * reg, wl, dl


def make_variables(state, dtype=torch.float32, requires_grad=True):
    """state: dict name -> numpy array (e.g. the product VariableStore.state_dict())"""
    V = OrderedDict()
    for k, a in state.items():
        t = torch.from_numpy(np.array(a)).to(dtype)
        trainable = not (k.endswith("moving_mean") or k.endswith("moving_variance"))
        if requires_grad and trainable:
            t.requires_grad_(True)
        V[k] = t
    return V


def segmenter_train_step(V, opt_state, x, y, keep_prob, seed, lr=1e-3, t=1, miu_cross=1.0, miu_dice=1.0, reg_coeff=1e-4):
    """one sess.run(optimizer) of source_segmenter.py:484-489 with Adam: returns (cost, grads dict)"""
    for v in V.values():
        if v.requires_grad:
            v.grad = None
    logits = segmenter_forward(V, x, keep_prob, True, True, seed)
    cost, reg, wl, dl = segmenter_cost(V, logits, y, miu_cross, miu_dice, reg_coeff)
    (cost + reg).backward()
    grads = {}
    with torch.no_grad():
        for k, v in V.items():
            if not v.requires_grad:
                continue
            g = v.grad if v.grad is not None else torch.zeros_like(v)
            grads[k] = g.clone()
            m, vv = opt_state.setdefault(k, (torch.zeros_like(v), torch.zeros_like(v)))
            T.adam_update(v, g, m, vv, lr, t)
    return cost.detach(), grads, logits.detach()
