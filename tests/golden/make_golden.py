#!/usr/bin/env python
"""Generate tests/golden/*.npz|json by executing the REFERENCE'S OWN Python (read from /root/reference at generation
time; nothing is copied into the repo) over a small numpy/torch stand-in for the `tensorflow` module.

What this pins: the op SEQUENCES and wiring of the reference — ops.PS (ops.py:3-27), layers.residual_block / DR_block /
conv_bn_relu2d / conv2d(SYMMETRIC) (layers.py), lib._label_decomp / _dice / _jaccard (lib.py),
source_segmenter.Full_DRN.create_network + _get_cost (source_segmenter.py:48-273, exec'd from the file text because the
module itself has SyntaxErrors at lines 611/619), adversarial.Full_DRN's create_zip_network / create_second_half /
create_classifier / create_mask_critic / _get_cost (adversarial.py:127-476, exec'd the same way because __init__ dies on
`self.predicter`), and the TF variable names recorded in lists/{old_bn_list,pred_bn_list,half_zip_*_vars}.  What it cannot pin: the arithmetic of the TF ops themselves (TF-1.4 is not installable here) — the
stand-in implements them with oracle.tf_ops, i.e. the same restatement the oracle uses ("parity unpinned").

Run in the build container:  python tests/golden/make_golden.py
"""
import contextlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import tf_ops as T  # noqa: E402


# ------------------------------------------------------------------------------------------------------------------
# fake tensorflow: eager numpy tensors with the handful of TF-1.4 APIs the reference's hot path touches
class _Dim(int):
    @property
    def value(self):
        return int(self)


class _Shape(object):
    def __init__(self, s):
        self.s = tuple(s)

    def as_list(self):
        return list(self.s)

    def __getitem__(self, i):
        return _Dim(self.s[i])


class FT(np.ndarray):
    def get_shape(self):
        return _Shape(self.shape)

    def eval(self):
        return np.asarray(self)


def ft(a, dtype=None):
    return np.asarray(a, dtype=dtype).view(FT)


class Graph(object):
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.vars = {}            # name -> FT, creation order
        self.nscope, self.vscope = [], []
        self.uniq = {}
        self.training_flags = {}
        self.name_init = None     # optional (name, shape) -> array: variable values as a pure function of the TF name

    def unique(self, base, var_scope):
        pre = "/".join(s for s in (self.vscope if var_scope else self.nscope) if s)
        key = (pre, base, var_scope)
        k = self.uniq.get(key, 0)
        self.uniq[key] = k + 1
        leaf = base if k == 0 else "%s_%d" % (base, k)
        return (pre + "/" + leaf) if pre else leaf


G = None


def make_tf(seed=0, weight_scale=None):
    global G
    G = Graph(seed)
    tf = types.ModuleType("tensorflow")
    tf.float32, tf.int32, tf.int64, tf.string = np.float32, np.int32, np.int64, str
    tf.AUTO_REUSE = "auto"

    def t32(x):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))

    tf.reset_default_graph = lambda: None
    tf.constant = lambda v, shape=None, **k: ft(np.full(shape, v) if shape is not None else v)
    tf.cast = lambda x, dt: ft(np.asarray(x).astype(dt))
    tf.reshape = lambda x, s: ft(np.reshape(np.asarray(x), s))
    tf.transpose = lambda x, p: ft(np.transpose(np.asarray(x), p))
    tf.split = lambda x, n, axis=0: [ft(a) for a in np.split(np.asarray(x), n, axis)]
    tf.squeeze = lambda x: ft(np.squeeze(np.asarray(x)))
    tf.concat = lambda xs, axis, name=None: ft(np.concatenate([np.asarray(a) for a in xs], axis))
    tf.expand_dims = lambda x, a: ft(np.expand_dims(np.asarray(x), a))
    tf.exp = lambda x: ft(np.exp(np.asarray(x)))
    tf.matmul = lambda a, b: ft(np.asarray(a, dtype=np.float32) @ np.asarray(b, dtype=np.float32))
    tf.log = lambda x: ft(np.log(np.asarray(x)))
    tf.reduce_sum = lambda x, axis=None, keep_dims=False: ft(np.sum(np.asarray(x), axis=axis, keepdims=keep_dims, dtype=np.float32))
    tf.reduce_mean = lambda x, axis=None: ft(np.mean(np.asarray(x), axis=axis, dtype=np.float32))
    tf.tile = lambda x, m: ft(np.tile(np.asarray(x), [int(v) for v in m]))
    tf.stack = lambda xs: [int(v) for v in xs]
    tf.shape = lambda x: np.asarray(np.asarray(x).shape)
    tf.div = lambda a, b, name=None: ft(np.asarray(a) / np.asarray(b))
    tf.add = lambda a, b: ft(np.asarray(a) + np.asarray(b))
    tf.equal = lambda a, b: np.array_equal(a, b)
    tf.clip_by_value = lambda x, lo, hi, name=None: ft(np.clip(np.asarray(x), np.float32(lo), np.float32(hi)))
    tf.argmax = lambda x, axis: ft(np.argmax(np.asarray(x), axis).astype(np.int64))
    tf.one_hot = lambda idx, depth, axis=-1: ft(np.eye(depth, dtype=np.float32)[np.asarray(idx)])
    tf.FixedLenFeature = lambda *a, **k: None

    def confusion_matrix(labels, pred, num_classes):
        cm = np.zeros((num_classes, num_classes), np.int64)
        np.add.at(cm, (np.asarray(labels), np.asarray(pred)), 1)
        return ft(cm)
    tf.confusion_matrix = confusion_matrix

    def pad(x, paddings, mode="CONSTANT"):
        p = [(int(a), int(b)) for a, b in np.asarray(paddings)]
        return ft(np.pad(np.asarray(x), p, mode="symmetric" if mode == "SYMMETRIC" else "constant"))
    tf.pad = pad

    def truncated_normal(shape, stddev=1.0):
        w = T_truncated(G.rng, tuple(shape), stddev)
        if weight_scale is not None:
            w = weight_scale(w)
        return ft(w)
    tf.truncated_normal = truncated_normal
    tf.truncated_normal_initializer = lambda stddev=1.0: (lambda shape: truncated_normal(shape, stddev))

    def Variable(initial, trainable=True, name=None):
        nm = G.unique(name or "Variable", False)
        v = ft(np.array(initial, dtype=np.float32))
        if G.name_init is not None and v.ndim >= 2:
            v = ft(G.name_init(nm, v.shape))
        G.vars[nm] = v
        return v
    tf.Variable = Variable

    def get_variable(name, shape, initializer=None, trainable=True):
        pre = "/".join(s for s in G.vscope if s)
        nm = (pre + "/" + name) if pre else name
        if nm not in G.vars:
            G.vars[nm] = initializer(shape) if G.name_init is None else ft(G.name_init(nm, tuple(shape)))
        return G.vars[nm]
    tf.get_variable = get_variable

    @contextlib.contextmanager
    def name_scope(name):
        G.nscope.append(name)
        try:
            yield name
        finally:
            G.nscope.pop()

    @contextlib.contextmanager
    def variable_scope(name, reuse=None):
        G.vscope.append(name)
        G.nscope.append(name)
        try:
            yield name
        finally:
            G.nscope.pop()
            G.vscope.pop()
    tf.name_scope, tf.variable_scope = name_scope, variable_scope
    tf.placeholder = lambda *a, **k: None
    tf.placeholder_with_default = lambda default, shape=None, name=None: default

    nn = types.ModuleType("tensorflow.nn")

    def conv2d(x, W, strides, padding):
        return ft(T.conv2d(t32(x), t32(W), int(strides[1]), 1, padding).numpy())

    def atrous_conv2d(x, W, rate, padding):
        return ft(T.conv2d(t32(x), t32(W), 1, int(rate), padding).numpy())

    def dropout(x, keep_prob):
        if float(keep_prob) != 1.0:
            raise RuntimeError("golden fixtures are generated with keep_prob == 1 (TF's dropout RNG is not reproducible)")
        return x
    nn.conv2d, nn.atrous_conv2d, nn.dropout = conv2d, atrous_conv2d, dropout
    nn.leaky_relu = lambda x, alpha=0.2: ft(T.leaky_relu(t32(x), alpha).numpy())
    nn.relu = lambda x: ft(np.maximum(np.asarray(x), 0))
    nn.max_pool = lambda x, ksize, strides, padding: ft(T.max_pool2(t32(x)).numpy())
    nn.softmax = lambda x: ft(torch.softmax(t32(x), -1).numpy())
    nn.l2_loss = lambda w: np.float32(T.l2_loss(t32(w)))
    tf.nn = nn

    contrib = types.ModuleType("tensorflow.contrib")
    layers = types.ModuleType("tensorflow.contrib.layers")

    def batch_norm(x, is_training=True, decay=0.999, scale=False, center=True, scope=None, variables_collections=None,
                   updates_collections="x", trainable=True):
        assert abs(decay - 0.90) < 1e-12 and scale and center and updates_collections is None
        pre = "/".join(s for s in G.vscope if s)
        base = ((pre + "/" + scope) if pre else scope) if scope is not None else G.unique("BatchNorm", True)
        C = np.asarray(x).shape[-1]
        for leaf, val in (("beta", 0.0), ("gamma", 1.0), ("moving_mean", 0.0), ("moving_variance", 1.0)):
            if base + "/" + leaf not in G.vars:
                G.vars[base + "/" + leaf] = ft(np.full((C,), val, np.float32))
        g, b = G.vars[base + "/gamma"], G.vars[base + "/beta"]
        mm, mv = torch.from_numpy(np.asarray(G.vars[base + "/moving_mean"])), torch.from_numpy(np.asarray(G.vars[base + "/moving_variance"]))
        y = T.batch_norm(t32(x), t32(g), t32(b), mm, mv, bool(is_training))
        return ft(y.numpy())
    layers.batch_norm = batch_norm
    contrib.layers = layers
    contrib.framework = types.ModuleType("tensorflow.contrib.framework")
    tf.contrib = contrib
    return tf


def T_truncated(rng, shape, stddev):
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def load_reference_module(name, tf):
    """import /root/reference/<name>.py against the fake tf (and stubs for nibabel / tf.python.debug)"""
    sys.modules["tensorflow"] = tf
    py = types.ModuleType("tensorflow.python")
    py.debug = types.ModuleType("tensorflow.python.debug")
    sys.modules["tensorflow.python"] = py
    sys.modules["tensorflow.python.debug"] = py.debug
    sys.modules.setdefault("nibabel", types.ModuleType("nibabel"))
    sys.modules.setdefault("pdb", __import__("pdb"))
    src = open(os.path.join(REF, name + ".py")).read()
    mod = types.ModuleType("ref_" + name)
    mod.__file__ = os.path.join(REF, name + ".py")
    exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    return mod


def he(w):
    return (w * (np.sqrt(2.0 / (w.shape[0] * w.shape[1] * w.shape[2])) / 0.01 * 0.9)).astype(np.float32) if w.ndim == 4 else w


def name_init(name, shape):
    """variable values as a pure function of the TF variable name (so that the test side can rebuild them without replaying the
    creation order): N(0,1) from default_rng(crc32(name)), scaled sqrt(2/fan_in)*0.9 for filters, 1/sqrt(fan_in) for FC matrices"""
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    w = rng.standard_normal(size=tuple(shape))
    fan = int(np.prod(shape[:-1]))
    return (w * (np.sqrt(2.0 / fan) * 0.9 if len(shape) == 4 else 1.0 / np.sqrt(fan))).astype(np.float32)


def golden_adaptation_graph(out, meta):
    """adversarial.Full_DRN's four graph builders + _get_cost (adversarial.py:127-476), exec'd from the file text because __init__ itself
    dies on `self.predicter` (line 95): MR front, CT front, shared second half on both, feature critic and mask critic on both domains,
    WGAN losses and L2 terms.  keep_prob 1 everywhere (TF's dropout RNG is not reproducible), critic BN on batch statistics."""
    tf = make_tf(9)
    G.name_init = name_init
    layers = load_reference_module("layers", tf)
    ops = load_reference_module("ops", tf)
    lib = load_reference_module("lib", tf)
    lines = open(os.path.join(REF, "adversarial.py")).read().split("\n")
    cls_src = "\n".join(lines[43:476])           # `class Full_DRN(object):` ... end of _get_cost (file lines 44-476)
    ns = {"tf": tf, "np": np, "raw_size": [256, 256, 3], "volume_size": [256, 256, 3], "label_size": [256, 256, 1],
          "_dice_eval": lib._dice_eval}
    for m in (layers, ops):
        ns.update({k: v for k, v in m.__dict__.items() if not k.startswith("__")})
    exec(compile(cls_src, "adversarial.py[44:476]", "exec"), ns)
    Full_DRN = ns["Full_DRN"]
    B = 2
    rng = np.random.default_rng(33)
    mr = rng.standard_normal((B, 256, 256, 3)).astype(np.float32)
    ct = (rng.standard_normal((B, 256, 256, 3)) * 1.2 + 0.1).astype(np.float32)
    net = Full_DRN.__new__(Full_DRN)
    net.n_class, net.batch_size = 5, B
    net.mr_front_weights, net.ct_front_weights, net.cls_weights, net.m_cls_weights, net.joint_weights = [], [], [], [], []
    net.mr, net.ct, net.keep_prob = ft(mr), ft(ct), 1.0
    mr_c4, ct_c4, mr_c6, ct_c6 = net.create_zip_network(input_channel=3, feature_base=16, num_cls=5, keep_prob=1.0, main_bn=False,
                                                        main_trainable=False, adapt_bn=True, adapt_trainable=True)
    with tf.variable_scope("", reuse=tf.AUTO_REUSE):
        ct_c9, ct_b8, ct_b7, ct_logits = net.create_second_half(ct_c6, feature_base=16, input_channel=3, num_cls=5, keep_prob=1.0,
                                                                joint_bn=False, joint_trainable=False)
        mr_c9, mr_b8, mr_b7, mr_logits = net.create_second_half(mr_c6, feature_base=16, input_channel=3, num_cls=5, keep_prob=1.0,
                                                                joint_bn=False, joint_trainable=False)
    with tf.variable_scope("cls_scope", reuse=tf.AUTO_REUSE):
        ct_cls = net.create_classifier(ct_c4, ct_c6, ct_b7, ct_c9, ct_logits, keep_prob=1.0)
        mr_cls = net.create_classifier(mr_c4, mr_c6, mr_b7, mr_c9, mr_logits, keep_prob=1.0)
    with tf.variable_scope("mask_cls_scope", reuse=tf.AUTO_REUSE):
        ct_mask = net.create_mask_critic(ct_logits, num_cls=5, keep_prob=1.0)
        mr_mask = net.create_mask_critic(mr_logits, num_cls=5, keep_prob=1.0)
    ck = {"miu_dis": 0.002, "miu_gen": 0.002, "lambda_mask_loss": 0.3, "regularizer": 1e-4, "gan_regularizer": 1e-4}
    dis_loss, gen_loss, fixed_reg, dis_reg, gen_reg = net._get_cost(ct_logits, mr_logits, ct_cls, mr_cls, ct_mask, mr_mask, dict(ck))

    def names_of(ws):
        return [[k for k, v in G.vars.items() if v is w][0] for w in ws]
    meta["adv_var_order"] = list(G.vars.keys())
    meta["adv_var_shapes"] = {k: list(np.asarray(v).shape) for k, v in G.vars.items()}
    meta["adv_lists"] = {"mr_front_weights": names_of(net.mr_front_weights), "ct_front_weights": names_of(net.ct_front_weights),
                         "cls_weights": names_of(net.cls_weights), "m_cls_weights": names_of(net.m_cls_weights),
                         "joint_weights": names_of(net.joint_weights)}
    meta["adv_cost_kwargs"] = ck
    meta["adv_scalars"] = {"dis_loss": float(dis_loss), "gen_loss": float(gen_loss), "fixed_coeff_reg": float(fixed_reg),
                           "dis_reg": float(dis_reg), "gen_reg": float(gen_reg)}
    for tag, v in (("ct_cls", ct_cls), ("mr_cls", mr_cls), ("ct_mask", ct_mask), ("mr_mask", mr_mask)):
        out["adv_" + tag] = np.asarray(v, dtype=np.float32)
    out["adv_ct_logits_sub"] = np.asarray(ct_logits)[:, ::16, ::16, :].copy()
    out["adv_mr_logits_sub"] = np.asarray(mr_logits)[:, ::16, ::16, :].copy()
    out["adv_ct_c4_sub"] = np.asarray(ct_c4)[:, ::4, ::4, ::8].copy()
    out["adv_mr_c6_sub"] = np.asarray(mr_c6)[:, ::4, ::4, ::16].copy()
    G.name_init = None


def main():
    out = {}
    meta = {"generator": "tests/golden/make_golden.py", "reference": "carrenD/Medical-Cross-Modality-Domain-Adaptation @ /root/reference"}

    # ---- ops.PS ---------------------------------------------------------------------------------------------------
    tf = make_tf(0)
    ops = load_reference_module("ops", tf)
    for tag, (B, a, b, r, nc) in {"ps_a": (2, 3, 2, 2, 2), "ps_b": (3, 2, 3, 8, 2), "ps_c": (2, 2, 2, 4, 5)}.items():
        x = np.arange(B * a * b * nc * r * r, dtype=np.float32).reshape(B, a, b, nc * r * r)
        y = ops.PS(ft(x), r, n_channel=nc, batch_size=B)
        out[tag + "_out"] = np.asarray(y)
        meta[tag] = [B, a, b, r, nc]

    # ---- lib helpers ------------------------------------------------------------------------------------------------
    lib = load_reference_module("lib", tf)
    rng = np.random.default_rng(7)
    lab = rng.integers(0, 7, size=(2, 9, 11)).astype(np.float32)      # labels 5,6 >= num_cls -> all-zero rows
    out["label_in"] = lab
    out["label_onehot"] = lib._label_decomp(5, lab)
    cm = rng.integers(0, 50, size=(5, 5))
    cm[:, 3] = 0
    cm[3, :] = 0                                                          # empty class -> the ==0 branches
    out["cm"] = cm
    out["cm_dice"] = lib._dice(cm)
    out["cm_jaccard"] = lib._jaccard(cm)

    # ---- layers.* blocks (structure pin; arithmetic = oracle) ----------------------------------------------------------
    tf = make_tf(11)
    layers = load_reference_module("layers", tf)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 8, 8, 8)).astype(np.float32)
    out["blk_x"] = x
    w1 = layers.weight_variable([3, 3, 8, 16], stddev=0.2)
    w2 = layers.weight_variable([3, 3, 16, 16], stddev=0.2)
    out["blk_rb_inc"] = np.asarray(layers.residual_block(ft(x), w1, w2, 1.0, inc_dim=True, is_train=True, leak=True))
    w3 = layers.weight_variable([3, 3, 8, 8], stddev=0.2)
    w4 = layers.weight_variable([3, 3, 8, 8], stddev=0.2)
    out["blk_drb"] = np.asarray(layers.DR_block(ft(x), w3, w4, rate=2, keep_prob=1.0, is_train=True, leak=True))
    out["blk_rb_infer_relu"] = np.asarray(layers.residual_block(ft(x), w3, w4, 1.0, is_train=False, leak=False, scope="myscope"))
    w5 = layers.weight_variable([5, 5, 8, 4], stddev=0.2)
    out["blk_conv_sym"] = np.asarray(layers.conv2d(ft(x), w5, 1.0, padding="SYMMETRIC"))
    w6 = layers.weight_variable([5, 5, 8, 4], stddev=0.2)
    out["blk_cbr_s2"] = np.asarray(layers.conv_bn_relu2d(ft(x), w6, 1.0, strides=[1, 2, 2, 1], is_train=True, leak=True))
    for k, v in G.vars.items():
        out["blkvar|" + k.replace("/", "|")] = np.asarray(v)
    meta["blk_var_order"] = list(G.vars.keys())

    # ---- Full_DRN.create_network + _get_cost from the file text ---------------------------------------------------------
    tf = make_tf(5, weight_scale=he)
    layers = load_reference_module("layers", tf)
    ops = load_reference_module("ops", tf)
    lib = load_reference_module("lib", tf)
    lines = open(os.path.join(REF, "source_segmenter.py")).read().split("\n")
    cls_src = "\n".join(lines[47:273])          # `class Full_DRN(object):` ... end of _dice_loss_fun (file lines 48-273)
    ns = {"tf": tf, "np": np, "raw_size": [256, 256, 3], "volume_size": [256, 256, 3], "label_size": [256, 256, 1],
          "_dice_eval": lib._dice_eval}
    for m in (layers, ops):
        ns.update({k: v for k, v in m.__dict__.items() if not k.startswith("__")})
    exec(compile(cls_src, "source_segmenter.py[48:273]", "exec"), ns)
    Full_DRN = ns["Full_DRN"]
    B = 2
    rng = np.random.default_rng(21)
    xin = rng.standard_normal((B, 256, 256, 3)).astype(np.float32)
    yy, xx = np.mgrid[0:256, 0:256]
    labm = np.zeros((B, 256, 256), np.float32)
    for bb in range(B):
        for c in range(1, 5):
            cy, cx = rng.integers(40, 216, 2)
            ry, rx = rng.integers(12, 40, 2)
            labm[bb][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1] = c
    yin = lib._label_decomp(5, labm)

    net = Full_DRN.__new__(Full_DRN)
    net.n_class, net.batch_size, net.conv_weights = 5, B, []
    net.x, net.y, net.keep_prob = ft(xin), ft(yin), 1.0
    logits = net.create_network(input_size=[256, 256, 3], input_channel=3, num_cls=5, feature_base=16, keep_prob=1.0,
                                adapt_module=True, main_bn=True, main_trainable=True, adapt_bn=True, adapt_trainable=True)
    net.predicter = layers.pixel_wise_softmax_2(logits)
    net.compact_pred = tf.argmax(net.predicter, 3)
    cost, reg = net._get_cost(logits, {"cross_flag": True, "miu_cross": 1.0, "dice_flag": True, "miu_dice": 1.0, "regularizer": 1e-4})
    names = list(G.vars.keys())
    conv_weight_names = []
    for w in net.conv_weights:
        conv_weight_names.append([k for k, v in G.vars.items() if v is w][0])
    meta["seg_var_order"] = names
    meta["seg_conv_weights"] = conv_weight_names
    meta["seg_seed"] = 5
    meta["seg_scalars"] = {"cost": float(cost), "reg": float(reg), "weighted_loss": float(net.weighted_loss), "dice_loss": float(net.dice_loss),
                           "dice_eval": float(net.dice_eval)}
    out["seg_x"] = xin.astype(np.float16)        # inputs are re-generated by seed in the test; kept (fp16) only as a checksum aid
    out["seg_label"] = labm.astype(np.uint8)
    lg = np.asarray(logits)
    out["seg_logits_sub"] = lg[:, ::8, ::8, :].copy()
    out["seg_argmax"] = np.asarray(net.compact_pred).astype(np.uint8)
    meta["seg_var_l2norm"] = {k: float(np.sqrt((np.asarray(v, dtype=np.float64) ** 2).sum())) for k, v in G.vars.items() if "Variable" in k}

    # ---- the adaptation graph (both domains, both critics, WGAN losses) from adversarial.py's own builders ---------------
    golden_adaptation_graph(out, meta)

    # ---- TF variable names recorded by the reference authors -----------------------------------------------------------
    for f in ("old_bn_list", "pred_bn_list", "half_zip_mri_vars", "half_zip_ct_vars"):
        meta["list_" + f] = [l.strip() for l in open(os.path.join(REF, "lists", f)) if l.strip()]

    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
    print("wrote golden.npz (%d arrays) and golden.json" % len(out))


if __name__ == "__main__":
    main()
