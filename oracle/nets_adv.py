"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/tf_ops.py header; PARITY UNPINNED for TF arithmetic).

CPU restatement of the adaptation graph of the reference's adversarial.py:
  zip network (MR `group_1..6` + CT `adapt_1..6`, adversarial.py:127-271), shared second half (273-318),
  feature critic (320-400), mask critic (402-443), WGAN losses + L2 (445-476), RMSProp steps + weight clip (633-656, 852-881).
Variables: dict {TF variable name: torch CPU tensor}, same names as the product's VariableStore.
Dropout stream ids are consumed in graph-construction order (one per conv call), like the product.
"""
import torch

from . import tf_ops as T

CRITIC_KEEP = 0.75


class _Ctx(object):
    """units (optional list): one record per conv(-dropout-BN-shortcut-activation) unit — its input, filter USE, BN scope, shortcut,
    conv accumulator, post-dropout tensor and output, each a distinct autograd node with retain_grad() — plus records of the data
    movement between units (PS, critic-input assembly, max-pool, the critics' final matmul), so that after backward() a test can hand
    every kernel of the product its OWN saved input and its OWN upstream gradient (teacher-forced per-kernel check,
    tests/test_gpu_teacher_forced_adv.py).  The critics' filters are used twice per discriminator step (CT and MR pass): `w_use` is a
    per-use node, so its .grad is this unit's filter gradient, not the sum."""

    def __init__(self, V, keep_prob, seed, critic_keep=CRITIC_KEEP, units=None):
        self.V, self.keep, self.seed, self.sid = V, keep_prob, seed, 0
        self.critic_keep = critic_keep
        self.units = units
        self.branch = ""

    def tap(self, t):
        """a distinct node for this USE of `t` (a tensor feeding a conv and a shortcut / a critic input otherwise accumulates both)"""
        if self.units is None or t is None or not (t.requires_grad and torch.is_grad_enabled()):
            return t
        u = t + 0
        u.retain_grad()
        return u

    def record(self, kind, **kw):
        if self.units is not None:
            kw.update(kind=kind, branch=self.branch)
            for v in kw.values():
                if torch.is_tensor(v) and v.requires_grad and not v.is_leaf:
                    v.retain_grad()
            self.units.append(kw)

    def unit(self, x, wname, scope=None, train=False, stride=1, dil=1, padding="SAME", keep=None, shortcut=None, act=True):
        """conv -> dropout [-> BN(scope)] [-> + shortcut (channel zero-padded)] [-> leaky-ReLU]: layers.conv2d / conv_bn_relu2d / the two
        halves of residual_block and DR_block (layers.py:9-45, 64-93, 145-189)"""
        V = self.V
        keep = self.keep if keep is None else keep
        xin, sc, w = self.tap(x), self.tap(shortcut), self.tap(V[wname])
        yc = T.conv2d(xin, w, stride, dil, padding)
        s = self.sid
        self.sid += 1
        yd = y = T.dropout(yc, keep, self.seed, s)
        gam = bet = None
        if scope is not None:
            gam, bet = self.tap(V[scope + "/gamma"]), self.tap(V[scope + "/beta"])
            y = T.batch_norm(y, gam, bet, V[scope + "/moving_mean"], V[scope + "/moving_variance"], train)
        if sc is not None:
            cin, cout = sc.shape[-1], y.shape[-1]
            y = (T.pad_channels(sc, cin // 2) if cout != cin else sc) + y
        if act:
            y = T.leaky_relu(y)
        self.record("conv", w=wname, w_use=w, bn=scope, gamma_use=gam, beta_use=bet, x=xin, shortcut=sc, out=y, conv=yc, dropped=yd, stride=stride, dil=dil,
                    padding=padding, keep=keep, sid=s, act=act, is_train=bool(train) if scope is not None else None)
        return y

    def conv(self, x, wname, stride=1, dil=1, padding="SAME", keep=None):
        return self.unit(x, wname, None, False, stride, dil, padding, keep, act=False)

    def cbr(self, x, wname, scope, train, stride=1, dil=1, padding="SAME", keep=None):
        return self.unit(x, wname, scope, train, stride, dil, padding, keep)

    def rb(self, x, w1, w2, scope, train, dil=1, keep=None):
        inner = self.unit(x, w1, scope + "_1", train, 1, dil, keep=keep)
        return self.unit(inner, w2, scope + "_2", train, 1, dil, keep=keep, shortcut=x)

    def pool(self, x):
        xin = self.tap(x)
        y = T.max_pool2(xin)
        self.record("pool", x=xin, out=y)
        return y

    def ps(self, x, r, nc):
        xin = self.tap(x)
        y = T.PS(xin, r, nc)
        self.record("ps", x=xin, out=y, r=r, nc=nc)
        return y

    def fc(self, h, wname):
        """tf.matmul(tf.reshape(h, [-1, D]), w) (adversarial.py:395-397, 438-440)"""
        xin, w = self.tap(h), self.tap(self.V[wname])
        y = xin.reshape(xin.shape[0], -1) @ w
        self.record("fc", w=wname, w_use=w, x=xin, out=y)
        return y


def _front(c, x, mr, train):
    """group_1..6 (mr=True) or adapt_1..6: returns (conv4_2, conv6_2)"""
    V = c.V
    g = (lambda k: "group_%d" % k) if mr else (lambda k: "adapt_%d" % k)
    bn = (lambda k, j: "%s/pred_%d_%d" % (g(k), k, j)) if mr else (lambda k, j: "%s/adapt_%d_%d" % (g(k), k, j))
    w = lambda k, i: g(k) + "/Variable" + ("" if i == 0 else "_%d" % i)
    h = c.conv(x, w(1, 0))
    h = c.rb(h, w(1, 1), w(1, 2), bn(1, 1) if mr else "adapt_1/adapt_1", train)
    h = c.pool(h)
    h = c.rb(h, w(2, 0), w(2, 1), bn(2, 1) if mr else "adapt_2/adapt_2", train)
    h = c.pool(h)
    c4 = None
    for k in (3, 4, 5, 6):
        h = c.rb(h, w(k, 0), w(k, 1), bn(k, 1), train)
        h = c.rb(h, w(k, 2), w(k, 3), bn(k, 2), train)
        if k == 3:
            h = c.pool(h)
        if k == 4:
            c4 = h
    return c4, h


def _second_half(c, x, train, n_class=5):
    V = c.V
    w = lambda k, i: "group_%d/Variable" % k + ("" if i == 0 else "_%d" % i)
    h = c.rb(x, w(7, 0), w(7, 1), "group_7/pred_7_1", train)
    b7 = c.rb(h, w(7, 2), w(7, 3), "group_7/pred_7_2", train)
    h = c.rb(b7, w(8, 0), w(8, 1), "group_8/pred_8_1", train, dil=2)
    b8 = c.rb(h, w(8, 2), w(8, 3), "group_8/pred_8_2", train, dil=2)
    h = c.cbr(b8, w(9, 0), "group_9/pred_9_1", train)
    c9 = c.cbr(h, w(9, 1), "group_9/pred_9_2", train)
    h = c.conv(c9, "group_10/Variable", padding="SYMMETRIC")
    h = c.ps(h, 8, n_class * 8)
    logits = c.conv(h, "output/Variable", padding="SYMMETRIC", keep=1.0)
    return c9, b8, b7, logits


def _classifier(c, c4, c6, b7, c9, logits):
    p = "cls_scope/"
    am = torch.argmax(logits.detach(), dim=-1, keepdim=True).to(logits.dtype)
    f4, f6, f7, f9, lg = c.ps(c4, 8, 2), c.ps(c6, 8, 4), c.ps(b7, 8, 8), c.ps(c9, 8, 8), c.tap(logits)
    ins = [c.tap(f4), c.tap(f6), c.tap(f7), c.tap(f9), lg]
    x = torch.cat([ins[0].repeat(1, 1, 1, 3), ins[1], ins[2], ins[3], lg, am], dim=3)
    c.record("critic_input", ins=ins, out=x)
    spec = [(1, 3, 2), (2, 5, 2), (3, 3, 2), (4, 3, 2), (5, 5, 4)]
    h = x
    for k, kd, sd in spec:
        s = p + "cls_%d/" % k
        h = c.rb(h, s + "Variable", s + "Variable_1", s + "cls_%d" % k, True, keep=c.critic_keep)
        h = c.cbr(h, s + "Variable_2", s + "cls_%d_3" % k, True, stride=sd, keep=c.critic_keep)
    h = c.cbr(h, p + "cls_6/Variable", p + "cls_6/cls_6", True, stride=2, padding="SYMMETRIC", keep=c.critic_keep)
    return c.fc(h, p + "cls_out/Variable")


def _mask_critic(c, logits):
    p = "mask_cls_scope/"
    h = c.cbr(logits, p + "mask_cls_1/Variable", p + "mask_cls_1/mask_cls_1", True, stride=2, keep=c.critic_keep)
    s = p + "mask_cls_2/"
    h = c.rb(h, s + "Variable", s + "Variable_1", s + "m_cls_2", True, keep=c.critic_keep)
    h = c.cbr(h, s + "Variable_2", s + "m_cls_2_3", True, stride=4, keep=c.critic_keep)
    s = p + "mask_cls_3/"
    h = c.rb(h, s + "Variable", s + "Variable_1", s + "m_cls_3", True, keep=c.critic_keep)
    h = c.cbr(h, s + "Variable_2", s + "m_cls_3_3", True, stride=4, keep=c.critic_keep)
    h = c.cbr(h, p + "mask_cls_4/Variable", p + "mask_cls_4/m_cls_4", True, stride=4, padding="SYMMETRIC", keep=c.critic_keep)
    return c.fc(h, p + "m_cls_out/Variable")


def adv_forward(V, mr, ct, keep_prob, mr_front_bn=False, joint_bn=False, ct_front_bn=False, seed=0, segmenter_no_grad=False,
                critic_keep=CRITIC_KEEP, units=None):
    """the graph of adversarial.py:82-119 for the fed branches (mr or ct may be None).  critic_keep: the critics' dropout keep
    probability — 0.75 in the reference (the builders' default argument, adversarial.py:320,402); 1.0 only for the golden fixtures."""
    c = _Ctx(V, keep_prob, seed, critic_keep, units)
    out = {}
    ctx = torch.no_grad() if segmenter_no_grad else torch.enable_grad()
    with ctx:
        z = {}
        if mr is not None:
            c.branch = "mr"
            z["mr"] = _front(c, mr, True, mr_front_bn)
        if ct is not None:
            c.branch = "ct"
            z["ct"] = _front(c, ct, False, ct_front_bn)
        feats = {}
        for br in ("ct", "mr"):
            if br in z:
                c.branch = br
                feats[br] = _second_half(c, z[br][1], joint_bn)
    for br in ("ct", "mr"):
        if br in z:
            c.branch = br
            c9, b8, b7, lg = feats[br]
            out[br + "_cls"] = _classifier(c, z[br][0], z[br][1], b7, c9, lg)
            out[br + "_logits"] = lg
    for br in ("ct", "mr"):
        if br in z:
            c.branch = br
            out[br + "_mask"] = _mask_critic(c, feats[br][3])
    return out


def wgan_losses(o, miu_dis=0.002, miu_gen=0.002, lam=0.3):
    """adversarial.py:455-474 (without the L2 terms)"""
    dis = gen = None
    if "mr_cls" in o:
        dis = -miu_dis * (o["mr_cls"] - o["ct_cls"]).mean() + lam * (-miu_dis * (o["mr_mask"] - o["ct_mask"]).mean())
    gen = -miu_gen * o["ct_cls"].mean() + lam * (-miu_gen * o["ct_mask"].mean())
    return dis, gen


def l2_coefficient(name, which, miu=0.002, gan_reg=1e-4, lam=0.3, sub_iter=1):
    """effective L2 coefficient on `name` inside dis_reg / gen_reg (adversarial.py:467-474, 644, 650): critic weight lists are
    appended once per builder call (CT and MR) -> multiplicity 2; ct_front_weights once; BN variables never."""
    if "Variable" not in name:
        return 0.0
    if which == "dis":
        if name.startswith("cls_scope/"):
            return gan_reg * miu * 2.0 / sub_iter
        if name.startswith("mask_cls_scope/"):
            return gan_reg * miu * 2.0 * lam / sub_iter
        return 0.0
    return gan_reg * miu / sub_iter if name.startswith("adapt_") else 0.0


def _set_grad(V, pred):
    for k, v in V.items():
        v.requires_grad_(bool(pred(k)) and not k.endswith(("moving_mean", "moving_variance")))
        v.grad = None


def dis_train_step(V, ms, mr, ct, keep_prob, seed, lr=3e-4, lam=0.3, miu=0.002, sub_iter=1):
    """sess.run(dis_optimizer) + sess.run(clip_op) of adversarial.py:852-861: segmenter frozen with every BN in inference mode, both
    critics on batch statistics, RMSProp over cls_vars, then the +-0.03 clip of the critics' filters.  ms: dict of RMSProp slots
    (created at 1.0 on first use).  Returns (dis_loss, grads)."""
    _set_grad(V, lambda k: "cls" in k)
    o = adv_forward(V, mr, ct, keep_prob, seed=seed, segmenter_no_grad=True)
    dis, _ = wgan_losses(o, miu_dis=miu, lam=lam)
    dis.backward()
    grads = {}
    with torch.no_grad():
        for k, v in V.items():
            if not v.requires_grad:
                continue
            g = (v.grad if v.grad is not None else torch.zeros_like(v)) + l2_coefficient(k, "dis", miu=miu, lam=lam, sub_iter=sub_iter) * v
            grads[k] = g
            T.rmsprop_update(v, g, ms.setdefault(k, torch.ones_like(v)), lr)
            if "Variable" in k:
                v.clamp_(-0.03, 0.03)
    return dis.detach(), grads, o


def gen_train_step(V, ms, ct, keep_prob, seed, lr=3e-4, lam=0.3, miu=0.002, sub_iter=1):
    """sess.run(gen_optimizer) of adversarial.py:875-881: CT front in BN-training mode, RMSProp over adapt_vars"""
    _set_grad(V, lambda k: k.startswith("adapt_"))
    o = adv_forward(V, None, ct, keep_prob, ct_front_bn=True, seed=seed)
    _, gen = wgan_losses(o, miu_gen=miu, lam=lam)
    gen.backward()
    grads = {}
    with torch.no_grad():
        for k, v in V.items():
            if not v.requires_grad:
                continue
            g = (v.grad if v.grad is not None else torch.zeros_like(v)) + l2_coefficient(k, "gen", miu=miu, lam=lam, sub_iter=sub_iter) * v
            grads[k] = g
            T.rmsprop_update(v, g, ms.setdefault(k, torch.ones_like(v)), lr)
    return gen.detach(), grads, o


def joint_train_step(V, ms_dis, ms_gen, mr, ct, keep_prob, seed, lr=3e-4):
    """the joint step BASELINE.json's metric is quoted on: 1 discriminator update on (B MR, B CT) + clip, then 1 generator update on
    the B CT slices (adversarial.py:839-882 with dis_sub_iter = gen_sub_iter = 1)"""
    d, _, _ = dis_train_step(V, ms_dis, mr, ct, keep_prob, seed, lr)
    g, _, _ = gen_train_step(V, ms_gen, ct, keep_prob, seed + 1, lr)
    return d, g
