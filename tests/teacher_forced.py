"""Shared machinery of the teacher-forced per-kernel parity tests (tests/test_gpu_teacher_forced_adv.py).

The oracle (oracle/nets_adv.py, `units=` recorder) runs a whole step once; every record then hands the product's kernels THEIR OWN saved
inputs and THEIR OWN upstream gradient, exactly as functional.ConvBNActFn / Conv2dDropFn / PSFn / CriticInputFn / MaxPool2Fn would
call them, and each kernel must reproduce the oracle's result of that same operation to 1e-4 of max|ref| (north_star's bar).  Why the
forcing is needed (leaky-ReLU slope flips amplify float32 round-off in whole-step comparisons) is measured in
tests/test_gpu_teacher_forced.py.
"""
import numpy as np
import torch

from conftest import pkg

TOL = 1e-4
EPS = 1e-3


from parity_util import batch_moments64, rel  # noqa: E402,F401  (rel: float64 on the device the product's result lives on)


class Checker(object):
    """walks the records of one oracle step; `sd0` = the variable values (incl. BN moving statistics) the step started from"""

    def __init__(self, dev, V, sd0, seed, tag):
        self.K = pkg("kernels")
        self.dev, self.V, self.sd0, self.seed, self.tag = dev, V, sd0, seed, tag
        self.rows = []          # (label, {check: error}, kernel class)
        self.kernels = set()    # which product kernel families were exercised (reported, and asserted by the tests)
        self.flips = 0
        self.pool_ties = 0

    def f32(self, t):
        return t.detach().float().contiguous().to(self.dev)

    def var32(self, name):
        return torch.from_numpy(np.asarray(self.sd0[name], dtype=np.float32)).to(self.dev)

    # ------------------------------------------------------------------------------------------------------------------------------
    def conv_unit(self, u):
        K, f32 = self.K, self.f32
        V = self.V
        w64, xin, sc, out, yc, yd = V[u["w"]], u["x"], u["shortcut"], u["out"], u["conv"], u["dropped"]
        keep, sid, stride, dil = float(u["keep"]), int(u["sid"]), int(u["stride"]), int(u["dil"])
        xd, wd = f32(xin), f32(w64)
        R = wd.shape[0]
        xin_d, padding = xd, u["padding"]
        if padding == "SYMMETRIC":          # layers._prepad: tf.pad SYMMETRIC materialised, then a VALID convolution
            xin_d, padding = K.sympad_fwd(xd, R // 2), "VALID"
        g = K.conv_geom(tuple(xin_d.shape), tuple(wd.shape), stride, dil, padding)
        errs = {}
        label = "%s[%s] %dx%d s%d d%d %d->%d@%d %s" % (u["w"], u["branch"], R, R, stride, dil, wd.shape[2], wd.shape[3], xd.shape[1],
                                                        "train" if u["is_train"] else ("infer" if u["bn"] else "nobn"))
        alpha = 0.2 if u["act"] else -1.0
        # ---- the convolution (+ dropout) alone
        conv_hip = K.conv2d_fwd(xin_d, wd, g, keep, self.seed, sid)
        errs["conv"] = rel(conv_hip, yd)
        dyc = yc.grad if yc.requires_grad else None          # gradient w.r.t. the conv accumulator (dropout mask applied)
        P = yd.numel() // yd.shape[-1]
        scd = f32(sc) if sc is not None else None
        need_sc = sc.shape[-1] if (sc is not None and sc.requires_grad and sc.grad is not None) else 0
        if u["bn"] is not None:
            b = u["bn"]
            gam, bet = f32(V[b + "/gamma"]), f32(V[b + "/beta"])
            mm0, mv0 = self.var32(b + "/moving_mean"), self.var32(b + "/moving_variance")
            xc = f32(yd)
            if u["is_train"]:
                m64, v64 = batch_moments64(yd, self.dev)
                # batch statistics the way the product gets them: from the convolution's epilogue where the planner offers it
                mm, mv = mm0.clone(), mv0.clone()
                if K.conv_stats_parts(g) > 0:
                    _, parts = K.conv2d_fwd_stats(xin_d, wd, g, mm, keep, self.seed, sid)      # shift = the moving mean, updated in place
                    mean_h, var_h = K.bn_stats_finish(parts, mm, P, mm, mv, 0.9)             # by the same call, as in ConvBNActFn
                    self.kernels.add("conv_fwd_stats")
                else:
                    mean_h, var_h = K.bn_stats_update(conv_hip, mm, mv, 0.9)
                    self.kernels.add("bn_stats")
                scale = float(v64.sqrt().max()) + 1e-30
                errs["mean"] = float((mean_h.cpu().double() - m64).abs().max()) / scale      # against the spread, not against |mean| ~ 0
                errs["var"] = rel(var_h, v64)
                bessel = P / (P - 1.0) if P > 1 else 1.0
                errs["mov_mean"] = float((mm.cpu().double() - (mm0.cpu().double() * 0.9 + 0.1 * m64)).abs().max()) / scale
                errs["mov_var"] = rel(mv, mv0.cpu().double() * 0.9 + 0.1 * v64 * bessel)
                mean_d, var_d = m64.float().to(self.dev), v64.float().to(self.dev)
                out_hip = K.bn_apply(xc, mean_d, var_d, gam, bet, scd, EPS, alpha)
                errs["out"] = rel(out_hip, out)
                self.flips += int(((out_hip > 0) != (f32(out) > 0)).sum())
                if out.requires_grad and out.grad is not None:
                    dxc, dgamma, dbeta, dsc = K.bn_bwd(f32(out.grad), f32(out), xc, mean_d, var_d, gam, need_sc, EPS, alpha, True, keep,
                                                       self.seed, sid)
                    errs["dconv"] = rel(dxc, dyc)
                    gu, bu = u["gamma_use"], u["beta_use"]
                    if gu.requires_grad and gu.grad is not None:
                        errs["dgamma"], errs["dbeta"] = rel(dgamma, gu.grad), rel(dbeta, bu.grad)
                    if need_sc:
                        errs["dshortcut"] = rel(dsc, sc.grad)
                    self.kernels.add("bn_bwd_train")
                    if sc is None and alpha >= 0.0:
                        # the product does not keep `out` for such a unit: the kernels recompute the activation's sign from the BN
                        # input.  Must equal, BIT FOR BIT, the backward that is handed the kernels' own forward output.
                        a = K.bn_bwd(f32(out.grad), out_hip, xc, mean_d, var_d, gam, 0, EPS, alpha, True, keep, self.seed, sid)
                        b_ = K.bn_bwd(f32(out.grad), None, xc, mean_d, var_d, gam, 0, EPS, alpha, True, keep, self.seed, sid, beta=bet)
                        errs["resign_bits"] = float(sum(int((u_ != v_).sum()) for u_, v_ in zip(a[:3], b_[:3])))
                        self.kernels.add("bn_bwd_resign")
            else:
                # frozen BN: the product folds it (+ shortcut + leaky-ReLU) into the convolution's epilogue (pnp_conv2d_fwd_bn)
                out_hip = K.conv2d_fwd_bn(xin_d, wd, g, K.bn_fold(gam, bet, mm0, mv0, EPS), scd, alpha, keep, self.seed, sid)
                errs["out_fused"] = rel(out_hip, out)
                self.kernels.add("conv_fwd_bn_fused")
                if out.requires_grad and out.grad is not None:
                    outd = f32(out)
                    dxc, dsc = K.bn_bwd_apply(f32(out.grad), outd, outd, mm0, mv0, gam, None, P, need_sc, EPS, alpha, False, keep, self.seed,
                                              sid)
                    errs["dconv"] = rel(dxc, dyc)
                    if need_sc:
                        errs["dshortcut"] = rel(dsc, sc.grad)
                    self.kernels.add("bn_bwd_frozen")
        elif dyc is not None and yd.grad is not None:
            errs["dconv"] = rel(K.dropout(f32(yd.grad), keep, self.seed, sid) if keep < 1.0 else f32(yd.grad), dyc)
        # ---- conv backward kernels on the oracle's d(conv accumulator)
        if dyc is not None:
            dyc_d = f32(dyc)
            wu = u["w_use"]
            if wu.requires_grad and wu.grad is not None:
                errs["dw"] = rel(K.conv2d_wgrad(xin_d, dyc_d, g), wu.grad)
                into = f32(wu.grad).clone()                    # the gradient-sink form: ADD into a slot that already holds a contribution
                K.conv2d_wgrad(xin_d, dyc_d, g, into=into)
                errs["dw_acc"] = rel(into, 2.0 * wu.grad)
                self.kernels.add("wgrad_s%d_k%d" % (stride, R))
            if xin.requires_grad and xin.grad is not None:
                dx = K.conv2d_dgrad(dyc_d, wd, g)
                if u["padding"] == "SYMMETRIC":
                    dx = K.sympad_bwd(dx, R // 2)
                errs["dx"] = rel(dx, xin.grad)
                self.kernels.add("dgrad_s%d_k%d" % (stride, R))
                if u["padding"] == "SAME" and stride == 1 and sc is None and u["bn"] is not None:
                    # head of a residual block: the shortcut gradient is added by the data-gradient kernel (functional.ResLink)
                    res = f32(xin.grad)
                    errs["dx_add"] = rel(K.conv2d_dgrad(dyc_d, wd, g, residual=res), 2.0 * xin.grad)
        mode = "train-BN" if u["is_train"] else ("frozen-BN" if u["bn"] else "no-BN")
        pad = "SYM" if u["padding"] == "SYMMETRIC" else "SAME"
        self.rows.append((label, errs, "conv %dx%d s%d d%d %s %s" % (R, R, stride, dil, pad, mode)))

    def fc_unit(self, u):
        K, f32 = self.K, self.f32
        x, w, out = u["x"], self.V[u["w"]], u["out"]
        B = x.shape[0]
        D = w.shape[0]
        xd, wd = f32(x).reshape(B, 1, 1, D), f32(w).reshape(1, 1, D, 1)
        g = K.conv_geom((B, 1, 1, D), (1, 1, D, 1), 1, 1, "VALID")
        errs = {"out": rel(K.conv2d_fwd(xd, wd, g).reshape(B, 1), out)}
        if out.requires_grad and out.grad is not None:
            dy = f32(out.grad).reshape(B, 1, 1, 1)
            wu = u["w_use"]
            if wu.requires_grad and wu.grad is not None:
                errs["dw"] = rel(K.conv2d_wgrad(xd, dy, g).reshape(D, 1), wu.grad)
            if x.requires_grad and x.grad is not None:
                errs["dx"] = rel(K.conv2d_dgrad(dy, wd, g).reshape(x.shape), x.grad)
        self.kernels.add("fc")
        self.rows.append(("%s[%s] fc %d" % (u["w"], u["branch"], D), errs, "fc (1x1 conv)"))

    def pool_unit(self, u):
        K, f32 = self.K, self.f32
        x, out = u["x"], u["out"]
        xd = f32(x)
        errs = {"out": rel(K.maxpool2_fwd(xd), out)}
        if out.requires_grad and out.grad is not None:
            dx = K.maxpool2_bwd(xd, f32(out.grad))
            ref = x.grad.float()
            bad = (dx.cpu() != ref)
            nbad = int(bad.sum())
            if nbad and x.dtype == torch.float64:
                # a 2x2 window whose two largest float64 values round to the same float32 routes its gradient to another pixel: count
                # them (each contributes 2 differing elements) and take them out of the comparison
                self.pool_ties += nbad // 2
                assert nbad <= 16, "max-pool backward differs at %d elements" % nbad
                dx = torch.where(bad.to(self.dev), ref.to(self.dev), dx)
            errs["dx"] = rel(dx, ref)
        self.kernels.add("maxpool")
        self.rows.append(("max_pool[%s] %d@%d" % (u["branch"], x.shape[-1], x.shape[1]), errs, "max_pool"))

    def ps_unit(self, u):
        K, f32 = self.K, self.f32
        x, out = u["x"], u["out"]
        errs = {"out": rel(K.ps_fwd(f32(x), u["r"], u["nc"]), out)}
        if out.requires_grad and out.grad is not None:
            errs["dx"] = rel(K.ps_bwd(f32(out.grad), u["r"], u["nc"]), x.grad)
        self.kernels.add("ps")
        self.rows.append(("PS[%s] %d->%d" % (u["branch"], x.shape[-1], u["nc"]), errs, "PS"))

    def critic_input_unit(self, u):
        K, f32 = self.K, self.f32
        ins, out = u["ins"], u["out"]
        a, b, c, d, lg = (f32(t) for t in ins)
        got = K.critic_input_fwd(a, 3, b, c, d, lg)
        # last channel = float(argmax of the logits): a differing label is accepted only at float64 near-ties of the logits
        l64 = ins[4].detach().double()
        top2 = torch.topk(l64, 2, dim=-1).values
        near_tie = (top2[..., 0] - top2[..., 1]) <= 1e-5 * float(l64.abs().max())
        lab_bad = (got[..., -1].cpu().double() != out[..., -1].detach().double()) & ~near_tie
        errs = {"out": rel(got[..., :-1], out[..., :-1]), "argmax_channel": float(int(lab_bad.sum()))}
        if out.requires_grad and out.grad is not None:
            need = tuple(bool(t.requires_grad and t.grad is not None) for t in ins)
            grads = K.critic_input_bwd(f32(out.grad), tuple(tuple(t.shape) for t in ins), 3, need)
            for name, gk, t, n in zip(("d_f4", "d_f6", "d_f7", "d_f9", "d_logits"), grads, ins, need):
                if n:
                    errs[name] = rel(gk, t.grad)
            self.kernels.add("critic_input_bwd")
        self.kernels.add("critic_input_fwd")
        self.rows.append(("critic_input[%s]" % u["branch"], errs, "critic_input"))

    # ------------------------------------------------------------------------------------------------------------------------------
    def run(self, units):
        for u in units:
            getattr(self, {"conv": "conv_unit", "fc": "fc_unit", "pool": "pool_unit", "ps": "ps_unit",
                           "critic_input": "critic_input_unit"}[u["kind"]])(u)
        return self

    def report(self):
        worst = {}
        for label, errs, _ in self.rows:
            for k_, e in errs.items():
                if e > worst.get(k_, (-1.0, ""))[0]:
                    worst[k_] = (e, label)
        for k_, (e, w) in sorted(worst.items()):
            print("teacher-forced %s: worst %-14s %.3e at %s" % (self.tag, k_, e, w))
        for cls, checks in sorted(self.class_table().items()):
            print("teacher-forced %s: class %-34s %s" % (self.tag, cls, "  ".join("%s %.1e" % (k_, e) for k_, e in sorted(checks.items()))))
        print("teacher-forced %s: %d records; kernel families: %s; leaky-ReLU sign differences HIP vs oracle forward: %d; max-pool float32 ties: %d"
              % (self.tag, len(self.rows), ", ".join(sorted(self.kernels)), self.flips, self.pool_ties))
        return worst

    def class_table(self):
        """{kernel class: {check: worst error}} — the table DESIGN.md §2 quotes"""
        out = {}
        for _, errs, cls in self.rows:
            d = out.setdefault(cls, {})
            for k_, e in errs.items():
                d[k_] = max(d.get(k_, 0.0), e)
        return out

    def assert_ok(self):
        bad = [(w, {k_: "%.2e" % e for k_, e in errs.items() if e >= TOL}) for w, errs, _ in self.rows if any(e >= TOL for e in errs.values())]
        assert not bad, "kernels beyond %.0e (%s): %s" % (TOL, self.tag, bad[:8])
