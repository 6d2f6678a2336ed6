"""Drop-in for the reference's adversarial.py: PnP-AdaNet's adaptation model (`Full_DRN`, adversarial.py:44-574) and its GAN
`Trainer` (576-946) on MI355X HIP kernels.

Graph (same builders, same TF variable names):
  create_zip_network   MR early layers `group_1..6` (frozen) and the CT domain-adaptation module `adapt_1..6`
  create_second_half   shared higher layers `group_7..10`, `output` (frozen), applied to CT then MR features
  create_classifier    feature-domain critic `cls_scope/cls_*` on PS-flattened features + logits + argmax (32 channels)
  create_mask_critic   mask-domain critic `mask_cls_scope/mask_cls_*` on the raw logits
  _get_cost            WGAN losses, L2 terms (adversarial.py:445-476)
Reference defects that are NOT replicated: `self.predictor`/`self.predicter` typo (adversarial.py:101-102);
replicated on purpose: `joint_weights` is never filled (joint_reg == 0), the critics always run dropout .75 and batch statistics
(Python default args at adversarial.py:320,402), second-half / critic weight lists are appended once per call (2x L2 weight).
"""
import logging
import os
import time

import numpy as np
import torch

from . import kernels as K
from .functional import Conv2dDropFn, CriticInputFn, WganLossFn, fan_out, wgrad_overlap
from .layers import (DR_block, conv2d, conv_bn_relu2d, max_pool2d, residual_block, sharable_weight_variable, weight_variable)
from .lib import _dice_eval, _label_decomp
from .ops import PS
from .parallel import barrier, rank_seed
from .variables import VariableStore

raw_size = [256, 256, 3]
volume_size = [256, 256, 3]
label_size = [256, 256, 1]
CRITIC_KEEP_PROB = 0.75      # Python default argument of create_classifier / create_mask_critic (adversarial.py:320,402)


class Full_DRN(object):
    def __init__(self, channels, n_class, batch_size, cost_kwargs={}, network_config={}, **kwargs):
        self.n_class = n_class
        self.batch_size = batch_size
        self.channels = channels
        self.device = torch.device(kwargs.get("device", "cuda"))
        self.world_size = int(kwargs.get("world_size", 1))
        self.feature_base = int(kwargs.get("feature_base", 16))
        self.network_config = dict(network_config)
        self.mr_front_trainable = self.network_config.get("mr_front_trainable", False)
        self.ct_front_trainable = self.network_config.get("ct_front_trainable", True)
        self.joint_trainable = self.network_config.get("joint_trainable", False)
        self.cls_trainable = self.network_config.get("cls_trainable", True)
        self.m_cls_trainable = self.network_config.get("m_cls_trainable", True)
        self.cost_kwargs = dict(cost_kwargs)
        self._parse_cost(dict(cost_kwargs))

        self.store = VariableStore(self.device, seed=int(kwargs.get("seed", 0)))
        m = torch.empty((batch_size, volume_size[0], volume_size[1], channels), device="meta")
        out = self._graph(m, m, keep_prob=1.0, mr_front_bn=False, joint_bn=False, ct_front_bn=True, record=True)
        assert tuple(out["ct_cls"].shape) == (batch_size, 1) and tuple(out["mr_mask"].shape) == (batch_size, 1)
        self._weight_lists = {k: self._names_of(v) for k, v in self._lists.items()}
        self.store.finalize()
        self._get_variables_by_scope()
        # last fetched values
        self.dis_loss = self.ct_gen_loss = None
        self.ct_logits = self.mr_logits = None

    # ---- helpers ----------------------------------------------------------------------------------------------------
    def _names_of(self, tensors):
        by_ptr = {id(v.tensor): v.name for v in self.store.vars.values()}
        return [by_ptr[id(t)] for t in tensors]

    def _parse_cost(self, ck):
        """adversarial.py:447-449, 462, 467"""
        self.miu_dis = float(ck["miu_dis"]) if "miu_dis" in ck else 0.002
        self.miu_gen = float(ck["miu_gen"]) if "miu_gen" in ck else 0.002
        lam = ck.pop("lambda_mask_loss", 1.0)
        self.lambda_mask_loss = 1.0 if lam is None else float(lam)
        self.reg_coeff = ck.pop("regularizer", 1.0e-4)
        self.gan_reg_coeff = ck.pop("gan_regularizer", 1.0e-4)

    def _fc(self, x, w):
        """tf.matmul(tf.reshape(x, [-1, D]), w) (adversarial.py:395-397, 438-440) as a 1x1 'convolution' on the MFMA kernel"""
        B = x.shape[0]
        D = w.shape[0]
        if x.is_meta:
            return torch.empty((B, 1), device="meta")
        g = K.conv_geom((B, 1, 1, D), (1, 1, D, 1), 1, 1, "VALID")
        y = Conv2dDropFn.apply(x.reshape(B, 1, 1, D), w.view(1, 1, D, 1), g, 1.0, 0, 0, torch.is_grad_enabled())
        return y.reshape(B, 1)

    # ---- adversarial.py:127-271 ------------------------------------------------------------------------------------------
    def create_zip_network(self, mr, ct, main_bn, main_trainable, adapt_bn, adapt_trainable, num_cls, feature_base=16, input_channel=3,
                           keep_prob=0.75):
        """MR early layers and the CT adaptation module.  Pass mr=None / ct=None to skip a branch (what TF's graph pruning
        does when only one of them is fetched)."""
        st = self.store
        fb = feature_base
        res = {}
        # (scope, bn-scope prefix, variable factory) for the two branches: identical topology, separate variables
        for branch, x in (("mr", mr), ("ct", ct)):
            if x is None:
                continue
            is_mr = branch == "mr"
            grp = "group_%d" if is_mr else "adapt_%d"
            trainable = main_trainable if is_mr else adapt_trainable
            bn = main_bn if is_mr else adapt_bn
            wl = self._lists["mr_front_weights" if is_mr else "ct_front_weights"]
            cnt = [0]

            def wv(shape):
                if is_mr:
                    return weight_variable(shape=shape, trainable=trainable)
                name = "Variable" if cnt[0] == 0 else "Variable_%d" % cnt[0]
                cnt[0] += 1
                return sharable_weight_variable(shape=shape, trainable=trainable, name=name)

            def bns(k, j=None):
                # MR: 'pred_k_j' ; CT: 'adapt_k' for single-block groups, 'adapt_k_j' otherwise (adversarial.py:137..262)
                if is_mr:
                    return "pred_%d_%d" % (k, j)
                return "adapt_%d" % k if j is None else "adapt_%d_%d" % (k, j)

            with st.variable_scope(grp % 1):
                cnt[0] = 0
                w1_1 = wv([3, 3, input_channel, fb])
                conv1_1 = conv2d(x, w1_1, keep_prob)
                wr1_1, wr1_2 = wv([3, 3, fb, fb]), wv([3, 3, fb, fb])
                block1_1 = residual_block(conv1_1, wr1_1, wr1_2, keep_prob, is_train=bn, leak=True, bn_trainable=trainable,
                                          scope=bns(1, 1) if is_mr else bns(1))
                out1 = max_pool2d(block1_1, n=2)
                wl += [w1_1, wr1_1, wr1_2]
            with st.variable_scope(grp % 2):
                cnt[0] = 0
                wr2_1, wr2_2 = wv([3, 3, fb, fb * 2]), wv([3, 3, fb * 2, fb * 2])
                block2_1 = residual_block(out1, wr2_1, wr2_2, inc_dim=True, keep_prob=keep_prob, leak=True, is_train=bn,
                                          bn_trainable=trainable, scope=bns(2, 1) if is_mr else bns(2))
                out2 = max_pool2d(block2_1, n=2)
                wl += [wr2_1, wr2_2]
            h = out2
            chans = {3: (fb * 2, fb * 4, True), 4: (fb * 4, fb * 8, False), 5: (fb * 8, fb * 16, False), 6: (fb * 16, fb * 16, False)}
            for k in (3, 4, 5, 6):
                cin, cout, pool = chans[k]
                with st.variable_scope(grp % k):
                    cnt[0] = 0
                    wa, wb = wv([3, 3, cin, cout]), wv([3, 3, cout, cout])
                    b1 = residual_block(h, wa, wb, keep_prob, inc_dim=(cin != cout), is_train=bn, leak=True, bn_trainable=trainable,
                                        scope=bns(k, 1))
                    wc, wd = wv([3, 3, cout, cout]), wv([3, 3, cout, cout])
                    b2 = residual_block(b1, wc, wd, keep_prob=keep_prob, is_train=bn, leak=True, bn_trainable=trainable, scope=bns(k, 2))
                    wl += [wa, wb, wc, wd]
                    h = max_pool2d(b2, n=2) if pool else b2
                if k == 4:          # conv4_2 feeds group 5 AND the feature critic
                    h, res[branch + "_c4"] = fan_out(b2)
            res[branch + "_c6"] = h
        return res

    # ---- adversarial.py:273-318 ------------------------------------------------------------------------------------------
    def create_second_half(self, input_feature, joint_bn, joint_trainable, num_cls, feature_base=16, input_channel=3, keep_prob=0.75):
        st = self.store
        fb = feature_base
        sw = lambda shape, name: sharable_weight_variable(shape=shape, trainable=joint_trainable, name=name)
        wl = self._lists["mr_front_weights"]     # sic: the reference appends the shared weights to mr_front_weights, once per call
        with st.variable_scope('group_7'):
            wr7_1, wr7_2 = sw([3, 3, fb * 16, fb * 32], "Variable"), sw([3, 3, fb * 32, fb * 32], "Variable_1")
            block7_1 = residual_block(input_feature, wr7_1, wr7_2, keep_prob=keep_prob, leak=True, inc_dim=True, is_train=joint_bn,
                                      bn_trainable=joint_trainable, scope='pred_7_1')
            wr7_3, wr7_4 = sw([3, 3, fb * 32, fb * 32], "Variable_2"), sw([3, 3, fb * 32, fb * 32], "Variable_3")
            block7_2 = residual_block(block7_1, wr7_3, wr7_4, keep_prob=keep_prob, leak=True, is_train=joint_bn,
                                      bn_trainable=joint_trainable, scope='pred_7_2')
            b7_seg, block7_2 = fan_out(block7_2)        # feeds group 8 AND the feature critic (gradients summed by pnp_add)
            wl += [wr7_1, wr7_2, wr7_3, wr7_4]
        with st.variable_scope('group_8'):
            wr8_1, wr8_2 = sw([3, 3, fb * 32, fb * 32], "Variable"), sw([3, 3, fb * 32, fb * 32], "Variable_1")
            block8_1 = DR_block(b7_seg, wr8_1, wr8_2, keep_prob=keep_prob, leak=True, is_train=joint_bn, rate=2,
                                bn_trainable=joint_trainable, scope='pred_8_1')
            wr8_3, wr8_4 = sw([3, 3, fb * 32, fb * 32], "Variable_2"), sw([3, 3, fb * 32, fb * 32], "Variable_3")
            block8_2 = DR_block(block8_1, wr8_3, wr8_4, keep_prob=keep_prob, leak=True, is_train=joint_bn, rate=2,
                                bn_trainable=joint_trainable, scope='pred_8_2')
            wl += [wr8_1, wr8_2, wr8_3, wr8_4]
        with st.variable_scope('group_9'):
            w9_1 = sw([3, 3, fb * 32, fb * 32], "Variable")
            conv9_1 = conv_bn_relu2d(block8_2, w9_1, keep_prob, leak=True, is_train=joint_bn, bn_trainable=joint_trainable, scope='pred_9_1')
            w9_2 = sw([3, 3, fb * 32, fb * 32], "Variable_1")
            conv9_2 = conv_bn_relu2d(conv9_1, w9_2, keep_prob, leak=True, is_train=joint_bn, bn_trainable=joint_trainable, scope='pred_9_2')
            c9_seg, conv9_2 = fan_out(conv9_2)          # feeds group 10 AND the feature critic
            wl += [w9_1, w9_2]
        with st.variable_scope('group_10'):
            w10_1 = sw([3, 3, fb * 32, 8 * 8 * num_cls * 8], "Variable")
            conv10_1 = conv2d(c9_seg, w10_1, keep_prob_=keep_prob, padding='SYMMETRIC')
            wl.append(w10_1)
            flat_conv10_1 = PS(conv10_1, r=8, n_channel=num_cls * 8, batch_size=self.batch_size)
        with st.variable_scope('output'):
            w11_1 = sw([5, 5, num_cls * 8, num_cls], "Variable")
            logits = conv2d(flat_conv10_1, w11_1, keep_prob_=1., padding='SYMMETRIC')
        return conv9_2, block8_2, block7_2, logits

    # ---- adversarial.py:320-400 ------------------------------------------------------------------------------------------
    def create_classifier(self, input_conv4, input_conv6, input_b7, input_conv9, seg_logits, feature_base=16, keep_prob=CRITIC_KEEP_PROB,
                          cls_bn=True, cls_trainable=True):
        st = self.store
        fb = feature_base
        sw = lambda shape, name: sharable_weight_variable(shape=shape, trainable=cls_trainable, name=name)
        wl = self._lists["cls_weights"]
        with st.variable_scope('cls_0'):
            f4 = PS(input_conv4, r=8, n_channel=2, batch_size=self.batch_size)
            f6 = PS(input_conv6, r=8, n_channel=4, batch_size=self.batch_size)
            f7 = PS(input_b7, r=8, n_channel=8, batch_size=self.batch_size)
            f9 = PS(input_conv9, r=8, n_channel=8, batch_size=self.batch_size)
            # tile(f4, 3) | f6 | f7 | f9 | logits | float(argmax logits): one fused kernel instead of 5 concats (adversarial.py:326-335)
            if seg_logits.is_meta:
                input_comp = torch.empty(tuple(seg_logits.shape[:3]) + (2 * 3 + 4 + 8 + 8 + seg_logits.shape[3] + 1,), device="meta")
            else:
                input_comp = CriticInputFn.apply(f4, f6, f7, f9, seg_logits, 3)
        spec = [(1, fb * 2, fb * 4, 3, 2, True), (2, fb * 4, fb * 8, 5, 2, True), (3, fb * 8, fb * 16, 3, 2, True),
                (4, fb * 16, fb * 32, 3, 2, True), (5, fb * 32, fb * 32, 5, 4, False)]
        h = input_comp
        for k, cin, cout, kd, sd, inc in spec:
            with st.variable_scope('cls_%d' % k):
                wa, wb = sw([3, 3, cin, cout], "Variable"), sw([3, 3, cout, cout], "Variable_1")
                blk = residual_block(h, wa, wb, keep_prob=keep_prob, inc_dim=inc, is_train=cls_bn, bn_trainable=cls_trainable,
                                     scope='cls_%d' % k, leak=True)
                wd = sw([kd, kd, cout, cout], "Variable_2")
                h = conv_bn_relu2d(blk, wd, keep_prob, strides=[1, sd, sd, 1], is_train=cls_bn, bn_trainable=cls_trainable,
                                   scope='cls_%d_3' % k, leak=True)
                wl += [wa, wb, wd]
        with st.variable_scope('cls_6'):
            wr6_1c = sw([3, 3, fb * 32, fb * 32], "Variable")
            conv_6c = conv_bn_relu2d(h, wr6_1c, strides=[1, 2, 2, 1], keep_prob=keep_prob, padding="SYMMETRIC", scope='cls_6',
                                     is_train=cls_bn, bn_trainable=cls_trainable, leak=True)
            wl.append(wr6_1c)
        with st.variable_scope('cls_out'):
            wc_out = sw([fb * 32 * 4, 1], "Variable")
            cls_logits = self._fc(conv_6c, wc_out)
            wl.append(wc_out)
        return cls_logits

    # ---- adversarial.py:402-443 ------------------------------------------------------------------------------------------
    def create_mask_critic(self, input_mask, feature_base=16, keep_prob=CRITIC_KEEP_PROB, num_cls=5, m_cls_bn=True, m_cls_trainable=True):
        st = self.store
        fb = feature_base
        sw = lambda shape, name: sharable_weight_variable(shape=shape, trainable=m_cls_trainable, name=name)
        wl = self._lists["m_cls_weights"]
        with st.variable_scope('mask_cls_1'):
            wr1_1m = sw([3, 3, num_cls, fb], "Variable")
            out1m = conv_bn_relu2d(input_mask, wr1_1m, keep_prob, strides=[1, 2, 2, 1], is_train=m_cls_bn, bn_trainable=m_cls_trainable,
                                   scope='mask_cls_1', leak=True)
            wl.append(wr1_1m)
        with st.variable_scope('mask_cls_2'):
            wa, wb = sw([3, 3, fb, fb], "Variable"), sw([3, 3, fb, fb], "Variable_1")
            blk = residual_block(out1m, wa, wb, keep_prob=keep_prob, inc_dim=False, is_train=m_cls_bn, bn_trainable=m_cls_trainable,
                                 scope='m_cls_2', leak=True)
            wd = sw([5, 5, fb, fb * 2], "Variable_2")
            out2m = conv_bn_relu2d(blk, wd, keep_prob, strides=[1, 4, 4, 1], is_train=m_cls_bn, bn_trainable=m_cls_trainable,
                                   scope='m_cls_2_3', leak=True)
            wl += [wa, wb, wd]
        with st.variable_scope('mask_cls_3'):
            wa, wb = sw([3, 3, fb * 2, fb * 4], "Variable"), sw([3, 3, fb * 4, fb * 4], "Variable_1")
            blk = residual_block(out2m, wa, wb, keep_prob=keep_prob, inc_dim=True, is_train=m_cls_bn, bn_trainable=m_cls_trainable,
                                 scope='m_cls_3', leak=True)
            wd = sw([5, 5, fb * 4, fb * 8], "Variable_2")
            out3m = conv_bn_relu2d(blk, wd, keep_prob, strides=[1, 4, 4, 1], is_train=m_cls_bn, bn_trainable=m_cls_trainable,
                                   scope='m_cls_3_3', leak=True)
            wl += [wa, wb, wd]
        with st.variable_scope('mask_cls_4'):
            wr4_1m = sw([5, 5, fb * 8, fb * 16], "Variable")
            conv_4m = conv_bn_relu2d(out3m, wr4_1m, strides=[1, 4, 4, 1], keep_prob=keep_prob, padding="SYMMETRIC", scope='m_cls_4',
                                     is_train=m_cls_bn, bn_trainable=m_cls_trainable, leak=True)
            wl.append(wr4_1m)
        with st.variable_scope('m_cls_out'):
            wm_out = sw([fb * 16 * 4, 1], "Variable")
            m_cls_logits = self._fc(conv_4m, wm_out)
            wl.append(wm_out)
        return m_cls_logits

    # ---- the graph of adversarial.py:82-119, executed eagerly ---------------------------------------------------------------
    def _graph(self, mr, ct, keep_prob, mr_front_bn, joint_bn, ct_front_bn, record=False, segmenter_no_grad=False, drop_seed=0, critics=True, critic_keep=None):
        """mr / ct: [B,256,256,3] (either may be None: pruned branch).  Returns dict of critic logits and segmenter logits."""
        st = self.store
        ckw = {} if critic_keep is None else {"keep_prob": critic_keep}     # None: the builders' own default (0.75, adversarial.py:320,402)
        self._lists = {"mr_front_weights": [], "ct_front_weights": [], "cls_weights": [], "m_cls_weights": [], "joint_weights": []}
        nc = self.n_class
        out = {}
        with st.as_default():
            st.begin_trace(drop_seed)
            seg_ctx = torch.no_grad() if segmenter_no_grad else _Null()
            with seg_ctx:
                z = self.create_zip_network(mr, ct, main_bn=mr_front_bn, main_trainable=self.mr_front_trainable, adapt_bn=ct_front_bn,
                                            adapt_trainable=self.ct_front_trainable, num_cls=nc, feature_base=self.feature_base,
                                            input_channel=self.channels, keep_prob=keep_prob)
                feats = {}
                for br in ("ct", "mr"):          # CT first, then MR (adversarial.py:91-92)
                    if br + "_c6" in z:
                        # conv6_2 feeds the shared half AND the feature critic: two autograd edges, gradients summed by pnp_add (fan_out)
                        c6_seg, z[br + "_c6"] = fan_out(z[br + "_c6"]) if critics else (z[br + "_c6"], z[br + "_c6"])
                        feats[br] = self.create_second_half(c6_seg, feature_base=self.feature_base, input_channel=3, num_cls=nc,
                                                            keep_prob=keep_prob, joint_bn=joint_bn, joint_trainable=self.joint_trainable)
            for br in feats:
                out[br + "_logits"] = feats[br][3]
            if not critics:         # fetches that do not reach the critics (test_eval): TF prunes them, incl. their BN moving-average updates
                return out
            with st.variable_scope("cls_scope"):
                for br in ("ct", "mr"):
                    if br in feats:
                        c9, b8, b7, lg = feats[br]
                        lg_cls, lg_mask = fan_out(lg)          # the logits feed both critics
                        feats[br] = (c9, b8, b7, lg_mask)
                        out[br + "_cls"] = self.create_classifier(z[br + "_c4"], z[br + "_c6"], b7, c9, lg_cls, feature_base=self.feature_base,
                                                                  cls_trainable=self.cls_trainable, **ckw)
                        out[br + "_logits"] = lg
            with st.variable_scope("mask_cls_scope"):
                for br in ("ct", "mr"):
                    if br in feats:
                        out[br + "_mask"] = self.create_mask_critic(feats[br][3], feature_base=self.feature_base, num_cls=nc,
                                                                    m_cls_trainable=self.m_cls_trainable, **ckw)
        return out

    # ---- adversarial.py:478-501 ------------------------------------------------------------------------------------------
    def _get_variables_by_scope(self):
        self.adapt_vars, self.cls_vars, self.seg_vars, self.mri_seg_vars = [], [], [], []
        for v in self.store.vars.values():
            if "cls" in v.name:
                self.cls_vars.append(v)
            elif "adapt" in v.name:
                self.adapt_vars.append(v)
            elif "output" in v.name:
                self.seg_vars.append(v)
                self.mri_seg_vars.append(v)
            elif "group" in v.name:
                self.mri_seg_vars.append(v)

    def l2_tables(self, dis_sub_iter, gen_sub_iter):
        """per-chunk L2 coefficients of the two optimisers (adversarial.py:463-474, 644, 650): the weight lists hold every critic
        weight TWICE (the builders run for CT and MR), so the effective coefficient is gan_reg*miu*count/sub_iter."""
        wl = self._weight_lists
        cnt_cls = {n: wl["cls_weights"].count(n) for n in set(wl["cls_weights"])}
        cnt_m = {n: wl["m_cls_weights"].count(n) for n in set(wl["m_cls_weights"])}
        cnt_ct = {n: wl["ct_front_weights"].count(n) for n in set(wl["ct_front_weights"])}
        gd = self.gan_reg_coeff * self.miu_dis / float(dis_sub_iter)
        gg = self.gan_reg_coeff * self.miu_gen / float(gen_sub_iter)
        dis = self.store.chunk_table(lambda v: gd * (cnt_cls.get(v.name, 0) + self.lambda_mask_loss * cnt_m.get(v.name, 0)), np.float32)
        gen = self.store.chunk_table(lambda v: gg * cnt_ct.get(v.name, 0), np.float32)
        return dis, gen

    def _activate(self, group):
        """only `group` ('cls' / 'adapt') receives gradients this step == minimize(..., var_list=...) (adversarial.py:643-652)"""
        for v in self.store.trainable():
            want = (group == "cls" and "cls" in v.name) or (group == "adapt" and "adapt" in v.name and "cls" not in v.name)
            v.tensor.requires_grad_(want)

    def dis_loss_and_grads(self, mr, ct, keep_prob, drop_seed=0):
        """discriminator step graph (adversarial.py:852-859): segmenter BN all in inference mode, critics batch-stat; backward through
        the two critics only (their inputs are constants for this step)."""
        self._activate("cls")
        self.store.zero_grad()
        o = self._graph(mr, ct, keep_prob, mr_front_bn=False, joint_bn=False, ct_front_bn=False, segmenter_no_grad=True, drop_seed=drop_seed)
        lam, mu = self.lambda_mask_loss, self.miu_dis
        use_mask = lam != 0.0
        loss = WganLossFn.apply(o["ct_cls"], o["mr_cls"], o["ct_mask"] if use_mask else None, o["mr_mask"] if use_mask else None,
                                (mu, -mu, lam * mu, -lam * mu), 1.0 / self.world_size)
        with wgrad_overlap():
            loss.backward(self.store.unit_grad(1))
        self.dis_loss = loss.detach()
        self.ct_logits, self.mr_logits = o["ct_logits"], o["mr_logits"]
        self.critic_scores = {k: o[k].detach() for k in ("ct_cls", "mr_cls", "ct_mask", "mr_mask") if o.get(k) is not None}
        return self.dis_loss

    def gen_loss_and_grads(self, ct, keep_prob, drop_seed=0):
        """generator step graph (adversarial.py:875-881): CT front in BN-training mode, gradients to `adapt_*` only."""
        self._activate("adapt")
        self.store.zero_grad()
        o = self._graph(None, ct, keep_prob, mr_front_bn=False, joint_bn=False, ct_front_bn=True, drop_seed=drop_seed)
        lam, mu = self.lambda_mask_loss, self.miu_gen
        use_mask = lam != 0.0
        loss = WganLossFn.apply(o["ct_cls"], None, o["ct_mask"] if use_mask else None, None, (-mu, 0.0, -lam * mu, 0.0), 1.0 / self.world_size)
        with wgrad_overlap():
            loss.backward(self.store.unit_grad(1))
        self.ct_gen_loss = loss.detach()
        self.ct_logits = o["ct_logits"]
        self.critic_scores = {k: o[k].detach() for k in ("ct_cls", "ct_mask") if o.get(k) is not None}
        return self.ct_gen_loss

    def evaluate(self, ct, ct_y, mr, mr_y, keep_prob=1.0, detail=False):
        """monitoring fetches that are not TensorBoard summaries (adversarial.py:101-116, 948-991): CT / MR hard Dice of the current
        segmenter, with `detail` the CT confusion matrix.  Every segmenter BN in inference mode; the critics are not evaluated (nothing
        here consumes their scores), so the forward has no side effect on any variable."""
        from . import lib
        with torch.no_grad():
            o = self._graph(mr, ct, keep_prob, mr_front_bn=False, joint_bn=False, ct_front_bn=False, critics=False)
            self.predicter, self.compact_pred = K.softmax_argmax(o["ct_logits"].contiguous())
            self.mr_seg_valid, self.compact_mr_valid = K.softmax_argmax(o["mr_logits"].contiguous())
            self.ct_dice_eval, self.ct_dice_eval_arr = _dice_eval(self.compact_pred, ct_y, self.n_class)
            self.mr_dice_eval, self.mr_dice_eval_arr = _dice_eval(self.compact_mr_valid, mr_y, self.n_class)
            if detail:
                self.compact_y, self.confusion_matrix = lib.compact_and_confusion(ct_y, self.compact_pred)
        return float(self.ct_dice_eval), float(self.mr_dice_eval)

    def predict_ct(self, ct, ct_y):
        """sess.run([compact_pred, confusion_matrix], {ct, ct_y, keep_prob: 1, *_bn: False}) of adversarial.py:1035-1037"""
        from . import lib
        with torch.no_grad():
            o = self._graph(None, ct, 1.0, mr_front_bn=False, joint_bn=False, ct_front_bn=False, critics=False)
            self.predicter, self.compact_pred = K.softmax_argmax(o["ct_logits"].contiguous())
            self.compact_y, self.confusion_matrix = lib.compact_and_confusion(ct_y, self.compact_pred)
        return self.compact_pred, self.confusion_matrix

    # ---- checkpoints / phase hand-off (own .npz format keyed by the TF names) -------------------------------------
    def save(self, path):
        from .lib import atomic_savez
        return atomic_savez(path, **{k.replace("/", "|"): v for k, v in self.store.state_dict().items()})

    def restore(self, sess_or_none, model_path, no_gan=False, clear_rms=False):
        """adversarial.py:503-574: name-matched restore.  no_gan: only the main variables — names containing 'group' or 'output' and
        neither 'adapt' nor 'cls' (the baseline's BatchNorm_k statistics arrive through load_baseline).  clear_rms concerns the optimiser
        slots, which live in the Trainer here (Trainer.restore_optimizer)."""
        with np.load(model_path) as z:
            sd = {k.replace("|", "/"): z[k] for k in z.files}
        if no_gan:
            sd = {k: v for k, v in sd.items() if ("group" in k or "output" in k) and "adapt" not in k and "cls" not in k}
        self.store.load_state_dict({k: v for k, v in sd.items() if k in self.store.vars}, strict=False)

    def load_baseline(self, segmenter_state, old_bn_list=None, new_bn_list=None, adapt_var_list=None, mr_var_list=None):
        """Phase hand-off of train_gan.py --phase pre-train (adversarial.py:706-765): conv weights of the source segmenter by name,
        its BatchNorm_k variables onto the pred_* scopes (old_bn_list -> new_bn_list, positional), then MR early layers copied onto
        the CT adaptation module (mr_var_list -> adapt_var_list, positional)."""
        st = self.store
        strip = lambda n: n.split(":")[0]
        sd = {}
        for k, v in segmenter_state.items():
            if "/Variable" in k and k in st.vars:
                sd[k] = v
        if old_bn_list is None:
            old_bn_list = [k for k in segmenter_state if k.startswith("BatchNorm")]
        if new_bn_list is None:
            new_bn_list = [k for k in st.vars if "/pred_" in k]
        by_suffix = {}
        for k in st.vars:
            by_suffix.setdefault(k.split("/", 1)[-1] if k.startswith("group_") else k, k)
        for o, n in zip(old_bn_list, new_bn_list):
            o, n = strip(o), strip(n)
            tgt = n if n in st.vars else by_suffix.get(n)
            if tgt is None or o not in segmenter_state:
                raise KeyError("cannot map BN variable %s -> %s" % (o, n))
            sd[tgt] = segmenter_state[o]
        st.load_state_dict(sd, strict=False)
        if mr_var_list is None:
            mr_var_list = [k for k in st.vars if k.startswith("group_") and int(k.split("/")[0].split("_")[1]) <= 6]
            adapt_var_list = [k for k in st.vars if k.startswith("adapt_")]
        cur = st.state_dict()
        st.load_state_dict({strip(a): cur[strip(m)] for m, a in zip(mr_var_list, adapt_var_list)}, strict=False)


contour_map = {"bg": 0, "la_myo": 1, "la_blood": 2, "lv_blood": 3, "aa": 4}     # adversarial.py:19-25
verbose = True


def _host(t):
    return None if t is None else float(t)


class _Null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class RMSPropOptimizer(object):
    """tf.train.RMSPropOptimizer(lr, decay=.9, momentum=0, epsilon=1e-10).minimize(loss, var_list=...) over the chunks of the flat arena
    selected by `mask` (adversarial.py:643-652); the `rms` slot starts at ONE like TF's."""

    def __init__(self, store, learning_rate, l2_table, mask, decay=0.9, epsilon=1e-10):
        self.store, self.lr, self.decay, self.eps = store, float(learning_rate), decay, epsilon
        self.ms = torch.ones_like(store.arena)
        self.l2, self.mask = l2_table, mask
        self._mask_host = mask.detach().cpu().numpy() if mask is not None else None
        self._ranges = store.written_ranges(self._mask_host) if store.arena.is_cuda else None      # what a step writes (kernels.weights_changed)

    def step(self):
        K.rmsprop_step(self.store.arena, self.store.grad_arena, self.ms, self.l2, self.mask, self.lr, self.decay, self.eps, ranges=self._ranges)

    # the slot and the learning rate are TF variables in the reference, i.e. part of every checkpoint (tf.train.Saver)
    def _mine(self, v):
        """variables of this optimiser's var_list (the chunks its mask selects)"""
        from ._lib import OPT_CHUNK
        return bool(self._mask_host[v.offset // OPT_CHUNK]) if self._mask_host is not None else True

    def state_dict(self):
        """'<variable>|RMSProp' per variable of the var_list — keyed by NAME: the arena layout changes between the pre-train graph
        (adapt_* frozen) and the train-gan graph, the names do not"""
        d = {"lr": np.float64(self.lr)}
        d.update(self.store.slots_to_dict(self.ms, "RMSProp", self._mine))
        return d

    def load_state_dict(self, sd, slots=True, lr=True):
        """-> (restored, missing) variable names of the var_list"""
        done, missing = ([], [])
        if slots:
            done, missing = self.store.slots_from_dict(self.ms, sd, "RMSProp", self._mine)
        if lr and "lr" in sd:
            self.lr = float(sd["lr"])
        return done, missing


class Trainer(object):
    """Train the adaptation model (adversarial.py:576-946).  Data sources: lists of .tfrecords files or objects with
    `next_batch(B) -> ([B,256,256,4] float32, ids)` (image channels 0:3, label map in channel 3)."""

    def __init__(self, net, mr_train_list, mr_val_list, ct_train_list, ct_val_list, adapt_var_list=None, mr_var_list=None, old_bn_list=None,
                 new_bn_list=None, test_label_list=None, test_nii_list=None, num_cls=None, batch_size=6, opt_kwargs={}, train_config={},
                 reducer=None, shard=None):
        self.net = net
        self.shard = shard              # (rank, world_size) under data parallelism
        self.rank = shard[0] if shard else 0
        self.test_label_list, self.test_nii_list = test_label_list, test_nii_list
        self.batch_size = batch_size
        self.num_cls = num_cls
        self.opt_kwargs = dict(opt_kwargs)
        self.train_config = dict(train_config)
        self.ct_train_list, self.ct_val_list = ct_train_list, ct_val_list
        self.mr_train_list, self.mr_val_list = mr_train_list, mr_val_list
        self.adapt_var_list, self.mr_var_list = adapt_var_list, mr_var_list
        self.old_bn_list, self.new_bn_list = old_bn_list, new_bn_list
        self.reducer = reducer
        self.dis_optimizer = self.gen_optimizer = None
        self.clip_mask = None
        self.global_step = 0
        self.step_times = []
        self.loss_dict = {}

    def next_batch(self, source, capacity=120, num_threads=2, min_after_dequeue=30):
        """adversarial.py:607-631 (shuffle_batch: 2 reader threads, capacity 120, min_after_dequeue 30)"""
        from .tfrecord import SliceQueue
        if hasattr(source, "next_batch"):
            return source
        return SliceQueue(source, self.batch_size, capacity=capacity, min_after_dequeue=min_after_dequeue, num_threads=num_threads,
                          shard=self.shard)

    def _feeder(self, source):
        from .feeder import DeviceFeeder
        return DeviceFeeder(self.next_batch(source), self.batch_size, self.num_cls, self.net.device)

    def _get_optimizer(self):
        """adversarial.py:633-656"""
        net, st = self.net, self.net.store
        lr = self.opt_kwargs.pop("learning_rate", None)
        self.LR_refresh = lr
        dsi = self.train_config.get('dis_sub_iter', 1) or 1
        gsi = self.train_config.get('gen_sub_iter', 1) or 1
        l2_dis, l2_gen = net.l2_tables(dsi, gsi)
        m_dis = st.chunk_table(lambda v: 1 if "cls" in v.name else 0, np.uint8)
        m_gen = st.chunk_table(lambda v: 1 if ("adapt" in v.name and "cls" not in v.name) else 0, np.uint8)
        self.dis_optimizer = RMSPropOptimizer(st, lr, l2_dis, m_dis, **self.opt_kwargs)
        self.gen_optimizer = RMSPropOptimizer(st, lr, l2_gen, m_gen, **self.opt_kwargs)
        # clip_op: every cls variable whose name contains "Variable" (conv / FC weights, not BN) to [-0.03, 0.03]
        self.clip_mask = st.chunk_table(lambda v: 1 if ("cls" in v.name and "Variable" in v.name) else 0, np.uint8)
        self._clip_ranges = st.written_ranges(self.clip_mask.cpu().numpy()) if st.arena.is_cuda else None
        return self.dis_optimizer, self.gen_optimizer

    def save_checkpoint(self, output_path):
        """lib._save (tf.train.Saver over ALL variables, lib.py:23-29): the model variables plus the RMSProp slots, the two learning
        rates and the global step"""
        from .lib import atomic_savez
        ck = os.path.join(output_path, "checkpoint.npz")
        self.net.save(ck)
        d, g = self.dis_optimizer.state_dict(), self.gen_optimizer.state_dict()
        slots = {k: v for src in (d, g) for k, v in src.items() if k != "lr"}       # the two var_lists are disjoint (cls* / adapt*)
        atomic_savez(os.path.join(output_path, "optimizer.npz"), kind="rmsprop", dis_lr=d["lr"], gen_lr=g["lr"],
                     global_step=np.int64(self.global_step), **slots)
        return ck

    def restore_optimizer(self, restored_path, clear_rms, lr_update):
        """adversarial.py:503-574, 803-805: RMSProp slots come back unless clear_rms, matched BY VARIABLE NAME like tf.train.Saver
        (so the critic slots warmed up by --phase pre-train survive into --phase train-gan, whose graph also trains adapt_* and
        therefore lays its arena out differently); the learning rates come back unless lr_update (LR_refresh replaces them)."""
        f = os.path.join(restored_path, "optimizer.npz")
        if not os.path.exists(f):
            return False
        with np.load(f) as z:
            if "kind" not in z.files or str(z["kind"]) != "rmsprop":
                logging.warning("optimizer state in %s is not RMSProp state of the adaptation graph: slots start fresh" % f)
                return False                                   # a checkpoint of another graph (e.g. the source segmenter's)
            sd = {k: z[k] for k in z.files}
            dd, dm = self.dis_optimizer.load_state_dict(dict(sd, lr=sd["dis_lr"]), slots=not clear_rms, lr=not lr_update)
            gd, gm = self.gen_optimizer.load_state_dict(dict(sd, lr=sd["gen_lr"]), slots=not clear_rms, lr=not lr_update)
            self.global_step = int(z["global_step"])
        if not clear_rms:
            if not (dd or gd):
                raise RuntimeError("restore with clear_rms=False asked for the RMSProp slots, but %s holds none for this graph" % f)
            if dm or gm:
                logging.warning("RMSProp slots not in %s for %d variables (e.g. %s): they start at 1.0 like a fresh tf slot" % (
                    f, len(dm) + len(gm), (dm + gm)[0]))
        return True

    def capture_steps(self, mr_batch, ct_batch, dropout):
        """Step capture (step_capture.py): from now on dis_step / gen_step on batches of these shapes and this keep probability are ONE
        hipGraph launch each.  The reference's counterpart is the single sess.run per step (adversarial.py:852-881).  The two warm-up
        steps per graph are real updates.  Not under data parallelism."""
        if self.reducer is not None:
            raise RuntimeError("capture_steps: not under data parallelism (the bucketed all-reduce runs on a side stream)")
        from .step_capture import CapturedStep
        if self.dis_optimizer is None:
            self._get_optimizer()
        self._cap = None
        # by-value arguments frozen into the recordings: the RMSProp learning rates (train() decays them, restore_optimizer may replace them)
        cap = {"dropout": float(dropout), "mr": tuple(mr_batch.shape), "ct": tuple(ct_batch.shape),
               "lr": (float(self.dis_optimizer.lr), float(self.gen_optimizer.lr))}
        g0 = self.global_step
        cap["dis"] = CapturedStep(lambda m, c: self.dis_step(m, c, dropout, 0), [mr_batch, ct_batch])
        cap["gen"] = CapturedStep(lambda c: self.gen_step(c, dropout, 0), [ct_batch])
        self.global_step = g0 + 4      # the two recordings counted themselves; the 2 + 2 warm-up steps are real
        self._cap = cap
        return cap

    def _captured(self, which, dropout, shapes):
        cap = getattr(self, "_cap", None)
        if cap is None or cap["dropout"] != float(dropout) or any(cap[k] != tuple(s) for k, s in shapes.items()):
            return None
        if cap["lr"] != (float(self.dis_optimizer.lr), float(self.gen_optimizer.lr)):
            # a replay would apply the learning rate of the recording: run eagerly from here on (capture_steps() again to re-record)
            if not cap.get("warned"):
                cap["warned"] = True
                logging.warning("captured GAN steps were recorded with learning rates %s, now %s: running eagerly (call capture_steps again to re-record)"
                                % (cap["lr"], (self.dis_optimizer.lr, self.gen_optimizer.lr)))
            return None
        return cap[which]

    def dis_step(self, mr_batch, ct_batch, dropout, seed):
        """sess.run(dis_optimizer) + sess.run(clip_op) (adversarial.py:852-861)"""
        g = self._captured("dis", dropout, {"mr": mr_batch.shape, "ct": ct_batch.shape})
        if g is not None:
            self.global_step += 1
            K.weights_changed()
            return g.replay(seed, mr_batch, ct_batch)
        loss = self.net.dis_loss_and_grads(mr_batch, ct_batch, dropout, drop_seed=seed)
        if self.reducer is not None:
            self.reducer.allreduce()
        self.dis_optimizer.step()
        K.clip(self.net.store.arena, self.clip_mask, -0.03, 0.03, ranges=self._clip_ranges)
        self.global_step += 1
        return loss

    def gen_step(self, ct_batch, dropout, seed):
        """sess.run(gen_optimizer) (adversarial.py:875-881)"""
        g = self._captured("gen", dropout, {"ct": ct_batch.shape})
        if g is not None:
            self.global_step += 1
            K.weights_changed()
            return g.replay(seed, ct_batch)
        loss = self.net.gen_loss_and_grads(ct_batch, dropout, drop_seed=seed)
        if self.reducer is not None:
            self.reducer.allreduce()
        self.gen_optimizer.step()
        self.global_step += 1
        return loss

    def train(self, output_path, restore=True, restored_path=None, training_iters=200, epochs=1000, dropout=0.75, display_step=5):
        """adversarial.py:767-946 (schedule, sub-iteration growth, periodic save + lr decay)"""
        self.output_path = output_path
        os.makedirs(output_path, exist_ok=True)
        save_path = os.path.join(output_path, "checkpoint.npz")      # the reference returns a Saver prefix (model.cpkt); this is the real file
        if epochs == 0:
            return save_path
        if self.dis_optimizer is None:
            self._get_optimizer()
        tc = self.train_config
        if restore and restored_path is not None:
            ck = os.path.join(restored_path, "checkpoint.npz")
            if os.path.exists(ck):
                self.net.restore(None, ck, no_gan=bool(tc.get("restore_from_baseline")), clear_rms=bool(tc.get("clear_rms")))
                if not tc.get("restore_from_baseline"):
                    self.restore_optimizer(restored_path, clear_rms=bool(tc.get("clear_rms")), lr_update=bool(tc.get("lr_update")))
        from .metrics import ScalarLog
        self.scalars = ScalarLog(output_path, self.rank)
        ct_feed, mr_feed = self._feeder(self.ct_train_list), self._feeder(self.mr_train_list)
        ct_val, mr_val = self._feeder(self.ct_val_list), self._feeder(self.mr_val_list)
        dis_interval, gen_interval = tc.get('dis_interval', 1), tc.get('gen_interval', 1)
        dis_sub_iter, gen_sub_iter = tc.get('dis_sub_iter', 1), tc.get('gen_sub_iter', 1)
        dis_inc, gen_inc = tc.get('dis_sub_iter_inc', 0), tc.get('gen_sub_iter_inc', 0)
        upd = tc.get('iter_upd_interval', 999999999999)
        seed = 1 + rank_seed(self.rank)
        try:
            for epoch in range(epochs):
                for step in range(epoch * training_iters, (epoch + 1) * training_iters):
                    start = time.time()
                    n_dis = n_gen = 0
                    if dis_interval != 0 and (step % dis_interval == 0) and step != 0:
                        n_dis = dis_sub_iter
                        for _ in range(dis_sub_iter):
                            ct_x = ct_feed.next()[0]
                            mr_x = mr_feed.next()[0]
                            self.dis_step(mr_x, ct_x, dropout, seed)
                            seed += 1
                    if gen_interval != 0 and (step % gen_interval == 0) and step != 0:
                        n_gen = gen_sub_iter
                        for _ in range(gen_sub_iter):
                            ct_x = ct_feed.next()[0]
                            self.gen_step(ct_x, dropout, seed)
                            seed += 1
                    if (step % upd == 0) and step != 0:
                        dis_sub_iter += dis_inc
                        gen_sub_iter += gen_inc
                    self.step_times.append(time.time() - start)
                    self.scalars.write("gan_step", step=step, epoch=epoch, host_time_s=self.step_times[-1], dis_updates=n_dis, gen_updates=n_gen)
                    logging.info("Training step %s epoch %s has been finished! Time elapsed %s seconds" % (step, epoch, time.time() - start))
                    if step % display_step == 0:
                        self.output_minibatch_stats(step, *ct_feed.next()[:2], *mr_feed.next()[:2])                # a training batch ...
                        self.output_minibatch_stats(step, *ct_val.next()[:2], *mr_val.next()[:2], detail=True)     # ... and a validation batch
                    if step % tc.get("checkpoint_space", 100) == 0 and step != 0:
                        if self.rank == 0:
                            self.save_checkpoint(output_path)
                        f = tc.get('lr_decay_factor', 1.0)
                        self.dis_optimizer.lr *= f
                        self.gen_optimizer.lr *= f
        finally:                 # reader threads, pinned buffers and copy streams go away also when a step raises
            for f in (ct_feed, mr_feed, ct_val, mr_val):
                f.close()
            self.scalars.close()
        if self.rank == 0:
            self.save_checkpoint(output_path)
        barrier()
        return save_path

    def output_minibatch_stats(self, step, ct_batch, ct_batch_y, mr_batch, mr_batch_y, detail=False):
        """adversarial.py:948-991 without the TensorBoard writers: Dice of both domains, confusion matrix + per-organ report on `detail`"""
        from .lib import _indicator_eval
        ct_d, mr_d = self.net.evaluate(ct_batch, ct_batch_y, mr_batch, mr_batch_y, detail=detail)
        self.loss_dict["val" if detail else "train"] = (step, ct_d, mr_d)
        if getattr(self, "scalars", None) is not None:      # evaluate() has just synchronised: the last losses cost nothing to read here
            self.scalars.write("val_eval" if detail else "train_eval", step=step, ct_dice=ct_d, mr_dice=mr_d,
                               dis_loss=_host(getattr(self.net, "dis_loss", None)), gen_loss=_host(getattr(self.net, "ct_gen_loss", None)))
        if detail:
            _indicator_eval(self.net.confusion_matrix, verbose=verbose)

    # -- volume inference -------------------------------------------------------------------------------
    def _predict_batch(self, vol, slice_y):
        from .lib import label_decomp_device
        dev = self.net.device
        y = label_decomp_device(self.num_cls, torch.from_numpy(slice_y).to(dev))
        pred, cm = self.net.predict_ct(torch.from_numpy(vol).to(dev), y)
        return pred.cpu().numpy(), cm

    def test_eval(self, sess=None, output_path=".", flip_correction=True):
        """adversarial.py:993-1052: CT test volumes through adapt_* front + shared second half; writes cm.csv"""
        from . import volume_eval as ve
        os.makedirs(os.path.join(output_path, "dense_pred"), exist_ok=True)
        self.test_pair_list = list(zip(self.test_label_list, self.test_nii_list))
        sample_eval_list, all_cm = ve.test_eval(self._predict_batch, self.test_label_list, self.test_nii_list, self.net.batch_size,
                                                self.num_cls, flip_correction, shuffle=True)
        np.savetxt(os.path.join(output_path, "cm.csv"), all_cm)
        return self.sample_metric_stddev(sample_eval_list)

    def test_model(self, this_model, output_path):
        """adversarial.py:1097-1108: restore a checkpoint (.npz of this package), then test_eval"""
        os.makedirs(output_path, exist_ok=True)
        self.net.restore(None, this_model)
        logging.info("model has been loaded!")
        dice, jac = self.test_eval(None, output_path)
        logging.info("testing finished")
        return dice, jac

    def sample_metric_stddev(self, sample_eval_list):
        """adversarial.py:1054-1084"""
        from . import volume_eval as ve
        return ve.sample_metric_stddev(sample_eval_list, self.num_cls, contour_map)
