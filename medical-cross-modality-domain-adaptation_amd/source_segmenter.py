"""Drop-in for the reference's source_segmenter.py: the source-domain dilated residual segmenter
(`Full_DRN`, source_segmenter.py:48-300) and its `Trainer` (303-570), executing on MI355X HIP kernels.

What changes with respect to TF-1.4 graph mode: the graph-building code of `create_network` is re-executed
eagerly every step against the variables created at construction time (symbolic build pass on `meta` tensors);
`sess.run(optimizer, feed_dict)` becomes `Full_DRN.train_step(x, y, keep_prob)`.  Variable NAMES follow TF
(`group_k/Variable[_j]`, `BatchNorm[_k]/{beta,gamma,moving_mean,moving_variance}`), including the reference's
L2-regulariser quirk (wr4_4 listed twice, wr4_3 never: source_segmenter.py:132-135).
"""
import logging
import os
import shutil
import time

import numpy as np
import torch

from . import kernels as K
from . import lib
from .functional import SegLossFn, sync_now, wgrad_overlap
from .layers import (DR_block, conv2d, conv_bn_relu2d, max_pool2d, pixel_wise_softmax_2, residual_block, weight_variable)
from .lib import _dice_eval, _indicator_eval, _label_decomp
from .ops import PS
from .parallel import barrier, rank_seed
from .variables import VariableStore

raw_size = [256, 256, 3]     # source_segmenter.py:34-36
volume_size = [256, 256, 3]
label_size = [256, 256, 1]

contour_map = {"bg": 0, "lv_myo": 1, "la_blood": 2, "lv_blood": 3, "aa": 4}
verbose = True


class Full_DRN(object):
    """Dilated Residual Network (source_segmenter.py:48-300).

    :param channels:    number of channels in the input image, set as 3
    :param n_class:     number of output labels, set as 5
    :param batch_size:  batch size (kept for signature compatibility; any batch works)
    kwargs: device (default cuda:current), seed (variable init seed), world_size (gradient scale 1/world_size)
    """

    def __init__(self, channels, n_class, batch_size, adapt_module=True, main_trainable=True, adapt_trainable=True,
                 cost_kwargs={}, **kwargs):
        self.n_class = n_class
        self.batch_size = batch_size
        self.channels = channels
        self.summaries = kwargs.get("summaries", True)
        self.device = torch.device(kwargs.get("device", "cuda"))
        self.world_size = int(kwargs.get("world_size", 1))
        self.conv_weights = []
        self.main_trainable = main_trainable
        self.adapt_trainable = adapt_trainable
        self.adapt_module = adapt_module
        self.feature_base = int(kwargs.get("feature_base", 16))

        self.store = VariableStore(self.device, seed=int(kwargs.get("seed", 0)))
        # symbolic build pass == the reference's graph construction: creates every variable, runs no kernel
        xm = torch.empty((batch_size, volume_size[0], volume_size[1], channels), device="meta")
        with self.store.as_default():
            self.store.begin_trace()
            logits = self.create_network(xm, input_size=raw_size, input_channel=channels, num_cls=n_class,
                                         feature_base=self.feature_base, keep_prob=1.0, adapt_module=adapt_module,
                                         main_bn=True, main_trainable=main_trainable, adapt_bn=True,
                                         adapt_trainable=adapt_trainable)
        assert tuple(logits.shape) == (batch_size, label_size[0], label_size[1], n_class), logits.shape
        self._conv_weight_names = self._names_of(self.conv_weights)
        self.store.finalize()
        self._parse_cost(dict(cost_kwargs))
        # L2 multiplicity per variable = how often it appears in conv_weights (source_segmenter.py:237)
        for v in self.store.vars.values():
            v.l2_mult = float(self._conv_weight_names.count(v.name))
        # outputs of the last forward (the tensors a TF user would fetch with sess.run)
        self.logits = self.predicter = self.compact_pred = self.compact_y = None
        self.cost = self.regularizer_loss = self.weighted_loss = self.dice_loss = None
        self.dice_eval = self.dice_eval_arr = None
        self.confusion_matrix = None

    # -- helpers -------------------------------------------------------------------------------------
    def _names_of(self, tensors):
        by_ptr = {}
        for v in self.store.vars.values():
            by_ptr[id(v.tensor)] = v.name
        return [by_ptr[id(t)] for t in tensors]

    def _parse_cost(self, cost_kwargs):
        """source_segmenter.py:217-221"""
        self.dice_flag = cost_kwargs.pop("dice_flag", True)
        self.cross_flag = cost_kwargs.pop("cross_flag", False)
        self.miu_dice = cost_kwargs.pop("miu_dice", None)
        self.miu_cross = cost_kwargs.pop("miu_cross", None)
        self.reg_coeff = cost_kwargs.pop("regularizer", 1e-4)

    def _w(self, name):
        return self.store.vars[name].tensor

    # -- source_segmenter.py:88-209 ---------------------------------------------------------------------
    def create_network(self, x, input_size, input_channel, num_cls, feature_base=16, keep_prob=0.75, main_bn=True,
                       main_trainable=True, adapt_module=True, adapt_bn=True, adapt_trainable=True):
        st = self.store
        fb = feature_base
        cw = []

        def wv(shape, trainable):
            w = weight_variable(shape=shape, trainable=trainable)
            return w

        with st.name_scope('group_1'):
            w1_1 = wv([3, 3, input_channel, fb], adapt_trainable)
            conv1_1 = conv2d(x, w1_1, keep_prob)
            wr1_1 = wv([3, 3, fb, fb], adapt_trainable)
            wr1_2 = wv([3, 3, fb, fb], adapt_trainable)
            block1_1 = residual_block(conv1_1, wr1_1, wr1_2, keep_prob, is_train=adapt_bn, leak=True, bn_trainable=adapt_trainable)
            out1 = max_pool2d(block1_1, n=2)
            cw += [w1_1, wr1_1, wr1_2]

        with st.name_scope('group_2'):
            wr2_1 = wv([3, 3, fb, fb * 2], adapt_trainable)
            wr2_2 = wv([3, 3, fb * 2, fb * 2], adapt_trainable)
            block2_1 = residual_block(out1, wr2_1, wr2_2, inc_dim=True, leak=True, keep_prob=keep_prob, is_train=adapt_bn,
                                      bn_trainable=adapt_trainable)
            out2 = max_pool2d(block2_1, n=2)
            cw += [wr2_1, wr2_2]

        with st.name_scope('group_3'):
            wr3_1 = wv([3, 3, fb * 2, fb * 4], adapt_trainable)
            wr3_2 = wv([3, 3, fb * 4, fb * 4], adapt_trainable)
            block3_1 = residual_block(out2, wr3_1, wr3_2, keep_prob, inc_dim=True, leak=True, is_train=adapt_bn,
                                      bn_trainable=adapt_trainable)
            wr3_3 = wv([3, 3, fb * 4, fb * 4], adapt_trainable)
            wr3_4 = wv([3, 3, fb * 4, fb * 4], adapt_trainable)
            block3_2 = residual_block(block3_1, wr3_3, wr3_4, keep_prob=keep_prob, leak=True, is_train=adapt_bn,
                                      bn_trainable=adapt_trainable)
            out3 = max_pool2d(block3_2, n=2)
            cw += [wr3_1, wr3_2, wr3_3, wr3_4]

        with st.name_scope('group_4'):
            wr4_1 = wv([3, 3, fb * 4, fb * 8], adapt_trainable)
            wr4_2 = wv([3, 3, fb * 8, fb * 8], adapt_trainable)
            block4_1 = residual_block(out3, wr4_1, wr4_2, keep_prob, inc_dim=True, leak=True, is_train=adapt_bn,
                                      bn_trainable=adapt_trainable)
            wr4_3 = wv([3, 3, fb * 8, fb * 8], adapt_trainable)
            wr4_4 = wv([3, 3, fb * 8, fb * 8], adapt_trainable)
            block4_2 = residual_block(block4_1, wr4_3, wr4_4, keep_prob, is_train=adapt_bn, leak=True,
                                      bn_trainable=adapt_trainable)
            # reference quirk kept on purpose: wr4_4 twice, wr4_3 never (source_segmenter.py:132-135)
            cw += [wr4_1, wr4_2, wr4_4, wr4_4]

        with st.name_scope('group_5'):
            wr5_1 = wv([3, 3, fb * 8, fb * 16], main_trainable)
            wr5_2 = wv([3, 3, fb * 16, fb * 16], main_trainable)
            block5_1 = residual_block(block4_2, wr5_1, wr5_2, keep_prob=keep_prob, leak=True, inc_dim=True, is_train=main_bn,
                                      bn_trainable=main_trainable)
            wr5_3 = wv([3, 3, fb * 16, fb * 16], main_trainable)
            wr5_4 = wv([3, 3, fb * 16, fb * 16], main_trainable)
            block5_2 = residual_block(block5_1, wr5_3, wr5_4, keep_prob=keep_prob, leak=True, is_train=main_bn,
                                      bn_trainable=main_trainable)
            cw += [wr5_1, wr5_2, wr5_3, wr5_4]

        with st.name_scope('group_6'):
            wr6_1 = wv([3, 3, fb * 16, fb * 16], main_trainable)
            wr6_2 = wv([3, 3, fb * 16, fb * 16], main_trainable)
            block6_1 = residual_block(block5_2, wr6_1, wr6_2, keep_prob=keep_prob, leak=True, is_train=main_bn,
                                      bn_trainable=main_trainable)
            wr6_3 = wv([3, 3, fb * 16, fb * 16], main_trainable)
            wr6_4 = wv([3, 3, fb * 16, fb * 16], main_trainable)
            block6_2 = residual_block(block6_1, wr6_3, wr6_4, keep_prob=keep_prob, leak=True, is_train=main_bn,
                                      bn_trainable=main_trainable)
            cw += [wr6_1, wr6_2, wr6_3, wr6_4]

        with st.name_scope('group_7'):
            wr7_1 = wv([3, 3, fb * 16, fb * 32], main_trainable)
            wr7_2 = wv([3, 3, fb * 32, fb * 32], main_trainable)
            block7_1 = residual_block(block6_2, wr7_1, wr7_2, keep_prob=keep_prob, leak=True, inc_dim=True, is_train=main_bn,
                                      bn_trainable=main_trainable)
            wr7_3 = wv([3, 3, fb * 32, fb * 32], main_trainable)
            wr7_4 = wv([3, 3, fb * 32, fb * 32], main_trainable)
            block7_2 = residual_block(block7_1, wr7_3, wr7_4, keep_prob=keep_prob, leak=True, is_train=main_bn,
                                      bn_trainable=main_trainable)
            cw += [wr7_1, wr7_2, wr7_3, wr7_4]

        with st.name_scope('group_8'):
            wr8_1 = wv([3, 3, fb * 32, fb * 32], main_trainable)
            wr8_2 = wv([3, 3, fb * 32, fb * 32], main_trainable)
            block8_1 = DR_block(block7_2, wr8_1, wr8_2, keep_prob=keep_prob, leak=True, rate=2, is_train=main_bn,
                                bn_trainable=main_trainable)
            wr8_3 = wv([3, 3, fb * 32, fb * 32], main_trainable)
            wr8_4 = wv([3, 3, fb * 32, fb * 32], main_trainable)
            block8_2 = DR_block(block8_1, wr8_3, wr8_4, keep_prob=keep_prob, leak=True, rate=2, is_train=main_bn,
                                bn_trainable=main_trainable)
            cw += [wr8_1, wr8_2, wr8_3, wr8_4]

        with st.name_scope('group_9'):
            w9_1 = wv([3, 3, fb * 32, fb * 32], main_trainable)
            conv9_1 = conv_bn_relu2d(block8_2, w9_1, keep_prob, is_train=main_bn, bn_trainable=main_trainable, leak=True)
            w9_2 = wv([3, 3, fb * 32, fb * 32], main_trainable)
            conv9_2 = conv_bn_relu2d(conv9_1, w9_2, keep_prob, is_train=main_bn, bn_trainable=main_trainable, leak=True)
            cw += [w9_1, w9_2]

        with st.name_scope('group_10'):
            local_size = 8 * 8   # r^2
            w10_1 = wv([3, 3, fb * 32, local_size * num_cls * 8], main_trainable)
            conv10_1 = conv2d(conv9_2, w10_1, keep_prob_=keep_prob, padding='SYMMETRIC')
            cw.append(w10_1)
            flat_conv10_1 = PS(conv10_1, r=8, n_channel=num_cls * 8, batch_size=self.batch_size)

        with st.name_scope('output'):
            w11_1 = wv([5, 5, num_cls * 8, num_cls], main_trainable)
            logits = conv2d(flat_conv10_1, w11_1, keep_prob_=1., padding='SYMMETRIC')
            cw.append(w11_1)

        self.conv_weights = cw
        return logits

    # -- eager "sess.run" -------------------------------------------------------------------------------
    def forward(self, x, keep_prob=1.0, main_bn=True, adapt_bn=True, drop_seed=0):
        """logits for a batch x [B,256,256,channels] (torch CUDA float32, NHWC)."""
        with self.store.as_default():
            self.store.begin_trace(drop_seed)
            logits = self.create_network(x, input_size=raw_size, input_channel=self.channels, num_cls=self.n_class,
                                         feature_base=self.feature_base, keep_prob=keep_prob, adapt_module=self.adapt_module,
                                         main_bn=main_bn, main_trainable=self.main_trainable, adapt_bn=adapt_bn,
                                         adapt_trainable=self.adapt_trainable)
        self.logits = logits
        return logits

    def _get_cost(self, logits, y):
        """source_segmenter.py:211-239: returns (loss vector [total, xent, dice]); element 0 is the backward root."""
        mc = float(self.miu_cross) if self.cross_flag is True else 0.0
        md = float(self.miu_dice) if self.dice_flag is True else 0.0
        out = SegLossFn.apply(logits, y, mc, md, 1.0 / self.world_size, sync_now())
        return out

    def evaluate(self, x, y, keep_prob=1.0, main_bn=True, adapt_bn=True, want_confusion=False):
        """forward + every monitoring output of the reference graph (predicter, compact_pred, cost, dice_eval ...)."""
        with torch.no_grad():
            logits = self.forward(x, keep_prob, main_bn, adapt_bn)
            lv = self._get_cost(logits, y)
            self.cost, self.weighted_loss, self.dice_loss = lv[0], lv[1], lv[2]
            self.predicter, self.compact_pred = K.softmax_argmax(logits.contiguous(), want_prob=True)
            self.compact_y, cm = lib.compact_and_confusion(y, self.compact_pred if want_confusion else None)
            self.dice_eval, self.dice_eval_arr = _dice_eval(self.compact_pred, y, self.n_class)
            self.regularizer_loss = self.l2_regularizer()
            if want_confusion:
                self.confusion_matrix = cm
        return self.cost

    def l2_regularizer(self):
        """reg_coeff * sum_i l2_loss(conv_weights[i]) (source_segmenter.py:237-239) as a 1-element device tensor"""
        if not hasattr(self, "_l2_table"):
            self._l2_table = self.store.chunk_table(lambda v: self.reg_coeff * v.l2_mult, np.float32)
        return K.l2_loss(self.store.arena, self._l2_table)

    def loss_and_grads(self, x, y, keep_prob, main_bn=True, adapt_bn=True, drop_seed=0):
        """fwd + bwd of cost (the L2 term's gradient is applied inside the optimiser kernel)."""
        self.store.zero_grad()
        logits = self.forward(x, keep_prob, main_bn, adapt_bn, drop_seed)
        lv = self._get_cost(logits, y)
        with wgrad_overlap():
            lv.backward(self.store.unit_grad(3))      # SegLossFn: element 0 is the root (cost); the upstream gradient is taken to be 1
        self.cost, self.weighted_loss, self.dice_loss = lv[0].detach(), lv[1].detach(), lv[2].detach()
        return self.cost

    # -- checkpoints (own format, keyed by the TF variable names) ------------------------------
    def save(self, path):
        return lib.atomic_savez(path, **{k.replace("/", "|"): v for k, v in self.store.state_dict().items()})

    def restore(self, sess_or_none, model_path):
        """source_segmenter.py:275-300 (relaxed name-matched restore); reads this package's .npz checkpoints."""
        with np.load(model_path) as z:
            sd = {k.replace("|", "/"): z[k] for k in z.files}
        self.store.load_state_dict(sd, strict=False)
        logging.info("Model restored from file: %s" % model_path)


class AdamOptimizer(object):
    """tf.train.AdamOptimizer(lr).minimize(cost + reg) over the store's flat arena (source_segmenter.py:378)."""

    def __init__(self, store, learning_rate, l2_table, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.store, self.lr, self.b1, self.b2, self.eps = store, float(learning_rate), beta1, beta2, epsilon
        self.m = torch.zeros_like(store.arena)
        self.v = torch.zeros_like(store.arena)
        self.l2 = l2_table
        self.t = 0

    def step(self):
        self.t += 1
        K.adam_step(self.store.arena, self.store.grad_arena, self.m, self.v, self.l2, None, self.lr, self.b1, self.b2, self.eps, self.t)

    # slots, beta powers (t) and the learning-rate variable are part of every TF checkpoint (tf.train.Saver, lib.py:23-29)
    def state_dict(self):
        """slots by variable name ('<variable>|Adam', '<variable>|Adam_1': TF's slot names), never by arena position"""
        d = {"t": np.int64(self.t), "lr": np.float64(self.lr)}
        d.update(self.store.slots_to_dict(self.m, "Adam"))
        d.update(self.store.slots_to_dict(self.v, "Adam_1"))
        return d

    def load_state_dict(self, sd, lr=True):
        """-> (restored, missing) variable names"""
        done, missing = self.store.slots_from_dict(self.m, sd, "Adam")
        done_v, missing_v = self.store.slots_from_dict(self.v, sd, "Adam_1")
        self.t = int(sd["t"])
        if lr:
            self.lr = float(sd["lr"])
        return [n for n in done if n in set(done_v)], sorted(set(missing) | set(missing_v))


class MomentumOptimizer(object):
    """tf.train.MomentumOptimizer with staircase exponential decay (source_segmenter.py:359-373)."""

    def __init__(self, store, learning_rate, l2_table, momentum, decay_rate, decay_steps):
        self.store, self.lr0, self.mom, self.decay_rate, self.decay_steps = store, float(learning_rate), momentum, decay_rate, decay_steps
        self.acc = torch.zeros_like(store.arena)
        self.l2 = l2_table
        self.t = 0

    @property
    def lr(self):
        return self.lr0 * (self.decay_rate ** (self.t // max(self.decay_steps, 1)))

    @lr.setter
    def lr(self, value):          # tf.assign(learning_rate_node, ...) of the periodic checkpoint block (source_segmenter.py:512-513)
        self.lr0 = float(value) / (self.decay_rate ** (self.t // max(self.decay_steps, 1)))

    def step(self):
        K.momentum_step(self.store.arena, self.store.grad_arena, self.acc, self.l2, None, self.lr, self.mom)
        self.t += 1

    def state_dict(self):
        d = {"t": np.int64(self.t), "lr0": np.float64(self.lr0)}
        d.update(self.store.slots_to_dict(self.acc, "Momentum"))
        return d

    def load_state_dict(self, sd, lr=True):
        done, missing = self.store.slots_from_dict(self.acc, sd, "Momentum")
        self.t = int(sd["t"])
        if lr:
            self.lr0 = float(sd["lr0"])
        return done, missing


class Trainer(object):
    """Train a network instance (source_segmenter.py:303-570).

    :param net: the network instance to train
    :param train_list / val_list: lists of .tfrecords files (one 256x256x3 slice each), or any object with a
           `next_batch(batch_size) -> np.ndarray [B,256,256,4]` method (image channels 0:3, label map in channel 3)
    """

    def __init__(self, net, train_list, val_list, num_cls, batch_size, test_nii_list=None, test_label_list=None,
                 optimizer="momentum", opt_kwargs={}, num_epochs=100, checkpoint_space=500, lr_update_flag=False,
                 reducer=None, shard=None):
        self.net = net
        self.shard = shard              # (rank, world_size) under data parallelism: this rank's share of the file lists
        self.rank = shard[0] if shard else 0
        self.seed_offset = rank_seed(self.rank)
        self.batch_size = batch_size
        self.num_cls = num_cls
        self.checkpoint_space = checkpoint_space
        self.opt_kwargs = dict(opt_kwargs)
        self.optimizer = optimizer
        self.train_list = train_list
        self.val_list = val_list
        self.test_label_list = test_label_list
        self.test_nii_list = test_nii_list
        self.lr_update_flag = lr_update_flag
        self.loss_dict = {}
        self.reducer = reducer          # parallel.GradReducer or None
        self.opt = None
        self.global_step = 0
        self.step_times = []

    def next_batch(self, source, capacity=120, num_threads=4, min_after_dequeue=30, label_type='float'):
        """source_segmenter.py:331-355: returns an iterator yielding (pair_feed [B,256,256,4], fids)."""
        from .tfrecord import SliceQueue
        if hasattr(source, "next_batch"):
            return source
        return SliceQueue(source, self.batch_size, capacity=capacity, min_after_dequeue=min_after_dequeue, num_threads=num_threads,
                          shard=self.shard)

    def save_checkpoint(self, output_path):
        """lib._save (tf.train.Saver over ALL variables, lib.py:23-29): model variables + optimiser slots / step / learning rate.
        Each file is written next to its destination and renamed into place, so a reader never sees a half-written checkpoint;
        `checkpoint-<global_step>.npz` keeps the per-step history the reference's `global_step=` argument produces."""
        ck = os.path.join(output_path, "checkpoint.npz")
        self.net.save(ck)
        lib.atomic_savez(os.path.join(output_path, "optimizer.npz"), kind=self.optimizer, global_step=np.int64(self.global_step),
                         **self.opt.state_dict())
        if os.path.exists(ck):
            shutil.copyfile(ck, os.path.join(output_path, "checkpoint-%d.npz" % self.global_step))
        return ck

    def restore_optimizer(self, restored_path):
        """source_segmenter.py:275-300, 460-462: whatever the checkpoint holds comes back, matched BY VARIABLE NAME (as tf.train.Saver
        does); lr_update_flag keeps the configured rate.  False when the folder holds no state for this kind of optimiser."""
        f = os.path.join(restored_path, "optimizer.npz")
        if not os.path.exists(f):
            return False
        with np.load(f) as z:
            if "kind" not in z.files or str(z["kind"]) != self.optimizer:
                logging.warning("optimizer state in %s is not %s state: slots start fresh" % (f, self.optimizer))
                return False
            done, missing = self.opt.load_state_dict({k: z[k] for k in z.files if k not in ("kind", "global_step")},
                                                     lr=self.lr_update_flag is not True)
            if not done:
                logging.warning("optimizer state in %s matches no variable of this graph: slots start fresh" % f)
                return False
            if missing:
                logging.warning("optimizer slots not found in %s for %d variables (e.g. %s): they start fresh" % (f, len(missing), missing[0]))
            self.global_step = int(z["global_step"])
        return True

    def _feeder(self, source):
        """dequeue -> pinned staging -> async H2D -> on-device one-hot, one batch ahead of the step (feeder.DeviceFeeder)"""
        from .feeder import DeviceFeeder
        return DeviceFeeder(self.next_batch(source), self.batch_size, self.num_cls, self.net.device)

    def _log_step(self, pending):
        """the reference fetches `cost` with every sess.run; here the host reads step k's loss after step k+1 has been queued,
        so the read never drains the GPU"""
        step, epoch, loss, start = pending
        lv = float(loss)
        now = time.time()
        self.step_times.append(now - max(start, self._last_done))
        self._last_done = now
        if getattr(self, "scalars", None) is not None:
            self.scalars.write("train_step", step=step, epoch=epoch, loss=lv, step_time_s=self.step_times[-1])
        logging.info("Training at step %s epoch %s , loss is %0.4f" % (str(step), str(epoch), lv))
        logging.info("Time elapsed %s seconds" % (str(self.step_times[-1])))

    def _get_optimizer(self, training_iters):
        """source_segmenter.py:357-381"""
        net = self.net
        l2 = net.store.chunk_table(lambda v: net.reg_coeff * v.l2_mult, np.float32)
        if self.optimizer == "momentum":
            lr = self.opt_kwargs.pop("learning_rate", 0.2)
            decay_rate = self.opt_kwargs.pop("decay_rate", 0.95)
            momentum = self.opt_kwargs.pop("momentum", 0.2)
            return MomentumOptimizer(net.store, lr, l2, momentum, decay_rate, training_iters)
        elif self.optimizer == "adam":
            lr = self.opt_kwargs.pop("learning_rate", None)
            self._new_LR = lr
            return AdamOptimizer(net.store, lr, l2, **self.opt_kwargs)
        raise ValueError("unknown optimizer %r" % (self.optimizer,))

    def capture_step(self, batch_x, batch_y, dropout):
        """Step capture (step_capture.py): from now on train_step on batches of these shapes and this keep probability is ONE hipGraph
        launch (the reference: one sess.run per step, source_segmenter.py:484-489).  Adam only (its per-step scalar travels in the
        captured step's device block); the two warm-up steps are real updates.  Not under data parallelism."""
        if self.reducer is not None:
            raise RuntimeError("capture_step: not under data parallelism (the bucketed all-reduce runs on a side stream)")
        if self.opt is None:
            self.opt = self._get_optimizer(100)
        if not isinstance(self.opt, AdamOptimizer):
            raise RuntimeError("capture_step: only the Adam optimiser's per-step scalars are carried by a captured step")
        from .step_capture import CapturedStep
        self._cap = None
        g0 = self.global_step
        cap = {"dropout": float(dropout), "x": tuple(batch_x.shape), "y": tuple(batch_y.shape)}
        cap["step"] = CapturedStep(lambda x_, y_: self.train_step(x_, y_, dropout, 0), [batch_x, batch_y], adam=self.opt)
        self.global_step = g0 + 2      # the recording counted itself; the two warm-up steps are real
        self._cap = cap
        return cap

    def train_step(self, batch_x, batch_y, dropout, step):
        """the accelerated unit: sess.run((optimizer, cost, lr), feed_dict) of source_segmenter.py:484-489"""
        net = self.net
        cap = getattr(self, "_cap", None)
        if cap is not None and cap["dropout"] == float(dropout) and cap["x"] == tuple(batch_x.shape) and cap["y"] == tuple(batch_y.shape):
            self.global_step += 1
            K.weights_changed()
            return cap["step"].replay(step + 1 + self.seed_offset, batch_x, batch_y)
        loss = net.loss_and_grads(batch_x, batch_y, dropout, main_bn=True, adapt_bn=True, drop_seed=step + 1 + self.seed_offset)
        if self.reducer is not None:
            self.reducer.allreduce(net.store.grad_arena)
        self.opt.step()
        self.global_step += 1
        return loss

    def train(self, output_path, restored_path=None, restore=False, training_iters=100, epochs=100, display_step=5,
              dropout=0.75):
        """source_segmenter.py:429-523"""
        # the reference returns "<output_path>/model.cpkt" (a tf.train.Saver prefix); here the checkpoint is this package's .npz,
        # and the returned path is the file test_choose_model / restore can be handed
        save_path = os.path.join(output_path, "checkpoint.npz")
        if epochs == 0:
            return save_path
        output_path = os.path.abspath(output_path)
        if not restore and self.rank == 0:
            shutil.rmtree(output_path, ignore_errors=True)
        os.makedirs(output_path, exist_ok=True)
        barrier()
        if self.opt is None:
            self.opt = self._get_optimizer(training_iters)
        if restore:
            if restored_path is None:
                raise Exception("No restore path is provided")
            ck = os.path.join(restored_path, "checkpoint.npz")
            if os.path.exists(ck):
                self.net.restore(None, ck)
                self.restore_optimizer(restored_path)
            else:
                print("Unable to restore, start from beginning")
            if self.lr_update_flag is True:
                self.opt.lr = self._new_LR
        from .metrics import ScalarLog
        self.scalars = ScalarLog(output_path, self.rank)
        feed_all = self._feeder(self.train_list)
        feed_val = self._feeder(self.val_list)
        self._last_done = 0.0
        pending = None
        try:
            for epoch in range(epochs):
                for step in range((epoch * training_iters), ((epoch + 1) * training_iters)):
                    start = time.time()
                    batch_x, batch_y, fid = feed_all.next()
                    loss = self.train_step(batch_x, batch_y, dropout, step)
                    if pending is not None:
                        self._log_step(pending)
                    pending = (step, epoch, loss, start) if verbose else None
                    if step % display_step == 0:
                        if pending is not None:
                            self._log_step(pending)
                            pending = None
                        self.output_minibatch_stats(step, batch_x, batch_y)
                        val_x, val_y, _ = feed_val.next()
                        self.val_stats(step, val_x, val_y, True)
                    if step % self.checkpoint_space == 0 and step > 10000:
                        if self.rank == 0:
                            self.save_checkpoint(output_path)
                        self.opt.lr = self.opt.lr * 0.9
                if pending is not None:
                    self._log_step(pending)
                    pending = None
                logging.info("Global step %s" % str(self.global_step))
        finally:
            feed_all.close()
            feed_val.close()
            self.scalars.close()
        logging.info("Optimization Finished!")
        if self.rank == 0:      # replicas hold identical weights; BN moving statistics are rank 0's (per-replica statistics)
            self.save_checkpoint(output_path)
        barrier()
        return save_path

    def output_minibatch_stats(self, step, batch_x, batch_y):
        """source_segmenter.py:525-539: logging-only forward (BN train mode, keep_prob 1: it DOES move the BN moving stats)"""
        self.net.evaluate(batch_x, batch_y, keep_prob=1.0, main_bn=True, adapt_bn=True)
        self.loss_dict["train"] = (step, float(self.net.cost), float(self.net.dice_eval))
        if getattr(self, "scalars", None) is not None:
            self.scalars.write("train_eval", step=step, cost=self.loss_dict["train"][1], dice=self.loss_dict["train"][2])

    def val_stats(self, step, batch_x, batch_y, detail=False):
        """source_segmenter.py:541-570"""
        self.net.evaluate(batch_x, batch_y, keep_prob=1.0, main_bn=False, adapt_bn=False, want_confusion=detail)
        if detail:
            _indicator_eval(self.net.confusion_matrix, verbose=verbose)
        self.loss_dict["val"] = (step, float(self.net.cost), float(self.net.dice_eval))
        if getattr(self, "scalars", None) is not None:
            self.scalars.write("val_eval", step=step, cost=self.loss_dict["val"][1], dice=self.loss_dict["val"][2])

    # -- volume inference -------------------------------------------------------------------------------
    def _predict_batch(self, vol, slice_y):
        dev = self.net.device
        x = torch.from_numpy(vol).to(dev)
        y = lib.label_decomp_device(self.num_cls, torch.from_numpy(slice_y).to(dev))
        self.net.evaluate(x, y, keep_prob=1.0, main_bn=False, adapt_bn=False, want_confusion=True)
        return self.net.compact_pred.cpu().numpy(), self.net.confusion_matrix

    def test_eval(self, sess=None, output_path=".", flip_correction=True, save_result=False):
        """source_segmenter.py:572-632: inference on the .nii test volumes -> (per-class mean Dice, see volume_eval for the 2nd value)"""
        from . import volume_eval as ve
        pred_folder = os.path.join(output_path, "test_pred")
        os.makedirs(pred_folder, exist_ok=True)
        self.test_pair_list = list(zip(self.test_label_list, self.test_nii_list))

        def on_sample(raw_y, tmp_y, nii_fid):
            if save_result is True:
                lib._save_nii_prediction(raw_y, tmp_y, nii_fid, pred_folder, out_bname="dense_pred_" + os.path.basename(nii_fid),
                                         num_cls=self.num_cls)
        sample_eval_list, _ = ve.test_eval(self._predict_batch, self.test_label_list, self.test_nii_list, self.net.batch_size, self.num_cls,
                                           flip_correction, shuffle=False, on_sample=on_sample)
        return self.sample_metric_stddev(sample_eval_list)

    def test_choose_model(self, this_model, output_path):
        """source_segmenter.py:666-675: restore a checkpoint (.npz of this package), then test_eval"""
        os.makedirs(output_path, exist_ok=True)
        self.net.restore(None, this_model)
        logging.info("model has been loaded!")
        dice, jac = self.test_eval(None, output_path)
        logging.info("testing finished")
        return dice, jac

    def sample_metric_stddev(self, sample_eval_list):
        """source_segmenter.py:634-664"""
        from . import volume_eval as ve
        return ve.sample_metric_stddev(sample_eval_list, self.num_cls, contour_map, quiet=not verbose)
