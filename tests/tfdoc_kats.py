"""Known answers written out BY HAND from TensorFlow's published semantics (API docs / op definitions of TF 1.x), not produced by
any code of this repository.  They are the only pin of the TF-1.4 arithmetic that is independent of oracle/tf_ops.py: TensorFlow
itself cannot run here and the reference ships no vectors.  Each entry cites the documentation text it follows and
the reference call site that relies on it.  tests/test_tfdoc_kats.py holds the oracle and the product's host arithmetic to these,
tests/test_gpu_tfdoc_kats.py the HIP kernels.
"""
import math

import numpy as np

# ---- 1. tf.pad(..., mode='SYMMETRIC')  (layers.py:23,72,91) -----------------------------------------------------------------------
# tf.pad API doc example (tensorflow/python/ops/array_ops.py, `pad` docstring):
#     t = tf.constant([[1, 2, 3], [4, 5, 6]]); paddings = tf.constant([[1, 1], [2, 2]])
#     tf.pad(t, paddings, "SYMMETRIC")  # [[2, 1, 1, 2, 3, 3, 2],
#                                       #  [2, 1, 1, 2, 3, 3, 2],
#                                       #  [5, 4, 4, 5, 6, 6, 5],
#                                       #  [5, 4, 4, 5, 6, 6, 5]]
# (and "REFLECT" gives [[6,5,4,5,6,5,4], ...]: SYMMETRIC repeats the edge sample, REFLECT does not)
PAD_SYM_IN = np.array([[1, 2, 3], [4, 5, 6]], np.float32)
PAD_SYM_PADDINGS = (1, 2)                       # rows (H) by 1, columns (W) by 2
PAD_SYM_OUT = np.array([[2, 1, 1, 2, 3, 3, 2],
                        [2, 1, 1, 2, 3, 3, 2],
                        [5, 4, 4, 5, 6, 6, 5],
                        [5, 4, 4, 5, 6, 6, 5]], np.float32)

# ---- 2. 'SAME' padding  (layers.py:18,67,86; adversarial.py:342-391, 409-434) -------------------------------------------------------
# TF API guide "Neural Network > Convolution" (api_guides/python/nn.md, section "Notes on SAME Convolution Padding"):
#     out_height = ceil(float(in_height) / float(strides[1]))
#     if (in_height % strides[1] == 0): pad_along_height = max(filter_height - strides[1], 0)
#     else:                             pad_along_height = max(filter_height - (in_height % strides[1]), 0)
#     pad_top = pad_along_height // 2 ; pad_bottom = pad_along_height - pad_top     ("...the extra padding is added at the bottom/right")
# tf.nn.atrous_conv2d doc: equivalent to a convolution with filters "upsampled" by inserting rate-1 zeros, i.e. effective filter
# height = filter_height + (filter_height - 1) * (rate - 1).
# rows: (in, filter, stride, rate) -> (out, pad_before, pad_after), every strided / dilated geometry on the reference path
SAME_CASES = [
    ((256, 3, 1, 1), (256, 1, 1)),      # every 3x3 stride-1 conv of the segmenter
    ((32, 3, 1, 2), (32, 2, 2)),        # group_8 dilated convs: effective filter 5
    ((256, 3, 2, 1), (128, 0, 1)),      # cls_1_3 / mask_cls_1: 256 % 2 == 0 -> pad_along = 3 - 2 = 1 -> (0, 1)
    ((128, 5, 2, 1), (64, 1, 2)),       # cls_2_3: 5 - 2 = 3 -> (1, 2)
    ((64, 3, 2, 1), (32, 0, 1)),        # cls_3_3
    ((32, 3, 2, 1), (16, 0, 1)),        # cls_4_3
    ((16, 5, 4, 1), (4, 0, 1)),         # cls_5_3: 5 - 4 = 1 -> (0, 1)
    ((128, 5, 4, 1), (32, 0, 1)),       # m_cls_2_3
    ((32, 5, 4, 1), (8, 0, 1)),         # m_cls_3_3
    ((7, 3, 2, 1), (4, 1, 1)),          # odd extent: 7 % 2 = 1 -> pad_along = 3 - 1 = 2 -> (1, 1)
    ((10, 5, 4, 1), (3, 1, 2)),         # 10 % 4 = 2 -> pad_along = 5 - 2 = 3 -> (1, 2)
]
# a numeric instance of the asymmetry, worked by hand: x = [1,2,3,4] along W (H = 1), filter [1,10,100] along W, stride 2, SAME:
# out_w = 2, pad_along = max(3-2,0) = 1 -> left 0, right 1:  y0 = 1*1 + 2*10 + 3*100 = 321 ; y1 = 3*1 + 4*10 + 0*100 = 43
# (symmetric "padding = k//2" would give y0 = 0*1 + 1*10 + 2*100 = 210, y1 = 2*1 + 3*10 + 4*100 = 432)
SAME_NUMERIC_X = np.array([1, 2, 3, 4], np.float32).reshape(1, 1, 4, 1)
SAME_NUMERIC_W = np.array([1, 10, 100], np.float32).reshape(1, 3, 1, 1)
SAME_NUMERIC_Y = np.array([321, 43], np.float32).reshape(1, 1, 2, 1)

# ---- 3. tf.contrib.layers.batch_norm(decay=0.9, updates_collections=None)  (layers.py:100) ------------------------------------------
# contrib batch_norm doc: epsilon default 0.001; "decay: Decay for the moving average"; moving averages updated by
# assign_moving_average: variable -= (1 - decay) * (variable - value); initial moving_mean 0, moving_variance 1; scale/center -> y =
# gamma * (x - mean) / sqrt(var + eps) + beta.  With the fused kernel (the path taken when updates_collections is None and the
# input is 4-D) the normalisation uses the POPULATION variance of the batch and the value fed to the moving variance is the
# Bessel-corrected one (FusedBatchNorm op: "batch_variance: ... to be used by TensorFlow to compute the running variance";
# tensorflow/core/kernels/fused_batch_norm_op.cc: rest_size_adjust = rest_size / (rest_size - 1)).
# One channel, four samples x = 1, 2, 3, 4:
#     mean = 2.5 ; population variance = (2.25 + .25 + .25 + 2.25)/4 = 1.25 ; Bessel-corrected = 5/3
#     y_i = (x_i - 2.5) / sqrt(1.25 + 0.001)              (gamma 1, beta 0)
#     moving_mean     <- 0 - 0.1 * (0 - 2.5)  = 0.25
#     moving_variance <- 1 - 0.1 * (1 - 5/3)  = 1.0666666...
BN_X = np.array([1, 2, 3, 4], np.float32).reshape(4, 1, 1, 1)
BN_Y = ((BN_X - 2.5) / math.sqrt(1.251)).astype(np.float32)
BN_MOVING_MEAN = 0.25
BN_MOVING_VAR = 1.0 + 0.1 * (5.0 / 3.0 - 1.0)
# inference mode afterwards: y = (x - moving_mean) / sqrt(moving_var + eps)
BN_Y_INFER = ((BN_X - BN_MOVING_MEAN) / math.sqrt(BN_MOVING_VAR + 1e-3)).astype(np.float32)

# ---- 4. gradient masks ------------------------------------------------------------------------------------------------------------
# tf.clip_by_value(t, lo, hi) (source_segmenter.py:252) is minimum(maximum(t, lo), hi) in TF 1.4 (clip_ops.py); MaximumGrad /
# MinimumGrad (math_grad.py: xmask = greater_equal(x, y) / less_equal(x, y)) send the gradient to the FIRST argument on ties.
# d clip(t, .005, 1) / dt at t = .001, .005, .5, 1, 1.2  ->  0, 1, 1, 1, 0
CLIP_T = np.array([0.001, 0.005, 0.5, 1.0, 1.2], np.float32)
CLIP_GRAD = np.array([0, 1, 1, 1, 0], np.float32)
# tf.nn.leaky_relu(x, alpha=0.2) (layers.py:12,35,166,187) = maximum(alpha * x, x) in TF 1.4 (nn_ops.py): slope alpha for x < 0,
# 1 for x > 0, and at the tie x == 0 the gradient goes to the first argument alpha * x  ->  alpha
LRELU_X = np.array([-2.0, 0.0, 3.0], np.float32)
LRELU_Y = np.array([-0.4, 0.0, 3.0], np.float32)
LRELU_GRAD = np.array([0.2, 0.2, 1.0], np.float32)
# tf.nn.max_pool gradient (MaxPoolGrad, maxpooling_op.cc): the whole gradient goes to the FIRST maximal element of the window in
# row-major scan order.  Window [[5, 5], [1, 5]] -> d = [[1, 0], [0, 0]]
POOL_X = np.array([[5, 5], [1, 5]], np.float32).reshape(1, 2, 2, 1)
POOL_DX = np.array([[1, 0], [0, 0]], np.float32).reshape(1, 2, 2, 1)

# ---- 5. optimisers ----------------------------------------------------------------------------------------------------------------
# tf.train.AdamOptimizer doc:  t <- t + 1 ; lr_t <- learning_rate * sqrt(1 - beta2^t) / (1 - beta1^t)
#                              m_t <- beta1 * m + (1 - beta1) * g ; v_t <- beta2 * v + (1 - beta2) * g * g
#                              variable <- variable - lr_t * m_t / (sqrt(v_t) + epsilon)        (epsilon "hat" form, 1e-8 default)
# w = 1, g = 0.5, first step, lr = 1e-3:  m = .05 ; v = 2.5e-4 ; lr_t = 1e-3 * sqrt(.001) / .1
ADAM_W0, ADAM_G, ADAM_LR = 1.0, 0.5, 1e-3
_lr_t = ADAM_LR * math.sqrt(1 - 0.999) / (1 - 0.9)
ADAM_W1 = ADAM_W0 - _lr_t * 0.05 / (math.sqrt(2.5e-4) + 1e-8)
# second step with the same gradient: m = .9*.05 + .1*.5 = .095 ; v = .999*2.5e-4 + .001*.25 = 4.9975e-4
_lr_t2 = ADAM_LR * math.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
ADAM_W2 = ADAM_W1 - _lr_t2 * 0.095 / (math.sqrt(4.9975e-4) + 1e-8)
# tf.train.RMSPropOptimizer doc (decay 0.9, momentum 0, epsilon 1e-10; adversarial.py:643-652):
#     mean_square <- decay * mean_square + (1 - decay) * g^2 ; mom <- momentum * mom + lr * g / sqrt(mean_square + epsilon)
#     variable <- variable - mom            (the "rms" slot is created with ones: rmsprop.py _create_slots, init_ops.ones_initializer)
# w = 1, g = 2, lr = 3e-4:  ms = .9 + .1 * 4 = 1.3
RMS_W0, RMS_G, RMS_LR = 1.0, 2.0, 3e-4
RMS_MS1 = 1.3
RMS_W1 = RMS_W0 - RMS_LR * RMS_G / math.sqrt(1.3 + 1e-10)
# tf.nn.l2_loss doc: output = sum(t ** 2) / 2 ; [1, 2, 3] -> 7
L2_T = np.array([1, 2, 3], np.float32)
L2_OUT = 7.0

# ---- 6. softmax cross-entropy pieces (source_segmenter.py:241-258) and tf.argmax --------------------------------------------------
# tf.nn.softmax doc: softmax = exp(logits) / reduce_sum(exp(logits), dim); logits (0, ln 3) -> (.25, .75)
SOFTMAX_Z = np.array([0.0, math.log(3.0)], np.float32)
SOFTMAX_P = np.array([0.25, 0.75], np.float32)
# tf.argmax doc ("Note that in case of ties the identity of the return value is not guaranteed") — the CPU/GPU kernels (Eigen
# argmax reducer) return the smallest index; the label maps of the reference rely on that only at exact ties
ARGMAX_Z = np.array([[1, 3, 3, 0, 3], [2, 2, 2, 2, 2]], np.float32)
ARGMAX_OUT = np.array([1, 0], np.int64)
