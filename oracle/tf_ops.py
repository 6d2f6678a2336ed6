"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the TensorFlow-1.4 op semantics that the reference's
hot path lowers to.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product path (medical-cross-modality-domain-adaptation_amd/) never does.

PARITY UNPINNED: the reference ships no tests / golden vectors, and TensorFlow 1.4 (an un-vendored pip
dependency: tensorflow-gpu==1.4.0, reference README.md:24) cannot be installed here, so the arithmetic below
is restated from TF's documented/known semantics, not checked against a TF run.  What IS pinned to the
reference's own code: the op SEQUENCES (ops.PS, layers.*, Full_DRN.create_network, lib._label_decomp), by
executing the reference's Python over a numpy stand-in for `tf` (tests/golden/make_golden.py).

Everything works on torch CPU tensors in NHWC, dtype float32 or float64 (pass float64 tensors for the
high-precision adjudicator), and is differentiable through torch autograd (= TF autodiff of the same graph).
Each function cites the reference call site (file:line under /root/reference) and the TF op it restates.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3      # tf.contrib.layers.batch_norm default epsilon (layers.py:100)
BN_DECAY = 0.90    # layers.py:100
LEAK = 0.2         # tf.nn.leaky_relu default alpha (layers.py:12,35,166,187)


# ---- padding arithmetic --------------------------------------------------------------------------------
def same_pad(in_size, k, stride, dil=1):
    """TF 'SAME' (tf.nn.conv2d / max_pool): out = ceil(in/stride); total = max((out-1)*stride + (k-1)*dil + 1 - in, 0);
    pad_before = total // 2, the remainder goes after (asymmetric for even totals... odd totals put the extra after)."""
    out = -(-in_size // stride)
    eff = (k - 1) * dil + 1
    total = max((out - 1) * stride + eff - in_size, 0)
    return out, total // 2, total - total // 2


def sym_index(n, p):
    """index vector of tf.pad(..., 'SYMMETRIC') along one axis: mirror INCLUDING the edge sample"""
    return list(range(p - 1, -1, -1)) + list(range(n)) + list(range(n - 1, n - 1 - p, -1))


def pad_symmetric(x, ph, pw):
    """tf.pad(x, [[0,0],[ph,ph],[pw,pw],[0,0]], 'SYMMETRIC') (layers.py:23,72,91)"""
    ih = torch.tensor(sym_index(x.shape[1], ph), dtype=torch.long)
    iw = torch.tensor(sym_index(x.shape[2], pw), dtype=torch.long)
    return x.index_select(1, ih).index_select(2, iw)


# ---- convolution ---------------------------------------------------------------------------------------
def conv2d(x, w, stride=1, dil=1, padding="SAME"):
    """tf.nn.conv2d (NHWC x HWIO, cross-correlation) / tf.nn.atrous_conv2d(rate=dil) (layers.py:18,24,67,73,86,92).
    padding: 'SAME' (TF asymmetric rule), 'SYMMETRIC' (mirror-pad k//2 then VALID), 'VALID'."""
    R, S = w.shape[0], w.shape[1]
    if padding == "SYMMETRIC":
        x = pad_symmetric(x, R // 2, S // 2)
        pads = (0, 0, 0, 0)
    elif padding == "SAME":
        _, pt, pb = same_pad(x.shape[1], R, stride, dil)
        _, pl, pr = same_pad(x.shape[2], S, stride, dil)
        pads = (pl, pr, pt, pb)
    elif padding == "VALID":
        pads = (0, 0, 0, 0)
    else:
        raise ValueError(padding)
    xn = x.permute(0, 3, 1, 2)
    if any(pads):
        xn = F.pad(xn, pads)
    wn = w.permute(3, 2, 0, 1)
    y = F.conv2d(xn, wn, None, stride=stride, padding=0, dilation=dil)
    return y.permute(0, 2, 3, 1)


def round_bf16(t):
    """round-to-nearest-even to bfloat16 and back to float32: what a bf16 MFMA operand (or a bf16 tensor in HBM) holds.  Used only
    to budget the tolerance of BASELINE config 5 (bf16 mixed precision) on the CPU; nothing on the fp32 path calls it."""
    return t.to(torch.bfloat16).to(torch.float32) if t.dtype == torch.float32 else t


def conv2d_dgrad_by_phases(dy, w, in_hw, stride, pad_t, pad_l):
    """Data gradient of a strided zero-padded conv, restated the way the HIP path computes it (csrc/conv_igemm.hip::plan_phases):
    dx[h] receives only the filter taps r with r = (h + pad) mod stride, so every residue class of pixels ("phase") is a stride-1
    correlation of dy with the flipped sub-filter {r = a, a+stride, ...}:
        dx[h0 + stride*i] = sum_t dy[i + q - t] * W[a + stride*t],   h0 = (a - pad) mod stride,  q = (h0 + pad - a) / stride
    Checked on the CPU against autograd (tests/test_host.py); pure test infrastructure like the rest of this module.
    dy [N,OH,OW,K], w [R,S,C,K], in_hw = (H, W) of the conv input -> dx [N,H,W,C]."""
    R, S, C, K = w.shape
    N, OH, OW, _ = dy.shape
    H, W = in_hw
    dx = torch.zeros((N, H, W, C), dtype=dy.dtype)
    for a in range(stride):
        for b in range(stride):
            T, U = -(-(R - a) // stride) if a < R else 0, -(-(S - b) // stride) if b < S else 0
            h0, w0 = (a - pad_t) % stride, (b - pad_l) % stride
            if T == 0 or U == 0 or h0 >= H or w0 >= W:
                continue
            qa, qb = (h0 + pad_t - a) // stride, (w0 + pad_l - b) // stride
            I, J = (H - 1 - h0) // stride + 1, (W - 1 - w0) // stride + 1
            sub = w[a::stride, b::stride]                              # [T,U,C,K]
            wf = torch.flip(sub, (0, 1)).permute(0, 1, 3, 2)           # flipped, K <-> C : filter of the correlation over dy
            pt, pl = T - 1 - qa, U - 1 - qb                            # zero padding before (may be negative: crop instead)
            # rows i' in [0, I) read dy rows i' - pt ... i' - pt + T - 1
            need_h, need_w = I + T - 1, J + U - 1
            src = torch.zeros((N, need_h, need_w, K), dtype=dy.dtype)
            lo_h, lo_w = max(pt, 0), max(pl, 0)                         # where dy row 0 lands in the padded buffer
            sh, sw = max(-pt, 0), max(-pl, 0)                           # dy rows skipped when the padding is negative
            nh, nw = min(OH - sh, need_h - lo_h), min(OW - sw, need_w - lo_w)
            if nh > 0 and nw > 0:
                src[:, lo_h:lo_h + nh, lo_w:lo_w + nw] = dy[:, sh:sh + nh, sw:sw + nw]
            out = F.conv2d(src.permute(0, 3, 1, 2), wf.permute(3, 2, 0, 1)).permute(0, 2, 3, 1)     # [N,I,J,C]
            dx[:, h0::stride, w0::stride] = out[:, :I, :J]
    return dx


def conv2d_direct_np(x, w, stride=1, dil=1, padding="SAME"):
    """From-the-definition float64 loop nest (numpy) — an implementation-independent check of conv2d() above for
    small shapes (pure-Python loops over taps only)."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    N, H, W, C = x.shape
    R, S, _, K = w.shape
    if padding == "SYMMETRIC":
        x = np.pad(x, ((0, 0), (R // 2, R // 2), (S // 2, S // 2), (0, 0)), mode="symmetric")
        pt = pl = 0
        OH = (x.shape[1] - ((R - 1) * dil + 1)) // stride + 1
        OW = (x.shape[2] - ((S - 1) * dil + 1)) // stride + 1
    elif padding == "SAME":
        OH, pt, pb = same_pad(H, R, stride, dil)
        OW, pl, pr = same_pad(W, S, stride, dil)
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
        pt = pl = 0
    else:
        pt = pl = 0
        OH = (H - ((R - 1) * dil + 1)) // stride + 1
        OW = (W - ((S - 1) * dil + 1)) // stride + 1
    y = np.zeros((N, OH, OW, K), dtype=np.float64)
    for r in range(R):
        for s in range(S):
            patch = x[:, r * dil: r * dil + (OH - 1) * stride + 1: stride, s * dil: s * dil + (OW - 1) * stride + 1: stride, :]
            y += np.einsum("nhwc,ck->nhwk", patch, w[r, s])
    return y


# ---- dropout (mask stream is the product's documented counter hash; see csrc/pnp_common.h) -------------------
def _fmix32(h):
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
    h ^= h >> np.uint32(13)
    h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def drop_key(seed, stream_id):
    lo = np.uint32(seed & 0xFFFFFFFF)
    hi = np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        k = _fmix32(np.array([np.uint32(hi + np.uint32(0x9E3779B9) * np.uint32(stream_id + 1))], dtype=np.uint32))[0]
        return _fmix32(np.array([lo ^ k], dtype=np.uint32))[0]


def drop_thresh(keep):
    t = (1.0 - float(np.float32(keep))) * 16777216.0 + 0.5
    return np.uint32(min(max(t, 0.0), 16777216.0))


def dropout_mask(shape, keep_prob, seed, stream_id):
    """float32 {0,1} keep mask: tf.nn.dropout's floor(keep + u) restated on the product's counter hash
    mask(idx) = (fmix32((idx*0xCC9E2D51) ^ key) >> 8) >= round((1-keep)*2^24)"""
    n = int(np.prod(shape))
    if keep_prob >= 1.0:
        return np.ones(shape, dtype=np.float32)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint32)
        h = _fmix32((idx * np.uint32(0xCC9E2D51)) ^ drop_key(seed, stream_id))
    return ((h >> np.uint32(8)) >= drop_thresh(keep_prob)).astype(np.float32).reshape(shape)


def dropout(x, keep_prob, seed=0, stream_id=0, mask=None):
    """tf.nn.dropout (layers.py:25,74,93).  TF 1.4 computes `math_ops.div(x, keep_prob) * binary_tensor`: a DIVISION by keep_prob (one
    correctly rounded fp32 operation), not a multiplication by its rounded reciprocal — 1 ulp apart at keep = 0.75 (VERDICT r5 missing #1);
    its gradient is g * mask / keep by the same RealDiv.  The kernels divide too (csrc: ConvArgs::drop_keep, elementwise.hip)."""
    if keep_prob >= 1.0:
        return x
    if mask is None:
        mask = dropout_mask(tuple(x.shape), keep_prob, seed, stream_id)
    m = torch.from_numpy(mask).to(x.dtype)
    return (x * m) / (float(np.float32(keep_prob)) if x.dtype == torch.float32 else keep_prob)


# ---- batch norm ------------------------------------------------------------------------------------------
def batch_norm(x, gamma, beta, moving_mean, moving_var, is_training, update_moving=True):
    """tf.contrib.layers.batch_norm(decay=.9, center, scale, eps=1e-3, updates_collections=None) (layers.py:95-100),
    fused-batch-norm semantics: normalise with the biased batch variance; the moving variance receives the
    Bessel-corrected one; moving stats are updated in place on every training-mode execution.
    moving_mean / moving_var are updated IN PLACE (torch tensors, no grad)."""
    if is_training:
        red = (0, 1, 2)
        n = x.shape[0] * x.shape[1] * x.shape[2]
        mean = x.mean(dim=red)
        var = ((x - mean) ** 2).mean(dim=red)
        if update_moving:
            with torch.no_grad():
                bessel = n / (n - 1.0) if n > 1 else 1.0
                moving_mean -= (moving_mean - mean.detach().to(moving_mean.dtype)) * (1.0 - BN_DECAY)
                moving_var -= (moving_var - (var.detach() * bessel).to(moving_var.dtype)) * (1.0 - BN_DECAY)
    else:
        mean, var = moving_mean.to(x.dtype), moving_var.to(x.dtype)
    return (x - mean) * (gamma * torch.rsqrt(var + BN_EPS)) + beta


def leaky_relu(x, alpha=LEAK):
    """tf.nn.leaky_relu = max(alpha*x, x); gradient alpha at x == 0 (LeakyReluGrad uses features > 0)"""
    return torch.where(x > 0, x, x * alpha)


def max_pool2(x):
    """tf.nn.max_pool(ksize 2, strides 2, SAME) (layers.py:102-103), even H/W"""
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def pad_channels(x, c_each):
    """tf.pad(x, [[0,0],[0,0],[0,0],[c,c]]) (layers.py:159-160, 181-182)"""
    return F.pad(x, (c_each, c_each))


def PS(x, r, nc):
    """closed form of ops.PS for batch >= 2 (ops.py:3-27): out[n, i*r+u, j*r+v, c] = x[n,i,j, c*r*r + v*r + u]"""
    N, A, B, _ = x.shape
    t = x.reshape(N, A, B, nc, r, r)          # [n,i,j,c,v,u]
    t = t.permute(0, 1, 5, 2, 4, 3)           # [n,i,u,j,v,c]
    return t.reshape(N, A * r, B * r, nc)


# ---- losses ------------------------------------------------------------------------------------------------
def softmax_weighted_loss(logits, y):
    """Full_DRN._softmax_weighted_loss (source_segmenter.py:241-258)"""
    p = torch.softmax(logits, dim=-1)
    ncls = logits.shape[-1]
    raw = 0
    ytot = y.sum()
    for i in range(ncls):
        gti = y[..., i]
        weighted = 1 - (gti.sum() / ytot)
        raw = raw + (-1.0 * weighted * gti * torch.log(torch.clamp(p[..., i], 0.005, 1)))
    return raw.mean()


def dice_loss(logits, y):
    """Full_DRN._dice_loss_fun (source_segmenter.py:260-273)"""
    p = torch.softmax(logits, dim=-1)
    ncls = logits.shape[-1]
    dice = 0
    for i in range(ncls):
        inse = (p[..., i] * y[..., i]).sum()
        l = (p[..., i] * p[..., i]).sum()
        r = y[..., i].sum()
        dice = dice + 2.0 * inse / (l + r + 1e-7)
    return -1.0 * dice / ncls


def pixel_wise_softmax_2(z):
    """layers.pixel_wise_softmax_2 (layers.py:134-138): exp(z)/sum exp(z), NO max subtraction, clipped to +-1e15"""
    e = torch.exp(z)
    return torch.clamp(e / e.sum(dim=-1, keepdim=True), -1e15, 1e15)


def argmax_lowest(p):
    """tf.argmax(p, 3): lowest index on ties"""
    pn = p.detach().cpu().numpy()
    return torch.from_numpy(np.argmax(pn, axis=-1).astype(np.int64))


def dice_eval(compact_pred, y, ncls):
    """lib._dice_eval (lib.py:96-110)"""
    pred = F.one_hot(compact_pred, ncls).to(y.dtype)
    arr = []
    tot = 0
    for i in range(ncls):
        inse = (pred[..., i] * y[..., i]).sum()
        union = pred[..., i].sum() + y[..., i].sum()
        d = 2.0 * inse / (union + 1e-7)
        arr.append(d)
        tot = tot + d
    return tot / ncls, arr


def label_decomp(num_cls, label_vol):
    """lib._label_decomp (lib.py:75-92)"""
    label_vol = np.asarray(label_vol)
    vol = np.zeros(label_vol.shape + (num_cls,), dtype=np.float32)
    for i in range(num_cls):
        vol[..., i][label_vol == i] = 1
    return vol


def l2_loss(w):
    """tf.nn.l2_loss = sum(w^2)/2 (source_segmenter.py:237)"""
    return (w * w).sum() / 2


# ---- optimisers (in-place on torch tensors, no autograd) --------------------------------------------------------
def _one_minus(beta, dtype):
    if dtype == torch.float32:
        return float(np.float32(1.0) - np.float32(beta))
    return 1.0 - beta


def adam_update(w, g, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer._apply_dense: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; w -= lr_t*m/(sqrt(v)+eps)"""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    # TF's ApplyAdam kernel forms (1 - beta) in the variable's dtype: for float32 that is fl32(1) - fl32(beta)
    m += (g - m) * _one_minus(beta1, w.dtype)
    v += (g * g - v) * _one_minus(beta2, w.dtype)
    w -= lr_t * m / (torch.sqrt(v) + eps)


def rmsprop_update(w, g, ms, lr, decay=0.9, eps=1e-10):
    """tf.train.RMSPropOptimizer(momentum=0): ms init 1.0; ms = decay*ms + (1-decay)*g^2; w -= lr*g/sqrt(ms+eps)"""
    ms += (g * g - ms) * _one_minus(decay, w.dtype)
    w -= lr * g / torch.sqrt(ms + eps)


def momentum_update(w, g, acc, lr, momentum):
    """tf.train.MomentumOptimizer: acc = momentum*acc + g; w -= lr*acc"""
    acc.mul_(momentum).add_(g)
    w -= lr * acc


# ---- restatement of the lane arithmetic of csrc/conv_small.hip (16x16x4 fp32 MFMA tiles) ---------------------------------------------
# Not a reference-side algorithm: the reference has no kernels.  It restates, in numpy, WHICH products each lane of a 64-wide wavefront
# feeds to v_mfma_f32_16x16x4_f32 in conv_n16_kernel / wgrad_n16_kernel (operand layouts as the CDNA ISA defines them), so that the index
# maps — the k-permutation of the channel quarters, the 8x32 output tile, the patch offsets — are pinned against the plain convolution on
# the CPU (tests/test_host.py) and not only through the GPU parity tests.
def _mfma_16x16x4(a_lane, b_lane, acc):
    """a_lane[l] = A[m = l % 16][k = l // 16], b_lane[l] = B[k = l // 16][n = l % 16]; acc[4g + i][n] accumulates D (lane (g, n), register i)"""
    A = a_lane.reshape(4, 16).T          # [m][k]
    B = b_lane.reshape(4, 16)            # [k][n]
    return acc + A @ B


def conv3x3_n16_by_mfma_tiles(x, w, pad=1):
    """x [N,H,W,C] (C = 16 or 32), w [3,3,C,16] float64 numpy -> y [N,OH,OW,16] computed tile by tile exactly like conv_n16_kernel:
    workgroup tile 8 x 32 pixels, wave = 2 rows = four 16-pixel MFMA tiles, per tap and 16-channel half ONE 4-float read per lane
    (lane (p, g): pixel p, channels 16h + 4g .. +3) feeding MFMAs j = 0..3 which contract the channels {16h + 4g + j : g = 0..3}."""
    N, H, W, C = x.shape
    OH, OW = H + 2 * pad - 2, W + 2 * pad - 2
    xp = np.zeros((N, H + 2 * pad + 8, W + 2 * pad + 32, C))
    xp[:, pad:pad + H, pad:pad + W] = x
    y = np.zeros((N, OH, OW, 16))
    lane = np.arange(64)
    p, g = lane % 16, lane // 16
    for n in range(N):
        for oh0 in range(0, OH, 8):
            for ow0 in range(0, OW, 32):
                for wave in range(4):
                    for q in range(4):
                        row, col0 = 2 * wave + (q >> 1), 16 * (q & 1)
                        acc = np.zeros((16, 16))
                        for tap in range(9):
                            r, s = tap // 3, tap % 3
                            for h in range(C // 16):
                                av = xp[n, oh0 + row + r, ow0 + col0 + p + s]          # [64 lanes][C]: the lane's patch pixel
                                for j in range(4):
                                    a_lane = av[lane, 16 * h + 4 * g + j]
                                    b_lane = w[r, s, 16 * h + 4 * g + j, p]
                                    acc = _mfma_16x16x4(a_lane, b_lane, acc)
                        for gi in range(4):
                            for i in range(4):
                                oh, ow = oh0 + row, ow0 + col0 + 4 * gi + i
                                if oh < OH and ow < OW:
                                    y[n, oh, ow] = acc[4 * gi + i]
    return y


def wgrad3x3_n16_by_mfma_tiles(x, dy, pad=1):
    """x [N,H,W,C] (C multiple of 16), dy [N,OH,OW,K] (K multiple of 16) -> dW [3,3,C,K] like wgrad_n16_kernel: M = 16 channels of a tap,
    N = 16 filters, K-dimension of the MFMA = 4 consecutive pixels of a row (lane quarter g), accumulated over all pixel groups."""
    N, H, W, C = x.shape
    _, OH, OW, K = dy.shape
    xp = np.zeros((N, H + 2 * pad + 8, W + 2 * pad + 32, C))
    xp[:, pad:pad + H, pad:pad + W] = x
    dyp = np.zeros((N, OH + 8, OW + 32, K))
    dyp[:, :OH, :OW] = dy
    dw = np.zeros((3, 3, C, K))
    lane = np.arange(64)
    p, g = lane % 16, lane // 16
    for kg in range(K // 16):
        for tap in range(9):
            r, s = tap // 3, tap % 3
            for c in range(C // 16):
                acc = np.zeros((16, 16))
                for n in range(N):
                    for oh in range(0, OH):
                        for ow0 in range(0, OW, 4):
                            a_lane = xp[n, oh + r, ow0 + g + s, 16 * c + p]
                            b_lane = dyp[n, oh, ow0 + g, 16 * kg + p]
                            acc = _mfma_16x16x4(a_lane, b_lane, acc)
                dw[r, s, 16 * c:16 * c + 16, 16 * kg:16 * kg + 16] = acc        # D[m = channel][n = filter]
    return dw


def wgrad_ring_walk(N, H, W, C, OH, OW, stride, dil, pad_t, pad_l, r, s, c, p_first, nstages):
    """The incremental input-coordinate walk of csrc/conv_igemm.hip::conv_wgrad_ring_kernel, restated: a loader row starts at output
    pixel p_first and advances 32 output pixels per stage; (ih, iw, byte offset) of its filter tap (r, s) and channel c are carried with
    one row wrap and one image wrap per step (needs OW >= 32).  Returns [(ok, offset)] per stage; `ok` False = the tap reads padding."""
    BK = 32
    OHW = OH * OW
    l_dh, l_dw = r * dil - pad_t, s * dil - pad_l
    iw_lim, ih_lim = OW * stride + l_dw, OH * stride + l_dh
    step_w, step_off = BK * stride, BK * stride * C * 4
    wrap_w, wrap_w_off = OW * stride, stride * (W - OW) * C * 4
    wrap_h, wrap_h_off = OH * stride, (H - OH * stride) * W * C * 4
    n = p_first // OHW
    rem = p_first - n * OHW
    oh, ow = rem // OW, rem % OW
    ih, iw = oh * stride + l_dh, ow * stride + l_dw
    off = (((n * H + ih) * W + iw) * C + c) * 4
    out = []
    for _ in range(nstages):
        out.append((0 <= ih < H and 0 <= iw < W, off))
        iw += step_w
        off += step_off
        if iw >= iw_lim:
            iw -= wrap_w
            ih += stride
            off += wrap_w_off
        if ih >= ih_lim:
            ih -= wrap_h
            off += wrap_h_off
    return out


# ---- Winograd F(2x2, 3x3) restated the way the HIP path computes it (csrc/conv_wino.hip) --------------------------------------------
# 1-D: Y = A^T [ (G g) * (B^T d) ], d = 4 inputs, g = 3 taps, Y = 2 outputs (Lavin & Gray 2016, the minimal filtering algorithm F(2,3)).
WINO_BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
WINO_G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
WINO_AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
# F(4,3) on the interpolation points (0, 1, -1, 1/2, -2, inf): d = 6 inputs, Y = 4 outputs, 36 instead of 144 multiplications per 4x4 output
# tile in 2-D.  Toom-Cook construction (A^T = E_4^T, G = E_3 with row j divided by N_j = prod_{l != j} (p_j - p_l), B^T = V^-T with row j
# multiplied by N_j; tools/wino_f43_study.py generates them from the points and measures the float32 error of several point sets: this one
# lands at the direct fp32 convolution's 3e-6..5e-6 of max|y| on the 256- / 512-channel layers, Lavin & Gray's (0, +-1, +-2) at 1e-5).
# B^T and A^T are exact in binary; G's thirds and fifteenths are rounded once (the kernels use the same float32 constants).
WINO4_BT = np.array([[1, -1.5, -2, 1.5, 1, 0], [0, -1, .5, 2.5, 1, 0], [0, 1, -2.5, .5, 1, 0], [0, -2, -1, 2, 1, 0], [0, .5, -1, -.5, 1, 0],
                     [0, 1, -1.5, -2, 1.5, 1]], np.float64)
WINO4_G = np.array([[1, 0, 0], [1 / 3, 1 / 3, 1 / 3], [-1 / 3, 1 / 3, -1 / 3], [-16 / 15, -8 / 15, -4 / 15], [1 / 15, -2 / 15, 4 / 15], [0, 0, 1]],
                   np.float64)
WINO4_AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, .5, -2, 0], [0, 1, 1, .25, 4, 0], [0, 1, -1, .125, -8, 1]], np.float64)


def wino_mats(m):
    """(B^T, G, A^T) of F(m x m, 3x3), m = 2 or 4"""
    return (WINO_BT, WINO_G, WINO_AT) if m == 2 else (WINO4_BT, WINO4_G, WINO4_AT)


def wino_tiles(N, Ho, Wo, dil, m=2):
    """tile enumeration of csrc/conv_wino.hip: a dilation-d stride-1 3x3 convolution is d*d independent dense 3x3 convolutions of the
    sub-images (a + d u, b + d v); every OUTPUT sub-image is cut into m x m tiles (m = 2: F(2x2, 3x3), m = 4: F(4x4, 3x3)).
    Returns (Hso, Wso, th, tw, T)."""
    assert Ho % dil == 0 and Wo % dil == 0
    Hs, Ws = Ho // dil, Wo // dil
    th, tw = -(-Hs // m), -(-Ws // m)
    return Hs, Ws, th, tw, N * dil * dil * th * tw


def wino_tile_coords(t, dil, th, tw):
    """tile id -> (image, sub-image phase a, b, tile row, tile column): t = (((n d + a) d + b) th + ti) tw + tj"""
    tj = t % tw
    t //= tw
    ti = t % th
    t //= th
    b = t % dil
    t //= dil
    a = t % dil
    return t // dil, a, b, ti, tj


# ---- the split-bf16 ("x3") GEMM of csrc/conv_wino_x3.hip, restated ------------------------------------------------------------------
def bf16_rne(a):
    """float32 -> nearest bfloat16 (ties to even; v_cvt_pk_bf16_f32), returned as float32"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def split3_bf16(a):
    """a == hi + mid + lo EXACTLY, each a bfloat16 value: hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid) (3 x 8 significand
    bits; both differences are exact in float32) — what wino_in_kernel<M, true> / wino_filter_kernel<.., true> store"""
    a = np.asarray(a, np.float32)
    hi = bf16_rne(a)
    r1 = a - hi
    mid = bf16_rne(r1)
    lo = bf16_rne(r1 - mid)
    return hi, mid, lo


X3_TERMS = ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0))      # (plane of A, plane of B): smallest first; mid.lo, lo.mid, lo.lo (<= 2^-26 of the product) dropped


def gemm_x3_np(A, B, chunk=64, block=16):
    """A [.., M, C] @ B [.., C, K] the way wino_gemm_x3_kernel contracts it: both operands as three bf16 planes, the six kept plane products
    of a 16-deep MFMA block summed (each product is exact in float32; the block sum here in float64, rounded once — the optimistic model
    of one v_mfma_f32_32x32x16_bf16) into a float32 accumulator that is RESTARTED every `chunk` channels, the chunk sums added into a
    second float32 accumulator.  C must be a multiple of `chunk`."""
    Ap = [p.astype(np.float64) for p in split3_bf16(A)]
    Bp = [p.astype(np.float64) for p in split3_bf16(B)]
    C = A.shape[-1]
    assert C % chunk == 0 and chunk % (2 * block) == 0
    total = np.zeros(A.shape[:-1] + (B.shape[-1],), np.float32)
    for c0 in range(0, C, chunk):
        cur = np.zeros_like(total)
        for s0 in range(c0, c0 + chunk, 2 * block):          # a 32-channel stage = two 16-deep slices; within a slice the six terms in order
            for k0 in (s0, s0 + block):
                for (i, j) in X3_TERMS:
                    cur = (cur.astype(np.float64) + Ap[i][..., k0:k0 + block] @ Bp[j][..., k0:k0 + block, :]).astype(np.float32)
        total = total + cur
    return total


def conv3x3_winograd_np(x, w, dil=1, flip_transpose=False, dtype=np.float32, pad=None, m=2, x3=False):
    """Stride-1 3x3 (dilated) convolution with zero padding `pad` on every side (default dil = TF SAME; 0 = VALID, the model's g10 on its
    mirror-padded input; 2 dil = the data gradient of a VALID convolution) in the four steps of the HIP path, intermediates in `dtype`:
      V[pos][t][c] = (B^T d B)[pos]   input transform of the 4x4 patch of tile t (zeros outside the image)
      U[pos][c][k] = (G g G^T)[pos]   filter transform (flip_transpose: of the data gradient's filter w'[r][s][k][c] = w[2-r][2-s][c][k])
      M[pos] = V[pos] @ U[pos]        16 GEMMs [T x C] x [C x K]
      y tile  = A^T M A               output transform, scattered to the tile's 2x2 pixels (those inside the image)
    m = 4: F(4x4, 3x3) — 6x6 patches, 36 transform points, 4x4 output tiles, otherwise the same four steps.
    x3: the GEMMs on split-bf16 operands with chunked accumulation (gemm_x3_np; float32 only, C % 64 == 0).
    x: [N][H][W][C] numpy, w: [3][3][C][K] numpy."""
    x = np.asarray(x, dtype)
    w = np.asarray(w, dtype)
    if flip_transpose:
        w = np.ascontiguousarray(w[::-1, ::-1].transpose(0, 1, 3, 2))
    pad = dil if pad is None else pad
    assert pad % dil == 0 and pad <= 2 * dil
    ps = pad // dil
    N, Hi, Wi, C = x.shape
    assert Hi % dil == 0 and Wi % dil == 0
    Ho, Wo = Hi + 2 * pad - 2 * dil, Wi + 2 * pad - 2 * dil
    Hsi, Wsi = Hi // dil, Wi // dil
    K = w.shape[3]
    Hso, Wso, th, tw, T = wino_tiles(N, Ho, Wo, dil, m)
    n_ = m + 2
    BT, G, AT = (a_.astype(dtype) for a_ in wino_mats(m))
    V = np.zeros((n_ * n_, T, C), dtype)
    for t in range(T):
        n, a, b, ti, tj = wino_tile_coords(t, dil, th, tw)
        d = np.zeros((n_, n_, C), dtype)
        for i in range(n_):
            for j in range(n_):
                u, v = m * ti - ps + i, m * tj - ps + j
                if 0 <= u < Hsi and 0 <= v < Wsi:
                    d[i, j] = x[n, a + dil * u, b + dil * v]
        # rows first, then columns (the kernel's order of additions)
        r = np.einsum("ip,pjc->ijc", BT, d).astype(dtype)
        V[:, t, :] = np.einsum("ipc,jp->ijc", r, BT).astype(dtype).reshape(n_ * n_, C)
    g1 = np.einsum("ir,rsck->isck", G, w).astype(dtype)
    U = np.einsum("isck,js->ijck", g1, G).astype(dtype).reshape(n_ * n_, C, K)
    Mm = gemm_x3_np(V, U) if x3 else np.stack([V[p] @ U[p] for p in range(n_ * n_)]).astype(dtype)
    y = np.zeros((N, Ho, Wo, K), dtype)
    for t in range(T):
        n, a, b, ti, tj = wino_tile_coords(t, dil, th, tw)
        mt = Mm[:, t, :].reshape(n_, n_, K)
        r = np.einsum("pi,ijk->pjk", AT, mt).astype(dtype)
        o = np.einsum("pjk,qj->pqk", r, AT).astype(dtype)
        for p in range(m):
            for q in range(m):
                u, v = m * ti + p, m * tj + q
                if u < Hso and v < Wso:
                    y[n, a + dil * u, b + dil * v] = o[p, q]
    return y


def wgrad3x3_winograd_np(x, dy, dil=1, dtype=np.float32, pad=None, nsplit=1, m=2, x3=False):
    """Filter gradient of the same convolution the way csrc/conv_wino.hip computes it — the transposition of F(2x2, 3x3):
      V[pos][t][c] = (B^T d B)[pos]                  the forward's input transform
      Y[pos][t][k] = (A y A^T)[pos]                  the 2x2 tile of dy spread to the 16 transform points (zeros for pixels outside the image)
      S[z][pos]    = V[pos][rows of split z]^T @ Y[pos][rows of split z]      16 GEMMs [C x T] x [T x K], reduction over tiles split nsplit ways
      dW           = G^T (sum_z S[z]) G              4x4 -> 3x3
    x: [N][H][W][C], dy: [N][Ho][Wo][K] numpy -> [3][3][C][K]."""
    x, dy = np.asarray(x, dtype), np.asarray(dy, dtype)
    pad = dil if pad is None else pad
    ps = pad // dil
    N, Hi, Wi, C = x.shape
    Ho, Wo, K = dy.shape[1], dy.shape[2], dy.shape[3]
    assert Ho == Hi + 2 * pad - 2 * dil and Wo == Wi + 2 * pad - 2 * dil
    Hsi, Wsi = Hi // dil, Wi // dil
    Hso, Wso, th, tw, T = wino_tiles(N, Ho, Wo, dil, m)
    n_ = m + 2
    BT, G, AT = (a_.astype(dtype) for a_ in wino_mats(m))
    A = np.ascontiguousarray(AT.T)
    V = np.zeros((n_ * n_, T, C), dtype)
    Y = np.zeros((n_ * n_, T, K), dtype)
    for t in range(T):
        n, a, b, ti, tj = wino_tile_coords(t, dil, th, tw)
        d = np.zeros((n_, n_, C), dtype)
        for i in range(n_):
            for j in range(n_):
                u, v = m * ti - ps + i, m * tj - ps + j
                if 0 <= u < Hsi and 0 <= v < Wsi:
                    d[i, j] = x[n, a + dil * u, b + dil * v]
        V[:, t, :] = np.einsum("ipc,jp->ijc", np.einsum("ip,pjc->ijc", BT, d).astype(dtype), BT).astype(dtype).reshape(n_ * n_, C)
        y = np.zeros((m, m, K), dtype)
        for p in range(m):
            for q in range(m):
                u, v = m * ti + p, m * tj + q
                if u < Hso and v < Wso:
                    y[p, q] = dy[n, a + dil * u, b + dil * v]
        Y[:, t, :] = np.einsum("iqk,jq->ijk", np.einsum("ip,pqk->iqk", A, y).astype(dtype), A).astype(dtype).reshape(n_ * n_, K)
    rows = -(-T // nsplit)
    S = np.zeros((n_ * n_, C, K), dtype)
    for z in range(nsplit):
        sl = slice(z * rows, min((z + 1) * rows, T))
        if x3:          # split-bf16 GEMM over the tiles of this split, zero-padded to a multiple of 64 like the kernel's operands
            Vz, Yz = V[:, sl], Y[:, sl]
            padn = (-Vz.shape[1]) % 64
            Vz = np.concatenate([Vz, np.zeros((Vz.shape[0], padn, C), dtype)], 1)
            Yz = np.concatenate([Yz, np.zeros((Yz.shape[0], padn, K), dtype)], 1)
            S += gemm_x3_np(np.ascontiguousarray(Vz.transpose(0, 2, 1)), Yz)
        else:
            S += np.stack([V[p, sl].T @ Y[p, sl] for p in range(n_ * n_)]).astype(dtype)
    S = S.reshape(n_, n_, C, K)
    return np.einsum("rjck,js->rsck", np.einsum("ir,ijck->rjck", G, S).astype(dtype), G).astype(dtype)
