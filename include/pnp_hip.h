/*
 * pnp_hip.h — C-ABI of libpnp_hip.so: the MI355X (gfx950) kernels behind the PnP-AdaNet
 * training hot path (dilated-residual segmenter fwd/bwd + Wasserstein critics).
 *
 * The reference (carrenD/Medical-Cross-Modality-Domain-Adaptation) has NO FFI of its own: every op is
 * a TensorFlow-1.4 graph op created in layers.py / ops.py / source_segmenter.py / adversarial.py.
 * Each entry point below therefore names the reference call site whose TF op it replaces
 * (file:line into /root/reference).  The Python host side binds these with ctypes
 * (see INTEGRATION.md); nothing in the signatures is a torch type.
 *
 * Conventions
 *   - tensors: dense float32, activations NHWC, filters HWIO (exactly the reference layout)
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated inside;
 *     scratch comes from a caller-provided workspace (size from pnp_*_workspace_bytes)
 *   - `stream` is a hipStream_t passed as void*; all calls are asynchronous w.r.t. the host
 *   - return 0 on success, <0 on error (PNP_E*), message via pnp_last_error(); never throws
 */
#ifndef PNP_HIP_H
#define PNP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNP_OK 0
#define PNP_EINVAL (-1)   /* bad argument / unsupported geometry */
#define PNP_ELAUNCH (-2)  /* hip launch error */
#define PNP_EWORKSPACE (-3) /* workspace too small */
#define PNP_ECOMM (-4)    /* RCCL missing or an RCCL call failed */

/* element types named at the ABI (convolution operand storage, collectives) */
#define PNP_DTYPE_F32 0
#define PNP_DTYPE_BF16 1
#define PNP_DTYPE_F64 2

#define PNP_PAD_ZERO 0      /* tf.nn.conv2d(padding='SAME') zero padding, layers.py:18,67 */
#define PNP_PAD_SYMMETRIC 1 /* tf.pad(x, k//2, 'SYMMETRIC') + VALID conv, layers.py:19-24,68-73 */

int pnp_abi_version(void);
const char* pnp_last_error(void);
/* cu_count, max clock (kHz), LDS bytes per CU, gcn arch name (<=63 chars) of `device` */
int pnp_device_info(int device, int* cu_count, int* clock_khz, int* lds_bytes, char* arch, int arch_len);

/* Kernel-level timing for the roofline figures of bench.py: while a class is enabled, the library records HIP events on the
 * launch stream around every launch of that class's main kernel (not around its helper kernels: filter flips, partial sums).
 * pnp_prof_summary waits for the events, returns one row per kernel symbol (named as rocprofv3 --kernel-trace prints it) with the
 * launch count, the summed duration and the summed ALGORITHMIC flops / bytes (2*N*OH*OW*R*S*C*K; 4*(|x|+|y|+|w|)), and clears
 * the records.  Returns the number of distinct symbols (may exceed max_rows; only max_rows are written). */
#define PNP_PROF_CONV_FWD 1   /* MFMA forward convolutions */
#define PNP_PROF_CONV_DGRAD 2 /* MFMA data gradients */
#define PNP_PROF_CONV_WGRAD 4 /* MFMA filter gradients */
#define PNP_PROF_CONV_DIRECT 8 /* vector-ALU convolutions for K <= 16 */
typedef struct pnp_prof_row {
    char name[128];
    int64_t launches;
    double ms, flops, bytes;
} pnp_prof_row;
int pnp_prof_enable(int32_t mask);
int pnp_prof_summary(pnp_prof_row* rows, int32_t max_rows);

/* Geometry shared by the three conv entry points (all describe the FORWARD convolution):
 *   x [N,H,W,C]  w [R,S,C,K]  y [N,OH,OW,K]
 *   y[n,oh,ow,k] = sum_{r,s,c} xpad[n, oh*stride - pad_t + r*dil, ow*stride - pad_l + s*dil, c] * w[r,s,c,k]
 *   pad_mode PNP_PAD_ZERO: out-of-range taps read 0 (TF SAME; pad_t/pad_l = the TF "pad before" amounts)
 *   pad_mode PNP_PAD_SYMMETRIC: out-of-range taps mirror including the edge (tf.pad SYMMETRIC), pad_t=pad_l=k//2
 * Dropout (tf.nn.dropout, layers.py:25,74,93) is fused in the forward epilogue:
 *   y *= mask(seed, stream_id, flat_index) / keep_prob    (keep_prob >= 1 disables it)
 * mask is the counter-hash stream documented at pnp_dropout (below). */
typedef struct pnp_conv_geom {
    int32_t N, H, W, C;     /* input */
    int32_t K, R, S;        /* filter count and size */
    int32_t OH, OW;         /* output spatial size */
    int32_t stride, dil;    /* same in both spatial dims (reference only uses square) */
    int32_t pad_t, pad_l;   /* pad before (top / left) */
    int32_t pad_mode;       /* PNP_PAD_* */
    int32_t dtype;          /* PNP_DTYPE_F32: the reference's arithmetic.  PNP_DTYPE_BF16 (BASELINE configs[4]): both operands of the
                             * contraction are rounded to bfloat16 (nearest-even) as they are staged, products accumulate in fp32;
                             * x / w / y / gradients stay float32 in memory (w = the fp32 master weights).  Layers off the MFMA fast
                             * paths (C not a multiple of 32, K <= 16, strided filter gradients) compute in fp32 either way. */
} pnp_conv_geom;

/* replaces tf.nn.conv2d (layers.py:18,24,67,73) and tf.nn.atrous_conv2d (layers.py:86,92) + tf.nn.dropout */
int pnp_conv2d_fwd(const float* x, const float* w, float* y, const pnp_conv_geom* g,
                   float keep_prob, uint64_t seed, uint32_t stream_id, void* stream);
/* Same with a workspace: layers with few output pixels (the critics' 4x4 / 2x2 maps, small batches) leave most of the 256 CUs
 * without a tile; given pnp_conv2d_fwd_workspace_bytes(g) bytes the reduction over R*S*C is split across workgroups and the
 * partial sums (fixed order: deterministic) pass through the workspace; dropout is then applied by the summing kernel.
 * workspace may be NULL / 0 bytes (== pnp_conv2d_fwd). */
size_t pnp_conv2d_fwd_workspace_bytes(const pnp_conv_geom* g);
int pnp_conv2d_fwd_ws(const float* x, const float* w, float* y, const pnp_conv_geom* g,
                      float keep_prob, uint64_t seed, uint32_t stream_id,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Training-mode conv -> dropout -> batch norm: the forward convolution leaves per-(pixel tile, wave row) partial sums of (y - shift[k])
 * and (y - shift[k])^2 over its output rows, so that the batch statistics cost no second pass over the activation
 * (pnp_bn_stats_finish combines them in double).  shift [K] (nullable = 0): any per-channel constant near the mean, e.g. the moving
 * mean — it only conditions the variance.  pnp_conv2d_fwd_stats_parts(g) = number of partial rows, 0 when this geometry's forward
 * cannot provide them (vector-ALU narrow-output kernels, reduction-split tiny layers): run pnp_bn_stats on the output instead.
 * parts: [parts][2][K] floats. */
int32_t pnp_conv2d_fwd_stats_parts(const pnp_conv_geom* g);
int pnp_conv2d_fwd_stats(const float* x, const float* w, float* y, const pnp_conv_geom* g,
                         float keep_prob, uint64_t seed, uint32_t stream_id,
                         const float* shift /*nullable*/, float* parts, size_t parts_bytes, void* stream);
/* The same with a workspace of pnp_conv2d_fwd_workspace_bytes(g) bytes.  Wide stride-1 3x3 layers (the segmenter's 256- / 512-channel
 * groups, source_segmenter.py:140-200) may be given to the Winograd F(2x2, 3x3) route (csrc/conv_wino.hip: 2.25x fewer multiplications,
 * transformed tensors through the workspace; same result up to fp32 rounding, ~1e-6 relative); its partial rows are the tile slabs of its
 * output transform, so the count comes from pnp_conv2d_fwd_stats_ws_parts(g).  Every other layer: identical to pnp_conv2d_fwd_stats.
 * pnp_conv2d_wino_chosen(g, kind) tells which route a layer takes (kind 0: forward; 1: data gradient; 2: filter gradient — g = the FORWARD geometry);
 * environment PNP_WINOGRAD = 0 never / 1 where the cost model says it pays / 2 wherever the geometry allows. */
int32_t pnp_conv2d_fwd_stats_ws_parts(const pnp_conv_geom* g);
int pnp_conv2d_fwd_stats_ws(const float* x, const float* w, float* y, const pnp_conv_geom* g,
                            float keep_prob, uint64_t seed, uint32_t stream_id,
                            const float* shift /*nullable*/, float* parts, size_t parts_bytes,
                            void* workspace, size_t workspace_bytes, void* stream);
int32_t pnp_conv2d_wino_chosen(const pnp_conv_geom* g, int32_t kind);   /* 0: direct kernels; else the route's output tile edge (2 or 4) */
/* sets the route policy at run time (0 / 1 / 2 as PNP_WINOGRAD; < 0: read only) and returns the previous one */
int32_t pnp_conv2d_wino_mode(int32_t mode);
/* Round 5: the route has two output tiles.  F(4x4, 3x3) — 36 multiplications per 4x4 output tile instead of the direct sum's 144
 * (F(2x2): 64), transformed tensors 2.25x the activations (F(2x2): 4x) — on the interpolation points (0, 1, -1, 1/2, -2, inf), whose
 * float32 error against the float64 convolution is the direct kernel's (3e-6..5e-6 of max|y| on the 256- / 512-channel layers;
 * profiles/r05_wino_f43_tolerance.txt).  tile = 2: F(2x2) only; 4 (the default, environment PNP_WINOGRAD_TILE): F(4x4) where its planner
 * takes the layer, else F(2x2), else the direct kernels; tile < 2: read only.  Returns the previous value. */
int32_t pnp_conv2d_wino_tile(int32_t tile);
/* Round 6: arithmetic of the route's forward / data-gradient GEMMs.  0: the fp32 matrix pipe (v_mfma_f32_32x32x2_f32).  1 (the default,
 * environment PNP_WINOGRAD_X3) where it pays (reductions over >= 256 channels), 2 wherever the shapes allow (C % 64 == 0): split-bf16 operands — every transformed value is stored as three bf16 planes whose sum IS the fp32 value, six of the
 * nine plane products (all terms above 2^-26 of the product) run on v_mfma_f32_32x32x16_bf16 (16x the fp32 pipe's rate) with fp32
 * accumulation in 64-channel chunks (csrc/conv_wino_x3.hip; tools/wino_bf16x3_study.py -> profiles/r06_wino_bf16x3_tolerance.txt: the
 * whole layer's error against float64 falls from 4e-6..6e-6 to 1e-6..2e-6, because the chunked chain is shorter than the fp32 pipe's).
 * mode < 0: read only.  Returns the previous mode.  Workspace queries and the transformed-filter cache follow the mode in force. */
int32_t pnp_conv2d_wino_x3(int32_t mode);
/* Round 6: the narrow layers of the route (3x3, stride 1, 32 or 64 input channels, 64 or 128 filters, output a multiple of 16 x 16) as DIRECT
 * split-bf16 convolutions (csrc/conv_x3_direct.hip): no transforms — a tile's halo patch is split into three bf16 planes once and the nine
 * taps are LDS offsets; six plane products per fragment pair on v_mfma_f32_32x32x16_bf16, one fp32 chain of 9 C terms (8e-7 of max|ref|).
 * Replaces, for those layers, the F(4x4) route that is bound by its 2.25x transformed tensors there (reference adversarial.py:337-366, the
 * critics' 64-channel blocks at 256^2 / 128^2).  0: off, 1 (default): where a launch fills the chip (>= 256 tile x filter-block items), 2: wherever the
 * shapes allow (environment PNP_X3_DIRECT); mode < 0: read only.  Returns the previous mode. */
int32_t pnp_conv2d_x3_direct(int32_t mode);
/* Transformed-filter cache of the route.  U = G g G^T (36 C K values per filter and pass: fp32, or three bf16 planes) only changes when the filter does: the caller
 * lends one buffer per (filter, pass) and reports weight writes; a launch whose filter has a valid entry skips wino_filter_kernel (the
 * frozen source segmenter / shared half of adversarial.py:839-882 never pay it again, a trained layer once per update instead of once per
 * pass).  pnp_conv2d_wino_filter_bytes(C, K): size of an entry that serves either tile (0: this shape never takes the route).
 * pnp_conv2d_wino_filter_bind(w, kind, U, bytes): kind 0 forward / 1 data gradient; U = null withdraws the entry, w = null all of them;
 * the buffer must outlive the binding.  pnp_weights_changed(lo, hi): the floats in [lo, hi) were (or are queued to be) written —
 * entries of filters inside are stale; lo = null: every entry.  Launches recorded into a hipGraph never touch the cache.
 * pnp_conv2d_wino_filter_stats: transforms skipped / run into an entry since the last reset. */
size_t pnp_conv2d_wino_filter_bytes(int32_t C, int32_t K);
int pnp_conv2d_wino_filter_bind(const float* w, int32_t kind, float* U, size_t bytes);
void pnp_weights_changed(const void* lo, const void* hi);
void pnp_conv2d_wino_filter_stats(int64_t* hits, int64_t* fills, int32_t reset);
/* the filter gradient (pnp_conv2d_wgrad / _wgrad_acc, given pnp_conv2d_wgrad_workspace_bytes) has its own switch (PNP_WINOGRAD_WGRAD,
 * same values); pnp_conv2d_wino_chosen(g, 2) tells its route */
int32_t pnp_conv2d_wino_wgrad_mode(int32_t mode);

/* Inference-mode conv -> dropout -> batch norm -> (+ shortcut) -> leaky-ReLU in ONE kernel (the monitoring forwards of
 * source_segmenter.py:525-570 / adversarial.py:948-991, every frozen-BN forward of the GAN steps, Trainer.test_eval):
 *   y = act( drop(conv(x,w)) * scale[k] + shift[k] + pad_channels(shortcut) ),  scale / shift from pnp_bn_fold.
 * shortcut [N*OH*OW, Cs] is zero-padded (K-Cs)/2 channels on each side (layers.py:159-165); alpha < 0: no activation. */
int pnp_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float* scale, float* shift,
                int32_t C, float eps, void* stream);
int pnp_conv2d_fwd_bn(const float* x, const float* w, float* y, const pnp_conv_geom* g,
                      float keep_prob, uint64_t seed, uint32_t stream_id,
                      const float* scale, const float* shift, const float* shortcut /*nullable*/, int32_t Cs, float alpha,
                      void* stream);
/* the same with a workspace of pnp_conv2d_fwd_workspace_bytes(g) bytes (nullable / too small: == pnp_conv2d_fwd_bn): the layers the
 * planner gives to the Winograd route take it, epilogue included */
int pnp_conv2d_fwd_bn_ws(const float* x, const float* w, float* y, const pnp_conv_geom* g,
                         float keep_prob, uint64_t seed, uint32_t stream_id,
                         const float* scale, const float* shift, const float* shortcut /*nullable*/, int32_t Cs, float alpha,
                         void* workspace, size_t workspace_bytes, void* stream);

/* gradient w.r.t. the conv input (TF autodiff of the ops above; Conv2DBackpropInput).
 * dy is the gradient w.r.t. the conv accumulator (i.e. AFTER the dropout mask has been applied by the caller).
 * workspace: pnp_conv2d_dgrad_workspace_bytes(g). */
size_t pnp_conv2d_dgrad_workspace_bytes(const pnp_conv_geom* g);
int pnp_conv2d_dgrad(const float* dy, const float* w, float* dx, const pnp_conv_geom* g,
                     void* workspace, size_t workspace_bytes, void* stream);
/* dx = data gradient + residual (residual [N,H,W,C], not aliasing dx): the gradient that reaches the conv input a second way — the
 * shortcut of layers.residual_block / DR_block (layers.py:145-189), where TF's autodiff emits an AddN.  Added in the epilogue of the
 * stride-1 MFMA kernels; other geometries add it with one more pass. */
int pnp_conv2d_dgrad_add(const float* dy, const float* w, const float* residual, float* dx, const pnp_conv_geom* g,
                         void* workspace, size_t workspace_bytes, void* stream);

/* gradient w.r.t. the filter (Conv2DBackpropFilter). dw [R,S,C,K] is overwritten. */
size_t pnp_conv2d_wgrad_workspace_bytes(const pnp_conv_geom* g);
int pnp_conv2d_wgrad(const float* x, const float* dy, float* dw, const pnp_conv_geom* g,
                     void* workspace, size_t workspace_bytes, void* stream);
/* dw += filter gradient: written straight into a gradient buffer that may already hold a contribution (filters shared by two
 * passes, adversarial.py:196,235; TF's AddN of the per-use gradients) — no separate accumulation kernel */
int pnp_conv2d_wgrad_acc(const float* x, const float* dy, float* dw, const pnp_conv_geom* g,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Naive one-thread-per-output direct convolution (fp32 fmaf chain in r,s,c order). On-device
 * cross-check for the MFMA kernels at sizes the CPU oracle cannot reach; not used by the product path. */
int pnp_conv2d_fwd_naive(const float* x, const float* w, float* y, const pnp_conv_geom* g, void* stream);

/* tf.nn.dropout (layers.py:25,74,93): y = x * keep_mask / keep.  TF's own RNG stream is not reproducible outside TF, so the
 * mask stream is specified here (csrc/pnp_common.h; restated in numpy in oracle/tf_ops.py):
 *   keep_mask(idx) = (fmix32((idx * 0xCC9E2D51) ^ key(seed, stream_id)) >> 8) >= round((1 - keep) * 2^24)
 * with fmix32 = the murmur3 finaliser and idx the flat element index.  Also used for the backward of the fused epilogue. */
int pnp_dropout(const float* x, float* y, size_t n, float keep_prob, uint64_t seed, uint32_t stream_id, void* stream);

/* tf.contrib.layers.batch_norm(decay=.9, eps=1e-3, updates_collections=None) (layers.py:95-100), fused batch-norm semantics.
 * x,y: [P,C] (P = N*H*W).  stats: mean[C], var[C] (biased).  ws: pnp_bn_workspace_bytes(P,C). */
size_t pnp_bn_workspace_bytes(int64_t P, int32_t C);
int pnp_bn_stats(const float* x, float* mean, float* var, int64_t P, int32_t C,
                 void* workspace, size_t workspace_bytes, void* stream);
/* pnp_bn_stats + pnp_bn_update_moving in one pass (the per-replica training-mode forward: one launch less per BN layer) */
int pnp_bn_stats_update(const float* x, float* mean, float* var, float* moving_mean, float* moving_var, int64_t P, int32_t C,
                        float decay, void* workspace, size_t workspace_bytes, void* stream);
/* mean / biased variance from the partials of pnp_conv2d_fwd_stats (same shift), optionally followed by the moving-average update
 * (moving_mean / moving_var nullable together; shift may alias moving_mean).  `parts` is CONSUMED: long lists (>= 512 partials) are
 * first compacted in place by many workgroups (double sums written back as float high/low pairs), then combined. */
int pnp_bn_stats_finish(float* parts, int32_t nparts, const float* shift, float* mean, float* var,
                        float* moving_mean, float* moving_var, int64_t P, int32_t C, float decay, void* stream);
/* moving_mean -= (1-decay)*(moving_mean-mean); moving_var likewise with var*P/(P-1) (Bessel) */
int pnp_bn_update_moving(float* moving_mean, float* moving_var, const float* mean, const float* var,
                         int64_t P, int32_t C, float decay, void* stream);
/* y = act( gamma*(x-mean)*rsqrt(var+eps) + beta + shortcut_padded ), act = leaky-ReLU(alpha) if alpha>=0 else identity.
 * shortcut (optional, may be NULL) has Cs channels, zero-padded (C-Cs)/2 on each side of the channel axis
 * (layers.py:159-165 residual_block: tf.pad(x,[..,[C/2,C/2]]) + add + leaky_relu). */
int pnp_bn_apply(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                 const float* shortcut, int32_t Cs, float* y, int64_t P, int32_t C, float eps, float alpha,
                 void* stream);
/* backward of pnp_bn_apply.  dz = dout * (out>0 ? 1 : alpha).  dbeta = sum dz, dgamma = sum dz*xhat.
 * out == NULL (allowed when no shortcut entered the activation and `beta` is given): the sign of `out` is RECOMPUTED from x as
 * fmaf(x-mean, gamma*rstd, beta) > 0 — bit for bit the value pnp_bn_apply activated — which saves one full read of the activation in
 * the reduction and in the apply pass (these kernels sit on the HBM roof).
 * training=1 : dx = gamma*rstd*(dz - dbeta/P - xhat*dgamma/P);  training=0 (frozen stats): dx = gamma*rstd*dz
 * dx is then multiplied by the dropout mask of the producing conv when keep_prob<1 (layers.py:25: conv->dropout->BN).
 * dshortcut (optional) receives dz restricted to the Cs un-padded channels. */
int pnp_bn_bwd(const float* dout, const float* out /*nullable*/, const float* x, const float* mean, const float* var,
               const float* gamma, const float* beta /*nullable unless out == NULL*/, float* dx, float* dgamma, float* dbeta, float* dshortcut, int32_t Cs,
               int64_t P, int32_t C, float eps, float alpha, int32_t training,
               float keep_prob, uint64_t seed, uint32_t stream_id,
               void* workspace, size_t workspace_bytes, void* stream);
/* Same, and the sums are ALSO added into dgamma_acc / dbeta_acc [C] (both or neither; e.g. the parameters' slots of a flat gradient
 * arena that may already hold another use's contribution) — no separate accumulation kernel per parameter.  dgamma / dbeta still
 * receive this call's own sums (the apply half needs them). */
int pnp_bn_bwd_acc(const float* dout, const float* out /*nullable*/, const float* x, const float* mean, const float* var,
                   const float* gamma, const float* beta /*nullable unless out == NULL*/, float* dx, float* dgamma, float* dbeta, float* dgamma_acc, float* dbeta_acc,
                   float* dshortcut, int32_t Cs, int64_t P, int32_t C, float eps, float alpha, int32_t training,
                   float keep_prob, uint64_t seed, uint32_t stream_id,
                   void* workspace, size_t workspace_bytes, void* stream);
/* The two halves of pnp_bn_bwd, for synchronised batch statistics under data parallelism: reduce the local
 * sums, all-reduce dgamma / dbeta across ranks (caller, RCCL), then apply with P_norm = the GLOBAL row count behind them.
 * pnp_bn_bwd == reduce followed by apply with P_norm = P. */
int pnp_bn_bwd_reduce(const float* dout, const float* out /*nullable*/, const float* x, const float* mean, const float* var,
                      const float* gamma /*nullable unless out == NULL*/, const float* beta /*nullable unless out == NULL*/, float* dgamma, float* dbeta, int64_t P, int32_t C, float eps, float alpha,
                      void* workspace, size_t workspace_bytes, void* stream);
int pnp_bn_bwd_apply(const float* dout, const float* out /*nullable*/, const float* x, const float* mean, const float* var,
                     const float* gamma, const float* beta /*nullable unless out == NULL*/, const float* dgamma, const float* dbeta, float* dx, float* dshortcut, int32_t Cs,
                     int64_t P, int64_t P_norm, int32_t C, float eps, float alpha, int32_t training,
                     float keep_prob, uint64_t seed, uint32_t stream_id, void* stream);

/* relu/leaky-relu of (a + shortcut) with no BN (not on the reference path; kept for the dead helpers) — omitted. */

/* tf.nn.max_pool(ksize 2, stride 2, SAME) (layers.py:102-103); H,W even. */
int pnp_maxpool2_fwd(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int pnp_maxpool2_bwd(const float* x, const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);

/* ops.PS (ops.py:3-27) closed form: out[n, i*r+u, j*r+v, c] = x[n, i, j, c*r*r + v*r + u];  x [N,A,B,nc*r*r] */
int pnp_ps_fwd(const float* x, float* y, int32_t N, int32_t A, int32_t B, int32_t r, int32_t nc, void* stream);
int pnp_ps_bwd(const float* dy, float* dx, int32_t N, int32_t A, int32_t B, int32_t r, int32_t nc, void* stream);

/* tf.pad(x, [[0,0],[p,p],[p,p],[0,0]], 'SYMMETRIC') (layers.py:23,72,91): xp[N,H+2p,W+2p,C], mirror including the edge.
 * The host path pre-pads SYMMETRIC convolutions with this and runs them as VALID convolutions on the tap-unrolled kernel
 * (pnp_conv2d_* also accept PNP_PAD_SYMMETRIC directly and fold the mirror into the gather). */
int pnp_sympad_fwd(const float* x, float* xp, int32_t N, int32_t H, int32_t W, int32_t C, int32_t p, void* stream);
/* backward of tf.pad(x, p, 'SYMMETRIC') in H and W: dx[N,H,W,C] from dxp[N,H+2p,W+2p,C] */
int pnp_sympad_bwd(const float* dxp, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t p, void* stream);

/* Segmentation loss of source_segmenter.py:211-273 (weighted cross-entropy + soft Dice), 5..8 classes.
 * logits, y (one-hot float, lib._label_decomp) : [P, ncls].
 * out[0]=miu_cross*xent + miu_dice*dice, out[1]=xent, out[2]=dice ; dlogits = d out[0] / d logits * gscale.
 * ws: pnp_seg_loss_workspace_bytes(P, ncls).  The per-class sums land in ws and are reused by the bwd call. */
size_t pnp_seg_loss_workspace_bytes(int64_t P, int32_t ncls);
int pnp_seg_loss_fwd(const float* logits, const float* y, float* out, int64_t P, int32_t ncls,
                     float miu_cross, float miu_dice, void* workspace, size_t workspace_bytes, void* stream);
int pnp_seg_loss_bwd(const float* logits, const float* y, float* dlogits, int64_t P, int32_t ncls,
                     float miu_cross, float miu_dice, float gscale,
                     const void* workspace, size_t workspace_bytes, void* stream);
/* Same with the pixel-mean normaliser given explicitly: under data parallelism with batch-global loss normalisers the caller
 * all-reduces the 32 double sums at the head of the workspace (class counts, Dice sums) and passes P_norm = global pixel count. */
int pnp_seg_loss_bwd_norm(const float* logits, const float* y, float* dlogits, int64_t P, int64_t P_norm, int32_t ncls,
                          float miu_cross, float miu_dice, float gscale,
                          const void* workspace, size_t workspace_bytes, void* stream);
/* pixel_wise_softmax_2 + tf.argmax (layers.py:134-138, source_segmenter.py:80-81): exp(z)/sum exp(z), lowest index on ties */
int pnp_softmax_argmax(const float* logits, float* prob /*nullable*/, int64_t* label, int64_t P, int32_t ncls, void* stream);
/* lib._dice_eval (lib.py:96-110): out[0]=mean dice, out[1..ncls]=per class; label = argmax map, y one-hot */
int pnp_dice_eval(const int64_t* label, const float* y, float* out, int64_t P, int32_t ncls,
                  void* workspace, size_t workspace_bytes, void* stream);

/* lib._label_decomp (lib.py:75-92): integer-valued float label map [P] -> one-hot float32 [P, ncls]; labels >= ncls give an all-zero
 * row.  (The reference does this on the host per dequeued batch; here it runs behind the H2D copy of the feeder.) */
int pnp_label_decomp(const float* label, float* onehot, int64_t P, int32_t ncls, void* stream);
/* compact_y = tf.argmax(y, 3) of the one-hot labels (lowest index on ties; source_segmenter.py:83) and
 * tf.confusion_matrix(compact_y, compact_pred, num_classes) (source_segmenter.py:85; rows = ground truth, columns = prediction) in one
 * pass.  compact_y [P] (nullable), pred [P] / cm [ncls*ncls] int64 (both or neither). */
int pnp_confusion_matrix(const float* y, const int64_t* pred, int64_t* compact_y, int64_t* cm, int64_t P, int32_t ncls, void* stream);
/* Synchronised batch statistics (data-parallel replicas of equally many rows): moments[0..C) = mean, moments[C..2C) = var + mean^2 in
 * double; the caller sums `moments` over the replicas (pnp_comm_allreduce, PNP_DTYPE_F64) and converts back with the replica count. */
int pnp_bn_moments(const float* mean, const float* var, double* moments, int32_t C, void* stream);
int pnp_bn_from_moments(const double* moments, int32_t world, float* mean, float* var, int32_t C, void* stream);

/* Optimisers over ONE flat fp32 arena.  The arena is cut in chunks of PNP_OPT_CHUNK elements; chunk_l2[c] is the
 * L2 coefficient (reg_coeff * multiplicity, source_segmenter.py:132-135,237) applied to that chunk: g += l2 * w.
 * chunk_mask[c]==0 skips the chunk (frozen variables).  Either table may be NULL. */
#define PNP_OPT_CHUNK 1024
/* tf.train.AdamOptimizer (source_segmenter.py:378): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w -= lr_t*m/(sqrt(v)+eps) */
int pnp_adam_step(float* w, const float* g, float* m, float* v, size_t n, const float* chunk_l2,
                  const uint8_t* chunk_mask, float lr, float beta1, float beta2, float eps, int32_t t, void* stream);
/* tf.train.RMSPropOptimizer(decay .9, momentum 0, eps 1e-10) (adversarial.py:643-652): ms=.9ms+.1g^2; w-=lr*g/sqrt(ms+eps) */
int pnp_rmsprop_step(float* w, const float* g, float* ms, size_t n, const float* chunk_l2, const uint8_t* chunk_mask,
                     float lr, float decay, float eps, void* stream);
/* tf.train.MomentumOptimizer (source_segmenter.py:370): acc = mom*acc + g ; w -= lr*acc */
int pnp_momentum_step(float* w, const float* g, float* acc, size_t n, const float* chunk_l2, const uint8_t* chunk_mask,
                      float lr, float momentum, void* stream);
/* tf.clip_by_value weight clipping (adversarial.py:654), chunk_mask selects the clipped chunks */
int pnp_clip(float* w, size_t n, const uint8_t* chunk_mask, float lo, float hi, void* stream);
/* sum over chunks of chunk_l2[c] * sum(w^2)/2  (tf.nn.l2_loss, source_segmenter.py:237) -> out[0] */
int pnp_l2_loss(const float* w, size_t n, const float* chunk_l2, float* out, void* workspace, size_t workspace_bytes, void* stream);
size_t pnp_reduce_workspace_bytes(size_t n);

/* Critic input assembly (adversarial.py:325-335): concat on C of [tile(a,3) | b | c | d | logits | float(argmax logits)] */
int pnp_critic_input_fwd(const float* a, int32_t Ca, int32_t tile_a, const float* b, int32_t Cb, const float* c, int32_t Cc,
                         const float* d, int32_t Cd, const float* logits, int32_t ncls, float* out, int64_t P, void* stream);
int pnp_critic_input_bwd(const float* dout, float* da, int32_t Ca, int32_t tile_a, float* db, int32_t Cb, float* dc, int32_t Cc,
                         float* dd, int32_t Cd, float* dlogits, int32_t ncls, int64_t P, void* stream);

/* WGAN critic losses (adversarial.py:455-459): out[0] = sum_i coef[i] * mean(logit_i[0..B)) over the (up to 4) non-NULL
 * critic-logit vectors  {ct_cls, mr_cls, ct_mask, mr_mask};  e.g. dis_loss: coef = {+miu, -miu, +lambda*miu, -lambda*miu}. */
int pnp_wgan_loss(const float* ct_cls, const float* mr_cls, const float* ct_mask, const float* mr_mask, int32_t B,
                  float c_ct_cls, float c_mr_cls, float c_ct_mask, float c_mr_mask, float* out, void* stream);
/* p[0..n) = value  (constant gradient of a mean: coef / B) */
int pnp_fill(float* p, size_t n, float value, void* stream);

/* ---- bf16-RESIDENT convolutions (BASELINE.json configs[4]: bf16 mixed precision; csrc/conv_bf16r.hip) ------------------------------
 * The same convolutions as pnp_conv2d_fwd / _dgrad (layers.py:18,24,67,73,86,92) with BOTH MFMA operands stored as bfloat16 in HBM:
 *   xh / dyh : bf16 copy [N,H,W,C] of an activation / upstream gradient, written by the kernel that produced the tensor (the `yh` /
 *              `dxh` outputs below, pnp_bn_apply_h, pnp_bn_bwd_apply_h) or by pnp_cast_bf16,
 *   w_oi     : bf16 shadow [R*S][K][C] of the fp32 master filter [R,S,C,K]  (forward: reduction index C contiguous),
 *   w_io     : bf16 shadow [R*S][C][K]                                      (data gradient: reduction index K contiguous; the kernel
 *              walks the taps in reverse — no flip / transpose launch),
 * both from pnp_filter_bf16 (either pointer nullable).  fp32 accumulation; y / dx are float32 as everywhere else, `yh` / `dxh`
 * (nullable) receive the same values rounded to bf16 (nearest-even) from the same epilogue.  Arithmetic: exactly "both operands of
 * every product rounded to bf16, products summed in fp32" — what PNP_DTYPE_BF16 means for the staged-rounding kernels too.
 * pnp_conv2d_bf16r_served(g, kind) (kind 0 forward, 1 data gradient): 1 when these kernels serve the geometry (zero padding, 3x3 /
 * forward 5x5, reduction channels % 32 == 0, output channels % 64 == 0, >= 4096 output pixels; data gradients of stride 1 directly, of
 * strided layers one launch per stride phase with the phase's sub-filter read out of the full shadow); everything else stays on the
 * fp32-storage entry points above. */
int pnp_cast_bf16(const float* x, void* y_bf16, size_t n, void* stream);
int pnp_filter_bf16(const float* w, void* w_io /*nullable*/, void* w_oi /*nullable*/, int32_t R, int32_t S, int32_t C, int32_t K,
                    void* stream);
int32_t pnp_conv2d_bf16r_served(const pnp_conv_geom* g, int32_t kind);
/* number of BN-statistics partial rows the resident forward leaves (its own tiles: not pnp_conv2d_fwd_stats_parts) */
int32_t pnp_conv2d_fwd_bf16r_stats_parts(const pnp_conv_geom* g);
/* forward + dropout, optionally (stat_parts != NULL) the BN statistics partials of pnp_conv2d_fwd_stats, optionally (scale != NULL) the
 * fused inference-mode BN + shortcut + leaky-ReLU of pnp_conv2d_fwd_bn */
int pnp_conv2d_fwd_bf16r(const void* xh, const void* w_oi, float* y, void* yh /*nullable*/, const pnp_conv_geom* g,
                         float keep_prob, uint64_t seed, uint32_t stream_id,
                         const float* stat_shift /*nullable*/, float* stat_parts /*nullable*/, size_t stat_parts_bytes,
                         const float* scale /*nullable*/, const float* shift, const float* shortcut /*nullable*/, int32_t Cs, float alpha,
                         void* stream);
/* dx = data gradient (+ residual, nullable, as pnp_conv2d_dgrad_add) */
int pnp_conv2d_dgrad_bf16r(const void* dyh, const void* w_io, const float* residual /*nullable*/, float* dx, void* dxh /*nullable*/,
                           const pnp_conv_geom* g, void* stream);

/* The elementwise producers with a bf16 SIDE OUTPUT (`yh` / `dxh`, nullable): the same values as the float32 output rounded to bf16, from
 * the same launch — what the next resident convolution reads.  Otherwise identical to pnp_bn_apply / pnp_bn_bwd_acc / pnp_bn_bwd_apply /
 * pnp_dropout (pnp_dropout_h: y may be NULL when only the bf16 copy is wanted). */
int pnp_bn_apply_h(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                   const float* shortcut /*nullable*/, int32_t Cs, float* y, void* yh, int64_t P, int32_t C, float eps, float alpha,
                   void* stream);
int pnp_bn_bwd_acc_h(const float* dout, const float* out, const float* x, const float* mean, const float* var, const float* gamma,
                     const float* beta, float* dx, void* dxh, float* dgamma, float* dbeta, float* dgamma_acc, float* dbeta_acc,
                     float* dshortcut, int32_t Cs, int64_t P, int32_t C, float eps, float alpha, int32_t training, float keep_prob,
                     uint64_t seed, uint32_t stream_id, void* workspace, size_t workspace_bytes, void* stream);
int pnp_bn_bwd_apply_h(const float* dout, const float* out, const float* x, const float* mean, const float* var, const float* gamma,
                       const float* beta, const float* dgamma, const float* dbeta, float* dx, void* dxh, float* dshortcut, int32_t Cs,
                       int64_t P, int64_t P_norm, int32_t C, float eps, float alpha, int32_t training, float keep_prob, uint64_t seed,
                       uint32_t stream_id, void* stream);
int pnp_dropout_h(const float* x, float* y /*nullable*/, void* yh, size_t n, float keep_prob, uint64_t seed, uint32_t stream_id,
                  void* stream);

/* filter gradient from bf16 x [N,H,W,C] and bf16 dy [N,OH,OW,K] (pnp_conv2d_bf16r_served(g, 2)): dw [R,S,C,K] float32 is overwritten
 * (accumulate = 0) or added to (accumulate != 0, as pnp_conv2d_wgrad_acc); workspace: pnp_conv2d_wgrad_bf16r_workspace_bytes(g) */
size_t pnp_conv2d_wgrad_bf16r_workspace_bytes(const pnp_conv_geom* g);
int pnp_conv2d_wgrad_bf16r(const void* xh, const void* dyh, float* dw, int32_t accumulate, const pnp_conv_geom* g,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---- step capture (one hipGraph launch per training step instead of ~1 400 kernel launches from the host) ---------------------------------
 * A captured step replays its launches with every by-value argument frozen.  Two scalars change every step: the dropout seed
 * (tf.nn.dropout's per-run mask, layers.py:25,74,93) and Adam's bias-corrected learning rate (tf.train.AdamOptimizer, source_segmenter.py:378).
 * With a 16-byte DEVICE block bound (pnp_step_params_bind(block); NULL unbinds: the default), every dropout-carrying entry point takes
 * its seed from the block instead of its `seed` argument — the mask stream is the same function of (seed, stream_id) either way — and
 * pnp_adam_step takes lr * sqrt(1 - beta2^t) / (1 - beta1^t) from the block instead of computing it from (lr, t).
 * pnp_step_params_set enqueues the write of the block on `stream` (ordered before the graph launch that follows it).
 * The binding is process-global state of the library (one training loop per process). */
int pnp_step_params_bind(const void* dev_block /*16 bytes, or NULL*/);
int pnp_step_params_set(void* dev_block, uint64_t drop_seed, float adam_lr_t, void* stream);

/* Data-parallel exchange step (new with respect to the single-GPU reference, train_segmenter.py:20 / train_gan.py:18): in-place
 * SUM all-reduce over RCCL (xGMI), one communicator per process / GPU.  librccl.so is resolved at run time: pnp_comm_load(path)
 * names the copy to bind (NULL: the one already mapped in this process, else the default search path); the other calls load it
 * on first use.  Bring-up: rank 0 calls pnp_comm_unique_id and ships the PNP_COMM_ID_BYTES to every rank out of band (the host
 * side uses the torchrun rendezvous store); then EVERY rank calls pnp_comm_init (collective; binds the current HIP device).
 * pnp_comm_allreduce enqueues on `stream` and returns; ordering against producers / consumers of `buf` is the caller's business
 * (events on that stream).  buf: n elements of dtype PNP_DTYPE_F32 or PNP_DTYPE_F64. */
#define PNP_COMM_ID_BYTES 128
int pnp_comm_load(const char* librccl_path /*nullable*/);
int pnp_comm_version(int* version);
int pnp_comm_unique_id(uint8_t* id /*[PNP_COMM_ID_BYTES]*/);
int pnp_comm_init(int32_t rank, int32_t world, const uint8_t* id, void** comm_out);
int pnp_comm_allreduce(void* comm, void* buf, size_t n, int32_t dtype, void* stream);
int pnp_comm_destroy(void* comm);

/* y = a*x + b*y elementwise (gradient fan-in adds) */
int pnp_axpby(const float* x, float* y, size_t n, float a, float b, void* stream);
/* out = x + y (out may alias neither): the gradient sum of a tensor that feeds two branches of the graph (TF autodiff's AddN) */
int pnp_add(const float* x, const float* y, float* out, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PNP_HIP_H */
