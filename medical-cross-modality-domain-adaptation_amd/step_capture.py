"""Step capture: one hipGraph launch per training step.

The reference hands TensorFlow ONE `sess.run(optimizer, feed_dict)` per step (adversarial.py:852-881, source_segmenter.py:484-489); the
eager host side here issues ~1 400 kernel launches per joint step through ctypes.  At B = 16 the GPU hides that (1.7 % of the step is
gaps between kernels), at the strong-scaling operating points (B = 2 per GPU, 1/8 of the kernel time, the same launch count) the host
becomes the bound.  `CapturedStep` records the launches of one step function — forward, backward, optimiser, clip — into a hipGraph
(torch.cuda.CUDAGraph: torch supplies the capture plumbing and the graph-private memory pool, every captured launch is libpnp_hip.so's)
and replays it.

What a replay cannot change is every by-value kernel argument.  Two scalars change per step:
  * the dropout seed — the kernels of a captured step read it from a 16-byte device block (pnp_step_params_bind; the mask stream stays
    the same function of (seed, call-site id), so a captured step with seed s draws exactly the masks the eager step with seed s draws),
  * Adam's bias-corrected learning rate — same block (`adam_lr_t`); RMSProp has no per-step scalar.
Input batches are copied into the captured step's static input tensors (one device copy per input), the loss comes back in a static
output tensor.  Host-side bookkeeping that the eager step does in Python (optimiser step counters) is done by `replay`.

Not captured: steps under data parallelism (the bucketed all-reduce lives on a side stream behind events), learning-rate schedules that
change `lr` by value (re-capture after the change), the profiler (pnp_prof_* records events around launches: probe eagerly).
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from . import kernels as K
import weakref


def _unpin():
    K.PINNED[0] -= 1
    if K.PINNED[0] <= 0:
        del K._ws_retired[:]         # no recording left that could replay their addresses


class CapturedStep(object):
    def __init__(self, fn, inputs, adam=None, warmup=2):
        """fn(*inputs) -> device tensor (the loss); `inputs`: example DEVICE tensors of the shapes every later call will have;
        adam: an AdamOptimizer whose step counter / learning rate feed `adam_lr_t` (None: the step has no Adam).
        The `warmup` calls before the recording are REAL steps on `inputs` (they train); the recording itself executes nothing."""
        lib = _lib.load()
        dev = inputs[0].device
        self.fn, self.adam = fn, adam
        self.static = [t.clone() for t in inputs]
        self.block = torch.zeros(4, dtype=torch.float32, device=dev)          # pnp_step_params: {u64 drop_seed, f32 adam_lr_t, f32 -}
        self._bp = ctypes.c_void_p(self.block.data_ptr())
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):                    # warm-up on a side stream (allocator state, lazy initialisations, caches)
            for i in range(warmup):
                self._pre(0x5EED + i, side)
                fn(*self.static)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        _lib.check(lib.pnp_step_params_bind(self._bp), "pnp_step_params_bind")
        t0 = adam.t if adam is not None else 0
        try:
            with torch.cuda.graph(self.graph):
                self.out = fn(*self.static)
        finally:
            _lib.check(lib.pnp_step_params_bind(None), "pnp_step_params_bind")
            if adam is not None:
                adam.t = t0                              # recording executes nothing: the counter moved, the weights did not
        self.replays = 0
        K.PINNED[0] += 1                                  # the graph holds addresses of cached filter shadows: nothing may be freed under it
        weakref.finalize(self, _unpin)

    def _lr_t(self):
        a = self.adam
        if a is None or a.t < 1:
            return 0.0
        # exactly pnp_adam_step's arithmetic: its lr / beta arguments are C floats, the formula runs in double, the result is a float
        lr, b1, b2 = (float(np.float32(v)) for v in (a.lr, a.b1, a.b2))
        return float(np.float32(lr * math.sqrt(1.0 - b2 ** a.t) / (1.0 - b1 ** a.t)))

    def _pre(self, seed, stream=None):
        st = ctypes.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        _lib.check(_lib.load().pnp_step_params_set(self._bp, int(seed) & 0xFFFFFFFFFFFFFFFF, float(self._lr_t()), st), "pnp_step_params_set")

    def replay(self, seed, *inputs):
        """one step with dropout seed `seed` on `inputs` (same shapes as at capture; tensors that ARE the static ones are not copied)"""
        for s, t in zip(self.static, inputs):
            if t is not s:
                s.copy_(t, non_blocking=True)
        if self.adam is not None:
            self.adam.t += 1
        self._pre(seed)
        # a replay rewrites the weights (optimiser / clip nodes) behind torch's back: ctypes writes do not bump version counters, so the
        # transformed-filter cache and the bf16 shadows are told here, not by every caller
        K.weights_changed()
        self.graph.replay()
        self.replays += 1
        return self.out.clone()          # (the static output tensor is overwritten by the next replay: callers may keep what they get)
