"""Drop-in for the reference's ops.py: the sub-pixel phase-shift upsampling `PS` (ops.py:3-27).

The reference builds PS out of reshape/transpose/split/squeeze/concat per channel group; its net effect for
batch_size >= 2 is the closed form
    out[n, i*r+u, j*r+v, c] = X[n, i, j, c*r*r + v*r + u]
(tests/golden pins this against a numpy emulation of the reference op sequence).  Here it is one gather kernel
(pnp_ps_fwd) and its inverse scatter for the gradient.
"""
import torch

from .functional import PSFn


def PS(X, r, n_channel=8, batch_size=10):
    """ops.py:23.  `batch_size` is accepted for signature compatibility; the closed form holds for any batch
    (the reference's batch_size==1 branch yields a transposed image and is never used with B>=2)."""
    if X.shape[-1] != n_channel * r * r:
        raise ValueError("PS: input has %d channels, expected n_channel*r*r = %d" % (X.shape[-1], n_channel * r * r))
    if X.is_meta:   # symbolic build pass
        return torch.empty((X.shape[0], X.shape[1] * r, X.shape[2] * r, n_channel), device="meta")
    return PSFn.apply(X, r, n_channel)


def _phase_shift(I, r, batch_size=10):
    """ops.py:3-21, the per-channel-group helper of PS: [B,a,b,r*r] -> [B,a*r,b*r,1] (closed form with one output channel)"""
    return PS(I, r, n_channel=1, batch_size=batch_size)
