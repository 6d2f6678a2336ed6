"""medical-cross-modality-domain-adaptation_amd — MI355X-native hot path of PnP-AdaNet
(carrenD/Medical-Cross-Modality-Domain-Adaptation): dilated-residual segmenter fwd/bwd + Wasserstein critics.

Host side = Python mirroring the reference's module surface (layers / ops / lib / source_segmenter / adversarial /
train_segmenter / train_gan); arithmetic = hand-written gfx950 HIP kernels in libpnp_hip.so behind the C-ABI of
include/pnp_hip.h, bound with ctypes (_lib.py).  The directory name contains '-' (it is fixed by the build
contract), so import it with importlib or through the `pnp_amd` alias module at the repository root.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "kernels", "functional", "variables", "layers", "ops", "lib", "source_segmenter"]
