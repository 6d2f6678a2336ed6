"""not gpu: tolerance budget for BASELINE config 5 (bf16 MFMA operands, fp32 accumulation: csrc/conv_bf16.hip / conv_bf16r.hip): the CPU oracle
with both operands of every convolution rounded to bfloat16, against the fp32 oracle, on a B=1 slice with He-scaled random filters
in inference mode.  The numbers bound what the bf16 conv path is held to by tests/test_gpu_bf16.py ("tolerance-checked Dice vs fp32")."""
import numpy as np
import torch

from oracle import nets
from oracle import tf_ops as T


def test_bf16_operand_rounding_budget():
    rng = np.random.default_rng(0)
    state = {}
    for k, s in nets.segmenter_variable_shapes().items():
        if "Variable" in k:
            state[k] = (rng.standard_normal(s) * np.sqrt(2.0 / (s[0] * s[1] * s[2])) * 0.9).astype(np.float32)
        elif k.endswith("moving_mean"):
            state[k] = (0.05 * rng.standard_normal(s)).astype(np.float32)
        elif k.endswith(("gamma", "moving_variance")):
            state[k] = (1.0 + 0.1 * rng.random(s)).astype(np.float32)
        else:
            state[k] = np.zeros(s, np.float32)
    V = nets.make_variables(state)
    x = torch.from_numpy(rng.standard_normal((1, 256, 256, 3)).astype(np.float32))
    with torch.no_grad():
        ref = nets.segmenter_forward(V, x, 1.0, main_bn=False, adapt_bn=False)
        low = nets.segmenter_forward(V, x, 1.0, main_bn=False, adapt_bn=False, operand_round=T.round_bf16)
    rel = float((low - ref).abs().max() / ref.abs().max())
    agree = float((low.argmax(3) == ref.argmax(3)).float().mean())
    # hard Dice of the bf16 label map against the fp32 label map, per class present
    a, b = low.argmax(3), ref.argmax(3)
    dices = []
    for c in range(5):
        pa, pb = (a == c), (b == c)
        if int(pb.sum()) > 0:
            dices.append(2.0 * float((pa & pb).sum()) / float(pa.sum() + pb.sum()))
    print("bf16 operands vs fp32, 33 conv layers, inference: logits max error %.3e of max|logit|, argmax agreement %.5f, "
          "label-map Dice vs fp32 (classes present): %s" % (rel, agree, ["%.4f" % d for d in dices]))
    assert T.round_bf16(torch.tensor([1.0 + 2.0 ** -9])).item() == 1.0 and T.round_bf16(torch.tensor([1.0 + 3 * 2.0 ** -9])).item() == 1.0 + 2.0 ** -7
    assert rel < 0.1 and agree > 0.97 and min(dices) > 0.95
