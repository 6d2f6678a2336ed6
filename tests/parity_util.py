"""Shared parity checks of the -m gpu tests."""
import torch


def argmax_mismatches(logits, logits64, noise_rel=1e-5):
    """Bit-exact label maps, adjudicated in float64: returns (n_mismatch, n_unexplained, worst_margin, noise).

    north_star asks for a bit-exact argmax label map.  Two float32 evaluations of the same graph can legitimately differ where the
    float64 top-2 margin of a pixel is below float32 round-off of the logits; a mismatch is "explained" only there
    (margin <= noise_rel * max|logit|), every other mismatch is a failure."""
    lg = torch.as_tensor(logits).detach().cpu()
    l64 = torch.as_tensor(logits64).detach().cpu().double()
    lab, lab64 = lg.argmax(-1), l64.argmax(-1)
    top2 = torch.topk(l64, 2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    mism = lab != lab64
    noise = noise_rel * float(l64.abs().max())
    worst = float(margin[mism].max()) if bool(mism.any()) else 0.0
    return int(mism.sum()), int((mism & (margin > noise)).sum()), worst, noise


def assert_argmax_exact(logits, logits64, what="argmax", noise_rel=1e-5):
    n, bad, worst, noise = argmax_mismatches(logits, logits64, noise_rel)
    print("%s: %d mismatching pixels of %d (largest fp64 margin among them %.3e, fp32 noise floor %.3e), unexplained %d" % (
        what, n, torch.as_tensor(logits).shape.numel() // torch.as_tensor(logits).shape[-1], worst, noise, bad))
    assert bad == 0, "%s: %d label-map mismatches at pixels whose float64 margin exceeds float32 noise" % (what, bad)


def rel(a, b):
    """max|a - b| / max|b| in float64.  When either side lives on the GPU the comparison runs THERE (round 6: widening 67 M-element maps to
    float64 on the host was 60 of the 231 s of the B = 16 adaptation walk and 22 of the 50 s of the segmenter's).  Same value as on the host, bit for bit:
    the widening and the subtraction are exact / correctly rounded element-wise operations and a maximum does not depend on its order."""
    a, b = torch.as_tensor(a).detach(), torch.as_tensor(b).detach()
    dev = a.device if a.is_cuda else (b.device if b.is_cuda else None)
    if dev is not None:
        a, b = a.to(dev), b.to(dev)
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def batch_moments64(y, dev):
    """per-channel mean and biased variance of a [.., C] map in float64, reduced on `dev`; returned on the host"""
    y64 = y.detach().to(dev).double().reshape(-1, y.shape[-1])
    m = y64.mean(0)
    v = ((y64 - m) ** 2).mean(0)
    return m.cpu(), v.cpu()
