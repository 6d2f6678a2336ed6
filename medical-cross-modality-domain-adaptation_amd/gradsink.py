"""Gradient sinks: parameter gradients written by the kernels straight into the flat gradient arena.

TF-1.4's autodiff materialises one gradient tensor per use of a variable and sums them with AddN; torch's engine does the same with an
`AccumulateGrad` node per parameter, i.e. one elementwise add kernel per variable per step into `.grad` (376 variables in the
adaptation graph: ~250 launches of 4-6 us per joint step).  A parameter that lives in a `VariableStore` arena registers its gradient
slot here instead; the backward passes in functional.py hand that slot to the kernel that produces the gradient
(pnp_conv2d_wgrad_acc, pnp_bn_bwd_acc: "add into"), return None to the engine, and tell the sink when the last use of the step has
been written (the data-parallel reducer launches a bucket's all-reduce from that callback).

The arena is zeroed once per step (`VariableStore.zero_grad`), which also re-arms the use counters.

Contract (the step functions of source_segmenter.py / adversarial.py follow it):
  * `zero_grad()` comes BEFORE the forward pass of the step.  The usual torch order forward -> zero_grad -> backward would wipe the
    uses the forward recorded; `done()` then RAISES (a count below zero) instead of announcing a bucket early.
  * a use is recorded only when the CALLER of `Function.apply` is recording a tape (`taped = torch.is_grad_enabled()` evaluated at the
    call site — inside `Function.forward` grad mode is always off and `ctx.needs_input_grad` is true under `no_grad` too), so
    monitoring / evaluation forwards leave the counters alone.
  * anything that asks the ENGINE for parameter gradients — `torch.autograd.grad(loss, params)`, tensor hooks on a weight,
    `retain_graph` + a second backward — gets None for every store variable and adds into the arena as a side effect.  Wrap such code
    in `with gradsink.disabled():` (the kernels then return the gradients to the engine, which accumulates into `.grad` itself).
"""
import contextlib
import os
import weakref

ENABLED = os.environ.get("PNP_GRAD_SINKS", "1") != "0"


@contextlib.contextmanager
def disabled():
    """parameter gradients through the autograd engine (AccumulateGrad) for the forward passes recorded inside this block"""
    global ENABLED
    prev = ENABLED
    ENABLED = False
    try:
        yield
    finally:
        ENABLED = prev


class Sink(object):
    __slots__ = ("param", "pending", "ready")

    def __init__(self, param):
        self.param = weakref.ref(param)     # the store's own tensor object: while it lives nothing else can sit at its address
        self.pending = 0                    # uses recorded by forward passes whose backward has not run yet
        self.ready = None                   # callable(): every use of this step has been written

    def grad(self):
        p = self.param()
        return None if p is None else p.grad


_SINKS = {}     # data_ptr of the parameter -> Sink
_READY = [0]    # sinks with a `ready` hook set (kept by set_ready / register / lookup): has_ready_hooks() is asked per filter gradient


def has_ready_hooks():
    """does any LIVE sink announce completed gradients to a data-parallel reducer (then gradients must complete on the compute stream).
    O(1) without data parallelism; with hooks set, sinks whose store is gone are pruned first so that a dead data-parallel store cannot
    switch the side-stream overlap off for the rest of the process"""
    if _READY[0] <= 0:
        return False
    for k in [k for k, s in _SINKS.items() if s.ready is not None and s.param() is None]:
        del _SINKS[k]
        _READY[0] -= 1
    return _READY[0] > 0


def register(param):
    """param: a leaf tensor whose .grad is (a view of) the buffer the kernels may add into"""
    old = _SINKS.get(param.data_ptr())
    if old is not None and old.ready is not None:
        _READY[0] -= 1
    s = Sink(param)
    _SINKS[param.data_ptr()] = s
    return s


def lookup(t):
    if not ENABLED:
        return None
    s = _SINKS.get(t.data_ptr())
    if s is None:
        return None
    p = s.param()
    # (same address and element count = the same variable, possibly viewed with another shape: the critics' [D, 1] matmul weights run as
    # 1x1 convolutions [1, 1, D, 1])
    if p is None or p.grad is None or p.data_ptr() != t.data_ptr() or p.numel() != t.numel():
        if p is None:
            if _SINKS.pop(t.data_ptr(), None) is not None and s.ready is not None:          # the store that owned this address is gone
                _READY[0] -= 1
        return None
    return s


def use(t, taped=True):
    """forward pass: `t` will receive a gradient from this call site -> its sink (or None), with the use recorded.
    taped: the caller of Function.apply is recording a tape (a backward pass can follow); False -> no sink, nothing recorded"""
    if not taped:
        return None
    s = lookup(t)
    if s is not None:
        s.pending += 1
    return s


def done(s):
    """backward pass: one use has been added into the slot"""
    s.pending -= 1
    if s.pending < 0:
        s.pending = 0
        raise RuntimeError("gradient sink: a backward pass wrote a parameter gradient whose forward use is not on record — "
                           "zero_grad() (which re-arms the use counters) must run BEFORE the forward pass of the step, not between "
                           "forward and backward")
    if s.pending == 0 and s.ready is not None:
        s.ready()


def set_ready(param, fn):
    s = _SINKS.get(param.data_ptr())
    if s is not None and s.param() is param:
        _READY[0] += (fn is not None) - (s.ready is not None)
        s.ready = fn
        return True
    return False


def rearm(params):
    for p in params:
        s = _SINKS.get(p.data_ptr())
        if s is not None:
            s.pending = 0
