"""CPU: host logic behind the gradient sinks (gradsink.py), the reducer's once-per-variable hooks (parallel.GradReducer) and the
arithmetic identity the in-place high/low compaction of long partial lists relies on (csrc/elementwise.hip colreduce_compact_kernel).
The kernels themselves are held to the autograd engine / to float64 sums by the -m gpu tests; this file pins the bookkeeping."""
import gc

import numpy as np
import torch

from conftest import pkg


def _param(n=8):
    p = torch.zeros(n, requires_grad=True)
    p.grad = torch.zeros(n)
    return p


def test_sink_counts_uses_and_announces_the_last_one():
    gs = pkg("gradsink")
    p = _param()
    s = gs.register(p)
    fired = []
    assert gs.set_ready(p, lambda: fired.append("ready"))
    a, b = gs.use(p), gs.use(p)                 # two call sites consume the variable in this step (shared critic filters)
    assert a is s and b is s and s.pending == 2
    assert s.grad() is p.grad                   # the slot the kernels add into is the variable's own .grad view
    gs.done(s)
    assert fired == [] and s.pending == 1       # the first use's gradient is in: not final yet
    gs.done(s)
    assert fired == ["ready"] and s.pending == 0
    # a forward whose backward never ran must not leak into the next step: zero_grad re-arms
    gs.use(p)
    gs.rearm([p])
    assert s.pending == 0
    gs.use(p)
    gs.done(s)
    assert fired == ["ready", "ready"]


def test_sink_refuses_the_forward_zero_grad_backward_order_and_untaped_uses():
    """(a) a use recorded by a forward under no_grad (monitoring / evaluation: no backward will follow) must not count —
    `taped` is evaluated by the caller of Function.apply; (b) zero_grad BETWEEN forward and backward wipes the recorded uses: the
    backward's done() must raise instead of clamping at zero and announcing a shared variable after its first use."""
    import pytest
    gs = pkg("gradsink")
    p = _param()
    s = gs.register(p)
    fired = []
    gs.set_ready(p, lambda: fired.append(1))
    assert gs.use(p, taped=False) is None and s.pending == 0
    a, b = gs.use(p, True), gs.use(p, True)          # forward: two uses
    gs.rearm([p])                                     # zero_grad in the wrong place
    with pytest.raises(RuntimeError, match="zero_grad"):
        gs.done(a)
    assert fired == []                                # nothing was announced early
    with gs.disabled():
        assert gs.use(p) is None and gs.lookup(p) is None
    assert gs.lookup(p) is s


def test_sink_lookup_is_by_storage_and_dies_with_its_variable():
    gs = pkg("gradsink")
    p = _param(16)
    gs.register(p)
    alias = p.detach()                          # what Function.forward / saved_tensors hand back: another object, same storage
    assert gs.lookup(alias) is not None and gs.lookup(alias).param() is p
    assert gs.lookup(p[4:]) is None             # another address
    assert gs.lookup(torch.zeros(16)) is None   # unrelated tensor
    other = torch.zeros(4)
    assert not gs.set_ready(other, lambda: None)
    ptr = p.data_ptr()
    keep_storage = p.detach()                   # keep the memory alive so that the address cannot be handed to a new tensor
    del p, alias
    gc.collect()
    assert gs.lookup(keep_storage) is None      # the owning variable is gone: a stale entry never matches (and is dropped)
    assert ptr not in gs._SINKS


def test_sinks_can_be_switched_off():
    gs = pkg("gradsink")
    p = _param()
    gs.register(p)
    try:
        gs.ENABLED = False
        assert gs.lookup(p) is None and gs.use(p) is None
    finally:
        gs.ENABLED = True
    assert gs.use(p) is not None
    gs.rearm([p])


def test_reducer_hook_counts_each_variable_once_per_step():
    """A gradient that went through a sink announces itself AND the engine's AccumulateGrad still runs the post-accumulate hook for the
    undefined gradient it was handed (measured on the GPU box: every variable fired twice, buckets were reduced half-filled).  The hook
    must count a variable once, in whichever order the two arrive, and re-arm on reset()."""
    par, var = pkg("parallel"), pkg("variables")
    store = var.VariableStore("cpu", seed=0)
    with store.as_default():
        for i in range(6):
            store.get("v%d" % i, (300,), 0.0, True, "weight")
    store.finalize()
    red = par.GradReducer(store, bucket_bytes=3 * 1024 * 4, overlap=True)       # CPU, no process group: overlap is off, hooks by hand
    assert not red.overlap and len(red.buckets) == 2
    launched = []
    red._launch = lambda b: launched.append(b)
    hooks = {v.name: red._make_hook(v) for v in store.trainable()}
    order = [v.name for v in reversed(store.trainable())]                       # gradients arrive last-created first
    for name in order[:3]:
        hooks[name](None)           # sink
        hooks[name](None)           # engine, same step
    assert launched == [0] and red._count[0] == 0 and red._count[1] == 3
    for name in order[3:]:
        hooks[name](None)
    assert launched == [0, 1]
    red.reset()
    assert red._count == red._remaining and not red._seen
    hooks[order[0]](None)
    assert red._count[0] == 2


def test_high_low_float_pairs_carry_a_double_sum():
    """colreduce_compact_kernel writes a slab's double sum s back as (h, l) = (float(s), float(s - h)); the final kernel adds h and l in
    double.  That reproduces s to ~2^-47 relative (two 24-bit mantissas), overflow stays infinite instead of turning into NaN."""
    rng = np.random.default_rng(0)
    s = rng.standard_normal(100000) * np.exp(rng.uniform(-20, 60, 100000))
    h = s.astype(np.float32)
    lo = (s - h.astype(np.float64)).astype(np.float32)
    back = h.astype(np.float64) + lo.astype(np.float64)
    assert np.max(np.abs(back - s) / np.abs(s)) < 2.0 ** -46
    big = np.float64(1e40)
    with np.errstate(over="ignore", invalid="ignore"):
        hb = np.float32(big)
        lb = np.float32(0.0) if not np.isfinite(hb) else np.float32(big - np.float64(hb))      # the kernel's guard
    assert np.isinf(hb) and np.float64(hb) + np.float64(lb) == np.inf


def test_bn_statistics_versions_are_forgotten_when_a_new_store_takes_the_addresses():
    """functional._STAT_VERSION is keyed by the address of a moving-statistics buffer; a store that dies leaves its counters behind and
    the next store's arena may land on the same addresses (round-3 review): finalize() drops every counter inside the new arena."""
    F = pkg("functional")
    F._STAT_VERSION.clear()
    F._STAT_VERSION.update({1000: 3, 1016: 1, 5000: 7})
    F.forget_stat_versions(1000, 1020)
    assert F._STAT_VERSION == {5000: 7}
    variables = pkg("variables")
    st = variables.VariableStore("cpu", seed=0)
    mm = st.get("bn/moving_mean", (8,), 0.0, False, "bn_stat")
    st.get("w", (3, 3, 4, 8), 0.5, True, "weight")
    st.finalize()
    base = st.state_arena.data_ptr()
    assert st.vars["bn/moving_mean"].tensor.data_ptr() == base
    F._STAT_VERSION[base] = 9                       # as if a dead store had bumped a buffer at this address
    st2 = variables.VariableStore("cpu", seed=0)
    st2.get("bn/moving_mean", (8,), 0.0, False, "bn_stat")
    st2.finalize()
    assert st2.state_arena.data_ptr() not in F._STAT_VERSION
    F._STAT_VERSION.clear()
