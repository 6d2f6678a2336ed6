"""-m gpu: step capture (step_capture.py) — a training step recorded into ONE hipGraph replays bit for bit what the eager step does:
same kernels, same order, the dropout seed and Adam's bias-corrected learning rate read from the step's device block."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu
COST = {"cross_flag": True, "miu_cross": 1.0, "dice_flag": True, "miu_dice": 1.0, "regularizer": 1e-4}


def _blob(rng, B):
    lab = np.zeros((B, 256, 256), np.int64)
    yy, xx = np.mgrid[0:256, 0:256]
    for b in range(B):
        for c in range(1, 5):
            cy, cx = rng.integers(40, 216, 2)
            lab[b][((yy - cy) / 30.0) ** 2 + ((xx - cx) / 22.0) ** 2 <= 1] = c
    out = np.zeros(lab.shape + (5,), np.float32)
    for i in range(5):
        out[..., i][lab == i] = 1
    return out


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_captured_segmenter_step_equals_eager_bit_for_bit(dev, dtype):
    """bf16: the recording must hold its own operand casts — kernels.bf16_of skips the per-tensor shadow while a stream is
    capturing, so a replay reads the bf16 copy of THIS step's static input, not the warm-up's."""
    F = pkg("functional")
    F.set_conv_dtype(dtype)
    try:
        _segmenter_capture_case(dev)
    finally:
        F.set_conv_dtype("f32")


def _segmenter_capture_case(dev):
    ss = pkg("source_segmenter")
    from dp_sync_common import scaled_state
    B = 2
    rng = np.random.default_rng(3)
    xs = [torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dev) for _ in range(3)]
    ys = [torch.from_numpy(_blob(rng, B)).to(dev) for _ in range(3)]

    def make():
        net = ss.Full_DRN(channels=3, n_class=5, batch_size=B, device=dev, cost_kwargs=dict(COST), seed=0)
        tr = ss.Trainer(net, None, None, num_cls=5, batch_size=B, optimizer="adam", opt_kwargs={"learning_rate": 1e-3})
        tr.opt = tr._get_optimizer(10)
        return net, tr
    net_e, tr_e = make()
    state0 = scaled_state(net_e)
    net_e.store.load_state_dict(state0)
    losses_e = [float(tr_e.train_step(xs[i], ys[i], 0.75, 10 + i)) for i in range(3)]
    w_e = net_e.store.arena.detach().cpu().clone()
    st_e = net_e.store.state_arena.detach().cpu().clone()

    net_c, tr_c = make()
    net_c.store.load_state_dict(state0)
    tr_c.capture_step(xs[0], ys[0], 0.75)                      # two real warm-up steps + the recording
    assert tr_c.global_step == 2 and tr_c.opt.t == 2
    net_c.store.load_state_dict(state0)                        # back to the start: weights, BN moving statistics, Adam slots, counters
    tr_c.opt.m.zero_(); tr_c.opt.v.zero_(); tr_c.opt.t = 0; tr_c.global_step = 0
    losses_c = [float(tr_c.train_step(xs[i], ys[i], 0.75, 10 + i)) for i in range(3)]
    assert tr_c._cap["step"].replays == 3 and tr_c.opt.t == 3 and tr_c.global_step == 3
    assert losses_c == losses_e, (losses_c, losses_e)
    assert torch.equal(net_c.store.arena.detach().cpu(), w_e)              # three Adam updates: identical weights
    assert torch.equal(net_c.store.state_arena.detach().cpu(), st_e)       # and identical BN moving statistics
    # the seed really reaches the captured kernels: another seed, another mask, another loss
    l_a = float(tr_c.train_step(xs[0], ys[0], 0.75, 100))
    net_c.store.load_state_dict(state0)
    l_b = float(tr_c.train_step(xs[0], ys[0], 0.75, 101))
    assert l_a != l_b
    # a batch of another shape falls back to the eager step
    n0 = tr_c._cap["step"].replays
    tr_c.train_step(xs[0][:1].repeat(4, 1, 1, 1), ys[0][:1].repeat(4, 1, 1, 1), 0.75, 5)
    assert tr_c._cap["step"].replays == n0


def test_captured_gan_steps_equal_eager_bit_for_bit(dev):
    adv = pkg("adversarial")
    from test_gpu_adversarial import COST as GCOST, NETCFG, he_state
    B = 2
    rng = np.random.default_rng(0)
    mr = torch.from_numpy(rng.standard_normal((B, 256, 256, 3)).astype(np.float32)).to(dev)
    ct = torch.from_numpy((rng.standard_normal((B, 256, 256, 3)) * 1.2 + 0.1).astype(np.float32)).to(dev)

    def make():
        net = adv.Full_DRN(channels=3, n_class=5, batch_size=B, cost_kwargs=dict(GCOST), network_config=dict(NETCFG), device=dev, seed=1)
        tr = adv.Trainer(net, None, None, None, None, num_cls=5, batch_size=B, opt_kwargs={"learning_rate": 3e-4},
                         train_config={"dis_sub_iter": 1, "gen_sub_iter": 1})
        tr._get_optimizer()
        return net, tr
    net_e, tr_e = make()
    sd = he_state(net_e, 7)
    net_e.store.load_state_dict(sd)

    def run(tr):
        out = []
        for i in range(2):
            out.append(float(tr.dis_step(mr, ct, 0.75, 2 * i + 1)))
            out.append(float(tr.gen_step(ct, 0.75, 2 * i + 2)))
        return out
    le = run(tr_e)
    w_e = net_e.store.arena.detach().cpu().clone()
    net_c, tr_c = make()
    net_c.store.load_state_dict(sd)
    tr_c.capture_steps(mr, ct, 0.75)
    # A workspace that has to grow AFTER a step was recorded (another batch shape, the other step's warm-up) must not pull the recorded
    # address from under the hipGraph (round 6: the bf16 joint step at B = 16 faulted on replay — the generator step's warm-up had grown
    # the filter-gradient side stream's buffer, and the discriminator graph still wrote into the freed one).  Force it: grow the side
    # stream's buffer, hand every cached block back to the driver, overwrite what the allocator gives out next.
    K, F = pkg("kernels"), pkg("functional")
    side = F.wgrad_side_stream()
    assert side is not None and K.PINNED[0] >= 2
    old = K._ws_cache[(dev.index, "main", side.cuda_stream)]
    p_old, n_old = old.data_ptr(), old.numel()
    del old
    with torch.cuda.stream(side):
        K.workspace(2 * n_old + (64 << 20), dev)
    assert any(r.data_ptr() == p_old for r in K._ws_retired)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = torch.full((n_old + (64 << 20),), 0x7F, dtype=torch.uint8, device=dev)
    net_c.store.load_state_dict(sd)
    for o in (tr_c.dis_optimizer, tr_c.gen_optimizer):          # RMSProp slots back to their initial value (ones: TF's initial ms)
        o.ms.copy_(torch.ones_like(o.ms))
    lc = run(tr_c)
    assert tr_c._cap["dis"].replays == 2 and tr_c._cap["gen"].replays == 2
    assert lc == le, (lc, le)
    assert torch.equal(net_c.store.arena.detach().cpu(), w_e)
    assert int(junk.min()) == 0x7F and int(junk.max()) == 0x7F          # nothing replayed into memory the recordings no longer own
