"""-m gpu, round 6 (ADVICE r5 on the captured fine-tuning step):
  * GraphedLossStep pins the net's PackArena: forward-only calls between two replays cannot free the buffers the graph replays on;
  * an eager optimizer's zero_grad(set_to_none=True) between replays does not lose the captured gradients;
  * an optimizer that has already trained keeps its state through the capture's warm-up;
  * the differentiable last DDIM step does not split input channels by batch (a trajectory's bits do not depend on its batch).
Gates: bit equality where the same kernels run on the same buffers."""
import pytest
import torch

import safediffcon_amd as sdc
from oracle.detweights import det_params, det_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _spec(net):
    return [(k, tuple(v.shape)) for k, v in net.state_dict().items()]


def _tokamak(dim=32, seed=61, timesteps=50):
    net = sdc.Unet1D(dim=dim, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
    net.load_state_dict(det_params(_spec(net), seed))
    net.to(DEV)
    return net, sdc.GaussianDiffusionTokamak(net, seq_length=128, nt=122, timesteps=timesteps).to(DEV)


def _inputs(B, shape, seed):
    state = det_tensor((B, *shape), seed, 0.3).to(DEV)
    w = (det_tensor((B,), seed + 1, 0.2) + 1.0).to(DEV)
    t = torch.randint(0, 50, (B,), generator=torch.Generator().manual_seed(seed + 2)).to(DEV)
    noise = det_tensor((B, *shape), seed + 3).to(DEV)
    return state, w, t, noise


def test_graphed_step_survives_forward_only_calls_between_replays():
    """three forward-only forward_train calls on the same net used to let PackArena.begin() drop the flipped
    (data-gradient) layouts and reallocate buffer + table under the captured sdc_pack_batch_run.  Now the arena is frozen while a
    GraphedLossStep lives: same tensors before and after, and the replay still equals the eager step bit for bit."""
    net, gd = _tokamak()
    state, w, t, noise = _inputs(8, (12, 128), 62)
    params = [p for p in net.parameters() if p.requires_grad]
    step = sdc.GraphedLossStep(gd, state, weight=w, t=t, noise=noise)
    arena = net._trainer().arena
    assert arena.frozen == 1
    buf_ptr, tab_ptr, n_keys = arena.buf.data_ptr(), arena.table.data_ptr(), len(arena.index)
    l1 = step().clone()
    g1 = [g.clone() for g in step.grads]
    with torch.no_grad():
        for _ in range(4):                                   # validation-style calls: no backward, the flip layouts go idle
            gd.p_losses(state, t, noise=noise, mean=False)
    # junk allocations that would land on a freed arena buffer
    junk = [torch.full((arena.buf.numel(),), float("nan"), device=DEV) for _ in range(3)]
    assert (arena.buf.data_ptr(), arena.table.data_ptr(), len(arena.index)) == (buf_ptr, tab_ptr, n_keys)
    l2 = step().clone()
    assert torch.equal(l1, l2) and all(torch.equal(a, b) for a, b in zip(g1, step.grads))
    for p in params:
        p.grad = None
    le = (w * gd.p_losses(state, t, noise=noise, mean=False)).mean()
    le.backward()
    assert torch.equal(le.detach(), l2) and all(torch.equal(p.grad, g) for p, g in zip(params, g1))
    del junk
    step.close()
    assert arena.frozen == 0
    with pytest.raises(RuntimeError):
        step()


def test_graphed_step_reseats_grads_after_zero_grad():
    """optimizer.zero_grad() (set_to_none=True by default; the reference's loops call it every iteration) drops
    p.grad; the next replay seats the captured gradient tensors again so an eager optimizer.step() moves the weights"""
    net, gd = _tokamak(seed=63)
    state, w, t, noise = _inputs(4, (12, 128), 64)
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-4)
    step = sdc.GraphedLossStep(gd, state, weight=w, t=t, noise=noise)
    step()
    opt.zero_grad()
    assert all(p.grad is None for p in params)
    before = [p.detach().clone() for p in params]
    step()
    assert all(p.grad is g for p, g in zip(params, step.grads))
    opt.step()
    moved = sum(int(not torch.equal(a, p.detach())) for a, p in zip(before, params))
    assert moved > len(params) // 2


def test_graphed_step_keeps_state_of_a_trained_optimizer():
    """an optimizer that has already stepped keeps its moments and step count through the capture's warm-up step"""
    net, gd = _tokamak(seed=65)
    state, w, t, noise = _inputs(4, (12, 128), 66)
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-5, capturable=True)
    (w * gd.p_losses(state, t, noise=noise, mean=False)).mean().backward()
    opt.step()
    opt.step()
    snap = {p: {k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)} for p, st in opt.state.items()}
    assert all(float(s["step"]) == 2.0 for s in snap.values())
    weights = [p.detach().clone() for p in params]
    step = sdc.GraphedLossStep(gd, state, weight=w, t=t, noise=noise, optimizer=opt)
    for p, st in opt.state.items():                          # (capture itself replays nothing)
        for k, v in snap[p].items():
            assert torch.equal(st[k], v), k
    assert all(torch.equal(a, p.detach()) for a, p in zip(weights, params))
    step()
    assert all(float(st["step"]) == 3.0 for st in opt.state.values())


def test_differentiable_last_step_is_batch_invariant():
    """sample(enable_grad=True) runs its last DDIM step through forward_train; inside it conv_raw must not take the
    batch-dependent split over input channels, so sample 0 of a batch of 2 equals sample 0 of a batch of 16 bit for bit"""
    from safediffcon_amd import autograd
    net = sdc.Unet2D(dim=16, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
    net.load_state_dict(det_params(_spec(net), 67))
    net.to(DEV)
    x = det_tensor((16, 3, 16, 128), 68).to(DEV)
    tt = torch.full((16,), 7, device=DEV, dtype=torch.long)
    with autograd.batch_invariant():
        big = net.forward_train(x, tt).detach()
        small = net.forward_train(x[:2].contiguous(), tt[:2]).detach()
    assert torch.equal(big[:2], small)
    # and the fine-tuning path still may split (documented: gradients are gated at 1e-9 MSE against the reference, not bitwise)
    assert autograd.SPLIT_SMALL_GRIDS
