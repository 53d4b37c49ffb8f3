"""GPU: deferred parameter gradients (autograd.deferred_param_grads / flush_param_grads, gridmm_multi_grad_accumulate).
A fine-tuning iteration runs ONE backward through all navigation steps of a rollout (map_nav_src/r2r/agent_base.py:190-199):
every parameter receives one gradient per step.  Pinned here: summing them with one multi-tensor launch gives the .grad
that autograd's sequential AccumulateGrad adds give, BIT for bit -- for 3 and for 9 uses of a parameter (more than the 7
sources of one record), for the row blocks of a fused q | k | v group, for LayerNorm parameters, and for a parameter that
also receives a gradient through a plain torch op."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(steps, defer):
    from gridmm_amd import autograd as ag
    dev = torch.device("cuda")
    torch.manual_seed(0)
    ag.WEIGHTS.clear()
    w = torch.nn.Parameter(torch.randn(192, 128, device=dev) * 0.05)
    b = torch.nn.Parameter(torch.randn(192, device=dev) * 0.05)
    q, k, v = (torch.nn.Parameter(torch.randn(128, 128, device=dev) * 0.05) for _ in range(3))
    qb, kb, vb = (torch.nn.Parameter(torch.randn(128, device=dev) * 0.05) for _ in range(3))
    ln = torch.nn.LayerNorm(192).to(dev)
    odd = torch.nn.Parameter(torch.randn(128, device=dev))           # used by a torch op AND nothing else
    params = [w, b, q, k, v, qb, kb, vb, ln.weight, ln.bias, odd]
    loss = 0
    for t in range(steps):
        x = torch.randn(3, 9, 128, device=dev) + odd
        y = ag.layer_norm(ag.linear(x, w, b), ln)
        z = ag.linear_group(x, [q, k, v], [qb, kb, vb])
        loss = loss + (y * y).sum() + z.pow(3).sum() + (b * b).sum()         # b: kernel gradient + a plain torch gradient
    if defer:
        with ag.deferred_param_grads():
            loss.backward()
        ag.flush_param_grads()
    else:
        loss.backward()
    torch.cuda.synchronize()
    return [p.grad.clone() for p in params]


@pytest.mark.parametrize("steps", [1, 3, 9, 16])
def test_deferred_sum_is_bit_identical_to_sequential_accumulation(steps):
    ref = _run(steps, False)
    got = _run(steps, True)
    for i, (a, c) in enumerate(zip(ref, got)):
        if i == 1:          # b: its torch-op gradients are summed before the kernels' ones instead of interleaved with them
            assert torch.allclose(a, c, rtol=1e-5, atol=1e-6), (steps, (a - c).abs().max().item())
        else:
            assert torch.equal(a, c), (steps, i, (a - c).abs().max().item())


def test_nothing_pending_after_a_flush_and_outside_the_region():
    from gridmm_amd import autograd as ag
    _run(2, True)
    assert not ag.DEFERRED.pending and not ag.DEFERRED.active
    _run(2, False)
    assert not ag.DEFERRED.pending
