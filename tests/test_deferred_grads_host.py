"""CPU: the host bookkeeping of autograd.deferred_param_grads (the GPU sum itself: tests/test_hip_deferred_grads.py).  Outside
the region, and for anything that is not a leaf fp32 Parameter on the GPU, hand() passes the gradient through untouched -- a
CPU run never defers (and never reaches the kernel); the region nests and restores its state on exceptions."""
import pytest
import torch


def test_hand_passes_through_outside_the_region_and_for_cpu_tensors():
    from gridmm_amd import autograd as ag
    p = torch.nn.Parameter(torch.zeros(4, 4))
    g = torch.ones(4, 4)
    assert ag.DEFERRED.hand(p, g) is g and not ag.DEFERRED.pending
    with ag.deferred_param_grads():
        assert ag.DEFERRED.active
        assert ag.DEFERRED.hand(p, g) is g                    # CPU gradient: autograd's own accumulation
        assert ag.DEFERRED.hand(p, None) is None
        assert ag.DEFERRED.hand(torch.zeros(4, 4), g) is g     # not a Parameter
        with ag.deferred_param_grads():
            pass
        assert ag.DEFERRED.active                              # the inner region restored the outer state
    assert not ag.DEFERRED.active and not ag.DEFERRED.pending
    ag.flush_param_grads()                                     # nothing pending: no library call, no error


def test_region_restores_its_state_on_an_exception():
    from gridmm_amd import autograd as ag
    with pytest.raises(RuntimeError):
        with ag.deferred_param_grads():
            raise RuntimeError("boom")
    assert not ag.DEFERRED.active
