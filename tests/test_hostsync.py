"""CPU: hostsync -- the tape of host decisions that a captured training step replays (gridmm_amd/train_graph.py)."""
import pytest
import torch

from gridmm_amd import hostsync as hs


def _decisions(x, mask, calls):
    def longest():
        calls.append("max")
        return int(x.abs().sum(1).argmax())
    k = hs.host(longest)
    rows = hs.select(x, mask)
    return k, rows


def test_record_then_replay_returns_the_taped_values_without_running_the_callables():
    x = torch.arange(12.0).reshape(4, 3)
    mask = torch.tensor([True, False, True, True])
    calls = []
    k0, r0 = _decisions(x, mask, calls)                 # plain mode: runs where it stands, x[mask] exactly
    assert calls == ["max"] and k0 == 3 and torch.equal(r0, x[mask])
    with hs.record() as tape:
        k1, r1 = _decisions(x, mask, calls)
    assert calls == ["max", "max"] and k1 == 3 and torch.equal(r1, x[mask]) and len(tape) == 2
    y = x * 2                                           # same metadata (mask), new data
    with hs.replay(tape):
        k2, r2 = _decisions(y, ~mask, calls)            # the mask is NOT looked at again: the taped row ids are used
    assert calls == ["max", "max"] and k2 == 3 and torch.equal(r2, y[mask])
    assert hs.MODE is None


def test_select_is_differentiable_like_boolean_indexing():
    x = torch.randn(5, 4, requires_grad=True)
    mask = torch.tensor([False, True, True, False, True])
    with hs.record() as tape:
        hs.select(x, mask).square().sum().backward()
    g_rec = x.grad.clone()
    x.grad = None
    x[mask].square().sum().backward()
    assert torch.equal(g_rec, x.grad)
    x.grad = None
    with hs.replay(tape):
        hs.select(x, mask).square().sum().backward()
    assert torch.equal(g_rec, x.grad)


def test_replay_checks_that_the_whole_tape_was_consumed():
    with hs.record() as tape:
        hs.host(lambda: 1)
        hs.host(lambda: 2)
    with pytest.raises(AssertionError):
        with hs.replay(tape):
            assert hs.host(lambda: 0) == 1
    assert hs.MODE is None
    with hs.replay(tape):
        assert hs.host(lambda: 0) == 1 and hs.host(lambda: 0) == 2
