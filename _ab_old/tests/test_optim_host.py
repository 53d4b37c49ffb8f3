"""Host side of the pre-training optimizer against the reference-generated golden (tests/golden/optim_reduced.npz):
learning-rate schedule (pretrain_src/optim/sched.py:17-30) and the decay / no-decay grouping by parameter name
(optim/misc.py:12-37).  The update itself is a HIP kernel: tests/test_hip_optim.py."""
import json
from types import SimpleNamespace

from conftest import load_golden


def test_lr_schedule_and_param_groups_match_reference():
    from oracle import gen_golden as GG
    from gridmm_amd.optim import build_optimizer, get_lr_sched, warmup_linear
    fx = load_golden("optim_reduced.npz")
    o = json.loads(str(fx["cfg"]))
    opts = SimpleNamespace(optim="adamw", learning_rate=o["learning_rate"], betas=o["betas"], weight_decay=o["weight_decay"],
                           warmup_steps=o["warmup_steps"], num_train_steps=o["num_train_steps"])
    for step in range(1, o["steps"] + 1):
        assert abs(get_lr_sched(step, opts) - float(fx["lr"][step - 1])) < 1e-15
    assert get_lr_sched(o["num_train_steps"], opts) == 1e-8 and get_lr_sched(o["num_train_steps"] + 5, opts) == 1e-8   # floor
    assert warmup_linear(0, 3, 9) == 0.0
    model = GG.OptimToy()
    opt = build_optimizer(model, opts)
    groups = [[n for n, p in model.named_parameters() if any(p is q for q in g["params"])] for g in opt.param_groups]
    assert groups == json.loads(str(fx["decay"]))
    assert "LayerNorm.weight" in groups[1] and "dense.bias" in groups[1] and "dense.weight" in groups[0]
