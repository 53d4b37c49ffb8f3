"""(run by tests/test_hip_train_graph.py in a subprocess with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, see train_graph.py)
GPU: the pre-training step as one hipGraph (train_graph.GraphedTrainStep) against the eager step it captures
(pretrain_loop.PreTrainer.train_step; reference loop pretrain_src/train_r2r.py:231-303): same losses, same gradient norms,
same parameters after several optimizer steps (warm-up lr schedule and AdamW bias correction advancing per replay),
for each of the three tasks; with dropout on, replays draw new masks and two identically seeded runs agree."""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TASKS = ("mlm", "mrc", "sap")


def _setup(drop):
    from gridmm_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from gridmm_amd.synthetic import batch_to, make_pretrain_batch
    from gridmm_amd.vilmodel import default_config
    dev = torch.device("cuda")
    cfg = default_config(use_lang2visn_attn=True, pretrain_tasks=list(TASKS), image_prob_size=1000, obj_prob_size=0,
                         num_l_layers=2, num_pano_layers=1, num_x_layers=2, hidden_dropout_prob=drop,
                         attention_probs_dropout_prob=drop)
    torch.manual_seed(0)
    model = GlocalTextPathCMTPreTraining(cfg).to(dev)
    batches = {t: batch_to(make_pretrain_batch(np.random.RandomState(i), 4, t, max_steps=3, L=40, vocab=30000,
                                               image_prob_size=1000, n_pts=(588, 588 * 2)), dev)
               for i, t in enumerate(TASKS)}
    return model, batches


def case_equals_eager(task):
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.train_graph import GraphedTrainStep
    model, batches = _setup(0.0)
    ma, mb = copy.deepcopy(model), copy.deepcopy(model)
    ta, tb = PreTrainer(ma, default_opts(warmup_steps=10)), PreTrainer(mb, default_opts(warmup_steps=10))
    for _ in range(2):                                   # what the graphed object runs while it is built: record + warm-up
        ta.train_step(batches[task], task)
    g = GraphedTrainStep(tb, batches[task], task)
    assert tb.global_step == ta.global_step == 2
    for _ in range(4):
        la, na = ta.train_step(batches[task], task)
        lb, nb = g()
        assert torch.isfinite(lb).all()
        assert torch.allclose(la, lb, rtol=2e-5, atol=2e-5), float((la - lb).abs().max())
        assert abs(float(na) - float(nb)) <= 2e-5 * float(na)
    assert tb.global_step == ta.global_step == 6
    for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
        tol = 2e-3 if pa.dtype == torch.float16 else 1e-4     # (fp16 grid_proj: one ulp of its rounding)
        assert torch.allclose(pa.float(), pb.float(), rtol=0, atol=tol), (n, float((pa.float() - pb.float()).abs().max()))
    # the optimizer state advanced like the eager one's
    sa, sb = ta.optimizer.state, tb.optimizer.state
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        if pa in sa:
            assert sa[pa]["step"] == sb[pb]["step"]
    # eager use after replays sees the replayed weights (packed-weight caches are re-validated)
    la, _ = ta.train_step(batches[task], task)
    lb, _ = tb.train_step(batches[task], task)
    assert torch.allclose(la, lb, rtol=5e-4, atol=5e-4)


def case_dropout():
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.train_graph import GraphedTrainStep
    model, batches = _setup(0.1)
    runs = []
    for _ in range(2):
        m = copy.deepcopy(model)
        tr = PreTrainer(m, default_opts(warmup_steps=10, learning_rate=0.0))     # lr 0: only the masks differ between replays
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        g = GraphedTrainStep(tr, batches["sap"], "sap")
        runs.append([g()[0].clone() for _ in range(3)])
    a, b = runs
    assert all(torch.isfinite(x).all() for x in a)
    assert not torch.equal(a[0], a[1]) and not torch.equal(a[1], a[2])          # new dropout masks every replay
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-5)                       # same seeds -> same masks


if __name__ == "__main__":
    case = sys.argv[1]
    if case == "dropout":
        case_dropout()
    else:
        case_equals_eager(case)
    print("ok", case)
