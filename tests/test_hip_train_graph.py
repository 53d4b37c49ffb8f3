"""GPU: the pre-training step as one hipGraph (train_graph.GraphedTrainStep) against the eager step it captures.  The
cases (tests/train_graph_cases.py) run in subprocesses on the runtime's DEFAULT settings (pre-recorded graph packets): the
graphs hold kernel nodes only, which is what the default needs (gridmm_amd/train_graph.py, KERNEL NODES ONLY); one case
keeps DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 covered."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(case, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "GRIDMM_TRAIN_GRAPH_ANY_RUNTIME")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(HERE, "train_graph_cases.py"), case], env=env,
                          capture_output=True, text=True, timeout=1200)


@pytest.mark.parametrize("case", ["mlm", "mrc", "sap", "dropout", "two_graphs", "fp16_grid_proj", "full_size", "trajectory",
                                  "segments_sap", "segments_mlm", "dist2", "rccl1", "alternate_full", "nonkernel"])
def test_graphed_training_step(case):
    r = _run(case, {})
    assert r.returncode == 0 and ("ok " + case) in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_graphed_training_step_with_packet_capture_off():
    r = _run("sap", {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"})
    assert r.returncode == 0 and "ok sap" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
