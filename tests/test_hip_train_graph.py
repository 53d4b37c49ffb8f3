"""GPU: the pre-training step as one hipGraph (train_graph.GraphedTrainStep) against the eager step it captures.  The
cases (tests/train_graph_cases.py) run in a subprocess whose environment carries DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, which
must be in place before the HIP runtime starts (gridmm_amd/train_graph.py explains why)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(case, env_extra):
    env = dict(os.environ, **env_extra)
    return subprocess.run([sys.executable, os.path.join(HERE, "train_graph_cases.py"), case], env=env,
                          capture_output=True, text=True, timeout=1200)


@pytest.mark.parametrize("case", ["mlm", "mrc", "sap", "dropout", "two_graphs", "fp16_grid_proj", "full_size", "trajectory",
                                  "segments_sap", "segments_mlm", "dist2"])
def test_graphed_training_step(case):
    r = _run(case, {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"})
    assert r.returncode == 0 and ("ok " + case) in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_refuses_without_the_runtime_setting():
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    r = subprocess.run([sys.executable, os.path.join(HERE, "train_graph_cases.py"), "mlm"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0 and "DEBUG_CLR_GRAPH_PACKET_CAPTURE" in r.stderr
