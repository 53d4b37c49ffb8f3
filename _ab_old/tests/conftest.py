import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")
    torch.set_num_threads(int(os.environ.get("GRIDMM_TEST_THREADS", "1")))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_state_dict(fx, device="cpu"):
    """Regenerate the deterministic weights a nav_* fixture was produced with."""
    from oracle.ref_harness import det_tensor
    names = json.loads(str(fx["param_names"]))
    shapes = json.loads(str(fx["param_shapes"]))
    seed = int(fx["weight_seed"])
    sd = {}
    for k, s in zip(names, shapes):
        if k.endswith("position_ids"):
            sd[k] = torch.arange(s[-1]).view(*s)
        else:
            sd[k] = det_tensor(k, s, seed)
        sd[k] = sd[k].to(device)
    return sd


def golden_nav_batch(fx, device="cpu"):
    """Rebuild the navigation batch dict stored in a nav_reduced*.npz fixture."""
    batch = {"vp_obj_masks": None}
    B = None
    for k in fx.files:
        if not k.startswith("in_"):
            continue
        name = k[3:]
        if name in ("gmap_vpids", "vp_cand_vpids"):
            batch[name] = json.loads(str(fx[k]))
        elif name.startswith("grid_fts_") or name.startswith("grid_map_"):
            continue
        else:
            batch[name] = torch.from_numpy(fx[k]).to(device)
    B = batch["txt_embeds"].shape[0]
    batch["grid_fts"] = [torch.from_numpy(fx["in_grid_fts_%d" % b]).to(device) for b in range(B)]
    batch["grid_map"] = [torch.from_numpy(fx["in_grid_map_%d" % b]).to(device) for b in range(B)]
    return batch


@pytest.fixture(scope="session")
def has_reference():
    return os.path.isdir("/root/reference/map_nav_src")
