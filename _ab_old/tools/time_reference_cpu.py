#!/usr/bin/env python
"""Times the REFERENCE ITSELF (MrZihan/GridMM, imported read-only from /root/reference through oracle/ref_harness.py)
on this host's CPU cores -- build container only; the reference's files never travel to the GPU box.

What is timed, at the reference's NATIVE shape (12 views x 49 patches x 768-D observations: the
reference hard-codes 768 and cannot take the 512-D BASELINE slab), full-size model (161 M parameters, random init):
  * EnvBatch.getGlobalMap + get_gridmap_pos_fts (map_nav_src/r2r/env.py:242-374), B episodes at memory depth t
    (NumPy, one episode after the other, as env.py:392-398 does);
  * GlocalTextPathNavCMT.forward('navigation') (map_nav_src/models/vilmodel.py:782-918) on the same B episodes,
    torch CPU, torch.set_num_threads(k) for k in {1, all cores}.
Output: one JSON object (steps/s = B / (fill + forward seconds) for the best k) + `lscpu` model name; paste into
BASELINE.md §5.   usage: python tools/time_reference_cpu.py [--batch 8] [--mem-steps 5] [--repeats 3]
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as R, gen_golden as GG                    # noqa: E402
from gridmm_amd import synthetic as S                                    # noqa: E402


def cpu_model():
    try:
        for line in subprocess.check_output(["lscpu"], text=True).splitlines():
            if line.startswith("Model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--mem-steps", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=3)
    a = ap.parse_args()
    assert R.reference_available(), "needs /root/reference (build container)"
    B, t = a.batch, a.mem_steps
    rs = np.random.RandomState(0)
    eps = [S.make_observations(rs, S.NATIVE, t, feat_scale=0.35) for _ in range(B)]
    depth_db, clip_db, info = {}, {}, {}
    for e, obs in enumerate(eps):
        for k, o in enumerate(obs):
            key = "s%d_v%d" % (e, k)
            depth_db[key] = GG._full_depth(o["depth"])
            clip = np.zeros((12, 50, 768), np.float16)
            clip[:, 1:] = o["feats"].reshape(12, 49, 768)
            clip_db[key] = clip
            info[key] = {"x": o["x"], "y": o["y"]}

    def fill_all():
        """B episodes stepped to depth t through the reference's EnvBatch; the LAST step of every episode is timed
        (= one navigation step at memory depth t)."""
        env = R.RefGridEnv(B, depth_db, clip_db, info)
        for e, obs in enumerate(eps):
            for k in range(t - 1):
                env.step(e, "s%d" % e, "v%d" % k, obs[k]["heading"])
        t0 = time.perf_counter()
        outs = [env.step(e, "s%d" % e, "v%d" % (t - 1), eps[e][t - 1]["heading"]) for e in range(B)]
        return time.perf_counter() - t0, outs

    fill_s, outs = min((fill_all() for _ in range(a.repeats)), key=lambda x: x[0])
    model = R.build_ref_model(seed=0)                                       # full size: BertConfig defaults + vlnbert_init
    batch = S.make_nav_batch(rs, B, L=80, G=20, n_visited=6, V1=37, n_cand=4, min_len=30)
    batch["grid_fts"] = [torch.from_numpy(o[0]) for o in outs]
    batch["grid_map"] = [torch.from_numpy(o[1]) for o in outs]
    batch["gridmap_pos_fts"] = torch.from_numpy(np.stack([o[2] for o in outs]))
    res = {}
    ncpu = os.cpu_count() or 1
    for k in sorted({1, ncpu}):
        torch.set_num_threads(k)
        with torch.no_grad():
            model("navigation", batch)                                      # warm-up
            ts = []
            for _ in range(a.repeats):
                t0 = time.perf_counter()
                model("navigation", batch)
                ts.append(time.perf_counter() - t0)
        res[k] = float(np.median(ts))
    best_k = min(res, key=res.get)
    out = {"cpu": cpu_model(), "logical_cpus": ncpu, "batch": B, "mem_steps": t, "points_per_episode": 588 * t,
           "shape": "native 12x49x768, L=80, G=20, V=37, full-size model (161 M parameters, random init)",
           "getGlobalMap_s_per_batch": fill_s, "forward_navigation_s_per_batch": {str(k): v for k, v in res.items()},
           "best_threads": best_k, "steps_per_s": B / (fill_s + res[best_k]),
           "steps_per_s_forward_only": B / res[best_k]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
