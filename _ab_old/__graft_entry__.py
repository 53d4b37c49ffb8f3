"""Driver entry points: build() compiles every HIP source for gfx950 in-tree; smoke() runs one tiny
fill_gridmap + forward('navigation') on cuda:0 and checks it against the CPU oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build():
    """hipcc --offload-arch=gfx950 for gridmm_amd/csrc/*.hip -> gridmm_amd/libgridmm_hip.so (cross-compiles
    without a GPU), then the oracle's C restatement and, when /root/reference is present, nothing from it needs
    compiling (the reference is pure Python).  Finally import the package and bind the C-ABI."""
    env = dict(os.environ)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "gridmm_amd", "csrc"), "-j8"], env=env)
    omk = os.path.join(ROOT, "oracle", "Makefile")
    if os.path.exists(omk):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], env=env)
    import gridmm_amd  # noqa: F401
    from gridmm_amd import _lib
    _lib.load()


def smoke():
    import json
    import numpy as np
    import torch
    from gridmm_amd import _lib, synthetic as S
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    from oracle import navcmt_oracle as O, gridmap_oracle as G
    from oracle.ref_harness import det_tensor

    if not os.path.exists(_lib.LIB_PATH):
        build()
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    dev = torch.device("cuda:0")
    cfg = default_config(num_l_layers=1, num_pano_layers=1, num_x_layers=2, intermediate_size=256, vocab_size=1000)
    model = GlocalTextPathNavCMT(cfg).eval()
    sd = {k: (det_tensor(k, v.shape, 1) if v.dtype.is_floating_point else v) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    model.to(dev)

    rs = np.random.RandomState(0)
    B, T = 2, 2
    mem = GridMemoryBatch(B, S.NATIVE, max_steps=T, device=dev)
    oracles = [G.GridMemory(G.NATIVE) for _ in range(B)]
    eps = [S.make_observations(rs, S.NATIVE, T, feat_scale=0.35) for _ in range(B)]
    for t in range(T):
        mem.step(np.stack([e[t]["depth"].reshape(-1) for e in eps]), np.stack([e[t]["feats"] for e in eps]),
                 [(e[t]["x"], e[t]["y"]) for e in eps], [e[t]["heading"] for e in eps])
        ref = [oracles[b].step(eps[b][t]["depth"], eps[b][t]["feats"], eps[b][t]["x"], eps[b][t]["y"],
                               eps[b][t]["heading"]) for b in range(B)]
    for b in range(B):
        n = ref[b][1].shape[0]
        assert np.array_equal(mem.cell_id[b, :n].cpu().numpy(), ref[b][1].astype(np.int16)), "cell ids differ"
    batch = S.make_nav_batch(rs, B, L=20, G=8, n_visited=3, V1=10, n_cand=3, min_len=8)
    cpu = dict(batch, grid_fts=[torch.from_numpy(r[0]) for r in ref], grid_map=[torch.from_numpy(r[1]) for r in ref],
               gridmap_pos_fts=torch.from_numpy(np.stack([r[2] for r in ref])))
    with torch.no_grad():
        want = O.forward_navigation(sd, cpu)
    got = model("navigation", dict(S.batch_to(batch, dev), grid_memory=mem, grid_fts=None, grid_map=None,
                                   gridmap_pos_fts=None))
    torch.cuda.synchronize()
    worst = 0.0
    for k in ("global_logits", "local_logits", "fused_logits", "grid_logits"):
        a, w = got[k].cpu(), want[k]
        f = torch.isfinite(w)
        assert torch.equal(f, torch.isfinite(a)), k
        worst = max(worst, float((a[f] - w[f]).abs().max()))
    assert worst < 1e-3, worst
    print(json.dumps({"smoke": "ok", "max_logit_err_vs_oracle": worst}))


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
