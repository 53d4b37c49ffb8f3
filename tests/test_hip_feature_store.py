"""GPU, SURVEY §8 row f3 end to end: reference-format HDF5 files (written by h5py / libhdf5, tests/golden/hdf5/) ->
convert_reference_files -> PackedStore.gather -> pinned host buffers -> async upload -> GridMemoryBatch.step on the device:
cell ids bit-exact vs the oracle grid memory fed with the reference's own slices of the same files
(map_nav_src/r2r/env.py:80-113, 279-303), navigation logits equal to the reference's list form and to the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
HERE = os.path.join(ROOT, "tests", "golden", "hdf5")


def test_store_to_device_grid_memory_matches_oracle_and_list_form(tmp_path):
    from gridmm_amd import feature_store as FS, synthetic as S
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    from oracle import gridmap_oracle as G, navcmt_oracle as O
    from oracle.ref_harness import det_tensor
    dev = torch.device("cuda")
    out = os.path.join(str(tmp_path), "obs.gmm")
    info_path = os.path.join(HERE, "viewpoint_info_small.json")
    FS.convert_reference_files(os.path.join(HERE, "clip_small.hdf5"), os.path.join(HERE, "depth_small.hdf5"), info_path, out)
    st = FS.PackedStore(out)
    info = json.load(open(info_path))
    keys = sorted(info)
    assert len(keys) == 3
    clip, depth = np.load(os.path.join(HERE, "clip_small.npz")), np.load(os.path.join(HERE, "depth_small.npz"))

    # two episodes walking the three viewpoints in different orders, headings in 30-degree steps
    walks = [[keys[0], keys[1], keys[2]], [keys[2], keys[0], keys[1]]]
    heads = [[0.0, np.pi / 6, -np.pi / 3], [np.pi / 2, np.pi, 5 * np.pi / 6]]
    B, T = 2, 3
    mem = GridMemoryBatch(B, S.NATIVE, max_steps=T, device=dev)
    oracles = [G.GridMemory(G.NATIVE) for _ in range(B)]
    pin_d = torch.empty(B, 588, dtype=torch.uint16).pin_memory()
    pin_f = torch.empty(B, 588, 768, dtype=torch.float16).pin_memory()
    ref = None
    for t in range(T):
        d, f, poses = st.gather([walks[b][t] for b in range(B)])
        pin_d.numpy()[:] = d
        pin_f.numpy()[:] = f
        mem.step(pin_d.to(dev, non_blocking=True), pin_f.to(dev, non_blocking=True), poses, [heads[b][t] for b in range(B)])
        torch.cuda.synchronize()           # the pinned buffers are rewritten next step
        ref = []
        for b in range(B):
            key = walks[b][t]
            # the reference's own expressions on the raw file contents
            sem = clip[key][:, :50].astype(np.float16)[:, 1:].reshape(-1, 768)
            full = depth[key][:, :128 * 128].astype(np.uint16).reshape(36, 128, 128)
            ref.append(oracles[b].step(G.sample_depth(full, G.NATIVE, slice(12, 24)), sem, info[key]["x"], info[key]["y"],
                                       heads[b][t]))
        for b in range(B):
            n = ref[b][1].shape[0]
            assert np.array_equal(mem.cell_id[b, :n].cpu().numpy(), ref[b][1].astype(np.int16)), (t, b)
            assert np.array_equal(mem.slab[b, :n].cpu().numpy(), ref[b][0])
            assert np.abs(mem.pos_fts[b].cpu().numpy() - ref[b][2]).max() < 2e-6

    # the same walk with the store resident in HBM (DeviceStore: observations gathered on the device, no PCIe traffic)
    ds = FS.DeviceStore.from_packed(st, dev)
    mem2 = GridMemoryBatch(B, S.NATIVE, max_steps=T, device=dev)
    mem2.track_cmax = True             # the occupied-cell count follows every step to a pinned word (varlen hint)
    for t in range(T):
        d, poses = ds.append(mem2, [walks[b][t] for b in range(B)])
        mem2.step(d, None, poses, [heads[b][t] for b in range(B)])
    torch.cuda.synchronize()
    assert torch.equal(mem2.cell_id, mem.cell_id) and torch.equal(mem2.slab, mem.slab) and torch.equal(mem2.perm, mem.perm)
    assert torch.equal(mem2.pos_fts, mem.pos_fts)

    cfg = default_config(num_l_layers=1, num_pano_layers=1, num_x_layers=2, intermediate_size=256, vocab_size=1000)
    model = GlocalTextPathNavCMT(cfg).eval()
    sd = {k: (det_tensor(k, v.shape, 1) if v.dtype.is_floating_point else v) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    model.to(dev)
    batch = S.make_nav_batch(np.random.RandomState(7), B, L=20, G=8, n_visited=3, V1=10, n_cand=3, min_len=8)
    with torch.no_grad():
        got = model("navigation", dict(S.batch_to(batch, dev), grid_memory=mem, grid_fts=None, grid_map=None,
                                       gridmap_pos_fts=None))
        fts, gmaps, pos = mem.as_reference_obs()
        lst = model("navigation", dict(S.batch_to(batch, dev), grid_fts=fts, grid_map=gmaps, gridmap_pos_fts=pos))
        want = O.forward_navigation(sd, dict(batch, grid_fts=[torch.from_numpy(r[0]) for r in ref],
                                             grid_map=[torch.from_numpy(r[1]) for r in ref],
                                             gridmap_pos_fts=torch.from_numpy(np.stack([r[2] for r in ref]))))
        # varlen map sequence chosen from the memory's tracked count (no read-back inside the call)
        occupied = max(len(np.unique(r[1][r[1] >= 0])) for r in ref)
        assert mem2.cmax_hint() == occupied
        model.varlen_buckets = tuple(sorted({min(196, (occupied + 15) // 16 * 16), 196}))
        var = model("navigation", dict(S.batch_to(batch, dev), grid_memory=mem2, grid_fts=None, grid_map=None,
                                       gridmap_pos_fts=None))
        model.varlen_buckets = None
    for k in ("global_logits", "local_logits", "fused_logits", "grid_logits"):
        f = torch.isfinite(got[k])
        assert torch.equal(f, torch.isfinite(var[k])) and (got[k][f] - var[k][f]).abs().max() < 2e-5, k
    for k in ("global_logits", "local_logits", "fused_logits", "grid_logits"):
        a, l, w = got[k].cpu(), lst[k].cpu(), want[k]
        f = torch.isfinite(w)
        assert torch.equal(f, torch.isfinite(a)) and torch.equal(f, torch.isfinite(l)), k
        assert (a[f] - l[f]).abs().max() < 1e-5, k          # device-resident memory == the reference's list form
        assert (a[f] - w[f]).abs().max() < 2e-4, k          # == the oracle on the reference's slices
