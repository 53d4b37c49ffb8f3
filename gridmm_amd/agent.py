"""GMapNavAgent: the reference's per-episode navigation loop around forward(mode, batch).

Restates (relative to /root/reference/map_nav_src/r2r):
  _language_variable          agent.py:36-49      _panorama_feature_variable   agent.py:51-94
  _nav_gmap_variable          agent.py:96-169     _nav_vp_variable             agent.py:171-205
  _teacher_action             agent.py:207-236    make_equiv_action            agent.py:238-255
  rollout                     agent.py:268-451    (criterion: agent_base.py:141, CE sum, ignore -100)
Same call order and mode strings (language -> [panorama -> graph update -> navigation -> action] x t),
same `nav_inputs` keys / shapes / dtypes, same stop handling and best-stop-node backtrack.  What changes
is where the data lives: tensors are built on `device` directly, and when the environment owns a
device-resident grid memory it is handed to the model as batch['grid_memory'] instead of re-uploading the
whole point history every step (agent.py:168).

`vln_bert` is any callable (mode, batch) -> outputs with the reference's contract
(gridmm_amd.vilmodel.GlocalTextPathNavCMT on the GPU; tests also drive it with the CPU oracle).
"""
import contextlib
import os
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

from .graph_utils import TopoMap, TopoMapBatch


def default_args(**over):
    """The subset of r2r/parser.py the loop reads."""
    a = dict(max_action_len=15, ignoreid=-100, image_feat_size=768, angle_feat_size=4, fusion="dynamic",
             enc_full_graph=True, act_visited_nodes=False, expl_max_ratio=0.6, detailed_output=False,
             # training (r2r/parser.py): DAgger by default in scripts/run_r2r.sh
             train_alg="dagger", ml_weight=0.2, expl_sample=False, lr=1e-5, optim="adamW", feat_dropout=0.4,
             dropout=0.5)
    a.update(over)
    return SimpleNamespace(**a)


def pad_tensors(tensors, lens=None, pad=0):
    """utils/ops.py: B x [T, ...] -> (B, Tmax, ...)."""
    if lens is None:
        lens = [t.size(0) for t in tensors]
    max_len = max(lens)
    out = tensors[0].new_full((len(tensors), max_len) + tuple(tensors[0].shape[1:]), pad)
    for i, (t, l) in enumerate(zip(tensors, lens)):
        out[i, :l] = t
    return out


def gen_seq_masks(seq_lens, max_len=None):
    if max_len is None:
        max_len = int(max(seq_lens))
    return torch.arange(max_len, device=seq_lens.device).unsqueeze(0) < seq_lens.unsqueeze(1)


class GMapNavAgent:
    def __init__(self, args, env, vln_bert, device="cuda"):
        """vln_bert: the model the loop calls as vln_bert(mode, batch).  A bare GlocalTextPathNavCMT is wrapped in
        model.VLNBert so that train() applies the reference's environment feature dropout (models/model.py:19,29-31);
        a VLNBert, or any other callable (the tests drive the loop with the reference model), is used as given."""
        from .vilmodel import GlocalTextPathNavCMT
        if isinstance(vln_bert, GlocalTextPathNavCMT):
            from .model import VLNBert
            vln_bert = VLNBert(args, vln_bert=vln_bert)
        self.args, self.env, self.vln_bert = args, env, vln_bert
        self.device = torch.device(device)
        self.scanvp_cands = {}
        self.feedback = "argmax"
        self.loss = 0.0
        self.logs = {"entropy": [], "IL_loss": []}
        self.trace = None      # optional list: per-step dict(nav_inputs / nav_outs / a_t) for parity tests
        # the per-episode methods below restate the reference's collation line by line; the loop itself runs their
        # batched equivalents (collate.NavCollator: same dictionaries, ~10x less host time) unless fast_collate is off
        self.fast_collate = True
        from .collate import NavCollator
        self.collator = NavCollator(args, self.device)
        self._graphs = None

    # node / view axes padded to these sizes when the model calls are replayed from hipGraphs (one graph per shape key)
    NODE_BUCKETS = (8, 12, 16, 24, 32, 48, 64, 96, 128)
    VIEW_BUCKETS = (36, 40, 48, 64)

    def enable_graph_replay(self, model=None):
        """Inference rollouts on the HIP model: 'panorama' and the shape-dependent half of 'navigation' are replayed from
        hipGraphs keyed by shape (graph.NavigationGraphs / PanoramaGraphs); the collator pads the node and view axes to
        buckets so that a rollout touches a handful of keys.  Same logits on the real rows (masked padding)."""
        from .collate import NavCollator
        from .graph import LanguageGraphs, NavigationGraphs, PanoramaGraphs
        model = model if model is not None else getattr(self.vln_bert, "vln_bert", self.vln_bert)
        self.collator = NavCollator(self.args, self.device, node_buckets=self.NODE_BUCKETS, view_buckets=self.VIEW_BUCKETS)
        self._graphs = (PanoramaGraphs(model), NavigationGraphs(model), LanguageGraphs(model))
        self.fast_collate = True

    def _model_call(self, mode, batch):
        if self._graphs is not None and not torch.is_grad_enabled():
            if mode == "panorama":
                out = self._graphs[0](batch)
                return (out[0].clone(), out[1].clone()) if self.trace is not None else out
            if mode == "navigation":
                out = self._graphs[1](batch)
                return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()} if self.trace is not None else out
            if mode == "language" and batch["txt_ids"].is_cuda:
                return self._graphs[2](batch)
        return self.vln_bert(mode, batch)

    # ---- collation -----------------------------------------------------------------------------
    def _upload(self, a):
        """Host array -> device tensor that owns its memory, through the collator's pinned ring (one asynchronous copy; a
        `torch.from_numpy(a).to(device)` is a pageable upload that waits for the whole stream)."""
        st = self.collator.stage
        if not st.cuda:
            return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        st.flush()
        st.owned = True
        t = st.put(a)
        st.flush()
        return t

    def _language_variable(self, obs):
        lens = [len(ob["instr_encoding"]) for ob in obs]
        seq = np.zeros((len(obs), max(lens)), dtype=np.int64)
        mask = np.zeros((len(obs), max(lens)), dtype=bool)
        for i, ob in enumerate(obs):
            seq[i, :lens[i]] = ob["instr_encoding"]
            mask[i, :lens[i]] = True
        return {"txt_ids": self._upload(seq), "txt_masks": self._upload(mask)}

    def _panorama_feature_variable(self, obs):
        fs = self.args.image_feat_size
        b_img, b_loc, b_types, b_lens, b_cands = [], [], [], [], []
        for ob in obs:
            img, ang, types, cands, used = [], [], [], [], set()
            for cc in ob["candidate"]:
                img.append(cc["feature"][:fs])
                ang.append(cc["feature"][fs:])
                types.append(1)
                cands.append(cc["viewpointId"])
                used.add(int(cc["pointId"]))
            img.extend([x[:fs] for k, x in enumerate(ob["feature"]) if k not in used])
            ang.extend([x[fs:] for k, x in enumerate(ob["feature"]) if k not in used])
            types.extend([0] * (36 - len(used)))
            img, ang = np.stack(img, 0), np.stack(ang, 0)
            box = np.array([[1, 1, 1]] * len(img)).astype(np.float32)
            b_img.append(torch.from_numpy(img))
            b_loc.append(torch.from_numpy(np.concatenate([ang, box], 1)))
            b_types.append(torch.LongTensor(types))
            b_cands.append(cands)
            b_lens.append(len(img))
        return {
            "view_img_fts": pad_tensors(b_img).to(self.device), "loc_fts": pad_tensors(b_loc).to(self.device),
            "nav_types": pad_tensors(b_types).to(self.device), "view_lens": torch.LongTensor(b_lens).to(self.device),
            "cand_vpids": b_cands, "obj_img_fts": None, "obj_lens": None,
        }

    def _nav_gmap_variable(self, obs, gmaps):
        B = len(obs)
        b_vpids, b_lens, b_embeds, b_steps, b_pos, b_visited, b_pair, no_vp_left = [], [], [], [], [], [], [], []
        for i, gmap in enumerate(gmaps):
            visited, unvisited = [], []
            for k in gmap.nodes():
                if self.args.act_visited_nodes:
                    (visited if k == obs[i]["viewpoint"] else unvisited).append(k)
                else:
                    (visited if gmap.visited(k) else unvisited).append(k)
            no_vp_left.append(len(unvisited) == 0)
            if self.args.enc_full_graph:
                vpids = [None] + visited + unvisited
                vmask = [0] + [1] * len(visited) + [0] * len(unvisited)
            else:
                vpids = [None] + unvisited
                vmask = [0] * len(vpids)
            steps = [gmap.step_id.get(vp, 0) for vp in vpids]
            emb = [gmap.embedding(vp) for vp in vpids[1:]]
            emb = torch.stack([torch.zeros_like(emb[0])] + emb, 0)
            pos = gmap.pos_features(obs[i]["viewpoint"], vpids, obs[i]["heading"], obs[i]["elevation"])
            pair = gmap.pair_distances(vpids)
            b_embeds.append(emb)
            b_steps.append(torch.LongTensor(steps))
            b_pos.append(torch.from_numpy(pos))
            b_pair.append(torch.from_numpy(pair))
            b_visited.append(torch.BoolTensor(vmask))
            b_vpids.append(vpids)
            b_lens.append(len(vpids))
        lens = torch.LongTensor(b_lens)
        G = int(lens.max())
        pair_d = torch.zeros(B, G, G)
        for i in range(B):
            pair_d[i, :b_lens[i], :b_lens[i]] = b_pair[i]
        out = {
            "gmap_vpids": b_vpids, "gmap_img_embeds": pad_tensors(b_embeds),
            "gmap_step_ids": pad_tensors(b_steps).to(self.device), "gmap_pos_fts": pad_tensors(b_pos).to(self.device),
            "gmap_visited_masks": pad_tensors(b_visited, pad=False).to(self.device),
            "gmap_pair_dists": pair_d.to(self.device), "gmap_masks": gen_seq_masks(lens).to(self.device),
            "no_vp_left": no_vp_left,
        }
        out.update(self._grid_variable(obs))
        return out

    def _grid_variable(self, obs):
        out = {}
        mem = getattr(self.env, "grid_memory", None)
        if mem is not None and getattr(mem, "slab", None) is not None:
            out.update(grid_memory=mem, grid_fts=None, grid_map=None, gridmap_pos_fts=None)   # device-resident
        else:
            out.update(grid_fts=[ob["grid_fts"].to(self.device) for ob in obs],
                       grid_map=[ob["grid_map"].to(self.device) for ob in obs],
                       gridmap_pos_fts=torch.stack([ob["gridmap_pos_fts"] for ob in obs], 0).to(self.device))
        return out

    def _nav_vp_variable(self, obs, gmaps, pano_embeds, cand_vpids, view_lens, nav_types):
        B = len(obs)
        vp_img = torch.cat([torch.zeros_like(pano_embeds[:, :1]), pano_embeds], 1)
        b_pos = []
        for i, gmap in enumerate(gmaps):
            cand = gmap.pos_features(obs[i]["viewpoint"], cand_vpids[i], obs[i]["heading"], obs[i]["elevation"])
            start = gmap.pos_features(obs[i]["viewpoint"], [gmap.start_vp], obs[i]["heading"], obs[i]["elevation"])
            pos = np.zeros((vp_img.size(1), 14), dtype=np.float32)
            pos[:, :7] = start
            pos[1:len(cand) + 1, 7:] = cand
            b_pos.append(torch.from_numpy(pos))
        nav_masks = torch.cat([torch.ones(B, 1, dtype=torch.bool, device=self.device), nav_types == 1], 1)
        return {
            "vp_img_embeds": vp_img, "vp_pos_fts": pad_tensors(b_pos).to(self.device),
            "vp_masks": gen_seq_masks(view_lens + 1), "vp_nav_masks": nav_masks,
            "vp_cand_vpids": [[None] + x for x in cand_vpids], "vp_obj_masks": None,
        }

    def _teacher_action(self, obs, vpids, ended, visited_masks=None):
        a = np.zeros(len(obs), dtype=np.int64)
        for i, ob in enumerate(obs):
            if ended[i]:
                a[i] = self.args.ignoreid
            elif ob["viewpoint"] == ob["gt_path"][-1]:
                a[i] = 0
            else:
                scan, cur = ob["scan"], ob["viewpoint"]
                best, best_d = self.args.ignoreid, float("inf")
                for j, vp in enumerate(vpids[i]):
                    if j > 0 and (visited_masks is None or not visited_masks[i][j]):
                        d = self.env.shortest_distances[scan][vp][ob["gt_path"][-1]] + \
                            self.env.shortest_distances[scan][cur][vp]
                        if d < best_d:
                            best_d, best = d, j
                a[i] = best
        self._teacher_host = a                      # (the loop needs the actions on the host too: no read-back)
        return self._upload(a)

    def make_equiv_action(self, a_t, gmaps, obs, traj):
        for i, ob in enumerate(obs):
            action = a_t[i]
            if action is not None:
                traj[i]["path"].append(gmaps[i].route(ob["viewpoint"], action))
                prev = traj[i]["path"][-2][-1] if len(traj[i]["path"][-1]) == 1 else traj[i]["path"][-1][-2]
                viewidx = self.scanvp_cands["%s_%s" % (ob["scan"], prev)][action]
                self.env.teleport(i, ob["scan"], action, (viewidx % 12) * math.radians(30),
                                  (viewidx // 12 - 1) * math.radians(30))

    def _update_scanvp_cands(self, obs):
        for ob in obs:
            d = self.scanvp_cands.setdefault("%s_%s" % (ob["scan"], ob["viewpoint"]), {})
            for cand in ob["candidate"]:
                d[cand["viewpointId"]] = int(cand["pointId"])

    # ---- the loop --------------------------------------------------------------------------------
    def _tick(self, name, t0):
        """bench.py's rollout leg: with self.timers (a dict) set, wall time per section incl. a device synchronize."""
        if self.timers is None:
            return 0.0
        import time
        if self.timers_sync:
            torch.cuda.synchronize()
        t = time.perf_counter()
        if t0:
            self.timers[name] = self.timers.get(name, 0.0) + (t - t0)
        return t

    timers = None
    defer_grads = bool(int(os.environ.get("GRIDMM_DEFER_GRADS", "1")))   # A/B switch (autograd.deferred_param_grads)
    timers_sync = True      # False: host-side time per section only (the device wait lands in the section that reads results)

    def rollout(self, train_ml=None, reset=True):
        """One rollout of the environment's next mini-batch (agent.py:268-451).  The body is a generator that yields once
        per navigation step, right after the step's model calls have been ENQUEUED and before their results are read:
        rollout() simply drives it to the end; interleaved_rollouts() alternates several of them so that one batch's host
        work runs under another batch's device work."""
        gen = self._rollout_gen(train_ml, reset)
        try:
            while True:
                next(gen)
        except StopIteration as e:
            return e.value

    def _rollout_gen(self, train_ml=None, reset=True):
        t0 = self._tick(None, 0.0)
        obs = self.env.reset() if reset else self.env._get_obs()
        t0 = self._tick("env (grid memory step + observation dicts)", t0)
        self._update_scanvp_cands(obs)
        B = len(obs)
        if self.fast_collate:      # the B maps as rows of one set of arrays: the collator reads them in whole-array passes
            tbatch = TopoMapBatch([ob["viewpoint"] for ob in obs])
            gmaps = tbatch.maps
        else:
            tbatch, gmaps = None, [TopoMap(ob["viewpoint"]) for ob in obs]
        self.collator.reset(B)
        if self._graphs is not None:
            tok = self._graphs[1].weights_token(self._graphs[1].model)
            for g in self._graphs:
                g.validate(tok)            # weights updated since the graphs were captured (training between evaluations)?
        if tbatch is not None:
            tbatch.observe_all(obs)
        else:
            for i, ob in enumerate(obs):
                gmaps[i].observe(ob)
        traj = [{"instr_id": ob["instr_id"], "path": [[ob["viewpoint"]]], "details": {}} for ob in obs]

        language_inputs = self._language_variable(obs)
        txt_embeds = self._model_call("language", language_inputs)
        t0 = self._tick("language", t0)

        ended = np.array([False] * B)
        just_ended = np.array([False] * B)
        ml_loss = 0.0
        # Teacher-forced training rollouts need nothing from the device to go on: the actions come from the ground-truth path.
        # The stop probabilities (read every step by the reference, agent.py:353-356, for the best-stop-node bookkeeping of
        # `traj`) stay on the device and are resolved after the loop -- or not at all inside train(), which discards `traj` --
        # so the host never waits for the stream and can run a whole iteration ahead of the device.
        lazy = train_ml is not None and self.feedback == "teacher" and self.trace is None and self.device.type == "cuda"
        pending = []

        for t in range(self.args.max_action_len):
            self.nav_steps = getattr(self, "nav_steps", 0) + 1
            for i, gmap in enumerate(gmaps):
                if not ended[i]:
                    if tbatch is not None:
                        tbatch.mark_step(i, obs[i]["viewpoint"], t + 1)
                    else:
                        gmap.step_id[obs[i]["viewpoint"]] = t + 1

            fast = self.fast_collate
            mem = getattr(self.env, "grid_memory", None)
            if self._graphs is not None and not torch.is_grad_enabled() and getattr(mem, "slab", None) is not None:
                # the half of 'navigation' that needs only the instruction and the grid memory starts now: the device works
                # through it while the host collates the panorama / graph inputs
                self._graphs[1].begin(txt_embeds, language_inputs["txt_masks"], mem)
                t0 = self._tick("navigation", t0)      # (profiled rollouts: the early front half counts as navigation)
            self.collator.keep_inputs = self.trace is not None
            pano_inputs = self.collator.panorama(obs) if fast else self._panorama_feature_variable(obs)
            t0 = self._tick("host: collate panorama inputs", t0)
            pano_embeds, pano_masks = self._model_call("panorama", pano_inputs)
            t0 = self._tick("panorama", t0)
            if fast:
                self.collator.update_embeddings(obs, gmaps, ended, pano_embeds, pano_masks, pano_inputs["cand_vpids"])
                nav_inputs = self.collator.navigation(obs, gmaps, pano_embeds, pano_inputs["cand_vpids"],
                                                      pano_inputs["view_lens"], pano_inputs["nav_types"])
                nav_inputs.update(self._grid_variable(obs))
            else:
                avg_pano = torch.sum(pano_embeds * pano_masks.unsqueeze(2), 1) / torch.sum(pano_masks, 1, keepdim=True)
                for i, gmap in enumerate(gmaps):
                    if not ended[i]:
                        gmap.add_embedding(obs[i]["viewpoint"], avg_pano[i], overwrite=True)
                        for j, cvp in enumerate(pano_inputs["cand_vpids"][i]):
                            if not gmap.visited(cvp):
                                gmap.add_embedding(cvp, pano_embeds[i, j])
                nav_inputs = self._nav_gmap_variable(obs, gmaps)
                nav_inputs.update(self._nav_vp_variable(obs, gmaps, pano_embeds, pano_inputs["cand_vpids"],
                                                        pano_inputs["view_lens"], pano_inputs["nav_types"]))
            nav_inputs.update({"txt_embeds": txt_embeds, "txt_masks": language_inputs["txt_masks"]})
            t0 = self._tick("host: TopoMap update + collate navigation inputs", t0)
            nav_outs = self._model_call("navigation", nav_inputs)
            t0 = self._tick("navigation", t0)

            if self.args.fusion == "local":
                nav_logits, nav_vpids = nav_outs["local_logits"], nav_inputs["vp_cand_vpids"]
            elif self.args.fusion == "global":
                nav_logits, nav_vpids = nav_outs["global_logits"], nav_inputs["gmap_vpids"]
            else:
                nav_logits, nav_vpids = nav_outs["fused_logits"], nav_inputs["gmap_vpids"]
            nav_probs = torch.softmax(nav_logits, 1)
            yield t                                   # (the device works on this step; another rollout may use the host now)
            if self.feedback == "argmax" and train_ml is None:
                # inference: the step's ONE device-to-host read -- stop probabilities and the arg-max actions together
                a_t = nav_logits.max(1)[1].detach()
                both = torch.stack([nav_probs[:, 0].detach().double(), a_t.double()]).cpu().numpy()
                stop_probs, a_t_pre = both[0].astype(np.float32), both[1].astype(np.int64)
            elif lazy:
                a_t_pre, stop_probs = None, None
                pending.append({"probs": nav_probs[:, 0].detach(), "vps": [ob["viewpoint"] for ob in obs],
                                "active": ~ended, "ended_now": []})
            else:
                a_t_pre = None
                stop_probs = nav_probs[:, 0].detach().cpu().numpy()   # one D2H per step (reference: B .item() calls)
            if stop_probs is not None:
                for i, gmap in enumerate(gmaps):
                    if not ended[i]:
                        gmap.stop_score[obs[i]["viewpoint"]] = {"stop": float(stop_probs[i])}

            nav_targets = None
            if train_ml is not None or self.feedback == "teacher":
                vm = None
                if self.args.fusion != "local":          # (the batched collator hands the mask over on the host as well)
                    vm = nav_inputs.get("gmap_visited_masks_host")
                    vm = nav_inputs["gmap_visited_masks"].cpu().numpy() if vm is None else vm
                nav_targets = self._teacher_action(obs, nav_vpids, ended, visited_masks=vm)
            if train_ml is not None:
                ml_loss = ml_loss + F.cross_entropy(nav_logits, nav_targets, ignore_index=self.args.ignoreid,
                                                    reduction="sum")

            if self.feedback == "teacher":
                a_t = nav_targets
            elif self.feedback == "argmax":
                if a_t_pre is None:
                    a_t = nav_logits.max(1)[1].detach()
            elif self.feedback == "sample":
                c = torch.distributions.Categorical(nav_probs)
                self.logs["entropy"].append(c.entropy().sum().item())
                a_t = c.sample().detach()
            elif self.feedback == "expl_sample":        # agent.py:385-395
                a_t = nav_probs.max(1)[1].detach()
                rand_explores = np.random.rand(B) > self.args.expl_max_ratio
                if self.args.fusion == "local":
                    cpu_nav_masks = nav_inputs["vp_nav_masks"].cpu().numpy()
                else:
                    cpu_nav_masks = (nav_inputs["gmap_masks"] & nav_inputs["gmap_visited_masks"].logical_not()).cpu().numpy()
                for i in range(B):
                    if rand_explores[i]:
                        a_t[i] = int(np.random.choice(np.arange(len(cpu_nav_masks[i]))[cpu_nav_masks[i]]))
            else:
                raise ValueError("Invalid feedback option: %s" % self.feedback)

            if self.feedback in ("teacher", "sample"):
                a_t_stop = [ob["viewpoint"] == ob["gt_path"][-1] for ob in obs]
                a_t_host = self._teacher_host if self.feedback == "teacher" else a_t.cpu().numpy()
            elif a_t_pre is not None:
                a_t_host, a_t_stop = a_t_pre, a_t_pre == 0
            else:
                a_t_host = a_t.cpu().numpy()
                a_t_stop = a_t_host == 0

            if self.trace is not None:
                self.trace.append({"t": t, "pano_inputs": pano_inputs, "nav_inputs": nav_inputs, "nav_outs": nav_outs, "a_t": a_t_host.copy(),
                                   "ended": ended.copy(), "nav_vpids": nav_vpids})

            cpu_a_t = []
            for i in range(B):
                if a_t_stop[i] or ended[i] or nav_inputs["no_vp_left"][i] or (t == self.args.max_action_len - 1):
                    cpu_a_t.append(None)
                    just_ended[i] = True
                else:
                    cpu_a_t.append(nav_vpids[i][a_t_host[i]])

            self.make_equiv_action(cpu_a_t, gmaps, obs, traj)
            for i in range(B):
                if (not ended[i]) and just_ended[i]:
                    if lazy:
                        pending[-1]["ended_now"].append(i)
                        continue
                    stop_node, stop_score = None, {"stop": -float("inf")}
                    for k, v in gmaps[i].stop_score.items():
                        if v["stop"] > stop_score["stop"]:
                            stop_score, stop_node = v, k
                    if stop_node is not None and obs[i]["viewpoint"] != stop_node:
                        traj[i]["path"].append(gmaps[i].route(obs[i]["viewpoint"], stop_node))

            t0 = self._tick("host: action selection + env step", t0)
            obs = self.env._get_obs()
            t0 = self._tick("env (grid memory step + observation dicts)", t0)
            self._update_scanvp_cands(obs)
            if tbatch is not None:
                tbatch.observe_all(obs, ~ended)
            else:
                for i, ob in enumerate(obs):
                    if not ended[i]:
                        gmaps[i].observe(ob)
            ended[:] = np.logical_or(ended, np.array([x is None for x in cpu_a_t]))
            if ended.all():
                break

        if pending and not getattr(self, "_defer_host_reads", False):
            self._resolve_stops(pending, gmaps, traj)      # ONE read-back for the whole rollout (train() skips even that)
        if train_ml is not None:
            ml_loss = ml_loss * train_ml / B
            self.loss = self.loss + ml_loss
            if torch.is_tensor(ml_loss) and getattr(self, "_defer_host_reads", False):
                self._pending_logs.append(ml_loss.detach())           # train() converts them once, after its last iteration
            else:
                self.logs["IL_loss"].append(float(ml_loss.detach()) if torch.is_tensor(ml_loss) else float(ml_loss))
        return traj

    @staticmethod
    def _resolve_stops(pending, gmaps, traj):
        """The stop-score bookkeeping of agent.py:353-356 / 425-436 for a rollout whose per-step stop probabilities were left
        on the device: replayed in step order from one read-back (an ended episode's map and trajectory do not change after
        its last step, so the deferred back-track to the best stop node appends the same route)."""
        probs = torch.stack([r["probs"] for r in pending]).float().cpu().numpy()
        for r, pr in zip(pending, probs):
            for i in np.nonzero(r["active"])[0]:
                gmaps[i].stop_score[r["vps"][i]] = {"stop": float(pr[i])}
            for i in r["ended_now"]:
                stop_node, stop_score = None, {"stop": -float("inf")}
                for k, v in gmaps[i].stop_score.items():
                    if v["stop"] > stop_score["stop"]:
                        stop_score, stop_node = v, k
                if stop_node is not None and r["vps"][i] != stop_node:
                    traj[i]["path"].append(gmaps[i].route(r["vps"][i], stop_node))

    @staticmethod
    def interleaved_rollouts(agents, streams=None):
        """Run one rollout() of every agent, interleaved at their per-step yield points: while agent A's 'navigation' runs on
        the device, agent B collates its inputs on the host (and vice versa).  The agents share the model but own their
        environment, grid memory, collator and graph caches; each runs on its own stream.  Results = [a.rollout() for a in
        agents] (same trajectories: the calls of one agent are never reordered)."""
        from . import ops
        streams = streams or [torch.cuda.Stream() for _ in agents] if torch.cuda.is_available() else [None] * len(agents)
        gens = [a._rollout_gen() for a in agents]
        out, live = [None] * len(agents), list(range(len(agents)))
        while live:
            for i in list(live):
                ctx = torch.cuda.stream(streams[i]) if streams[i] is not None else contextlib.nullcontext()
                with ctx:
                    try:
                        next(gens[i])
                    except StopIteration as e:
                        out[i] = e.value
                        live.remove(i)
        return out

    # ---- Seq2SeqAgent.test / .train (agent_base.py:150-211) ---------------------------------------
    def test(self, feedback="argmax", iters=None):
        """Evaluate once on each instruction of the environment (BaseAgent.test, agent_base.py:49-77)."""
        self.feedback = feedback
        self._set_mode(False)
        self.env.reset_epoch()
        self.results, looped = {}, False
        with torch.no_grad():
            while not looped and (iters is None or iters > 0):
                for traj in self.rollout():
                    if traj["instr_id"] in self.results:
                        looped = True
                    else:
                        self.results[traj["instr_id"]] = traj
                if iters is not None:
                    iters -= 1
        return [{"instr_id": k, "trajectory": v["path"]} for k, v in self.results.items()]

    def _set_mode(self, training):
        m = self.vln_bert
        if hasattr(m, "train"):
            m.train(training)
        mem = getattr(self.env, "grid_memory", None)
        if mem is not None and hasattr(mem, "keep_for_backward"):
            mem.keep_for_backward = bool(training)

    # ---- checkpoints in the reference's agent format (agent_base.py:213-259) ------------------------------------
    def _ckpt_parts(self):
        if not hasattr(self, "critic"):
            from .model import Critic
            self.critic = Critic(self.args).to(self.device)
        if not hasattr(self, "vln_bert_optimizer"):
            self.make_optimizer()
        if not hasattr(self, "critic_optimizer"):    # agent_base.py:122-139: the same optimizer class as vln_bert's
            opt = {"rms": torch.optim.RMSprop, "adam": torch.optim.Adam, "adamW": torch.optim.AdamW,
                   "sgd": torch.optim.SGD}[self.args.optim]
            self.critic_optimizer = opt(self.critic.parameters(), lr=self.args.lr)
        return (("vln_bert", self.vln_bert, self.vln_bert_optimizer), ("critic", self.critic, self.critic_optimizer))

    def save(self, epoch, path):
        """{'vln_bert': {epoch, state_dict, optimizer}, 'critic': {...}} -- what the reference's Seq2SeqAgent.save writes."""
        import os
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save({name: {"epoch": epoch + 1, "state_dict": m.state_dict(), "optimizer": o.state_dict()}
                    for name, m, o in self._ckpt_parts()}, path)

    def load(self, path):
        """Loads parameters (and, with args.resume_optimizer, the optimizer states) of a checkpoint in that format, written
        by this agent or by the reference -- a DistributedDataParallel 'module.' prefix on either side is reconciled and
        keys the model does not have are skipped, as agent_base.py:230-259 does.  Returns the epoch to resume from."""
        states = torch.load(path, map_location="cpu")
        for name, model, opt in self._ckpt_parts():
            have = model.state_dict()
            given = states[name]["state_dict"]
            if set(have) != set(given):
                mine_ddp = next(iter(have)).startswith("module.")
                theirs_ddp = next(iter(given)).startswith("module.")
                if theirs_ddp and not mine_ddp:
                    given = {k[len("module."):] if k.startswith("module.") else k: v for k, v in given.items()}
                elif mine_ddp and not theirs_ddp:
                    given = {"module." + k: v for k, v in given.items()}
                given = {k: v for k, v in given.items() if k in have}
            have.update(given)
            model.load_state_dict(have)
            if getattr(self.args, "resume_optimizer", False):
                defaults = [dict(g) for g in opt.param_groups]
                opt.load_state_dict(states[name]["optimizer"])
                from .optim import AdamW
                if isinstance(opt, AdamW):    # a torch.optim.AdamW state (reference checkpoint): restore what it does not carry
                    for g, d in zip(opt.param_groups, defaults):
                        for k in ("correct_bias", "decay_first"):
                            g.setdefault(k, d[k])
                    for st in opt.state.values():
                        if torch.is_tensor(st.get("step")):
                            st["step"] = int(st["step"].item())
        return states["vln_bert"]["epoch"] - 1

    def make_optimizer(self):
        """agent_base.py:122-139: one optimizer over all vln_bert parameters at args.lr."""
        if self.args.optim == "adamW":        # scripts/run_r2r.sh; torch.optim.AdamW semantics on the fused HIP step
            from .optim import AdamW
            self.vln_bert_optimizer = AdamW(list(self.vln_bert.parameters()), lr=self.args.lr, betas=(0.9, 0.999),
                                            eps=1e-8, weight_decay=0.01, decay_first=True)
        else:
            opt = {"rms": torch.optim.RMSprop, "adam": torch.optim.Adam, "sgd": torch.optim.SGD}[self.args.optim]
            self.vln_bert_optimizer = opt(self.vln_bert.parameters(), lr=self.args.lr)
        from .dist import GradientReducer
        self.grad_reducer = GradientReducer(self.vln_bert.parameters())
        return self.vln_bert_optimizer

    def train(self, n_iters, feedback="teacher"):
        """agent_base.py:164-211: per iteration zero_grad -> rollout(s) accumulate self.loss -> backward ->
        [all-reduce(mean) of the gradients across ranks, the DDP step] -> clip_grad_norm 40 -> optimizer step."""
        if not hasattr(self, "vln_bert_optimizer"):
            self.make_optimizer()
        self.feedback = feedback
        self._set_mode(True)
        self.losses = []
        from . import autograd as ag
        # no device read-back inside the loop: the losses / IL logs of all iterations are fetched once at the end, so the host
        # issues iteration i + 1's forward while the device is still in iteration i's backward (GRIDMM_TRAIN_SYNC=1: per-iteration
        # read-back as before, A/B)
        self._defer_host_reads = os.environ.get("GRIDMM_TRAIN_SYNC", "0") in ("", "0")
        self._pending_logs, dev_losses = [], []
        try:
            self._train_iterations(n_iters, ag, dev_losses)
        finally:
            self._defer_host_reads = False
            if self._pending_logs:
                self.logs["IL_loss"].extend(float(v) for v in torch.stack([x.float() for x in self._pending_logs]).cpu())
            self._pending_logs = []
        if dev_losses:
            self.losses = [float(v) for v in torch.stack([x.float() for x in dev_losses]).cpu()]
        return self.losses

    def _train_iterations(self, n_iters, ag, dev_losses):
        for _ in range(n_iters):
            self.vln_bert_optimizer.zero_grad()
            self.loss = 0
            if self.args.train_alg == "imitation":
                self.feedback = "teacher"
                self.rollout(train_ml=1.0)
            elif self.args.train_alg == "dagger":
                if self.args.ml_weight != 0:
                    self.feedback = "teacher"
                    self.rollout(train_ml=self.args.ml_weight)
                self.feedback = "expl_sample" if self.args.expl_sample else "sample"
                self.rollout(train_ml=1)
            else:
                raise NotImplementedError("train_alg %r (the A2C branch, train_rl=True, is not used by the released "
                                          "GridMM scripts)" % self.args.train_alg)
            # one backward through every step of the rollout(s): each parameter gets a gradient per step -- kept aside by the
            # kernels' autograd Functions and summed by ONE launch instead of one add per parameter and step
            with ag.deferred_param_grads() if self.defer_grads else contextlib.nullcontext():
                self.loss.backward()
            ag.flush_param_grads()
            self.grad_reducer.reduce()
            if self.args.optim == "adamW":
                self.vln_bert_optimizer.step(max_grad_norm=40.0)          # clip_grad_norm_(40) fused into the step
            else:
                torch.nn.utils.clip_grad_norm_(self.vln_bert.parameters(), 40.0)
                self.vln_bert_optimizer.step()
            if self._defer_host_reads and torch.is_tensor(self.loss):
                dev_losses.append(self.loss.detach())
            else:
                self.losses.append(float(self.loss.detach()) if torch.is_tensor(self.loss) else float(self.loss))
