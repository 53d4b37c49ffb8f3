"""Batched host-side collation for GMapNavAgent.rollout: the same `pano_inputs` / `nav_inputs` dictionaries as
agent.py's per-episode restatement of the reference (map_nav_src/r2r/agent.py:51-205), built with one vectorised pass
over the batch instead of per-node tensor operations.

What the reference does per step and episode: stacks 36 view features one by one, keeps one tensor per graph node for
the running mean of its embeddings (models/graph_utils.py:125-137) and stacks them again every step, and calls
get_pos_fts three times (graph nodes, candidates, start node).  At B = 32 that is ~2 000 tiny tensor operations per
step -- 34 ms of host time around a 4 ms model call (bench.py `rollout`, round-3 first measurement).  Here:

  * node embeddings live in ONE device tensor per rollout (B, node slots, H) of running sums + host-side counts;
    a step applies all overwrites with one index_put and all accumulations with another, and reads
    `gmap_img_embeds` back with one gather (count 1 -> multiplied by 1.0: the stored row, exactly);
  * position features of the graph nodes, the candidates and the start node of all episodes go through one
    `graph_utils.batched_pos_features` call (same arithmetic and precisions as TopoMap.pos_features);
  * panorama inputs are assembled into padded numpy arrays with one fancy-index per episode and uploaded once.

agent.GMapNavAgent uses it by default (`fast_collate`); tests/test_agent_loop.py checks it against the per-episode
restatement on the same observations, key by key.
"""
import numpy as np
import torch

from . import _lib
from .graph_utils import UNREACHABLE, batched_pos_features


class _Stager:
    """Host arrays of one step -> ONE asynchronous host-to-device copy.  put() copies an array into the current slot of a
    small ring of PINNED buffers and returns the matching view of the slot's device mirror; flush() ships the used bytes.
    A `torch.from_numpy(a).to(device)` per array (the form this replaces: ~12 per step) is a blocking copy each, i.e. the
    host waits for everything already queued on the stream -- the early-launched front half of 'navigation' and the
    'panorama' graph -- before it can go on collating.  Views are valid until the ring comes round (3 flushes later; their
    consumers were enqueued on the same stream long before).  On a CPU device put() degrades to a plain copy."""

    def __init__(self, device, nbytes=1 << 20, ring=3):
        self.device, self.cuda = torch.device(device), torch.device(device).type == "cuda"
        self.ring, self.pos, self.used = [], 0, 0
        # owned = True: the views handed out by put() are slices of a FRESH device allocation per flush cycle (they own it and
        # are never overwritten) instead of the slot's recycled mirror -- for rollouts under autograd, whose backward reads the
        # inputs of every step after the ring has come round many times.  Same single asynchronous copy per flush: the host
        # never waits for the stream, so a training loop can run ahead of the device (agent.train).
        self.owned, self._cycle_owned, self._own = False, False, None
        if self.cuda:
            for _ in range(ring):
                self.ring.append([torch.empty(nbytes, dtype=torch.uint8).pin_memory(),
                                  torch.empty(nbytes, dtype=torch.uint8, device=self.device), None])

    def _slot(self):
        return self.ring[self.pos]

    MAX_RING = 64
    GROW = 8
    _spare = ()

    def _begin_cycle(self):
        host, dev, ev = self._slot()
        if self.owned and ev is not None and not ev.query() and len(self.ring) < self.MAX_RING:
            # the copy issued from this slot has not run yet (a training loop whose host is a whole iteration ahead of the
            # device): a new slot here instead of a wait.  Inference rollouts read the action back every step, so their ring
            # of three never waits long and never grows (a pinned allocation costs milliseconds)
            n = host.numel()
            if not self._spare or self._spare[-1].numel() != n:
                from .pinned import pinned_slots
                self._spare = pinned_slots(n, min(self.GROW, self.MAX_RING - len(self.ring)))   # one pinned allocation for several slots
            # (owned cycles never use the slot's device mirror: an empty placeholder until a recycled cycle needs it)
            self.ring.insert(self.pos, [self._spare.pop(), None, None])
            host, dev, ev = self._slot()
        if ev is not None:
            ev.synchronize()                          # the copy issued from this slot `ring` flushes ago has completed
        self._cycle_owned = self.owned
        self._own = torch.empty(host.numel(), dtype=torch.uint8, device=self.device) if self.owned else None

    def put(self, a):
        a = np.ascontiguousarray(a)
        if not self.cuda:
            return torch.from_numpy(a.copy())
        t = torch.from_numpy(a)
        n = a.nbytes
        if self.used and self._cycle_owned != self.owned:
            self.flush()                              # (a cycle is either recycled or owned)
        o = (self.used + 15) // 16 * 16
        host = self._slot()[0]
        if o + n > host.numel():
            self.flush()                              # (a step that outgrows the slot: ship what is there, start the next slot)
            if n > self._slot()[0].numel():
                self.ring[self.pos][0] = torch.empty(2 * n, dtype=torch.uint8).pin_memory()
                self.ring[self.pos][1] = torch.empty(2 * n, dtype=torch.uint8, device=self.device)
            o = 0
        if self.used == 0:
            self._begin_cycle()
        host, dev, _ = self._slot()
        if self._cycle_owned:
            dev = self._own
        elif dev is None:                             # a slot added by an owned cycle: its mirror is made on first recycled use
            dev = self.ring[self.pos][1] = torch.empty(host.numel(), dtype=torch.uint8, device=self.device)
        if n:
            host[o:o + n].view(t.dtype).view(t.shape).copy_(t)
        self.used = o + n
        return dev[o:o + n].view(t.dtype).view(t.shape)

    def flush(self):
        if not self.cuda or self.used == 0:
            return
        host, dev, _ = self._slot()
        if self._cycle_owned:
            dev = self._own
        dev[:self.used].copy_(host[:self.used], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.ring[self.pos][2] = ev
        self.pos, self.used, self._own = (self.pos + 1) % len(self.ring), 0, None


class NavCollator:
    def __init__(self, args, device, node_slots=48, node_buckets=None, view_buckets=None):
        """node_buckets / view_buckets: ascending sizes to which the graph-node axis G / the view axis V are padded
        (masked rows: the model's outputs on the real rows do not change) so that the per-step shapes come from a small
        set -- what graph.NavigationGraphs keys its captured graphs by.  None: the reference's shapes (max over the batch)."""
        self.args, self.device = args, torch.device(device)
        self.node_slots = node_slots
        self.node_buckets, self.view_buckets = node_buckets, view_buckets
        self.pool = None              # (B, slots, H) running sums; slot 0 stays zero (stop token / padding)
        self.cnt = None               # (B, slots) host counts
        self.stage = _Stager(self.device)
        # panorama blocks are pure functions of (scan, viewpoint, view index) -- the reference keeps them in its
        # buffered_state_dict / ImageFeaturesDB (map_nav_src/r2r/env.py:529-575, utils/data.py) -- so the COLLATED block of
        # an observation (candidate views first, then the views that face no candidate; image and location features,
        # types, length, candidate ids) is kept in device tables, one slot per key: a step uploads only the blocks it has
        # not seen and gathers the batch on the device.  pano_cache=False restores the per-step assembly.
        # The tables are BOUNDED: at most `pano_cache_slots` blocks (default 4096: ~0.6 GB at 48 views x 768 fp32), least
        # recently used blocks are overwritten beyond that -- a full R2R sweep touches tens of thousands of distinct
        # (viewpoint, view index) states and an unbounded table would grow into several GB.  clear_panorama_cache() drops them.
        self.pano_cache = True
        self.pano_cache_slots = 4096
        self._pano = None

    def reset(self, batch_size):
        self.pool = None
        self.cnt = np.zeros((batch_size, self.node_slots), dtype=np.int32)

    def clear_panorama_cache(self):
        """Drop the device tables of collated panorama blocks (e.g. between splits); they are rebuilt on demand."""
        self._pano = None

    @staticmethod
    def _bucket(n, buckets):
        for b in buckets or ():
            if n <= b:
                return int(b)
        return n

    keep_inputs = False     # True: every uploaded array owns its device memory (training rollouts keep the inputs of all
    #                         steps for backward; traced rollouts keep them for inspection) instead of a view of the ring

    def _dev(self, a):
        # under autograd (or when the caller keeps the inputs) the views own their device memory; either way the bytes travel
        # through the pinned ring in ONE asynchronous copy per flush -- never a pageable (stream-synchronising) upload
        self.stage.owned = bool(self.keep_inputs or torch.is_grad_enabled())
        return self.stage.put(a)

    # ---- agent.py:51-94 ---------------------------------------------------------------------------
    @staticmethod
    def _pano_block(ob, fs):
        """One observation's collated panorama: (img (n, fs), loc (n, A + 3), types (n,), candidate ids)."""
        feat = np.asarray(ob["feature"])
        A = feat.shape[1] - fs
        pids = [int(cc["pointId"]) for cc in ob["candidate"]]
        nc = len(pids)
        rest = np.ones(36, dtype=bool)
        n = nc + 36 - len(set(pids))
        img = np.zeros((n, fs), dtype=np.float32)
        loc = np.zeros((n, A + 3), dtype=np.float32)
        types = np.zeros(n, dtype=np.int64)
        if nc:                                          # candidates' views first, then the views that face no candidate
            rest[pids] = False
            cf = np.stack([cc["feature"] for cc in ob["candidate"]])
            img[:nc], loc[:nc, :A] = cf[:, :fs], cf[:, fs:]
            types[:nc] = 1
        img[nc:], loc[nc:, :A] = feat[rest, :fs], feat[rest, fs:]
        loc[:, A:] = 1.0                                # the constant "box" columns of valid rows
        return img, loc, types, [cc["viewpointId"] for cc in ob["candidate"]]

    def panorama(self, obs):
        fs, B = self.args.image_feat_size, len(obs)
        if self.pano_cache and all("viewIndex" in ob and "scan" in ob for ob in obs):
            return self._panorama_cached(obs)
        blocks = [self._pano_block(ob, fs) for ob in obs]
        lens = np.fromiter((len(b[2]) for b in blocks), dtype=np.int64, count=B)
        self._view_lens = lens
        V = self._bucket(int(lens.max()), self.view_buckets)
        img = np.zeros((B, V, fs), dtype=np.float32)
        loc = np.zeros((B, V, blocks[0][1].shape[1]), dtype=np.float32)
        types = np.zeros((B, V), dtype=np.int64)
        for i, (bi, bl, bt, _) in enumerate(blocks):
            n = len(bt)
            img[i, :n], loc[i, :n], types[i, :n] = bi, bl, bt
        out = {
            "view_img_fts": self._dev(img), "loc_fts": self._dev(loc),
            "nav_types": self._dev(types), "view_lens": self._dev(lens),
            "cand_vpids": [b[3] for b in blocks], "obj_img_fts": None, "obj_lens": None,
        }
        self.stage.flush()
        return out

    def _panorama_cached(self, obs):
        fs, B = self.args.image_feat_size, len(obs)
        P = self._pano
        if P is None:
            A3 = np.asarray(obs[0]["feature"]).shape[1] - fs + 3
            P = self._pano = {"slot": {}, "key_of": [], "cands": [], "lens": [], "tick": np.zeros(0, dtype=np.int64), "clock": 0,
                              "cap": 0, "vmax": 48, "A3": A3, "img": None, "loc": None, "types": None}
        keys = [(ob["scan"], ob["viewpoint"], int(ob["viewIndex"])) for ob in obs]
        P["clock"] += 1
        miss = {}
        for k, ob in zip(keys, obs):
            if k not in P["slot"] and k not in miss:
                miss[k] = self._pano_block(ob, fs)
        if miss:
            limit = max(int(self.pano_cache_slots), 2 * B)             # (a batch must fit beside the blocks it replaces)
            need_v = max(len(b[2]) for b in miss.values())
            used = len(P["key_of"])
            n_new = min(used + len(miss), limit)
            if n_new > P["cap"] or need_v > P["vmax"]:                 # grow the device tables (doubling, up to the limit)
                cap = max(256, P["cap"])
                while cap < n_new:
                    cap *= 2
                cap = min(cap, limit)
                vmax = P["vmax"]
                while vmax < need_v:
                    vmax += 16
                new = {"img": torch.zeros(cap, vmax, fs, dtype=torch.float32, device=self.device),
                       "loc": torch.zeros(cap, vmax, P["A3"], dtype=torch.float32, device=self.device),
                       "types": torch.zeros(cap, vmax, dtype=torch.int64, device=self.device)}
                for name, t in new.items():
                    if P[name] is not None:
                        t[:P["cap"], :P["vmax"]] = P[name]
                    P[name] = t                                         # (the old table is released here)
                tick = np.zeros(cap, dtype=np.int64)
                tick[:len(P["tick"])] = P["tick"]
                P["cap"], P["vmax"], P["tick"] = cap, vmax, tick
            # slots of the new blocks: fresh ones while the table has room, then the least recently used blocks that this
            # batch does not read (their keys leave the index)
            m, vmax = len(miss), P["vmax"]
            fresh = list(range(used, min(used + m, P["cap"])))
            if len(fresh) < m:
                busy = {P["slot"][k] for k in keys if k in P["slot"]}
                order = np.argsort(P["tick"][:used], kind="stable")
                victims = [int(sl) for sl in order if int(sl) not in busy][:m - len(fresh)]
                for sl in victims:
                    del P["slot"][P["key_of"][sl]]
                fresh += victims
            img = np.zeros((m, vmax, fs), dtype=np.float32)
            loc = np.zeros((m, vmax, P["A3"]), dtype=np.float32)
            types = np.zeros((m, vmax), dtype=np.int64)
            for j, (k, (bi, bl, bt, cands)) in enumerate(miss.items()):
                n, sl = len(bt), fresh[j]
                img[j, :n], loc[j, :n], types[j, :n] = bi, bl, bt
                P["slot"][k] = sl
                if sl == len(P["key_of"]):
                    P["key_of"].append(k); P["cands"].append(cands); P["lens"].append(n)
                else:
                    P["key_of"][sl], P["cands"][sl], P["lens"][sl] = k, cands, n
            contiguous = fresh == list(range(fresh[0], fresh[0] + m))
            # first sight of a viewpoint: its collated block goes up through the pinned ring too (asynchronous); owned device
            # memory, because the four uploads below span more flush cycles than the ring has recycled mirrors
            self.stage.owned = True
            ix = None if contiguous else self.stage.put(np.asarray(fresh, dtype=np.int64))
            srcs = {}
            for name, a in (("img", img), ("loc", loc), ("types", types)):
                self.stage.flush()                                     # (blocks of up to B panoramas: one slot each)
                srcs[name] = self.stage.put(a)
            self.stage.flush()
            for name in ("img", "loc", "types"):
                src = srcs[name]
                if contiguous:
                    P[name][fresh[0]:fresh[0] + m].copy_(src)
                else:
                    P[name].index_copy_(0, ix, src)
        slots = np.fromiter((P["slot"][k] for k in keys), dtype=np.int64, count=B)
        P["tick"][slots] = P["clock"]
        lens = np.fromiter((P["lens"][sl] for sl in slots), dtype=np.int64, count=B)
        self._view_lens = lens
        V = self._bucket(int(lens.max()), self.view_buckets)
        sd, ld = self._dev(slots), self._dev(lens)
        ld._gridmm_host_max = int(lens.max())           # the mask width, known here: the model need not read it back
        self.stage.flush()
        if V > P["vmax"]:                                               # (a view bucket wider than the table rows)
            pad = lambda t: torch.nn.functional.pad(t, (0, 0, 0, V - t.shape[1]) if t.dim() == 3 else (0, V - t.shape[1]))   # noqa: E731
        else:
            pad = lambda t: t                                           # noqa: E731
        return {
            "view_img_fts": pad(torch.index_select(P["img"][:, :V], 0, sd)),
            "loc_fts": pad(torch.index_select(P["loc"][:, :V], 0, sd)),
            "nav_types": pad(torch.index_select(P["types"][:, :V], 0, sd)), "view_lens": ld,
            "cand_vpids": [P["cands"][sl] for sl in slots], "obj_img_fts": None, "obj_lens": None,
        }

    # ---- agent.py:300-311 (graph node embeddings) ---------------------------------------------------
    def _grow(self, need):
        slots = self.cnt.shape[1]
        while slots <= need:
            slots *= 2
        cnt = np.zeros((self.cnt.shape[0], slots), dtype=np.int32)
        cnt[:, :self.cnt.shape[1]] = self.cnt
        self.cnt = cnt
        if self.pool is not None:
            pool = self.pool.new_zeros(self.pool.shape[0], slots, self.pool.shape[2])
            pool[:, :self.pool.shape[1]] = self.pool
            self.pool = pool

    def _apply(self, src, sets, adds):
        """sets / adds: rows (episode, node slot, source row): pool[e, slot] = / += src[e, row].  (episode, slot) pairs are
        unique within one call (the callers split a step whose node repeats)."""
        if torch.is_grad_enabled() and (src.requires_grad or self.pool.requires_grad):
            for ops, accumulate in ((sets, False), (adds, True)):       # differentiable rollouts: out of place
                if len(ops) == 0:
                    continue
                ix = self._dev(np.asarray(ops, dtype=np.int64).T.copy())
                self.stage.flush()
                self.pool = self.pool.index_put((ix[0], ix[1]), src[ix[0], ix[2]], accumulate=accumulate)
            return
        # inference: flat row indices, one upload, index_select / index_copy_ / index_add_ (one kernel each; index_put_
        # sorts its indices first: ~0.5 ms of host time per call on this stack)
        S, V1, H = self.pool.shape[1], src.shape[1], src.shape[2]
        parts = []
        for ops in (sets, adds):
            o = np.asarray(ops, dtype=np.int64).reshape(-1, 3)
            parts += [o[:, 0] * S + o[:, 1], o[:, 0] * V1 + o[:, 2]]
        ns, na = len(parts[0]), len(parts[2])
        if ns + na == 0:
            return
        ix = self._dev(np.concatenate(parts))
        self.stage.flush()
        flat, rows = self.pool.view(-1, H), src.reshape(-1, H)
        if ns:
            flat.index_copy_(0, ix[:ns], rows.index_select(0, ix[ns:2 * ns]))
        if na:
            flat.index_add_(0, ix[2 * ns:2 * ns + na], rows.index_select(0, ix[2 * ns + na:]))

    def update_embeddings(self, obs, gmaps, ended, pano_embeds, pano_masks, cand_vpids):
        """Current node := mean of its panorama (overwrite); every unvisited candidate += its view embedding."""
        avg = torch.sum(pano_embeds * pano_masks.unsqueeze(2), 1) / torch.sum(pano_masks, 1, keepdim=True)
        src = torch.cat([avg.unsqueeze(1), pano_embeds], 1)               # row 0: the panorama mean
        if self.pool is None:
            self.pool = src.new_zeros(src.shape[0], self.cnt.shape[1], src.shape[2])
        tb = getattr(gmaps[0], "_batch", None)
        if tb is not None and all(getattr(g, "_batch", None) is tb for g in gmaps):
            cid, cm = self._cand_ids(gmaps, cand_vpids)
            srt = np.sort(np.where(cm, cid, -1 - np.arange(cid.shape[1])[None]), axis=1)
            if not (srt[:, 1:] == srt[:, :-1]).any():            # (a node twice among one panorama's candidates: loop form)
                return self._update_embeddings_batched(tb, obs, gmaps, ended, src, cid, cm)
        sets, adds, touched = [], [], set()
        for i, g in enumerate(gmaps):
            if ended[i]:
                continue
            if g.n + 1 >= self.cnt.shape[1]:
                self._grow(g.n + 1)
            items = [(g.index(obs[i]["viewpoint"]) + 1, 0, True)]
            items += [(g.index(c) + 1, j + 1, False) for j, c in enumerate(cand_vpids[i]) if not g.visited(c)]
            for slot, row, overwrite in items:
                if (i, slot) in touched:                                   # same node twice in one step: keep the order
                    self._apply(src, sets, adds)
                    sets, adds, touched = [], [], set()
                touched.add((i, slot))
                if overwrite or self.cnt[i, slot] == 0:
                    sets.append((i, slot, row))
                    self.cnt[i, slot] = 1
                else:
                    adds.append((i, slot, row))
                    self.cnt[i, slot] += 1
        self._apply(src, sets, adds)

    def _cand_ids(self, gmaps, cand_vpids):
        memo = getattr(self, "_cid_memo", None)
        if memo is not None and memo[0] is cand_vpids and memo[1] is gmaps[0]:      # update_embeddings and navigation of one step
            return memo[2]
        out = self._cand_ids_build(gmaps, cand_vpids)
        self._cid_memo = (cand_vpids, gmaps[0], out)
        return out

    def _cand_ids_build(self, gmaps, cand_vpids):
        B = len(gmaps)
        nc = np.fromiter((len(c) for c in cand_vpids), dtype=np.int64, count=B)
        cid = np.zeros((B, max(int(nc.max()) if B else 0, 1)), dtype=np.int64)
        for i, (g, cs) in enumerate(zip(gmaps, cand_vpids)):
            if cs:
                cid[i, :len(cs)] = [g._id[c] for c in cs]
        return cid, np.arange(cid.shape[1])[None] < nc[:, None]

    def _update_embeddings_batched(self, tb, obs, gmaps, ended, src, cid, cm):
        B = len(obs)
        bi = np.arange(B)
        if self.cnt.shape[1] <= tb.cap:
            self._grow(tb.cap)
        live = ~np.asarray(ended, dtype=bool)
        cur = np.fromiter((g.index(ob["viewpoint"]) for g, ob in zip(gmaps, obs)), dtype=np.int64, count=B)
        lb = bi[live]
        sets = [np.stack([lb, cur[live] + 1, np.zeros(len(lb), dtype=np.int64)], 1)]     # current node := panorama mean
        self.cnt[lb, cur[live] + 1] = 1
        take = cm & live[:, None] & ~tb.seen[bi[:, None], cid]                            # unvisited candidates
        i, j = np.nonzero(take)
        slot = cid[i, j] + 1
        first = self.cnt[i, slot] == 0
        sets.append(np.stack([i[first], slot[first], j[first] + 1], 1))
        adds = np.stack([i[~first], slot[~first], j[~first] + 1], 1)
        self.cnt[i[first], slot[first]] = 1
        np.add.at(self.cnt, (i[~first], slot[~first]), 1)
        self._apply(src, np.concatenate(sets), adds)

    # ---- agent.py:96-205 ----------------------------------------------------------------------------
    def navigation(self, obs, gmaps, pano_embeds, cand_vpids, view_lens, nav_types, grid_memory=None):
        batch = getattr(gmaps[0], "_batch", None)
        if batch is not None and all(getattr(g, "_batch", None) is batch for g in gmaps):
            return self._navigation_batched(batch, obs, gmaps, pano_embeds, cand_vpids, view_lens, nav_types)
        return self._navigation_loop(obs, gmaps, pano_embeds, cand_vpids, view_lens, nav_types)

    native = True       # False: the NumPy form below (kept as the readable restatement; tests compare the two)

    def _navigation_batched(self, tb, obs, gmaps, pano_embeds, cand_vpids, view_lens, nav_types):
        """The step's nav_inputs from the (B, cap, ...) arrays of a graph_utils.TopoMapBatch in two native calls
        (gridmm_collate_nav_plan / _fill, csrc/hostutil.hip: node order, position features of graph nodes / candidates /
        start node, pair distances, step ids, masks, embedding slots, fusion maps); what stays in Python: the viewpoint-name
        lists the caller needs.  ~100 small NumPy calls per step before (`_navigation_batched_numpy`)."""
        if not self.native:
            return self._navigation_batched_numpy(tb, obs, gmaps, pano_embeds, cand_vpids, view_lens, nav_types)
        lib = _lib.load()
        a, B, cap = self.args, len(obs), tb.cap
        cur = np.fromiter((g.index(ob["viewpoint"]) for g, ob in zip(gmaps, obs)), dtype=np.int64, count=B)
        start = np.fromiter((g.index(g.start_vp) for g in gmaps), dtype=np.int64, count=B)
        n = np.ascontiguousarray(tb.n, dtype=np.int64)
        seen = np.ascontiguousarray(tb.seen).view(np.uint8)
        order = np.empty((B, cap), dtype=np.int64)
        m, n_vis, n_unv = (np.empty(B, dtype=np.int64) for _ in range(3))
        seen_eff = np.empty((B, cap), dtype=np.uint8)
        P = lambda x: x.ctypes.data       # noqa: E731
        M = lib.gridmm_collate_nav_plan(P(seen), P(n), P(cur), B, cap, int(bool(a.enc_full_graph)), int(bool(a.act_visited_nodes)),
                                        P(order), P(m), P(n_vis), P(n_unv), P(seen_eff))
        if M < 0:
            _lib.check(M, "gridmm_collate_nav_plan")
        G = self._bucket(1 + M, self.node_buckets)
        cid, cm = self._cand_ids(gmaps, cand_vpids)
        nc = cm.sum(1).astype(np.int64)
        Cw = cid.shape[1]
        V1 = pano_embeds.shape[1] + 1
        afs = int(getattr(a, "angle_feat_size", 4))
        F = afs + 3
        if self.cnt.shape[1] <= cap:
            self._grow(cap)
        bh = np.array([float(ob["heading"]) for ob in obs], dtype=np.float64)
        be = np.array([float(ob["elevation"]) for ob in obs], dtype=np.float64)
        gpos = np.empty((B, G, F), dtype=np.float32)
        vpos = np.empty((B, V1, 2 * F), dtype=np.float32)
        pair = np.empty((B, G, G), dtype=np.float32)
        steps, slot = np.empty((B, G), dtype=np.int64), np.empty((B, G), dtype=np.int64)
        visited, gmask = np.empty((B, G), dtype=np.uint8), np.empty((B, G), dtype=np.uint8)
        inv = np.empty((B, G), dtype=np.float32)
        cand_of_node = np.empty((B, G), dtype=np.int32)
        cand_visited = np.empty((B, V1), dtype=np.uint8)
        cnt = np.ascontiguousarray(self.cnt, dtype=np.int32)
        rc = lib.gridmm_collate_nav_fill(P(tb.pos), P(tb.dist), P(tb.via), P(tb.step), P(order), P(m), P(n_vis), P(seen_eff),
                                         P(cur), P(start), P(cid), P(nc), P(bh), P(be), P(cnt), B, cap, Cw, cnt.shape[1], G, V1,
                                         afs, int(bool(a.enc_full_graph)), P(gpos), P(vpos), P(pair), P(steps), P(visited),
                                         P(slot), P(inv), P(gmask), P(cand_of_node), P(cand_visited))
        if rc > 0:
            i, j = divmod(rc - 1, G)
            raise KeyError("graph node without an embedding: %s" % gmaps[i].names[int(order[i, j - 1])])
        _lib.check(rc, "gridmm_collate_nav_fill")
        vpids = [[None] + [g.names[k] for k in order[i, :m[i]]] for i, g in enumerate(gmaps)]
        slot_d, inv_d = self._dev(slot), self._dev(inv)
        out = {
            "gmap_vpids": vpids, "gmap_img_embeds": None, "gmap_step_ids": self._dev(steps),
            "gmap_pos_fts": self._dev(gpos), "gmap_visited_masks": self._dev(visited.view(np.bool_)),
            "gmap_pair_dists": self._dev(pair), "gmap_masks": self._dev(gmask.view(np.bool_)),
            "no_vp_left": [bool(v == 0) for v in n_unv],
            "fusion_maps": (self._dev(cand_of_node), self._dev(cand_visited)),
            "gmap_visited_masks_host": visited.view(np.bool_),      # (the teacher's action needs it on the host)
        }
        out = self._vp_part(out, pano_embeds, cand_vpids, view_lens, nav_types, vpos)
        self.stage.flush()                         # every host array of the step: one asynchronous upload
        rows = torch.arange(B, device=self.device).unsqueeze(1)
        out["gmap_img_embeds"] = self.pool[rows, slot_d] * inv_d.unsqueeze(2)
        return out

    def _navigation_batched_numpy(self, tb, obs, gmaps, pano_embeds, cand_vpids, view_lens, nav_types):
        """The same dictionary from the (B, cap, ...) arrays of a graph_utils.TopoMapBatch: node order, geometry, pair
        distances, step ids, embedding slots and the fusion maps of all episodes in one pass of whole-array operations
        (the per-episode Python that is left: name lists for the caller, routes with more than one leg)."""
        a, B, cap = self.args, len(obs), tb.cap
        bi = np.arange(B)
        n = tb.n
        cur = np.fromiter((g.index(ob["viewpoint"]) for g, ob in zip(gmaps, obs)), dtype=np.int64, count=B)
        start = np.fromiter((g.index(g.start_vp) for g in gmaps), dtype=np.int64, count=B)
        ar = np.arange(cap)
        valid = ar[None] < n[:, None]
        if a.act_visited_nodes:
            seen = np.zeros((B, cap), dtype=bool)
            seen[bi, cur] = True
        else:
            seen = tb.seen & valid
        n_vis_all = seen.sum(1)
        if a.enc_full_graph:               # [visited in id order | unvisited in id order]
            key = np.where(valid, np.where(seen, ar[None], cap + ar[None]), 4 * cap)
            m, n_vis = n.copy(), n_vis_all
        else:                              # unvisited only
            key = np.where(valid & ~seen, ar[None], 4 * cap)
            m, n_vis = n - n_vis_all, np.zeros(B, dtype=np.int64)
        n_unv = n - n_vis_all
        order = np.argsort(key, axis=1, kind="stable")
        M = int(m.max())
        G = self._bucket(1 + M, self.node_buckets)
        ids = order[:, :M]                                             # (B, M) node ids, garbage past m[b]
        jm = np.arange(M)[None] < m[:, None]
        ids = np.where(jm, ids, 0)
        cid, cm = self._cand_ids(gmaps, cand_vpids)
        C = int(cm.sum(1).max()) if B else 0
        tgt = np.concatenate([ids, cid, start[:, None]], 1)            # (B, T)
        tm = np.concatenate([jm, cm, np.ones((B, 1), dtype=bool)], 1)
        T = tgt.shape[1]
        delta = tb.pos[bi[:, None], tgt] - tb.pos[bi, cur][:, None, :]
        is_cur = tgt == cur[:, None]
        graph = np.where(is_cur, 0.0, tb.dist[bi[:, None], cur[:, None], tgt])
        graph = np.where(np.isfinite(graph), graph, float(UNREACHABLE))
        via = tb.via[bi[:, None], cur[:, None], tgt]
        # routes with more than one leg are unrolled along the pivots (FloydGraph.path) -- one native call for the batch
        # (gridmm_route_lengths; ~350 Python recursions per step before)
        hops = np.empty((B, T), dtype=np.float64)
        tgt_c, tm_c = np.ascontiguousarray(tgt, dtype=np.int64), np.ascontiguousarray(tm, dtype=np.uint8)
        _lib.check(_lib.load().gridmm_route_lengths(tb.via.ctypes.data, B, cap, cur.ctypes.data, tgt_c.ctypes.data,
                                                    tm_c.ctypes.data, T, hops.ctypes.data), "gridmm_route_lengths")
        bh = np.array([float(ob["heading"]) for ob in obs])
        be = np.array([float(ob["elevation"]) for ob in obs])
        feats = batched_pos_features(delta.reshape(-1, 3), np.repeat(bh, T), np.repeat(be, T), graph.reshape(-1),
                                     hops.reshape(-1)).reshape(B, T, -1)
        F = feats.shape[2]
        V1 = pano_embeds.shape[1] + 1
        gpos = np.zeros((B, G, F), dtype=np.float32)
        gpos[:, 0] = np.array([0, 1, 0, 1] + [0] * (F - 4), dtype=np.float32)      # the stop token
        gpos[:, 1:M + 1] = feats[:, :M] * jm[:, :, None]
        vpos = np.zeros((B, V1, 2 * F), dtype=np.float32)
        vpos[:, :, :F] = feats[:, T - 1][:, None, :]
        if C:
            vpos[:, 1:C + 1, F:] = feats[:, M:M + C] * cm[:, :C, None]
        pair = np.zeros((B, G, G), dtype=np.float32)
        if M:
            sub = tb.dist[bi[:, None, None], ids[:, :, None], ids[:, None, :]]
            sub = np.where(np.isfinite(sub), sub, float(UNREACHABLE))
            sub[:, np.arange(M), np.arange(M)] = 0.0
            pair[:, 1:M + 1, 1:M + 1] = sub * (jm[:, :, None] & jm[:, None, :])
        steps = np.zeros((B, G), dtype=np.int64)
        steps[:, 1:M + 1] = tb.step[bi[:, None], ids] * jm
        visited = np.zeros((B, G), dtype=bool)
        visited[:, 1:M + 1] = np.arange(M)[None] < n_vis[:, None]
        slot = np.zeros((B, G), dtype=np.int64)
        slot[:, 1:M + 1] = (ids + 1) * jm
        if self.cnt.shape[1] <= cap:
            self._grow(cap)
        c = self.cnt[bi[:, None], slot[:, 1:M + 1]]
        if ((c == 0) & jm).any():
            i, j = [int(v[0]) for v in np.nonzero((c == 0) & jm)]
            raise KeyError("graph node without an embedding: %s" % gmaps[i].names[int(ids[i, j])])
        inv = np.ones((B, G), dtype=np.float32)
        inv[:, 1:M + 1] = np.where(jm, np.float32(1.0) / np.maximum(c, 1).astype(np.float32), np.float32(1.0))
        # fusion maps (vilmodel.py:884-899): candidate column of every unvisited node, candidates that are visited nodes
        cand_of_node = np.full((B, G), -2, dtype=np.int32)
        cand_visited = np.zeros((B, V1), dtype=np.uint8)
        if C:
            cvis = seen[bi[:, None], cid] & cm if a.enc_full_graph else np.zeros_like(cm)
            cand_visited[:, 1:C + 1] = cvis[:, :C]
            eq = (ids[:, :, None] == cid[:, None, :]) & (cm & ~cvis)[:, None, :]                    # (B, M, C)
            last = cid.shape[1] - 1 - np.argmax(eq[:, :, ::-1], axis=2)                             # last match wins
            col = np.where(eq.any(2), last + 1, -1)
        else:
            col = np.full((B, M), -1, dtype=np.int64)
        unv_node = jm & (np.arange(M)[None] >= n_vis[:, None])
        cand_of_node[:, 1:M + 1] = np.where(unv_node, col, -2)
        vpids = [[None] + [g.names[k] for k in ids[i, :m[i]]] for i, g in enumerate(gmaps)]
        lens = m + 1
        slot_d, inv_d = self._dev(slot), self._dev(inv)
        out = {
            "gmap_vpids": vpids, "gmap_img_embeds": None, "gmap_step_ids": self._dev(steps),
            "gmap_pos_fts": self._dev(gpos), "gmap_visited_masks": self._dev(visited), "gmap_visited_masks_host": np.asarray(visited, dtype=bool),
            "gmap_pair_dists": self._dev(pair),
            "gmap_masks": self._dev(np.arange(G)[None] < lens[:, None]), "no_vp_left": [bool(v == 0) for v in n_unv],
            "fusion_maps": (self._dev(cand_of_node), self._dev(cand_visited)),
        }
        out = self._vp_part(out, pano_embeds, cand_vpids, view_lens, nav_types, vpos)
        self.stage.flush()                         # every host array of the step: one asynchronous upload
        rows = torch.arange(B, device=self.device).unsqueeze(1)
        out["gmap_img_embeds"] = self.pool[rows, slot_d] * inv_d.unsqueeze(2)
        return out

    def _vp_part(self, out, pano_embeds, cand_vpids, view_lens, nav_types, vpos):
        B, V1 = pano_embeds.shape[0], pano_embeds.shape[1] + 1
        vp_img = torch.cat([torch.zeros_like(pano_embeds[:, :1]), pano_embeds], 1)
        nav_masks = torch.cat([torch.ones(B, 1, dtype=torch.bool, device=self.device), nav_types == 1], 1)
        vl = view_lens + 1
        out.update({
            "vp_img_embeds": vp_img, "vp_pos_fts": self._dev(vpos),
            "vp_masks": torch.arange(V1, device=vl.device).unsqueeze(0) < vl.unsqueeze(1),
            "vp_nav_masks": nav_masks, "vp_cand_vpids": [[None] + x for x in cand_vpids], "vp_obj_masks": None,
        })
        return out

    def _navigation_loop(self, obs, gmaps, pano_embeds, cand_vpids, view_lens, nav_types):
        """Per-episode form (maps that own their arrays)."""
        a, B = self.args, len(obs)
        recs, deltas, graphs, hopss, base_h, base_e = [], [], [], [], [], []
        for i, g in enumerate(gmaps):
            cur = g.index(obs[i]["viewpoint"])
            n = g.n
            if a.act_visited_nodes:
                vis = np.array([cur], dtype=np.int64)
                unv = np.array([k for k in range(n) if k != cur], dtype=np.int64)
            else:
                seen = g.seen[:n]
                vis, unv = np.flatnonzero(seen), np.flatnonzero(~seen)
            ids = np.concatenate([vis, unv]) if a.enc_full_graph else unv
            n_vis = len(vis) if a.enc_full_graph else 0
            cid = np.fromiter((g.index(c) for c in cand_vpids[i]), dtype=np.int64, count=len(cand_vpids[i]))
            tgt = np.concatenate([ids, cid, np.array([g.index(g.start_vp)], dtype=np.int64)])
            d, gr, hp = g.relative_geometry(cur, tgt)
            deltas.append(d)
            graphs.append(gr)
            hopss.append(hp)
            base_h.append(np.full(len(tgt), float(obs[i]["heading"])))
            base_e.append(np.full(len(tgt), float(obs[i]["elevation"])))
            recs.append((g, ids, n_vis, len(unv), len(cid)))
        feats = batched_pos_features(np.concatenate(deltas), np.concatenate(base_h), np.concatenate(base_e),
                                     np.concatenate(graphs), np.concatenate(hopss))
        F = feats.shape[1]
        G = self._bucket(1 + max(len(r[1]) for r in recs), self.node_buckets)
        V1 = pano_embeds.shape[1] + 1
        cand_of_node = np.full((B, G), -2, dtype=np.int32)     # integer form of the vpid-keyed fusion loops
        cand_visited = np.zeros((B, V1), dtype=np.uint8)       # (vilmodel.py:884-899; GlocalTextPathNavCMT._fusion_index_maps)
        gpos = np.zeros((B, G, F), dtype=np.float32)
        pair = np.zeros((B, G, G), dtype=np.float32)
        steps = np.zeros((B, G), dtype=np.int64)
        visited = np.zeros((B, G), dtype=bool)
        slot = np.zeros((B, G), dtype=np.int64)
        inv = np.ones((B, G), dtype=np.float32)
        vpos = np.zeros((B, V1, 2 * F), dtype=np.float32)
        lens = np.zeros(B, dtype=np.int64)
        stop_row = np.array([0, 1, 0, 1] + [0] * (F - 4), dtype=np.float32)  # the stop token: zero angles and distances
        vpids, no_vp_left, o = [], [], 0
        for i, (g, ids, n_vis, n_unv, nc) in enumerate(recs):
            m = len(ids)
            lens[i] = m + 1
            gpos[i, 0] = stop_row
            gpos[i, 1:m + 1] = feats[o:o + m]
            vpos[i, 1:nc + 1, F:] = feats[o + m:o + m + nc]
            vpos[i, :, :F] = feats[o + m + nc]
            o += m + nc + 1
            if m:
                sub = g.dist[np.ix_(ids, ids)]
                sub = np.where(np.isfinite(sub), sub, float(UNREACHABLE))
                np.fill_diagonal(sub, 0.0)
                pair[i, 1:m + 1, 1:m + 1] = sub
            names = [g.names[k] for k in ids]
            steps[i, 1:m + 1] = [g.step_id.get(v, 0) for v in names]
            visited[i, 1:1 + n_vis] = True
            slot[i, 1:m + 1] = ids + 1
            c = self.cnt[i, ids + 1]
            if (c == 0).any():
                raise KeyError("graph node without an embedding: %s" % names[int(np.flatnonzero(c == 0)[0])])
            inv[i, 1:m + 1] = np.float32(1.0) / c.astype(np.float32)
            vpids.append([None] + names)
            no_vp_left.append(n_unv == 0)
            seen_names = set(names[:n_vis])
            col = {}
            for j, cv in enumerate(cand_vpids[i]):
                if cv in seen_names:
                    cand_visited[i, j + 1] = 1
                else:
                    col[cv] = j + 1
            for j in range(n_vis, m):
                cand_of_node[i, j + 1] = col.get(names[j], -1)
        slot_d, inv_d = self._dev(slot), self._dev(inv)
        out = {
            "gmap_vpids": vpids, "gmap_img_embeds": None, "gmap_step_ids": self._dev(steps),
            "gmap_pos_fts": self._dev(gpos), "gmap_visited_masks": self._dev(visited), "gmap_visited_masks_host": np.asarray(visited, dtype=bool),
            "gmap_pair_dists": self._dev(pair),
            "gmap_masks": self._dev(np.arange(G)[None] < lens[:, None]), "no_vp_left": no_vp_left,
            "fusion_maps": (self._dev(cand_of_node), self._dev(cand_visited)),
        }
        out = self._vp_part(out, pano_embeds, cand_vpids, view_lens, nav_types, vpos)
        self.stage.flush()                         # every host array of the step: one asynchronous upload
        rows = torch.arange(B, device=self.device).unsqueeze(1)
        out["gmap_img_embeds"] = self.pool[rows, slot_d] * inv_d.unsqueeze(2)
        return out
