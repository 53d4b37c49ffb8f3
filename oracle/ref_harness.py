"""Reference-import harness.  TEST INFRASTRUCTURE — runs ONLY in the build container.

Imports MrZihan/GridMM's own Python from /root/reference (read-only) with the
harness-side stand-ins SURVEY.md Appendix C lists, so that

  * oracle/gen_golden.py can dump golden input/output vectors into tests/golden/
  * tests can validate the oracle restatement (oracle/*.py) against the real thing
    when /root/reference is present (it is absent on the GPU box).

Nothing here ships in the product path and nothing here is copied from the
reference: it only *calls* it.

Reference entry points driven:
  map_nav_src/models/vilmodel.py:676   GlocalTextPathNavCMT
  map_nav_src/r2r/env.py:267           EnvBatch.getGlobalMap
  map_nav_src/r2r/env.py:242           EnvBatch.get_gridmap_pos_fts
  map_nav_src/r2r/env.py:115           get_rel_position
"""
import os
import sys
import types
import zlib

import numpy as np
import torch

REF_ROOT = os.environ.get("GRIDMM_REFERENCE", "/root/reference")
REF_NAV = os.path.join(REF_ROOT, "map_nav_src")


def reference_available():
    return os.path.isdir(REF_NAV)


# --------------------------------------------------------------------------- shims
class _AttrDict(dict):
    """Stand-in for easydict.EasyDict (attribute access on a dict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def install_shims():
    if "easydict" not in sys.modules:
        m = types.ModuleType("easydict")
        m.EasyDict = _AttrDict
        sys.modules["easydict"] = m
    for name in ("MatterSim", "cv2", "h5py", "imutils", "jsonlines", "line_profiler"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)


def use_tree(tree):
    """Make `tree` (map_nav_src or pretrain_src) THE reference tree for top-level imports: both trees have a `utils`
    package (and `models` / `model`, `optim`, `data`), so whichever was imported first would shadow the other's --
    r2r/env.py:15 `from utils.data import ...` found pretrain_src/utils once an optimiser fixture had run in the same
    process.  Puts `tree` first on sys.path, drops the other reference trees from it, and forgets every cached module
    that was loaded from another reference tree (module objects already handed out keep working: they hold their own
    references)."""
    tree = os.path.abspath(tree)
    root = os.path.abspath(REF_ROOT) + os.sep
    sys.path[:] = [p for p in sys.path if not os.path.abspath(p or ".").startswith(root)]
    sys.path.insert(0, tree)
    for name, m in list(sys.modules.items()):
        if name.startswith("_ref_") or name.startswith("vlnce_baselines"):
            continue                       # the VLN-CE modules are imported under private package names
        where = getattr(m, "__file__", None)
        if where is None:
            try:
                where = (list(getattr(m, "__path__", [])) or [None])[0]   # namespace packages (no __init__.py)
            except Exception:
                where = None
        if where and os.path.abspath(where).startswith(root) and not os.path.abspath(where).startswith(tree + os.sep):
            del sys.modules[name]
    import importlib
    importlib.invalidate_caches()


_VIL = None
_ENV = None


def import_vilmodel():
    """Import the reference map_nav_src/models/vilmodel.py (transformers 5.x shims)."""
    global _VIL
    if _VIL is not None:
        return _VIL
    install_shims()
    use_tree(REF_NAV)
    from models import vilmodel  # noqa: the reference's module

    # transformers 5.x: init_weights()/post_init() are incompatible with this class.
    vilmodel.BertPreTrainedModel.init_weights = lambda self: None
    _VIL = vilmodel
    return vilmodel


def import_env():
    """Import the reference map_nav_src/r2r/env.py with simulator stand-ins."""
    global _ENV
    if _ENV is not None:
        return _ENV
    install_shims()
    use_tree(REF_NAV)
    from r2r import env  # noqa

    _ENV = env
    return env


# --------------------------------------------------------------------------- config / weights
def make_config(**over):
    """BertConfig() + the attributes of map_nav_src/models/vlnbert_init.py:38-56."""
    import transformers

    cfg = transformers.BertConfig()
    cfg.max_action_steps = 100
    cfg.image_feat_size = 768
    cfg.angle_feat_size = 4
    cfg.obj_feat_size = 0
    cfg.obj_loc_size = 3
    cfg.num_l_layers = 9
    cfg.num_pano_layers = 2
    cfg.num_x_layers = 4
    cfg.graph_sprels = True
    cfg.glocal_fuse = True
    cfg.fix_lang_embedding = False
    cfg.fix_pano_embedding = False
    cfg.fix_local_branch = False
    cfg.update_lang_bert = True
    cfg.output_attentions = True
    cfg.output_hidden_states = False
    cfg.pred_head_dropout_prob = 0.1
    cfg.use_lang2visn_attn = False
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def det_tensor(name, shape, seed=0):
    """Deterministic parameter values keyed by parameter name (never committed).

    Legacy RandomState is bit-stable across numpy versions.
    matrices ~ N(0, .04), biases ~ N(0, .02), LayerNorm weight ~ 1 + N(0, .05).
    """
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    x = rs.standard_normal(size=tuple(shape)).astype(np.float32)
    lname = name.lower()
    is_ln = ("layernorm" in lname or "layer_norm" in lname or ".norm" in lname
             or lname.endswith("net.2.weight") or lname.endswith("net.2.bias")
             or lname.endswith("embeddings.1.weight") or lname.endswith("embeddings.1.bias"))
    if len(shape) >= 2:
        x *= 0.04
    elif is_ln and name.endswith("weight"):
        x = 1.0 + 0.05 * x
    else:
        x *= 0.02
    return torch.from_numpy(np.ascontiguousarray(x))


def det_state_dict(model, seed=0):
    sd = {}
    for k, v in model.state_dict().items():
        if not v.dtype.is_floating_point:
            sd[k] = v.clone()
            continue
        sd[k] = det_tensor(k, v.shape, seed).to(v.dtype)
    return sd


def build_ref_model(seed=0, **cfg_over):
    vil = import_vilmodel()
    cfg = make_config(**cfg_over)
    m = vil.GlocalTextPathNavCMT(cfg).eval()
    m.load_state_dict(det_state_dict(m, seed))
    return m


# --------------------------------------------------------------------------- pre-training twin
REF_PRETRAIN = os.path.join(REF_ROOT, "pretrain_src")
_PRE = None


def import_pretrain():
    """Import pretrain_src/model/{pretrain_cmt,vilmodel}.py (package name `model`)."""
    global _PRE
    if _PRE is not None:
        return _PRE
    install_shims()
    use_tree(REF_PRETRAIN)
    from model import pretrain_cmt, vilmodel as pvil  # noqa: the reference's modules

    pvil.BertPreTrainedModel.init_weights = lambda self: None
    # transformers 5.x has no _tie_or_clone_weights: tie by hand after construction (pretrain_cmt.py:66-69)
    pretrain_cmt.GlocalTextPathCMTPreTraining.tie_weights = lambda self: None
    _PRE = pretrain_cmt
    return pretrain_cmt


PRETRAIN_REDUCED = dict(num_l_layers=1, num_pano_layers=1, num_x_layers=2, intermediate_size=64, vocab_size=2000,
                        use_lang2visn_attn=True, pretrain_tasks=["mlm", "mrc", "sap"], image_prob_size=50,
                        obj_prob_size=0, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


def build_ref_pretrain_model(seed=0, **cfg_over):
    """GlocalTextPathCMTPreTraining (pretrain_cmt.py:38-69) with deterministic weights, MLM decoder tied."""
    pre = import_pretrain()
    cfg = make_config(**dict(PRETRAIN_REDUCED, **cfg_over))
    m = pre.GlocalTextPathCMTPreTraining(cfg)
    sd = det_state_dict(m, seed)
    if "mlm" in cfg.pretrain_tasks:
        sd["mlm_head.predictions.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    m.load_state_dict(sd)
    if "mlm" in cfg.pretrain_tasks:
        m.mlm_head.predictions.decoder.weight = m.bert.embeddings.word_embeddings.weight
    return m


# --------------------------------------------------------------------------- env driver
class _HistArray(np.ndarray):
    """ndarray whose `== []` is a scalar False (numpy-2 fix for env.py:298)."""

    def __eq__(self, other):
        if isinstance(other, list):
            return False
        return np.ndarray.__eq__(self, other)

    __hash__ = None


class _FakeLoc:
    def __init__(self, vp):
        self.viewpointId = vp


class _FakeState:
    def __init__(self, scan, vp, heading):
        self.scanId = scan
        self.location = _FakeLoc(vp)
        self.heading = heading


class _FakeSim:
    def __init__(self):
        self.state = None

    def getState(self):
        return [self.state]


class _DictDB:
    def __init__(self, d):
        self.d = d

    def get_image_feature(self, scan, vp):
        return self.d["%s_%s" % (scan, vp)]


class RefGridEnv:
    """Drives the reference EnvBatch.getGlobalMap without a simulator.

    depth_db:  {key: uint16 (36, 128, 128, 1)} as DepthFeaturesDB returns (env.py:93)
    clip_db:   {key: float16 (12, 50, 768)}  as SemanticFeaturesDB returns
    vp_info:   {key: {"x": float, "y": float}}
    """

    def __init__(self, batch_size, depth_db, clip_db, vp_info):
        E = import_env()
        eb = E.EnvBatch.__new__(E.EnvBatch)
        eb.batch_size = batch_size
        eb.sims = [_FakeSim() for _ in range(batch_size)]
        eb.global_semantic = [[] for _ in range(batch_size)]
        eb.global_position_x = [[] for _ in range(batch_size)]
        eb.global_position_y = [[] for _ in range(batch_size)]
        eb.global_mask = [[] for _ in range(batch_size)]
        eb.max_x = [-10000 for _ in range(batch_size)]
        eb.min_x = [10000 for _ in range(batch_size)]
        eb.max_y = [-10000 for _ in range(batch_size)]
        eb.min_y = [10000 for _ in range(batch_size)]
        eb.heading = [0 for _ in range(batch_size)]
        eb.global_map = [[] for _ in range(batch_size)]
        eb.DepthDB = _DictDB(depth_db)
        eb.SemanticDB = _DictDB(clip_db)
        eb.viewpoint_info = vp_info
        self.eb = eb

    def step(self, i, scan, vp, heading):
        """One getGlobalMap call for episode i, state written back as getStates does (env.py:397)."""
        eb = self.eb
        eb.sims[i].state = _FakeState(scan, vp, heading)
        (tid, sem, gx, gy, gm, gmap, mx, mnx, my, mny, pos_fts) = eb.getGlobalMap(i)
        eb.global_semantic[i] = np.asarray(sem).view(_HistArray)
        eb.global_position_x[i] = gx
        eb.global_position_y[i] = gy
        eb.global_mask[i] = gm
        eb.global_map[i] = gmap
        eb.max_x[i], eb.min_x[i], eb.max_y[i], eb.min_y[i] = mx, mnx, my, mny
        return np.asarray(sem), np.array(gmap, copy=True), np.asarray(pos_fts)


# --------------------------------------------------------------------------- VLN-CE twin (row a12)
class _Anything:
    """Permissive stand-in for habitat / gym / timm / torchvision symbols: attribute access, calling,
    decorating and even sub-classing all succeed (harness-side only; nothing numeric goes through it)."""

    def __init__(self, name="stub"):
        self.__name__ = name

    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything(k)

    def __call__(self, *a, **kw):
        if len(a) == 1 and isinstance(a[0], type) and not kw:
            return a[0]          # used as a class decorator
        return _Anything("call")

    def __mro_entries__(self, bases):
        return (object,)

    def __iter__(self):
        return iter(())


class _AnyModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything(k)


_VLNCE = None


def import_vlnce_policy():
    """Import VLN_CE/vlnce_baselines/models/Policy_ViewSelection_GridMap.py with stand-ins for every
    simulator / vision dependency; only its pure NumPy methods (getGlobalMap, get_rel_position,
    get_gridmap_pos_fts: :632-825) are ever called."""
    global _VLNCE
    if _VLNCE is not None:
        return _VLNCE
    import importlib.util
    root = os.path.join(REF_ROOT, "VLN_CE")
    for name in ("gym", "habitat", "habitat_baselines", "habitat_baselines.common",
                 "habitat_baselines.common.baseline_registry", "habitat_baselines.rl", "habitat_baselines.rl.models",
                 "habitat_baselines.rl.models.rnn_state_encoder", "habitat_baselines.rl.ppo",
                 "habitat_baselines.rl.ppo.policy", "timm", "timm.data", "timm.data.transforms_factory",
                 "torchvision", "torchvision.transforms", "cv2", "imutils", "waypoint_prediction",
                 "waypoint_prediction.utils", "vlnce_baselines.models.gridmap", "vlnce_baselines.models.gridmap.vlnbert_init",
                 "vlnce_baselines.common", "vlnce_baselines.common.aux_losses", "vlnce_baselines.models.encoders",
                 "vlnce_baselines.models.encoders.instruction_encoder", "vlnce_baselines.models.encoders.resnet_encoders",
                 "vlnce_baselines.models.policy"):
        if name not in sys.modules:
            sys.modules[name] = _AnyModule(name)
    for pkg, sub in (("vlnce_baselines", "vlnce_baselines"), ("vlnce_baselines.models", "vlnce_baselines/models")):
        if pkg not in sys.modules or isinstance(sys.modules[pkg], _AnyModule):
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(root, sub)]
            sys.modules[pkg] = m
    spec = importlib.util.spec_from_file_location(
        "vlnce_baselines.models.Policy_ViewSelection_GridMap",
        os.path.join(root, "vlnce_baselines", "models", "Policy_ViewSelection_GridMap.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _VLNCE = mod
    return mod


_VLNCE_VIL = None


def import_vlnce_vilmodel():
    """Import VLN_CE/vlnce_baselines/models/gridmap/vilmodel.py as a standalone package (`_ref_vlnce_gridmap`), with
    `timm` stubbed and the two vision towers (CLIP, ViT: row f4, not on this path) replaced by empty modules."""
    global _VLNCE_VIL
    if _VLNCE_VIL is not None:
        return _VLNCE_VIL
    import importlib
    import transformers  # noqa: F401  (must be imported before the timm stand-in exists: it probes find_spec("timm"))
    from transformers import BertPreTrainedModel  # noqa: F401
    install_shims()
    for name in ("timm", "timm.data", "timm.data.transforms_factory"):
        if name not in sys.modules:
            sys.modules[name] = _AnyModule(name)
    pkg = types.ModuleType("_ref_vlnce_gridmap")
    pkg.__path__ = [os.path.join(REF_ROOT, "VLN_CE", "vlnce_baselines", "models", "gridmap")]
    sys.modules["_ref_vlnce_gridmap"] = pkg
    vil = importlib.import_module("_ref_vlnce_gridmap.vilmodel")
    vil.BertPreTrainedModel.init_weights = lambda self: None
    vil.CLIP = lambda **kw: torch.nn.Identity()
    vil.timm = types.SimpleNamespace(create_model=lambda *a, **kw: torch.nn.Identity())
    _VLNCE_VIL = vil
    return vil


def build_ref_vlnce_model(seed=0, **cfg_over):
    vil = import_vlnce_vilmodel()
    cfg = make_config(**cfg_over)
    m = vil.GlocalTextPathNavCMT(cfg).eval()
    m.load_state_dict(det_state_dict(m, seed))
    return m


class RefVlnceGridEnv:
    """Drives the reference VLN-CE GridMap.getGlobalMap (Policy_ViewSelection_GridMap.py:689-825)."""

    def __init__(self, batch_size, dataset="R2R", max_dist=25):
        mod = import_vlnce_policy()
        mod.DATASET, mod.MAX_DIST = dataset, max_dist
        g = object.__new__(mod.GridMap)
        g.global_fts = [[] for _ in range(batch_size)]
        g.global_position_x = [[] for _ in range(batch_size)]
        g.global_position_y = [[] for _ in range(batch_size)]
        g.global_mask = [[] for _ in range(batch_size)]
        g.global_map_index = [[] for _ in range(batch_size)]
        g.max_x = [-10000 for _ in range(batch_size)]
        g.min_x = [10000 for _ in range(batch_size)]
        g.max_y = [-10000 for _ in range(batch_size)]
        g.min_y = [10000 for _ in range(batch_size)]
        g.headings = [0 for _ in range(batch_size)]
        self.g = g

    def step(self, i, position, heading, depth_full, grid_ft):
        """depth_full (12,256,256) float32 metres; grid_ft (12,50,768)."""
        g = self.g
        g.headings[i] = heading
        (fts, gx, gy, gm, gmap, mx, mnx, my, mny, pos) = g.getGlobalMap(i, position, heading, depth_full, grid_ft, None)
        g.global_fts[i] = np.asarray(fts).view(_HistArray)
        g.global_position_x[i], g.global_position_y[i], g.global_mask[i] = gx, gy, gm
        g.global_map_index[i] = gmap
        g.max_x[i], g.min_x[i], g.max_y[i], g.min_y[i] = mx, mnx, my, mny
        return np.asarray(fts), np.array(gmap, copy=True), np.asarray(pos)
