"""TEST INFRASTRUCTURE ONLY -- imports the upstream reference (read-only, /root/reference) as a
CPU oracle inside the build container.  Never imported by the product path, never shipped to
the GPU box (the reference does not exist there).  Used by oracle/gen_golden.py to mint the
fixtures under tests/golden/ and by the optional ``refcheck`` tests that compare the build's
own CPU restatement (oracle/*.py) with the reference op by op.

The reference needs torchvision / mmseg / mmengine / ftfy / cv2 / BLIP at *import* time only
(never on the vision hot path); those are absent here, so placeholder modules are installed in
``sys.modules`` first (recipe: SURVEY.md Appendix A).  No pretrained tag is ever resolved:
``create_model`` is rebound to build random-init fp32 CPU models which then receive the
build's deterministic weights.
"""
from __future__ import annotations

import importlib
import importlib.machinery
import os
import sys
import types

REF_ROOT = os.environ.get("SEGEARTH_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "segmentor.py"))


class _PermissiveMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return cls()


class _Permissive(metaclass=_PermissiveMeta):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Permissive()


def _stub(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []  # behave as a package so sub-imports resolve through sys.modules
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_installed = False


def install_stubs() -> None:
    global _installed
    if _installed:
        return
    import torch
    import torch.nn as nn
    try:
        import transformers  # noqa: F401  must precede the torchvision stub (find_spec probe)
    except Exception:
        pass

    if "torchvision" not in sys.modules:
        _stub("torchvision")
        _stub("torchvision.ops")
        _stub("torchvision.ops.misc", FrozenBatchNorm2d=nn.BatchNorm2d)
        names = ("Normalize", "Compose", "RandomResizedCrop", "InterpolationMode", "ToTensor",
                 "Resize", "CenterCrop", "ColorJitter", "Grayscale")
        _stub("torchvision.transforms", **{n: _Permissive for n in names})
        _stub("torchvision.transforms.functional")
    if "ftfy" not in sys.modules:
        _stub("ftfy", fix_text=lambda s: s)
    if "cv2" not in sys.modules:
        _stub("cv2")

    class BaseSegmentor(nn.Module):
        def __init__(self, data_preprocessor=None, **kw):
            super().__init__()
            self.data_preprocessor = data_preprocessor

    class SegDataPreProcessor:
        def __init__(self, **kw):
            self.kw = kw

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

    class PixelData:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    _stub("mmseg")
    _stub("mmseg.models")
    _stub("mmseg.models.segmentors", BaseSegmentor=BaseSegmentor)
    _stub("mmseg.models.data_preprocessor", SegDataPreProcessor=SegDataPreProcessor)
    _stub("mmseg.registry", MODELS=_Registry(), DATASETS=_Registry())
    _stub("mmengine")
    _stub("mmengine.structures", PixelData=PixelData)
    _stub("BLIP")
    _stub("BLIP.models")
    _stub("BLIP.models.blip_retrieval", blip_retrieval=None)
    _installed = True


_ref_modules = {}


def ref(module: str):
    """Import ``module`` from the reference tree (e.g. ``ref('outlier_suppression')``)."""
    if module in _ref_modules:
        return _ref_modules[module]
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    install_stubs()
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # The repo root ships drop-in files with the SAME module names as the reference
    # (segmentor.py, segearth_segmentor.py).  Make sure the reference's win for this import.
    shadow = {"segmentor", "segearth_segmentor"}
    top = module.split(".")[0]
    saved_path = list(sys.path)
    if top in shadow:
        sys.path[:] = [REF_ROOT] + [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(_repo_root())]
        sys.modules.pop(top, None)
    try:
        m = importlib.import_module(module)
    finally:
        sys.path[:] = saved_path
    if top in shadow:
        # keep a private handle; free the public name for the build's own drop-in module
        sys.modules.pop(top, None)
    _ref_modules[module] = m
    return m


def _repo_root() -> str:
    return os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_clip(model_name: str, quick_gelu: bool = True):
    """Random-init fp32 CPU CLIP from the reference factory; never touches a pretrained tag."""
    factory = ref("open_clip.factory")
    return factory.create_model(model_name, pretrained=None, precision="fp32", device="cpu",
                                force_quick_gelu=quick_gelu)


def build_vision_tower(cfg, state_dict_np, quick_gelu=None):
    """Reference ``VisionTransformer`` with arbitrary dims (reference open_clip/transformer.py:341-362)
    loaded with the build's deterministic weights."""
    import torch
    tr = ref("open_clip.transformer")
    qg = cfg.quick_gelu if quick_gelu is None else quick_gelu
    act = tr.QuickGELU if qg else torch.nn.GELU
    vt = tr.VisionTransformer(
        image_size=cfg.image_size, patch_size=cfg.patch, width=cfg.width, layers=cfg.layers,
        heads=cfg.heads, mlp_ratio=cfg.mlp_ratio, output_dim=cfg.embed_dim,
        act_layer=act, norm_layer=tr.LayerNorm, output_tokens=False)
    sd = {k: torch.from_numpy(v.copy()) for k, v in state_dict_np.items()}
    vt.load_state_dict(sd, strict=True)
    return vt.eval()
