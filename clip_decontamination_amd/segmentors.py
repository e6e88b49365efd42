"""Host-side implementation shared by the two drop-in entry points (repo-root ``segmentor.py`` /
``segearth_segmentor.py``).  Same class names, constructor kwargs, public methods and return types as the
reference (segmentor.py:25-546, segearth_segmentor.py:22-326); the bodies call the HIP library.

What differs, and why (none of it is on the hot path):
  * weights: every pretrained tag of the reference resolves to a download (segmentor.py:69-128), impossible
    offline.  The drop-in takes ``checkpoint=`` (a local state dict holding ``visual.*`` tensors, loaded with
    ``weights_only=True``) or, with ``synthetic_ok=True``, the deterministic synthetic weights used for parity.
  * text features: ``tokenizer=`` (callable: list[str] -> int ids [n, 77]; the BPE vocabulary is a downloaded
    asset) together with a ``checkpoint`` that holds the text tower runs the reference's own route
    (segmentor.py:157-174) on the HIP text tower (``sg_text_encode``).  Alternatives: ``text_features=`` ([Q,E]
    tensor / .npy / .pt path) or ``text_encoder=`` (callable: list[str] -> [n,E] tensor); the 80-template
    prompt ensemble is applied here in every case.
  * ``precision=`` (default ``"f16x2"``): the arithmetic of the tower.  The reference runs ``.half()`` on the GPU (segmentor.py:467) and fp32
    on the CPU; the default here is the most faithful mode that clears the 50 Mpix/s target -- two f16 planes per operand, three MFMAs
    per product: the fp32 path's logits (max |dlogit| 5e-7 on the bench tiles) and labels (identical up to fp32 ties) at ~95 Mpix/s for
    ViT-L/14.  ``"bf16"`` (225 Mpix/s, 5e-3 / 99.4 % labels), ``"f16"``, ``"fp8"`` trade accuracy for speed; ``"f32"`` is the f32-MFMA parity mode.
"""
from __future__ import annotations

import os
import warnings
from typing import Callable, Optional

import numpy as np
import torch
import torch.nn as nn

from . import weights as Wt
from .engine import (HipCLIP, HipVisionTower, OutlierSuppressionModule, SelfAttentionEnhancementModule,
                     SimilarityEnhancementModule, compute_padsize)
from .pipeline import SegPipeline
from .prompts import ensemble_prompts

try:  # the reference's plugin registry, when mmseg is installed
    from mmseg.models.segmentors import BaseSegmentor as _Base
    from mmseg.models.data_preprocessor import SegDataPreProcessor
    from mmseg.registry import MODELS
    from mmengine.structures import PixelData
    HAVE_MMSEG = True
except Exception:  # stand-alone use (tests, bench, serving)
    HAVE_MMSEG = False

    class _Base(nn.Module):
        def __init__(self, data_preprocessor=None, **_):
            super().__init__()
            self.data_preprocessor = data_preprocessor

    class SegDataPreProcessor:
        def __init__(self, mean=None, std=None, bgr_to_rgb=False, **_):
            self.mean, self.std, self.bgr_to_rgb = mean, std, bgr_to_rgb

    class PixelData:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

    MODELS = _Registry()


def get_cls_idx(path):
    """One class per line, commas separate synonym queries (reference segmentor.py:611-622)."""
    with open(path, "r") as f:
        name_sets = f.readlines()
    class_names, class_indices = [], []
    for idx, line in enumerate(name_sets):
        names_i = line.split(",")
        class_names += names_i
        class_indices += [idx for _ in range(len(names_i))]
    class_names = [item.replace("\n", "") for item in class_names]
    return class_names, class_indices


def _vit_name(clip_type: str, vit_type: str) -> str:
    """The architecture the reference would have built for (clip_type, vit_type) (segmentor.py:69-128)."""
    if "tiny" in vit_type:
        return vit_type
    if "B" in vit_type:
        return "ViT-B-16" if clip_type in ("CLIP", "OpenCLIP", "MetaCLIP") else "ViT-B-32"
    if "L" in vit_type:
        return "ViT-L-14"
    if "H" in vit_type:
        return "ViT-H-14"
    raise ValueError(f"unsupported vit_type {vit_type!r}")


def _load_full_state(path: str):
    """The whole CLIP state dict (text tower included); loaders that execute nothing from the file."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return {k: torch.from_numpy(z[k]) for k in z.files}
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return sd["state_dict"] if isinstance(sd, dict) and "state_dict" in sd else sd


def _load_state(path: str):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    elif path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            sd = {k: torch.from_numpy(z[k]) for k in z.files}
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "state_dict" in sd:
            sd = sd["state_dict"]
    vis = {k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}
    return vis if vis else sd


def _to_bool(value):
    if isinstance(value, str):
        return value.strip().lower() in ("1", "true", "yes", "y", "on")
    return bool(value)


class _HipSegmentorBase(_Base):
    """Everything both reference classes share: construction of the tower, text features, slide / feature /
    post-process methods."""

    _to_bool = staticmethod(_to_bool)

    def _setup(self, clip_type, vit_type, model_type, name_path, device, ignore_residual, prob_thd, logit_scale, slide_stride,
               slide_crop, cls_token_lambda, bg_idx, apply_sim_feat_up, sim_feat_up_cfg, global_debias_factor=0.0,
               checkpoint=None, text_features=None, text_encoder: Optional[Callable] = None, precision="f16x2",
               synthetic_ok=False, tiles_per_launch=None, jbu_checkpoint_ok=True, tokenizer: Optional[Callable] = None, tile_group=None):
        self.tile_group = tile_group                         # opt-in tile sharding (pipeline.resolve_tile_group); None = off
        if clip_type == "BLIP":
            raise NotImplementedError("clip_type='BLIP' is a different backbone (vendored BLIP) and is out of scope for the HIP path")
        self.clip_type, self.vit_type, self.model_type = clip_type, vit_type, model_type
        dev = torch.device(device) if not isinstance(device, torch.device) else device
        if dev.type != "cuda":
            raise RuntimeError("the drop-in segmentors run on the GPU only (HIP hot path, no CPU fallback)")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        quick = clip_type in ("CLIP", "MetaCLIP")            # openai / *-quickgelu tags (model.py:116,518)
        cfg = Wt.vit_config(_vit_name(clip_type, vit_type), quick_gelu=None if "tiny" in vit_type else quick)
        checkpoint = checkpoint or os.environ.get("SEGEARTH_CLIP_CHECKPOINT")
        if checkpoint:
            state = _load_state(checkpoint)
        elif synthetic_ok:
            warnings.warn("SegEarth drop-in: using deterministic SYNTHETIC vision weights (no pretrained checkpoint given)")
            state = Wt.make_vit_weights(cfg, seed=0)
        else:
            raise RuntimeError("no vision checkpoint: pass checkpoint=<path to a state dict with visual.* tensors> "
                               "(or set SEGEARTH_CLIP_CHECKPOINT); pretrained tags cannot be downloaded here. "
                               "synthetic_ok=True selects the synthetic parity weights.")
        visual = HipVisionTower(cfg, state, precision=precision, device=dev)
        visual.gem_ignore_residual = ignore_residual
        self.net = HipCLIP(visual)
        self.patch_size = visual.patch_size
        self.cls_token_lambda = cls_token_lambda
        self.global_debias_factor = global_debias_factor
        self.bg_idx = bg_idx
        self.apply_sim_feat_up = apply_sim_feat_up

        query_words, query_idx = get_cls_idx(name_path)
        self.num_queries = len(query_words)
        self.num_classes = max(query_idx) + 1
        self.query_idx = torch.tensor(query_idx, dtype=torch.int64, device=dev)
        E = cfg.embed_dim
        if text_features is not None:
            tf = text_features
            if isinstance(tf, str):
                tf = torch.from_numpy(np.load(tf, allow_pickle=False)) if tf.endswith(".npy") else torch.load(tf, map_location="cpu", weights_only=True)
            tf = torch.as_tensor(tf).float()
        elif text_encoder is not None:
            feats = []
            with torch.no_grad():
                for qw in query_words:                                   # segmentor.py:159-173
                    f = torch.as_tensor(text_encoder(ensemble_prompts(qw))).float()
                    f = f / f.norm(dim=-1, keepdim=True)
                    f = f.mean(dim=0)
                    feats.append((f / f.norm()).unsqueeze(0))
            tf = torch.cat(feats, 0)
        elif tokenizer is not None:
            # the reference's own route (segmentor.py:157-174): tokenizer -> net.encode_text, here on the HIP text tower
            from .engine import HipTextTower
            full = _load_full_state(checkpoint) if checkpoint else None
            if not full or "token_embedding.weight" not in full:
                raise RuntimeError("tokenizer= given but the checkpoint holds no text tower (token_embedding.weight ...)")
            tcfg = Wt.TEXT_CONFIGS[cfg.name] if cfg.name in Wt.TEXT_CONFIGS else None
            if tcfg is None:
                raise RuntimeError(f"no text configuration known for {cfg.name}")
            if tcfg.quick_gelu != quick:
                tcfg = Wt.TextConfig(**{**tcfg.__dict__, "quick_gelu": quick})
            self.net.text = HipTextTower(tcfg, full, precision="f32", device=dev)
            tf = self.net.text.query_features(tokenizer, query_words).cpu()
        elif synthetic_ok:
            warnings.warn("SegEarth drop-in: using SYNTHETIC text features")
            tf = torch.from_numpy(Wt.make_text_features(self.num_queries, E))
        else:
            raise RuntimeError("no text features: pass text_features=[Q,E], text_encoder=callable, or tokenizer=callable together "
                               "with a checkpoint that holds the text tower")
        if tuple(tf.shape) != (self.num_queries, E):
            raise ValueError(f"text features have shape {tuple(tf.shape)}, expected ({self.num_queries}, {E})")
        self.query_features = tf.to(dev)
        self.dtype = self.query_features.dtype
        self.ignore_residual = ignore_residual
        self.logit_scale, self.prob_thd = logit_scale, prob_thd
        self.slide_stride, self.slide_crop = slide_stride, slide_crop
        self.upsampler = None
        if apply_sim_feat_up:
            from .upsampler import HipJBU
            self.feat_dim = E
            cfgu = sim_feat_up_cfg or {}
            self.upsampler = HipJBU.from_config(cfgu.get("model_name", "jbu_one"), E, cfgu.get("model_path"), dev,
                                                synthetic_ok=synthetic_ok, precision=precision)
        # tiles per tower launch: None = the machine-filling size for this tower and tile (about ten rounds of 256 x 256 GEMM tiles on the 256 CUs:
        # 119 tiles of 512 for ViT-L/14, 95 for ViT-H/14, capped at 128; measured 228 Mpix/s at 119 tiles against 213 at 32)
        if tiles_per_launch is None:
            vc = visual.cfg
            crop = slide_crop if isinstance(slide_crop, int) and slide_crop > 0 else 512
            n_tok = (-(-crop // vc.patch)) ** 2 + 1
            tiles_per_launch = max(1, min(128, (2560 // max(1, -(-vc.width // 256))) * 256 // n_tok))
        self._tiles_per_launch = tiles_per_launch
        self._pipe = None
        return visual

    # -- pipeline object (rebuilt lazily so attribute edits after construction are honoured) ---------------------
    def _pipeline(self) -> SegPipeline:
        return SegPipeline(self.net, self.query_features, self.query_idx, model_type=self.model_type,
                           ignore_residual=self.ignore_residual, cls_token_lambda=self.cls_token_lambda,
                           global_debias_factor=self.global_debias_factor, logit_scale=self.logit_scale, prob_thd=self.prob_thd,
                           bg_idx=self.bg_idx, apply_similarity_enhancement=getattr(self, "apply_similarity_enhancement", False),
                           upsampler=self.upsampler, tiles_per_launch=self._tiles_per_launch,
                           cross_tile_fusion=getattr(self, "cross_tile_fusion_cfg", None), apply_ctd=getattr(self, "apply_ctd", False),
                           tile_group=getattr(self, "tile_group", None), apply_layer_fusion=getattr(self, "apply_layer_fusion", False),
                           layer_fusion_lambda=getattr(self, "layer_fusion_lambda", 0.5))

    def forward_feature(self, img, logit_size=None, tile_h_idx=None, tile_w_idx=None):
        """Reference segmentor.py:286-392.  img [B,3,H,W] -> logits [B,Q,h,w]."""
        if type(img) == list:
            img = img[0]
        return self._pipeline().forward_feature(img.float(), logit_size)

    def forward_slide(self, img, img_metas, stride=112, crop_size=224):
        """Reference segmentor.py:394-451.  One image [1,3,H,W] (or a list holding one [3,H,W])."""
        if type(img) == list:
            img = img[0].unsqueeze(0)
        pipe = self._pipeline()
        outs = []
        for b in range(img.shape[0]):
            ori = img_metas[b]["ori_shape"][:2] if img_metas is not None else None
            outs.append(pipe.forward_slide(img[b].float(), stride, crop_size, ori))
        return torch.cat(outs, 0)

    @torch.no_grad()
    def predict(self, inputs, data_samples):
        """Reference segmentor.py:453-473."""
        if data_samples is not None:
            batch_img_metas = [data_sample.metainfo for data_sample in data_samples]
        else:
            batch_img_metas = [dict(ori_shape=inputs.shape[2:], img_shape=inputs.shape[2:], pad_shape=inputs.shape[2:],
                                    padding_size=[0, 0, 0, 0])] * inputs.shape[0]
        inputs = inputs.float()
        if self.slide_crop > 0:
            seg_logits = self.forward_slide(inputs, batch_img_metas, self.slide_stride, self.slide_crop)
        else:
            seg_logits = self.forward_feature(inputs, batch_img_metas[0]["ori_shape"])
        return self.postprocess_result(seg_logits, data_samples)

    def postprocess_result(self, seg_logits, data_samples):
        """Reference segmentor.py:475-499: class probabilities [K,H,W] + labels [1,H,W] per image."""
        pipe = self._pipeline()
        batch_size = seg_logits.shape[0]
        for i in range(batch_size):
            probs, seg_pred = pipe.postprocess(seg_logits[i].float())
            if data_samples is None:
                return seg_pred
            data_samples[i].set_data({"seg_logits": PixelData(**{"data": probs}), "pred_sem_seg": PixelData(**{"data": seg_pred})})
            if getattr(self, "result_dir", None) or getattr(self, "heatmap_dir", None):        # segmentor.py:501-531
                self._write_maps(probs, seg_pred, data_samples[i], i)
        return data_samples

    def _generate_palette(self, n):
        """Reference segmentor.py:568-578: deterministic HSV palette, the background class dimmed."""
        import colorsys
        palette = []
        for idx in range(n):
            r, g, b = colorsys.hsv_to_rgb((idx / max(1, n)) % 1.0, 0.75, 1.0 if idx != self.bg_idx else 0.2)
            palette.append([int(r * 255), int(g * 255), int(b * 255)])
        return np.array(palette, dtype=np.uint8)

    def _write_maps(self, probs, seg_pred, sample, i):
        """Colour mask -> result_dir/<stem>.png, confidence map -> heatmap_dir/<stem>.png.  The images are rendered on the device
        (sg_render_maps); only the PNG encoding is host work.  Without OpenCV the reference's own fall-back colour ramp is used."""
        from . import ops
        from PIL import Image
        meta = sample.metainfo if hasattr(sample, "metainfo") else {}
        stem = None
        for key in ("img_path", "ori_path", "filename", "ori_filename"):
            if key in meta and meta[key]:
                stem = os.path.splitext(os.path.basename(meta[key]))[0]
                break
        if stem is None:
            stem = f"sample_{i}"
        palette = getattr(self, "_palette_cache", None)
        if palette is None or len(palette) < self.num_classes:
            palette = self._palette_cache = self._generate_palette(self.num_classes)
        mask, heat = ops.render_maps(seg_pred, probs, torch.from_numpy(palette), want_mask=bool(self.result_dir), want_heat=bool(self.heatmap_dir))
        if self.result_dir:
            os.makedirs(self.result_dir, exist_ok=True)
            Image.fromarray(mask.cpu().numpy()).save(os.path.join(self.result_dir, f"{stem}.png"))
        if self.heatmap_dir:
            os.makedirs(self.heatmap_dir, exist_ok=True)
            Image.fromarray(heat.cpu().numpy()).save(os.path.join(self.heatmap_dir, f"{stem}.png"))

    def compute_padsize(self, H: int, W: int, patch_size: int):
        return compute_padsize(H, W, patch_size)

    # mmseg abstract hooks the reference leaves empty (segmentor.py:548-566)
    def _forward(self, data_samples=None):
        """ """

    def inference(self, img, batch_img_metas):
        """ """

    def encode_decode(self, inputs, batch_img_metas):
        """ """

    def extract_feat(self, inputs):
        """ """

    def loss(self, inputs, data_samples):
        """ """


class SegmentorEx(_HipSegmentorBase):
    """Drop-in of reference ``segmentor.SegmentorEx`` (constructor kwargs of segmentor.py:33-63)."""

    def __init__(self, clip_type, vit_type, model_type, name_path, device=torch.device("cuda"), ignore_residual=True, prob_thd=0.0,
                 logit_scale=50, slide_stride=112, slide_crop=224, cls_token_lambda=0.0, global_debias_factor=0.0, bg_idx=0,
                 apply_sim_feat_up=False, sim_feat_up_cfg=dict(model_name="jbu_one", model_path="your/model/path"),
                 apply_ctd=False, apply_outlier_suppression=False, outlier_suppression_cfg=None,
                 apply_self_attn_enhancement=False, self_attn_enhancement_cfg=None, apply_layer_fusion=False,
                 layer_fusion_lambda=0.5, layer_fusion_threshold=0.7, apply_similarity_enhancement=False,
                 similarity_enhancement_cfg=None, result_dir=None, heatmap_dir=None,
                 # -- drop-in extras (see module docstring) --
                 checkpoint=None, text_features=None, text_encoder=None, precision="f16x2", synthetic_ok=False, tiles_per_launch=None,
                 tokenizer=None, cross_tile_fusion_cfg=None, tile_group=None):
        data_preprocessor = SegDataPreProcessor(mean=list(Wt.PIXEL_MEAN), std=list(Wt.PIXEL_STD), bgr_to_rgb=True)
        super().__init__(data_preprocessor=data_preprocessor)
        if model_type == "GEM":
            raise ValueError("model_type='GEM' crashes in the reference SegmentorEx (it unpacks (cls, feats), SURVEY.md R5); "
                             "use segearth_segmentor.Segmentor with cls_token_lambda=0")
        visual = self._setup(clip_type, vit_type, model_type, name_path, device, ignore_residual, prob_thd, logit_scale, slide_stride,
                             slide_crop, cls_token_lambda, bg_idx, _to_bool(apply_sim_feat_up), sim_feat_up_cfg, global_debias_factor,
                             checkpoint, text_features, text_encoder, precision, synthetic_ok, tiles_per_launch, tokenizer=tokenizer,
                             tile_group=tile_group)
        # opt-in extra: kwargs of the reference's CrossTileFusion (cross_tile_fusion.py:24-60), which the reference never calls (R2)
        self.cross_tile_fusion_cfg = cross_tile_fusion_cfg
        self.apply_ctd = _to_bool(apply_ctd)                                # segmentor.py:184-194, 339-365: DBSCAN + cluster debias, on the device here
        # attention-map layer fusion (segmentor.py:189-193, 302-304): the reference's own code only runs it with one head;
        # built here with the semantics that case pins (oracle/vit.py).  layer_fusion_threshold is unused by the reference.
        self.apply_layer_fusion, self.layer_fusion_lambda, self.layer_fusion_threshold = _to_bool(apply_layer_fusion), layer_fusion_lambda, layer_fusion_threshold
        self.apply_similarity_enhancement = _to_bool(apply_similarity_enhancement)
        if self.apply_similarity_enhancement:                               # segmentor.py:196-220
            c = dict(similarity_weight=1.0, temperature=1.0, add_self_similarity=True)
            c.update(similarity_enhancement_cfg or {})
            visual.similarity_enhancer = SimilarityEnhancementModule(**c)
        self.apply_self_attn_enhancement = _to_bool(apply_self_attn_enhancement)
        if self.apply_self_attn_enhancement:                                # segmentor.py:222-249
            c = dict(enhancement_strength=0.1, min_self_attn_threshold=0.15, mode="feature")
            c.update(self_attn_enhancement_cfg or {})
            visual.self_attn_enhancer = SelfAttentionEnhancementModule(**c)
        self.apply_outlier_suppression = _to_bool(apply_outlier_suppression)
        if self.apply_outlier_suppression:                                  # segmentor.py:251-274
            c = dict(top_k=10)
            c.update(outlier_suppression_cfg or {})
            visual.outlier_suppressor = OutlierSuppressionModule(top_k=c["top_k"])
        self.result_dir, self.heatmap_dir = result_dir, heatmap_dir


class Segmentor(_HipSegmentorBase):
    """Drop-in of reference ``segearth_segmentor.Segmentor`` (segearth_segmentor.py:23-41): no refiners, no global
    debias; the CLS token is only requested when ``cls_token_lambda != 0``; GEM runs here."""

    def __init__(self, clip_type, vit_type, model_type, name_path, device=torch.device("cuda"), ignore_residual=True, prob_thd=0.0,
                 logit_scale=50, slide_stride=112, slide_crop=224, cls_token_lambda=0, bg_idx=0, apply_sim_feat_up=True,
                 sim_feat_up_cfg=dict(model_name="jbu_one", model_path="your/model/path"),
                 checkpoint=None, text_features=None, text_encoder=None, precision="f16x2", synthetic_ok=False, tiles_per_launch=None,
                 tokenizer=None, cross_tile_fusion_cfg=None, tile_group=None, apply_outlier_suppression=False, outlier_suppression_cfg=None):
        data_preprocessor = SegDataPreProcessor(mean=list(Wt.PIXEL_MEAN), std=list(Wt.PIXEL_STD), bgr_to_rgb=True)
        super().__init__(data_preprocessor=data_preprocessor)
        if model_type == "GEM" and cls_token_lambda != 0:
            raise ValueError("GEM returns no CLS token (gem_utils.py:198-199): cls_token_lambda must be 0 (SURVEY.md R5)")
        self._setup(clip_type, vit_type, model_type, name_path, device, ignore_residual, prob_thd, logit_scale, slide_stride,
                    slide_crop, cls_token_lambda, bg_idx, apply_sim_feat_up, sim_feat_up_cfg, 0.0,
                    checkpoint, text_features, text_encoder, precision, synthetic_ok, tiles_per_launch, tokenizer=tokenizer,
                    tile_group=tile_group)
        # opt-in extra: kwargs of the reference's CrossTileFusion (cross_tile_fusion.py:24-60), which the reference never calls (R2)
        self.cross_tile_fusion_cfg = cross_tile_fusion_cfg
        self.output_cls_token = cls_token_lambda != 0
        self.apply_similarity_enhancement = False
        # opt-in extra (BASELINE configs[2]: "GEM self-self attn + outlier_suppression" in ONE forward).  The reference cannot run that
        # composition (SegmentorEx crashes on GEM and this class has no refiners); the definition built here -- detection on the
        # ordinary stream's attention of block L-2, suppression on the GEM stream before ln_post -- is written down in DESIGN.md section 7 and
        # restated in oracle/vit.py::gem_forward.  Same kwargs as SegmentorEx (segmentor.py:251-274).
        self.apply_outlier_suppression = _to_bool(apply_outlier_suppression)
        if self.apply_outlier_suppression:
            c = dict(top_k=10)
            c.update(outlier_suppression_cfg or {})
            self.net.visual.outlier_suppressor = OutlierSuppressionModule(top_k=c["top_k"])
