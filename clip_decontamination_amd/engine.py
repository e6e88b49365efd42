"""Host side of the vision tower: owns a libsegearth_hip context and mirrors the reference's seams

    net.encode_image(img, model_type, ignore_residual, output_cls_token=True, ...)   (open_clip/model.py:265-286)
    net.visual(img)                                                                  (GEM, gem/gem_utils.py:159-199)
    net.visual.{similarity_enhancer, outlier_suppressor, self_attn_enhancer}         (segmentor.py:216,245,270)

so that the drop-in segmentors read like the reference's.  All arithmetic happens in the HIP library.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import ForwardOpts, MODEL_TYPES, TileBatch, VitDesc, check
from .ops import precision_id, ptr, stream_ptr
from .weights import TextConfig, VitConfig


def compute_padsize(H: int, W: int, patch_size: int) -> Tuple[int, int, int, int]:
    """Symmetric zero padding to a patch multiple -> (l, r, t, b).  Reference segmentor.py:534-546."""
    l = r = t = b = 0
    if W % patch_size:
        lr = patch_size - (W % patch_size)
        l = lr // 2
        r = lr - l
    if H % patch_size:
        tb = patch_size - (H % patch_size)
        t = tb // 2
        b = tb - t
    return l, r, t, b


# Refiner "modules": plain config holders with the reference modules' constructor signatures.
@dataclass
class SimilarityEnhancementModule:            # reference similarity_enhancement.py:16-35
    similarity_weight: float = 1.0
    temperature: float = 1.0
    add_self_similarity: bool = True


@dataclass
class OutlierSuppressionModule:               # reference outlier_suppression.py:64-81
    top_k: int = 10
    contamination_temp: float = 0.1


@dataclass
class SelfAttentionEnhancementModule:         # reference self_attention_enhancement.py:33-46
    enhancement_strength: float = 0.1
    min_self_attn_threshold: float = 0.15
    mode: str = "feature"
    top_k: int = 10

    def __post_init__(self):
        assert self.mode in ["feature", "attention"], f"Mode must be 'feature' or 'attention', got {self.mode}"


class HipVisionTower:
    """The ``net.visual`` of the drop-in: a HIP context + the refiner attributes the reference installs."""

    def __init__(self, cfg: VitConfig, state_dict: Dict[str, "np.ndarray | torch.Tensor"], precision="bf16",
                 device: "torch.device | str | int" = "cuda:0"):
        self.lib = _lib.load()                      # raises if the HIP extension is not built
        if not torch.cuda.is_available():
            raise RuntimeError("HipVisionTower needs a GPU: the hot path has no CPU implementation")
        self.cfg = cfg
        self.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if self.device.type != "cuda":
            raise RuntimeError(f"HipVisionTower needs a cuda (ROCm) device, got {self.device}")
        self.precision = precision_id(precision)
        self.patch_size = (cfg.patch, cfg.patch)
        self.output_dim = cfg.embed_dim
        self.similarity_enhancer: Optional[SimilarityEnhancementModule] = None
        self.outlier_suppressor: Optional[OutlierSuppressionModule] = None
        self.self_attn_enhancer: Optional[SelfAttentionEnhancementModule] = None
        self.gem_depth = 7
        self.gem_ignore_residual = True
        self._ws: Optional[torch.Tensor] = None
        desc = VitDesc(cfg.width, cfg.layers, cfg.heads, cfg.patch, cfg.embed_dim, cfg.grid0, cfg.mlp_width,
                       int(cfg.quick_gelu), self.precision)
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.sg_create(C.byref(self._ctx), self.device.index or 0, C.byref(desc)), "sg_create")
            self.load_state_dict(state_dict)

    def __del__(self):
        ctx = getattr(self, "_ctx", None)
        if ctx is not None and ctx.value:
            self.lib.sg_destroy(ctx)
            self._ctx = C.c_void_p()

    # -- weights -----------------------------------------------------------------------------------
    def load_state_dict(self, state_dict) -> None:
        """Keys as in the reference's ``net.visual.state_dict()`` (a ``visual.`` prefix is accepted)."""
        with torch.cuda.device(self.device):
            s = stream_ptr(self.device)
            for name, value in state_dict.items():
                if name.startswith("visual."):
                    name = name[len("visual."):]
                t = torch.as_tensor(value) if not torch.is_tensor(value) else value
                t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
                check(self.lib.sg_vit_set_tensor(self._ctx, name.encode(), ptr(t), t.numel(), s), f"sg_vit_set_tensor({name})")
            torch.cuda.current_stream(self.device).synchronize()     # staging tensors may now be freed
            check(self.lib.sg_vit_finalize(self._ctx, s), "sg_vit_finalize")

    # -- options --------------------------------------------------------------------------------------
    def forward_opts(self, model_type: str, ignore_residual: bool = True, apply_similarity_enhancement: bool = True,
                     apply_layer_fusion: bool = False, layer_fusion_lambda: float = 0.5) -> ForwardOpts:
        if model_type not in MODEL_TYPES:
            raise ValueError(f"unknown model_type {model_type!r}")
        o = ForwardOpts()
        # attention-map layer fusion (transformer.py:598-607,630-637,647-690; semantics pinned by the one-head reference fixture,
        # oracle/vit.py): not defined for GEM (its forward is replaced wholesale, gem_utils.py:159-199)
        o.layer_fusion_enabled = int(bool(apply_layer_fusion) and model_type != "GEM")
        o.layer_fusion_lambda = float(layer_fusion_lambda)
        o.model_type = MODEL_TYPES[model_type]
        o.ignore_residual = int(bool(ignore_residual))
        o.gem_depth = self.gem_depth
        se = self.similarity_enhancer if apply_similarity_enhancement else None
        # the reference only captures mid-layer features when apply_similarity_enhancement is passed (transformer.py:594)
        if se is not None:
            o.similarity_enabled, o.similarity_weight = 1, float(se.similarity_weight)
            o.similarity_temperature, o.similarity_add_self = float(se.temperature), int(bool(se.add_self_similarity))
        else:
            o.similarity_weight, o.similarity_temperature, o.similarity_add_self = 1.0, 1.0, 1
        if self.outlier_suppressor is not None:
            o.outlier_enabled, o.outlier_top_k = 1, int(self.outlier_suppressor.top_k)
            o.outlier_contamination_temp = float(self.outlier_suppressor.contamination_temp)
        if self.self_attn_enhancer is not None:
            sa = self.self_attn_enhancer
            o.selfattn_enabled, o.selfattn_mode, o.selfattn_top_k = 1, int(sa.mode == "attention"), int(sa.top_k)
            o.selfattn_strength, o.selfattn_threshold = float(sa.enhancement_strength), float(sa.min_self_attn_threshold)
        if model_type == "GEM":
            o.ignore_residual = int(bool(self.gem_ignore_residual))
            # GEM's replaced forward (gem_utils.py:159-199) has no hook positions: of the refiners only the outlier suppressor has a
            # defined place in it here (BASELINE configs[2] as one forward) -- the others stay off, as in the reference
            o.similarity_enabled = o.selfattn_enabled = o.layer_fusion_enabled = 0
        return o

    def _workspace(self, nbytes: int):
        """One arena per stream: forwards issued on different streams may overlap on the device and must not share scratch."""
        if self._ws is None:
            self._ws = {}
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes + 256:
            self._ws[key] = None
            ws = self._ws[key] = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        p = ws.data_ptr()
        off = (-p) % 256
        return C.c_void_p(p + off), ws.numel() - off

    # -- the tower ------------------------------------------------------------------------------------
    def forward_tiles(self, scene: torch.Tensor, windows: torch.Tensor, tile_hw: Tuple[int, int], opts: ForwardOpts,
                      scene_index: Optional[torch.Tensor] = None):
        """scene: f32 [3,H,W] / [B,3,H,W] normalised planes, or u8 [H,W,3] / [B,H,W,3] raw RGB.
        windows: int32 [T,4] (y1,y2,x1,x2) on the device.  Returns (cls [T,E] or None, tokens [T,gh*gw,E])."""
        if scene.is_cuda and scene.device != self.device:
            raise RuntimeError(f"scene lives on {scene.device}, the tower on {self.device}")
        with torch.cuda.device(self.device):          # the tower's device, whatever the caller's current device is
            return self._forward_tiles(scene, windows, tile_hw, opts, scene_index)

    def _forward_tiles(self, scene, windows, tile_hw, opts, scene_index):
        if scene.dtype == torch.uint8:
            fmt = _lib.IMG_U8_NHWC
            H, W = scene.shape[-3], scene.shape[-2]
            batched = scene.dim() == 4
        else:
            fmt = _lib.IMG_F32_NCHW
            scene = scene.float()
            H, W = scene.shape[-2], scene.shape[-1]
            batched = scene.dim() == 4
        scene = scene.contiguous()
        if not scene.is_cuda:
            raise RuntimeError("scene must live on the GPU")
        th, tw = tile_hw
        P = self.cfg.patch
        l, r, t, b = compute_padsize(th, tw, P)
        gh, gw = (th + t + b) // P, (tw + l + r) // P
        T = windows.shape[0]
        windows = windows.to(device=self.device, dtype=torch.int32).contiguous()
        tb = TileBatch()
        tb.scene, tb.format, tb.scene_h, tb.scene_w = scene.data_ptr(), fmt, H, W
        tb.windows = windows.data_ptr()
        if scene_index is not None:
            scene_index = scene_index.to(device=self.device, dtype=torch.int32).contiguous()
            tb.scene_index = scene_index.data_ptr()
        tb.scene_stride = 3 * H * W if batched else 0
        tb.n_tiles, tb.tile_h, tb.tile_w, tb.pad_l, tb.pad_t, tb.grid_h, tb.grid_w = T, th, tw, l, t, gh, gw
        E = self.cfg.embed_dim
        gem = opts.model_type == MODEL_TYPES["GEM"]
        cls = None if gem else torch.empty(T, E, dtype=torch.float32, device=self.device)
        tokens = torch.empty(T, gh * gw, E, dtype=torch.float32, device=self.device)
        need = self.lib.sg_vit_workspace_bytes(self._ctx, T, gh, gw, C.byref(opts))
        wp, wn = self._workspace(need)
        check(self.lib.sg_vit_forward(self._ctx, C.byref(tb), C.byref(opts), ptr(cls), ptr(tokens), wp, wn, stream_ptr(self.device)),
              "sg_vit_forward")
        return cls, tokens

    def _whole_image_windows(self, img: torch.Tensor):
        B, _, H, W = img.shape
        win = torch.tensor([[0, H, 0, W]] * B, dtype=torch.int32, device=self.device)
        idx = torch.arange(B, dtype=torch.int32, device=self.device)
        return win, idx

    def __call__(self, img: torch.Tensor):
        """GEM entry point, the reference's ``self.net.visual(img)``: patch tokens [B,n,E], no CLS (R5)."""
        win, idx = self._whole_image_windows(img)
        _, tok = self.forward_tiles(img, win, tuple(img.shape[-2:]), self.forward_opts("GEM"), idx)
        return tok


class HipTextTower:
    """The text half of ``CLIP`` (open_clip/model.py:288-306): a ``sg_text`` context; ``encode_text(ids)`` runs in HIP.

    Init-time producer of ``query_features`` (segmentor.py:157-174).  Token ids come from the caller's tokenizer (the BPE
    vocabulary is a downloaded asset, out of scope); everything after the ids is computed here."""

    def __init__(self, cfg: TextConfig, state_dict, precision="f32", device: "torch.device | str | int" = "cuda:0"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("HipTextTower needs a GPU: there is no CPU implementation")
        self.cfg = cfg
        self.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        self.precision = precision_id(precision)
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.sg_text_create(C.byref(self._ctx), self.device.index or 0, cfg.width, cfg.layers, cfg.heads, cfg.context_length,
                                          cfg.vocab_size, cfg.embed_dim, int(cfg.quick_gelu), self.precision), "sg_text_create")
            s = stream_ptr(self.device)
            wanted = ("token_embedding.", "positional_embedding", "transformer.", "ln_final.", "text_projection")
            for name, value in state_dict.items():
                if name.startswith("visual.") or not name.startswith(wanted):
                    continue                                             # full CLIP state dicts carry visual.* and logit_scale too
                t = torch.as_tensor(value) if not torch.is_tensor(value) else value
                t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
                check(self.lib.sg_text_set_tensor(self._ctx, name.encode(), ptr(t), t.numel(), s), f"sg_text_set_tensor({name})")
            torch.cuda.current_stream(self.device).synchronize()

    def __del__(self):
        ctx = getattr(self, "_ctx", None)
        if ctx is not None and ctx.value:
            self.lib.sg_text_destroy(ctx)
            self._ctx = C.c_void_p()

    def encode_text(self, text: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """text: int [S, context_length] token ids -> [S, E] f32."""
        if text.dim() != 2 or text.shape[1] != self.cfg.context_length:
            raise ValueError(f"token ids must be [S, {self.cfg.context_length}], got {tuple(text.shape)}")
        if text.numel() and (int(text.min()) < 0 or int(text.max()) >= self.cfg.vocab_size):
            raise IndexError(f"token id outside [0, {self.cfg.vocab_size}) (the reference's nn.Embedding raises too)")
        ids = text.to(device=self.device, dtype=torch.int32).contiguous()
        S = ids.shape[0]
        out = torch.empty(S, self.cfg.embed_dim, dtype=torch.float32, device=self.device)
        if S == 0:
            return out
        with torch.cuda.device(self.device):
            need = self.lib.sg_text_workspace_bytes(self._ctx, S)
            ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            base = ws.data_ptr() + (-ws.data_ptr()) % 256
            check(self.lib.sg_text_encode(self._ctx, ptr(ids), S, ptr(out), C.c_void_p(base), need, stream_ptr(self.device)), "sg_text_encode")
        return torch.nn.functional.normalize(out, dim=-1) if normalize else out

    def query_features(self, tokenizer, query_words: Sequence[str]) -> torch.Tensor:
        """segmentor.py:157-174: 80-template prompt ensemble per query word, all words in ONE encode launch."""
        from .prompts import ensemble_prompts
        batches = [torch.as_tensor(tokenizer(ensemble_prompts(qw))) for qw in query_words]
        n = [b.shape[0] for b in batches]
        f = self.encode_text(torch.cat(batches, 0))
        f = f / f.norm(dim=-1, keepdim=True)
        rows = [c.mean(dim=0) for c in f.split(n, dim=0)]
        q = torch.stack(rows)
        return q / q.norm(dim=-1, keepdim=True)


class HipCLIP:
    """The ``net`` of the drop-in segmentors: ``.visual`` + ``.encode_image`` (+ ``.encode_text``) with the reference signatures."""

    def __init__(self, visual: HipVisionTower, text: Optional[HipTextTower] = None):
        self.visual = visual
        self.text = text

    def encode_text(self, text, normalize: bool = False):
        if self.text is None:
            raise RuntimeError("this HipCLIP was built without text-tower weights")
        return self.text.encode_text(text, normalize)

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def encode_image(self, image, model_type, ignore_residual: bool = False, output_cls_token: bool = False, normalize: bool = False,
                     apply_layer_fusion: bool = False, layer_fusion_lambda: float = 0.5, layer_fusion_threshold: float = 0.7,
                     apply_similarity_enhancement: bool = False):
        v = self.visual
        win, idx = v._whole_image_windows(image)
        opts = v.forward_opts(model_type, ignore_residual, apply_similarity_enhancement, apply_layer_fusion, layer_fusion_lambda)
        cls, tok = v.forward_tiles(image, win, tuple(image.shape[-2:]), opts, idx)
        if normalize:
            cls = torch.nn.functional.normalize(cls, dim=-1)
            tok = torch.nn.functional.normalize(tok, dim=-1)
        return (cls, tok) if output_cls_token else tok
