"""Host side of the SimFeatUp JBU upsampler: owns an ``sg_jbu`` context (reference
simfeatup_dev/upsamplers.py: ``get_upsampler('jbu_one' | 'jbu_stack', dim)`` + ``load_state_dict``) and produces
per-pixel class logits for a batch of tiles.  Arithmetic happens in libsegearth_hip.so."""
from __future__ import annotations

import ctypes as C
import os
import warnings
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, ops, weights as Wt
from ._lib import TileBatch, check
from .ops import precision_id, ptr, stream_ptr

KINDS = {"jbu_one": 0, "jbu_stack": 1}


def get_upsampler(upsampler: str, dim: int, device="cuda:0", precision="f32") -> "HipJBU":
    """Same factory name as the reference (upsamplers.py:353-369); only the JBU variants ride the HIP path."""
    if upsampler not in KINDS:
        raise ValueError(f"Unknown upsampler {upsampler}" if upsampler not in ("bilinear", "resize_conv", "carafe", "sapa", "ifa")
                         else f"upsampler {upsampler!r} is outside the HIP hot path (needs sapa / mmcv native ops; no config uses it)")
    return HipJBU(upsampler, dim, device, precision)


class HipJBU:
    def __init__(self, model_name: str, feat_dim: int, device="cuda:0", precision="f32"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("HipJBU needs a GPU: the hot path has no CPU implementation")
        self.model_name, self.feat_dim = model_name, feat_dim
        self.device = torch.device(device)
        # the upsampler has a parity (f32) and a throughput (bf16) path: fp8 / f16 towers feed the bf16 one; the exact two-plane f16 tower gets the
        # f32 kernels with their three linears on the two-plane f16 GEMM (f32-grade results, SG_PREC_F16X2)
        pid = precision_id(precision)
        self.precision = pid if pid in (_lib.PREC_F32, _lib.PREC_F16X2) else _lib.PREC_BF16
        self._ctx = C.c_void_p()
        self._ws = {}
        self.tiles_per_launch = 8            # tiles per JBU launch (workspace ~2.2 GB per 512-pixel tile)
        self.fused_tail = True               # bf16 mode: sg_jbu_logits (no [S^2, C] feature map); False = upsample + cosine_logits
        with torch.cuda.device(self.device):
            check(self.lib.sg_jbu_create(C.byref(self._ctx), self.device.index or 0, KINDS[model_name], feat_dim), "sg_jbu_create")

    def __del__(self):
        ctx = getattr(self, "_ctx", None)
        if ctx is not None and ctx.value:
            self.lib.sg_jbu_destroy(ctx)
            self._ctx = C.c_void_p()

    @classmethod
    def from_config(cls, model_name, feat_dim, model_path, device, synthetic_ok=False, precision="f32"):
        up = cls(model_name, feat_dim, device, precision)
        if model_path and os.path.exists(model_path):
            ckpt = torch.load(model_path, map_location="cpu", weights_only=True)["state_dict"]
            up.load_state_dict({k[10:]: v for k, v in ckpt.items()})       # strips 'upsampler.' (segmentor.py:282)
        elif synthetic_ok:
            warnings.warn(f"SimFeatUp: checkpoint {model_path!r} not found, using SYNTHETIC upsampler weights")
            up.load_state_dict(Wt.make_jbu_weights(model_name, feat_dim, seed=3))
        else:
            raise FileNotFoundError(f"SimFeatUp checkpoint {model_path!r} not found (the configured jbu_one blob is not in the tree)")
        return up

    def load_state_dict(self, state_dict: Dict[str, "np.ndarray | torch.Tensor"], strict: bool = True):
        with torch.cuda.device(self.device):
            s = stream_ptr(self.device)
            for name, value in state_dict.items():
                t = torch.as_tensor(value).detach().to(device=self.device, dtype=torch.float32).contiguous().reshape(-1)
                check(self.lib.sg_jbu_set_tensor(self._ctx, name.encode(), ptr(t), t.numel(), s), f"sg_jbu_set_tensor({name})")
            torch.cuda.current_stream(self.device).synchronize()

    def _workspace(self, nbytes):
        """One arena per stream (as HipVisionTower._workspace): launches on different streams may overlap and must not share scratch."""
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes + 256:
            self._ws[key] = None
            ws = self._ws[key] = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        p = ws.data_ptr()
        off = (-p) % 256
        return C.c_void_p(p + off), ws.numel() - off

    def __call__(self, source: torch.Tensor, guidance: torch.Tensor) -> torch.Tensor:
        """Reference call shape: source [B,C,h,w], guidance [B,3,H,W] -> [B,C,16h,16w] (upsamplers.py:278-325)."""
        B, Cc, h, w = source.shape
        tok = source.permute(0, 2, 3, 1).reshape(B, h * w, Cc)
        out = self.upsample_tokens(tok, guidance, h, w)
        return out.reshape(B, 16 * h, 16 * w, Cc).permute(0, 3, 1, 2)

    def upsample_tokens(self, tokens: torch.Tensor, guidance: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
        """tokens [B, gh*gw, C] pixel-major, guidance [B,3,GH,GW] -> [B, 16gh*16gw, C]."""
        tokens = tokens.contiguous().float()
        guidance = guidance.contiguous().float()
        B, n, Cc = tokens.shape
        assert n == gh * gw and Cc == self.feat_dim
        with torch.cuda.device(self.device):
            out = torch.empty(B, 256 * n, Cc, dtype=torch.float32, device=self.device)
            need = self.lib.sg_jbu_workspace_bytes(self._ctx, B, gh, gw)
            wp, wn = self._workspace(need)
            check(self.lib.sg_jbu_upsample(self._ctx, ptr(tokens), ptr(guidance), B, gh, gw, guidance.shape[-2], guidance.shape[-1],
                                           self.precision, ptr(out), wp, wn, stream_ptr(self.device)), "sg_jbu_upsample")
        return out

    def logits(self, tokens, cls, scene, windows, tile_hw, pad_lt, grid, text, global_debias_factor, cls_token_lambda, scene_index=None,
               padded_hw=None):
        """Per-pixel logits [T, Q, 16gh, 16gw] for the tiles of one launch: global debias -> JBU -> cosine logits
        (reference order, segmentor.py:317-379; the reference runs B=1, :369-372).  ``tiles_per_launch`` tiles go through the
        upsampler per launch -- no per-tile host loop; the chunk bounds the workspace (C x S^2 f32 = 0.5 GB per 512-pixel tile)."""
        T = tokens.shape[0]
        gh, gw = grid
        th, tw = tile_hw
        l, t = pad_lt
        # the guidance is the normalised, zero-padded tile; JBU upsamples 16x regardless of the ViT's patch size
        ph, pw = padded_hw if padded_hw is not None else (th + 2 * t, tw + 2 * l)
        fmt = _lib.IMG_U8_NHWC if scene.dtype == torch.uint8 else _lib.IMG_F32_NCHW
        scene = scene.contiguous()
        H, W = (scene.shape[-3], scene.shape[-2]) if fmt == _lib.IMG_U8_NHWC else (scene.shape[-2], scene.shape[-1])
        Q = text.shape[0]
        with torch.cuda.device(self.device):
            if cls is not None and global_debias_factor != 0:
                tokens = ops.global_debias(tokens, cls, float(global_debias_factor))
            outs = []
            windows = windows.to(device=self.device, dtype=torch.int32).contiguous()
            if scene_index is not None:
                scene_index = scene_index.to(device=self.device, dtype=torch.int32).contiguous()
            step = max(1, int(self.tiles_per_launch))
            for i in range(0, T, step):
                c = min(step, T - i)
                tb = TileBatch()
                tb.scene, tb.format, tb.scene_h, tb.scene_w = scene.data_ptr(), fmt, H, W
                wi = windows[i:i + c].contiguous()
                tb.windows = wi.data_ptr()
                if scene_index is not None:
                    si = scene_index[i:i + c].contiguous()
                    tb.scene_index = si.data_ptr()
                    tb.scene_stride = 3 * H * W
                tb.n_tiles, tb.tile_h, tb.tile_w, tb.pad_l, tb.pad_t, tb.grid_h, tb.grid_w = c, th, tw, l, t, gh, gw
                guid = torch.empty(c, 3, ph, pw, dtype=torch.float32, device=self.device)
                check(self.lib.sg_extract_tiles(C.byref(tb), ph, pw, ptr(guid), stream_ptr(self.device)), "sg_extract_tiles")
                lam = float(cls_token_lambda) if cls is not None else 0.0
                if self.fused_tail and self.precision == _lib.PREC_BF16 and self.feat_dim % 64 == 0 and self.feat_dim >= 512 and Q <= 32:
                    # throughput mode: JBU + L2-norm + x T^T in one call; the [S^2, C] feature map never reaches HBM
                    lg = torch.empty(c, Q, 256 * gh * gw, dtype=torch.float32, device=self.device)
                    tk = tokens[i:i + c].contiguous().float()
                    ci = None if (cls is None or lam == 0.0) else cls[i:i + c].contiguous().float()
                    need = self.lib.sg_jbu_workspace_bytes(self._ctx, c, gh, gw)
                    wp, wn = self._workspace(need)
                    check(self.lib.sg_jbu_logits(self._ctx, ptr(tk), ptr(guid), c, gh, gw, ph, pw, self.precision, ptr(text), Q, ptr(ci), lam,
                                                 ptr(lg), wp, wn, stream_ptr(self.device)), "sg_jbu_logits")
                else:
                    feats = self.upsample_tokens(tokens[i:i + c], guid, gh, gw)                      # [c, 256 n, C]
                    lg = ops.cosine_logits(feats, None if cls is None else cls[i:i + c], text, 0.0, lam, two_plane=self.precision == _lib.PREC_F16X2)
                outs.append(lg.reshape(c, Q, 16 * gh, 16 * gw))
        return torch.cat(outs, 0) if len(outs) > 1 else outs[0]
