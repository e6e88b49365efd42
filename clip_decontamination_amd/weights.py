"""Deterministic synthetic weights for the CLIP vision tower and the JBU upsampler.

No pretrained checkpoint can be fetched in this pipeline, so parity and
throughput are measured on weights produced here.  The generator is pure numpy (PCG64,
one stream per tensor name) so that this container (where the reference is imported as the
oracle) and the GPU box (where it is absent) regenerate bit-identical tensors.

Tensor names and shapes follow the ``visual.*`` state dict of the reference's vendored
open_clip ``VisionTransformer`` (reference open_clip/transformer.py:372-442) so the same
dict can be ``load_state_dict``-ed into the reference when minting golden fixtures.
Scales follow the commented-out init in reference open_clip/transformer.py:508-518.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, asdict
from typing import Dict

import numpy as np


@dataclass(frozen=True)
class VitConfig:
    """Architecture hyper-parameters (reference open_clip/model_configs/*.json)."""
    name: str
    width: int          # D
    layers: int         # L
    heads: int          # H
    patch: int          # P
    embed_dim: int      # E
    image_size: int = 224
    mlp_ratio: float = 4.0
    quick_gelu: bool = True   # OpenAI / MetaCLIP weights (reference model.py:116,518)

    @property
    def grid0(self) -> int:
        return self.image_size // self.patch

    @property
    def head_dim(self) -> int:
        return self.width // self.heads

    @property
    def mlp_width(self) -> int:
        return int(self.width * self.mlp_ratio)

    def to_dict(self):
        return asdict(self)


VIT_CONFIGS: Dict[str, VitConfig] = {
    # reference open_clip/model_configs/ViT-B-16.json, ViT-L-14.json, ViT-H-14.json, ViT-B-32.json
    "ViT-B-16": VitConfig("ViT-B-16", 768, 12, 12, 16, 512),
    "ViT-B-32": VitConfig("ViT-B-32", 768, 12, 12, 32, 512),
    "ViT-L-14": VitConfig("ViT-L-14", 1024, 24, 16, 14, 768),
    "ViT-H-14": VitConfig("ViT-H-14", 1280, 32, 16, 14, 1024, quick_gelu=False),
    # tiny shapes used by the golden fixtures and CPU-side tests
    "tiny-8": VitConfig("tiny-8", 64, 4, 2, 8, 32, image_size=32),
    "tiny-16": VitConfig("tiny-16", 128, 5, 2, 16, 64, image_size=64),
    "tiny-gem": VitConfig("tiny-gem", 64, 8, 2, 8, 32, image_size=32),     # >= 7 layers for GEM depth 7
    "tiny-gelu": VitConfig("tiny-gelu", 64, 4, 2, 8, 32, image_size=32, quick_gelu=False),
    # ONE head: the only head count at which the reference's apply_layer_fusion + outlier suppressor path runs (R9) -- pins its semantics
    "tiny-1h": VitConfig("tiny-1h", 64, 4, 1, 8, 32, image_size=32),
}


def vit_config(vit_type: str, quick_gelu: bool | None = None) -> VitConfig:
    """Resolve the loose ``vit_type`` strings the reference segmentors accept
    (reference segmentor.py:69-104: ``'B' in vit_type`` / ``'L' in vit_type`` / ``'H' in vit_type``)."""
    key = vit_type.replace("/", "-")
    if key in VIT_CONFIGS:
        cfg = VIT_CONFIGS[key]
    elif "tiny" in key:
        cfg = VIT_CONFIGS[key]
    elif "B" in key:
        cfg = VIT_CONFIGS["ViT-B-32" if "32" in key else "ViT-B-16"]
    elif "L" in key:
        cfg = VIT_CONFIGS["ViT-L-14"]
    elif "H" in key:
        cfg = VIT_CONFIGS["ViT-H-14"]
    else:
        raise ValueError(f"unknown vit_type {vit_type!r}")
    if quick_gelu is not None and quick_gelu != cfg.quick_gelu:
        cfg = VitConfig(**{**cfg.to_dict(), "quick_gelu": quick_gelu})
    return cfg


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))


def _normal(seed, key, shape, std, mean=0.0):
    a = _rng(seed, key).standard_normal(shape, dtype=np.float32)
    a *= np.float32(std)
    if mean:
        a += np.float32(mean)
    return a


def make_vit_weights(cfg: VitConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """fp32 numpy state dict of the vision tower, keys as in ``net.visual.state_dict()``."""
    D, L, P, E = cfg.width, cfg.layers, cfg.patch, cfg.embed_dim
    M = cfg.mlp_width
    scale = D ** -0.5
    proj_std = (D ** -0.5) * ((2 * L) ** -0.5)
    attn_std = D ** -0.5
    fc_std = (2 * D) ** -0.5
    w: Dict[str, np.ndarray] = {}
    w["conv1.weight"] = _normal(seed, "conv1.weight", (D, 3, P, P), (3 * P * P) ** -0.5)
    w["class_embedding"] = _normal(seed, "class_embedding", (D,), scale)
    w["positional_embedding"] = _normal(seed, "positional_embedding", (cfg.grid0 ** 2 + 1, D), scale)
    for ln in ("ln_pre", "ln_post"):
        w[f"{ln}.weight"] = _normal(seed, f"{ln}.weight", (D,), 0.1, 1.0)
        w[f"{ln}.bias"] = _normal(seed, f"{ln}.bias", (D,), 0.05)
    for i in range(L):
        p = f"transformer.resblocks.{i}."
        for ln in ("ln_1", "ln_2"):
            w[p + f"{ln}.weight"] = _normal(seed, p + f"{ln}.weight", (D,), 0.1, 1.0)
            w[p + f"{ln}.bias"] = _normal(seed, p + f"{ln}.bias", (D,), 0.05)
        w[p + "attn.in_proj_weight"] = _normal(seed, p + "attn.in_proj_weight", (3 * D, D), attn_std)
        w[p + "attn.in_proj_bias"] = _normal(seed, p + "attn.in_proj_bias", (3 * D,), 0.02)
        w[p + "attn.out_proj.weight"] = _normal(seed, p + "attn.out_proj.weight", (D, D), proj_std)
        w[p + "attn.out_proj.bias"] = _normal(seed, p + "attn.out_proj.bias", (D,), 0.02)
        w[p + "mlp.c_fc.weight"] = _normal(seed, p + "mlp.c_fc.weight", (M, D), fc_std)
        w[p + "mlp.c_fc.bias"] = _normal(seed, p + "mlp.c_fc.bias", (M,), 0.02)
        w[p + "mlp.c_proj.weight"] = _normal(seed, p + "mlp.c_proj.weight", (D, M), proj_std)
        w[p + "mlp.c_proj.bias"] = _normal(seed, p + "mlp.c_proj.bias", (D,), 0.02)
    w["proj"] = _normal(seed, "proj", (D, E), scale)
    return w


@dataclass(frozen=True)
class TextConfig:
    """Text-tower hyper-parameters (reference open_clip/model.py CLIPTextCfg; model_configs/*.json "text_cfg")."""
    name: str
    width: int
    layers: int
    heads: int
    embed_dim: int
    context_length: int = 77
    vocab_size: int = 49408
    quick_gelu: bool = True


TEXT_CONFIGS: Dict[str, TextConfig] = {
    "ViT-B-16": TextConfig("ViT-B-16", 512, 12, 8, 512),
    "ViT-B-32": TextConfig("ViT-B-32", 512, 12, 8, 512),
    "ViT-L-14": TextConfig("ViT-L-14", 768, 12, 12, 768),
    "ViT-H-14": TextConfig("ViT-H-14", 1024, 24, 16, 1024, quick_gelu=False),
    "tiny-text": TextConfig("tiny-text", 64, 3, 2, 32, context_length=12, vocab_size=100),
    "tiny-8": TextConfig("tiny-8", 64, 3, 2, 32, context_length=12, vocab_size=100),          # text half paired with the tiny-8 ViT
    "tiny-text-gelu": TextConfig("tiny-text-gelu", 64, 2, 2, 32, context_length=77, vocab_size=100, quick_gelu=False),
}


def make_text_weights(cfg: TextConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """fp32 numpy state dict of the text tower, keys as the text part of ``CLIP.state_dict()``."""
    D, L, E = cfg.width, cfg.layers, cfg.embed_dim
    M = 4 * D
    proj_std = (D ** -0.5) * ((2 * L) ** -0.5)
    w: Dict[str, np.ndarray] = {}
    w["token_embedding.weight"] = _normal(seed, "t.token_embedding.weight", (cfg.vocab_size, D), 0.3)
    w["positional_embedding"] = _normal(seed, "t.positional_embedding", (cfg.context_length, D), 0.1)
    for i in range(L):
        p = f"transformer.resblocks.{i}."
        for ln in ("ln_1", "ln_2"):
            w[p + f"{ln}.weight"] = _normal(seed, "t." + p + f"{ln}.weight", (D,), 0.1, 1.0)
            w[p + f"{ln}.bias"] = _normal(seed, "t." + p + f"{ln}.bias", (D,), 0.05)
        w[p + "attn.in_proj_weight"] = _normal(seed, "t." + p + "attn.in_proj_weight", (3 * D, D), D ** -0.5)
        w[p + "attn.in_proj_bias"] = _normal(seed, "t." + p + "attn.in_proj_bias", (3 * D,), 0.02)
        w[p + "attn.out_proj.weight"] = _normal(seed, "t." + p + "attn.out_proj.weight", (D, D), proj_std)
        w[p + "attn.out_proj.bias"] = _normal(seed, "t." + p + "attn.out_proj.bias", (D,), 0.02)
        w[p + "mlp.c_fc.weight"] = _normal(seed, "t." + p + "mlp.c_fc.weight", (M, D), (2 * D) ** -0.5)
        w[p + "mlp.c_fc.bias"] = _normal(seed, "t." + p + "mlp.c_fc.bias", (M,), 0.02)
        w[p + "mlp.c_proj.weight"] = _normal(seed, "t." + p + "mlp.c_proj.weight", (D, M), proj_std)
        w[p + "mlp.c_proj.bias"] = _normal(seed, "t." + p + "mlp.c_proj.bias", (D,), 0.02)
    w["ln_final.weight"] = _normal(seed, "t.ln_final.weight", (D,), 0.1, 1.0)
    w["ln_final.bias"] = _normal(seed, "t.ln_final.bias", (D,), 0.05)
    w["text_projection"] = _normal(seed, "t.text_projection", (D, E), D ** -0.5)
    return w


def make_token_ids(cfg: TextConfig, n_seq: int, seed: int = 11) -> np.ndarray:
    """[n_seq, context_length] int32 rows shaped like the CLIP tokenizer's output: SOT, body, EOT (= the largest id), zero padding."""
    g = _rng(seed, f"tokens.{cfg.name}.{n_seq}")
    ids = np.zeros((n_seq, cfg.context_length), np.int32)
    sot, eot = cfg.vocab_size - 2, cfg.vocab_size - 1
    for i in range(n_seq):
        n_body = int(g.integers(1, cfg.context_length - 1))
        ids[i, 0] = sot
        ids[i, 1:1 + n_body] = g.integers(1, cfg.vocab_size - 2, n_body)
        ids[i, 1 + n_body] = eot
    return ids


def make_text_features(num_queries: int, embed_dim: int, seed: int = 7) -> np.ndarray:
    """Unit-norm rows standing in for the prompt-ensembled text embeddings
    (reference segmentor.py:157-174 produces ``query_features [Q, E]``)."""
    t = _normal(seed, f"text.{num_queries}.{embed_dim}", (num_queries, embed_dim), 1.0)
    t /= np.linalg.norm(t, axis=-1, keepdims=True)
    return t.astype(np.float32)


def make_jbu_weights(model_name: str, feat_dim: int, seed: int = 3) -> Dict[str, np.ndarray]:
    """fp32 state dict of ``JBUOne`` / ``JBUStack`` (reference simfeatup_dev/upsamplers.py:202-325).

    Key names match ``get_upsampler(name, dim).state_dict()`` of the reference so the dict
    can be loaded into it with ``strict=True`` when minting fixtures.
    """
    if model_name == "jbu_one":
        ups = {"up": 5}
    elif model_name == "jbu_stack":
        ups = {f"up{i}": 3 for i in range(1, 5)}
    else:
        raise ValueError(f"Unknown upsampler {model_name}")
    key_dim, gdim = 32, 3
    w: Dict[str, np.ndarray] = {}
    for name, r in ups.items():
        d2 = (2 * r + 1) ** 2
        p = name + "."
        w[p + "range_temp"] = np.asarray(0.3, dtype=np.float32)
        w[p + "sigma_spatial"] = np.asarray(0.8, dtype=np.float32)
        w[p + "range_proj.0.weight"] = _normal(seed, p + "range_proj.0.weight", (key_dim, gdim, 1, 1), 0.6)
        w[p + "range_proj.0.bias"] = _normal(seed, p + "range_proj.0.bias", (key_dim,), 0.1)
        w[p + "range_proj.3.weight"] = _normal(seed, p + "range_proj.3.weight", (key_dim, key_dim, 1, 1), key_dim ** -0.5)
        w[p + "range_proj.3.bias"] = _normal(seed, p + "range_proj.3.bias", (key_dim,), 0.1)
        w[p + "fixup_proj.0.weight"] = _normal(seed, p + "fixup_proj.0.weight", (d2, d2 + gdim, 1, 1), (d2 + gdim) ** -0.5)
        w[p + "fixup_proj.0.bias"] = _normal(seed, p + "fixup_proj.0.bias", (d2,), 0.05)
        w[p + "fixup_proj.3.weight"] = _normal(seed, p + "fixup_proj.3.weight", (d2, d2, 1, 1), d2 ** -0.5)
        w[p + "fixup_proj.3.bias"] = _normal(seed, p + "fixup_proj.3.bias", (d2,), 0.05)
    w["fixup_proj.1.weight"] = _normal(seed, "fixup_proj.1.weight", (feat_dim, feat_dim, 1, 1), feat_dim ** -0.5)
    w["fixup_proj.1.bias"] = _normal(seed, "fixup_proj.1.bias", (feat_dim,), 0.05)
    return w


def make_tiles_u8(num_tiles: int, size: int = 512, seed: int = 1234, smooth: bool = False) -> np.ndarray:
    """Synthetic uint8 NHWC tiles.  ``smooth`` gives low-frequency content
    (8x8 noise upsampled) so that JBU guidance and outlier statistics are non-degenerate."""
    rng = np.random.default_rng(seed)
    if not smooth:
        return rng.integers(0, 256, size=(num_tiles, size, size, 3), dtype=np.uint8)
    low = rng.random((num_tiles, 9, 9, 3), dtype=np.float32)
    ys = np.linspace(0, 8, size, dtype=np.float32)
    y0 = np.clip(np.floor(ys).astype(np.int64), 0, 7)
    fy = (ys - y0)[None, :, None, None]
    rows = low[:, y0] * (1 - fy) + low[:, y0 + 1] * fy
    fx = (ys - y0)[None, None, :, None]
    img = rows[:, :, y0] * (1 - fx) + rows[:, :, y0 + 1] * fx
    img = img * 200.0 + rng.random(img.shape, dtype=np.float32) * 55.0
    return np.clip(img, 0, 255).astype(np.uint8)


# SegDataPreProcessor constants (reference segmentor.py:64-67); applied to RGB on the 0-255 scale.
PIXEL_MEAN = (122.771, 116.746, 104.094)
PIXEL_STD = (68.501, 66.632, 70.323)


def normalize_tiles(tiles_u8_nhwc: np.ndarray) -> np.ndarray:
    """uint8 NHWC RGB -> float32 NCHW normalised, the tensor ``predict`` receives."""
    x = tiles_u8_nhwc.astype(np.float32)
    x = (x - np.asarray(PIXEL_MEAN, np.float32)) / np.asarray(PIXEL_STD, np.float32)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))
