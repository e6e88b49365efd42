"""TEST INFRASTRUCTURE ONLY (the checker, never the product path).

CPU fp32 restatement of the segmentor-level hot path of the reference:
``forward_feature`` (segmentor.py:286-392 / segearth_segmentor.py:163-220),
``forward_slide`` (segmentor.py:394-451), ``postprocess_result`` (segmentor.py:475-499),
``compute_padsize`` (segmentor.py:534-546).  Pinned by tests/golden/segment_*.npz which were
produced by the reference's own ``SegmentorEx`` / ``Segmentor`` methods.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import jbu as jbu_oracle
from . import vit as vit_oracle


def compute_padsize(H: int, Wd: int, patch: int):
    l = r = t = b = 0
    if Wd % patch:
        lr = patch - (Wd % patch)
        l = lr // 2
        r = lr - l
    if H % patch:
        tb = patch - (H % patch)
        t = tb // 2
        b = tb - t
    return l, r, t, b


@dataclass
class SegOracle:
    """Holds what ``SegmentorEx.__init__`` would hold (weights, text features, knobs)."""
    cfg: object
    weights: Dict[str, torch.Tensor]
    text: torch.Tensor                      # query_features [Q,E], unit norm
    query_idx: torch.Tensor                 # int64 [Q] -> class id
    model_type: str = "SegEarth"
    ignore_residual: bool = True
    cls_token_lambda: float = 0.0
    global_debias_factor: float = 0.0
    similarity_cfg: Optional[dict] = None
    outlier_cfg: Optional[dict] = None
    self_attn_cfg: Optional[dict] = None
    jbu_weights: Optional[Dict[str, torch.Tensor]] = None
    logit_scale: float = 50.0
    prob_thd: float = 0.0
    bg_idx: int = 0
    slide_stride: int = 112
    slide_crop: int = 224
    gem_depth: int = 7
    segearth_variant: bool = False          # segearth_segmentor.Segmentor semantics (no debias/refiners)
    apply_ctd: bool = False                 # Cluster-Then-Debias (segmentor.py:339-365)

    @property
    def num_queries(self):
        return self.text.shape[0]

    # ---- segmentor.py:286-392 ---------------------------------------------------------------
    def forward_feature(self, img, logit_size=None):
        P = self.cfg.patch
        if self.model_type == "GEM":
            feats = vit_oracle.gem_forward(self.weights, self.cfg, img, self.ignore_residual, self.gem_depth,
                                           outlier_cfg=self.outlier_cfg)       # composition of BASELINE configs[2], defined in gem_forward
            cls = None
        else:
            cls, feats = vit_oracle.vit_forward(
                self.weights, self.cfg, img, self.model_type, self.ignore_residual,
                similarity_cfg=self.similarity_cfg, outlier_cfg=self.outlier_cfg, self_attn_cfg=self.self_attn_cfg)
        cls_logits = None
        if cls is not None:
            cls = cls / cls.norm(dim=-1, keepdim=True)
            cls_logits = cls @ self.text.T
        gh, gw = img.shape[-2] // P, img.shape[-1] // P
        H, Wd = img.shape[-2:]
        if self.global_debias_factor != 0 and cls is not None:                 # :322-336
            fn = feats / feats.norm(dim=-1, keepdim=True)
            cn = cls / cls.norm(dim=-1, keepdim=True)
            sim = (fn * cn.unsqueeze(1)).sum(-1)
            feats = feats - cls.unsqueeze(1) * (sim.unsqueeze(-1) * self.global_debias_factor)
        if self.apply_ctd and cls is not None:                                 # :339-365
            from oracle import ctd as ctd_oracle
            feats, _ = ctd_oracle.ctd_debias(feats, cls, eps=1.1, min_samples=11, factor=-1.5)
        if self.jbu_weights is not None:                                       # :368-372 (B=1)
            E = feats.shape[-1]
            src = feats.permute(0, 2, 1).reshape(1, E, gh, gw)
            up = jbu_oracle.jbu_forward(self.jbu_weights, src, img)
            # P=16: 16g == S, the reference's view(1, C, image_w*image_h).  P=14 (the reference cannot run it) uses the
            # build's documented definition -- JBU to 16g, logits there, bilinear resize to S below ("parity unpinned").
            gh, gw = up.shape[-2:]
            feats = up.reshape(1, E, gh * gw).permute(0, 2, 1)
        feats = feats / feats.norm(dim=-1, keepdim=True)
        logits = feats @ self.text.T
        if self.cls_token_lambda != 0 and cls_logits is not None:
            logits = logits + cls_logits.unsqueeze(1) * self.cls_token_lambda
        Q = logits.shape[-1]
        logits = logits.permute(0, 2, 1).reshape(-1, Q, gh, gw)
        size = img.shape[-2:] if logit_size is None else logit_size
        return F.interpolate(logits, size=size, mode="bilinear")

    # ---- segmentor.py:394-451 ---------------------------------------------------------------
    def tile_windows(self, H, Wd, stride=None, crop=None):
        s = self.slide_stride if stride is None else stride
        c = self.slide_crop if crop is None else crop
        hg = max(H - c + s - 1, 0) // s + 1
        wg = max(Wd - c + s - 1, 0) // s + 1
        wins = []
        for hi in range(hg):
            for wi in range(wg):
                y2 = min(hi * s + c, H)
                x2 = min(wi * s + c, Wd)
                wins.append((max(y2 - c, 0), y2, max(x2 - c, 0), x2))
        return wins

    def forward_slide(self, img, ori_shape=None, stride=None, crop=None):
        B, _, H, Wd = img.shape
        preds = img.new_zeros((B, self.num_queries, H, Wd))
        count = img.new_zeros((B, 1, H, Wd))
        for (y1, y2, x1, x2) in self.tile_windows(H, Wd, stride, crop):
            tile = img[:, :, y1:y2, x1:x2]
            th, tw = tile.shape[-2:]
            pad = compute_padsize(th, tw, self.cfg.patch)
            if any(pad):
                tile = F.pad(tile, pad)
            lg = self.forward_feature(tile)
            if any(pad):
                lg = lg[:, :, pad[2]:pad[2] + th, pad[0]:pad[0] + tw]
            preds[:, :, y1:y2, x1:x2] += lg
            count[:, :, y1:y2, x1:x2] += 1
        assert (count == 0).sum() == 0
        preds = preds / count
        size = (H, Wd) if ori_shape is None else tuple(ori_shape)
        return F.interpolate(preds, size=size, mode="bilinear")

    # ---- segmentor.py:475-499 ---------------------------------------------------------------
    def postprocess(self, seg_logits):
        """seg_logits [Q,H,W] (one image) -> (class probabilities [K,H,W], labels int64 [1,H,W])."""
        p = torch.softmax(seg_logits * self.logit_scale, dim=0)
        K = int(self.query_idx.max()) + 1
        if K != self.query_idx.numel():
            onehot = F.one_hot(self.query_idx, K).T.view(K, -1, 1, 1).to(p.dtype)
            p = (p.unsqueeze(0) * onehot).max(1)[0]
        pred = p.argmax(0, keepdim=True)
        pred[p.max(0, keepdim=True)[0] < self.prob_thd] = self.bg_idx
        return p, pred

    def predict(self, img):
        if self.slide_crop > 0:
            lg = self.forward_slide(img)
        else:
            lg = self.forward_feature(img, img.shape[-2:])
        return self.postprocess(lg[0])


def read_class_file(path: str):
    """segmentor.py:611-622 ``get_cls_idx``: one class per line, commas separate synonyms."""
    names, idx = [], []
    with open(path) as f:
        for i, line in enumerate(f.readlines()):
            parts = line.split(",")
            names += [s.replace("\n", "") for s in parts]
            idx += [i] * len(parts)
    return names, idx
