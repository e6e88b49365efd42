"""TEST INFRASTRUCTURE ONLY (the checker, never the product path).

CPU fp32 restatement of the reference's training-free token refinements:
outlier suppression, similarity enhancement, self-attention enhancement, cross-tile fusion.
Pinned against the reference modules by tests/golden fixtures (oracle/gen_golden.py).
Citations are to /root/reference.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

NEIGHBOUR_OFFSETS = ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1))


def similarity_map(patches, temperature: float = 1.0, add_self_similarity: bool = True):
    """similarity_enhancement.py:37-66: cosine self-similarity of mid-layer patch tokens, fp32.
    patches [B,n,D] -> [B,n,n]."""
    f = F.normalize(patches.float(), p=2, dim=-1)
    s = torch.bmm(f, f.transpose(1, 2)) / temperature
    if not add_self_similarity:
        s = s * (1 - torch.eye(s.shape[-1]).unsqueeze(0))
    return s


def outlier_ratio(attn, n: int):
    """outlier_suppression.py:44-53: A[cls,i] / (A[i,i] + 1e-8) for the n patch tokens."""
    diag = torch.diagonal(attn, dim1=-2, dim2=-1)[:, 1:1 + n]
    return attn[:, 0, 1:1 + n] / (diag + 1e-8)


def detect_outliers(attn, n: int, top_k: int = 10):
    """outlier_suppression.py:15-61 -> int64 [B, min(k,n)] (descending ratio)."""
    return torch.topk(outlier_ratio(attn, n), k=min(top_k, n), largest=True, dim=1).indices


def _neighbourhood(grid_b, idx, gh: int, gw: int):
    """Clamped 8-neighbour gather shared by outlier suppression and weak-token replacement.
    grid_b [D,gh,gw]; idx [k] -> centre feats [k,D], neighbour feats [k,8,D], coords, cos, weights."""
    rows = torch.div(idx, gw, rounding_mode="trunc")
    cols = idx % gw
    off = torch.tensor(NEIGHBOUR_OFFSETS, dtype=torch.long)
    ny = (rows[:, None] + off[None, :, 0]).clamp(0, gh - 1)
    nx = (cols[:, None] + off[None, :, 1]).clamp(0, gw - 1)
    centre = grid_b[:, rows, cols].T                               # [k,D]
    nbr = grid_b[:, ny, nx].permute(1, 2, 0)                       # [k,8,D]
    cos = (F.normalize(nbr, p=2, dim=2) * F.normalize(centre, p=2, dim=1).unsqueeze(1)).sum(2)
    wts = torch.softmax((1.0 - cos).clamp(min=0.0), dim=1)
    return rows, cols, ny, nx, centre, nbr, cos, wts


def suppress_outliers(grid, idx, contamination_temp: float = 0.1):
    """outlier_suppression.py:115-214.  grid [B,D,gh,gw], idx [B,k] -> new grid.
    All reads come from the ORIGINAL map; neighbour writes are overwrite / last-writer-wins in
    (outlier, neighbour) order, skipping cells equal to the outlier itself; outlier cells last."""
    B, D, gh, gw = grid.shape
    out = grid.clone()
    if idx.numel() == 0:
        return out
    for b in range(B):
        rows, cols, ny, nx, centre, nbr, cos, wts = _neighbourhood(grid[b], idx[b], gh, gw)
        replacement = (nbr * wts.unsqueeze(2)).sum(1)              # [k,D]
        sigma = (cos * contamination_temp).clamp(0, 1)
        cleaned = nbr - centre.unsqueeze(1) * sigma.unsqueeze(2)   # [k,8,D]
        for i in range(idx.shape[1]):
            for j in range(8):
                y, x = int(ny[i, j]), int(nx[i, j])
                if y != int(rows[i]) or x != int(cols[i]):
                    out[b, :, y, x] = cleaned[i, j]
        out[b, :, rows, cols] = replacement.T
    return out


def replace_weak_tokens(grid, idx):
    """self_attention_enhancement.py:247-324: neighbour-mean replacement, no decontamination."""
    B, D, gh, gw = grid.shape
    out = grid.clone()
    if idx.numel() == 0:
        return out
    for b in range(B):
        rows, cols, _, _, _, nbr, _, wts = _neighbourhood(grid[b], idx[b], gh, gw)
        out[b, :, rows, cols] = (nbr * wts.unsqueeze(2)).sum(1).T
    return out


def self_attention_enhance(grid, attn, enhancement_strength: float = 0.1, min_self_attn_threshold: float = 0.15,
                           mode: str = "feature", top_k: int = 10):
    """self_attention_enhancement.py:71-245 for the spatial ([B,D,gh,gw], patch-only) call shape
    used by the ViT hook (open_clip/transformer.py:698-718)."""
    B, D, gh, gw = grid.shape
    n = attn.shape[1] - 1
    diag = torch.diagonal(attn, dim1=-2, dim2=-1)[:, 1:1 + n]
    if mode == "feature":
        weak = torch.topk(diag, k=min(top_k, n), largest=False, dim=1).indices
        return replace_weak_tokens(grid, weak)
    # attention mode (:152-245): boost the diagonal, L1-renormalise rows, A'.[0;X]
    boost = (min_self_attn_threshold - diag).clamp(min=0.0) * enhancement_strength
    a = attn.clone()
    ar = torch.arange(1, n + 1)
    a[:, ar, ar] += boost
    a = a / (a.sum(-1, keepdim=True) + 1e-8)
    seq = grid.reshape(B, D, gh * gw).permute(0, 2, 1)
    seq = torch.cat([torch.zeros(B, 1, D, dtype=seq.dtype), seq], 1)
    out = torch.bmm(a, seq)[:, 1:]
    return out.permute(0, 2, 1).reshape(B, D, gh, gw)


# --------------------------------------------------------------------------------------------
# Cross-tile fusion (cross_tile_fusion.py; unwired in the reference)
# --------------------------------------------------------------------------------------------

def fuse_weighted(cur, nbr, fusion_strength: float, eps: float = 1e-6):
    """cross_tile_fusion.py:185-236 (adaptive branch).  cur [B,a,C], nbr [B,b,C]."""
    cn = cur / (cur.norm(dim=-1, keepdim=True) + eps)
    nn_ = nbr / (nbr.norm(dim=-1, keepdim=True) + eps)
    sim = torch.bmm(cn, nn_.transpose(1, 2))
    thr = sim.mean(-1, keepdim=True) + sim.std(-1, keepdim=True)
    margin = torch.relu(sim - thr)
    raw = margin.pow(2)
    wts = raw / (raw.sum(-1, keepdim=True) + eps)
    local = margin.mean(-1, keepdim=True).clamp(0.0, 1.0)
    agg = torch.bmm(wts, nbr)
    s = fusion_strength * local
    return cur * (1 - s) + agg * s


def fuse_attention(cur, nbr, fusion_strength: float):
    """cross_tile_fusion.py:143-183."""
    C = cur.shape[-1]
    both = torch.cat([cur, nbr], 1)
    a = torch.softmax(torch.bmm(cur, both.transpose(1, 2)) / (C ** 0.5), -1)
    return cur * (1 - fusion_strength) + torch.bmm(a, both) * fusion_strength


class CrossTileFusionOracle:
    """Behavioural restatement of ``CrossTileFusion.forward`` for B=1 contiguous inputs
    (cross_tile_fusion.py:238-320).  The reference extracts strips with ``reshape``: the
    row strips ('top'/'bottom') are views of the tile and therefore see later in-place
    writes, the column strips ('left'/'right') are copies taken before any fusion.  This
    class makes that aliasing explicit: cached 'bottom'/'top' = rows AFTER all fusions of
    the tile, cached 'left'/'right' = columns BEFORE any fusion."""

    def __init__(self, fusion_mode="weighted", cache_boundary_width=2, fusion_strength=0.3):
        self.mode, self.bw, self.strength = fusion_mode, cache_boundary_width, fusion_strength
        self.cache: Dict[Tuple[int, int], Dict[str, torch.Tensor]] = {}

    def reset_cache(self):
        self.cache.clear()

    def __call__(self, feats, h_idx, w_idx, gh, gw):
        B, N, C = feats.shape
        assert B == 1
        bw = self.bw
        g = feats.clone().view(B, gh, gw, C)
        pre_left = g[:, :, :bw].reshape(B, -1, C).clone()
        pre_right = g[:, :, -bw:].reshape(B, -1, C).clone()
        fuse = (lambda c, n: fuse_attention(c, n, self.strength)) if self.mode == "attention" \
            else (lambda c, n: fuse_weighted(c, n, self.strength))
        opposite = {"top": "bottom", "bottom": "top", "left": "right", "right": "left"}
        where = {"top": (h_idx - 1, w_idx), "bottom": (h_idx + 1, w_idx),
                 "left": (h_idx, w_idx - 1), "right": (h_idx, w_idx + 1)}
        for d in ("top", "bottom", "left", "right"):
            if where[d] not in self.cache:
                continue
            nbr = self.cache[where[d]][opposite[d]]
            if d == "top":       # row strips are live views: current values
                g[:, :bw] = fuse(g[:, :bw].reshape(B, -1, C), nbr).view(B, bw, gw, C)
            elif d == "bottom":
                g[:, -bw:] = fuse(g[:, -bw:].reshape(B, -1, C), nbr).view(B, bw, gw, C)
            elif d == "left":    # column strips are the pre-fusion copies
                g[:, :, :bw] = fuse(pre_left, nbr).view(B, gh, bw, C)
            else:
                g[:, :, -bw:] = fuse(pre_right, nbr).view(B, gh, bw, C)
        self.cache[(h_idx, w_idx)] = {
            "top": g[:, :bw].reshape(B, -1, C).clone(), "bottom": g[:, -bw:].reshape(B, -1, C).clone(),
            "left": pre_left, "right": pre_right}
        return g.reshape(B, N, C)
