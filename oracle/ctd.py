"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's Cluster-Then-Debias step (oracle, not product).

Follows CTD.py as ``segmentor.py:339-365`` drives it: ``cluster_patch_tokens_dbscan(metric='euclidean', eps=1.1,
min_samples=11)`` (CTD.py:147-296; only the labels are used there) and ``adaptive_debiasing(items, labels, cls, factor=-1.5)``
(CTD.py:299-366).  DBSCAN itself is scikit-learn 1.7.2 in the reference (CTD.py:127-140, brute-force radius neighbours in
float64 on float32 points); ``dbscan_labels`` restates its published algorithm (core points = >= min_samples neighbours within
eps, self included; clusters = connected components of core points numbered by their smallest core index; a border point joins
the lowest-numbered cluster that has a core point within eps of it; the rest is noise, -1).  Pinned by tests/golden/ctd.npz,
minted by oracle/gen_golden.py from the reference functions themselves.
"""
from __future__ import annotations

import numpy as np
import torch


def ctd_points(feats: torch.Tensor) -> torch.Tensor:
    """CTD.py:212-219 (euclidean, use_spatial=False, feat_weight=1) then dbscan()'s own normalisation (CTD.py:104)."""
    f = feats.float()
    f = f / (f.norm(dim=-1, keepdim=True) + 1.1)
    return f / (f.norm(dim=-1, keepdim=True) + 1e-8)


def neighbour_matrix(points: np.ndarray, eps: float) -> np.ndarray:
    """sklearn's brute-force radius query on float32 data: squared distances by the Gram expansion in float64."""
    x = points.astype(np.float64)
    nx = (x * x).sum(1)
    d2 = nx[:, None] + nx[None, :] - 2.0 * (x @ x.T)
    return d2 <= float(eps) ** 2, d2


def dbscan_labels(points: np.ndarray, eps: float, min_samples: int) -> np.ndarray:
    nb, _ = neighbour_matrix(points, eps)
    n = points.shape[0]
    core = nb.sum(1) >= min_samples
    labels = np.full(n, -1, np.int64)
    cid = 0
    for i in range(n):                                     # sklearn _dbscan_inner: index order, depth-first expansion
        if labels[i] != -1 or not core[i]:
            continue
        stack = [i]
        while stack:
            j = stack.pop()
            if labels[j] == -1:
                labels[j] = cid
                if core[j]:
                    stack.extend(k for k in np.nonzero(nb[j])[0] if labels[k] == -1)
        cid += 1
    return labels


def adaptive_debiasing(items: torch.Tensor, labels: torch.Tensor, bias: torch.Tensor, factor: float, eps: float = 1.1) -> torch.Tensor:
    """CTD.py:299-366: items [B,n,C], labels [B,n] (-1 noise), bias [B,C]."""
    out = items.clone().float()
    for b in range(items.shape[0]):
        lab = labels[b]
        valid = lab >= 0
        if not bool(valid.any()):
            continue
        ids = lab[valid]
        K = int(ids.max()) + 1
        pl = items[b, valid].float()
        sums = torch.zeros(K, items.shape[-1]).index_add_(0, ids, pl)
        counts = torch.zeros(K).index_add_(0, ids, torch.ones(ids.shape[0]))
        protos = sums / counts.clamp_min(1.0).unsqueeze(1)
        pu = protos / (protos.norm(dim=-1, keepdim=True) + eps)
        cv = bias[b].float()
        cu = cv / (cv.norm(dim=-1, keepdim=True) + eps)
        sims = (pu * cu.unsqueeze(0)).sum(-1).clamp(-1.0, 1.0)
        out[b, valid] = out[b, valid] + sims[ids].unsqueeze(1) * (factor * cv).unsqueeze(0)
    return out


def ctd_debias(feats: torch.Tensor, cls: torch.Tensor, eps: float = 1.1, min_samples: int = 11, factor: float = -1.5):
    """segmentor.py:339-365 -> (features [B,n,C], labels int64 [B,n])."""
    pts = ctd_points(feats)
    labels = torch.stack([torch.from_numpy(dbscan_labels(pts[b].numpy(), eps, min_samples)) for b in range(feats.shape[0])], 0)
    return adaptive_debiasing(feats, labels, cls, factor), labels


def make_clustered_tokens(B: int, n: int, C: int, seed: int = 0, n_centers: int = 5, spread: float = 0.55, noise_frac: float = 0.15):
    """Synthetic patch tokens with cluster structure (several DBSCAN clusters, border points and noise at eps 1.1 / 11)."""
    g = np.random.default_rng(seed)
    out = np.empty((B, n, C), np.float32)
    for b in range(B):
        centers = g.standard_normal((n_centers, C)).astype(np.float32)
        centers /= np.linalg.norm(centers, axis=1, keepdims=True)
        which = g.integers(0, n_centers, n)
        x = centers[which] + spread * g.standard_normal((n, C)).astype(np.float32) / np.sqrt(C) * g.uniform(0.5, 2.5, (n, 1)).astype(np.float32)
        noise = g.random(n) < noise_frac
        x[noise] = g.standard_normal((int(noise.sum()), C)).astype(np.float32) / np.sqrt(C)
        out[b] = x * g.uniform(0.5, 3.0, (n, 1)).astype(np.float32)
    return out
