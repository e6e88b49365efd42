"""TEST INFRASTRUCTURE ONLY (the checker, never the product path).

CPU fp32 restatement of the reference's *modified* open_clip vision tower forward, written
functionally over a flat weight dict (keys as in ``clip_decontamination_amd.weights``).
Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import it.

Parity status: PINNED -- ``tests/test_oracle_vs_golden.py`` checks every function here against
fixtures minted by running the reference itself in the build container
(``oracle/gen_golden.py``; the reference holds no numeric fixtures of its own).

Reference citations are to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import refine

W = Dict[str, torch.Tensor]

SELF_SELF_TYPES = ("vanilla", "MaskCLIP", "ClearCLIP", "SCLIP", "SegEarth", "SFP", "Experimental",
                   "NACLIP", "NOnly", "GAV")


def to_torch(weights_np) -> W:
    return {k: torch.from_numpy(v.copy()) if not torch.is_tensor(v) else v for k, v in weights_np.items()}


def activation(x, quick_gelu: bool):
    # open_clip/transformer.py:35-38 (QuickGELU) / nn.GELU (exact erf)
    return x * torch.sigmoid(1.702 * x) if quick_gelu else F.gelu(x)


def layer_norm(x, w: W, prefix: str):
    # open_clip/transformer.py:26-32; eps = nn.LayerNorm default 1e-5
    return F.layer_norm(x, (x.shape[-1],), w[prefix + ".weight"], w[prefix + ".bias"], 1e-5)


def resized_pos_embed(pos, g0: int, gh: int, gw: int):
    """open_clip/transformer.py:777-795 -- bicubic, align_corners=False, the ``+0.1`` scale hack."""
    if gh == g0 and gw == g0:
        return pos
    D = pos.shape[-1]
    grid = pos[1:].reshape(1, g0, g0, D).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=((gh + 0.1) / g0, (gw + 0.1) / g0), mode="bicubic")
    assert grid.shape[-2] == gh and grid.shape[-1] == gw
    return torch.cat([pos[:1], grid.permute(0, 2, 3, 1).reshape(gh * gw, D)], 0)


def gem_resized_pos_embed(pos, g0: int, gh: int, gw: int):
    """gem/gem_utils.py:12-43 -- ``size=`` form with antialias=True (identity when unchanged)."""
    if gh == g0 and gw == g0:
        return pos
    D = pos.shape[-1]
    grid = pos[1:].reshape(1, g0, g0, D).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(gh, gw), mode="bicubic", antialias=True)
    return torch.cat([pos[:1], grid.permute(0, 2, 3, 1).reshape(gh * gw, D)], 0)


def embed(w: W, cfg, img, gem: bool = False):
    """Patch embed + class token + positional embedding + ln_pre.
    open_clip/transformer.py:559-574.  img [B,3,H,W] (H, W multiples of P) -> x [B,N,D]."""
    B, _, H, Wd = img.shape
    P, D = cfg.patch, cfg.width
    gh, gw = H // P, Wd // P
    x = F.conv2d(img, w["conv1.weight"], stride=P)            # [B,D,gh,gw]
    x = x.reshape(B, D, gh * gw).permute(0, 2, 1)
    x = torch.cat([w["class_embedding"].view(1, 1, D).expand(B, -1, -1), x], 1)
    pos = (gem_resized_pos_embed if gem else resized_pos_embed)(w["positional_embedding"], cfg.grid0, gh, gw)
    x = x + pos
    return layer_norm(x, w, "ln_pre"), gh, gw


def split_heads(t, H):
    B, N, D = t.shape
    return t.view(B, N, H, D // H).permute(0, 2, 1, 3)         # [B,H,N,dh]


def merge_heads(t):
    B, H, N, dh = t.shape
    return t.permute(0, 2, 1, 3).reshape(B, N, H * dh)


def qkv_proj(w: W, p: str, xn, H):
    qkv = F.linear(xn, w[p + "attn.in_proj_weight"], w[p + "attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)                            # rows ordered q|k|v
    return split_heads(q, H), split_heads(k, H), split_heads(v, H)


def out_proj(w: W, p: str, ctx):
    return F.linear(merge_heads(ctx), w[p + "attn.out_proj.weight"], w[p + "attn.out_proj.bias"])


def mlp(w: W, p: str, xn, quick_gelu):
    h = F.linear(xn, w[p + "mlp.c_fc.weight"], w[p + "mlp.c_fc.bias"])
    return F.linear(activation(h, quick_gelu), w[p + "mlp.c_proj.weight"], w[p + "mlp.c_proj.bias"])


def res_block(w: W, cfg, i: int, x, need_weights: bool = False):
    """open_clip/transformer.py:234-254 with nn.MultiheadAttention (q pre-scaled by dh^-1/2).
    Returns (x, head-averaged attention [B,N,N] or None)."""
    p = f"transformer.resblocks.{i}."
    H = cfg.heads
    q, k, v = qkv_proj(w, p, layer_norm(x, w, p + "ln_1"), H)
    a = torch.softmax((q * (cfg.head_dim ** -0.5)) @ k.transpose(-1, -2), dim=-1)
    x = x + out_proj(w, p, a @ v)
    x = x + mlp(w, p, layer_norm(x, w, p + "ln_2"), cfg.quick_gelu)
    return x, (a.mean(1) if need_weights else None)


def gaussian_bias(gh: int, gw: int, std: float = 1.0):
    """open_clip/transformer.py:797-820: Gaussian window centred on each patch, zero row/col for CLS."""
    ys = torch.arange(gh, dtype=torch.float32)
    xs = torch.arange(gw, dtype=torch.float32)
    c = 1.0 / (std * math.sqrt(2.0))
    dy = (ys[:, None] - ys[None, :]) * c                      # [gh,gh]
    dx = (xs[:, None] - xs[None, :]) * c
    g = torch.exp(-(dy[:, None, :, None] ** 2 + dx[None, :, None, :] ** 2))   # [gh,gw,gh,gw]
    g = g.reshape(gh * gw, gh * gw)
    out = torch.zeros(gh * gw + 1, gh * gw + 1)
    out[1:, 1:] = g
    return out


def last_block_attention(w: W, cfg, i: int, xn, model_type: str, sim_bias=None, sim_weight: float = 1.0,
                         gh: Optional[int] = None, gw: Optional[int] = None):
    """open_clip/transformer.py:822-940 (``custom_attn``) on ``xn = ln_1(x)``.
    ``sim_bias`` is the similarity map [B,n,n] (fp32) or None; it is zero-padded for the CLS
    row/column and broadcast over heads (similarity_enhancement.py:78-124)."""
    p = f"transformer.resblocks.{i}."
    H, scale = cfg.heads, cfg.head_dim ** -0.5
    q, k, v = qkv_proj(w, p, xn, H)
    B, _, N, _ = q.shape

    def bias(s):
        if sim_bias is None:
            return s
        pad = torch.zeros(B, 1, N, N, dtype=sim_bias.dtype)
        pad[:, 0, 1:, 1:] = sim_bias
        return s + sim_weight * pad.to(s.dtype)

    def ss(t):
        return (t @ t.transpose(-1, -2)) * scale

    if model_type == "vanilla":
        a = torch.softmax(bias((q @ k.transpose(-1, -2)) * scale), -1)
    elif model_type == "MaskCLIP":
        a = torch.eye(N).expand(B, H, N, N)
    elif model_type == "ClearCLIP":
        a = torch.softmax(bias(ss(q)), -1)
    elif model_type == "SCLIP":
        a = torch.softmax(bias(ss(q)), -1) + torch.softmax(bias(ss(k)), -1)
    elif model_type == "SegEarth":
        a = torch.softmax(bias(ss(q)), -1) + torch.softmax(bias(ss(k)), -1) + torch.softmax(bias(ss(v)), -1)
    elif model_type == "SFP":
        a = torch.softmax(bias(0.5 * (ss(q) + ss(k))), -1)
    elif model_type == "Experimental":
        # transformer.py:896-902: softmax, add map to the PROBABILITIES, softmax again (R8)
        a = torch.softmax(ss(k) + ss(q), -1)
        a = torch.softmax(bias(a), -1)
    elif model_type in ("NACLIP", "NOnly", "GAV"):
        omega = gaussian_bias(gh, gw).unsqueeze(0).unsqueeze(0)
        if model_type == "NACLIP":
            s = ss(k)
        else:
            # transformer.py:925,928: omega scaled by |q_i||k_j| * scale
            qn, kn = q.norm(dim=-1), k.norm(dim=-1)               # [B,H,N]
            omega = omega * scale * (qn.unsqueeze(-1) * kn.unsqueeze(-2))
            s = torch.zeros(B, H, N, N) if model_type == "NOnly" else (q @ k.transpose(-1, -2)) * scale
        a = torch.softmax(s + omega, -1)
    else:
        raise ValueError(f"unknown model_type {model_type}")
    return out_proj(w, p, a @ v)


def vit_forward(w: W, cfg, img, model_type: str = "SegEarth", ignore_residual: bool = True,
                similarity_cfg: Optional[dict] = None, outlier_cfg: Optional[dict] = None,
                self_attn_cfg: Optional[dict] = None, capture: Optional[dict] = None, layer_fusion: Optional[dict] = None):
    """open_clip/transformer.py:538-775 with ``last_n_layers=1``, ``output_cls_token=True``.
    Returns (cls [B,E], tokens [B,n,E]).

    ``layer_fusion``: None = ``apply_layer_fusion=False``; else ``{"lambda": l}`` = the reference's attention-map fusion
    (transformer.py:598-607, 630-637, 647-690) with its view(N, heads, L, L) read as what it can only mean: the attention of
    ``blk(x, need_weights=True)`` is ALREADY head-averaged [B,L,L] (nn.MultiheadAttention default), so the fused map is used as is.
    The reference reaches that reading exactly when heads == 1 (fixture vit_tiny-1h); for heads > 1 its view() raises (R9).
      * every ordinary block (and, with ignore_residual, the last block's own ``blk(x)``) contributes A_l; acc = l*acc + (1-l)*A_l;
      * the capture of block L-2's attention for the refiners is NOT taken (the ``elif`` at :609), so the self-attention enhancer and
        the outlier SUPPRESSION are skipped; with an outlier suppressor installed its top_k selects columns of the fused map to
        zero, rows are L1-renormalised (+1e-8) and ``output = attn @ output`` (all N tokens, CLS included); without one the fused
        map is discarded (the forward equals the unfused one).  ``layer_fusion_threshold`` is unused by the reference.

    ``similarity_cfg`` / ``outlier_cfg`` / ``self_attn_cfg``: None = module not installed,
    else the kwargs of the corresponding reference module.
    ``capture``: optional dict that receives intermediate tensors (for golden fixtures).
    """
    L = cfg.layers
    x, gh, gw = embed(w, cfg, img)
    mid_idx = (L - 1) // 2                                     # transformer.py:593
    want_attn = outlier_cfg is not None and layer_fusion is None   # transformer.py:609 (R6); :598 takes precedence over the elif
    lam = None if layer_fusion is None else float(layer_fusion.get("lambda", 0.5))
    x_mid, attn, acc = None, None, None
    for i in range(L - 1):
        if i == mid_idx and similarity_cfg is not None:
            x_mid = x.clone()
        x, a = res_block(w, cfg, i, x, need_weights=(lam is not None) or (want_attn and i == L - 2))
        if lam is not None:
            acc = a if acc is None else lam * acc + (1.0 - lam) * a       # transformer.py:601-607
        elif a is not None:
            attn = a
    sim = None
    sim_weight = 1.0
    if similarity_cfg is not None:
        sim = refine.similarity_map(x_mid[:, 1:], temperature=similarity_cfg.get("temperature", 1.0),
                                    add_self_similarity=similarity_cfg.get("add_self_similarity", True))
        sim_weight = similarity_cfg.get("similarity_weight", 1.0)
    p = f"transformer.resblocks.{L - 1}."
    out = last_block_attention(w, cfg, L - 1, layer_norm(x, w, p + "ln_1"), model_type, sim, sim_weight, gh, gw)
    if not ignore_residual:                                    # transformer.py:641-643
        out = x + out
        out = out + mlp(w, p, layer_norm(out, w, p + "ln_2"), cfg.quick_gelu)
    if lam is not None and ignore_residual:                    # transformer.py:630-637: the last block's ordinary attention joins the EMA
        _, a = res_block(w, cfg, L - 1, x, need_weights=True)
        acc = a if acc is None else lam * acc + (1.0 - lam) * a
    if capture is not None:
        capture.update(x_pre_last=x, x_mid=x_mid, attn=attn, sim=sim, last_out=out.clone(), gh=gh, gw=gw, fused_attn=acc)
    B, N, D = out.shape
    if lam is not None and acc is not None and outlier_cfg is not None:       # transformer.py:647-690
        idx = refine.detect_outliers(acc, gh * gw, outlier_cfg.get("top_k", 10))
        mask = torch.ones(B, N)
        for b in range(B):
            mask[b, idx[b] + 1] = 0.0
        am = acc * mask.unsqueeze(1)
        am = am / (am.sum(dim=-1, keepdim=True) + 1e-8)
        out = torch.bmm(am, out)
        if capture is not None:
            capture.update(fusion_idx=idx)
    if attn is not None and self_attn_cfg is not None:         # transformer.py:698-718
        grid = out[:, 1:].permute(0, 2, 1).reshape(B, D, gh, gw)
        grid = refine.self_attention_enhance(grid, attn, **self_attn_cfg)
        out = torch.cat([out[:, :1], grid.reshape(B, D, gh * gw).permute(0, 2, 1)], 1)
    if attn is not None and outlier_cfg is not None:           # transformer.py:721-742
        grid = out[:, 1:].permute(0, 2, 1).reshape(B, D, gh, gw)
        idx = refine.detect_outliers(attn, gh * gw, outlier_cfg.get("top_k", 10))
        grid = refine.suppress_outliers(grid, idx, outlier_cfg.get("contamination_temp", 0.1))
        out = torch.cat([out[:, :1], grid.reshape(B, D, gh * gw).permute(0, 2, 1)], 1)
        if capture is not None:
            capture.update(outlier_idx=idx)
    y = layer_norm(out, w, "ln_post") @ w["proj"]              # transformer.py:765-770
    if capture is not None:
        capture.update(refined=out)
    return y[:, 0], y[:, 1:]


# --------------------------------------------------------------------------------------------
# GEM (gem/gem_utils.py, gem/gem_wrapper.py)
# --------------------------------------------------------------------------------------------

def gem_block(w: W, cfg, i: int, x_gem, x, ignore_residual: bool, capture: Optional[dict] = None):
    """gem/gem_utils.py:60-123,132-153 with ss_attn_iter=1, ss_attn_temp=None.  ``capture`` (composition only, see gem_forward)
    receives the head-averaged attention [B,N,N] of the ORDINARY stream of this block."""
    p = f"transformer.resblocks.{i}."
    H, scale = cfg.heads, cfg.head_dim ** -0.5
    xn = layer_norm(x, w, p + "ln_1")
    q, k, v = qkv_proj(w, p, xn, H)
    a_ori = torch.softmax((q @ k.transpose(-1, -2)) * scale, -1)
    if capture is not None:
        capture["attn"] = a_ori.mean(dim=1)                    # what nn.MultiheadAttention(need_weights=True) hands back
    ori = out_proj(w, p, a_ori @ v)
    inv_temp = (xn.norm(dim=-1).mean(dim=-1) * scale).view(-1, 1, 1, 1)   # per image (:79-81)

    def stream(t):
        t = F.normalize(t, dim=-1)
        t = torch.softmax((t @ t.transpose(-1, -2)) * inv_temp, -1) @ t
        t = F.normalize(t, dim=-1)
        return torch.softmax((t @ t.transpose(-1, -2)) * inv_temp, -1) @ v

    gem = out_proj(w, p, (stream(v) + stream(k) + stream(q)) / 3)
    x_ori = x + ori
    x_ori = x_ori + mlp(w, p, layer_norm(x_ori, w, p + "ln_2"), cfg.quick_gelu)
    x_gem = gem if ignore_residual else x_gem + gem
    return x_gem, x_ori


def gem_forward(w: W, cfg, img, ignore_residual: bool = True, depth: int = 7, outlier_cfg: Optional[dict] = None):
    """gem/gem_utils.py:159-199: tokens [B,n,E] of the GEM stream (no CLS, R5).

    ``outlier_cfg`` -- BASELINE configs[2] names "GEM self-self attn + outlier_suppression" as ONE forward.  The reference cannot run it
    (SegmentorEx crashes on GEM, R5; GEM's replaced forward ignores the suppressor), so this is a DEFINED composition of two stages that
    are each pinned to the reference (parity of the composition itself is unpinned):
      * detection reads what the ordinary forward reads (transformer.py:609-610): the head-averaged attention of block L-2 -- here the
        ORDINARY ("ori") stream of that block, the only stream that runs nn.MultiheadAttention's q k^T attention;
      * suppression acts where the ordinary forward applies it (transformer.py:721-742): on the patch tokens of the final feature map
        before ln_post -- here the GEM stream x_gem -- through the unchanged OutlierSuppressionModule (outlier_suppression.py:83-214)."""
    L = cfg.layers
    x, gh, gw = embed(w, cfg, img, gem=True)
    first = L - (depth - 1)
    attn = None
    for i in range(first):
        x, a = res_block(w, cfg, i, x, need_weights=outlier_cfg is not None and i == L - 2)
        attn = a if a is not None else attn
    x_gem = x
    for i in range(first, L):
        cap = {} if (outlier_cfg is not None and i == L - 2) else None
        x_gem, x = gem_block(w, cfg, i, x_gem, x, ignore_residual, cap)
        if cap is not None:
            attn = cap["attn"]
    if outlier_cfg is not None and attn is not None:
        B, N, D = x_gem.shape
        grid = x_gem[:, 1:].permute(0, 2, 1).reshape(B, D, gh, gw)
        idx = refine.detect_outliers(attn, gh * gw, outlier_cfg.get("top_k", 10))
        grid = refine.suppress_outliers(grid, idx, outlier_cfg.get("contamination_temp", 0.1))
        x_gem = torch.cat([x_gem[:, :1], grid.reshape(B, D, gh * gw).permute(0, 2, 1)], 1)
    y = layer_norm(x_gem, w, "ln_post") @ w["proj"]
    return y[:, 1:]
