"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's text tower (oracle, not product).

Follows open_clip/model.py:288-306 (``CLIP.encode_text``), :239-245 (``build_causal_mask``), open_clip/transformer.py:942-954
(``text_global_pool`` 'argmax') and the standard ``ResidualAttentionBlock`` (transformer.py:199-264).  Pinned by
tests/golden/text_*.npz, produced by oracle/gen_golden.py from the reference's own ``CLIP.encode_text``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a


def encode_text(w, cfg, tokens, normalize: bool = False) -> torch.Tensor:
    """tokens int [S, ctx] -> [S, E] (fp32)."""
    w = {k: _t(v).float() for k, v in w.items()}
    tokens = _t(tokens).long()
    S, N = tokens.shape
    D, H = cfg.width, cfg.heads
    dh = D // H
    x = w["token_embedding.weight"][tokens] + w["positional_embedding"][:N]              # model.py:291-293
    mask = torch.full((N, N), float("-inf")).triu_(1)                                     # model.py:239-245
    for i in range(cfg.layers):
        p = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (D,), w[p + "ln_1.weight"], w[p + "ln_1.bias"], 1e-5)
        qkv = h @ w[p + "attn.in_proj_weight"].T + w[p + "attn.in_proj_bias"]
        q, k, v = (t.reshape(S, N, H, dh).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
        a = torch.softmax((q * dh ** -0.5) @ k.transpose(-1, -2) + mask, dim=-1)
        ctx = (a @ v).permute(0, 2, 1, 3).reshape(S, N, D)
        x = x + ctx @ w[p + "attn.out_proj.weight"].T + w[p + "attn.out_proj.bias"]
        h = F.layer_norm(x, (D,), w[p + "ln_2.weight"], w[p + "ln_2.bias"], 1e-5)
        h = h @ w[p + "mlp.c_fc.weight"].T + w[p + "mlp.c_fc.bias"]
        h = h * torch.sigmoid(1.702 * h) if cfg.quick_gelu else F.gelu(h)
        x = x + h @ w[p + "mlp.c_proj.weight"].T + w[p + "mlp.c_proj.bias"]
    x = F.layer_norm(x, (D,), w["ln_final.weight"], w["ln_final.bias"], 1e-5)             # model.py:297
    pooled = x[torch.arange(S), tokens.argmax(dim=-1)]                                    # transformer.py:949-951
    out = pooled @ w["text_projection"]                                                   # model.py:299-303
    return F.normalize(out, dim=-1) if normalize else out


def query_features(w, cfg, token_batches) -> torch.Tensor:
    """Prompt-ensemble averaging of segmentor.py:157-174: one [n_templates, ctx] id batch per query word -> [Q, E]."""
    rows = []
    for tok in token_batches:
        f = encode_text(w, cfg, tok)
        f = f / f.norm(dim=-1, keepdim=True)
        f = f.mean(dim=0)
        rows.append(f / f.norm())
    return torch.stack(rows)
