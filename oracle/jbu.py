"""TEST INFRASTRUCTURE ONLY (the checker, never the product path).

CPU fp32 restatement of the SimFeatUp joint bilateral upsampler
(/root/reference/simfeatup_dev/upsamplers.py:202-325) over a flat weight dict with the key names of
``get_upsampler(name, dim).state_dict()``.

The adaptive convolution is FeatUp's CUDA extension ``featup.adaptive_conv_cuda`` (third party,
NOT vendored, version unpinned -- not in the reference's requirements.txt).  Its published
semantics are restated from the reference's own in-tree torch form
``adaptive_conv_py_simple`` (upsamplers.py:14-25):
    out[b,c,y,x] = sum_{i,j<d} in[b,c,y+i,x+j] * filt[b,y,x,i,j]
No reference test covers that op, so it is pinned only through ``adaptive_conv_py_simple``
run in the build container (tests/golden/jbu_*.npz).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

W = Dict[str, torch.Tensor]


def adaptive_conv(padded, filters):
    """padded [B,C,h+d-1,w+d-1], filters [B,h,w,d,d] -> [B,C,h,w], without the 12 GB unfold."""
    B, C, _, _ = padded.shape
    _, h, w, d, _ = filters.shape
    out = torch.zeros(B, C, h, w, dtype=padded.dtype)
    for i in range(d):
        for j in range(d):
            out += padded[:, :, i:i + h, j:j + w] * filters[:, None, :, :, i, j]
    return out


def radius_of(w: W, up: str) -> int:
    d2 = w[up + ".fixup_proj.3.bias"].shape[0]
    d = int(round(d2 ** 0.5))
    return (d - 1) // 2


def range_kernel(w: W, up: str, guidance, r: int):
    """upsamplers.py:230-238."""
    d = 2 * r + 1
    proj = F.conv2d(guidance, w[up + ".range_proj.0.weight"], w[up + ".range_proj.0.bias"])
    proj = F.conv2d(F.gelu(proj), w[up + ".range_proj.3.weight"], w[up + ".range_proj.3.bias"])
    B, K, gh, gw = proj.shape
    pp = F.pad(proj, [r] * 4, mode="reflect")
    temp = w[up + ".range_temp"].exp().clamp(1e-4, 1e4)
    dots = []
    for i in range(d):
        for j in range(d):
            dots.append((pp[:, :, i:i + gh, j:j + gw] * proj).sum(1))
    return torch.softmax(temp * torch.stack(dots, 1), dim=1)       # [B,d*d,gh,gw]


def spatial_kernel(w: W, up: str, r: int):
    """upsamplers.py:240-251 (meshgrid 'ij')."""
    d = 2 * r + 1
    t = torch.linspace(-1, 1, d)
    dist = t[:, None] ** 2 + t[None, :] ** 2
    return torch.exp(-dist / (2 * w[up + ".sigma_spatial"] ** 2)).reshape(1, d * d, 1, 1)


def combined_kernel(w: W, up: str, guidance, r: int):
    """upsamplers.py:258-266 -> [B,gh,gw,d,d]."""
    d = 2 * r + 1
    k = range_kernel(w, up, guidance, r) * spatial_kernel(w, up, r)
    k = k / k.sum(1, keepdim=True).clamp(1e-7)
    f = F.conv2d(torch.cat([k, guidance], 1), w[up + ".fixup_proj.0.weight"], w[up + ".fixup_proj.0.bias"])
    f = F.conv2d(F.gelu(f), w[up + ".fixup_proj.3.weight"], w[up + ".fixup_proj.3.bias"])
    k = k + 0.1 * f
    B, _, gh, gw = k.shape
    return k.permute(0, 2, 3, 1).reshape(B, gh, gw, d, d)


def jbu_stage(w: W, up: str, source, guidance):
    """JBULearnedRange.forward, upsamplers.py:253-275.  source [B,C,h,w], guidance [B,3,2h,2w]."""
    r = radius_of(w, up)
    gh, gw = guidance.shape[-2:]
    k = combined_kernel(w, up, guidance, r)
    hr = F.interpolate(source, size=(gh, gw), mode="bicubic", align_corners=False)
    hr = F.pad(hr, [r] * 4, mode="reflect")
    return adaptive_conv(hr, k.to(hr.dtype))


def jbu_forward(w: W, source, guidance_full):
    """JBUOne.forward :304-325 / JBUStack.forward :278-301 (which one is decided by the keys)."""
    ups = ["up"] * 4 if "up.range_temp" in w else [f"up{i}" for i in range(1, 5)]
    x = source
    for up in ups:
        h, ww = x.shape[-2:]
        small = F.adaptive_avg_pool2d(guidance_full, (h * 2, ww * 2))
        x = jbu_stage(w, up, x, small)
    return F.conv2d(x, w["fixup_proj.1.weight"], w["fixup_proj.1.bias"]) * 0.1 + x
