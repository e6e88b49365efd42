"""Thin torch-tensor wrappers over the C ABI (one Python function per exported op).

PyTorch is plumbing here: it owns device memory and the stream; every computation happens in
libsegearth_hip.so.  Nothing in this module falls back to torch arithmetic.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import PREC_BF16, PREC_F16, PREC_F16X2, PREC_F32, PREC_FP8, MODEL_TYPES, check


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libsegearth_hip ops need device tensors (no CPU fallback exists)")


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """The caller's current stream ON ``device`` (default: the current device)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on_tensor_device(fn):
    """Run an op on the device its first GPU tensor argument lives on: the context-free entry points launch on the CURRENT
    device, so a tensor on cuda:1 in a process whose current device is cuda:0 must switch first (stream_ptr() then picks that
    device's current stream)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if torch.is_tensor(a) and a.is_cuda:
                with torch.cuda.device(a.device):
                    return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    return wrapper


def precision_id(precision) -> int:
    if precision in (PREC_F32, "f32", "fp32", torch.float32):
        return PREC_F32
    if precision in (PREC_BF16, "bf16", torch.bfloat16):
        return PREC_BF16
    if precision in (PREC_FP8, "fp8"):
        return PREC_FP8
    if precision in (PREC_F16, "f16", "fp16", "half", torch.float16):
        return PREC_F16
    if precision in (PREC_F16X2, "f16x2", "exact"):       # two-plane f16: f32-grade results on the f16 matrix pipe (include/segearth_hip.h)
        return PREC_F16X2
    raise ValueError(f"unknown precision {precision!r}")


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous().float()


def scratch(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256) + 256, dtype=torch.uint8, device=device)


def _aligned(buf: torch.Tensor) -> Tuple[C.c_void_p, int]:
    p = buf.data_ptr()
    off = (-p) % 256
    return C.c_void_p(p + off), buf.numel() - off


@on_tensor_device
def linear(A, W, bias=None, residual=None, act: int = 0, precision="bf16"):
    """act(A @ W^T + bias) (+ residual); act 0 none / 1 QuickGELU / 2 erf-GELU."""
    lib = _lib.load()
    A, W = _f32(A), _f32(W)
    _require_gpu(A, W, bias, residual)
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    Kp = (K + 63) // 64 * 64
    buf = scratch((M + N) * Kp * 4 + 1024, A.device)      # 4 bytes per element covers every operand form (two-plane f16 included)
    sp, sn = _aligned(buf)
    bias = None if bias is None else _f32(bias)
    residual = None if residual is None else _f32(residual)
    check(lib.sg_op_linear(ptr(A), ptr(W), ptr(bias), ptr(residual), ptr(out), M, N, K, act, precision_id(precision), sp, sn,
                           stream_ptr()), "sg_op_linear")
    return out


@on_tensor_device
def quantize_rows_fp8(x):
    """f32 [rows, D] -> (uint8 e4m3 [rows, D], f32 scales [rows]); scale = max|row| / 448."""
    lib = _lib.load()
    x = _f32(x)
    _require_gpu(x)
    rows, D = x.shape
    q = torch.empty(rows, D, dtype=torch.uint8, device=x.device)
    sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib.sg_quantize_rows_fp8(ptr(x), rows, D, ptr(q), ptr(sc), stream_ptr()), "sg_quantize_rows_fp8")
    return q, sc


@on_tensor_device
def ln_chain(A, W1, b1, x, gamma, beta, W2, b2, act: int = 0, precision="bf16", fold: bool = True):
    """x_new = x + A @ W1^T + b1;  y = act(LayerNorm(x_new) @ W2^T + b2): the residual GEMM -> LayerNorm -> GEMM chain of a block, with the
    LayerNorm folded into the two GEMMs (fold) or as its own pass.  Returns (x_new, y), both f32."""
    lib = _lib.load()
    A, W1, b1, gamma, beta, W2, b2 = (_f32(t) for t in (A, W1, b1, gamma, beta, W2, b2))
    x = _f32(x).clone()
    M, K1 = A.shape
    D, N2 = W1.shape[0], W2.shape[0]
    y = torch.empty(M, N2, dtype=torch.float32, device=A.device)
    nbytes = lib.sg_op_ln_chain_scratch_bytes(M, K1, D, N2)
    buf = scratch(nbytes, A.device)
    sp, sn = _aligned(buf)
    check(lib.sg_op_ln_chain(ptr(A), ptr(W1), ptr(b1), ptr(x), ptr(gamma), ptr(beta), ptr(W2), ptr(b2), ptr(y), M, K1, D, N2, act,
                             precision_id(precision), int(fold), sp, sn, stream_ptr()), "sg_op_ln_chain")
    return x, y


@on_tensor_device
def gemm_fp8_mx(a8, w8, sw, sa=None, a_mx=None, bias=None, residual=None, act: int = 0, out_bf16: bool = False, mx_out: bool = False):
    """The MXFP8 forms of the fp8 GEMM (include/segearth_hip.h: sg_gemm_fp8_mx_raw) on caller-quantised operands.  a8 [M,K] u8 with
    either per-row scales `sa` [M] or E8M0 block scales `a_mx` [K/128, M, 4] u8; w8 [N,K] u8 with per-row scales sw [N].
    Returns C, or (c8 [M,N] u8, c_scale [N/128, M, 4] u8) with mx_out."""
    lib = _lib.load()
    M, K = a8.shape
    N = w8.shape[0]
    bias = None if bias is None else _f32(bias)
    residual = None if residual is None else _f32(residual)
    if mx_out:
        c8 = torch.empty(M, N, dtype=torch.uint8, device=a8.device)
        cs = torch.zeros(N // 128, M, 4, dtype=torch.uint8, device=a8.device)
        out = None
    else:
        out = torch.empty(M, N, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=a8.device)
        c8 = cs = None
    check(lib.sg_gemm_fp8_mx_raw(ptr(a8), ptr(sa), ptr(a_mx), ptr(w8), ptr(sw), ptr(bias), ptr(residual), ptr(out), ptr(c8), ptr(cs),
                                 M, N, K, act, int(out_bf16), stream_ptr()), "sg_gemm_fp8_mx_raw")
    return (c8, cs) if mx_out else out


@on_tensor_device
def linear_fp8(A, W, bias=None, residual=None, act: int = 0, out_bf16: bool = False):
    """act(dequant(fp8(A) @ fp8(W)^T) + bias) (+ residual) on the fp8 MFMA path; A [M,K], W [N,K] f32, K % 128 == 0."""
    lib = _lib.load()
    a8, sa = quantize_rows_fp8(A)
    w8, sw = quantize_rows_fp8(W)
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=A.device)
    bias = None if bias is None else _f32(bias)
    residual = None if residual is None else _f32(residual)
    check(lib.sg_gemm_fp8_raw(ptr(a8), ptr(sa), ptr(w8), ptr(sw), ptr(bias), ptr(residual), ptr(out), M, N, K, act, int(out_bf16),
                              stream_ptr()), "sg_gemm_fp8_raw")
    return out, (a8, sa, w8, sw)


@on_tensor_device
def layernorm(x, gamma, beta, eps: float = 1e-5):
    lib = _lib.load()
    x, gamma, beta = _f32(x), _f32(gamma), _f32(beta)
    _require_gpu(x, gamma, beta)
    D = x.shape[-1]
    y = torch.empty_like(x)
    check(lib.sg_op_layernorm(ptr(x), ptr(gamma), ptr(beta), ptr(y), x.numel() // D, D, eps, stream_ptr()), "sg_op_layernorm")
    return y


@on_tensor_device
def attention(qkv, heads: int, variant: str = "vanilla", sim=None, sim_weight: float = 1.0, precision="bf16",
              want_stats: bool = False):
    """Multi-term attention over packed qkv [B,N,3D] -> ctx [B,N,D] (+ head-averaged A[0,:], diag(A))."""
    lib = _lib.load()
    qkv = _f32(qkv)
    _require_gpu(qkv, sim)
    B, N, D3 = qkv.shape
    D = D3 // 3
    ctx = torch.empty(B, N, D, dtype=torch.float32, device=qkv.device)
    a_cls = torch.empty(B, N, dtype=torch.float32, device=qkv.device) if want_stats else None
    a_diag = torch.empty(B, N, dtype=torch.float32, device=qkv.device) if want_stats else None
    pid = precision_id(precision)
    buf = scratch(lib.sg_op_attention_scratch_bytes(B, N, D, heads, pid), qkv.device)
    sp, sn = _aligned(buf)
    sim = None if sim is None else _f32(sim)
    check(lib.sg_op_attention(ptr(qkv), B, N, D, heads, MODEL_TYPES[variant], ptr(sim), float(sim_weight), ptr(ctx), ptr(a_cls),
                              ptr(a_diag), pid, sp, sn, stream_ptr()), "sg_op_attention")
    return (ctx, a_cls, a_diag) if want_stats else ctx


@on_tensor_device
def similarity_map(patches, temperature: float = 1.0, add_self_similarity: bool = True, precision="f32"):
    """patches [B,n,D] -> cosine self-similarity [B,n,n] (similarity_enhancement.py:37-66)."""
    lib = _lib.load()
    patches = _f32(patches)
    _require_gpu(patches)
    B, n, D = patches.shape
    sim = torch.empty(B, n, n, dtype=torch.float32, device=patches.device)
    buf = scratch(B * n * D * 4 + 1024, patches.device)
    sp, sn = _aligned(buf)
    check(lib.sg_similarity_map(ptr(patches), n * D, D, B, n, D, float(temperature), int(add_self_similarity),
                                precision_id(precision), ptr(sim), sp, sn, stream_ptr()), "sg_similarity_map")
    return sim


@on_tensor_device
def outlier_suppress(feats, attn_cls, attn_diag, gh: int, gw: int, top_k: int = 10, contamination_temp: float = 0.1):
    """feats [B,gh*gw,D] (returns a refined copy) + the selected indices [B,k]."""
    lib = _lib.load()
    feats = _f32(feats).clone()
    attn_cls, attn_diag = _f32(attn_cls), _f32(attn_diag)
    _require_gpu(feats, attn_cls, attn_diag)
    B, n, D = feats.shape
    k = min(top_k, n)
    idx = torch.empty(B, k, dtype=torch.int32, device=feats.device)
    buf = scratch(lib.sg_outlier_scratch_bytes(B, D, k), feats.device)
    sp, _ = _aligned(buf)
    check(lib.sg_outlier_suppress(ptr(feats), ptr(attn_cls), ptr(attn_diag), B, gh, gw, D, top_k, float(contamination_temp),
                                  ptr(idx), sp, stream_ptr()), "sg_outlier_suppress")
    return feats, idx


@on_tensor_device
def weak_token_replace(feats, attn_diag, gh: int, gw: int, top_k: int = 10):
    lib = _lib.load()
    feats = _f32(feats).clone()
    attn_diag = _f32(attn_diag)
    _require_gpu(feats, attn_diag)
    B, n, D = feats.shape
    k = min(top_k, n)
    idx = torch.empty(B, k, dtype=torch.int32, device=feats.device)
    buf = scratch(lib.sg_outlier_scratch_bytes(B, D, k), feats.device)
    sp, _ = _aligned(buf)
    check(lib.sg_weak_token_replace(ptr(feats), ptr(attn_diag), B, gh, gw, D, top_k, ptr(idx), sp, stream_ptr()),
          "sg_weak_token_replace")
    return feats, idx


@on_tensor_device
def cosine_logits(tokens, cls, text, global_debias_factor: float = 0.0, cls_token_lambda: float = 0.0, two_plane: bool = False):
    """tokens [B,n,E], cls [B,E] or None, text [Q,E] -> logits [B,Q,n] (segmentor.py:309-336,374-386).
    ``two_plane`` (no global debias): the per-pixel form of the exact tower mode -- the product on the f16 matrix pipe, operands as two f16 planes."""
    lib = _lib.load()
    tokens, text = _f32(tokens), _f32(text)
    cls = None if cls is None else _f32(cls)
    _require_gpu(tokens, text, cls)
    B, n, E = tokens.shape
    Q = text.shape[0]
    out = torch.empty(B, Q, n, dtype=torch.float32, device=tokens.device)
    if two_plane and float(global_debias_factor) == 0.0:
        check(lib.sg_cosine_logits_two_plane(ptr(tokens), ptr(cls), ptr(text), B, n, E, Q, float(cls_token_lambda), ptr(out), stream_ptr()),
              "sg_cosine_logits_two_plane")
        return out
    check(lib.sg_cosine_logits(ptr(tokens), ptr(cls), ptr(text), B, n, E, Q, float(global_debias_factor), float(cls_token_lambda),
                               ptr(out), stream_ptr()), "sg_cosine_logits")
    return out


@on_tensor_device
def stitch(tile_logits, windows, up_hw, pad_tl, canvas_hw):
    """tile_logits [T,Q,gh,gw]; windows int32 [T,4] (y1,y2,x1,x2) -> canvas [Q,H,W]."""
    lib = _lib.load()
    tile_logits = _f32(tile_logits)
    windows = windows.to(device=tile_logits.device, dtype=torch.int32).contiguous()
    _require_gpu(tile_logits)
    T, Q, gh, gw = tile_logits.shape
    H, W = canvas_hw
    canvas = torch.empty(Q, H, W, dtype=torch.float32, device=tile_logits.device)
    check(lib.sg_stitch(ptr(tile_logits), ptr(windows), T, Q, gh, gw, up_hw[0], up_hw[1], pad_tl[0], pad_tl[1], H, W, ptr(canvas),
                        stream_ptr()), "sg_stitch")
    return canvas


@on_tensor_device
def resize_bilinear(src, size):
    """[C,h,w] -> [C,H,W], align_corners=False."""
    lib = _lib.load()
    src = _f32(src)
    _require_gpu(src)
    Cc, h, w = src.shape
    H, W = size
    dst = torch.empty(Cc, H, W, dtype=torch.float32, device=src.device)
    check(lib.sg_resize_bilinear(ptr(src), Cc, h, w, ptr(dst), H, W, stream_ptr()), "sg_resize_bilinear")
    return dst


@on_tensor_device
def postprocess(logits, query_idx, num_classes: int, logit_scale: float, prob_thd: float, bg_idx: int, want_probs: bool = True):
    """logits [Q,H,W] -> (probs [K,H,W] or None, labels int64 [1,H,W]) (segmentor.py:475-489)."""
    lib = _lib.load()
    logits = _f32(logits)
    _require_gpu(logits)
    Q, H, W = logits.shape
    qi = query_idx.to(device=logits.device, dtype=torch.int32).contiguous()
    probs = torch.empty(num_classes, H, W, dtype=torch.float32, device=logits.device) if want_probs else None
    labels = torch.empty(1, H, W, dtype=torch.int64, device=logits.device)
    check(lib.sg_postprocess(ptr(logits), ptr(qi), Q, num_classes, H, W, float(logit_scale), float(prob_thd), int(bg_idx),
                             ptr(probs), ptr(labels), stream_ptr()), "sg_postprocess")
    return probs, labels


@on_tensor_device
def render_maps(labels, probs, palette, want_mask: bool = True, want_heat: bool = True):
    """labels int64 [1,H,W] or [H,W], probs [K,H,W], palette uint8 [K,3] -> (mask uint8 [H,W,3] | None, heat uint8 [H,W,3] | None)."""
    lib = _lib.load()
    labels = labels.reshape(labels.shape[-2], labels.shape[-1]).contiguous()
    _require_gpu(labels, probs)
    H, W = labels.shape
    K = int(palette.shape[0])
    pal = palette.to(device=labels.device, dtype=torch.uint8).contiguous()
    probs = None if probs is None else _f32(probs)
    mask = torch.empty(H, W, 3, dtype=torch.uint8, device=labels.device) if want_mask else None
    heat = torch.empty(H, W, 3, dtype=torch.uint8, device=labels.device) if want_heat else None
    check(lib.sg_render_maps(ptr(labels), ptr(probs), ptr(pal), K, H, W, ptr(mask), ptr(heat), stream_ptr()), "sg_render_maps")
    return mask, heat


@on_tensor_device
def adaptive_conv(inp, filters):
    """FeatUp AdaptiveConv.apply: inp [B,C,h+d-1,w+d-1], filters [B,h,w,d,d] -> [B,C,h,w]."""
    lib = _lib.load()
    inp, filters = _f32(inp), _f32(filters)
    _require_gpu(inp, filters)
    B, Cc = inp.shape[:2]
    _, h, w, d, _ = filters.shape
    out = torch.empty(B, Cc, h, w, dtype=torch.float32, device=inp.device)
    check(lib.sg_adaptive_conv(ptr(inp), ptr(filters), B, Cc, h, w, d, ptr(out), stream_ptr()), "sg_adaptive_conv")
    return out


@on_tensor_device
def cross_tile_fusion(tokens, hg: int, wg: int, gh: int, gw: int, boundary_width: int = 2, fusion_mode: str = "weighted",
                      fusion_strength: float = 0.3):
    """tokens [hg*wg, gh*gw, C] (tiles in raster order) -> fused copy (reference cross_tile_fusion.py, B=1 tile-by-tile semantics)."""
    lib = _lib.load()
    tokens = _f32(tokens).clone()
    _require_gpu(tokens)
    T, n, Cc = tokens.shape
    assert T == hg * wg and n == gh * gw
    buf = scratch(lib.sg_cross_tile_scratch_bytes(T, gh, gw, Cc, boundary_width), tokens.device)
    sp, _ = _aligned(buf)
    check(lib.sg_cross_tile_fusion(ptr(tokens), hg, wg, gh, gw, Cc, boundary_width, 0 if fusion_mode == "weighted" else 1,
                                   float(fusion_strength), sp, stream_ptr()), "sg_cross_tile_fusion")
    return tokens


@on_tensor_device
def global_debias(tokens, cls, factor: float):
    """tokens - cls_hat * (cos(tokens, cls_hat) * factor)  (segmentor.py:322-336); cls is normalised inside."""
    lib = _lib.load()
    tokens, cls = _f32(tokens), _f32(cls)
    _require_gpu(tokens, cls)
    B, n, E = tokens.shape
    out = torch.empty_like(tokens)
    check(lib.sg_global_debias(ptr(tokens), ptr(cls), B, n, E, float(factor), ptr(out), stream_ptr()), "sg_global_debias")
    return out


@on_tensor_device
def ctd_debias(tokens, cls, eps: float = 1.1, min_samples: int = 11, factor: float = -1.5, want_labels: bool = True, normalize_cls: bool = False):
    """Cluster-Then-Debias (segmentor.py:339-365): tokens [B,n,C], CLS features [B,C] (unit unless ``normalize_cls``) ->
    (debiased copy, labels int32 [B,n])."""
    lib = _lib.load()
    tokens = _f32(tokens).clone()
    cls = _f32(cls)
    _require_gpu(tokens, cls)
    B, n, Cc = tokens.shape
    labels = torch.empty(B, n, dtype=torch.int32, device=tokens.device) if want_labels else None
    need = lib.sg_ctd_scratch_bytes(B, n, Cc)
    buf = scratch(need, tokens.device)
    sp, sn = _aligned(buf)
    check(lib.sg_ctd_debias(ptr(tokens), ptr(cls), B, n, Cc, float(eps), int(min_samples), float(factor), int(normalize_cls), ptr(labels), sp, sn,
                            stream_ptr()), "sg_ctd_debias")
    return tokens, labels


class CrossTileSteps:
    """The three device steps of the sharded cross-tile fusion (sg_cross_tile_pack / _fuse / _apply) for a rank holding tiles
    [tile0, tile0+n_local) of a scene whose tile grid is ``wg`` wide.  ``pipeline.sharded_cross_tile_fusion`` drives them and
    does the two strip all-gathers in between; the CPU tests replace this class with a torch stand-in."""

    def __init__(self, gh: int, gw: int, C_: int, boundary_width: int, fusion_mode: str, fusion_strength: float, wg: int):
        self.lib = _lib.load()
        self.gh, self.gw, self.C, self.bw, self.wg = gh, gw, C_, boundary_width, wg
        self.mode = 0 if fusion_mode == "weighted" else 1
        self.strength = float(fusion_strength)

    def strip_len(self, which: int) -> int:
        return self.gh * self.bw if which == 0 else self.bw * self.gw

    @on_tensor_device
    def pack(self, tokens, tile0: int, which: int, left_result=None):
        _require_gpu(tokens, left_result)
        out = torch.empty(tokens.shape[0], self.strip_len(which), self.C, dtype=torch.float32, device=tokens.device)
        check(self.lib.sg_cross_tile_pack(ptr(tokens), ptr(left_result), tokens.shape[0], tile0, self.wg, self.gh, self.gw, self.C, self.bw,
                                          which, ptr(out), stream_ptr()), "sg_cross_tile_pack")
        return out

    @on_tensor_device
    def fuse(self, tokens, tile0: int, nbr_strips, which: int):
        _require_gpu(tokens, nbr_strips)
        res = torch.zeros(tokens.shape[0], self.strip_len(which), self.C, dtype=torch.float32, device=tokens.device)
        check(self.lib.sg_cross_tile_fuse(ptr(tokens), ptr(nbr_strips), tokens.shape[0], tile0, self.wg, self.gh, self.gw, self.C, self.bw,
                                          self.mode, self.strength, which, ptr(res), stream_ptr()), "sg_cross_tile_fuse")
        return res

    @on_tensor_device
    def apply(self, tokens, tile0: int, left_result, top_result):
        _require_gpu(tokens, left_result, top_result)
        check(self.lib.sg_cross_tile_apply(ptr(tokens), ptr(left_result), ptr(top_result), tokens.shape[0], tile0, self.wg, self.gh, self.gw,
                                           self.C, self.bw, stream_ptr()), "sg_cross_tile_apply")
        return tokens
