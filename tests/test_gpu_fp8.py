"""fp8 (OCP e4m3) MFMA path: the raw GEMM against torch's float8_e4m3fn arithmetic, and the SG_PREC_FP8 tower against the fp32
oracle (agreement reported, SURVEY.md §8d: 'bit-exact required in fp32 parity mode; % agreement reported for bf16/fp8')."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from clip_decontamination_amd import weights as Wt
from oracle import vit as OV, segment as OS                # checker only

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rnd(*shape, seed=0, scale=1.0):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(np.float32) * scale)


def test_quantize_rows_matches_torch_e4m3():
    from clip_decontamination_amd import ops
    x = rnd(37, 256, seed=1) * torch.logspace(-3, 2, 37).view(37, 1)
    x[5] = 0.0                                             # an all-zero row: scale 1, bytes 0
    q, sc = ops.quantize_rows_fp8(x.to(DEV))
    ref_sc = x.abs().amax(dim=1) / 448.0
    ref_sc[ref_sc == 0] = 1.0
    assert torch.allclose(sc.cpu(), ref_sc, rtol=1e-6, atol=0)
    deq = q.cpu().view(torch.float8_e4m3fn).float()
    want = (x / ref_sc.view(-1, 1)).to(torch.float8_e4m3fn).float()
    # the hardware multiplies by 1/scale where torch divides: allow one e4m3 step on the (rare) values that sit on a rounding boundary
    step = torch.maximum(want.abs() * 2 ** -3, torch.tensor(2.0 ** -9))
    assert ((deq - want).abs() <= step).all()
    assert (deq == want).float().mean().item() > 0.995


@pytest.mark.parametrize("M,N,K", [(257, 384, 128), (1370, 1024, 1024), (2740, 3072, 1024), (130, 256, 512)])
@pytest.mark.parametrize("act,res,obf", [(0, False, False), (1, False, True), (0, True, False)])
def test_fp8_gemm_exact_on_its_own_operands(M, N, K, act, res, obf):
    """The MFMA result must equal the f32 product of the de-quantised operands (products of e4m3 values are exact in f32; only
    the accumulation order differs): tolerance 1e-4 relative to the output scale."""
    from clip_decontamination_amd import ops
    A, W = rnd(M, K, seed=2).to(DEV), (rnd(N, K, seed=3) * K ** -0.5).to(DEV)
    bias = rnd(N, seed=4).to(DEV)
    R = rnd(M, N, seed=5).to(DEV) if res else None
    out, (a8, sa, w8, sw) = ops.linear_fp8(A, W, bias, R, act, out_bf16=obf)
    a = a8.view(torch.float8_e4m3fn).float() * sa.view(-1, 1)
    w = w8.view(torch.float8_e4m3fn).float() * sw.view(-1, 1)
    ref = (a.double() @ w.double().T).float() + bias
    if act == 1:
        ref = ref * torch.sigmoid(1.702 * ref)
    if R is not None:
        ref = ref + R
    tol = (2e-2 if obf else 2e-4) * ref.abs().max().item()
    assert (out.float() - ref).abs().max().item() < tol
    # and the quantisation error itself stays where e4m3 puts it (3 mantissa bits): a few percent of the output scale
    exact = A @ W.T + bias
    if act == 1:
        exact = exact * torch.sigmoid(1.702 * exact)
    if R is not None:
        exact = exact + R
    assert (out.float() - exact).abs().max().item() < 0.08 * exact.abs().max().item()


def _mx_decode(c8, cs):
    """e4m3 bytes [M,N] + E8M0 block scales [N/128, M, 4] -> f32 [M,N]."""
    M, N = c8.shape
    e = cs.permute(1, 0, 2).reshape(M, N // 32).to(torch.int32)                    # [M, N/32]
    sc = torch.pow(2.0, (e - 127).double()).float()
    return c8.view(torch.float8_e4m3fn).float() * sc.repeat_interleave(32, dim=1), e


@pytest.mark.parametrize("M,N,K,act", [(1370, 1024, 1024, 0), (2055, 4096, 1024, 1), (1024, 256, 128, 2)])
def test_fp8_gemm_mx_output(M, N, K, act):
    """The block-scaled epilogue: every 32-column block of act(A.W^T + bias) is stored as e4m3 times the smallest power of two that
    keeps the block inside +-448 -- tight scale, values within half an e4m3 step of the f32 result."""
    from clip_decontamination_amd import ops
    A, W = rnd(M, K, seed=2).to(DEV), (rnd(N, K, seed=3) * K ** -0.5).to(DEV)
    A = A * torch.logspace(-2, 2, M, device=DEV).view(M, 1)                         # rows of very different magnitude
    bias = rnd(N, seed=4).to(DEV)
    a8, sa = ops.quantize_rows_fp8(A)
    w8, sw = ops.quantize_rows_fp8(W)
    ref = ops.gemm_fp8_mx(a8, w8, sw, sa=sa, bias=bias, act=act)                    # same kernel, f32 output
    c8, cs = ops.gemm_fp8_mx(a8, w8, sw, sa=sa, bias=bias, act=act, mx_out=True)
    got, e = _mx_decode(c8, cs)
    blk = ref.view(M, N // 32, 32).abs().amax(dim=2)                                # block absmax of the f32 result
    qmax = (got.view(M, N // 32, 32).abs().amax(dim=2) / torch.pow(2.0, (e - 127).float().to(DEV)))
    live = blk > 1e-30
    assert (qmax[live] <= 448).all() and (qmax[live] > 208).all()                   # the scale is the tightest power of two (448/2 less one step)
    step = torch.maximum(ref.abs() * 2.0 ** -4, (blk * 2.0 ** -9).repeat_interleave(32, dim=1))   # half a step; subnormal floor relative to the block scale
    assert ((got - ref).abs() <= step * 1.001).all()


@pytest.mark.parametrize("M,N,K,res,obf", [(1370, 1024, 4096, True, False), (1024, 256, 128, False, False), (3000, 1280, 5120, True, True)])
def test_fp8_gemm_mx_operand_exact(M, N, K, res, obf):
    """A with per-(row, 32-element) E8M0 scales through the scale operand of the MFMA: equals the f64 product of the decoded operands."""
    from clip_decontamination_amd import ops
    g = torch.Generator().manual_seed(5)
    a_f = rnd(M, K, seed=6)
    a8 = a_f.to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)                      # values in [-4, 4]: plain e4m3 bytes
    a_mx = torch.randint(118, 136, (K // 128, M, 4), generator=g, dtype=torch.uint8).to(DEV)
    W = (rnd(N, K, seed=7) * K ** -0.5).to(DEV)
    w8, sw = ops.quantize_rows_fp8(W)
    bias = rnd(N, seed=8).to(DEV)
    R = rnd(M, N, seed=9).to(DEV) if res else None
    out = ops.gemm_fp8_mx(a8, w8, sw, a_mx=a_mx, bias=bias, residual=R, out_bf16=obf)
    a, _ = _mx_decode(a8, a_mx)
    w = w8.view(torch.float8_e4m3fn).float() * sw.view(-1, 1)
    ref = (a.double() @ w.double().T).float() + bias
    if R is not None:
        ref = ref + R
    tol = (2e-2 if obf else 2e-4) * ref.abs().max().item()
    assert (out.float() - ref).abs().max().item() < tol


def test_fp8_mx_chain_matches_row_scaled_chain():
    """fc -> MX hand-off -> proj (what the tower's MLP does in SG_PREC_FP8) against the f32 MLP: the block scales are finer than the
    per-row ones, so the error must not exceed the row-scaled chain's."""
    from clip_decontamination_amd import ops
    M, D, H = 2055, 1024, 4096
    x = rnd(M, D, seed=11).to(DEV)
    W1, b1 = (rnd(H, D, seed=12) * D ** -0.5).to(DEV), rnd(H, seed=13).to(DEV) * 0.1
    W2, b2 = (rnd(D, H, seed=14) * H ** -0.5).to(DEV), rnd(D, seed=15).to(DEV) * 0.1
    x8, sx = ops.quantize_rows_fp8(x)
    w18, s1 = ops.quantize_rows_fp8(W1)
    w28, s2 = ops.quantize_rows_fp8(W2)
    h8, hs = ops.gemm_fp8_mx(x8, w18, s1, sa=sx, bias=b1, act=1, mx_out=True)
    y_mx = ops.gemm_fp8_mx(h8, w28, s2, a_mx=hs, bias=b2)
    h = ops.gemm_fp8_mx(x8, w18, s1, sa=sx, bias=b1, act=1)
    hq, sh = ops.quantize_rows_fp8(h)
    y_row = ops.gemm_fp8_mx(hq, w28, s2, sa=sh, bias=b2)
    hx = x @ W1.T + b1
    exact = (hx * torch.sigmoid(1.702 * hx)) @ W2.T + b2
    e_mx = (y_mx - exact).abs().mean().item()
    e_row = (y_row - exact).abs().mean().item()
    print(f"MLP chain mean |err|: MX hand-off {e_mx:.3e}, row-scaled {e_row:.3e}")
    assert e_mx <= e_row * 1.05


def _tower(name, prec):
    from clip_decontamination_amd.engine import HipVisionTower, HipCLIP
    cfg = Wt.vit_config(name)
    return cfg, HipCLIP(HipVisionTower(cfg, Wt.make_vit_weights(cfg, seed=0), precision=prec, device=DEV))


@pytest.mark.parametrize("mt", ["SegEarth", "Experimental", "vanilla"])
def test_tiny_tower_fp8_vs_oracle(mt):
    """tiny-16 (width 128, 5 layers): fp8 linears in the 4 ordinary blocks; token features vs the fp32 oracle."""
    cfg, net = _tower("tiny-16", "fp8")
    img = rnd(2, 3, 64, 64, seed=7)
    cls, tok = net.encode_image(img.to(DEV), mt, True, output_cls_token=True)
    with torch.no_grad():
        rc, rt = OV.vit_forward(OV.to_torch(Wt.make_vit_weights(cfg, seed=0)), cfg, img, mt, True)
    cos = F.cosine_similarity(tok.cpu(), rt, dim=-1)
    print(f"[fp8 tiny-16 {mt}] token cosine vs fp32 oracle: min {cos.min().item():.4f} mean {cos.mean().item():.4f}")
    assert cos.min().item() > 0.97 and cos.mean().item() > 0.99


def test_config5_h14_fp8_cross_tile_fusion_xbd():
    """BASELINE configs[4] with its fp8: ViT-H/14 (31 ordinary blocks on fp8 MFMA), xBD 2 queries, CrossTileFusion over 2 x 2 tiles.
    Compared with the fp32 oracle composition; agreement is reported (no reference behaviour exists for fp8, R11)."""
    import segmentor
    from oracle import refine as OR
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    names = os.path.join(ROOT, "configs", "cls_xBD.txt")
    words, qidx = segmentor.get_cls_idx(names)
    E = Wt.vit_config("ViT-H/14").embed_dim
    text = torch.from_numpy(Wt.make_text_features(len(words), E))
    ctf = dict(fusion_mode="weighted", cache_boundary_width=2, fusion_strength=0.3)
    seg = segmentor.SegmentorEx(clip_type="CLIP", vit_type="ViT-H/14", name_path=names, device=torch.device(DEV), precision="fp8",
                                synthetic_ok=True, text_features=text, model_type="SegEarth", global_debias_factor=0.2, prob_thd=0.0,
                                slide_crop=224, slide_stride=224, apply_sim_feat_up=False, cross_tile_fusion_cfg=ctf)
    cfg = seg.net.visual.cfg
    u8 = Wt.make_tiles_u8(1, 448, seed=25, smooth=True)
    img = torch.from_numpy(Wt.normalize_tiles(u8))
    logits = seg.forward_slide(img.to(DEV), [dict(ori_shape=(448, 448))], 224, 224)
    w = OV.to_torch(Wt.make_vit_weights(cfg, seed=0))
    o = OS.SegOracle(cfg, w, text, torch.tensor(qidx), model_type="SegEarth", global_debias_factor=0.2, slide_crop=224, slide_stride=224)
    fus = OR.CrossTileFusionOracle("weighted", 2, 0.3)
    g = 224 // cfg.patch
    canvas = torch.zeros(1, text.shape[0], 448, 448)
    with torch.no_grad():
        for t, (y1, y2, x1, x2) in enumerate(o.tile_windows(448, 448)):
            cls, tok = OV.vit_forward(w, cfg, img[:, :, y1:y2, x1:x2], "SegEarth", True)
            tok = fus(tok, t // 2, t % 2, g, g)
            cn = cls / cls.norm(dim=-1, keepdim=True)
            fn = tok / tok.norm(dim=-1, keepdim=True)
            tok = tok - cn.unsqueeze(1) * ((fn * cn.unsqueeze(1)).sum(-1, keepdim=True) * 0.2)
            tok = tok / tok.norm(dim=-1, keepdim=True)
            lg = (tok @ text.T).permute(0, 2, 1).reshape(1, -1, g, g)
            canvas[:, :, y1:y2, x1:x2] = F.interpolate(lg, size=(224, 224), mode="bilinear")
    pred = seg.postprocess_result(logits, None).cpu()
    ref_pred = o.postprocess(canvas[0])[1]
    err = (logits.cpu() - canvas).abs().max().item()
    agree = (pred == ref_pred).float().mean().item()
    print(f"[config5 H/14 fp8 + cross-tile fusion] max|dlogit| = {err:.2e}, label agreement = {agree:.4f}")
    assert err < 6e-3 and agree > 0.96                      # measured (r2): 2.6e-3, 0.981
