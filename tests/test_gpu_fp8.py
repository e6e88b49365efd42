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
