"""BASELINE.json `configs` 2-5 as parity cases at the REAL model shapes, through the drop-in classes, against the CPU oracle
(oracle/, pinned to the reference by tests/golden/*) on the same seeded inputs.  Scenes are cut down to a few tiles so the
oracle finishes in seconds on the GPU box's host cores; the kernels, shapes per tile and code paths are the configs' own.

Tolerances: f32 parity mode and the two-plane f16 mode (f16x2: f32-grade arithmetic on the f16 matrix pipe)
            max|dlogit| < 1e-3 and label maps identical up to fp32 ties of the oracle (north_star);
            bf16 / f16 modes bounded at about 2x the measured worst case (HALF_TOL below; measured values are printed).
config 5 names fp8: its fp8 run (SG_PREC_FP8: QKV / fc / proj linears on v_mfma_scale_f32_16x16x128_f8f6f4) lives in
tests/test_gpu_fp8.py; here the same case runs in f32 / bf16 / f16."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from clip_decontamination_amd import weights as Wt
from oracle import segment as OS, vit as OV, refine as OR          # checker only

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = dict(similarity_weight=1.0, temperature=1.0, add_self_similarity=True)


# throughput modes: (max|dlogit|, label agreement) bounds = about 2x the worst measured over configs 2-5 (values printed per case)
# measured (r2): bf16 max|dlogit| <= 1.4e-3, agreement >= 0.9947;  f16 <= 1.1e-4 (3.3e-4 with the bf16 JBU of config 4), >= 0.9992
HALF_TOL = {"bf16": (3e-3, 0.989), "f16": (7e-4, 0.998)}


def names(f):
    return os.path.join(ROOT, "configs", f)


def scene(H, W, seed):
    u8 = Wt.make_tiles_u8(1, max(H, W), seed=seed, smooth=True)[:, :H, :W]
    return torch.from_numpy(Wt.normalize_tiles(u8))               # [1,3,H,W] f32


def setup_module(_m):
    torch.set_num_threads(min(16, os.cpu_count() or 1))


def build(cls_name, vit, name_file, precision, **kw):
    import segmentor, segearth_segmentor
    cls = segmentor.SegmentorEx if cls_name == "SegmentorEx" else segearth_segmentor.Segmentor
    cfg = Wt.vit_config(vit)
    words, qidx = segmentor.get_cls_idx(names(name_file))
    text = torch.from_numpy(Wt.make_text_features(len(words), cfg.embed_dim))
    seg = cls(clip_type="CLIP", vit_type=vit, name_path=names(name_file), device=torch.device("cuda:0"), precision=precision,
              synthetic_ok=True, text_features=text, **kw)
    return seg.net.visual.cfg, seg, text, torch.tensor(qidx)      # the architecture the drop-in resolved (activation included)


def compare(seg, logits, ref_logits, oracle, tag, prec):
    pred = seg.postprocess_result(logits, None).cpu()
    ref_pred = oracle.postprocess(ref_logits[0])[1]
    err = (logits.cpu() - ref_logits).abs().max().item()
    agree = (pred == ref_pred).float().mean().item()
    print(f"[{tag} {prec}] max|dlogit| = {err:.2e}, label agreement = {agree:.4f}")
    if prec in ("f32", "f16x2"):                               # the two exact modes: the north_star bar itself
        assert err < 1e-3
        if agree != 1.0:
            # labels may differ only where the ORACLE itself is at a tie within the fp32 noise of the logits (two classes, or
            # the arg-max probability against prob_thd, closer than 1e-4 in probability): an arg-max there is not defined
            # to better than the rounding of either side.
            probs = oracle.postprocess(ref_logits[0])[0]
            bad = (pred != ref_pred)[0]
            assert bad.float().mean().item() < 1e-4
            pmax = probs.max(0)[0][bad]
            ours = probs[:, bad].gather(0, pred[0][bad][None])[0]
            tie = (pmax - ours).abs() < 1e-4
            thd = (pmax - oracle.prob_thd).abs() < 1e-4
            assert bool((tie | thd).all()), "label mismatch away from a tie"
    else:
        tol_err, tol_agree = HALF_TOL[prec]
        assert err < tol_err and agree >= tol_agree


@pytest.mark.parametrize("prec", ["f32", "f16x2", "bf16", "f16"])
def test_config2_b16_512_slide_potsdam(prec):
    """configs[1]: ViT-B/16, 512 tiles at stride 256, Potsdam (8 queries / 6 classes), the shipped refiner stack."""
    kw = dict(model_type="Experimental", global_debias_factor=0.2, apply_similarity_enhancement=True, similarity_enhancement_cfg=SIM,
              apply_outlier_suppression=True, outlier_suppression_cfg=dict(top_k=30), prob_thd=0.1, bg_idx=5, slide_crop=512,
              slide_stride=256, apply_sim_feat_up=False)
    cfg, seg, text, qidx = build("SegmentorEx", "ViT-B/16", "cls_potsdam.txt", prec, **kw)
    img = scene(768, 768, 21)
    logits = seg.forward_slide(img.cuda(), [dict(ori_shape=(768, 768))], 256, 512)
    o = OS.SegOracle(cfg, OV.to_torch(Wt.make_vit_weights(cfg, seed=0)), text, qidx, model_type="Experimental", global_debias_factor=0.2,
                     similarity_cfg=SIM, outlier_cfg=dict(top_k=30), prob_thd=0.1, bg_idx=5, slide_crop=512, slide_stride=256)
    with torch.no_grad():
        ref = o.forward_slide(img)
    compare(seg, logits, ref, o, "config2 B/16 512-slide", prec)


@pytest.mark.parametrize("prec", ["f32", "f16x2", "bf16", "f16"])
def test_config3_l14_gem_loveda(prec):
    """configs[2], GEM half: ViT-L/14 through segearth_segmentor.Segmentor(model_type='GEM') (the only reference class where GEM
    runs), LoveDA 9 queries / 7 classes."""
    cfg, seg, text, qidx = build("Segmentor", "ViT-L/14", "cls_loveda.txt", prec, model_type="GEM", cls_token_lambda=0.0,
                                 slide_crop=224, slide_stride=112, apply_sim_feat_up=False, prob_thd=0.3)
    img = scene(224, 336, 22)                                       # 1 x 2 tiles
    logits = seg.forward_slide(img.cuda(), [dict(ori_shape=(224, 336))], 112, 224)
    o = OS.SegOracle(cfg, OV.to_torch(Wt.make_vit_weights(cfg, seed=0)), text, qidx, model_type="GEM", prob_thd=0.3, slide_crop=224,
                     slide_stride=112, segearth_variant=True)
    with torch.no_grad():
        ref = o.forward_slide(img)
    compare(seg, logits, ref, o, "config3 L/14 GEM", prec)


@pytest.mark.parametrize("prec", ["f32", "f16x2", "bf16", "f16"])
def test_config3_l14_gem_plus_outlier_loveda(prec):
    """configs[2] as ONE forward: ViT-L/14, GEM dual-stream blocks AND OutlierSuppressionModule(top_k=30), LoveDA.  The reference cannot run
    this composition; the definition (detection on the ordinary stream's head-averaged attention of
    block L-2, suppression on the GEM stream before ln_post) is restated in oracle/vit.py::gem_forward from the two stages that ARE
    pinned to the reference -- end-to-end parity of the composition is therefore 'unpinned', the per-stage pins are the fixtures."""
    cfg, seg, text, qidx = build("Segmentor", "ViT-L/14", "cls_loveda.txt", prec, model_type="GEM", cls_token_lambda=0.0,
                                 slide_crop=224, slide_stride=112, apply_sim_feat_up=False, prob_thd=0.3,
                                 apply_outlier_suppression=True, outlier_suppression_cfg=dict(top_k=30))
    img = scene(224, 336, 26)                                       # 1 x 2 tiles
    logits = seg.forward_slide(img.cuda(), [dict(ori_shape=(224, 336))], 112, 224)
    o = OS.SegOracle(cfg, OV.to_torch(Wt.make_vit_weights(cfg, seed=0)), text, qidx, model_type="GEM", prob_thd=0.3, slide_crop=224,
                     slide_stride=112, segearth_variant=True, outlier_cfg=dict(top_k=30))
    with torch.no_grad():
        ref = o.forward_slide(img)
        plain = OS.SegOracle(cfg, OV.to_torch(Wt.make_vit_weights(cfg, seed=0)), text, qidx, model_type="GEM", prob_thd=0.3, slide_crop=224,
                             slide_stride=112, segearth_variant=True).forward_slide(img)
    assert (ref - plain).abs().max().item() > 1e-3                  # the suppression does change the map: the composition is not a no-op
    compare(seg, logits, ref, o, "config3 L/14 GEM + outlier k=30", prec)


@pytest.mark.parametrize("prec", ["f32", "f16x2", "bf16", "f16"])
def test_config3_l14_outlier_loveda(prec):
    """configs[2], outlier-suppression half: ViT-L/14 SegmentorEx + OutlierSuppressionModule(top_k=30), LoveDA."""
    kw = dict(model_type="SegEarth", global_debias_factor=0.2, apply_outlier_suppression=True, outlier_suppression_cfg=dict(top_k=30),
              prob_thd=0.3, slide_crop=224, slide_stride=112, apply_sim_feat_up=False)
    cfg, seg, text, qidx = build("SegmentorEx", "ViT-L/14", "cls_loveda.txt", prec, **kw)
    img = scene(224, 336, 23)
    logits = seg.forward_slide(img.cuda(), [dict(ori_shape=(224, 336))], 112, 224)
    o = OS.SegOracle(cfg, OV.to_torch(Wt.make_vit_weights(cfg, seed=0)), text, qidx, model_type="SegEarth", global_debias_factor=0.2,
                     outlier_cfg=dict(top_k=30), prob_thd=0.3, slide_crop=224, slide_stride=112)
    with torch.no_grad():
        ref = o.forward_slide(img)
    compare(seg, logits, ref, o, "config3 L/14 outlier k=30", prec)


@pytest.mark.parametrize("prec", ["f32", "f16x2", "bf16", "f16"])
def test_config4_l14_jbu_isaid(prec):
    """configs[3]: ViT-L/14 + SimFeatUp JBU (jbu_one, the shipped base config), iSAID 16 queries; per-pixel logits."""
    kw = dict(model_type="SegEarth", global_debias_factor=0.2, prob_thd=0.4, slide_crop=224, slide_stride=112, apply_sim_feat_up=True,
              sim_feat_up_cfg=dict(model_name="jbu_one", model_path=None))
    cfg, seg, text, qidx = build("SegmentorEx", "ViT-L/14", "cls_isaid.txt", prec, **kw)
    img = scene(224, 224, 24)
    logits = seg.forward_slide(img.cuda(), [dict(ori_shape=(224, 224))], 112, 224)
    o = OS.SegOracle(cfg, OV.to_torch(Wt.make_vit_weights(cfg, seed=0)), text, qidx, model_type="SegEarth", global_debias_factor=0.2,
                     jbu_weights=OV.to_torch(Wt.make_jbu_weights("jbu_one", cfg.embed_dim, seed=3)), prob_thd=0.4, slide_crop=224,
                     slide_stride=112)
    with torch.no_grad():
        ref = o.forward_slide(img)
    compare(seg, logits, ref, o, "config4 L/14 + JBU", prec)


@pytest.mark.parametrize("prec", ["f32", "f16x2", "bf16", "f16"])
def test_config5_h14_cross_tile_fusion_xbd(prec):
    """configs[4] without its fp8 (not built): ViT-H/14 (erf-GELU, head dim 80), xBD 2 queries, CrossTileFusion('weighted') over a
    2 x 2 tile scene.  Oracle = per-tile oracle tokens -> the reference module's sequential semantics -> logits -> stitch."""
    ctf = dict(fusion_mode="weighted", cache_boundary_width=2, fusion_strength=0.3)
    kw = dict(model_type="SegEarth", global_debias_factor=0.2, prob_thd=0.0, slide_crop=224, slide_stride=224, apply_sim_feat_up=False,
              cross_tile_fusion_cfg=ctf)
    cfg, seg, text, qidx = build("SegmentorEx", "ViT-H/14", "cls_xBD.txt", prec, **kw)
    img = scene(448, 448, 25)
    logits = seg.forward_slide(img.cuda(), [dict(ori_shape=(448, 448))], 224, 224)
    w = OV.to_torch(Wt.make_vit_weights(cfg, seed=0))
    o = OS.SegOracle(cfg, w, text, qidx, model_type="SegEarth", global_debias_factor=0.2, slide_crop=224, slide_stride=224)
    fus = OR.CrossTileFusionOracle("weighted", 2, 0.3)
    g = 224 // cfg.patch
    canvas = torch.zeros(1, text.shape[0], 448, 448)
    with torch.no_grad():
        for t, (y1, y2, x1, x2) in enumerate(o.tile_windows(448, 448)):
            cls, tok = OV.vit_forward(w, cfg, img[:, :, y1:y2, x1:x2], "SegEarth", True)
            tok = fus(tok, t // 2, t % 2, g, g)
            cn = cls / cls.norm(dim=-1, keepdim=True)
            fn = tok / tok.norm(dim=-1, keepdim=True)
            tok = tok - cn.unsqueeze(1) * ((fn * cn.unsqueeze(1)).sum(-1, keepdim=True) * 0.2)       # segmentor.py:322-336
            tok = tok / tok.norm(dim=-1, keepdim=True)
            lg = (tok @ text.T).permute(0, 2, 1).reshape(1, -1, g, g)
            canvas[:, :, y1:y2, x1:x2] = F.interpolate(lg, size=(224, 224), mode="bilinear")
    compare(seg, logits, canvas, o, "config5 H/14 + cross-tile fusion", prec)


@pytest.mark.parametrize("prec", ["f32", "f16x2"])
def test_jbu_after_cross_tile_fusion_vs_oracle(prec):
    """tokens -> CrossTileFusion -> global debias -> JBU -> per-pixel cosine logits -> stitch (segmentor.py:368-372 behind
    cross_tile_fusion.py:238-320; the reference never calls the fusion, R2 -- the composition is the build's wiring of two pinned stages).
    A patch-16 tower (the upsampler is 16x), 2 x 2 tiles."""
    from clip_decontamination_amd.engine import HipVisionTower, HipCLIP
    from clip_decontamination_amd.pipeline import SegPipeline
    from clip_decontamination_amd.upsampler import HipJBU
    from oracle import jbu as OJ
    cfg = Wt.vit_config("tiny-16")
    wnp = Wt.make_vit_weights(cfg, seed=0)
    jnp = Wt.make_jbu_weights("jbu_stack", cfg.embed_dim, seed=3)
    qidx = [0, 0, 1, 2, 3, 4, 5, 5]
    text = torch.from_numpy(Wt.make_text_features(len(qidx), cfg.embed_dim))
    up = HipJBU("jbu_stack", cfg.embed_dim, "cuda:0", prec)
    up.load_state_dict(jnp)
    ctf = dict(fusion_mode="weighted", cache_boundary_width=1, fusion_strength=0.4)
    pipe = SegPipeline(HipCLIP(HipVisionTower(cfg, wnp, precision=prec, device="cuda:0")), text, torch.tensor(qidx), model_type="SegEarth",
                       global_debias_factor=0.2, prob_thd=0.1, bg_idx=5, cross_tile_fusion=ctf, upsampler=up)
    S, g = 48, 48 // cfg.patch
    img = torch.from_numpy(np.random.default_rng(31).standard_normal((1, 3, 2 * S, 2 * S), dtype=np.float32))
    out = pipe.forward_slide(img[0].cuda(), S, S).cpu()
    w, jw = OV.to_torch(wnp), OV.to_torch(jnp)
    fus = OR.CrossTileFusionOracle("weighted", 1, 0.4)
    canvas = torch.zeros(1, len(qidx), 2 * S, 2 * S)
    with torch.no_grad():
        for t in range(4):
            y1, x1 = (t // 2) * S, (t % 2) * S
            tile = img[:, :, y1:y1 + S, x1:x1 + S]
            cls, tok = OV.vit_forward(w, cfg, tile, "SegEarth", True)
            tok = fus(tok, t // 2, t % 2, g, g)
            cn = cls / cls.norm(dim=-1, keepdim=True)
            fn = tok / tok.norm(dim=-1, keepdim=True)
            tok = tok - cn.unsqueeze(1) * ((fn * cn.unsqueeze(1)).sum(-1, keepdim=True) * 0.2)       # segmentor.py:322-336
            src = tok.permute(0, 2, 1).reshape(1, -1, g, g)
            feats = OJ.jbu_forward(jw, src, tile)                                                   # [1, E, 16 g, 16 g] = the tile size
            f = feats.reshape(1, feats.shape[1], -1).permute(0, 2, 1)
            f = f / f.norm(dim=-1, keepdim=True)
            canvas[:, :, y1:y1 + S, x1:x1 + S] = (f @ text.T).permute(0, 2, 1).reshape(1, -1, S, S)
    err = (out - canvas).abs().max().item()
    print(f"[{prec}] JBU after cross-tile fusion: max|dlogit| = {err:.2e}")
    assert err < 1e-3


def test_sharded_cross_tile_steps_equal_single_call():
    """sg_cross_tile_pack / _fuse / _apply driven as three simulated ranks (strips concatenated instead of all-gathered)
    == sg_cross_tile_fusion over the whole scene == the oracle's sequential semantics."""
    from clip_decontamination_amd import ops
    from clip_decontamination_amd.pipeline import partition
    hg, wg, gp, C = 3, 4, 9, 40
    tok = torch.from_numpy(np.random.default_rng(3).standard_normal((hg * wg, gp * gp, C)).astype(np.float32)).cuda()
    T, world = hg * wg, 3
    for mode in ("weighted", "attention"):
        whole = ops.cross_tile_fusion(tok, hg, wg, gp, gp, 2, mode, 0.5)
        steps = ops.CrossTileSteps(gp, gp, C, 2, mode, 0.5, wg)
        parts = [partition(T, world, r) for r in range(world)]
        local = [tok[a:b].clone() for a, b in parts]
        right_all = torch.cat([steps.pack(local[r], parts[r][0], 0) for r in range(world)], 0)
        left = [steps.fuse(local[r], parts[r][0], right_all, 0) for r in range(world)]
        bottom_all = torch.cat([steps.pack(local[r], parts[r][0], 1, left[r]) for r in range(world)], 0)
        top = [steps.fuse(local[r], parts[r][0], bottom_all, 1) for r in range(world)]
        out = torch.cat([steps.apply(local[r], parts[r][0], left[r], top[r]) for r in range(world)], 0)
        assert torch.equal(out, whole)
        o = OR.CrossTileFusionOracle(mode, 2, 0.5)
        ref = torch.stack([o(tok[t:t + 1].cpu().clone(), t // wg, t % wg, gp, gp)[0] for t in range(T)], 0)
        assert (out.cpu() - ref).abs().max().item() < 3e-5
