"""TEST INFRASTRUCTURE ONLY.  Mints the fixtures under tests/golden/ by running the REFERENCE
(imported read-only from /root/reference through oracle/refimport.py) on the build's
deterministic synthetic weights and inputs.  Runs only in the build container; the fixtures are
data (inputs' seeds + expected outputs), never reference source.

    python -m oracle.gen_golden [--only tiny|layer_fusion|refine|jbu|jbu_real|segment|text|ctd|real]

Every fixture is cross-checked here against the build's own CPU restatement (oracle/*.py) so a
drift between the two fails at mint time, and again in tests/test_oracle_vs_golden.py.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
import zlib

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from clip_decontamination_amd import weights as Wt   # noqa: E402
from oracle import refimport as R                    # noqa: E402
from oracle import vit as OV, refine as OR, jbu as OJ, segment as OS   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
MODEL_TYPES = ["vanilla", "MaskCLIP", "ClearCLIP", "SCLIP", "SegEarth", "SFP", "Experimental", "NACLIP", "NOnly", "GAV"]


def save(name, **arrs):
    os.makedirs(GOLD, exist_ok=True)
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def close(a, b, tol, what):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    err = (a.float() - b.float()).abs().max().item()
    print(f"    restatement vs reference [{what}]: max|d| = {err:.3e}")
    assert err <= tol, f"{what}: {err} > {tol}"


def rand_img(seed, B, H, Wd):
    g = np.random.default_rng(seed)
    return torch.from_numpy(g.standard_normal((B, 3, H, Wd), dtype=np.float32))


# ---------------------------------------------------------------------------------------------
def ref_clip(cfg, wnp):
    """Reference ``CLIP`` (open_clip/model.py) whose ``visual`` carries the build's weights."""
    M = R.ref("open_clip.model")
    vis = M.CLIPVisionCfg(layers=cfg.layers, width=cfg.width, head_width=cfg.head_dim, patch_size=cfg.patch,
                          image_size=cfg.image_size, mlp_ratio=cfg.mlp_ratio)
    txt = M.CLIPTextCfg(context_length=8, vocab_size=64, width=32, heads=2, layers=1)
    net = M.CLIP(cfg.embed_dim, vis, txt, quick_gelu=cfg.quick_gelu)
    net.visual.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in wnp.items()}, strict=True)
    return net.eval()


def install_refiners(net, similarity_cfg=None, outlier_cfg=None, self_attn_cfg=None):
    """What reference segmentor.py:196-274 does to ``net.visual``."""
    for a in ("similarity_enhancer", "outlier_suppressor", "self_attn_enhancer"):
        if hasattr(net.visual, a):
            delattr(net.visual, a)
    if similarity_cfg is not None:
        net.visual.similarity_enhancer = R.ref("similarity_enhancement").SimilarityEnhancementModule(**similarity_cfg)
    if outlier_cfg is not None:
        net.visual.outlier_suppressor = R.ref("outlier_suppression").OutlierSuppressionModule(**outlier_cfg)
    if self_attn_cfg is not None:
        net.visual.self_attn_enhancer = R.ref("self_attention_enhancement").SelfAttentionEnhancementModule(**self_attn_cfg)


def ref_segmentor(cls_name, net, cfg, text, query_idx, **kw):
    """A reference segmentor object without running its ``__init__`` (which would resolve
    pretrained tags); attributes set exactly as ``__init__`` would (segmentor.py:131-191)."""
    mod = R.ref("segmentor" if cls_name == "SegmentorEx" else "segearth_segmentor")
    cls = getattr(mod, cls_name)
    s = cls.__new__(cls)
    torch.nn.Module.__init__(s)
    s.net = net
    s.clip_type = "CLIP"
    s.vit_type = cfg.name
    s.model_type = kw.get("model_type", "SegEarth")
    s.apply_sim_feat_up = False
    s.cls_token_lambda = kw.get("cls_token_lambda", 0.0)
    s.output_cls_token = s.cls_token_lambda != 0
    s.global_debias_factor = kw.get("global_debias_factor", 0.0)
    s.bg_idx = kw.get("bg_idx", 0)
    s.patch_size = net.visual.patch_size
    s.query_idx = torch.as_tensor(query_idx, dtype=torch.int64)
    s.num_queries = len(query_idx)
    s.num_classes = max(query_idx) + 1
    s.query_features = torch.as_tensor(text)
    s.dtype = torch.float32
    s.ignore_residual = kw.get("ignore_residual", True)
    s.logit_scale = kw.get("logit_scale", 50)
    s.prob_thd = kw.get("prob_thd", 0.0)
    s.slide_stride = kw.get("slide_stride", 112)
    s.slide_crop = kw.get("slide_crop", 224)
    s.apply_ctd = kw.get("apply_ctd", False)
    s.apply_layer_fusion = False
    s.layer_fusion_lambda = 0.5
    s.layer_fusion_threshold = 0.7
    s.apply_similarity_enhancement = kw.get("similarity_cfg") is not None
    s.apply_self_attn_enhancement = kw.get("self_attn_cfg") is not None
    s.apply_outlier_suppression = kw.get("outlier_cfg") is not None
    s.result_dir = s.heatmap_dir = None
    if cls_name == "SegmentorEx":
        install_refiners(net, kw.get("similarity_cfg"), kw.get("outlier_cfg"), kw.get("self_attn_cfg"))
    return s.eval()


# ---------------------------------------------------------------------------------------------
def gen_tiny():
    """Tiny ViT: every model_type, refiner hooks, bicubic pos-embed branch, both activations."""
    print("[tiny] vision tower")
    for cname in ("tiny-8", "tiny-gelu"):
        cfg = Wt.vit_config(cname)
        wnp = Wt.make_vit_weights(cfg, seed=0)
        w = OV.to_torch(wnp)
        net = ref_clip(cfg, wnp)
        img = rand_img(11, 2, 48, 48)              # g = 6 != g0 = 4 -> bicubic pos-embed branch
        img_native = rand_img(12, 1, 32, 32)
        out = {"img": img, "img_native": img_native}
        with torch.no_grad():
            for mt in (MODEL_TYPES if cname == "tiny-8" else ["SegEarth"]):
                install_refiners(net)
                im = img[:1] if mt in ("NOnly",) else img   # reference NOnly is batch-1 only (transformer.py:924)
                cls, tok = net.encode_image(im, mt, True, output_cls_token=True)
                ocls, otok = OV.vit_forward(w, cfg, im, mt, True)
                close(otok, tok, 2e-5, f"{cname}/{mt}/tokens")
                close(ocls, cls, 2e-5, f"{cname}/{mt}/cls")
                out[f"{mt}.cls"], out[f"{mt}.tokens"] = cls, tok
            # residual kept in the last block (ignore_residual=False)
            cls, tok = net.encode_image(img, "SegEarth", False, output_cls_token=True)
            ocls, otok = OV.vit_forward(w, cfg, img, "SegEarth", False)
            close(otok, tok, 2e-5, f"{cname}/SegEarth+residual")
            out["SegEarth.res.cls"], out["SegEarth.res.tokens"] = cls, tok
            cls, tok = net.encode_image(img_native, "SegEarth", True, output_cls_token=True)
            ocls, otok = OV.vit_forward(w, cfg, img_native, "SegEarth", True)
            close(otok, tok, 2e-5, f"{cname}/native grid")
            out["native.cls"], out["native.tokens"] = cls, tok
            if cname == "tiny-8":
                sim_cfg = dict(similarity_weight=1.0, temperature=1.0, add_self_similarity=True)
                out_cfg = dict(top_k=5)
                sa_cfg = dict(enhancement_strength=0.1, min_self_attn_threshold=0.15, mode="feature", top_k=4)
                combos = {
                    "sim": (sim_cfg, None, None), "out": (None, out_cfg, None),
                    "sim_out": (sim_cfg, out_cfg, None), "all": (sim_cfg, out_cfg, sa_cfg),
                    "sa_only": (None, None, sa_cfg),
                    "sa_attn": (sim_cfg, out_cfg, dict(enhancement_strength=0.3, min_self_attn_threshold=0.15, mode="attention", top_k=4)),
                    "sim2": (dict(similarity_weight=0.5, temperature=2.0, add_self_similarity=False), None, None),
                }
                for tag, (sc, oc, ac) in combos.items():
                    for mt in ("SegEarth", "Experimental", "ClearCLIP"):
                        install_refiners(net, sc, oc, ac)
                        cls, tok = net.encode_image(img, mt, True, output_cls_token=True,
                                                    apply_similarity_enhancement=sc is not None)
                        cap = {}
                        ocls, otok = OV.vit_forward(w, cfg, img, mt, True, similarity_cfg=sc, outlier_cfg=oc,
                                                    self_attn_cfg=ac, capture=cap)
                        close(otok, tok, 3e-5, f"{tag}/{mt}/tokens")
                        out[f"{tag}.{mt}.cls"], out[f"{tag}.{mt}.tokens"] = cls, tok
                        if tag == "sim_out" and mt == "Experimental":
                            # intermediates from the restatement (already shown equal end-to-end)
                            out["inter.attn_cls_row"] = cap["attn"][:, 0]
                            out["inter.attn_diag"] = torch.diagonal(cap["attn"], dim1=-2, dim2=-1)
                            out["inter.sim"] = cap["sim"]
                            out["inter.x_pre_last"] = cap["x_pre_last"]
                            out["inter.last_out"] = cap["last_out"]
                            out["inter.outlier_idx"] = cap["outlier_idx"]
                            out["inter.refined"] = cap["refined"]
                install_refiners(net)
        save(f"vit_{cname}", **out)

    print("[tiny] GEM")
    cfg = Wt.vit_config("tiny-gem")
    wnp = Wt.make_vit_weights(cfg, seed=0)
    w = OV.to_torch(wnp)
    out = {}
    with torch.no_grad():
        for ign in (True, False):
            net = ref_clip(cfg, wnp)
            gw = R.ref("gem.gem_wrapper").GEMWrapper(model=net, tokenizer=None, depth=7, ignore_residual=ign)
            for nm, im in (("g6", rand_img(21, 2, 48, 48)), ("g4", rand_img(22, 1, 32, 32))):
                tok = gw.model.visual(im)
                otok = OV.gem_forward(w, cfg, im, ign, 7)
                close(otok, tok, 3e-5, f"gem ign={ign} {nm}")
                out[f"{nm}.ign{int(ign)}.tokens"] = tok
                out[f"{nm}.img"] = im
    save("vit_tiny-gem", **out)


def gen_layer_fusion():
    """apply_layer_fusion (transformer.py:598-607,630-637,647-690).  With an outlier suppressor installed the reference's
    ``attn_accumulated.view(N, num_heads, L, L)`` only works when heads == 1 (the attention it accumulates is already head-averaged,
    R9): a ONE-head tiny tower pins the semantics; without a suppressor the fusion is a no-op on any head count (pinned on tiny-8)."""
    print("[layer_fusion] reference apply_layer_fusion=True")
    out = {}
    with torch.no_grad():
        cfg = Wt.vit_config("tiny-1h")
        wnp = Wt.make_vit_weights(cfg, seed=0)
        w = OV.to_torch(wnp)
        net = ref_clip(cfg, wnp)
        img = rand_img(31, 2, 48, 48)
        out["img"] = img
        sim_cfg = dict(similarity_weight=1.0, temperature=1.0, add_self_similarity=True)
        for tag, sc, lam, ign in (("lf", None, 0.5, True), ("lf_sim", sim_cfg, 0.3, True), ("lf_res", None, 0.5, False)):
            for mt in ("SegEarth", "Experimental"):
                install_refiners(net, sc, dict(top_k=5), None)
                cls, tok = net.encode_image(img, mt, ign, output_cls_token=True, apply_layer_fusion=True, layer_fusion_lambda=lam,
                                            apply_similarity_enhancement=sc is not None)
                cap = {}
                ocls, otok = OV.vit_forward(w, cfg, img, mt, ign, similarity_cfg=sc, outlier_cfg=dict(top_k=5), layer_fusion={"lambda": lam},
                                            capture=cap)
                close(otok, tok, 3e-5, f"tiny-1h {tag}/{mt} tokens")
                close(ocls, cls, 3e-5, f"tiny-1h {tag}/{mt} cls")
                out[f"{tag}.{mt}.cls"], out[f"{tag}.{mt}.tokens"] = cls, tok
                if tag == "lf" and mt == "SegEarth":
                    out["inter.fused_attn"] = cap["fused_attn"]
                    out["inter.fusion_idx"] = cap["fusion_idx"]
        install_refiners(net)
        # no suppressor: fusion computed and discarded (any head count)
        cfg8 = Wt.vit_config("tiny-8")
        wnp8 = Wt.make_vit_weights(cfg8, seed=0)
        net8 = ref_clip(cfg8, wnp8)
        install_refiners(net8)
        cls, tok = net8.encode_image(img, "SegEarth", True, output_cls_token=True, apply_layer_fusion=True)
        cls0, tok0 = net8.encode_image(img, "SegEarth", True, output_cls_token=True)
        close(tok, tok0, 1e-5, "tiny-8 fusion without suppressor == no fusion (need_weights=True takes the unfused MHA path: rounding only)")
        ocls, otok = OV.vit_forward(OV.to_torch(wnp8), cfg8, img, "SegEarth", True, layer_fusion={"lambda": 0.5})
        close(otok, tok, 3e-5, "tiny-8 noop tokens")
        out["noop8.tokens"] = tok
        # heads > 1 with a suppressor: the reference raises (recorded, not a fixture)
        install_refiners(net8, None, dict(top_k=5), None)
        try:
            net8.encode_image(img, "SegEarth", True, output_cls_token=True, apply_layer_fusion=True)
            print("    reference with heads=2 + suppressor: ran (unexpected)")
        except RuntimeError as e:
            print(f"    reference with heads=2 + suppressor raises, as SURVEY R9 says: {str(e)[:90]}")
    save("vit_tiny-1h", **out)


# ---------------------------------------------------------------------------------------------
def gen_refine():
    print("[refine] modules")
    g = torch.Generator().manual_seed(5)
    out = {}
    osm = R.ref("outlier_suppression")
    B, D, gh, gw = 2, 24, 7, 7
    n = gh * gw
    grid = torch.randn(B, D, gh, gw, generator=g)
    attn = torch.softmax(torch.randn(B, n + 1, n + 1, generator=g) * 2.0, -1)
    # force corner / edge / adjacent outliers so the clamped-neighbour and write-order rules matter
    for b in range(B):
        for t in (0, 1, gw - 1, n - 1, 24, 25):
            attn[b, 0, 1 + t] += 0.5 + 0.01 * t
    idx_ref = osm.detect_outliers_by_attention(attn, n, 8)
    close(OR.detect_outliers(attn, n, 8), idx_ref, 0, "outlier indices")
    mod = osm.OutlierSuppressionModule(top_k=8)
    res = mod(grid, attn, gh, gw)
    close(OR.suppress_outliers(grid, idx_ref, 0.1), res, 1e-6, "outlier suppression")
    out.update(grid=grid, attn=attn, outlier_idx=idx_ref, suppressed=res)
    # 4-D (per-head) attention input form
    attn4 = torch.softmax(torch.randn(B, 3, n + 1, n + 1, generator=g), -1)
    idx4 = osm.detect_outliers_by_attention(attn4, n, 5)
    close(OR.detect_outliers(attn4.mean(1), n, 5), idx4, 0, "4-D attention indices")
    out.update(attn4=attn4, outlier_idx4=idx4)

    sam = R.ref("self_attention_enhancement")
    for mode in ("feature", "attention"):
        m = sam.SelfAttentionEnhancementModule(enhancement_strength=0.3, min_self_attn_threshold=0.05, mode=mode, top_k=6)
        r = m(grid, attn, gh, gw)
        o = OR.self_attention_enhance(grid, attn, 0.3, 0.05, mode, 6)
        close(o, r, 1e-5, f"self-attention enhancement [{mode}]")
        out[f"selfattn_{mode}"] = r

    sem = R.ref("similarity_enhancement")
    feats = torch.randn(2, n, 40, generator=g)
    for tag, kw in (("a", dict(similarity_weight=1.0, temperature=1.0, add_self_similarity=True)),
                    ("b", dict(similarity_weight=0.7, temperature=0.5, add_self_similarity=False))):
        m = sem.SimilarityEnhancementModule(**kw)
        s = m.compute_similarity_map(feats)
        close(OR.similarity_map(feats, kw["temperature"], kw["add_self_similarity"]), s, 1e-6, f"similarity map {tag}")
        out[f"sim_{tag}"] = s
    out["sim_feats"] = feats

    ctf = R.ref("cross_tile_fusion")
    C, pg = 16, 6
    tiles = torch.randn(2, 3, 1, pg * pg, C, generator=g)           # 2x3 tiles, raster order
    for mode in ("weighted", "attention"):
        m = ctf.CrossTileFusion(fusion_mode=mode, cache_boundary_width=2, fusion_strength=0.3)
        o = OR.CrossTileFusionOracle(mode, 2, 0.3)
        res = torch.zeros_like(tiles)
        for hi in range(2):
            for wi in range(3):
                r = m(tiles[hi, wi].clone(), hi, wi, pg, pg)
                oo = o(tiles[hi, wi].clone(), hi, wi, pg, pg)
                close(oo, r, 1e-5, f"cross-tile fusion [{mode}] tile ({hi},{wi})")
                res[hi, wi] = r
        out[f"ctf_{mode}"] = res
    out["ctf_tiles"] = tiles
    save("refine", **out)


# ---------------------------------------------------------------------------------------------
def gen_jbu():
    print("[jbu] upsamplers (reference torch form of the adaptive conv)")
    U = R.ref("simfeatup_dev.upsamplers")
    U.AdaptiveConv = type("AC", (), {"apply": staticmethod(U.adaptive_conv_py_simple)})
    out = {}
    g = torch.Generator().manual_seed(9)
    for name, C, gs in (("jbu_one", 16, 5), ("jbu_stack", 24, 4)):
        wnp = Wt.make_jbu_weights(name, C, seed=3)
        up = U.get_upsampler(name, C)
        up.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in wnp.items()}, strict=True)
        up.eval()
        src = torch.randn(1, C, gs, gs, generator=g)
        S = gs * 16
        # smooth + noise guidance so the range kernel is neither flat nor degenerate
        low = torch.randn(1, 3, 6, 6, generator=g)
        guid = F.interpolate(low, size=(S, S), mode="bicubic", align_corners=False) + 0.2 * torch.randn(1, 3, S, S, generator=g)
        with torch.no_grad():
            r = up(src, guid)
            w = OV.to_torch(wnp)
            o = OJ.jbu_forward(w, src, guid)
            close(o, r, 5e-5, f"{name} end-to-end")
            stage = up.up if name == "jbu_one" else up.up1
            small = F.adaptive_avg_pool2d(guid, (gs * 2, gs * 2))
            k_ref = stage.get_range_kernel(small)
            upn = "up" if name == "jbu_one" else "up1"
            rr = OJ.radius_of(w, upn)
            close(OJ.range_kernel(w, upn, small, rr), k_ref, 1e-5, f"{name} range kernel")
            s1 = stage(src, small)
            close(OJ.jbu_stage(w, upn, src, small), s1, 2e-5, f"{name} stage 1")
        out.update({f"{name}.src": src, f"{name}.guidance": guid, f"{name}.out": r, f"{name}.stage1": s1,
                    f"{name}.range1": k_ref, f"{name}.small1": small})
    # stand-alone adaptive conv (boundary op mirrored by sg_adaptive_conv)
    x = torch.randn(2, 5, 12 + 6, 9 + 6, generator=g)
    f = torch.randn(2, 12, 9, 7, 7, generator=g)
    r = U.adaptive_conv_py_simple(x, f)
    close(OJ.adaptive_conv(x, f), r, 1e-5, "adaptive conv")
    out.update(ac_in=x, ac_filt=f, ac_out=r)
    save("jbu", **out)


def gen_jbu_real():
    """The reference's JBUStack (dim 512) on the TRAINED COCO-Stuff checkpoint it ships (simfeatup_dev/weights/, loaded as at
    segmentor.py:281-283 but with weights_only=True): real range_temp / sigma_spatial / fixup weights exercise the clamp and
    softmax regimes synthetic weights do not (upsamplers.py:230-275).  The fixture stores the weights (data) + input + output."""
    print("[jbu_real] JBUStack(512) with the shipped clip_jbu_stack_cocostuff.ckpt")
    U = R.ref("simfeatup_dev.upsamplers")
    U.AdaptiveConv = type("AC", (), {"apply": staticmethod(U.adaptive_conv_py_simple)})
    ck = torch.load("/root/reference/simfeatup_dev/weights/clip_jbu_stack_cocostuff.ckpt", map_location="cpu", weights_only=True)["state_dict"]
    sd = {k[10:]: v.float() for k, v in ck.items() if k.startswith("upsampler.")}
    C, gs = 512, 4
    up = U.get_upsampler("jbu_stack", C)
    up.load_state_dict(sd, strict=True)
    up.eval()
    g = torch.Generator().manual_seed(21)
    src = torch.randn(1, C, gs, gs, generator=g)
    S = gs * 16
    low = torch.randn(1, 3, 5, 5, generator=g)
    guid = F.interpolate(low, size=(S, S), mode="bicubic", align_corners=False) + 0.3 * torch.randn(1, 3, S, S, generator=g)
    with torch.no_grad():
        r = up(src, guid)
        o = OJ.jbu_forward(sd, src, guid)
        close(o, r, 2e-4 * float(r.abs().max()), "jbu_stack real weights end-to-end")
        small = F.adaptive_avg_pool2d(guid, (gs * 2, gs * 2))
        k_ref = up.up1.get_range_kernel(small)
        close(OJ.range_kernel(sd, "up1", small, OJ.radius_of(sd, "up1")), k_ref, 1e-5, "real range kernel")
    print("    range_temp:", [float(sd[f"up{i}.range_temp"]) for i in range(1, 5)], " sigma:", [float(sd[f"up{i}.sigma_spatial"]) for i in range(1, 5)])
    # the [1,512,64,64] output is 8 MB: keep the first 64 channels in full and, over ALL channels, two per-pixel moments
    out = {"src": src, "guidance": guid, "out_c64": r[0, :64], "out_sum": r[0].sum(0), "out_sq": (r[0] * r[0]).sum(0), "range1": k_ref}
    out.update({"w." + k: v for k, v in sd.items()})
    save("jbu_real", **out)


# ---------------------------------------------------------------------------------------------
POTSDAM_QIDX = [0, 0, 1, 2, 3, 4, 5, 5]          # configs/cls_potsdam.txt: 8 queries -> 6 classes


def gen_segment():
    print("[segment] forward_feature / forward_slide / postprocess via the reference segmentors")
    cfg = Wt.vit_config("tiny-8")
    wnp = Wt.make_vit_weights(cfg, seed=0)
    w = OV.to_torch(wnp)
    text = Wt.make_text_features(8, cfg.embed_dim)
    out = {}
    sim_cfg = dict(similarity_weight=1.0, temperature=1.0, add_self_similarity=True)
    cases = {
        # name: (class, kwargs, image HxW)
        "ex_base": ("SegmentorEx", dict(model_type="Experimental", global_debias_factor=0.2, similarity_cfg=sim_cfg,
                                        outlier_cfg=dict(top_k=6), prob_thd=0.1, bg_idx=5, slide_crop=32, slide_stride=16), (72, 88)),
        "ex_pad": ("SegmentorEx", dict(model_type="SegEarth", global_debias_factor=0.2, cls_token_lambda=-0.3,
                                       prob_thd=0.0, slide_crop=36, slide_stride=20), (60, 75)),
        "se_plain": ("Segmentor", dict(model_type="SegEarth", cls_token_lambda=-0.3, slide_crop=32, slide_stride=16), (48, 64)),
        "ex_small": ("SegmentorEx", dict(model_type="ClearCLIP", slide_crop=32, slide_stride=16), (24, 40)),   # image < crop
        # Cluster-Then-Debias (sklearn DBSCAN inside the reference), 8x8 patch grid per tile
        "ex_ctd": ("SegmentorEx", dict(model_type="SegEarth", global_debias_factor=0.2, apply_ctd=True, prob_thd=0.1, bg_idx=5,
                                       slide_crop=64, slide_stride=32), (96, 128)),
    }
    with torch.no_grad():
        for name, (cls_name, kw, (H, Wd)) in cases.items():
            net = ref_clip(cfg, wnp)
            s = ref_segmentor(cls_name, net, cfg, text, POTSDAM_QIDX, **kw)
            img = rand_img(zlib.crc32(name.encode()) % 1000, 1, H, Wd)
            metas = [dict(ori_shape=(H, Wd))]
            logits = s.forward_slide(img, metas, s.slide_stride, s.slide_crop)
            pred = s.postprocess_result(logits.clone(), None)
            okw = dict(model_type=kw.get("model_type"), cls_token_lambda=kw.get("cls_token_lambda", 0.0),
                       global_debias_factor=kw.get("global_debias_factor", 0.0),
                       similarity_cfg=kw.get("similarity_cfg"), outlier_cfg=kw.get("outlier_cfg"),
                       prob_thd=kw.get("prob_thd", 0.0), bg_idx=kw.get("bg_idx", 0),
                       slide_crop=kw["slide_crop"], slide_stride=kw["slide_stride"], apply_ctd=kw.get("apply_ctd", False))
            o = OS.SegOracle(cfg, w, torch.from_numpy(text), torch.tensor(POTSDAM_QIDX), **okw)
            ol = o.forward_slide(img)
            close(ol, logits, 3e-5, f"{name} slide logits")
            prob, opred = o.postprocess(ol[0])
            agree = (opred == pred).float().mean().item()
            print(f"    {name}: argmax agreement {agree:.4f}")
            assert agree == 1.0
            # one-tile forward_feature with an explicit logit_size (slide_crop == 0 path)
            fc = kw["slide_crop"] if kw.get("apply_ctd") else 32
            ff = s.forward_feature(img[:, :, :fc, :fc], (40, 44))
            close(o.forward_feature(img[:, :, :fc, :fc], (40, 44)), ff, 3e-5, f"{name} forward_feature")
            out.update({f"{name}.img": img, f"{name}.logits": logits, f"{name}.pred": pred, f"{name}.ff": ff})
            if name == "ex_base":
                # label / confidence images by the reference's own helpers (segmentor.py:568-608); its OpenCV-less branch of
                # _to_colormap is the one exercised (cv2 is a stub here; the JET table lives inside OpenCV)
                mod = R.ref("segmentor")
                saved, mod.cv2 = mod.cv2, None
                try:
                    probs_ref = o.postprocess(ol[0])[0]
                    out["viz.mask"] = s._colorize_mask(pred.squeeze(0).numpy())
                    out["viz.heat"] = s._to_colormap(probs_ref.max(dim=0)[0].numpy())
                    out["viz.palette"] = s._generate_palette(s.num_classes)
                finally:
                    mod.cv2 = saved
        # GEM through segearth_segmentor.Segmentor (the only class where GEM runs, R5)
        gcfg = Wt.vit_config("tiny-gem")
        gnp = Wt.make_vit_weights(gcfg, seed=0)
        net = ref_clip(gcfg, gnp)
        gwrap = R.ref("gem.gem_wrapper").GEMWrapper(model=net, tokenizer=None, depth=7, ignore_residual=True)
        s = ref_segmentor("Segmentor", gwrap.model, gcfg, text, POTSDAM_QIDX, model_type="GEM", slide_crop=32, slide_stride=16)
        img = rand_img(77, 1, 48, 56)
        logits = s.forward_slide(img, [dict(ori_shape=(48, 56))], 16, 32)
        o = OS.SegOracle(gcfg, OV.to_torch(gnp), torch.from_numpy(text), torch.tensor(POTSDAM_QIDX), model_type="GEM",
                         slide_crop=32, slide_stride=16)
        close(o.forward_slide(img), logits, 3e-5, "GEM slide logits")
        out.update({"gem.img": img, "gem.logits": logits})
    out["text"] = text
    save("segment", **out)


# ---------------------------------------------------------------------------------------------
def gen_text():
    """text_<cfg>.npz: the reference's CLIP.encode_text on synthetic text weights + tokenizer-shaped ids."""
    from oracle import text as OT
    M = R.ref("open_clip.model")
    for name, S in (("tiny-text", 9), ("tiny-text-gelu", 5)):
        tc = Wt.TEXT_CONFIGS[name]
        wnp = Wt.make_text_weights(tc, seed=0)
        vis = M.CLIPVisionCfg(layers=1, width=32, head_width=16, patch_size=8, image_size=16)
        txt = M.CLIPTextCfg(context_length=tc.context_length, vocab_size=tc.vocab_size, width=tc.width, heads=tc.heads, layers=tc.layers)
        net = M.CLIP(tc.embed_dim, vis, txt, quick_gelu=tc.quick_gelu).eval()
        sd = net.state_dict()
        for k, v in wnp.items():
            assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
            sd[k] = torch.from_numpy(v.copy())
        net.load_state_dict(sd, strict=True)
        ids = Wt.make_token_ids(tc, S)
        ids[1, 3] = ids[1].max()                       # a duplicated maximum: argmax must take the first
        with torch.no_grad():
            ref = net.encode_text(torch.from_numpy(ids).long())
            refn = net.encode_text(torch.from_numpy(ids).long(), normalize=True)
        mine = OT.encode_text(wnp, tc, ids)
        close(mine, ref, 2e-5, f"text {name}")
        close(OT.encode_text(wnp, tc, ids, normalize=True), refn, 2e-5, f"text {name} normalized")
        save(f"text_{name}", tokens=ids, features=ref.numpy(), features_normalized=refn.numpy())


def gen_ctd():
    """ctd.npz: the reference's Cluster-Then-Debias step exactly as segmentor.py:339-365 calls it (sklearn DBSCAN inside)."""
    from oracle import ctd as OC
    CTD = R.ref("CTD")
    out = {}
    for tag, (n_side, C, spread, nc, seed) in {"a": (12, 32, 0.3, 4, 4), "b": (20, 48, 0.35, 7, 10)}.items():
        n = n_side * n_side
        x = torch.from_numpy(OC.make_clustered_tokens(2, n, C, seed=seed, spread=spread, n_centers=nc))
        g = np.random.default_rng(100 + seed)
        cls = torch.from_numpy(g.standard_normal((2, C)).astype(np.float32))
        cls = cls / cls.norm(dim=-1, keepdim=True)
        feats = x.clone()
        _, labels = CTD.cluster_patch_tokens_dbscan(feats, grid_hw=(n_side, n_side),
                                                   cfg_dict={"max_points": 8192, "metric": "euclidean", "eps": 1.1, "min_samples": 11})
        ref = CTD.adaptive_debiasing(items=feats.clone(), labels=labels, bias=cls, factor=-1.5)
        mine, mlab = OC.ctd_debias(x, cls)
        assert torch.equal(mlab, labels), "DBSCAN restatement differs from the reference"
        close(mine, ref, 1e-6, f"ctd {tag}")
        for b in range(2):                                       # the fixture must not hinge on a rounding-level tie
            _, d2 = OC.neighbour_matrix(OC.ctd_points(x)[b].numpy(), 1.1)
            assert np.abs(d2 - 1.21).min() > 1e-4
        print(f"    ctd {tag}: clusters per tile {[int(l.max()) + 1 for l in labels]}, noise {[int((l < 0).sum()) for l in labels]}")
        out.update({f"{tag}.tokens": x.numpy(), f"{tag}.cls": cls.numpy(), f"{tag}.labels": labels.numpy().astype(np.int32),
                    f"{tag}.out": ref.numpy()})
    save("ctd", **out)


def gen_real():
    """Real-size towers: patch-grid logits, CLS logits and argmax maps on seeded tiles
    (inputs are regenerated from the seed on the GPU box; only outputs are stored)."""
    print("[real] B/16 and L/14 at 224 and 512 (reference CPU fp32; L/14@518 takes a few seconds)")
    sim_cfg = dict(similarity_weight=1.0, temperature=1.0, add_self_similarity=True)
    out = {}
    for vit, sizes in (("ViT-B-16", (224, 512)), ("ViT-L-14", (224, 512))):
        cfg = Wt.vit_config(vit)
        wnp = Wt.make_vit_weights(cfg, seed=0)
        w = OV.to_torch(wnp)
        net = ref_clip(cfg, wnp)
        text = Wt.make_text_features(8, cfg.embed_dim)
        for S in sizes:
            tiles = Wt.normalize_tiles(Wt.make_tiles_u8(1, S, seed=1234, smooth=True))
            img = torch.from_numpy(tiles)
            pad = OS.compute_padsize(S, S, cfg.patch)
            imgp = F.pad(img, pad) if any(pad) else img
            for mt in ("SegEarth", "Experimental"):
                s = ref_segmentor("SegmentorEx", net, cfg, text, POTSDAM_QIDX, model_type=mt, global_debias_factor=0.2,
                                  similarity_cfg=sim_cfg, outlier_cfg=dict(top_k=30), prob_thd=0.1, bg_idx=5)
                t0 = time.time()
                with torch.no_grad():
                    g = imgp.shape[-1] // cfg.patch
                    lg = s.forward_feature(imgp, (g, g))      # patch-grid logits (bilinear to same size = identity)
                dt = time.time() - t0
                o = OS.SegOracle(cfg, w, torch.from_numpy(text), torch.tensor(POTSDAM_QIDX), model_type=mt,
                                 global_debias_factor=0.2, similarity_cfg=sim_cfg, outlier_cfg=dict(top_k=30))
                with torch.no_grad():
                    ol = o.forward_feature(imgp, (g, g))
                close(ol, lg, 2e-4, f"{vit}@{S} {mt} patch logits ({dt:.2f}s ref)")
                key = f"{vit}.{S}.{mt}"
                out[key + ".logits"] = lg[0]
                out[key + ".argmax"] = lg[0].argmax(0).to(torch.uint8)
    save("real_logits", **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    steps = {"tiny": gen_tiny, "layer_fusion": gen_layer_fusion, "refine": gen_refine, "jbu": gen_jbu, "jbu_real": gen_jbu_real, "segment": gen_segment, "text": gen_text, "ctd": gen_ctd, "real": gen_real}
    for k, fn in steps.items():
        if not a.only or a.only == k:
            fn()
    print("done")
