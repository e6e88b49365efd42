"""GPU parity of the whole vision tower (sg_vit_forward through the C ABI) against the golden
fixtures minted from the reference and against the oracle restatement on seeded inputs.

Parity bar (BASELINE.json north_star): per-pixel class logits within 1e-3 of the reference's fp32
CPU path and arg-max identical -- asserted in SG_PREC_F32.  In SG_PREC_BF16 (throughput mode) operands
are rounded to bf16, so the test asserts a looser bound and reports arg-max agreement."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from clip_decontamination_amd import weights as Wt            # noqa: E402
from oracle import vit as OV, segment as OS                    # noqa: E402  (checker only)

DEV = "cuda:0"
# Throughput-mode bounds = about 2x the measured worst case on MI355X (gpurun logs of round 2; the measured value is printed by every
# test).  bf16 has 8 significant bits, f16 has 11: the f16 bounds are ~8x tighter.
# measured (r2): bf16 tiny tokens <= 8.8e-3, GEM 7.4e-3, real-size max|dlogit| 6.9e-2 (B/16@224: an outlier top-k pick that flips under
# rounding replaces a whole token) / q99 4.0e-3 / agreement >= 0.9847, sa_attn 7.7e-3;  f16: 1.1e-3, 1.2e-3, 1.06e-4 / 7.8e-5 / 1.0000, 1.0e-3
HALF_TOL = {
    "bf16": dict(tiny_tokens=0.018, gem=0.015, real_err=0.14, real_q99=8e-3, real_agree=0.97, sa_attn=0.016),
    "f16": dict(tiny_tokens=2.2e-3, gem=2.5e-3, real_err=2.5e-4, real_q99=1.6e-4, real_agree=0.998, sa_attn=2.2e-3),   # agreement: 1369 samples, two flips allowed
}
POTSDAM_QIDX = [0, 0, 1, 2, 3, 4, 5, 5]
SIM = dict(similarity_weight=1.0, temperature=1.0, add_self_similarity=True)


def tower(cfg_name, precision):
    from clip_decontamination_amd.engine import HipVisionTower, HipCLIP
    cfg = Wt.vit_config(cfg_name)
    return cfg, HipCLIP(HipVisionTower(cfg, Wt.make_vit_weights(cfg, seed=0), precision=precision, device=DEV))


def install(net, sim=None, out=None, sa=None):
    from clip_decontamination_amd import engine as E
    net.visual.similarity_enhancer = E.SimilarityEnhancementModule(**sim) if sim else None
    net.visual.outlier_suppressor = E.OutlierSuppressionModule(**out) if out else None
    net.visual.self_attn_enhancer = E.SelfAttentionEnhancementModule(**sa) if sa else None


def maxdiff(a, b):
    return (torch.as_tensor(a).float().cpu() - torch.as_tensor(b).float().cpu()).abs().max().item()


@pytest.fixture(scope="module", params=["f32", "f16x2"])
def tiny_f32(request):
    """The tiny tower in an EXACT mode: f32 MFMA parity mode, or the two-plane f16 mode (SG_PREC_F16X2: f32-grade results on the f16 matrix
    pipe) -- every test below holds both to the same fp32 bounds against the reference fixtures."""
    return tower("tiny-8", request.param)


@pytest.fixture(scope="module", params=["bf16", "f16"])
def tiny_bf16(request):
    """The tiny tower in a throughput mode: bf16 or f16 operands (SG_PREC_BF16 / SG_PREC_F16)."""
    cfg, net = tower("tiny-8", request.param)
    net.precision_tag = request.param
    return cfg, net


@pytest.mark.parametrize("mt", ["vanilla", "MaskCLIP", "ClearCLIP", "SCLIP", "SegEarth", "SFP", "Experimental", "NACLIP", "NOnly", "GAV"])
def test_tiny_model_types_f32(golden, tiny_f32, mt):
    cfg, net = tiny_f32
    install(net)
    g = golden("vit_tiny-8")
    img = torch.from_numpy(g["img"])
    if mt == "NOnly":
        img = img[:1]                     # the reference's NOnly branch only works for batch 1 (transformer.py:924)
    cls, tok = net.encode_image(img.to(DEV), mt, True, output_cls_token=True)
    assert maxdiff(tok, g[f"{mt}.tokens"]) < 1e-4
    assert maxdiff(cls, g[f"{mt}.cls"]) < 1e-4


@pytest.mark.parametrize("mt", ["vanilla", "MaskCLIP", "ClearCLIP", "SCLIP", "SegEarth", "SFP", "Experimental", "NACLIP", "NOnly", "GAV"])
def test_tiny_model_types_bf16(golden, tiny_bf16, mt):
    cfg, net = tiny_bf16
    install(net)
    g = golden("vit_tiny-8")
    img = torch.from_numpy(g["img"])
    if mt == "NOnly":
        img = img[:1]
    cls, tok = net.encode_image(img.to(DEV), mt, True, output_cls_token=True)
    ref = torch.from_numpy(g[f"{mt}.tokens"])
    rel = maxdiff(tok, ref) / ref.abs().max().item()
    print(f"[{net.precision_tag}] tiny {mt}: max rel token error {rel:.3e}")
    assert rel < HALF_TOL[net.precision_tag]["tiny_tokens"]


def test_tiny_residual_native_gelu_f32(golden, tiny_f32):
    cfg, net = tiny_f32
    install(net)
    g = golden("vit_tiny-8")
    cls, tok = net.encode_image(torch.from_numpy(g["img"]).to(DEV), "SegEarth", False, output_cls_token=True)
    assert maxdiff(tok, g["SegEarth.res.tokens"]) < 1e-4
    cls, tok = net.encode_image(torch.from_numpy(g["img_native"]).to(DEV), "SegEarth", True, output_cls_token=True)
    assert maxdiff(tok, g["native.tokens"]) < 1e-4 and maxdiff(cls, g["native.cls"]) < 1e-4
    cfg2, net2 = tower("tiny-gelu", "f32")
    g2 = golden("vit_tiny-gelu")
    cls, tok = net2.encode_image(torch.from_numpy(g2["img"]).to(DEV), "SegEarth", True, output_cls_token=True)
    assert maxdiff(tok, g2["SegEarth.tokens"]) < 1e-4


COMBOS = {"sim": (SIM, None, None), "out": (None, dict(top_k=5), None), "sim_out": (SIM, dict(top_k=5), None),
          "all": (SIM, dict(top_k=5), dict(enhancement_strength=0.1, min_self_attn_threshold=0.15, mode="feature", top_k=4)),
          "sa_only": (None, None, dict(enhancement_strength=0.1, min_self_attn_threshold=0.15, mode="feature", top_k=4)),
          "sa_attn": (SIM, dict(top_k=5), dict(enhancement_strength=0.3, min_self_attn_threshold=0.15, mode="attention", top_k=4)),
          "sim2": (dict(similarity_weight=0.5, temperature=2.0, add_self_similarity=False), None, None)}


@pytest.mark.parametrize("tag", list(COMBOS))
@pytest.mark.parametrize("mt", ["SegEarth", "Experimental", "ClearCLIP"])
def test_tiny_refiners_f32(golden, tiny_f32, tag, mt):
    cfg, net = tiny_f32
    sc, oc, ac = COMBOS[tag]
    install(net, sc, oc, ac)
    g = golden("vit_tiny-8")
    cls, tok = net.encode_image(torch.from_numpy(g["img"]).to(DEV), mt, True, output_cls_token=True,
                                apply_similarity_enhancement=sc is not None)
    assert maxdiff(tok, g[f"{tag}.{mt}.tokens"]) < 1e-4
    assert maxdiff(cls, g[f"{tag}.{mt}.cls"]) < 1e-4


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("f16x2", 1e-4), ("bf16", None), ("f16", None)])
def test_tiny_gem(golden, prec, tol):
    cfg, net = tower("tiny-gem", prec)
    g = golden("vit_tiny-gem")
    for ign in (True, False):
        net.visual.gem_ignore_residual = ign
        for nm in ("g6", "g4"):
            tok = net.visual(torch.from_numpy(g[f"{nm}.img"]).to(DEV))
            ref = torch.from_numpy(g[f"{nm}.ign{int(ign)}.tokens"])
            bound = tol if tol is not None else HALF_TOL[prec]["gem"] * ref.abs().max().item()
            print(f"[{prec}] GEM {nm} ign={ign}: max rel error {maxdiff(tok, ref) / ref.abs().max().item():.3e}")
            assert maxdiff(tok, ref) < bound, (prec, ign, nm, maxdiff(tok, ref))


def head_logits(net, cfg, img, mt, text, precision_tag):
    from clip_decontamination_amd import ops
    cls, tok = net.encode_image(img, mt, True, output_cls_token=True, apply_similarity_enhancement=True)
    lg = ops.cosine_logits(tok, cls, text, 0.2, 0.0)
    g = img.shape[-1] // cfg.patch
    return lg.reshape(1, -1, g, g)


@pytest.mark.parametrize("prec", ["f32", "f16x2"])
@pytest.mark.parametrize("vit,S", [("ViT-B-16", 224), ("ViT-B-16", 512), ("ViT-L-14", 224), ("ViT-L-14", 512)])
def test_real_size_parity_f32(golden, vit, S, prec):
    """BASELINE configs[0] (B/16, one 224 tile, 8 queries / 6 classes) and the L/14 512-tile shape:
    logits within 1e-3 of the reference fixture, arg-max bit-exact -- in parity mode and in the two-plane f16 mode."""
    cfg, net = tower(vit, prec)
    install(net, SIM, dict(top_k=30))
    g = golden("real_logits")
    text = torch.from_numpy(Wt.make_text_features(8, cfg.embed_dim)).to(DEV)
    img = torch.from_numpy(Wt.normalize_tiles(Wt.make_tiles_u8(1, S, seed=1234, smooth=True)))
    pad = OS.compute_padsize(S, S, cfg.patch)
    imgp = (F.pad(img, pad) if any(pad) else img).to(DEV)
    for mt in ("SegEarth", "Experimental"):
        lg = head_logits(net, cfg, imgp, mt, text, prec)[0]
        ref = torch.from_numpy(g[f"{vit}.{S}.{mt}.logits"])
        err = maxdiff(lg, ref)
        agree = (lg.argmax(0).cpu().to(torch.uint8) == torch.from_numpy(g[f"{vit}.{S}.{mt}.argmax"])).float().mean().item()
        print(f"[{prec}] {vit}@{S} {mt}: max|dlogit| = {err:.2e}, argmax agreement = {agree:.4f}")
        assert err < 1e-3
        assert agree == 1.0


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("vit,S", [("ViT-B-16", 224), ("ViT-L-14", 512)])
def test_real_size_parity_bf16(golden, vit, S, prec):
    """Throughput modes: report max|dlogit| and arg-max agreement against the fp32 reference fixture."""
    cfg, net = tower(vit, prec)
    install(net, SIM, dict(top_k=30))
    g = golden("real_logits")
    text = torch.from_numpy(Wt.make_text_features(8, cfg.embed_dim)).to(DEV)
    img = torch.from_numpy(Wt.normalize_tiles(Wt.make_tiles_u8(1, S, seed=1234, smooth=True)))
    pad = OS.compute_padsize(S, S, cfg.patch)
    imgp = (F.pad(img, pad) if any(pad) else img).to(DEV)
    for mt in ("SegEarth", "Experimental"):
        lg = head_logits(net, cfg, imgp, mt, text, prec)[0]
        ref = torch.from_numpy(g[f"{vit}.{S}.{mt}.logits"])
        err = maxdiff(lg, ref)
        agree = (lg.argmax(0).cpu().to(torch.uint8) == torch.from_numpy(g[f"{vit}.{S}.{mt}.argmax"])).float().mean().item()
        q99 = torch.quantile((lg.cpu() - ref).abs().flatten(), 0.99).item()
        print(f"[{prec}] {vit}@{S} {mt}: max|dlogit| = {err:.2e}, 99th percentile = {q99:.2e}, argmax agreement = {agree:.4f}")
        # the maximum is set by discrete events (an outlier top-k pick that flips under bf16 rounding replaces a whole token), so it is
        # bounded loosely; the bulk of the error is what bf16 operands must deliver
        assert err < HALF_TOL[prec]["real_err"]
        assert q99 < HALF_TOL[prec]["real_q99"]
        assert agree > HALF_TOL[prec]["real_agree"]


def test_u8_ingest_equals_float_ingest():
    """uint8 NHWC tiles with the fused (x-mean)/std == normalised float planes (segmentor.py:64-67)."""
    cfg, net = tower("tiny-8", "f32")
    install(net)
    u8 = Wt.make_tiles_u8(2, 40, seed=5)
    f = torch.from_numpy(Wt.normalize_tiles(u8)).to(DEV)
    v = net.visual
    win = torch.tensor([[0, 40, 0, 40], [0, 40, 0, 40]], dtype=torch.int32)
    idx = torch.tensor([0, 1], dtype=torch.int32)
    opts = v.forward_opts("SegEarth", True)
    c1, t1 = v.forward_tiles(f, win, (40, 40), opts, idx)
    c2, t2 = v.forward_tiles(torch.from_numpy(u8).to(DEV), win, (40, 40), opts, idx)
    assert maxdiff(t1, t2) < 1e-5 and maxdiff(c1, c2) < 1e-5


def test_windowed_tiles_equal_cropped_tiles():
    """On-device crop + zero pad (window into the scene) == feeding the cropped, F.pad-ded tile."""
    cfg, net = tower("tiny-8", "f32")
    install(net)
    v = net.visual
    scene = torch.from_numpy(np.random.default_rng(3).standard_normal((3, 70, 90), dtype=np.float32)).to(DEV)
    wins = [(0, 36, 0, 36), (34, 70, 54, 90), (10, 46, 20, 56)]
    opts = v.forward_opts("SegEarth", True)
    c, t = v.forward_tiles(scene, torch.tensor([[a, b, cc, d] for (a, b, cc, d) in wins], dtype=torch.int32), (36, 36), opts)
    for i, (y1, y2, x1, x2) in enumerate(wins):
        tile = scene[None, :, y1:y2, x1:x2]
        pad = OS.compute_padsize(36, 36, cfg.patch)
        ci, ti = net.encode_image(F.pad(tile, pad), "SegEarth", True, output_cls_token=True)
        assert maxdiff(t[i], ti[0]) < 1e-5 and maxdiff(c[i], ci[0]) < 1e-5


@pytest.mark.parametrize("mt", ["SegEarth", "Experimental"])
def test_tiny_selfattn_attention_mode_bf16(golden, tiny_bf16, mt):
    """mode='attention' in throughput mode: the head-averaged attention is rebuilt per image in f32 from the bf16 q, k."""
    cfg, net = tiny_bf16
    sc, oc, ac = COMBOS["sa_attn"]
    install(net, sc, oc, ac)
    g = golden("vit_tiny-8")
    cls, tok = net.encode_image(torch.from_numpy(g["img"]).to(DEV), mt, True, output_cls_token=True, apply_similarity_enhancement=True)
    ref = torch.from_numpy(g[f"sa_attn.{mt}.tokens"])
    print(f"[{net.precision_tag}] sa_attn {mt}: max rel error {maxdiff(tok, ref) / ref.abs().max().item():.3e}")
    assert maxdiff(tok, ref) < HALF_TOL[net.precision_tag]["sa_attn"] * ref.abs().max().item()


# ---- attention-map layer fusion (apply_layer_fusion, reference transformer.py:598-607,630-637,647-690) -----------------------------
@pytest.mark.parametrize("tag,sim,lam,ign", [("lf", None, 0.5, True), ("lf_sim", SIM, 0.3, True), ("lf_res", None, 0.5, False)])
@pytest.mark.parametrize("mt", ["SegEarth", "Experimental"])
@pytest.mark.parametrize("prec", ["f32", "f16x2"])
def test_layer_fusion_one_head_reference_fixture_f32(golden, tag, sim, lam, ign, mt, prec):
    """The one configuration the reference's own layer-fusion code runs in (heads == 1): fixture from the reference."""
    cfg, net = tower("tiny-1h", prec)
    install(net, sim, dict(top_k=5))
    g = golden("vit_tiny-1h")
    cls, tok = net.encode_image(torch.from_numpy(g["img"]).to(DEV), mt, ign, output_cls_token=True, apply_layer_fusion=True,
                                layer_fusion_lambda=lam, apply_similarity_enhancement=sim is not None)
    assert maxdiff(tok, g[f"{tag}.{mt}.tokens"]) < 1e-4 and maxdiff(cls, g[f"{tag}.{mt}.cls"]) < 1e-4


@pytest.mark.parametrize("prec", ["f32", "f16x2", "bf16", "f16"])
def test_layer_fusion_multi_head_vs_oracle(prec):
    """heads > 1: the definition the one-head case pins (per-image head-averaged maps, as nn.MultiheadAttention returns them), against the
    oracle restatement; and fusion without a suppressor leaves the output unchanged."""
    cfg, net = tower("tiny-8", prec)
    w = OV.to_torch(Wt.make_vit_weights(cfg, seed=0))
    img = torch.from_numpy(np.random.default_rng(41).standard_normal((2, 3, 48, 48), dtype=np.float32))
    for mt, sim in (("SegEarth", None), ("Experimental", SIM)):
        install(net, sim, dict(top_k=5))
        cls, tok = net.encode_image(img.to(DEV), mt, True, output_cls_token=True, apply_layer_fusion=True, layer_fusion_lambda=0.4,
                                    apply_similarity_enhancement=sim is not None)
        with torch.no_grad():
            rc, rt = OV.vit_forward(w, cfg, img, mt, True, similarity_cfg=sim, outlier_cfg=dict(top_k=5), layer_fusion={"lambda": 0.4})
        rel = maxdiff(tok, rt) / rt.abs().max().item()
        print(f"[{prec}] layer fusion {mt}: max rel token error {rel:.3e}")
        assert rel < {"f32": 1e-5, "f16x2": 1e-5, "bf16": 0.02, "f16": 3e-3}[prec]
    install(net)
    a = net.encode_image(img.to(DEV), "SegEarth", True, output_cls_token=True, apply_layer_fusion=True)[1]
    b = net.encode_image(img.to(DEV), "SegEarth", True, output_cls_token=True)[1]
    assert torch.equal(a, b)


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_folded_layernorm_tower_vs_separate_pass(prec):
    """ViT-B/16, 64 tiles (R = 12 608 rows: enough 256 x 256 tiles for the persistent GEMM path, where ln_1 / ln_2 are folded into the QKV / fc
    GEMMs; a handful of tiles would take the small-launch dispatch, which keeps the LayerNorm passes): the tower with the
    folding switched off (tuning code 34: LayerNorm as its own pass, as in round 1) must agree with the default to 2-byte rounding, and a
    block whose LayerNorm parameters are replaced AFTER sg_vit_finalize (no staged f32 weight to re-fold from) falls back to the pass."""
    import ctypes as C
    from clip_decontamination_amd import _lib
    lib = _lib.load()
    cfg, net = tower("ViT-B-16", prec)
    img = torch.from_numpy(Wt.normalize_tiles(Wt.make_tiles_u8(64, 224, seed=77, smooth=True))).to(DEV)
    cls_f, tok_f = net.encode_image(img, "SegEarth", True, output_cls_token=True)
    lib.sg_set_gemm_config(34)
    try:
        cls_u, tok_u = net.encode_image(img, "SegEarth", True, output_cls_token=True)
    finally:
        lib.sg_set_gemm_config(-1)
    scale = tok_u.abs().max().item()
    d = (tok_f - tok_u).abs().max().item()
    print(f"[{prec}] folded vs separate LayerNorm: max|dtoken| = {d:.3e} (|token| max {scale:.2f})")
    assert d < HALF_TOL[prec]["tiny_tokens"] * scale
    assert not torch.equal(tok_f, tok_u)                      # the two paths really are different code
    # replace one block's ln_2 scale after finalize: that block must leave the folded path (its folded weight would be stale)
    v = net.visual
    g = (torch.from_numpy(Wt.make_vit_weights(cfg, seed=0)["transformer.resblocks.3.ln_2.weight"]) * 1.5).to(DEV)
    _lib.check(lib.sg_vit_set_tensor(v._ctx, b"transformer.resblocks.3.ln_2.weight", C.c_void_p(g.data_ptr()), g.numel(), None), "set_tensor")
    _lib.check(lib.sg_vit_finalize(v._ctx, None), "finalize")
    torch.cuda.synchronize()
    _, tok_a = net.encode_image(img, "SegEarth", True, output_cls_token=True)
    lib.sg_set_gemm_config(34)
    try:
        _, tok_b = net.encode_image(img, "SegEarth", True, output_cls_token=True)
    finally:
        lib.sg_set_gemm_config(-1)
    assert (tok_a - tok_b).abs().max().item() < HALF_TOL[prec]["tiny_tokens"] * scale      # the new scale is honoured on both paths
    assert (tok_a - tok_f).abs().max().item() > 1e-3 * scale                                # and it did change the result
