"""The CPU restatement (oracle/) against the fixtures minted from the reference itself
(oracle/gen_golden.py).  This is what pins the oracle; the GPU parity tests then compare the HIP
path with the oracle.  Tolerances: fp32 CPU vs fp32 CPU, 5e-5 absolute on O(1) tensors."""
import numpy as np
import pytest
import torch

from clip_decontamination_amd import weights as Wt
from oracle import vit as OV, refine as OR, jbu as OJ, segment as OS

TOL = 5e-5
POTSDAM_QIDX = [0, 0, 1, 2, 3, 4, 5, 5]


def t(a):
    return torch.from_numpy(np.asarray(a))


def maxdiff(a, b):
    return (torch.as_tensor(a).float() - torch.as_tensor(b).float()).abs().max().item()


@pytest.fixture(scope="module")
def tiny():
    cfg = Wt.vit_config("tiny-8")
    return cfg, OV.to_torch(Wt.make_vit_weights(cfg, seed=0))


@pytest.mark.parametrize("mt", ["vanilla", "MaskCLIP", "ClearCLIP", "SCLIP", "SegEarth", "SFP", "Experimental",
                                "NACLIP", "NOnly", "GAV"])
def test_model_types(golden, tiny, mt):
    cfg, w = tiny
    g = golden("vit_tiny-8")
    img = t(g["img"])
    if mt == "NOnly":
        img = img[:1]
    with torch.no_grad():
        cls, tok = OV.vit_forward(w, cfg, img, mt, True)
    assert maxdiff(tok, g[f"{mt}.tokens"]) < TOL
    assert maxdiff(cls, g[f"{mt}.cls"]) < TOL


def test_residual_and_native_grid(golden, tiny):
    cfg, w = tiny
    g = golden("vit_tiny-8")
    with torch.no_grad():
        cls, tok = OV.vit_forward(w, cfg, t(g["img"]), "SegEarth", False)
        assert maxdiff(tok, g["SegEarth.res.tokens"]) < TOL
        cls, tok = OV.vit_forward(w, cfg, t(g["img_native"]), "SegEarth", True)
        assert maxdiff(tok, g["native.tokens"]) < TOL and maxdiff(cls, g["native.cls"]) < TOL


SIM = dict(similarity_weight=1.0, temperature=1.0, add_self_similarity=True)
OUT = dict(top_k=5)
SA = dict(enhancement_strength=0.1, min_self_attn_threshold=0.15, mode="feature", top_k=4)
COMBOS = {"sim": (SIM, None, None), "out": (None, OUT, None), "sim_out": (SIM, OUT, None), "all": (SIM, OUT, SA),
          "sa_only": (None, None, SA),
          "sa_attn": (SIM, OUT, dict(enhancement_strength=0.3, min_self_attn_threshold=0.15, mode="attention", top_k=4)),
          "sim2": (dict(similarity_weight=0.5, temperature=2.0, add_self_similarity=False), None, None)}


@pytest.mark.parametrize("tag", list(COMBOS))
@pytest.mark.parametrize("mt", ["SegEarth", "Experimental", "ClearCLIP"])
def test_refiner_hooks(golden, tiny, tag, mt):
    cfg, w = tiny
    g = golden("vit_tiny-8")
    sc, oc, ac = COMBOS[tag]
    with torch.no_grad():
        cls, tok = OV.vit_forward(w, cfg, t(g["img"]), mt, True, similarity_cfg=sc, outlier_cfg=oc, self_attn_cfg=ac)
    assert maxdiff(tok, g[f"{tag}.{mt}.tokens"]) < TOL
    assert maxdiff(cls, g[f"{tag}.{mt}.cls"]) < TOL


def test_self_attn_enhancer_alone_is_noop(golden):
    g = golden("vit_tiny-8")     # without an outlier suppressor no attention is captured
    assert maxdiff(g["sa_only.SegEarth.tokens"], g["SegEarth.tokens"]) == 0.0


def test_exact_gelu(golden):
    cfg = Wt.vit_config("tiny-gelu")
    w = OV.to_torch(Wt.make_vit_weights(cfg, seed=0))
    g = golden("vit_tiny-gelu")
    with torch.no_grad():
        cls, tok = OV.vit_forward(w, cfg, t(g["img"]), "SegEarth", True)
    assert maxdiff(tok, g["SegEarth.tokens"]) < TOL


def test_gem(golden):
    cfg = Wt.vit_config("tiny-gem")
    w = OV.to_torch(Wt.make_vit_weights(cfg, seed=0))
    g = golden("vit_tiny-gem")
    with torch.no_grad():
        for ign in (True, False):
            for nm in ("g6", "g4"):
                tok = OV.gem_forward(w, cfg, t(g[f"{nm}.img"]), ign, 7)
                assert maxdiff(tok, g[f"{nm}.ign{int(ign)}.tokens"]) < TOL


def test_refine_modules(golden):
    g = golden("refine")
    grid, attn = t(g["grid"]), t(g["attn"])
    n = grid.shape[-1] * grid.shape[-2]
    idx = OR.detect_outliers(attn, n, 8)
    assert torch.equal(idx, t(g["outlier_idx"]))
    assert maxdiff(OR.suppress_outliers(grid, idx, 0.1), g["suppressed"]) < 1e-6
    assert torch.equal(OR.detect_outliers(t(g["attn4"]).mean(1), n, 5), t(g["outlier_idx4"]))
    for mode in ("feature", "attention"):
        assert maxdiff(OR.self_attention_enhance(grid, attn, 0.3, 0.05, mode, 6), g[f"selfattn_{mode}"]) < 1e-5
    f = t(g["sim_feats"])
    assert maxdiff(OR.similarity_map(f, 1.0, True), g["sim_a"]) < 1e-6
    assert maxdiff(OR.similarity_map(f, 0.5, False), g["sim_b"]) < 1e-6


def test_planted_outliers_in_topk():
    """Property the reference's own test checks (test_outlier_suppression.py:24-46)."""
    torch.manual_seed(0)
    attn = torch.rand(2, 197, 197)
    attn = attn / attn.sum(-1, keepdim=True)
    planted = [10, 50, 100]
    for p in planted:
        attn[:, 0, p + 1] = 0.5
        attn[:, p + 1, p + 1] = 0.001
    idx = OR.detect_outliers(attn, 196, 10)
    assert idx.shape == (2, 10)
    for b in range(2):
        assert set(planted) <= set(idx[b].tolist())


@pytest.mark.parametrize("mode", ["weighted", "attention"])
def test_cross_tile_fusion(golden, mode):
    g = golden("refine")
    tiles = t(g["ctf_tiles"])
    o = OR.CrossTileFusionOracle(mode, 2, 0.3)
    for hi in range(2):
        for wi in range(3):
            r = o(tiles[hi, wi].clone(), hi, wi, 6, 6)
            assert maxdiff(r, g[f"ctf_{mode}"][hi, wi]) < 1e-5


@pytest.mark.parametrize("name", ["jbu_one", "jbu_stack"])
def test_jbu(golden, name):
    g = golden("jbu")
    C = g[f"{name}.src"].shape[1]
    w = OV.to_torch(Wt.make_jbu_weights(name, C, seed=3))
    with torch.no_grad():
        out = OJ.jbu_forward(w, t(g[f"{name}.src"]), t(g[f"{name}.guidance"]))
    assert maxdiff(out, g[f"{name}.out"]) < 1e-4
    assert maxdiff(OJ.adaptive_conv(t(g["ac_in"]), t(g["ac_filt"])), g["ac_out"]) < 1e-5


@pytest.mark.parametrize("tag,sim,lam,ign", [("lf", None, 0.5, True), ("lf_sim", SIM, 0.3, True), ("lf_res", None, 0.5, False)])
@pytest.mark.parametrize("mt", ["SegEarth", "Experimental"])
def test_layer_fusion_matches_the_one_head_reference(golden, tag, sim, lam, ign, mt):
    """apply_layer_fusion + outlier suppressor: the reference runs only with ONE head (its view(N, heads, L, L) of already
    head-averaged weights) -- that run pins the restatement (fixture vit_tiny-1h from oracle.gen_golden --only layer_fusion)."""
    g = golden("vit_tiny-1h")
    cfg = Wt.vit_config("tiny-1h")
    w = OV.to_torch(Wt.make_vit_weights(cfg, seed=0))
    with torch.no_grad():
        cls, tok = OV.vit_forward(w, cfg, t(g["img"]), mt, ign, similarity_cfg=sim, outlier_cfg=dict(top_k=5), layer_fusion={"lambda": lam})
    assert maxdiff(tok, g[f"{tag}.{mt}.tokens"]) < 2e-5 and maxdiff(cls, g[f"{tag}.{mt}.cls"]) < 2e-5


def test_layer_fusion_without_suppressor_is_a_noop(golden):
    g = golden("vit_tiny-1h")
    cfg = Wt.vit_config("tiny-8")
    w = OV.to_torch(Wt.make_vit_weights(cfg, seed=0))
    with torch.no_grad():
        _, tok = OV.vit_forward(w, cfg, t(g["img"]), "SegEarth", True, layer_fusion={"lambda": 0.5})
        _, tok0 = OV.vit_forward(w, cfg, t(g["img"]), "SegEarth", True)
    assert torch.equal(tok, tok0) and maxdiff(tok, g["noop8.tokens"]) < 2e-5


def test_jbu_on_the_trained_checkpoint(golden):
    """JBUStack(512) on the COCO-Stuff weights the reference ships (fixture minted by oracle.gen_golden --only jbu_real from the
    reference module + its own checkpoint): trained range_temp / sigma_spatial / fixup weights."""
    g = golden("jbu_real")
    w = {k[2:]: t(v) for k, v in g.items() if k.startswith("w.")}
    with torch.no_grad():
        out = OJ.jbu_forward(w, t(g["src"]), t(g["guidance"]))[0]
    scale = float(np.abs(g["out_c64"]).max())
    assert maxdiff(out[:64], g["out_c64"]) < 2e-4 * scale
    assert maxdiff(out.sum(0), g["out_sum"]) < 2e-4 * float(np.abs(g["out_sum"]).max()) + 1e-3
    assert maxdiff((out * out).sum(0), g["out_sq"]) < 2e-4 * float(np.abs(g["out_sq"]).max())


CASES = {
    "ex_base": dict(model_type="Experimental", global_debias_factor=0.2, similarity_cfg=SIM, outlier_cfg=dict(top_k=6),
                    prob_thd=0.1, bg_idx=5, slide_crop=32, slide_stride=16),
    "ex_pad": dict(model_type="SegEarth", global_debias_factor=0.2, cls_token_lambda=-0.3, slide_crop=36, slide_stride=20),
    "se_plain": dict(model_type="SegEarth", cls_token_lambda=-0.3, slide_crop=32, slide_stride=16),
    "ex_small": dict(model_type="ClearCLIP", slide_crop=32, slide_stride=16),
}


@pytest.mark.parametrize("name", list(CASES))
def test_slide_and_postprocess(golden, tiny, name):
    cfg, w = tiny
    g = golden("segment")
    o = OS.SegOracle(cfg, w, t(g["text"]), torch.tensor(POTSDAM_QIDX), **CASES[name])
    img = t(g[f"{name}.img"])
    with torch.no_grad():
        lg = o.forward_slide(img)
        assert maxdiff(lg, g[f"{name}.logits"]) < TOL
        _, pred = o.postprocess(lg[0])
        assert torch.equal(pred, t(g[f"{name}.pred"]))
        ff = o.forward_feature(img[:, :, :32, :32], (40, 44))
        assert maxdiff(ff, g[f"{name}.ff"]) < TOL


def test_padsize():
    assert OS.compute_padsize(512, 512, 14) == (3, 3, 3, 3)
    assert OS.compute_padsize(224, 224, 16) == (0, 0, 0, 0)
    assert OS.compute_padsize(36, 37, 8) == (1, 2, 2, 2)


def test_real_size_b16_224(golden):
    """ViT-B/16, one 224x224 tile, 8 queries / 6 classes -- BASELINE.json configs[0]."""
    g = golden("real_logits")
    cfg = Wt.vit_config("ViT-B-16")
    w = OV.to_torch(Wt.make_vit_weights(cfg, seed=0))
    text = torch.from_numpy(Wt.make_text_features(8, cfg.embed_dim))
    img = torch.from_numpy(Wt.normalize_tiles(Wt.make_tiles_u8(1, 224, seed=1234, smooth=True)))
    for mt in ("SegEarth", "Experimental"):
        o = OS.SegOracle(cfg, w, text, torch.tensor(POTSDAM_QIDX), model_type=mt, global_debias_factor=0.2,
                         similarity_cfg=SIM, outlier_cfg=dict(top_k=30))
        with torch.no_grad():
            lg = o.forward_feature(img, (14, 14))[0]
        assert maxdiff(lg, g[f"ViT-B-16.224.{mt}.logits"]) < 1e-4
        assert torch.equal(lg.argmax(0).to(torch.uint8), t(g[f"ViT-B-16.224.{mt}.argmax"]))


@pytest.mark.parametrize("name", ["tiny-text", "tiny-text-gelu"])
def test_text_tower(golden, name):
    """oracle/text.py vs the reference's CLIP.encode_text (open_clip/model.py:288-306)."""
    from oracle import text as OT
    g = golden(f"text_{name}")
    tc = Wt.TEXT_CONFIGS[name]
    w = Wt.make_text_weights(tc, seed=0)
    with torch.no_grad():
        assert maxdiff(OT.encode_text(w, tc, g["tokens"]), g["features"]) < TOL
        assert maxdiff(OT.encode_text(w, tc, g["tokens"], normalize=True), g["features_normalized"]) < TOL
    assert (g["tokens"] == Wt.make_token_ids(tc, g["tokens"].shape[0]))[0].all()      # ids regenerate from the seed


@pytest.mark.parametrize("tag", ["a", "b"])
def test_cluster_then_debias(golden, tag):
    """oracle/ctd.py vs the reference's CTD step (sklearn DBSCAN + adaptive_debiasing, segmentor.py:339-365)."""
    from oracle import ctd as OC
    g = golden("ctd")
    out, labels = OC.ctd_debias(t(g[f"{tag}.tokens"]), t(g[f"{tag}.cls"]))
    assert np.array_equal(labels.numpy(), g[f"{tag}.labels"])
    assert maxdiff(out, g[f"{tag}.out"]) < 1e-5
    assert labels.max() >= 1 and (labels < 0).any()          # several clusters and some noise: the case is not degenerate


def test_known_answer_centre_of_a_3x3_grid():
    """The one known-answer vector in the reference's tests (test_som.py:130-182, written for a module that no longer exists): the
    centre of a 3 x 3 grid holding 1..9 is an outlier and is replaced by the mean of its 8 neighbours, 5.0.  With every channel
    equal the cosine weights of mean_interpolation are uniform, so the shipped module must reproduce it."""
    grid = torch.arange(1, 10, dtype=torch.float32).reshape(1, 1, 3, 3).repeat(1, 4, 1, 1)
    grid[0, :, 1, 1] = 100.0                                   # the outlier value itself must not matter
    out = OR.suppress_outliers(grid, torch.tensor([[4]]), contamination_temp=0.0)
    assert torch.allclose(out[0, :, 1, 1], torch.full((4,), 5.0), atol=1e-6)
    weak = OR.replace_weak_tokens(grid, torch.tensor([[4]]))
    assert torch.allclose(weak[0, :, 1, 1], torch.full((4,), 5.0), atol=1e-6)


def test_layernorm_folding_identity_and_slice_statistics():
    """The algebra the 2-byte modes rely on (csrc/rowops.hip): LayerNorm(x; gamma, beta) @ W^T + b equals
    rstd * (x @ W'^T - mean * c) + b' with W' = gamma o W, c = row sums of W', b' = b + W @ beta; and per-64-column (sum, centred sum
    of squares) slices combine (Chan) to the row's mean and variance, whatever the mean is."""
    import numpy as np
    rng = np.random.default_rng(5)
    M, D, N = 37, 256, 48
    x = rng.standard_normal((M, D)) * np.logspace(-1, 1, M)[:, None] + rng.standard_normal((M, 1)) * 3.0
    g, be = 1 + 0.3 * rng.standard_normal(D), 0.2 * rng.standard_normal(D)
    W, b = rng.standard_normal((N, D)) / 16, rng.standard_normal(N)
    mu, var = x.mean(1, keepdims=True), x.var(1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + 1e-5)
    want = ((x - mu) * rstd * g + be) @ W.T + b
    Wf = W * g
    got = rstd * (x @ Wf.T - mu * Wf.sum(1)) + (b + W @ be)
    assert np.abs(got - want).max() < 1e-10 * np.abs(want).max()
    # slice statistics -> (mean, variance)
    sl = x.reshape(M, D // 64, 64)
    s = sl.sum(2)
    q = ((sl - s[..., None] / 64) ** 2).sum(2)
    mean = s.sum(1) / D
    m2 = (q + 64 * (s / 64 - mean[:, None]) ** 2).sum(1)
    assert np.abs(mean - mu[:, 0]).max() < 1e-12 * (1 + np.abs(mu).max())
    assert np.abs(m2 / D - var[:, 0]).max() < 1e-10 * var.max()


def test_two_plane_f16_product_is_f32_grade():
    """The arithmetic SG_PREC_F16X2 relies on (csrc/common.h split_h2, tools/h2_probe.hip on the hardware): x = hi + lo with hi = f16(x),
    lo = f16(x - hi) carries ~22 significant bits, and hi.hi + hi.lo + lo.hi accumulated in f32 matches an f32 dot product's error,
    three orders of magnitude below plain f16 -- including small operands whose lo plane lives in the f16 subnormals."""
    import numpy as np
    rng = np.random.default_rng(11)
    K = 1024
    for wscale in (1.0, 0.03):
        a = rng.standard_normal((16, K)).astype(np.float32)
        w = (rng.standard_normal((16, K)) * wscale).astype(np.float32)
        ah, wh = a.astype(np.float16), w.astype(np.float16)
        al, wl = (a - ah.astype(np.float32)).astype(np.float16), (w - wh.astype(np.float32)).astype(np.float16)
        rep = np.abs((ah.astype(np.float64) + al.astype(np.float64)) - a).max()
        assert rep <= max(2.0 ** -21 * np.abs(a).max(), 2.0 ** -24)               # representation error of an element
        f = lambda x: x.astype(np.float32)
        three = (f(ah) @ f(wh).T + f(ah) @ f(wl).T + f(al) @ f(wh).T).astype(np.float32)   # f32 accumulation of the three products
        ref = a.astype(np.float64) @ w.astype(np.float64).T
        e3 = np.abs(three - ref).max()
        e32 = np.abs((a @ w.T) - ref).max()
        e16 = np.abs((f(ah) @ f(wh).T) - ref).max()
        # f32-grade: within ~1e-6 of the result scale (numpy's blocked f32 matmul, e32, is itself a few 1e-7), > 100x below plain f16
        assert e3 < 1.5e-6 * np.abs(ref).max() and e3 < 10 * e32 + 1e-6 and e3 < e16 / 100, (wscale, e3, e32, e16)


def test_gem_plus_outlier_composition_definition(tiny=None):
    """BASELINE configs[2] as one forward (oracle/vit.py::gem_forward(outlier_cfg=...)): a composition the reference cannot run.
    Pins the DEFINITION on the tiny GEM tower: (1) without a suppressor it is the reference-pinned GEM forward (vit_tiny-gem fixture, covered
    above); (2) detection reads the ORDINARY stream's head-averaged attention of block L-2, also when that block is not a dual-stream one
    (gem_depth 2); (3) the suppression is exactly suppress_outliers on the GEM stream before ln_post."""
    from clip_decontamination_amd import weights as Wt
    import numpy as np
    cfg = Wt.vit_config("tiny-gem")
    w = OV.to_torch(Wt.make_vit_weights(cfg, seed=0))
    img = torch.from_numpy(np.random.default_rng(1).standard_normal((2, 3, 48, 48), dtype=np.float32))
    with torch.no_grad():
        plain = OV.gem_forward(w, cfg, img, True, 7)
        comp = OV.gem_forward(w, cfg, img, True, 7, outlier_cfg=dict(top_k=5))
        shallow = OV.gem_forward(w, cfg, img, True, 2, outlier_cfg=dict(top_k=5))
    assert comp.shape == plain.shape == shallow.shape
    changed = (comp - plain).abs().amax(dim=-1) > 1e-6                       # [B, n]: tokens the suppression touched
    n_changed = changed.sum(dim=1)
    assert bool((n_changed >= 5).all()) and bool((n_changed <= 5 * 9).all())  # k outliers + at most 8 neighbours each
    assert torch.isfinite(shallow).all()
