"""Run-to-run reproducibility of the HIP path: the same inputs must give the same BYTES on every call.

Nothing in the library uses atomics or an unordered reduction, so any difference between two calls is a defect (a race, an
uninitialised read, a hazard).  Round 3 found one this way: jbu_conv_lowres_kernel changed rare pixels from run to run once its
Keff arithmetic had been SLP-packed into v_pk_fma_f32 ('JBU reproducibility'); parity tests with bf16-sized
tolerances had not seen it.  The launches here are large enough that co-resident workgroups drift apart (more workgroups than the
chip holds at once), which is what that defect needed."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from clip_decontamination_amd import weights as Wt            # noqa: E402

DEV = "cuda:0"
RUNS = 4


def _same(outs, what):
    for i in range(1, len(outs)):
        a, b = outs[0], outs[i]
        if not torch.equal(a, b):
            d = (a != b)
            raise AssertionError(f"{what}: run {i} differs from run 0 in {int(d.sum())} of {d.numel()} values; max |d| = "
                                 f"{(a.float() - b.float()).abs().max().item():.3e}")


@pytest.mark.parametrize("name", ["jbu_one", "jbu_stack"])
@pytest.mark.parametrize("prec", ["bf16", "f32"])
def test_jbu_is_reproducible(name, prec):
    """32 x 32 tokens -> 512 x 512 (4096 workgroups in the last stage: eight rounds on 256 CUs x 2 slots)."""
    from clip_decontamination_amd.upsampler import get_upsampler
    C, g, B = 64, 32, 1
    g_ = torch.Generator().manual_seed(21)
    src = torch.randn(B, C, g, g, generator=g_).to(DEV)
    guid = torch.from_numpy(Wt.normalize_tiles(Wt.make_tiles_u8(B, 16 * g, seed=77, smooth=True))).to(DEV)
    up = get_upsampler(name, C, DEV, prec)
    up.load_state_dict(Wt.make_jbu_weights(name, C, seed=3))
    outs = []
    for _ in range(RUNS + 2):
        outs.append(up(src, guid).clone())
        torch.cuda.synchronize()
    _same(outs, f"JBU {name} {prec}")


@pytest.mark.parametrize("prec", ["bf16", "f16", "fp8", "f16x2", "f32"])
def test_tower_is_reproducible(prec):
    """ViT-B/16, 24 tiles of 224 x 224 through the dense-feature path (every GEMM, the folded LayerNorm, both attention kernels):
    the persistent GEMMs run their MFMA group and their epilogue group side by side on every SIMD."""
    from clip_decontamination_amd import ops
    from clip_decontamination_amd.engine import HipVisionTower, HipCLIP, SimilarityEnhancementModule, OutlierSuppressionModule
    cfg = Wt.vit_config("ViT-B-16")
    net = HipCLIP(HipVisionTower(cfg, Wt.make_vit_weights(cfg, seed=0), precision=prec, device=DEV))
    net.visual.similarity_enhancer = SimilarityEnhancementModule(similarity_weight=1.0, temperature=1.0, add_self_similarity=True)
    net.visual.outlier_suppressor = OutlierSuppressionModule(top_k=30)
    B = 24 if prec != "f32" else 4
    img = torch.from_numpy(Wt.normalize_tiles(Wt.make_tiles_u8(B, 224, seed=1234, smooth=True))).to(DEV)
    text = torch.from_numpy(Wt.make_text_features(8, cfg.embed_dim)).to(DEV)
    for mt in ("SegEarth", "Experimental"):
        outs = []
        for _ in range(RUNS):
            cls, tok = net.encode_image(img, mt, True, output_cls_token=True, apply_similarity_enhancement=True)
            outs.append(ops.cosine_logits(tok, cls, text, 0.2, 0.0).clone())
            torch.cuda.synchronize()
        _same(outs, f"tower {prec} {mt}")


@pytest.mark.parametrize("prec", ["bf16", "f16x2"])
def test_l14_tower_is_reproducible(prec):
    """ViT-L/14 on 518-pixel tiles (the headline shape: 1370 tokens, D = 1024), 8 tiles."""
    from clip_decontamination_amd import ops
    from clip_decontamination_amd.engine import HipVisionTower, HipCLIP, SimilarityEnhancementModule, OutlierSuppressionModule
    cfg = Wt.vit_config("ViT-L-14")
    net = HipCLIP(HipVisionTower(cfg, Wt.make_vit_weights(cfg, seed=0), precision=prec, device=DEV))
    net.visual.similarity_enhancer = SimilarityEnhancementModule(similarity_weight=1.0, temperature=1.0, add_self_similarity=True)
    net.visual.outlier_suppressor = OutlierSuppressionModule(top_k=30)
    img = torch.from_numpy(Wt.normalize_tiles(Wt.make_tiles_u8(8, 518, seed=99, smooth=True))).to(DEV)
    text = torch.from_numpy(Wt.make_text_features(8, cfg.embed_dim)).to(DEV)
    outs = []
    for _ in range(RUNS):
        cls, tok = net.encode_image(img, "Experimental", True, output_cls_token=True, apply_similarity_enhancement=True)
        outs.append(ops.cosine_logits(tok, cls, text, 0.2, 0.0).clone())
        torch.cuda.synchronize()
    _same(outs, f"L/14 tower {prec}")


def test_jbu_logits_tail_is_reproducible():
    """The fused JBU + cosine-logits tail (sg_jbu_logits: row-dot GEMM epilogue, bf16 chain) on a 512-channel map, 16 x 16 tokens x 4 tiles."""
    import ctypes as C
    from clip_decontamination_amd import _lib
    from clip_decontamination_amd.upsampler import get_upsampler
    from clip_decontamination_amd.ops import ptr, stream_ptr
    Cc, g, T, Q = 512, 16, 4, 16
    up = get_upsampler("jbu_one", Cc, DEV, "bf16")
    up.load_state_dict(Wt.make_jbu_weights("jbu_one", Cc, seed=3))
    gen = torch.Generator().manual_seed(5)
    tok = torch.randn(T, g * g, Cc, generator=gen).to(DEV)
    cls = torch.randn(T, Cc, generator=gen).to(DEV)
    text = F.normalize(torch.randn(Q, Cc, generator=gen), dim=-1).to(DEV)
    guid = torch.from_numpy(Wt.normalize_tiles(Wt.make_tiles_u8(T, 16 * g, seed=3, smooth=True))).to(DEV).contiguous()
    lib = _lib.load()
    P = 256 * g * g
    need = lib.sg_jbu_workspace_bytes(up._ctx, T, g, g)
    wp, wn = up._workspace(need)
    outs = []
    for _ in range(RUNS):
        lg = torch.empty(T, Q, P, dtype=torch.float32, device=DEV)
        _lib.check(lib.sg_jbu_logits(up._ctx, ptr(tok), ptr(guid), T, g, g, 16 * g, 16 * g, _lib.PREC_BF16, ptr(text), Q, ptr(cls), 0.2, ptr(lg), wp, wn,
                                     stream_ptr()), "sg_jbu_logits")
        torch.cuda.synchronize()
        outs.append(lg)
    _same(outs, "JBU fused logits tail")
