"""The drop-in segmentors end to end on the GPU (f32 parity mode) against fixtures produced by the REFERENCE
classes' own forward_slide / forward_feature / postprocess_result (tests/golden/segment.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POTSDAM = os.path.join(ROOT, "configs", "cls_potsdam.txt")
SIM = dict(similarity_weight=1.0, temperature=1.0, add_self_similarity=True)

CASES = {
    "ex_base": ("SegmentorEx", dict(model_type="Experimental", global_debias_factor=0.2, apply_similarity_enhancement=True,
                                    similarity_enhancement_cfg=SIM, apply_outlier_suppression=True, outlier_suppression_cfg=dict(top_k=6),
                                    prob_thd=0.1, bg_idx=5, slide_crop=32, slide_stride=16)),
    "ex_pad": ("SegmentorEx", dict(model_type="SegEarth", global_debias_factor=0.2, cls_token_lambda=-0.3, slide_crop=36, slide_stride=20)),
    "se_plain": ("Segmentor", dict(model_type="SegEarth", cls_token_lambda=-0.3, slide_crop=32, slide_stride=16, apply_sim_feat_up=False)),
    "ex_small": ("SegmentorEx", dict(model_type="ClearCLIP", slide_crop=32, slide_stride=16)),
    # Cluster-Then-Debias: the reference runs scikit-learn DBSCAN on the CPU per tile (segmentor.py:339-365)
    "ex_ctd": ("SegmentorEx", dict(model_type="SegEarth", global_debias_factor=0.2, apply_ctd=True, prob_thd=0.1, bg_idx=5,
                                   slide_crop=64, slide_stride=32)),
}


def build(cls_name, text, **kw):
    import segmentor, segearth_segmentor
    cls = segmentor.SegmentorEx if cls_name == "SegmentorEx" else segearth_segmentor.Segmentor
    return cls(clip_type="CLIP", vit_type="tiny-8", name_path=POTSDAM, device=torch.device("cuda:0"), precision="f32",
               synthetic_ok=True, text_features=text, **kw)


@pytest.mark.parametrize("name", list(CASES))
def test_slide_feature_postprocess_match_reference(golden, name):
    g = golden("segment")
    cls_name, kw = CASES[name]
    seg = build(cls_name, torch.from_numpy(g["text"]), **kw)
    img = torch.from_numpy(g[f"{name}.img"]).cuda()
    H, W = img.shape[-2:]
    logits = seg.forward_slide(img, [dict(ori_shape=(H, W))], seg.slide_stride, seg.slide_crop)
    assert (logits.cpu() - torch.from_numpy(g[f"{name}.logits"])).abs().max().item() < 1e-3
    pred = seg.postprocess_result(logits, None)
    assert torch.equal(pred.cpu(), torch.from_numpy(g[f"{name}.pred"]))
    fc = 64 if name == "ex_ctd" else 32
    ff = seg.forward_feature(img[:, :, :fc, :fc], (40, 44))
    assert (ff.cpu() - torch.from_numpy(g[f"{name}.ff"])).abs().max().item() < 1e-3
    # predict(): same path behind the mmseg entry point (data_samples=None returns the label map)
    pred2 = seg.predict(img, None)
    assert torch.equal(pred2.cpu(), torch.from_numpy(g[f"{name}.pred"]))


def test_gem_segmentor(golden):
    import segearth_segmentor
    g = golden("segment")
    seg = segearth_segmentor.Segmentor(clip_type="CLIP", vit_type="tiny-gem", model_type="GEM", name_path=POTSDAM,
                                       device=torch.device("cuda:0"), precision="f32", synthetic_ok=True,
                                       text_features=torch.from_numpy(g["text"]), slide_crop=32, slide_stride=16, apply_sim_feat_up=False)
    img = torch.from_numpy(g["gem.img"]).cuda()
    logits = seg.forward_slide(img, [dict(ori_shape=tuple(img.shape[-2:]))], 16, 32)
    assert (logits.cpu() - torch.from_numpy(g["gem.logits"])).abs().max().item() < 1e-3


def test_predict_with_data_samples_and_ori_shape_resize(golden):
    g = golden("segment")
    seg = build("SegmentorEx", torch.from_numpy(g["text"]), model_type="SegEarth", slide_crop=32, slide_stride=16, prob_thd=0.1, bg_idx=5)

    class Sample:
        def __init__(self, meta):
            self.metainfo = meta
            self.data = {}

        def set_data(self, d):
            self.data.update(d)

    img = torch.from_numpy(g["ex_base.img"]).cuda()
    H, W = img.shape[-2:]
    s = Sample(dict(ori_shape=(H + 9, W - 5)))
    out = seg.predict(img, [s])
    assert out[0].data["seg_logits"].data.shape == (6, H + 9, W - 5)
    assert out[0].data["pred_sem_seg"].data.shape == (1, H + 9, W - 5)
    assert out[0].data["pred_sem_seg"].data.dtype == torch.int64


def test_jbu_segmentor_smoke(golden):
    """apply_sim_feat_up=True (the shipped base config): per-pixel logits through the JBU; compared with the oracle
    composition (debias -> JBU -> cosine logits) on one tile."""
    from clip_decontamination_amd import weights as Wt
    from oracle import segment as OS, vit as OV
    g = golden("segment")
    text = torch.from_numpy(g["text"])
    cfg = Wt.vit_config("tiny-16")
    seg = __import__("segmentor").SegmentorEx(clip_type="CLIP", vit_type="tiny-16", model_type="SegEarth", name_path=POTSDAM,
                                               device=torch.device("cuda:0"), precision="f32", synthetic_ok=True,
                                               text_features=torch.from_numpy(Wt.make_text_features(8, cfg.embed_dim)),
                                               global_debias_factor=0.2, apply_sim_feat_up=True,
                                               sim_feat_up_cfg=dict(model_name="jbu_stack", model_path=None), slide_crop=64, slide_stride=32)
    img = torch.from_numpy(np.random.default_rng(4).standard_normal((1, 3, 64, 96), dtype=np.float32))
    logits = seg.forward_slide(img.cuda(), [dict(ori_shape=(64, 96))], 32, 64)
    o = OS.SegOracle(cfg, OV.to_torch(Wt.make_vit_weights(cfg, seed=0)), torch.from_numpy(Wt.make_text_features(8, cfg.embed_dim)),
                     torch.tensor([0, 0, 1, 2, 3, 4, 5, 5]), model_type="SegEarth", global_debias_factor=0.2,
                     jbu_weights=OV.to_torch(Wt.make_jbu_weights("jbu_stack", cfg.embed_dim, seed=3)), slide_crop=64, slide_stride=32)
    with torch.no_grad():
        ref = o.forward_slide(img)
    assert (logits.cpu() - ref).abs().max().item() < 1e-3


def test_label_and_confidence_images(golden, tmp_path):
    """result_dir / heatmap_dir (segmentor.py:501-531): images rendered by sg_render_maps == the reference's _colorize_mask /
    _to_colormap (its OpenCV-less branch) on the same prediction; PNG files named after the sample's img_path."""
    from PIL import Image
    g = golden("segment")
    _, kw = CASES["ex_base"]
    seg = build("SegmentorEx", torch.from_numpy(g["text"]), result_dir=str(tmp_path / "res"), heatmap_dir=str(tmp_path / "heat"), **kw)
    assert np.array_equal(seg._generate_palette(seg.num_classes), g["viz.palette"])

    class Sample:
        def __init__(self, meta):
            self.metainfo = meta

        def set_data(self, d):
            pass

    img = torch.from_numpy(g["ex_base.img"]).cuda()
    H, W = img.shape[-2:]
    seg.predict(img, [Sample(dict(ori_shape=(H, W), img_path="/data/potsdam/top_2_13.tif"))])
    mask = np.array(Image.open(tmp_path / "res" / "top_2_13.png"))
    heat = np.array(Image.open(tmp_path / "heat" / "top_2_13.png"))
    assert np.array_equal(mask, g["viz.mask"])
    # the grey level is uint8(p * 255): a probability within 1e-6 of a k/255 boundary may land one level off
    d = np.abs(heat.astype(np.int32) - g["viz.heat"].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
