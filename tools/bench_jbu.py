"""Throughput of the JBU (SimFeatUp) path: per-pixel logits through the upsampler, tile by tile (BASELINE configs[3] shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_decontamination_amd import weights as Wt
import segmentor

vit = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/16"
crop, stride = (512, 256) if len(sys.argv) < 3 else (int(sys.argv[2]), int(sys.argv[2]) // 2)
names = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "cls_isaid.txt")
words, _ = segmentor.get_cls_idx(names)
cfg = Wt.vit_config(vit)
text = torch.from_numpy(Wt.make_text_features(len(words), cfg.embed_dim))
seg = segmentor.SegmentorEx(clip_type="CLIP", vit_type=vit, name_path=names, device=torch.device("cuda:0"), precision="bf16", synthetic_ok=True,
                            text_features=text, model_type="SegEarth", global_debias_factor=0.2, prob_thd=0.4, slide_crop=crop, slide_stride=stride,
                            apply_sim_feat_up=True, sim_feat_up_cfg=dict(model_name="jbu_one", model_path=None), tiles_per_launch=16)
G = int(os.environ.get("JBU_GRID", "4"))                       # G x G tiles per scene (16 by default)
S = stride * (G - 1) + crop
scene = torch.from_numpy(Wt.make_tiles_u8(1, S, seed=1, smooth=True)[0]).cuda()
img = torch.from_numpy(Wt.normalize_tiles(Wt.make_tiles_u8(1, S, seed=1, smooth=True))).cuda()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = seg.forward_slide(img, [dict(ori_shape=(S, S))], stride, crop)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{vit} + jbu_one, {S}x{S} scene, {G * G} tiles of {crop}: {dt * 1e3:.1f} ms -> {G * G * crop * crop / dt / 1e6:.2f} Mpix/s", flush=True)
