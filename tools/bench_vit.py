"""Tile throughput of other towers / precisions than the headline (exploration aid): python tools/bench_vit.py ViT-H-14 fp8 64"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_decontamination_amd import weights as Wt
from clip_decontamination_amd.engine import HipVisionTower, HipCLIP
from clip_decontamination_amd.pipeline import SegPipeline, tile_windows

vit, prec, T = sys.argv[1], sys.argv[2], int(sys.argv[3])
cfg = Wt.vit_config(vit)
tower = HipVisionTower(cfg, Wt.make_vit_weights(cfg, seed=0), precision=prec, device="cuda:0")
text = torch.from_numpy(Wt.make_text_features(2, cfg.embed_dim))
pipe = SegPipeline(HipCLIP(tower), text, torch.tensor([0, 1]), model_type="SegEarth", global_debias_factor=0.2, tiles_per_launch=T)
scene = torch.from_numpy(Wt.make_tiles_u8(1, 2304, seed=1, smooth=True)[0]).cuda()
wins = tile_windows(2304, 2304, (256, 256), (512, 512))[:T]
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe.tile_logits(scene, wins, (512, 512))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{vit} {prec}: {T} tiles of 512 in {dt * 1e3:.1f} ms -> {T * 512 * 512 / dt / 1e6:.1f} Mpix/s", flush=True)
