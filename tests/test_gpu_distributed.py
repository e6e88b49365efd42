"""Two ranks (gloo, both on cuda:0 -- the box has one GPU) run the REAL pipeline with tile sharding: per-rank HIP towers, the strip
exchange of the sharded cross-tile fusion, the all-gather of tile logits and the stitch must reproduce the single-process canvas
bit for bit (same kernels, same per-tile launch shapes are not guaranteed, so f32 parity mode and a 1e-5 tolerance)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
QIDX = [0, 0, 1, 2, 3, 4, 5, 5]


def _pipe(ctf):
    from clip_decontamination_amd import weights as Wt
    from clip_decontamination_amd.engine import HipVisionTower, HipCLIP
    from clip_decontamination_amd.pipeline import SegPipeline
    cfg = Wt.vit_config("tiny-8")
    tower = HipVisionTower(cfg, Wt.make_vit_weights(cfg, seed=0), precision="f32", device="cuda:0")
    text = torch.from_numpy(Wt.make_text_features(len(QIDX), cfg.embed_dim))
    return SegPipeline(HipCLIP(tower), text, torch.tensor(QIDX), model_type="SegEarth", global_debias_factor=0.2, prob_thd=0.1, bg_idx=5,
                       cross_tile_fusion=ctf, tiles_per_launch=4)


def _scene():
    return torch.from_numpy(np.random.default_rng(11).standard_normal((3, 96, 128), dtype=np.float32))


def _worker(rank, world, port, ctf, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        pipe = _pipe(ctf)
        out = pipe.forward_slide(_scene().cuda(), 32, 32, group="world")   # 3 x 4 = 12 tiles of 32, partitioned over the ranks (opt-in)
        torch.cuda.synchronize()
        q.put((rank, out.cpu().numpy()))                           # by value: tensors travel as shared-memory fds that die with the child
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ctf", [None, dict(fusion_mode="weighted", cache_boundary_width=1, fusion_strength=0.4)])
def test_two_ranks_equal_one_process(ctf):
    single = _pipe(ctf).forward_slide(_scene().cuda(), 32, 32).cpu()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ctf, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        out = torch.from_numpy(out)
        assert out.shape == single.shape
        assert (out - single).abs().max().item() < 1e-5, f"rank {rank}"
