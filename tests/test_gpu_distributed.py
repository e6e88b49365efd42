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
    jbu = ctf == "jbu"
    cfg = Wt.vit_config("tiny-16" if jbu else "tiny-8")        # the upsampler is 16x: a patch-16 tower
    tower = HipVisionTower(cfg, Wt.make_vit_weights(cfg, seed=0), precision="f32", device="cuda:0")
    text = torch.from_numpy(Wt.make_text_features(len(QIDX), cfg.embed_dim))
    up = None
    if jbu:
        from clip_decontamination_amd.upsampler import HipJBU
        up = HipJBU("jbu_stack", cfg.embed_dim, "cuda:0", "f32")
        up.load_state_dict(Wt.make_jbu_weights("jbu_stack", cfg.embed_dim, seed=3))
    return SegPipeline(HipCLIP(tower), text, torch.tensor(QIDX), model_type="SegEarth", global_debias_factor=0.2, prob_thd=0.1, bg_idx=5,
                       cross_tile_fusion=None if jbu else ctf, tiles_per_launch=4, upsampler=up)


def _scene():
    return torch.from_numpy(np.random.default_rng(11).standard_normal((3, 96, 128), dtype=np.float32))


def _worker(rank, world, port, ctf, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        pipe = _pipe(ctf)
        st = 24 if ctf == "jbu" else 32                              # JBU case: overlapping tiles, so the halo exchange really carries tiles
        out = pipe.forward_slide(_scene().cuda(), st, 32, group="world")   # tiles of 32 partitioned over the ranks (opt-in)
        lab = pipe.segment_scene(_scene().cuda(), st, 32, group="world")   # band-local stitch + labels, only the label bands gathered
        torch.cuda.synchronize()
        q.put((rank, out.cpu().numpy(), lab.cpu().numpy()))                           # by value: tensors travel as shared-memory fds that die with the child
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ctf", [None, dict(fusion_mode="weighted", cache_boundary_width=1, fusion_strength=0.4), "jbu"])
def test_two_ranks_equal_one_process(ctf):
    """None / cross-tile fusion: all-gather of patch-grid logits; "jbu": per-pixel logits, point-to-point halo exchange.  Each rank
    stitches only its canvas band; canvas and labels gathered from the bands equal the single-process result."""
    st = 24 if ctf == "jbu" else 32
    sp = _pipe(ctf)
    single = sp.forward_slide(_scene().cuda(), st, 32).cpu()
    single_lab = sp.segment_scene(_scene().cuda(), st, 32).cpu()
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
    for rank, out, lab in res:
        out = torch.from_numpy(out)
        assert out.shape == single.shape
        assert (out - single).abs().max().item() < 1e-5, f"rank {rank}"
        assert (torch.from_numpy(lab) != single_lab).float().mean().item() < 1e-3, f"rank {rank} labels"
