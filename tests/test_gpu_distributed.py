"""Two ranks (gloo, both on cuda:0 -- the box has one GPU) run the REAL pipeline with tile sharding: per-rank HIP towers, the strip
exchange of the sharded cross-tile fusion, the all-gather of tile logits and the stitch must reproduce the single-process canvas
(f32 parity mode and the two-plane f16 mode: 1e-5; bf16: a rank's smaller launches may take other kernels, so a 2-byte bound).
The same comparison runs on RCCL (backend nccl, one GPU per rank) whenever the box has two GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
QIDX = [0, 0, 1, 2, 3, 4, 5, 5]


CTF = dict(fusion_mode="weighted", cache_boundary_width=1, fusion_strength=0.4)


def _pipe(ctf, prec="f32", dev="cuda:0"):
    from clip_decontamination_amd import weights as Wt
    from clip_decontamination_amd.engine import HipVisionTower, HipCLIP
    from clip_decontamination_amd.pipeline import SegPipeline
    jbu = ctf in ("jbu", "jbu_ctf")
    cfg = Wt.vit_config("tiny-16" if jbu else "tiny-8")        # the upsampler is 16x: a patch-16 tower
    tower = HipVisionTower(cfg, Wt.make_vit_weights(cfg, seed=0), precision=prec, device=dev)
    text = torch.from_numpy(Wt.make_text_features(len(QIDX), cfg.embed_dim))
    up = None
    if jbu:
        from clip_decontamination_amd.upsampler import HipJBU
        up = HipJBU("jbu_stack", cfg.embed_dim, dev, prec)
        up.load_state_dict(Wt.make_jbu_weights("jbu_stack", cfg.embed_dim, seed=3))
    fusion = CTF if ctf == "jbu_ctf" else (None if jbu else ctf)
    return SegPipeline(HipCLIP(tower), text, torch.tensor(QIDX), model_type="SegEarth", global_debias_factor=0.2, prob_thd=0.1, bg_idx=5,
                       cross_tile_fusion=fusion, tiles_per_launch=4, upsampler=up)


def _scene():
    return torch.from_numpy(np.random.default_rng(11).standard_normal((3, 96, 128), dtype=np.float32))


def _stride(ctf):
    return 24 if ctf == "jbu" else 32                                # JBU case: overlapping tiles, so the halo exchange really carries tiles


def _worker(rank, world, port, ctf, q, prec="f32", backend="gloo"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    gpu = rank if backend == "nccl" else 0                           # RCCL: one GPU per rank; gloo rehearsal: both ranks on cuda:0
    torch.cuda.set_device(gpu)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", gpu))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = f"cuda:{gpu}"
        pipe = _pipe(ctf, prec, dev)
        st = _stride(ctf)
        out = pipe.forward_slide(_scene().to(dev), st, 32, group="world")   # tiles of 32 partitioned over the ranks (opt-in)
        lab = pipe.segment_scene(_scene().to(dev), st, 32, group="world")   # band-local stitch + labels, only the label bands gathered
        torch.cuda.synchronize()
        q.put((rank, out.cpu().numpy(), lab.cpu().numpy()))                           # by value: tensors travel as shared-memory fds that die with the child
    finally:
        dist.destroy_process_group()


# per-precision bound on |sharded - single process| logits (and on the share of differing labels): exact modes run the same arithmetic
# whatever the launch size; in bf16 a rank's half-size launches may be dispatched to other GEMM tiles / a separate LayerNorm pass
TOL = {"f32": (1e-5, 1e-3), "f16x2": (1e-5, 1e-3), "bf16": (2e-2, 2e-2)}


def _run_two_ranks(ctf, prec, backend):
    st = _stride(ctf)
    sp = _pipe(ctf, prec)
    single = sp.forward_slide(_scene().cuda(), st, 32).cpu()
    single_lab = sp.segment_scene(_scene().cuda(), st, 32).cpu()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ctf, q, prec, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tol, tol_lab = TOL[prec]
    for rank, out, lab in res:
        out = torch.from_numpy(out)
        assert out.shape == single.shape
        d = (out - single).abs().max().item()
        print(f"[{backend} {prec} {ctf if isinstance(ctf, str) or ctf is None else 'ctf'}] rank {rank}: |sharded - single| = {d:.2e}")
        assert d < tol, f"rank {rank}"
        assert (torch.from_numpy(lab) != single_lab).float().mean().item() < tol_lab, f"rank {rank} labels"


@pytest.mark.parametrize("ctf", [None, CTF, "jbu", "jbu_ctf"])
def test_two_ranks_equal_one_process(ctf):
    """None / cross-tile fusion: all-gather of patch-grid logits; "jbu": per-pixel logits, point-to-point halo exchange; "jbu_ctf": the
    strips are exchanged, the fused tokens go through the upsampler, the per-pixel logits travel as halo tiles.  Each rank stitches only
    its canvas band; canvas and labels gathered from the bands equal the single-process result."""
    _run_two_ranks(ctf, "f32", "gloo")


@pytest.mark.parametrize("prec", ["f16x2", "bf16"])
def test_two_ranks_equal_one_process_other_precisions(prec):
    """The same comparison in the two-plane f16 mode (exact: same bound as f32) and in bf16 (2-byte bound: a rank's smaller launches may
    take a different kernel path than the single process's)."""
    _run_two_ranks(CTF, prec, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: this box has fewer than two GPUs")
@pytest.mark.parametrize("ctf", [None, CTF, "jbu"])
def test_two_ranks_on_rccl(ctf):
    """The sharded path on backend nccl (= RCCL over xGMI), one GPU per rank: all_gather_into_tensor of tile logits / strips, batch_isend_irecv
    of halo tiles.  Skipped on one-GPU boxes (the builder's); runs wherever the suite meets two GPUs."""
    _run_two_ranks(ctf, "f32", "nccl")
