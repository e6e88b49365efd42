"""Multi-rank host logic on CPU (gloo, world_size 2): tile partition + equal-sized all-gather + reassembly must give
the same tile-logit list as one process.  The per-tile compute is replaced by a deterministic stand-in (the HIP
kernels need a GPU); the collective, padding and ordering code is the product's."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clip_decontamination_amd.pipeline import SegPipeline, partition, tile_windows


def fake_tile_logits(self, scene, windows, tile_hw, scene_index=None, grid_of_tiles=None):
    out = []
    for (y1, y2, x1, x2) in windows:
        base = torch.arange(3 * 4 * 5, dtype=torch.float32).reshape(3, 4, 5)
        out.append(base + 1000.0 * y1 + x1)
    return torch.stack(out, 0)


def _worker(rank, world, port, n_tiles_expected, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pipe = SegPipeline.__new__(SegPipeline)
        pipe.device = torch.device("cpu")
        SegPipeline.tile_logits = fake_tile_logits
        wins = tile_windows(100, 130, (20, 20), (36, 36))
        assert len(wins) == n_tiles_expected
        got = pipe.gather_tile_logits(None, wins, (36, 36), world, rank, dist.group.WORLD)
        want = fake_tile_logits(pipe, None, wins, (36, 36))
        q.put((rank, bool(torch.equal(got, want)), tuple(got.shape)))
    finally:
        dist.destroy_process_group()


def _run_ranks(target, world, extra_args, attempts=3):
    """Spawn `world` gloo ranks; a rendezvous port can be taken between probing and binding, so a failed start is retried."""
    ctx = mp.get_context("spawn")
    for attempt in range(attempts):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=target, args=(r, world, port) + tuple(extra_args) + (q,)) for r in range(world)]
        for p in procs:
            p.start()
        res = []
        import queue as _queue, time as _time
        deadline = _time.time() + 180
        while len(res) < world and _time.time() < deadline:
            try:
                res.append(q.get(timeout=1.0))
            except _queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):       # a rank died: do not sit out the timeout
                    break
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        if len(res) == world and all(p.exitcode == 0 for p in procs):
            return res
    raise AssertionError(f"{target.__name__}: ranks failed {attempts} times")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_gather_equals_single_process(world):
    wins = tile_windows(100, 130, (20, 20), (36, 36))
    res = _run_ranks(_worker, world, (len(wins),))
    for rank, ok, shape in res:
        assert ok, f"rank {rank} reassembled a different tile list"
        assert shape[0] == len(wins)


def test_partition_covers_everything():
    for n in (1, 7, 9, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [partition(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_tile_windows_match_reference_rule():
    # last window shifted back inside the image (segmentor.py:418-423); image smaller than the crop -> one window
    assert tile_windows(512, 512, (112, 112), (224, 224))[-1] == (288, 512, 288, 512)
    assert len(tile_windows(512, 512, (112, 112), (224, 224))) == 16
    assert len(tile_windows(1024, 1024, (256, 256), (512, 512))) == 9
    assert tile_windows(100, 90, (112, 112), (224, 224)) == [(0, 100, 0, 90)]


# ---- cross-tile fusion under tile sharding: strip exchange -------------------------------------------------
class TorchCrossTileSteps:
    """CPU stand-in for ops.CrossTileSteps (same pack / fuse / apply contract), built on the oracle's fuse functions."""

    def __init__(self, gh, gw, C, bw, mode, strength, wg):
        self.gh, self.gw, self.C, self.bw, self.mode, self.strength, self.wg = gh, gw, C, bw, mode, strength, wg

    def strip_len(self, which):
        return self.gh * self.bw if which == 0 else self.bw * self.gw

    def pack(self, tokens, tile0, which, left_result=None):
        g = tokens.reshape(-1, self.gh, self.gw, self.C).clone()
        if which == 0:
            return g[:, :, -self.bw:].reshape(g.shape[0], -1, self.C)
        if left_result is not None:
            for i in range(g.shape[0]):
                if (tile0 + i) % self.wg > 0:
                    g[i, :, :self.bw] = left_result[i].reshape(self.gh, self.bw, self.C)
        return g[:, -self.bw:].reshape(g.shape[0], -1, self.C)

    def fuse(self, tokens, tile0, nbr_strips, which):
        from oracle import refine as OR
        g = tokens.reshape(-1, self.gh, self.gw, self.C)
        out = torch.zeros(g.shape[0], self.strip_len(which), self.C)
        f = (lambda c, n: OR.fuse_attention(c, n, self.strength)) if self.mode == "attention" else (lambda c, n: OR.fuse_weighted(c, n, self.strength))
        for i in range(g.shape[0]):
            t = tile0 + i
            hi, wi = divmod(t, self.wg)
            if which == 0 and wi > 0:
                out[i] = f(g[i, :, :self.bw].reshape(1, -1, self.C), nbr_strips[t - 1][None])[0]
            if which == 1 and hi > 0:
                out[i] = f(g[i, :self.bw].reshape(1, -1, self.C), nbr_strips[t - self.wg][None])[0]
        return out

    def apply(self, tokens, tile0, left_result, top_result):
        g = tokens.reshape(-1, self.gh, self.gw, self.C).clone()
        for i in range(g.shape[0]):
            hi, wi = divmod(tile0 + i, self.wg)
            if hi > 0:
                g[i, :self.bw] = top_result[i].reshape(self.bw, self.gw, self.C)
            if wi > 0:                                              # the left result wins the corner
                g[i, :, :self.bw] = left_result[i].reshape(self.gh, self.bw, self.C)
        return g.reshape(tokens.shape)


def _ctf_case(grid):
    (hg, wg), gp, C = grid, 7, 12
    tok = torch.from_numpy(__import__("numpy").random.default_rng(5).standard_normal((hg * wg, gp * gp, C)).astype("float32"))
    return hg, wg, gp, C, tok


def _ctf_worker(rank, world, port, mode, grid, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clip_decontamination_amd.pipeline import sharded_cross_tile_fusion, gather_blocks
        hg, wg, gp, C, tok = _ctf_case(grid)
        T = hg * wg
        lo, hi = partition(T, world, rank)
        steps = TorchCrossTileSteps(gp, gp, C, 2, mode, 0.5, wg)
        mine = sharded_cross_tile_fusion(tok[lo:hi].clone(), steps, T, world, rank)
        full = gather_blocks(mine, T, world, rank)
        q.put((rank, full.numpy()))                    # by value (tensors travel as shared-memory fds that die with the child)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,grid", [(2, "weighted", (3, 4)), (3, "attention", (3, 4)), (5, "weighted", (3, 4)),
                                             (3, "weighted", (1, 2))])      # last: more ranks than tiles, rank 2 owns nothing
def test_sharded_cross_tile_fusion_equals_sequential_reference_semantics(world, mode, grid):
    """Partitioned tiles + two strip all-gathers == the reference module fed tile by tile in raster order
    (oracle CrossTileFusionOracle, pinned to cross_tile_fusion.py by tests/golden/refine.npz)."""
    from oracle import refine as OR
    hg, wg, gp, C, tok = _ctf_case(grid)
    o = OR.CrossTileFusionOracle(mode, 2, 0.5)
    ref = torch.stack([o(tok[t:t + 1].clone(), t // wg, t % wg, gp, gp)[0] for t in range(hg * wg)], 0)
    res = _run_ranks(_ctf_worker, world, (mode, grid))
    for rank, full in res:
        full = torch.from_numpy(full)
        assert full.shape == ref.shape
        assert (full - ref).abs().max().item() < 1e-5, f"rank {rank}"


# ---- opt-in tile sharding, band-local stitch, halo exchange (pipeline.forward_slide / segment_scene) ---------------------------
Q_FAKE = 3


def scene_tile_logits(self, scene, windows, tile_hw, scene_index=None, grid_of_tiles=None):
    """Deterministic stand-in for the HIP tower: logits that depend on the scene CONTENT under the window, so two ranks holding
    different scenes produce different tiles.  Patch-grid form [T,Q,4,5]; per-pixel form [T,Q,th,tw] when an 'upsampler' is set."""
    out = []
    for (y1, y2, x1, x2) in windows:
        crop = scene[:, y1:y2, x1:x2]
        if getattr(self, "upsampler", None) is not None:
            out.append(torch.stack([crop[0] * (q + 1) + 0.01 * crop[1] - crop[2] * q for q in range(Q_FAKE)], 0))
        else:
            g = torch.nn.functional.adaptive_avg_pool2d(crop[None], (4, 5))[0]
            out.append(torch.stack([g[0] * (q + 1) + g[1] - g[2] * q for q in range(Q_FAKE)], 0))
    return torch.stack(out, 0)


def torch_stitch(self, tile_logits, windows, up_hw, pad_tl, canvas_hw):
    """CPU stand-in for ops.stitch with the kernel's contract: bilinear resize to up_hw, un-pad, mean over the covering tiles
    accumulated in raster order."""
    Q = tile_logits.shape[1]
    H, W = canvas_hw
    acc = torch.zeros(Q, H, W)
    cnt = torch.zeros(1, H, W)
    for t, (y1, y2, x1, x2) in enumerate(windows.tolist()):
        up = torch.nn.functional.interpolate(tile_logits[t][None], size=tuple(up_hw), mode="bilinear")[0]
        up = up[:, pad_tl[0]:pad_tl[0] + (y2 - y1), pad_tl[1]:pad_tl[1] + (x2 - x1)]
        ya, yb_ = max(y1, 0), min(y2, H)
        if yb_ <= ya:
            continue
        acc[:, ya:yb_, x1:x2] += up[:, ya - y1:yb_ - y1]
        cnt[:, ya:yb_, x1:x2] += 1
    assert (cnt == 0).sum() == 0
    return acc / cnt


def _fake_pipe(upsampler):
    from types import SimpleNamespace
    pipe = SegPipeline.__new__(SegPipeline)
    pipe.device = torch.device("cpu")
    pipe.visual = SimpleNamespace(cfg=SimpleNamespace(patch=4))
    pipe.num_queries = Q_FAKE
    pipe.upsampler = object() if upsampler else None
    pipe.cross_tile_fusion = None
    pipe.tile_group = None
    pipe.tile_logits = scene_tile_logits.__get__(pipe)
    pipe._stitch = torch_stitch.__get__(pipe)
    pipe.postprocess = lambda lg, want_probs=False: (None, lg.argmax(0, keepdim=True))
    return pipe


def _scene(seed, H=100, W=132):
    return torch.from_numpy(__import__("numpy").random.default_rng(seed).standard_normal((3, H, W)).astype("float32"))


def _slide_worker(rank, world, port, case, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if case == "image_parallel":
            # the reference's own multi-GPU launch: every rank holds a DIFFERENT image and no tile group is given -> no sharding
            pipe = _fake_pipe(False)
            out = pipe.forward_slide(_scene(100 + rank), 20, 36)
            q.put((rank, out.numpy()))
        elif case == "mismatch":
            pipe = _fake_pipe(False)
            try:
                pipe.forward_slide(_scene(7, H=100 + 4 * rank), 20, 36, group="world")
                q.put((rank, "no error"))
            except RuntimeError as e:
                q.put((rank, str(e)))
        else:
            ups = case.startswith("pixel")
            pipe = _fake_pipe(ups)
            pipe.tile_group = "world"
            canvas = pipe.forward_slide(_scene(7), 20, 36)
            labels = pipe.segment_scene(_scene(7), 20, 36)
            q.put((rank, canvas.numpy(), labels.numpy()))
    finally:
        dist.destroy_process_group()


def test_initialised_process_group_alone_does_not_shard_tiles():
    """with torch.distributed initialised for image-level data parallelism (dist_test.sh, DefaultSampler) and no
    tile group, each rank must get exactly the single-process result for ITS image."""
    res = _run_ranks(_slide_worker, 2, ("image_parallel",))
    for rank, out in res:
        want = _fake_pipe(False).forward_slide(_scene(100 + rank), 20, 36)
        assert torch.equal(torch.from_numpy(out), want), f"rank {rank}"


def test_tile_group_with_different_scenes_is_rejected():
    res = _run_ranks(_slide_worker, 2, ("mismatch",))
    for rank, msg in res:
        assert "different scenes" in msg, (rank, msg)


@pytest.mark.parametrize("world,case", [(2, "grid"), (3, "grid"), (2, "pixel"), (3, "pixel"), (5, "pixel")])
def test_sharded_slide_equals_single_process(world, case):
    """grid: all-gather of patch-grid tile logits; pixel: point-to-point halo exchange of per-pixel tile logits.  Both stitch only
    the rank's band; canvas and labels gathered from the bands must equal the single-process result bit for bit."""
    single = _fake_pipe(case == "pixel")
    want = single.forward_slide(_scene(7), 20, 36)
    want_lab = single.segment_scene(_scene(7), 20, 36)
    res = _run_ranks(_slide_worker, world, (case,))
    for rank, canvas, labels in res:
        assert torch.equal(torch.from_numpy(canvas), want), f"rank {rank} canvas"
        assert torch.equal(torch.from_numpy(labels), want_lab), f"rank {rank} labels"


def test_band_plan_covers_canvas_and_needs_only_lower_ranks():
    from clip_decontamination_amd.pipeline import band_plan
    for (H, W, s, c) in [(100, 132, 20, 36), (512, 512, 256, 512), (2304, 4352, 256, 512), (60, 60, 112, 224)]:
        wins = tile_windows(H, W, (s, s), (c, c))
        for world in (1, 2, 3, 5, 8, 16):
            yb, need = band_plan(wins, H, world)
            assert yb[0] == 0 and yb[-1] == H and all(yb[i] <= yb[i + 1] for i in range(world))
            for r in range(world):
                lo, hi = partition(len(wins), world, r)
                a, b = need[r]
                if yb[r + 1] > yb[r]:
                    assert a <= lo and b == hi
                    covering = [t for t, w in enumerate(wins) if w[1] > yb[r] and w[0] < yb[r + 1]]
                    assert covering and covering[0] >= a and covering[-1] < b, (H, world, r)


def test_launch_chunks_are_balanced_and_cover_the_tile_list():
    from clip_decontamination_amd.pipeline import launch_chunks
    assert launch_chunks(0, 32) == []
    assert launch_chunks(33, 32) == [(0, 17), (17, 33)]
    assert launch_chunks(32, 32) == [(0, 32)]
    assert launch_chunks(5, 1) == [(i, i + 1) for i in range(5)]
    for n in range(1, 200):
        for lim in (1, 7, 32, 128):
            ch = launch_chunks(n, lim)
            assert ch[0][0] == 0 and ch[-1][1] == n and all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
            sizes = [b - a for a, b in ch]
            assert max(sizes) <= lim and max(sizes) - min(sizes) <= 1 and len(ch) == -(-n // lim)
