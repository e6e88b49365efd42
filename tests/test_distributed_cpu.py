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


def fake_tile_logits(self, scene, windows, tile_hw, scene_index=None):
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
        got = pipe.gather_tile_logits(None, wins, (36, 36), world, rank)
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
        try:
            res = [q.get(timeout=180) for _ in procs]
        except Exception:
            pass
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
