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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_gather_equals_single_process(world):
    wins = tile_windows(100, 130, (20, 20), (36, 36))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, len(wins), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
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
