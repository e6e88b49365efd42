#!/usr/bin/env python3
"""Headline benchmark: segmented Mpix/s, ViT-L/14, 512-pixel tiles, sliding window (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic tiles, inputs resident in HBM:
  uint8 NHWC scene -> [tile crop + normalise + zero pad + im2col] -> ViT-L/14 (23 ordinary blocks, last block
  'Experimental' self-self attention + similarity map, outlier suppression k=30: configs/base_config.py)
  -> global debias + cosine logits (8 Potsdam queries) -> [all-gather of the per-tile logit maps over ranks]
  -> write-once stitch of this rank's canvas band -> softmax / synonym merge / arg-max labels.
Scaling is WEAK: every rank processes TILES_PER_RANK tiles of a scene that grows with the rank count;
value = all ranks' tiles * 512 * 512 / max-over-ranks time.

Also reported in the same JSON line:
  roofline     -- the dominant kernel (bf16 MFMA GEMM): algorithmic FLOPs of its launches / their HIP-event time
                  (events recorded inside the library on the launch stream during the timed steps)
  cpu_baseline -- the oracle (CPU restatement pinned to the reference) timed on this host's cores on a bounded sample
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from clip_decontamination_amd import weights as Wt                      # noqa: E402

POTSDAM_QIDX = [0, 0, 1, 2, 3, 4, 5, 5]      # configs/cls_potsdam.txt: 8 queries -> 6 classes
TILE, STRIDE = 512, 256
TILE_COLS, TILE_ROWS_PER_RANK = 16, 8        # default: 128 tiles per rank per step (a 2304 x 4352 scene band)
PEAK_BF16_TFLOPS = 2500.0                    # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def build_pipeline(device, precision, tiles_per_launch, upsampler="none"):
    from clip_decontamination_amd.engine import HipVisionTower, HipCLIP, SimilarityEnhancementModule, OutlierSuppressionModule
    from clip_decontamination_amd.pipeline import SegPipeline
    cfg = Wt.vit_config("ViT-L-14")
    tower = HipVisionTower(cfg, Wt.make_vit_weights(cfg, seed=0), precision=precision, device=device)
    tower.similarity_enhancer = SimilarityEnhancementModule(1.0, 1.0, True)
    tower.outlier_suppressor = OutlierSuppressionModule(top_k=30)
    text = torch.from_numpy(Wt.make_text_features(len(POTSDAM_QIDX), cfg.embed_dim))
    up = None
    if upsampler != "none":                              # BASELINE configs[3]: ViT-L/14 + SimFeatUp JBU (synthetic upsampler weights, seed 3)
        from clip_decontamination_amd.upsampler import HipJBU
        up = HipJBU(upsampler, cfg.embed_dim, device, precision)
        up.load_state_dict(Wt.make_jbu_weights(upsampler, cfg.embed_dim, seed=3))
    pipe = SegPipeline(HipCLIP(tower), text, torch.tensor(POTSDAM_QIDX), model_type="Experimental", ignore_residual=True,
                       global_debias_factor=0.2, prob_thd=0.1, bg_idx=5, apply_similarity_enhancement=True,
                       tiles_per_launch=tiles_per_launch, upsampler=up)
    return cfg, pipe


def cpu_baseline(cfg, n_tiles=3, keep=None):
    """The oracle (port of the reference path, fp32, torch CPU) on a bounded sample of the same workload.  ``keep`` (a list)
    receives (tile, oracle logits [Q,512,512], oracle labels) of the timed tiles, so the same tiles can be pushed through the HIP
    path and compared (the `parity` object of the bench line)."""
    from oracle import segment as OS, vit as OV
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))            # the GPU box gives one GPU a 16-core CPU share
    torch.set_num_threads(threads)
    w = OV.to_torch(Wt.make_vit_weights(cfg, seed=0))
    text = torch.from_numpy(Wt.make_text_features(len(POTSDAM_QIDX), cfg.embed_dim))
    o = OS.SegOracle(cfg, w, text, torch.tensor(POTSDAM_QIDX), model_type="Experimental", global_debias_factor=0.2,
                     similarity_cfg=dict(similarity_weight=1.0, temperature=1.0, add_self_similarity=True),
                     outlier_cfg=dict(top_k=30), prob_thd=0.1, bg_idx=5, slide_crop=TILE, slide_stride=STRIDE)
    tiles = torch.from_numpy(Wt.normalize_tiles(Wt.make_tiles_u8(n_tiles + 1, TILE, seed=1234, smooth=True)))
    with torch.no_grad():
        o.postprocess(o.forward_slide(tiles[:1])[0])                  # warm-up tile
        t0 = time.perf_counter()
        for i in range(1, n_tiles + 1):
            lg = o.forward_slide(tiles[i:i + 1])[0]
            _, lab = o.postprocess(lg)
            if keep is not None:
                keep.append((tiles[i], lg, lab))
        dt = time.perf_counter() - t0
    return {"value": n_tiles * TILE * TILE / dt / 1e6, "unit": "Mpix/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_tiles} tiles of {TILE}x{TILE} after 1 warm-up, {dt / n_tiles:.2f} s/tile, fp32 torch CPU"}


def main():
    global TILE_COLS, TILE_ROWS_PER_RANK
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f16", "f32", "fp8", "f16x2"],
                    help="bf16 = the headline configuration; f16 = IEEE f16 operands at the same MFMA rate (the reference's own GPU arithmetic); "
                         "f32 = parity mode; fp8 = ordinary-block linears on fp8 MFMA (reported separately, never the default)")
    ap.add_argument("--tile-cols", type=int, default=TILE_COLS, help="tiles per scene row")
    ap.add_argument("--tile-rows", type=int, default=TILE_ROWS_PER_RANK, help="tile rows per rank (weak scaling: the scene grows with the ranks)")
    ap.add_argument("--tiles-per-launch", type=int, default=0, help="0 = all of a rank's tiles in one launch of the tower")
    ap.add_argument("--streams", type=int, default=1, help="experiment: split a rank's tiles over this many HIP streams (tails of one half overlap the other)")
    ap.add_argument("--upsampler", default="none", choices=["none", "jbu_one", "jbu_stack"],
                    help="per-pixel logits through the SimFeatUp JBU upsampler (BASELINE configs[3]); multi-rank: halo tiles travel point to point")
    ap.add_argument("--tuning", type=int, default=None, help="experiment: a library tuning code for sg_set_gemm_config (34 = LayerNorm as its own pass, ...)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-self-check", action="store_true",
                    help="profiling runs only: skip the single-tile re-computations after the timed region, so that rocprofv3's per-kernel averages cover the timed launch shape alone")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsing ranks on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()
    if args.tuning is not None:
        from clip_decontamination_amd import _lib as _sg_lib
        _sg_lib.load().sg_set_gemm_config(int(args.tuning))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP-only (no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from clip_decontamination_amd import _lib, ops
    from clip_decontamination_amd.pipeline import band_plan, exchange_halo_tiles, gather_blocks, partition, tile_windows
    lib = _lib.load()
    jbu = args.upsampler != "none"
    if jbu and args.tile_rows == TILE_ROWS_PER_RANK and args.tile_cols == TILE_COLS:
        args.tile_rows, args.tile_cols = 4, 8            # per-pixel logits are 11 MB per tile: 32 tiles per rank per step by default
    TILE_COLS, TILE_ROWS_PER_RANK = args.tile_cols, args.tile_rows
    if args.tiles_per_launch <= 0:
        args.tiles_per_launch = min(TILE_COLS * TILE_ROWS_PER_RANK, 32 if jbu else 1 << 30)
    cfg, pipe = build_pipeline(device, args.precision, args.tiles_per_launch, args.upsampler)

    # scene: TILE_COLS x (TILE_ROWS_PER_RANK * world) tiles of 512 at stride 256, uint8 NHWC, resident in HBM
    rows_total = TILE_ROWS_PER_RANK * world
    H = STRIDE * (rows_total - 1) + TILE
    W = STRIDE * (TILE_COLS - 1) + TILE
    wins = tile_windows(H, W, (STRIDE, STRIDE), (TILE, TILE))
    assert len(wins) == rows_total * TILE_COLS
    my = wins[rank * TILE_ROWS_PER_RANK * TILE_COLS:(rank + 1) * TILE_ROWS_PER_RANK * TILE_COLS]
    band_h = STRIDE * TILE_ROWS_PER_RANK
    y_lo, y_hi = rank * band_h, (H if rank == world - 1 else (rank + 1) * band_h)
    # every rank only needs the scene rows its tiles touch; generate that slab deterministically per rank
    slab_lo, slab_hi = my[0][0], my[-1][1]
    rng_tiles = Wt.make_tiles_u8(1, max(W, slab_hi - slab_lo), seed=1234 + rank, smooth=True)[0]
    slab = torch.from_numpy(np.ascontiguousarray(rng_tiles[:slab_hi - slab_lo, :W])).to(device)
    my_local = [(y1 - slab_lo, y2 - slab_lo, x1, x2) for (y1, y2, x1, x2) in my]
    win_all = torch.tensor(wins, dtype=torch.int32, device=device)
    l, r, t, b = 3, 3, 3, 3                                           # compute_padsize(512, 512, 14)
    up = (TILE + t + b, TILE + l + r)
    T_all = len(wins)
    assert partition(T_all, world, rank) == (rank * len(my), (rank + 1) * len(my))       # the pipeline's own block partition
    plan = band_plan(wins, H, world)                                  # canvas bands + the tile range each band needs (pipeline.py)
    group = dist.group.WORLD if world > 1 else None

    side = [torch.cuda.Stream(device=device) for _ in range(args.streams)] if args.streams > 1 else []

    def tower():
        if not side:
            return pipe.tile_logits(slab, my_local, (TILE, TILE))
        cur = torch.cuda.current_stream()
        per = (len(my_local) + len(side) - 1) // len(side)
        parts = []
        for i, st in enumerate(side):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                parts.append(pipe.tile_logits(slab, my_local[i * per:(i + 1) * per], (TILE, TILE)))
        for st, part in zip(side, parts):
            cur.wait_stream(st)
            part.record_stream(cur)
        return torch.cat(parts, 0)

    last = {}

    def step():
        tl = tower()                                                  # [tiles of this rank, Q, 37, 37]  (or [.., Q, 592, 592] with the upsampler)
        last["tl"] = tl
        # The product's own sharding helpers (pipeline.sharded_canvas_band does exactly this on a scene every rank holds; here every
        # rank only holds the scene rows of its own tiles, so the steps are spelled out):
        #   patch-grid logits (44 kB per tile): ONE all_gather_into_tensor rebuilds the tile list everywhere (RCCL over xGMI);
        #   per-pixel logits (upsampler, 11 MB per tile): only the tiles that straddle a band edge travel, point to point.
        a_t, b_t = plan[1][rank]
        if world > 1 and not jbu:
            tiles = gather_blocks(tl, T_all, world, rank, group)[a_t:b_t]
        elif world > 1:
            tiles, a_t = exchange_halo_tiles(tl, wins, world, rank, group, plan)
        else:
            tiles = tl
        # this rank stitches + labels ITS band of canvas rows
        w_band = win_all[a_t:a_t + tiles.shape[0]].clone()
        w_band[:, 0:2] -= y_lo
        canvas = ops.stitch(tiles, w_band, up, (t, l), (y_hi - y_lo, W))
        probs, labels = pipe.postprocess(canvas, want_probs=False)
        return labels

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    check(lib.sg_profile_enable(1 << 16))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    lib.sg_profile_disable()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    def prof(cat):
        ms, fl, n, dr = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
        lib.sg_profile_read(cat, C.byref(ms), C.byref(fl), C.byref(n), C.byref(dr))
        return ms.value, fl.value, n.value, dr.value

    def pmc_traffic():
        """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (profiles/pmc_traffic.json, written by
        tools/pmc_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command); None if absent."""
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                t = json.load(f)
            return t.get("gemm_bf16_persist"), t.get("note")
        except (OSError, ValueError):
            return None, None

    # ---- correctness of the TIMED launch shape (outside the timed region) --------------------------------------------------
    # (a) first / middle / last tile of this rank recomputed ALONE (one-tile launches) must equal their rows of the batched launch
    def self_check():
        tl = last["tl"]
        picks = sorted({0, len(my_local) // 2, len(my_local) - 1})
        worst, exact = 0.0, True
        # a one-tile launch would normally take the small-launch dispatch (128 x 128 GEMM tiles, LayerNorm as its own pass); tuning code 36
        # keeps it on the kernels of the timed launch, so that what is compared is the 128-tile stream against a single tile of the SAME code
        lib.sg_set_gemm_config(36)
        try:
            for i in picks:
                alone = pipe.tile_logits(slab, [my_local[i]], (TILE, TILE))[0]
                d = (alone - tl[i]).abs().max().item()
                worst = max(worst, d)
                exact = exact and bool(torch.equal(alone, tl[i]))
        finally:
            lib.sg_set_gemm_config(-1 if args.tuning is None else int(args.tuning))
        if not (worst < 1e-3):
            raise SystemExit(f"bench self-check FAILED: a tile of the {len(my_local)}-tile launch differs from the same tile run alone by {worst}")
        return {"tiles_checked": len(picks), "max_dlogit_batched_vs_alone": worst, "bit_identical": exact}

    batched_check = None if args.no_self_check else self_check()

    tiles_total = len(wins) * args.steps
    value = tiles_total * TILE * TILE / dt / 1e6
    if rank == 0:
        g_ms, g_fl, g_n, g_drop = prof(3)                  # gemm_bf16_persist: the dominant kernel (every large ViT linear), all instantiations
        c_ms, c_fl, c_n, _ = prof(5)                       # ... of which: epilogue applies a folded LayerNorm (QKV, fc)
        p_ms, p_fl, p_n, _ = prof(6)                       # ... of which: epilogue also emits the next LayerNorm's operand + statistics (out-proj, proj)
        def _part(ms, fl, n):
            return {"launches": n, "avg_launch_us": round(ms * 1e3 / max(n, 1), 2), "achieved": round(fl / (ms * 1e-3) / 1e12, 2) if ms > 0 else 0.0}
        o_ms, o_fl, o_n, _ = prof(0)                       # the remaining bf16 GEMM launches (patch embed, proj, similarity map)
        a_ms, a_fl, a_n, _ = prof(1)
        f_ms, f_fl, f_n, _ = prof(4)                       # fp8 GEMM launches (only with --precision fp8)
        achieved = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        out = {
            "metric": "segmented Mpix/sec ViT-L/14 512-tile slide", "value": round(value, 3), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "ViT-L/14, 512x512x3 uint8 tiles at stride 256 (padded to 518, N=1370 tokens), "
                                   "model_type=Experimental + similarity enhancement + outlier suppression k=30 + global debias 0.2, "
                                   + ("SimFeatUp JBU upsampler to per-pixel logits, " if jbu else "") +
                                   "8 Potsdam queries / 6 classes, slide stitch + arg-max labels",
                       "tiles_per_step_per_gpu": TILE_ROWS_PER_RANK * TILE_COLS, "tiles_per_launch": args.tiles_per_launch,
                       "scene": f"{H}x{W}", "upsampler": args.upsampler,
                       "partition": (f"tile rows over {world} rank(s), " + ("point-to-point halo exchange of per-pixel tile logits" if jbu
                                     else "all-gather of patch-grid logits") + ", band-local stitch + labels")},
            "roofline": {"bound": "mfma", "kernel": "gemm_bf16_persist", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": pmc_traffic()[0], "traffic_note": pmc_traffic()[1],
                         "launches": g_n, "avg_launch_us": round(g_ms * 1e3 / max(g_n, 1), 2),
                         "algorithmic_gflop_per_launch": round(g_fl / max(g_n, 1) / 1e9, 3), "events_dropped": g_drop,
                         "by_epilogue": {"note": "the same kernel's three instantiations; the folded ones carry 46 of the step's 48 LayerNorm passes "
                                                 "(the separate pass was 220 us per LayerNorm), so their FLOP rate is not comparable with round 1's",
                                         "plain": _part(g_ms - c_ms - p_ms, g_fl - c_fl - p_fl, g_n - c_n - p_n),
                                         "folded_ln_consumer": _part(c_ms, c_fl, c_n), "folded_ln_producer": _part(p_ms, p_fl, p_n)},
                         "attention": {"kernel": "attn_kernel", "achieved": round(a_fl / (a_ms * 1e-3) / 1e12, 2) if a_ms > 0 else 0.0,
                                       "launches": a_n, "avg_launch_us": round(a_ms * 1e3 / max(a_n, 1), 2)},
                         "fp8_gemm": ({"kernel": "gemm_bf16_ring<256,256,2,4,2,FP8>", "achieved": round(f_fl / (f_ms * 1e-3) / 1e12, 2), "launches": f_n,
                                       "peak": 5000.0, "share_of_step_time": round(f_ms / (dt * 1e3), 4)} if f_n else None),
                         "share_of_step_time": {"gemm_bf16_persist": round(g_ms / (dt * 1e3), 4), "other_bf16_gemm": round(o_ms / (dt * 1e3), 4),
                                                "attention": round(a_ms / (dt * 1e3), 4)}},
        }
        out["self_check"] = batched_check
        if jbu:
            out["cpu_baseline"] = None
            out["cpu_baseline_note"] = ("not timed with the upsampler: the oracle's JBU (the reference's unfold form) needs ~12 GB and ~70 s per 512-pixel "
                                        "tile (BASELINE.md); parity of this path: tests/test_gpu_configs.py::test_config4_l14_jbu_isaid")
        elif world == 1 and not args.no_cpu_baseline:
            kept = []
            out["cpu_baseline"] = cpu_baseline(cfg, keep=kept)
            # (b) the CPU-baseline tiles through the HIP path (same precision mode as the timed run): logits vs the oracle's
            worst, agree, npx = 0.0, 0, 0
            for tile, lg_ref, lab_ref in kept:
                lg = pipe.forward_slide(tile.to(device), STRIDE, TILE)[0]
                _, lab = pipe.postprocess(lg, want_probs=False)
                worst = max(worst, (lg.cpu() - lg_ref).abs().max().item())
                agree += int((lab.cpu() == lab_ref).sum())
                npx += lab_ref.numel()
            out["parity"] = {"max_dlogit": worst, "label_agreement": agree / max(npx, 1), "n_tiles": len(kept),
                             "against": "oracle (fp32 CPU restatement pinned to the reference), the cpu_baseline tiles, per-pixel logits after stitch",
                             "mode": args.precision}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def check(rc):
    if rc != 0:
        raise RuntimeError("libsegearth_hip call failed")


if __name__ == "__main__":
    main()
