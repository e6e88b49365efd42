#!/usr/bin/env python3
"""Headline benchmark: segmented Mpix/s, ViT-L/14, 512-pixel tiles, sliding window (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]                      (N > 1 without a launcher: the script starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic tiles, inputs resident in HBM:
  uint8 NHWC scene -> [tile crop + normalise + zero pad + im2col] -> ViT-L/14 (23 ordinary blocks, last block
  'Experimental' self-self attention + similarity map, outlier suppression k=30: configs/base_config.py)
  -> global debias + cosine logits (8 Potsdam queries) -> [all-gather of the per-tile logit maps over ranks]
  -> write-once stitch of this rank's canvas band -> softmax / synonym merge / arg-max labels.
Scaling: WEAK by default (every rank processes 119 tiles -- 7 x 17 windows -- of a scene that grows with the rank count); `--scaling strong` = ONE fixed
8192 x 8192 scene (961 tiles) whose raster tile list is partitioned over the ranks (pipeline.partition).
value = all ranks' tiles * 512 * 512 / max-over-ranks time.

Also reported in the same JSON line:
  roofline     -- the dominant kernel (bf16 MFMA GEMM): algorithmic FLOPs of its launches / their HIP-event time
                  (events recorded inside the library on the launch stream during the timed steps)
  cpu_baseline -- the oracle (CPU restatement pinned to the reference) timed on this host's cores on a bounded sample
  parity       -- the cpu_baseline tiles through the HIP path in the benchmarked precision against the oracle
  exact_mode   -- (default line only) the same workload in the two-plane f16 mode (--precision f16x2: f32-grade arithmetic on the f16
                  matrix pipe), timed separately: the north_star's ">= 50 Mpix/s AND arg-max-identical" measured in one mode
`--config 2|3|4|5` run BASELINE.json's other configurations (ViT-B/16 slide; L/14 GEM + outlier suppression; L/14 + JBU; H/14 fp8 +
cross-tile fusion) through the same steps; they are reported separately, never as the default line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TILE, STRIDE = 512, 256
# weak scaling default: tiles per rank per step = rows x cols of 512-pixel windows at stride 256, chosen per tower so that the 256-row output
# tiles of the persistent GEMM fill whole rounds of the 256 CUs (T tiles x N tokens / 256 row tiles x N/256 column tiles, every linear of a
# block): ViT-L/14 7 x 17 = 119 tiles -> 637 row tiles -> 9.95 / 29.86 / 39.81 rounds (128 tiles: 10.70 / 32.11 / 42.81, i.e. a last round
# with 70 % / 11 % / 81 % of the chip busy).  The same workload per tile; only the launch size is picked for the machine.
TILE_GRID = {"ViT-L-14": (7, 17), "ViT-B-16": (7, 18), "ViT-H-14": (5, 19)}
TILE_COLS, TILE_ROWS_PER_RANK = 17, 7        # the ViT-L/14 default: 119 tiles per rank per step (a 2048 x 4608 scene band)
STRONG_SCENE = 8192                          # strong scaling: one 8192 x 8192 scene = 31 x 31 = 961 tiles
PEAK_BF16_TFLOPS = 2500.0                    # MI355X dense bf16 / f16 MFMA (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0
SIM = dict(similarity_weight=1.0, temperature=1.0, add_self_similarity=True)

# BASELINE.json `configs` as bench workloads (1-based as the tests name them; None = the headline line).  names = configs/cls_*.txt.
WORKLOADS = {
    None: dict(vit="ViT-L-14", model_type="Experimental", sim=True, outlier=30, debias=0.2, names="cls_potsdam.txt", prob_thd=0.1, bg_idx=5,
               upsampler="none", ctf=False, precision="bf16",
               label="ViT-L/14, model_type=Experimental + similarity enhancement + outlier suppression k=30 + global debias 0.2, 8 Potsdam queries / 6 classes"),
    2: dict(vit="ViT-B-16", model_type="Experimental", sim=True, outlier=30, debias=0.2, names="cls_potsdam.txt", prob_thd=0.1, bg_idx=5,
            upsampler="none", ctf=False, precision="bf16",
            label="BASELINE configs[1]: ViT-B/16, model_type=Experimental + similarity enhancement + outlier suppression k=30 + global debias 0.2, 8 Potsdam queries / 6 classes"),
    3: dict(vit="ViT-L-14", model_type="GEM", sim=False, outlier=30, debias=0.0, names="cls_loveda.txt", prob_thd=0.3, bg_idx=0,
            upsampler="none", ctf=False, precision="bf16",
            label="BASELINE configs[2]: ViT-L/14, GEM dual-stream blocks (depth 7) + outlier suppression k=30 in ONE forward (DESIGN.md section 7), 9 LoveDA queries / 7 classes"),
    4: dict(vit="ViT-L-14", model_type="SegEarth", sim=False, outlier=0, debias=0.2, names="cls_isaid.txt", prob_thd=0.4, bg_idx=0,
            upsampler="jbu_one", ctf=False, precision="bf16",
            label="BASELINE configs[3]: ViT-L/14 + SimFeatUp JBU (jbu_one) to per-pixel logits, global debias 0.2, 16 iSAID queries"),
    5: dict(vit="ViT-H-14", model_type="SegEarth", sim=False, outlier=0, debias=0.2, names="cls_xBD.txt", prob_thd=0.0, bg_idx=0,
            upsampler="none", ctf=True, precision="fp8",
            label="BASELINE configs[4]: ViT-H/14, fp8 linears, CrossTileFusion('weighted', boundary 2, strength 0.3), global debias 0.2, 2 xBD queries"),
}
CTF_CFG = dict(fusion_mode="weighted", cache_boundary_width=2, fusion_strength=0.3)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=None, choices=[2, 3, 4, 5],
                    help="run one of BASELINE.json's other configurations (2: B/16 slide, 3: L/14 GEM + outlier suppression, 4: L/14 + JBU, "
                         "5: H/14 fp8 + cross-tile fusion) instead of the headline workload; reported separately")
    ap.add_argument("--precision", default=None, choices=["bf16", "f16", "f32", "fp8", "f16x2"],
                    help="bf16 = the headline configuration; f16x2 = two f16 planes per operand, three MFMAs per product: the fp32 reference's results "
                         "(arg-max identical up to fp32 ties) on the matrix pipe; f16 = IEEE f16 operands (the reference's own GPU arithmetic); "
                         "f32 = parity mode on the f32 MFMA; fp8 = ordinary-block linears on fp8 MFMA (reported separately, never the default)")
    ap.add_argument("--vit", default=None, choices=["ViT-B-16", "ViT-L-14", "ViT-H-14"], help="override the workload's tower")
    ap.add_argument("--cross-tile-fusion", action="store_true", help="fuse boundary strips between neighbouring tiles before the head (strips all-gathered over ranks)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --tile-rows x --tile-cols tiles per rank, the scene grows with the ranks; strong: one fixed --scene x --scene image, tiles partitioned over the ranks")
    ap.add_argument("--scene", type=int, default=STRONG_SCENE, help="strong scaling: side of the square scene in pixels (8192 -> 961 tiles)")
    ap.add_argument("--tile-cols", type=int, default=0, help="weak scaling: tiles per scene row (0 = the tower's default grid, TILE_GRID)")
    ap.add_argument("--tile-rows", type=int, default=0, help="weak scaling: tile rows per rank (0 = the tower's default grid)")
    ap.add_argument("--tiles-per-launch", type=int, default=0, help="0 = up to the tower's machine-filling launch size (TILE_GRID: 119 tiles at ViT-L/14) per launch (32 with the upsampler)")
    ap.add_argument("--streams", type=int, default=1, help="experiment: split a rank's tiles over this many HIP streams (tails of one half overlap the other)")
    ap.add_argument("--upsampler", default=None, choices=["none", "jbu_one", "jbu_stack"],
                    help="per-pixel logits through the SimFeatUp JBU upsampler (BASELINE configs[3]); multi-rank: halo tiles travel point to point")
    ap.add_argument("--tuning", type=int, default=None, help="experiment: a library tuning code for sg_set_gemm_config (34 = LayerNorm as its own pass, ...)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-mode", action="store_true", help="skip the second (f16x2) measurement of the default line")
    ap.add_argument("--no-self-check", action="store_true",
                    help="profiling runs only: skip the single-tile re-computations after the timed region, so that rocprofv3's per-kernel averages cover the timed launch shape alone")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsing ranks on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    return ap.parse_args(argv)


def self_launch(args) -> None:
    """`python bench.py --gpus N` without a launcher: start N ranks as child processes (torch.distributed.run) BEFORE this process touches
    the GPU, and exit with their return code.  (Never a re-exec: a process that initialised the GPU must not exec.)"""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def workload_of(args):
    w = dict(WORKLOADS[args.config])
    if args.precision is not None:
        w["precision"] = args.precision
    if args.vit is not None:
        w["vit"] = args.vit
    if args.upsampler is not None:
        w["upsampler"] = args.upsampler
    if args.cross_tile_fusion:
        w["ctf"] = True
    return w


def query_idx_of(names):
    from clip_decontamination_amd.segmentors import get_cls_idx
    words, qidx = get_cls_idx(os.path.join(ROOT, "configs", names))
    return list(qidx)


def build_pipeline(device, w, precision, tiles_per_launch):
    import torch
    from clip_decontamination_amd import weights as Wt
    from clip_decontamination_amd.engine import HipVisionTower, HipCLIP, SimilarityEnhancementModule, OutlierSuppressionModule
    from clip_decontamination_amd.pipeline import SegPipeline
    cfg = Wt.vit_config(w["vit"])
    tower = HipVisionTower(cfg, Wt.make_vit_weights(cfg, seed=0), precision=precision, device=device)
    if w["sim"]:
        tower.similarity_enhancer = SimilarityEnhancementModule(**SIM)
    if w["outlier"]:
        tower.outlier_suppressor = OutlierSuppressionModule(top_k=w["outlier"])
    qidx = query_idx_of(w["names"])
    text = torch.from_numpy(Wt.make_text_features(len(qidx), cfg.embed_dim))
    up = None
    if w["upsampler"] != "none":                         # BASELINE configs[3]: ViT-L/14 + SimFeatUp JBU (synthetic upsampler weights, seed 3)
        from clip_decontamination_amd.upsampler import HipJBU
        up = HipJBU(w["upsampler"], cfg.embed_dim, device, precision)
        up.load_state_dict(Wt.make_jbu_weights(w["upsampler"], cfg.embed_dim, seed=3))
    pipe = SegPipeline(HipCLIP(tower), text, torch.tensor(qidx), model_type=w["model_type"], ignore_residual=True,
                       global_debias_factor=w["debias"], prob_thd=w["prob_thd"], bg_idx=w["bg_idx"], apply_similarity_enhancement=w["sim"],
                       tiles_per_launch=tiles_per_launch, upsampler=up, cross_tile_fusion=CTF_CFG if w["ctf"] else None)
    return cfg, pipe, qidx


def cpu_baseline(cfg, w, qidx, n_tiles=3, keep=None):
    """The oracle (port of the reference path, fp32, torch CPU) on a bounded sample of the same workload.  ``keep`` (a list)
    receives (tile, oracle logits [Q,512,512], oracle labels, oracle probabilities) of the timed tiles, so the same tiles can be pushed
    through the HIP path and compared (the `parity` object of the bench line)."""
    import torch
    from clip_decontamination_amd import weights as Wt
    from oracle import segment as OS, vit as OV
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))            # the GPU box gives one GPU a 16-core CPU share
    torch.set_num_threads(threads)
    wts = OV.to_torch(Wt.make_vit_weights(cfg, seed=0))
    text = torch.from_numpy(Wt.make_text_features(len(qidx), cfg.embed_dim))
    o = OS.SegOracle(cfg, wts, text, torch.tensor(qidx), model_type=w["model_type"], global_debias_factor=w["debias"],
                     similarity_cfg=SIM if w["sim"] else None, outlier_cfg=dict(top_k=w["outlier"]) if w["outlier"] else None,
                     prob_thd=w["prob_thd"], bg_idx=w["bg_idx"], slide_crop=TILE, slide_stride=STRIDE, segearth_variant=w["model_type"] == "GEM")
    tiles = torch.from_numpy(Wt.normalize_tiles(Wt.make_tiles_u8(n_tiles + 1, TILE, seed=1234, smooth=True)))
    with torch.no_grad():
        o.postprocess(o.forward_slide(tiles[:1])[0])                  # warm-up tile
        t0 = time.perf_counter()
        for i in range(1, n_tiles + 1):
            lg = o.forward_slide(tiles[i:i + 1])[0]
            probs, lab = o.postprocess(lg)
            if keep is not None:
                keep.append((tiles[i], lg, lab, probs))
        dt = time.perf_counter() - t0
    return {"value": n_tiles * TILE * TILE / dt / 1e6, "unit": "Mpix/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_tiles} tiles of {TILE}x{TILE} after 1 warm-up, {dt / n_tiles:.2f} s/tile, fp32 torch CPU"}, o


def parity_of(pipe, kept, oracle, device, mode):
    """The kept oracle tiles through the HIP path: max |dlogit|, label agreement, and how many of the differing pixels sit at a TIE of the
    oracle itself (two classes, or the arg-max probability against prob_thd, closer than 1e-4 in probability -- the bar of
    tests/test_gpu_configs.py::compare: an arg-max there is not defined to better than the fp32 rounding of either side)."""
    worst, agree, npx, off_tie = 0.0, 0, 0, 0
    for tile, lg_ref, lab_ref, probs in kept:
        lg = pipe.forward_slide(tile.to(device), STRIDE, TILE)[0]
        _, lab = pipe.postprocess(lg, want_probs=False)
        lab = lab.cpu()
        worst = max(worst, (lg.cpu() - lg_ref).abs().max().item())
        bad = (lab != lab_ref)[0]
        agree += int((~bad).sum())
        npx += lab_ref.numel()
        if bool(bad.any()):
            pmax = probs.max(0)[0][bad]
            ours = probs[:, bad].gather(0, lab[0][bad][None])[0]
            tie = ((pmax - ours).abs() < 1e-4) | ((pmax - oracle.prob_thd).abs() < 1e-4)
            off_tie += int((~tie).sum())
    return {"max_dlogit": worst, "label_agreement": agree / max(npx, 1), "labels_differing": npx - agree,
            "labels_differing_away_from_oracle_ties": off_tie, "argmax_identical_up_to_fp32_ties": off_tie == 0, "n_tiles": len(kept),
            "against": "oracle (fp32 CPU restatement pinned to the reference), the cpu_baseline tiles, per-pixel logits after stitch; a tie = the "
                       "oracle's own top-2 class probabilities (or arg-max probability and prob_thd) within 1e-4",
            "mode": mode}


def main():
    args = parse_args()
    self_launch(args)                                    # N > 1 and no launcher: children do the work, this process never touches the GPU
    import torch
    from clip_decontamination_amd import weights as Wt

    if args.tuning is not None:
        from clip_decontamination_amd import _lib as _sg_lib
        _sg_lib.load().sg_set_gemm_config(int(args.tuning))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or without a launcher")
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
    world_seen = dist.get_world_size() if dist is not None else 1

    from clip_decontamination_amd import _lib, ops
    from clip_decontamination_amd.engine import compute_padsize
    from clip_decontamination_amd.pipeline import band_plan, exchange_halo_tiles, gather_blocks, partition, tile_windows
    lib = _lib.load()
    w = workload_of(args)
    precision = w["precision"]
    jbu = w["upsampler"] != "none"
    if args.tile_rows <= 0 or args.tile_cols <= 0:
        # per-pixel logits (upsampler) are 11 MB per tile: 32 tiles per rank per step; otherwise the tower's machine-filling grid
        args.tile_rows, args.tile_cols = (4, 8) if jbu else TILE_GRID[w["vit"]]
    tile_cols, tile_rows = args.tile_cols, args.tile_rows

    # ---- scene geometry: weak = tile_cols x (tile_rows * world) tiles; strong = one fixed square scene ---------------------------------
    if args.scaling == "strong":
        H = W = int(args.scene)
    else:
        H = STRIDE * (tile_rows * world - 1) + TILE
        W = STRIDE * (tile_cols - 1) + TILE
    wins = tile_windows(H, W, (STRIDE, STRIDE), (TILE, TILE))
    hg = max(H - TILE + STRIDE - 1, 0) // STRIDE + 1
    wg = max(W - TILE + STRIDE - 1, 0) // STRIDE + 1
    T_all = len(wins)
    lo, hi = partition(T_all, world, rank)               # the pipeline's own block partition of the raster tile list
    my = wins[lo:hi] if hi > lo else [wins[0]]            # more ranks than tiles: a dummy so shapes agree
    n_mine = hi - lo
    if args.tiles_per_launch <= 0:
        args.tiles_per_launch = 32 if jbu else TILE_GRID[w["vit"]][0] * TILE_GRID[w["vit"]][1]      # the machine-filling launch size of the tower
    cfg, pipe, qidx = build_pipeline(device, w, precision, args.tiles_per_launch)
    plan = band_plan(wins, H, world)                                  # canvas bands + the tile range each band needs (pipeline.py)
    yb, need = plan
    y_lo, y_hi = yb[rank], yb[rank + 1]
    # every rank only holds the scene rows its tiles touch.  The scene is periodic (a 1024 x 1024 synthetic image repeated), generated on
    # the device, identical on every rank: scene[y, x] = base[y % 1024, x % 1024]
    slab_lo, slab_hi = my[0][0], my[-1][1]
    base = torch.from_numpy(Wt.make_tiles_u8(1, 1024, seed=1234, smooth=True)[0]).to(device)
    slab = base[torch.arange(slab_lo, slab_hi, device=device) % 1024][:, torch.arange(W, device=device) % 1024].contiguous()
    my_local = [(y1 - slab_lo, y2 - slab_lo, x1, x2) for (y1, y2, x1, x2) in my]
    win_all = torch.tensor(wins, dtype=torch.int32, device=device)
    P = cfg.patch
    l, r, t, b = compute_padsize(TILE, TILE, P)                       # (3, 3, 3, 3) at patch 14
    up = (TILE + t + b, TILE + l + r)
    group = dist.group.WORLD if world > 1 else None

    side = [torch.cuda.Stream(device=device) for _ in range(args.streams)] if args.streams > 1 else []

    def tower(p):
        if w["ctf"]:                                                  # tokens of all of this rank's tiles -> strip exchange -> fusion -> head
            if world > 1:
                return p._fused_tile_logits(slab, my_local, (TILE, TILE), (hg, wg), T_all, world, rank, group, n_real=n_mine)
            return p.tile_logits(slab, my_local, (TILE, TILE), grid_of_tiles=(hg, wg))
        if not side:
            return p.tile_logits(slab, my_local, (TILE, TILE))
        cur = torch.cuda.current_stream()
        per = (len(my_local) + len(side) - 1) // len(side)
        parts = []
        for i, st in enumerate(side):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                parts.append(p.tile_logits(slab, my_local[i * per:(i + 1) * per], (TILE, TILE)))
        for st, part in zip(side, parts):
            cur.wait_stream(st)
            part.record_stream(cur)
        return torch.cat(parts, 0)

    last = {}

    def step(p=None):
        p = pipe if p is None else p
        tl = tower(p)[:n_mine]                                        # [tiles of this rank, Q, 37, 37]  (or [.., Q, 592, 592] with the upsampler)
        last["tl"] = tl
        # The product's own sharding helpers (pipeline.sharded_canvas_band does exactly this on a scene every rank holds; here every
        # rank only holds the scene rows of its own tiles, so the steps are spelled out):
        #   patch-grid logits (44 kB per tile): ONE all_gather_into_tensor rebuilds the tile list everywhere (RCCL over xGMI);
        #   per-pixel logits (upsampler, 11 MB per tile): only the tiles that straddle a band edge travel, point to point.
        a_t, b_t = need[rank]
        if world > 1 and not jbu:
            tiles = gather_blocks(tl, T_all, world, rank, group)[a_t:b_t]
        elif world > 1:
            tiles, a_t = exchange_halo_tiles(tl, wins, world, rank, group, plan)
        else:
            tiles = tl
        if y_hi <= y_lo:
            return None
        # this rank stitches + labels ITS band of canvas rows
        w_band = win_all[a_t:a_t + tiles.shape[0]].clone()
        w_band[:, 0:2] -= y_lo
        canvas = ops.stitch(tiles, w_band, up, (t, l), (y_hi - y_lo, W))
        probs, labels = p.postprocess(canvas, want_probs=False)
        return labels

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(p, steps, warmup):
        for _ in range(warmup):
            step(p)
        barrier()
        check(lib.sg_profile_enable(1 << 16))
        t0 = time.perf_counter()
        for _ in range(steps):
            step(p)
        barrier()
        dt_ = time.perf_counter() - t0
        lib.sg_profile_disable()
        if world > 1:
            tmax = torch.tensor([dt_], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax.item())
        return dt_

    dt = timed(pipe, args.steps, args.warmup)

    def prof(cat):
        ms, fl, n, dr = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
        lib.sg_profile_read(cat, C.byref(ms), C.byref(fl), C.byref(n), C.byref(dr))
        return ms.value, fl.value, n.value, dr.value

    def pmc_traffic():
        """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (profiles/pmc_traffic.json, written by
        tools/pmc_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command) together with the git
        head those passes ran at; (None, None, None) if absent."""
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                tr = json.load(f)
            return tr.get("gemm_bf16_persist"), tr.get("note"), tr.get("measured_at_git_head")
        except (OSError, ValueError):
            return None, None, None

    # ---- correctness of the TIMED launch shape (outside the timed region) --------------------------------------------------
    # (a) first / middle / last tile of this rank recomputed ALONE (one-tile launches) must equal their rows of the batched launch
    def self_check():
        tl = last["tl"]
        picks = sorted({0, n_mine // 2, n_mine - 1}) if n_mine > 0 else []
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
        if world > 1:                                     # every rank learns the worst case, so that all of them stop together (no rank left in a barrier)
            red = torch.tensor([worst, 0.0 if exact else 1.0], dtype=torch.float64, device=device)
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
            worst, exact = float(red[0].item()), bool(red[1].item() == 0.0)
        if not (worst < 1e-3):
            raise SystemExit(f"bench self-check FAILED: a tile of a multi-tile launch differs from the same tile run alone by {worst} (worst over ranks)")
        return {"tiles_checked": len(picks), "max_dlogit_batched_vs_alone": worst, "bit_identical": exact}

    # cross-tile fusion couples a tile with its neighbours: a tile run alone is a different computation, so the check does not apply
    batched_check = None if (args.no_self_check or w["ctf"]) else self_check()

    tiles_total = T_all * args.steps
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
        h_ms, h_fl, h_n, h_drop = prof(7)                  # two-plane f16 GEMM launches (only with --precision f16x2)
        traffic, traffic_note, traffic_head = pmc_traffic()
        if precision == "f16x2":
            # dominant kernel of the exact mode: the two-plane GEMM.  `achieved` is ALGORITHMIC (2 M N K per launch); the kernel issues three
            # MFMAs per product, so the matrix pipe runs at 3x that figure and `frac` can reach 1/3 at most -- both are stated.
            achieved = h_fl / (h_ms * 1e-3) / 1e12 if h_ms > 0 else 0.0
            roof = {"bound": "mfma", "kernel": "two-plane f16 GEMM (gemm_h2)", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None,
                    "matrix_pipe_issued_tflops": round(3 * achieved, 2), "frac_of_issued_mfma": round(3 * achieved / PEAK_BF16_TFLOPS, 4),
                    "note": "achieved = algorithmic 2 M N K of the launches / their HIP-event time; every product is hi.hi + lo.hi + hi.lo = three f16 MFMAs",
                    "launches": h_n, "avg_launch_us": round(h_ms * 1e3 / max(h_n, 1), 2),
                    "algorithmic_gflop_per_launch": round(h_fl / max(h_n, 1) / 1e9, 3), "events_dropped": h_drop,
                    "attention": {"kernel": "attn_kernel<two-plane>", "achieved": round(a_fl / (a_ms * 1e-3) / 1e12, 2) if a_ms > 0 else 0.0,
                                  "launches": a_n, "avg_launch_us": round(a_ms * 1e3 / max(a_n, 1), 2)},
                    "share_of_step_time": {"gemm_two_plane": round(h_ms / (dt * 1e3), 4), "attention": round(a_ms / (dt * 1e3), 4)}}
        else:
            achieved = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
            roof = {"bound": "mfma", "kernel": "gemm_bf16_persist", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                    "traffic": traffic if (args.config is None and precision == "bf16") else None,
                    "traffic_note": traffic_note, "traffic_measured_at_git_head": traffic_head,
                    "launches": g_n, "avg_launch_us": round(g_ms * 1e3 / max(g_n, 1), 2),
                    "algorithmic_gflop_per_launch": round(g_fl / max(g_n, 1) / 1e9, 3), "events_dropped": g_drop,
                    "by_epilogue": {"note": "the same kernel's three instantiations; the folded ones carry the LayerNorm passes of the blocks "
                                            "(the separate pass was 220 us per LayerNorm), so their FLOP rate is not comparable with a plain GEMM's",
                                    "plain": _part(g_ms - c_ms - p_ms, g_fl - c_fl - p_fl, g_n - c_n - p_n),
                                    "folded_ln_consumer": _part(c_ms, c_fl, c_n), "folded_ln_producer": _part(p_ms, p_fl, p_n)},
                    "attention": {"kernel": "attn_kernel", "achieved": round(a_fl / (a_ms * 1e-3) / 1e12, 2) if a_ms > 0 else 0.0,
                                  "launches": a_n, "avg_launch_us": round(a_ms * 1e3 / max(a_n, 1), 2)},
                    "fp8_gemm": ({"kernel": "gemm_bf16_ring<256,256,2,4,2,FP8> / gemm_fp8_persist", "achieved": round(f_fl / (f_ms * 1e-3) / 1e12, 2), "launches": f_n,
                                  "peak": PEAK_FP8_TFLOPS, "frac": round(f_fl / (f_ms * 1e-3) / 1e12 / PEAK_FP8_TFLOPS, 4),
                                  "share_of_step_time": round(f_ms / (dt * 1e3), 4)} if f_n else None),
                    "share_of_step_time": {"gemm_bf16_persist": round(g_ms / (dt * 1e3), 4), "other_bf16_gemm": round(o_ms / (dt * 1e3), 4),
                                           "attention": round(a_ms / (dt * 1e3), 4)}}
        vit_label = {"ViT-B-16": "ViT-B/16", "ViT-L-14": "ViT-L/14", "ViT-H-14": "ViT-H/14"}[w["vit"]]
        out = {
            "metric": f"segmented Mpix/sec {vit_label} 512-tile slide",
            "value": round(value, 3), "unit": "Mpix/s",
            "n_gpus": world, "world_size_seen": world_seen, "backend": (args.backend + (" (ranks share cuda:0)" if args.share_gpu else "")) if world > 1 else None,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": precision, "data": "synthetic",
            "config": {"workload": w["label"] + f"; 512x512x3 uint8 tiles at stride 256 (padded to {up[0]}, N={(up[0] // P) ** 2 + 1} tokens), slide stitch + arg-max labels",
                       "baseline_config": args.config if args.config is not None else "headline (metric of BASELINE.json)",
                       "tiles_per_step_per_gpu": n_mine, "tiles_per_step": T_all, "tiles_per_launch": args.tiles_per_launch,
                       "scene": f"{H}x{W}", "upsampler": w["upsampler"], "cross_tile_fusion": bool(w["ctf"]),
                       "partition": (f"raster tile list in contiguous blocks over {world} rank(s), "
                                     + ("two all-gathers of boundary strips, " if w["ctf"] and world > 1 else "")
                                     + ("point-to-point halo exchange of per-pixel tile logits" if jbu else "all-gather of patch-grid logits")
                                     + ", band-local stitch + labels")},
            "roofline": roof,
        }
        out["self_check"] = batched_check
        if jbu or args.config in (3, 4, 5) or w["ctf"]:
            out["cpu_baseline"] = None
            out["cpu_baseline_note"] = ("not timed for this configuration (the oracle's JBU needs ~12 GB and ~70 s per 512-pixel tile, its GEM forward ~9 s per tile: "
                                        "BASELINE.md); parity of these paths at their real model shapes: tests/test_gpu_configs.py, tests/test_gpu_fp8.py")
        elif world == 1 and not args.no_cpu_baseline:
            kept = []
            out["cpu_baseline"], oracle = cpu_baseline(cfg, w, qidx, keep=kept)
            # (b) the CPU-baseline tiles through the HIP path (same precision mode as the timed run): logits vs the oracle's
            out["parity"] = parity_of(pipe, kept, oracle, device, precision)
            # (c) the default line also carries the EXACT mode: the same workload in two-plane f16 (f32-grade results), timed on its own
            if args.config is None and precision == "bf16" and not args.no_exact_mode and args.scaling == "weak" and not side:
                pipe.visual._ws = None                                    # the bf16 tower's workspace arena (6.5 GB) is no longer needed
                torch.cuda.empty_cache()
                _, pipe_x, _ = build_pipeline(device, w, "f16x2", args.tiles_per_launch)
                xs = max(2, min(args.steps, 3))
                dtx = timed(pipe_x, xs, 1)
                hx_ms, hx_fl, hx_n, _ = prof(7)
                ax_ms, ax_fl, ax_n, _ = prof(1)
                out["exact_mode"] = {
                    "dtype": "f16x2", "value": round(T_all * xs * TILE * TILE / dtx / 1e6, 3), "unit": "Mpix/s", "steps": xs, "warmup": 1,
                    "ms_per_step": round(dtx / xs * 1e3, 3),
                    "what": "the same workload with every GEMM / attention operand held as two f16 planes (hi + lo) and every product issued as three "
                            "f16 MFMAs into one f32 accumulator (SG_PREC_F16X2): the fp32 reference's results on the matrix pipe",
                    "gemm_two_plane": {"achieved_algorithmic_tflops": round(hx_fl / (hx_ms * 1e-3) / 1e12, 2) if hx_ms > 0 else 0.0, "launches": hx_n,
                                       "avg_launch_us": round(hx_ms * 1e3 / max(hx_n, 1), 2), "share_of_step_time": round(hx_ms / (dtx * 1e3), 4)},
                    "attention_two_plane": {"achieved_algorithmic_tflops": round(ax_fl / (ax_ms * 1e-3) / 1e12, 2) if ax_ms > 0 else 0.0, "launches": ax_n,
                                            "avg_launch_us": round(ax_ms * 1e3 / max(ax_n, 1), 2), "share_of_step_time": round(ax_ms / (dtx * 1e3), 4)},
                    "parity": parity_of(pipe_x, kept, oracle, device, "f16x2"),
                }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def check(rc):
    if rc != 0:
        raise RuntimeError("libsegearth_hip call failed")


if __name__ == "__main__":
    main()
