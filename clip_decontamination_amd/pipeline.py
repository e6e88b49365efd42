"""Sliding-window segmentation pipeline on top of the HIP library (host logic only).

Mirrors reference segmentor.py: ``forward_feature`` (:286-392), ``forward_slide`` (:394-451),
``postprocess_result`` (:475-499) -- but tiles are cropped / padded / embedded on the device in batches,
stitched write-once, and (opt-in: ``tile_group``) partitioned over the ranks of a torch.distributed group with ONE
all-gather of the per-tile patch-grid logit maps; per-pixel logit maps (JBU upsampler) travel point to
point instead, only the tiles that straddle a canvas-band edge.  Every rank stitches and labels its own canvas band.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import ops
from .engine import HipCLIP, compute_padsize


def tile_windows(H: int, W: int, stride: Tuple[int, int], crop: Tuple[int, int]) -> List[Tuple[int, int, int, int]]:
    """Raster-order windows (y1, y2, x1, x2); the last window of each axis is shifted back inside the
    image (reference segmentor.py:411-423)."""
    hs, ws = stride
    hc, wc = crop
    hg = max(H - hc + hs - 1, 0) // hs + 1
    wg = max(W - wc + ws - 1, 0) // ws + 1
    out = []
    for hi in range(hg):
        for wi in range(wg):
            y2 = min(hi * hs + hc, H)
            x2 = min(wi * ws + wc, W)
            out.append((max(y2 - hc, 0), y2, max(x2 - wc, 0), x2))
    return out


def partition(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition of the raster tile list: [lo, hi) for ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _is_nccl(group) -> bool:
    return torch.distributed.get_backend(group) == "nccl"


def resolve_tile_group(group):
    """Tile sharding is OPT-IN: ``None`` = this process handles the whole scene (the reference's own multi-GPU launch shards IMAGES
    over ranks, dist_test.sh -> mmengine DefaultSampler, so an initialised default process group must NOT trigger tile sharding);
    ``"world"`` = the default group; otherwise a torch.distributed ProcessGroup."""
    if group is None:
        return None
    if isinstance(group, str):
        if group != "world":
            raise ValueError(f"tile_group must be None, 'world' or a ProcessGroup, got {group!r}")
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            raise RuntimeError("tile_group='world' needs an initialised torch.distributed process group")
        return torch.distributed.group.WORLD
    return group


def check_same_scene(desc: Sequence[int], device, group) -> None:
    """Every rank of a tile group must hold the SAME scene geometry (H, W, tile count, crop, stride ...): ranks that hold
    different images would silently mix tiles of different scenes in the gather.  One tiny all-gather per scene."""
    world = torch.distributed.get_world_size(group)
    dev = device if _is_nccl(group) else torch.device("cpu")
    mine = torch.tensor(list(desc), dtype=torch.int64, device=dev)
    allv = torch.empty(world * mine.numel(), dtype=torch.int64, device=dev)
    torch.distributed.all_gather_into_tensor(allv, mine, group=group)
    allv = allv.reshape(world, -1).cpu()
    if not bool((allv == allv[0]).all()):
        raise RuntimeError("tile sharding: the ranks of the tile group hold different scenes "
                           f"(per-rank [H, W, tiles, crop_h, crop_w, stride_h, stride_w, Q]: {allv.tolist()}); "
                           "every rank must be given the same image")


def gather_blocks(local: torch.Tensor, n_items: int, world: int, rank: int, group=None) -> torch.Tensor:
    """All-gather of per-rank blocks of the raster tile list (``partition``): ``local`` holds this rank's items (possibly followed
    by padding); returns the full [n_items, ...] list on every rank.  Blocks are padded to ceil(n_items / world) so the
    collective is fixed-size: ONE all_gather_into_tensor (RCCL over xGMI on GPUs) into a preallocated buffer, which IS the result
    when the tile count divides evenly."""
    lo, hi = partition(n_items, world, rank)
    t_pad = (n_items + world - 1) // world
    tail = tuple(local.shape[1:])
    buf = local.new_empty((world * t_pad,) + tail)
    block = local.new_zeros((t_pad,) + tail)
    block[:hi - lo] = local[:hi - lo]
    torch.distributed.all_gather_into_tensor(buf, block, group=group)
    if n_items == world * t_pad:
        return buf
    parts = []
    for r in range(world):
        a, b = partition(n_items, world, r)
        parts.append(buf[r * t_pad:r * t_pad + (b - a)])
    return torch.cat(parts, 0)


def band_plan(wins: Sequence[Tuple[int, int, int, int]], H: int, world: int):
    """Canvas bands for a band-local stitch: rank r owns canvas rows [yb[r], yb[r+1]) where yb[r] is the first row of its first
    tile (raster order, ``partition``); need[r] = (a, b) is the contiguous range of tile ids that overlap the band.  Because y1 is
    non-decreasing in raster order, need[r] = [a_r, hi_r) with a_r <= lo_r: a rank only ever needs tiles of LOWER ranks (the
    halo), never of higher ones."""
    T = len(wins)
    yb = []
    for r in range(world):
        lo, hi = partition(T, world, r)
        yb.append(0 if r == 0 else (wins[lo][0] if hi > lo else H))
    yb.append(H)
    for r in range(world - 1, 0, -1):                      # empty blocks sit at the end; keep yb monotone
        yb[r] = min(yb[r], yb[r + 1])
    need = []
    for r in range(world):
        lo, hi = partition(T, world, r)
        if yb[r + 1] <= yb[r]:
            need.append((lo, lo))
            continue
        a = lo
        while a > 0 and wins[a - 1][1] > yb[r]:
            a -= 1
        need.append((a, hi))
    return yb, need


def exchange_halo_tiles(local: torch.Tensor, wins, world: int, rank: int, group=None, plan=None) -> Tuple[torch.Tensor, int]:
    """Band-local stitch, step 1: every rank receives the tile logits of LOWER ranks that overlap its canvas band, point to
    point (xGMI is point-to-point: only the tiles that straddle a band edge travel, instead of all-gathering [T,Q,S,S] per-pixel
    logits to everyone).  Returns (tile logits of tiles [a, hi) in raster order, a)."""
    T = len(wins)
    yb, need = plan if plan is not None else band_plan(wins, wins[-1][1], world)
    lo, hi = partition(T, world, rank)
    owner = [r for r in range(world) for _ in range(partition(T, world, r)[1] - partition(T, world, r)[0])]
    dist = torch.distributed
    peer = (lambda r: r) if group is None or group is dist.group.WORLD else (lambda r: dist.get_global_rank(group, r))
    staged = (not _is_nccl(group)) and local.is_cuda      # gloo moves host memory only
    a, b = need[rank]
    halo = local.new_empty((lo - a,) + tuple(local.shape[1:]), device="cpu" if staged else local.device)
    ops_, keep = [], []
    src = local.cpu() if staged else local
    for r in range(rank + 1, world):                      # my tiles that a higher rank's band needs
        ra, _ = need[r]
        s0, s1 = max(ra, lo), hi
        if partition(T, world, r)[1] > partition(T, world, r)[0] and s1 > s0 and ra < hi:
            t = src[s0 - lo:s1 - lo].contiguous()
            keep.append(t)
            ops_.append(dist.P2POp(dist.isend, t, peer(r), group))
    t0 = a
    while t0 < lo:                                        # tiles [a, lo) come from their owners, one message per owner
        o = owner[t0]
        t1 = min(partition(T, world, o)[1], lo)
        ops_.append(dist.P2POp(dist.irecv, halo[t0 - a:t1 - a], peer(o), group))
        t0 = t1
    if ops_:
        for w in dist.batch_isend_irecv(ops_):
            w.wait()
    halo = halo.to(local.device) if staged else halo
    return (torch.cat([halo, local[:hi - lo]], 0) if lo > a else local[:hi - lo]), a


def sharded_cross_tile_fusion(tokens: torch.Tensor, steps, n_tiles: int, world: int, rank: int, group=None) -> torch.Tensor:
    """Cross-tile boundary fusion when the raster tile list is partitioned over ranks.  ``tokens`` [n_local, n, C]
    are this rank's PRE-fusion patch tokens; a tile needs the right columns of its left neighbour and the final bottom rows of
    its upper neighbour, which may live on another rank: two all-gathers of packed strips (a few hundred kB per tile edge)
    replace the reference's per-process boundary cache.  ``steps`` = ops.CrossTileSteps (or a stand-in with pack/fuse/apply)."""
    lo, hi = partition(n_tiles, world, rank)
    assert tokens.shape[0] == hi - lo
    if hi == lo:                                   # more ranks than tiles: still take part in the collectives
        dummy = tokens.new_zeros((0, steps.strip_len(0), tokens.shape[-1]))
        gather_blocks(dummy, n_tiles, world, rank, group)
        gather_blocks(tokens.new_zeros((0, steps.strip_len(1), tokens.shape[-1])), n_tiles, world, rank, group)
        return tokens
    right_all = gather_blocks(steps.pack(tokens, lo, 0), n_tiles, world, rank, group)
    left_res = steps.fuse(tokens, lo, right_all, 0)
    bottom_all = gather_blocks(steps.pack(tokens, lo, 1, left_res), n_tiles, world, rank, group)
    top_res = steps.fuse(tokens, lo, bottom_all, 1)
    return steps.apply(tokens, lo, left_res, top_res)


def launch_chunks(n_tiles: int, limit: int) -> List[Tuple[int, int]]:
    """[start, stop) ranges of the tile list, one per launch of the tower: as few launches as `limit` tiles per launch allows, of EQUAL size
    (33 tiles at a limit of 32 run as 17 + 16, not 32 + 1 -- a one- or two-tile launch leaves most of the chip idle)."""
    if n_tiles <= 0:
        return []
    limit = max(1, int(limit))
    n = -(-n_tiles // limit)
    base, extra = divmod(n_tiles, n)
    out, a = [], 0
    for i in range(n):
        b = a + base + (1 if i < extra else 0)
        out.append((a, b))
        a = b
    return out


class SegPipeline:
    def __init__(self, net: HipCLIP, text: torch.Tensor, query_idx: torch.Tensor, model_type: str = "SegEarth",
                 ignore_residual: bool = True, cls_token_lambda: float = 0.0, global_debias_factor: float = 0.0,
                 logit_scale: float = 50.0, prob_thd: float = 0.0, bg_idx: int = 0, apply_similarity_enhancement: bool = False,
                 upsampler=None, tiles_per_launch: int = 32, cross_tile_fusion: Optional[dict] = None, apply_ctd: bool = False,
                 tile_group=None, apply_layer_fusion: bool = False, layer_fusion_lambda: float = 0.5):
        self.apply_layer_fusion, self.layer_fusion_lambda = bool(apply_layer_fusion), float(layer_fusion_lambda)
        self.net = net
        self.visual = net.visual
        self.device = self.visual.device
        self.text = text.to(device=self.device, dtype=torch.float32).contiguous()
        self.query_idx = query_idx.to(device=self.device, dtype=torch.int32).contiguous()
        self.num_queries = int(self.text.shape[0])
        self.num_classes = int(query_idx.max()) + 1
        self.model_type = model_type
        self.ignore_residual = ignore_residual
        self.cls_token_lambda = float(cls_token_lambda)
        self.global_debias_factor = float(global_debias_factor)
        self.logit_scale, self.prob_thd, self.bg_idx = float(logit_scale), float(prob_thd), int(bg_idx)
        self.apply_similarity_enhancement = apply_similarity_enhancement
        self.upsampler = upsampler
        self.tiles_per_launch = int(tiles_per_launch)
        # opt-in: the reference ships CrossTileFusion but never calls it; kwargs of its constructor
        self.cross_tile_fusion = cross_tile_fusion
        self.apply_ctd = bool(apply_ctd)                 # Cluster-Then-Debias (segmentor.py:339-365), SegmentorEx only
        # Tile sharding over ranks is opt-in (None = off, "world" = the default process group, or a ProcessGroup whose ranks
        # all hold the SAME scene).  It is never inferred from torch.distributed being initialised: under the reference's own
        # multi-GPU launch every rank holds a different image.
        self.tile_group = tile_group

    def _stitch(self, tile_logits, windows, up_hw, pad_tl, canvas_hw):
        return ops.stitch(tile_logits, windows, up_hw, pad_tl, canvas_hw)

    def _pre_head(self, tok, cls):
        """Global debias (+ CTD) when they cannot stay fused in the logits kernel: returns (tokens, remaining debias factor)."""
        f = self.global_debias_factor if cls is not None else 0.0
        if not getattr(self, "apply_ctd", False) or cls is None:
            return tok, f
        if f != 0:                                            # reference order: global debias, then cluster on the debiased features
            tok = ops.global_debias(tok, cls, f)
        tok, _ = ops.ctd_debias(tok, cls, eps=1.1, min_samples=11, factor=-1.5, want_labels=False, normalize_cls=True)
        return tok, 0.0

    # -- per-tile logits -----------------------------------------------------------------------------------
    def tile_logits(self, scene: torch.Tensor, windows: Sequence[Tuple[int, int, int, int]], tile_hw: Tuple[int, int],
                    scene_index: Optional[torch.Tensor] = None, grid_of_tiles: Optional[Tuple[int, int]] = None) -> torch.Tensor:
        """-> [T, Q, gh, gw] patch-grid logits (or [T, Q, H', W'] per-pixel logits with the JBU upsampler)."""
        v = self.visual
        P = v.cfg.patch
        th, tw = tile_hw
        l, r, t, b = compute_padsize(th, tw, P)
        gh, gw = (th + t + b) // P, (tw + l + r) // P
        opts = v.forward_opts(self.model_type, self.ignore_residual, self.apply_similarity_enhancement,
                              getattr(self, "apply_layer_fusion", False), getattr(self, "layer_fusion_lambda", 0.5))
        win = torch.tensor(list(windows), dtype=torch.int32, device=self.device).reshape(-1, 4)
        outs = []
        fuse = self.cross_tile_fusion is not None and grid_of_tiles is not None and grid_of_tiles[0] * grid_of_tiles[1] > 1
        if fuse:
            # boundary fusion couples neighbouring tiles: run the tower over the whole scene first, fuse, then the head
            cls_all, tok_all = [], []
            for i, j in launch_chunks(win.shape[0], self.tiles_per_launch):
                si = None if scene_index is None else scene_index[i:j]
                c_, t_ = v.forward_tiles(scene, win[i:j], tile_hw, opts, si)
                cls_all.append(c_); tok_all.append(t_)
            tok = ops.cross_tile_fusion(torch.cat(tok_all, 0), grid_of_tiles[0], grid_of_tiles[1], gh, gw,
                                        self.cross_tile_fusion.get("cache_boundary_width", 2),
                                        self.cross_tile_fusion.get("fusion_mode", "weighted"),
                                        self.cross_tile_fusion.get("fusion_strength", 0.3))
            cls = None if cls_all[0] is None else torch.cat(cls_all, 0)
            tok, f = self._pre_head(tok, cls)
            if self.upsampler is not None:
                # segmentor.py:368-372 after cross_tile_fusion.py:238-320: the FUSED tokens go through the upsampler (tokens -> fusion -> JBU -> logits)
                return self.upsampler.logits(tok, cls, scene, win, tile_hw, (l, t), (gh, gw), self.text, f, self.cls_token_lambda, scene_index,
                                             padded_hw=(th + t + b, tw + l + r))
            lg = ops.cosine_logits(tok, cls, self.text, f, self.cls_token_lambda if cls is not None else 0.0)
            return lg.reshape(win.shape[0], self.num_queries, gh, gw)
        for i, j in launch_chunks(win.shape[0], self.tiles_per_launch):
            w = win[i:j]
            si = None if scene_index is None else scene_index[i:j]
            cls, tok = v.forward_tiles(scene, w, tile_hw, opts, si)
            tok, f = self._pre_head(tok, cls)
            if self.upsampler is not None:
                outs.append(self.upsampler.logits(tok, cls, scene, w, tile_hw, (l, t), (gh, gw), self.text,
                                                  f, self.cls_token_lambda, si,
                                                  padded_hw=(th + t + b, tw + l + r)))
            else:
                lg = ops.cosine_logits(tok, cls, self.text, f, self.cls_token_lambda if cls is not None else 0.0)
                outs.append(lg.reshape(w.shape[0], self.num_queries, gh, gw))
        return torch.cat(outs, 0) if len(outs) > 1 else outs[0]

    # -- reference forward_slide -----------------------------------------------------------------------------
    def _geometry(self, scene, stride, crop):
        if scene.dtype == torch.uint8:
            H, W = int(scene.shape[0]), int(scene.shape[1])
        else:
            H, W = int(scene.shape[-2]), int(scene.shape[-1])
        stride = (stride, stride) if isinstance(stride, int) else tuple(stride)
        crop = (crop, crop) if isinstance(crop, int) else tuple(crop)
        wins = tile_windows(H, W, stride, crop)
        tile_hw = (wins[0][1] - wins[0][0], wins[0][3] - wins[0][2])
        P = self.visual.cfg.patch
        l, r, t, b = compute_padsize(tile_hw[0], tile_hw[1], P)
        up_hw = (tile_hw[0] + t + b, tile_hw[1] + l + r)
        hg = max(H - crop[0] + stride[0] - 1, 0) // stride[0] + 1
        wg = max(W - crop[1] + stride[1] - 1, 0) // stride[1] + 1
        return H, W, stride, crop, wins, tile_hw, (t, l), up_hw, (hg, wg)

    def forward_slide(self, scene: torch.Tensor, stride, crop, ori_shape=None, group=None) -> torch.Tensor:
        """scene: f32 [3,H,W] normalised planes or u8 [H,W,3].  Returns logits [1,Q,H_ori,W_ori] (on every rank of the tile
        group when sharding is on: ``group`` overrides ``self.tile_group``; None = this process does the whole scene)."""
        H, W, stride, crop, wins, tile_hw, pad_tl, up_hw, grid = self._geometry(scene, stride, crop)
        group = resolve_tile_group(group if group is not None else self.tile_group)
        if group is None:
            tl = self.tile_logits(scene, wins, tile_hw, grid_of_tiles=grid)
            win_dev = torch.tensor(wins, dtype=torch.int32, device=self.device)
            canvas = self._stitch(tl, win_dev, up_hw, pad_tl, (H, W))
        else:
            band, y0, yb = self.sharded_canvas_band(scene, stride, crop, group)
            canvas = self.gather_bands(band, yb, H, group)
        if ori_shape is not None and tuple(ori_shape) != (H, W):
            canvas = ops.resize_bilinear(canvas, tuple(ori_shape))
        return canvas.unsqueeze(0)

    def sharded_canvas_band(self, scene, stride, crop, group):
        """Tiles partitioned over the ranks of ``group`` -> this rank's band of the stitched canvas: (band [Q, rows, W], first row,
        band boundaries of all ranks).  Patch-grid logits (44 kB per tile) are rebuilt everywhere with ONE all-gather (RCCL over xGMI);
        per-pixel logits (upsampler: 16-32 MB per tile) travel point to point, only the tiles that straddle a band edge.  Either way each
        rank stitches ONLY its own band, so the tail scales with the ranks too.  The stitch itself is identical given identical tile
        logits (write-once: the covering tiles are averaged in raster order, and all of them are present); the tile logits of a rank
        equal the single process's bit for bit in parity mode (f32: tests/test_gpu_distributed.py), while the 2-byte modes may take a
        different kernel path for a different launch size (few tiles per launch: 128 x 128 GEMM tiles and a separate LayerNorm pass,
        DESIGN.md section 4), i.e. differ by 2-byte rounding (bounded in the same test file)."""
        H, W, stride, crop, wins, tile_hw, pad_tl, up_hw, grid = self._geometry(scene, stride, crop)
        world, rank = torch.distributed.get_world_size(group), torch.distributed.get_rank(group)
        check_same_scene([H, W, len(wins), crop[0], crop[1], stride[0], stride[1], self.num_queries], self.device, group)
        T = len(wins)
        lo, hi = partition(T, world, rank)
        plan = band_plan(wins, H, world)
        yb, need = plan
        if self.upsampler is None:
            tl = self.gather_tile_logits(scene, wins, tile_hw, world, rank, group, grid_of_tiles=grid)
            a, b = need[rank]
            tiles, first = tl[a:b], a
        else:
            mine = list(wins[lo:hi])
            if self.cross_tile_fusion is not None and T > 1:      # strips are exchanged first (two small all-gathers), then JBU on the fused tokens
                local = self._fused_tile_logits(scene, mine if mine else [wins[0]], tile_hw, grid, T, world, rank, group, n_real=hi - lo)
            else:
                local = self.tile_logits(scene, mine if mine else [wins[0]], tile_hw)[:hi - lo]
            tiles, first = exchange_halo_tiles(local, wins, world, rank, group, plan)
        y0, y1 = yb[rank], yb[rank + 1]
        if y1 <= y0:
            return tiles.new_zeros((self.num_queries, 0, W)), y0, yb
        w_band = torch.tensor(wins[first:first + tiles.shape[0]], dtype=torch.int32, device=self.device)
        w_band[:, 0:2] -= y0
        return self._stitch(tiles, w_band, up_hw, pad_tl, (y1 - y0, W)), y0, yb

    def gather_bands(self, band: torch.Tensor, yb, H: int, group) -> torch.Tensor:
        """[C, rows_r, W] bands of every rank -> the full [C, H, W] on every rank (one all_gather_into_tensor of bands padded to
        the tallest one).  Used for the drop-in's full-canvas return value and for the label map."""
        world = torch.distributed.get_world_size(group)
        hmax = max(yb[r + 1] - yb[r] for r in range(world))
        C_, W = band.shape[0], band.shape[-1]
        block = band.new_zeros((C_, hmax, W))
        block[:, :band.shape[1]] = band
        buf = band.new_empty((world, C_, hmax, W))
        torch.distributed.all_gather_into_tensor(buf.view(world * C_, hmax, W), block, group=group)   # output = concat along dim 0
        if world == 1:
            return buf[0, :, :H]
        return torch.cat([buf[r, :, :yb[r + 1] - yb[r]] for r in range(world)], 1)

    def segment_scene(self, scene: torch.Tensor, stride, crop, group=None, gather_labels: bool = True):
        """The whole path to labels for one scene.  Sharded: every rank computes its tiles, stitches + labels ITS band; only the
        int64 label band (8 B per pixel instead of 4Q) is gathered.  -> labels [1,H,W] (or (band labels, first row) when
        ``gather_labels`` is False)."""
        group = resolve_tile_group(group if group is not None else self.tile_group)
        if group is None:
            _, labels = self.postprocess(self.forward_slide(scene, stride, crop)[0], want_probs=False)
            return labels
        band, y0, yb = self.sharded_canvas_band(scene, stride, crop, group)
        if band.shape[1] > 0:
            _, lab = self.postprocess(band, want_probs=False)
        else:
            lab = torch.empty((1, 0, band.shape[-1]), dtype=torch.int64, device=band.device)
        if not gather_labels:
            return lab, y0
        return self.gather_bands(lab, yb, yb[-1], group)

    def gather_tile_logits(self, scene, wins, tile_hw, world, rank, group=None, grid_of_tiles=None) -> torch.Tensor:
        """Rank r computes a contiguous block of the raster tile list; one all-gather (RCCL over xGMI on GPUs)
        of equal-sized [T_pad, Q, gh, gw] blocks rebuilds the full list on every rank.  With cross-tile fusion the
        boundary strips are exchanged first (``sharded_cross_tile_fusion``)."""
        T = len(wins)
        lo, hi = partition(T, world, rank)
        mine = list(wins[lo:hi])
        fuse = getattr(self, "cross_tile_fusion", None) is not None and grid_of_tiles is not None and T > 1
        if fuse:
            local = self._fused_tile_logits(scene, mine if mine else [wins[0]], tile_hw, grid_of_tiles, T, world, rank, group,
                                            n_real=hi - lo)
        else:
            local = self.tile_logits(scene, mine if mine else [wins[0]], tile_hw)   # more ranks than tiles: a dummy so shapes agree
        return gather_blocks(local, T, world, rank, group)

    def _fused_tile_logits(self, scene, mine, tile_hw, grid_of_tiles, T, world, rank, group, n_real):
        v = self.visual
        P = v.cfg.patch
        l, r, t, b = compute_padsize(tile_hw[0], tile_hw[1], P)
        gh, gw = (tile_hw[0] + t + b) // P, (tile_hw[1] + l + r) // P
        opts = v.forward_opts(self.model_type, self.ignore_residual, self.apply_similarity_enhancement,
                              getattr(self, "apply_layer_fusion", False), getattr(self, "layer_fusion_lambda", 0.5))
        win = torch.tensor(list(mine), dtype=torch.int32, device=self.device).reshape(-1, 4)
        cls_all, tok_all = [], []
        for i, j in launch_chunks(win.shape[0], self.tiles_per_launch):
            c_, t_ = v.forward_tiles(scene, win[i:j], tile_hw, opts, None)
            cls_all.append(c_); tok_all.append(t_)
        tok = torch.cat(tok_all, 0)[:n_real].contiguous()
        cls = None if cls_all[0] is None else torch.cat(cls_all, 0)[:n_real].contiguous()
        cf = self.cross_tile_fusion
        steps = ops.CrossTileSteps(gh, gw, tok.shape[-1], cf.get("cache_boundary_width", 2), cf.get("fusion_mode", "weighted"),
                                   cf.get("fusion_strength", 0.3), grid_of_tiles[1])
        tok = sharded_cross_tile_fusion(tok, steps, T, world, rank, group)
        if n_real == 0:
            return tok.new_zeros((0, self.num_queries) + ((16 * gh, 16 * gw) if self.upsampler is not None else (gh, gw)))
        tok, f = self._pre_head(tok, cls)
        if self.upsampler is not None:                   # fused tokens -> JBU -> per-pixel logits of this rank's tiles (they travel as halo tiles)
            return self.upsampler.logits(tok, cls, scene, win[:n_real], tile_hw, (l, t), (gh, gw), self.text, f, self.cls_token_lambda, None,
                                         padded_hw=(tile_hw[0] + t + b, tile_hw[1] + l + r))
        lg = ops.cosine_logits(tok, cls, self.text, f, self.cls_token_lambda if cls is not None else 0.0)
        return lg.reshape(n_real, self.num_queries, gh, gw)

    # -- reference forward_feature (whole image / explicit logit size) ------------------------------------------
    def forward_feature(self, img: torch.Tensor, logit_size=None) -> torch.Tensor:
        """img [B,3,H,W] (H, W already patch multiples or not -- no padding is applied here, as in the
        reference) -> [B,Q,h,w] bilinearly resized to ``logit_size`` (default: the image size)."""
        B, _, H, W = img.shape
        P = self.visual.cfg.patch
        Hc, Wc = (H // P) * P, (W // P) * P           # conv1 (stride P) ignores a ragged remainder
        wins = [(0, Hc, 0, Wc)] * B
        idx = torch.arange(B, dtype=torch.int32, device=self.device)
        tl = self.tile_logits(img, wins, (Hc, Wc), idx)
        size = (H, W) if logit_size is None else tuple(logit_size)
        return torch.stack([ops.resize_bilinear(tl[i], size) for i in range(B)], 0)

    # -- reference postprocess_result --------------------------------------------------------------------------
    def postprocess(self, seg_logits: torch.Tensor, want_probs: bool = True):
        """seg_logits [Q,H,W] -> (probs [K,H,W], labels int64 [1,H,W])."""
        return ops.postprocess(seg_logits, self.query_idx, self.num_classes, self.logit_scale, self.prob_thd, self.bg_idx, want_probs)
