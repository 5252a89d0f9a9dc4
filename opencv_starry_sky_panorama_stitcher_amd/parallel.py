"""Multi-GPU panorama: images shard across GPUs, only the overlap bands of the pyramid sums are exchanged.

The reference has no distributed code; this is the MI355X-first design of section 8(e):

* ``shard_indices``: contiguous runs of images per rank (one process per GPU).
* Every rank feeds ITS frames into a multiband blender prepared with the GLOBAL panorama roi, which gives it partial
  sums (sum of (short)(L*w) per Laplacian level, sum of w) inside the bounding box ``bbox[r]`` of its frames' padded
  rectangles (the rectangles MultiBandBlender::feed snaps to multiples of 2^bands).
* ``plan_exchange``: for every ordered pair (s, d) the rectangle ``bbox[s] & bbox[d]``.  Rank s exports its partial sums
  of that rectangle for every level and sends them to d (point-to-point over the direct xGMI link: RCCL send/recv, no
  ring collective), d adds them (integer sums wrap mod 2^16 exactly as the sequential ``short +=`` does, so they are
  order independent; the f32 weight sums can differ from the single-GPU association by 1 ULP).
* After the exchange rank d holds COMPLETE sums over bbox[d] and normalises + collapses that rectangle locally.  The
  collapse of a rectangle is inexact only within 2*2^bands pixels of a rectangle edge that is not a panorama edge,
  and every frame roi lies at least 3*2^bands inside its padded rectangle -- so every pixel covered by one of rank
  d's frames comes out exactly as on a single GPU.  ``owner_map`` assigns each covered pixel to one such rank.

The plan and the protocol are backend agnostic (tests drive them with the CPU oracle over ``gloo``);
``HipOverlapExchange`` is the GPU implementation used by bench.py (torch.distributed ``nccl`` == RCCL).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

Rect = Tuple[int, int, int, int]


def shard_indices(n_images: int, world: int, rank: int) -> List[int]:
    """Contiguous run of image indices for ``rank`` (6 consecutive images per GPU in config 4)."""
    lo, hi = rank * n_images // world, (rank + 1) * n_images // world
    return list(range(lo, hi))


def effective_bands(pano_roi: Rect, num_bands: int) -> int:
    """MultiBandBlender::prepare: num_bands = min(requested, ceil(log2(max(w, h))))."""
    return min(int(num_bands), int(np.ceil(np.log(float(max(pano_roi[2], pano_roi[3]))) / np.log(2.0))))


def padded_pano_size(pano_roi: Rect, nb: int) -> Tuple[int, int]:
    m = 1 << nb
    w, h = pano_roi[2], pano_roi[3]
    return w + (m - w % m) % m, h + (m - h % m) % m


def padded_rect(corner: Tuple[int, int], size: Tuple[int, int], pano_roi: Rect, nb: int) -> Rect:
    """The rectangle MultiBandBlender::feed works on for one image, relative to the pano corner (blenders.cpp feed():
    grow by gap = 3*2^nb, clip to the padded pano, snap to multiples of 2^nb, shift back inside)."""
    m = 1 << nb
    pw, ph = padded_pano_size(pano_roi, nb)
    rx, ry, rbx, rby = pano_roi[0], pano_roi[1], pano_roi[0] + pw, pano_roi[1] + ph
    gap = 3 * m
    tnx, tny = max(rx, corner[0] - gap), max(ry, corner[1] - gap)
    bnx, bny = min(rbx, corner[0] + size[0] + gap), min(rby, corner[1] + size[1] + gap)
    tnx = rx + (((tnx - rx) >> nb) << nb)
    tny = ry + (((tny - ry) >> nb) << nb)
    width, height = bnx - tnx, bny - tny
    width += (m - width % m) % m
    height += (m - height % m) % m
    bnx, bny = tnx + width, tny + height
    dx, dy = max(bnx - rbx, 0), max(bny - rby, 0)
    tnx -= dx
    tny -= dy
    return (tnx - rx, tny - ry, width, height)


def rect_union(rects: Sequence[Rect]) -> Rect:
    x0 = min(r[0] for r in rects)
    y0 = min(r[1] for r in rects)
    x1 = max(r[0] + r[2] for r in rects)
    y1 = max(r[1] + r[3] for r in rects)
    return (x0, y0, x1 - x0, y1 - y0)


def rect_intersect(a: Rect, b: Rect):
    x0, y0 = max(a[0], b[0]), max(a[1], b[1])
    x1, y1 = min(a[0] + a[2], b[0] + b[2]), min(a[1] + a[3], b[1] + b[3])
    if x0 < x1 and y0 < y1:
        return (x0, y0, x1 - x0, y1 - y0)
    return None


@dataclass
class ExchangePlan:
    world: int
    nb: int
    pano_roi: Rect                       # global panorama roi (absolute coordinates)
    padded: Tuple[int, int]              # padded pano size
    bbox: List[Rect]                     # per rank, pano-relative, multiples of 2^nb
    image_rois: List[List[Rect]]         # per rank: its frames' rois, pano-relative (x, y, w, h)
    pairs: List[Tuple[int, int, Rect]] = field(default_factory=list)   # (src, dst, rect)

    def sends(self, rank: int):
        return [(d, r) for s, d, r in self.pairs if s == rank]

    def recvs(self, rank: int):
        return [(s, r) for s, d, r in self.pairs if d == rank]

    def bytes_sent(self, rank: int, float_mode: bool = False) -> int:
        per_px = (12 if float_mode else 6) + 4
        total = 0
        for _, r in self.sends(rank):
            for lvl in range(self.nb + 1):
                total += (r[2] >> lvl) * (r[3] >> lvl) * per_px
        return total


def plan_exchange(corners: Sequence[Tuple[int, int]], sizes: Sequence[Tuple[int, int]], owner: Sequence[int], world: int, num_bands: int) -> ExchangePlan:
    """corners/sizes: warpRoi of EVERY frame of the panorama (all ranks compute them: it is O(N) geometry);
    owner[i]: rank that holds frame i."""
    x0 = min(c[0] for c in corners)
    y0 = min(c[1] for c in corners)
    x1 = max(c[0] + s[0] for c, s in zip(corners, sizes))
    y1 = max(c[1] + s[1] for c, s in zip(corners, sizes))
    pano = (x0, y0, x1 - x0, y1 - y0)   # cv.detail.resultRoi
    nb = effective_bands(pano, num_bands)
    bbox, rois = [], []
    for r in range(world):
        mine = [i for i in range(len(corners)) if owner[i] == r]
        if not mine:
            raise ValueError(f"rank {r} holds no frame")
        bbox.append(rect_union([padded_rect(corners[i], sizes[i], pano, nb) for i in mine]))
        rois.append([(corners[i][0] - x0, corners[i][1] - y0, sizes[i][0], sizes[i][1]) for i in mine])
    plan = ExchangePlan(world, nb, pano, padded_pano_size(pano, nb), bbox, rois)
    for s in range(world):
        for d in range(world):
            if s == d:
                continue
            r = rect_intersect(bbox[s], bbox[d])
            if r is not None:
                plan.pairs.append((s, d, r))
    return plan


def owner_map(plan: ExchangePlan) -> np.ndarray:
    """int16 HxW (final pano size): the lowest rank one of whose frame rois covers the pixel, -1 where no frame does."""
    h, w = plan.pano_roi[3], plan.pano_roi[2]
    own = -np.ones((h, w), np.int16)
    for r in reversed(range(plan.world)):
        for (x, y, rw, rh) in plan.image_rois[r]:
            own[y:y + rh, x:x + rw] = r
    return own


def run_exchange(plan: ExchangePlan, rank: int, backend, dist, make_buffers) -> None:
    """The protocol.  ``backend.export(level, rect) -> (lap, w)`` and ``backend.import_(level, rect, lap, w)`` work on
    flat tensors created by ``make_buffers(level, rect) -> (lap, w)``; ``dist`` is torch.distributed."""
    ops, recv_slots, keep = [], [], []
    for d, rect in plan.sends(rank):
        for lvl in range(plan.nb + 1):
            lap, w = backend.export(lvl, rect)
            keep.append((lap, w))
            ops.append(dist.P2POp(dist.isend, lap, d))
            ops.append(dist.P2POp(dist.isend, w, d))
    for s, rect in plan.recvs(rank):
        for lvl in range(plan.nb + 1):
            lap, w = make_buffers(lvl, rect)
            recv_slots.append((lvl, rect, lap, w))
            ops.append(dist.P2POp(dist.irecv, lap, s))
            ops.append(dist.P2POp(dist.irecv, w, s))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for lvl, rect, lap, w in recv_slots:
        backend.import_(lvl, rect, lap, w)


class HipOverlapExchange:
    """bench.py's multi-GPU step: Composer.feed -> exchange over RCCL -> Composer.finish_region(bbox[rank])."""

    def __init__(self, composer, dist, torch, all_corners, all_sizes, owner, num_bands: int, float_mode: bool = False):
        from . import _lib

        self._lib = _lib
        self.c, self.dist, self.torch = composer, dist, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.plan = plan_exchange(all_corners, all_sizes, owner, self.world, num_bands)
        self.float_mode = float_mode
        composer.set_pano_roi(self.plan.pano_roi)
        self.blender = composer.blender_handle()
        self._bufs: Dict[Tuple[str, int, int, Rect], tuple] = {}

    def _buffers(self, kind: str, peer: int, lvl: int, rect: Rect):
        key = (kind, peer, lvl, rect)
        if key not in self._bufs:
            t = self.torch
            n = (rect[2] >> lvl) * (rect[3] >> lvl)
            lap = t.empty(n * 3, dtype=t.float32 if self.float_mode else t.int16, device="cuda")
            w = t.empty(n, dtype=t.float32, device="cuda")
            self._bufs[key] = (lap, w)
        return self._bufs[key]

    def run(self, frames) -> None:
        L, chk = self._lib.lib(), self._lib.check
        self.c.feed(frames)
        plan, rank, dist = self.plan, self.rank, self.dist
        ops, recv_slots = [], []
        for d, rect in plan.sends(rank):
            for lvl in range(plan.nb + 1):
                lap, w = self._buffers("s", d, lvl, rect)
                chk(L.ssp_blender_export_partial(self.blender, lvl, rect[0], rect[1], rect[2], rect[3], C.c_void_p(lap.data_ptr()), C.c_void_p(w.data_ptr())))
                ops.append(dist.P2POp(dist.isend, lap, d))
                ops.append(dist.P2POp(dist.isend, w, d))
        for s, rect in plan.recvs(rank):
            for lvl in range(plan.nb + 1):
                lap, w = self._buffers("r", s, lvl, rect)
                recv_slots.append((lvl, rect, lap, w))
                ops.append(dist.P2POp(dist.irecv, lap, s))
                ops.append(dist.P2POp(dist.irecv, w, s))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for lvl, rect, lap, w in recv_slots:
            chk(L.ssp_blender_import_partial(self.blender, lvl, rect[0], rect[1], rect[2], rect[3], C.c_void_p(lap.data_ptr()), C.c_void_p(w.data_ptr())))
        self.c.finish_region(plan.bbox[rank])


# ====================================================================================================================
# Strip exchange: send parts of the neighbours' planes instead of 13.3 bytes per pixel of partial pyramid sums
# ====================================================================================================================
# A collapsed pixel depends on the blended levels within 2 * 2^bands pixels, so a rank that OWNS a rectangle of the panorama blends the
# rectangle grown by that much (its REGION), exactly like a single GPU would (weights included: it visits the images in global order), from
# its own frames plus, for every foreign frame that reaches the region, a STRIP of that frame's planes:
#   levels=True  (default) the strip's rectangle of EVERY level of the owner's planes (8-bit image levels, float32 weight levels, aprons),
#                cut to the region grown by 2^bands -- one sample of the top level, what pyrUp reaches; the receiver builds nothing;
#   levels=False the bordered level-0 planes only (u8x3 image with its BORDER_REFLECT band, u8 mask; 4 bytes per pixel) over the region grown
#                by 4 * 2^bands -- a Laplacian / weight sample of level l depends on level 0 within 2 (2^(l+1) - 1) + 2^(l+1) < 4 * 2^bands
#                pixels --, the receiver rebuilding the Gaussian pyramids of the strips itself.

@dataclass
class StripPlan:
    world: int
    nb: int
    pano_roi: Rect
    padded: Tuple[int, int]
    owner: List[int]                      # per image
    prect: List[Rect]                     # per image: padded rectangle (pano-relative, multiples of 2^nb)
    owned: List[Rect]                     # per rank: rectangle it outputs (multiples of 2^nb)
    region: List[Rect]                    # per rank: rectangle it collapses (owned grown by the collapse halo)
    strips: List[Tuple[int, int, Rect]] = field(default_factory=list)   # (image, dst rank, rect): rect of image's planes dst needs
    cell_owner: np.ndarray = None         # int16 (padded_h / 2^nb, padded_w / 2^nb): owning rank of every cell, -1 = nobody
    levels: bool = True                   # strips carry every pyramid level (the receiver builds nothing) / level 0 only (it rebuilds the pyramids)

    def sends(self, rank: int):
        return [(i, d, r) for i, d, r in self.strips if self.owner[i] == rank]

    def recvs(self, rank: int):
        return [(i, self.owner[i], r) for i, d, r in self.strips if d == rank]

    def bytes_sent(self, rank: int, bytes_per_px: int = 4) -> int:
        """payload bytes of the strips a rank sends.  Level 0: 3 + 1 per pixel for 8-bit frames, 12 + 1 for float32 frames; all-level
        strips add the levels >= 1 at (bytes_per_px - 1) + 4 per sample (8-bit fed pyramids keep 8-bit levels; the weights are float32)."""
        px = sum(r[2] * r[3] for _, _, r in self.sends(rank))
        if not self.levels:
            return px * bytes_per_px
        return px * bytes_per_px + sum((r[2] >> l) * (r[3] >> l) * (bytes_per_px - 1 + 4) for _, _, r in self.sends(rank) for l in range(1, self.nb + 1))


def _grow(r: Rect, by: int, bound: Tuple[int, int]) -> Rect:
    x0, y0 = max(0, r[0] - by), max(0, r[1] - by)
    x1, y1 = min(bound[0], r[0] + r[2] + by), min(bound[1], r[1] + r[3] + by)
    return (x0, y0, x1 - x0, y1 - y0)


@dataclass
class FeedParts:
    """The feed units of a panorama (``feed_parts``): what the composers warp and feed, image by image."""
    corners: List[Tuple[int, int]]
    sizes: List[Tuple[int, int]]
    image: List[int]                      # per part: index of the frame it belongs to
    owner: List[int]                      # per part: rank that holds that frame
    pano_roi: Rect                        # cv.detail.resultRoi of the WHOLE rois (what the caller of the reference sees, sde.py:1807)
    image_rois: List[Rect]                # per frame: warper.warpRoi (sde.py:1696)


def feed_parts(cv, warp: str, warper_scale: float, frame_size: Tuple[int, int], Ks, Rs, owner: Sequence[int], num_bands: int) -> FeedParts:
    """What ``Composer`` feeds for these cameras: one rectangle per frame -- its warpRoi -- or two for a frame that straddles
    u = +-pi*scale, whose roi spans the full circle while its mask is set at the two ends only (``PyRotationWarper.liveParts``).  Every
    rank calls this for ALL frames of the panorama (geometry only) so that all ranks agree on the plan; ``num_bands`` is the REQUESTED band
    count, as handed to the composers."""
    w = cv.PyRotationWarper(warp, warper_scale)
    corners, sizes, image, part_owner, rois = [], [], [], [], []
    for i in range(len(Ks)):
        roi = tuple(int(v) for v in w.warpRoi(frame_size, Ks[i], Rs[i]))
        rois.append(roi)
        for part in w.liveParts(frame_size, Ks[i], Rs[i], num_bands):
            corners.append((part[0], part[1])); sizes.append((part[2], part[3])); image.append(i); part_owner.append(int(owner[i]))
    x0 = min(r[0] for r in rois)
    y0 = min(r[1] for r in rois)
    x1 = max(r[0] + r[2] for r in rois)
    y1 = max(r[1] + r[3] for r in rois)
    return FeedParts(corners, sizes, image, part_owner, (x0, y0, x1 - x0, y1 - y0), rois)


def _clusters(rects: Sequence[Rect]) -> List[List[int]]:
    """Connected components of rectangles that overlap or touch."""
    n = len(rects)
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for a in range(n):
        for b in range(a + 1, n):
            ra, rb = rects[a], rects[b]
            if ra[0] <= rb[0] + rb[2] and rb[0] <= ra[0] + ra[2] and ra[1] <= rb[1] + rb[3] and rb[1] <= ra[1] + ra[3]:
                parent[find(a)] = find(b)
    groups: Dict[int, List[int]] = {}
    for a in range(n):
        groups.setdefault(find(a), []).append(a)
    return list(groups.values())


def plan_strips(corners: Sequence[Tuple[int, int]], sizes: Sequence[Tuple[int, int]], owner: Sequence[int], world: int, num_bands: int,
                levels: bool = True, pano_roi: Rect = None) -> StripPlan:
    """corners / sizes / owner describe the FEED UNITS of the panorama: the frames' rois, or -- closed rings -- ``feed_parts``' rectangles
    (then ``pano_roi`` is OpenCV's resultRoi of the whole rois; default: the union of the units).
    ``levels``: all-level strips -- the sender ships the strip's rectangle of every pyramid level it has built anyway, so the rectangle is the
    receiver's region grown by 2^bands only (one pixel of the top level, pyrUp's reach; rectangles nest from level to level because everything is
    a multiple of 2^bands).  ``levels=False``: level-0 strips over the region grown by 4 * 2^bands, the receiver rebuilds their pyramids."""
    n = len(corners)
    x0 = min(c[0] for c in corners)
    y0 = min(c[1] for c in corners)
    x1 = max(c[0] + s[0] for c, s in zip(corners, sizes))
    y1 = max(c[1] + s[1] for c, s in zip(corners, sizes))
    pano = (x0, y0, x1 - x0, y1 - y0)
    if pano_roi is not None:
        pano_roi = tuple(int(v) for v in pano_roi)
        if not (pano_roi[0] <= x0 and pano_roi[1] <= y0 and pano_roi[0] + pano_roi[2] >= x1 and pano_roi[1] + pano_roi[3] >= y1):
            raise ValueError(f"plan_strips: pano_roi {pano_roi} does not contain every feed unit (union {pano})")
        pano = pano_roi
        x0, y0 = pano[0], pano[1]
    nb = effective_bands(pano, num_bands)
    m = 1 << nb
    padded = padded_pano_size(pano, nb)
    prect = [padded_rect(corners[i], sizes[i], pano, nb) for i in range(n)]
    rois = [(corners[i][0] - x0, corners[i][1] - y0, sizes[i][0], sizes[i][1]) for i in range(n)]
    # ownership on the 2^nb grid: a cell touched by frames of several ranks goes to the rank whose frames' bounding box it is
    # most central in (midlines between equally sized neighbours).  "Bounding box" = that of the rank's MAIN cluster of units: a rank that
    # holds a frame straddling u = +-pi*scale has a second, small unit at the other end of the panorama, which claims nothing while a
    # rank whose main cluster lies there touches the cell (a rank's owned cells then stay one compact rectangle).
    gw, gh = padded[0] // m, padded[1] // m
    best = np.full((gh, gw), np.inf)
    cell = -np.ones((gh, gw), np.int16)
    cx = (np.arange(gw) + 0.5) * m
    cy = (np.arange(gh) + 0.5) * m
    dists, any_touch = [], np.zeros((gh, gw), bool)
    for r in range(world):
        mine = [i for i in range(n) if owner[i] == r]
        if not mine:
            raise ValueError(f"rank {r} holds no frame")
        groups = _clusters([rois[i] for i in mine])
        main = max(groups, key=lambda g: sum(rois[mine[k]][2] * rois[mine[k]][3] for k in g))
        bb = rect_union([rois[mine[k]] for k in main])
        dist = np.abs(cx - (bb[0] + bb[2] / 2.0))[None, :] / (bb[2] / 2.0) + np.abs(cy - (bb[1] + bb[3] / 2.0))[:, None] / (bb[3] / 2.0)
        dists.append(dist)
        touched = np.zeros((gh, gw), bool)
        for k, i in enumerate(mine):
            rx, ry, rw, rh = rois[i]
            any_touch[ry // m:(ry + rh + m - 1) // m, rx // m:(rx + rw + m - 1) // m] = True
            if k in main:
                touched[ry // m:(ry + rh + m - 1) // m, rx // m:(rx + rw + m - 1) // m] = True
        take = touched & (dist < best)
        best[take] = dist[take]
        cell[take] = r
    orphan = any_touch & (cell < 0)       # touched by minor clusters only: to the rank whose main cluster is nearest
    if orphan.any():
        stack = np.stack(dists)
        cell[orphan] = np.argmin(stack[:, orphan], axis=0).astype(np.int16)
    owned, region = [], []
    for r in range(world):
        ys, xs = np.nonzero(cell == r)
        if len(ys) == 0:
            raise ValueError(f"rank {r} owns no part of the panorama")
        o = (int(xs.min()) * m, int(ys.min()) * m, (int(xs.max()) - int(xs.min()) + 1) * m, (int(ys.max()) - int(ys.min()) + 1) * m)
        owned.append(o)
        region.append(_grow(o, 2 * m, padded))
    plan = StripPlan(world, nb, pano, padded, list(owner), prect, owned, region, cell_owner=cell, levels=bool(levels))
    for d in range(world):
        need = _grow(region[d], m if levels else 4 * m, padded)
        for i in range(n):
            if owner[i] == d:
                continue
            s = rect_intersect(prect[i], need)
            if s is not None:
                plan.strips.append((i, d, s))
    return plan


def strip_owner_map(plan: StripPlan) -> np.ndarray:
    """int16 HxW (final pano size): rank that outputs each pixel (-1 where no frame lies)."""
    m = 1 << plan.nb
    full = np.repeat(np.repeat(plan.cell_owner, m, axis=0), m, axis=1)
    return full[:plan.pano_roi[3], :plan.pano_roi[2]]


class _DevBytes:
    """A tightly packed device byte buffer for one strip plane.  ``alloc(nbytes) -> (keepalive, device pointer)``."""

    def __init__(self, alloc):
        self._alloc = alloc
        self._cache: Dict[Tuple[str, int, int], tuple] = {}

    def get(self, kind: str, image: int, peer: int, nbytes: int):
        key = (kind, image, peer)
        if key not in self._cache:
            self._cache[key] = self._alloc(nbytes)
        return self._cache[key]


def _umat_alloc(nbytes: int):
    from .umat import UMat
    from . import _lib
    u = UMat(np.empty((1, nbytes), np.uint8))
    w, h, cn, depth = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    pitch, data = C.c_size_t(), C.c_void_p()
    _lib.check(_lib.lib().ssp_image_info(u._h, C.byref(w), C.byref(h), C.byref(cn), C.byref(depth), C.byref(pitch), C.byref(data)))
    return u, data.value


class StripExchangeBase:
    """Shared part of the strip protocol: Composer.feed -> export strips -> [transport] -> feed strips -> order -> finish_region."""

    def __init__(self, composer, plan: StripPlan, rank: int, alloc, plane_layout: bool = True):
        from . import _lib
        self._lib = _lib
        self.c, self.plan, self.rank = composer, plan, rank
        # plane layout: a strip travels with the row pitch and apron of a level-0 plane, the receive buffer is fed as the plane itself
        self.plane_layout = bool(plane_layout)
        _lib.check(_lib.lib().ssp_blender_set_strip_layout(composer.blender_handle(), int(self.plane_layout)))
        composer.set_pano_roi(plan.pano_roi)
        self.mine = [i for i in range(len(plan.owner)) if plan.owner[i] == rank]   # global indices of this rank's feed units, in feed order
        self.local = {g: k for k, g in enumerate(self.mine)}
        self.bufs = _DevBytes(alloc)
        self._check_parts()

    def _check_parts(self) -> None:
        """The plan's feed units of this rank must be what the composer feeds (frames that straddle u = +-pi*scale are fed as two units:
        build the plan from ``feed_parts``)."""
        parts = self.c.parts() if hasattr(self.c, "parts") else None
        if parts is None:
            return
        if len(parts) != len(self.mine):
            raise ValueError(f"strip exchange: rank {self.rank}'s composer feeds {len(parts)} units but the plan holds {len(self.mine)} for it "
                             "(plan closed rings with parallel.feed_parts)")
        m = 1 << self.plan.nb
        for k, g in enumerate(self.mine):
            want = padded_rect(parts[k][1][:2], parts[k][1][2:], self.plan.pano_roi, self.plan.nb)
            if tuple(want) != tuple(self.plan.prect[g]):
                raise ValueError(f"strip exchange: unit {g} of the plan {self.plan.prect[g]} is not what rank {self.rank}'s composer feeds as unit {k} {want}")

    def level_buffer_bytes(self, rect) -> int:
        """bytes of one all-level strip buffer (every level's image and weight planes, aprons included)."""
        v = C.c_size_t()
        self._lib.check(self._lib.lib().ssp_level_strip_buffer_bytes(int(rect[2]), int(rect[3]), int(self.plan.nb), int(bool(getattr(self.c, "float_frames", False))), C.byref(v)))
        return int(v.value)

    def buffer_bytes(self, rect, cn: int) -> int:
        """bytes of one strip buffer: cn = 3 the image strip (8-bit, or float32 when the composer works on float frames), 1 its mask."""
        bpp = cn * (4 if (cn == 3 and getattr(self.c, "float_frames", False)) else 1)
        v = C.c_size_t()
        self._lib.check(self._lib.lib().ssp_strip_buffer_bytes(int(rect[2]), int(rect[3]), bpp, int(self.plane_layout), C.byref(v)))
        return int(v.value)

    def _static(self):
        """The plan is static: buffers and ctypes argument arrays are built once."""
        if getattr(self, "_st", None) is None:
            nb = self.buffer_bytes
            if self.plan.levels:
                # one buffer per strip; the mask slot repeats it
                lb = self.level_buffer_bytes
                out = [(i, d, r, self.bufs.get("sl", i, d, lb(r)), None) for i, d, r in self.plan.sends(self.rank)]
                slots = [(i, s, r, self.bufs.get("rl", i, s, lb(r)), None) for i, s, r in self.plan.recvs(self.rank)]
                out = [(i, d, r, b, b) for i, d, r, b, _ in out]
                slots = [(i, s, r, b, b) for i, s, r, b, _ in slots]
            else:
                out = [(i, d, r, self.bufs.get("si", i, d, nb(r, 3)), self.bufs.get("sm", i, d, nb(r, 1))) for i, d, r in self.plan.sends(self.rank)]
                slots = [(i, s, r, self.bufs.get("ri", i, s, nb(r, 3)), self.bufs.get("rm", i, s, nb(r, 1))) for i, s, r in self.plan.recvs(self.rank)]

            def arrays(items):
                n = len(items)
                return (n, (C.c_int * max(4 * n, 1))(*[int(v) for _, _, r, _, _ in items for v in r]), (C.c_void_p * max(n, 1))(*[ib[1] for _, _, _, ib, _ in items]),
                        (C.c_void_p * max(n, 1))(*[mb[1] for _, _, _, _, mb in items]))
            keys = list(self.mine) + [i for i, _, _, _, _ in slots]
            self._st = dict(out=out, slots=slots, exp=arrays(out), imp=arrays(slots), origins=(C.c_int * max(len(slots), 1))(*[int(self.plan.prect[i][0]) for i, _, _, _, _ in slots]), feeds=(C.c_int * max(len(out), 1))(*[self.local[i] for i, _, _, _, _ in out]),
                            keys=(C.c_int * len(keys))(*keys), nkeys=len(keys))
        return self._st

    def traffic(self) -> dict:
        """Bytes of strip buffers this rank sends to / receives from every neighbour per panorama (the buffers as they travel: all levels with
        their aprons, or level-0 image + mask), and how many strips."""
        def size(r):
            return self.level_buffer_bytes(r) if self.plan.levels else self.buffer_bytes(r, 3) + self.buffer_bytes(r, 1)
        sent, recv = {}, {}
        for _, d, r in self.plan.sends(self.rank):
            sent[int(d)] = sent.get(int(d), 0) + size(r)
        for _, s, r in self.plan.recvs(self.rank):
            recv[int(s)] = recv.get(int(s), 0) + size(r)
        return {"sent_bytes": sent, "recv_bytes": recv, "strips_sent": len(self.plan.sends(self.rank)), "strips_received": len(self.plan.recvs(self.rank)),
                "protocol": "all-level strips" if self.plan.levels else "level-0 strips, pyramids rebuilt by the receiver"}

    def export_all(self):
        """-> [(image, dst, rect, (img_keep, img_ptr), (mask_keep, mask_ptr))] for every strip this rank sends (one batched copy launch)."""
        st = self._static()
        n, rects, imgs, masks = st["exp"]
        if n and self.plan.levels:
            self._lib.check(self._lib.lib().ssp_blender_export_level_strips(self.c.blender_handle(), n, st["feeds"], rects, imgs, int(self.plan.nb)))
        elif n:
            self._lib.check(self._lib.lib().ssp_blender_export_strips(self.c.blender_handle(), n, st["feeds"], rects, imgs, masks))
        return st["out"]

    def tensors(self, item):
        """the device buffers of one export_all / recv_slots entry that travel: (image, mask), or the one all-level buffer"""
        return (item[3][0],) if self.plan.levels else (item[3][0], item[4][0])

    def recv_slots(self):
        return self._static()["slots"]

    def finish(self, slots) -> None:
        L, chk = self._lib.lib(), self._lib.check
        st = self._static()
        blender = self.c.blender_handle()
        n, rects, imgs, masks = st["imp"]
        if n and self.plan.levels:
            chk(L.ssp_blender_feed_level_strips(blender, n, rects, st["origins"], imgs, int(self.plan.nb)))
        elif n:
            chk(L.ssp_blender_feed_strips(blender, n, rects, imgs, masks))
        chk(L.ssp_blender_order_feeds(blender, st["keys"], st["nkeys"]))
        self.c.finish_region(self.plan.region[self.rank])

    # the same in two halves, for a caller that builds the strips' pyramids together with another blender's pending images
    def import_strips(self) -> None:
        """received strips -> level-0 planes of this panorama's blender; their pyramids stay pending."""
        n, rects, imgs, masks = self._static()["imp"]
        if n and self.plan.levels:
            self._lib.check(self._lib.lib().ssp_blender_feed_level_strips(self.c.blender_handle(), n, rects, self._static()["origins"], imgs, int(self.plan.nb)))      # (nothing left pending)
        elif n:
            self._lib.check(self._lib.lib().ssp_blender_feed_strips_begin(self.c.blender_handle(), n, rects, imgs, masks))

    def collapse(self) -> None:
        st = self._static()
        self._lib.check(self._lib.lib().ssp_blender_order_feeds(self.c.blender_handle(), st["keys"], st["nkeys"]))
        self.c.finish_region(self.plan.region[self.rank])


def emulate_strip_exchange(exchanges: Sequence[StripExchangeBase], frames_per_rank) -> None:
    """All ranks of a StripPlan on ONE GPU (tests): device-to-device copies stand in for RCCL."""
    from . import _lib
    import ctypes
    hip = _lib.lib()
    levels = exchanges[0].plan.levels
    for ex, frames in zip(exchanges, frames_per_rank):
        ex.c.feed_planes(frames)
        if levels:
            ex.c.feed_pyramids()          # all-level strips are cut from the finished pyramids
    sent = {}
    for ex in exchanges:
        for i, d, r, ib, mb in ex.export_all():
            sent[(i, d)] = (ib, mb)
    for ex in exchanges:
        if not levels:
            ex.c.feed_pyramids()
    for ex in exchanges:
        slots = ex.recv_slots()
        for i, s, r, ib, mb in slots:
            src_i, src_m = sent[(i, ex.rank)]
            if levels:
                _lib.check(hip.ssp_device_copy(ctypes.c_void_p(ib[1]), ctypes.c_void_p(src_i[1]), ctypes.c_size_t(ex.level_buffer_bytes(r))))
                continue
            _lib.check(hip.ssp_device_copy(ctypes.c_void_p(ib[1]), ctypes.c_void_p(src_i[1]), ctypes.c_size_t(ex.buffer_bytes(r, 3))))
            _lib.check(hip.ssp_device_copy(ctypes.c_void_p(mb[1]), ctypes.c_void_p(src_m[1]), ctypes.c_size_t(ex.buffer_bytes(r, 1))))
        ex.finish(slots)


def strip_transport_begin(dist, sends, recvs):
    """Post one batched point-to-point exchange and return its requests.  ``sends`` = [(dst, tensors)], ``recvs`` =
    [(src, tensors)], both in the order of ``StripPlan.strips`` -- for every ordered pair of ranks the messages are posted in
    the same order on both sides."""
    ops = []
    for d, tensors in sends:
        for t in tensors:
            ops.append(dist.P2POp(dist.isend, t, d))
    for s, tensors in recvs:
        for t in tensors:
            ops.append(dist.P2POp(dist.irecv, t, s))
    return dist.batch_isend_irecv(ops) if ops else []


def strip_transport(dist, sends, recvs) -> None:
    for req in strip_transport_begin(dist, sends, recvs):
        req.wait()


class HipStripExchange(StripExchangeBase):
    """bench.py's multi-GPU step over RCCL (torch.distributed ``nccl``): point-to-point sends of the strips each neighbour needs."""

    def __init__(self, composer, dist, torch, all_corners, all_sizes, owner, num_bands: int, levels: bool = True, pano_roi: Rect = None):
        """all_corners / all_sizes / owner: the feed units of the WHOLE panorama (``feed_parts``: corners, sizes, owner and pano_roi)."""
        self.dist, self.torch = dist, torch
        self._tensors = []

        def alloc(nbytes: int):
            t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            return t, t.data_ptr()
        plan = plan_strips(all_corners, all_sizes, owner, dist.get_world_size(), num_bands, levels=levels, pano_roi=pano_roi)
        super().__init__(composer, plan, dist.get_rank(), alloc)

    def _check_stream(self) -> None:
        """RCCL orders its transfers against torch's CURRENT stream; the library's kernels must be queued on that same stream
        (``ssp_set_stream(torch.cuda.current_stream().cuda_stream)``), or a send could read a strip that is still being packed."""
        cur = C.c_void_p()
        self._lib.check(self._lib.lib().ssp_current_stream(C.byref(cur)))
        want = int(self.torch.cuda.current_stream().cuda_stream)
        if int(cur.value or 0) != want:
            raise RuntimeError(f"HipStripExchange: the library launches on stream {int(cur.value or 0):#x} but torch's current stream is {want:#x}; "
                               "call ssp_set_stream / ssp_use_stream with torch's stream before exchanging strips over RCCL")

    def begin(self, frames) -> None:
        """warp + level-0 borders of the own frames, export the strips and post the point-to-point messages."""
        if self.dist.get_backend() == "nccl":
            self._check_stream()
        # The receive buffers ARE planes of the previous panorama that went through this exchange: its (pyramid and) collapse
        # kernels read them.  Posting the next receives waits, on the device, for the event recorded behind that collapse -- stated
        # here instead of relying on where the process group happens to pick up the current stream.
        done = getattr(self, "_buffers_free", None)
        if done is not None:
            self.torch.cuda.current_stream().wait_event(done)
        self.c.feed_planes(frames)
        if self.plan.levels:
            self.c.feed_pyramids()        # all-level strips are cut from the finished pyramids
        out = self.export_all()
        slots = self.recv_slots()
        if getattr(self, "_msgs", None) is None:
            self._msgs = ([(it[1], self.tensors(it)) for it in out], [(it[1], self.tensors(it)) for it in slots])
            # RCCL takes device tensors and orders against the current stream by itself.  Any other backend (gloo rehearsals of
            # the N>1 path on one GPU) gets host copies: a device pointer is not something its transport promises to read.
            self._staged = self.dist.get_backend() != "nccl"
            if self._staged:
                self._host = tuple([(p, tuple(self.torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in ts)) for p, ts in side] for side in self._msgs)
        if self._staged:
            for (_, dev), (_, host) in zip(self._msgs[0], self._host[0]):
                for d, h in zip(dev, host):
                    h.copy_(d, non_blocking=True)
            self.torch.cuda.current_stream().synchronize()
            self._reqs = strip_transport_begin(self.dist, *self._host)
        else:
            self._reqs = strip_transport_begin(self.dist, *self._msgs)

    def own_pyramids(self) -> None:
        if not self.plan.levels:
            self.c.feed_pyramids()

    def _arrived(self) -> None:
        """Wait for this panorama's strips.  Timed both ways (``wait_stats``): on the host (gloo blocks here) and on the stream (RCCL makes the
        compute stream wait for its transfers: an event pair around the wait -- ~0 when the strips arrived while the GPU was busy)."""
        import time as _time
        timed = getattr(self, "_time_waits", False)
        if timed:
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record(self.torch.cuda.current_stream())
            t0 = _time.perf_counter()
        for req in self._reqs:
            req.wait()
        if timed:
            e1.record(self.torch.cuda.current_stream())
            self._waits.append((e0, e1, (_time.perf_counter() - t0) * 1e3))
        self._reqs = None
        if self._staged:
            for (_, dev), (_, host) in zip(self._msgs[1], self._host[1]):
                for d, h in zip(dev, host):
                    d.copy_(h, non_blocking=True)

    def time_waits(self, on: bool = True) -> None:
        self._time_waits, self._waits = bool(on), []

    def wait_stats(self) -> dict:
        """Mean time per panorama the collapse waited for its receives since ``time_waits()``: on the stream (event pair) and on the host."""
        waits = getattr(self, "_waits", [])
        if not waits:
            return {"panoramas": 0, "recv_wait_stream_ms": None, "recv_wait_host_ms": None}
        self.torch.cuda.synchronize()
        return {"panoramas": len(waits), "recv_wait_stream_ms": round(sum(a.elapsed_time(b) for a, b, _ in waits) / len(waits), 4),
                "recv_wait_host_ms": round(sum(h for _, _, h in waits) / len(waits), 4)}

    def _mark_buffers_free(self) -> None:
        if getattr(self, "_buffers_free", None) is None:
            self._buffers_free = self.torch.cuda.Event()
        self._buffers_free.record(self.torch.cuda.current_stream())

    def collapse(self) -> None:
        super().collapse()
        self._mark_buffers_free()

    def complete(self) -> None:
        """wait for the strips, build their pyramids, collapse the region."""
        self._arrived()
        self.finish(self.recv_slots())
        self._mark_buffers_free()

    def in_flight(self) -> bool:
        return getattr(self, "_reqs", None) is not None

    def run(self, frames) -> None:
        """One panorama, start to end: the transfer overlaps only the own pyramids."""
        self.begin(frames)
        self.own_pyramids()                                          # while the strips travel over xGMI
        self.complete()


class HipStripPipeline:
    """The multi-GPU step, software pipelined over TWO panoramas on ONE stream (double buffering: two composers, two sets of strip
    buffers).  A step (a) warps panorama k+1's frames, exports its strips and posts their messages, (b) finishes panorama k -- whose
    strips were posted a whole step ago -- and (c) builds panorama k+1's own pyramids; the pyramids of (b)'s strips and of (c) are
    one chain of launches, then (b)'s collapse.  Every step
    launches exactly one panorama's kernels, one after the other as in the serial order (nothing runs concurrently but the copy
    engines / RCCL), completes one panorama and leaves one in flight; a panorama's transfer has a full step to finish in."""

    def __init__(self, make_composer, dist, torch, all_corners, all_sizes, owner, num_bands: int, levels: bool = True, pano_roi: Rect = None):
        self.ex = [HipStripExchange(make_composer(), dist, torch, all_corners, all_sizes, owner, num_bands, levels=levels, pano_roi=pano_roi) for _ in range(2)]
        self.k = 0
        self.plan = self.ex[0].plan

    def step(self, frames) -> "HipStripExchange":
        cur, prev = self.ex[self.k & 1], self.ex[(self.k & 1) ^ 1]
        L, chk = cur._lib.lib(), cur._lib.check
        cur.begin(frames)
        finishing = prev.in_flight()
        if finishing:
            prev._arrived()
            prev.import_strips()
        if self.plan.levels:
            # all-level strips: begin() built panorama k+1's pyramids (its strips are cut from them), panorama k's strips arrive with theirs
            if finishing:
                prev.collapse()
            self.k += 1
            return prev
        # ONE chain of pyramid launches for the strips of panorama k and the own frames of panorama k+1 (two blenders): the
        # latency-bound small levels are walked once per step instead of twice
        chk(L.ssp_blender_feed_end_pair(cur.c.blender_handle(), prev.c.blender_handle() if finishing else None))
        if finishing:
            prev.collapse()
        self.k += 1
        return prev                                                  # the exchange whose panorama is complete now (after the first step)

    def drain(self) -> None:
        for e in (self.ex[self.k & 1], self.ex[(self.k & 1) ^ 1]):   # older first
            if e.in_flight():
                e.complete()


# ---- exposure compensation across ranks ------------------------------------------------------------------------
# The gains couple ALL images (one linear system), but their inputs are the seam-scale warps: ~0.1 MPix per frame.  Every rank
# warps its own frames at seam scale, the small images are gathered to every rank (all_gather_object: KBs to a few MB), every rank
# runs the SAME feed -- same inputs, same code, same gains -- and keeps the gains of its own frames.  No bulk collective.
def subset_compensator(cv, full, indices: Sequence[int]):
    """A compensator holding the gains of ``indices`` (in that order) of an already fed one: what a rank's Composer needs, whose
    frame i is global image indices[i].  Uses cv2's getMatGains / setMatGains pair."""
    if full is None or full.type == 0:
        return cv.detail.ExposureCompensator_createDefault(0)
    gains = full.getMatGains()
    local = cv.detail.ExposureCompensator_createDefault(full.type)
    local.setMatGains([gains[i] for i in indices])
    return local


def distributed_compensator(cv, dist, comp_type: int, owner: Sequence[int], corners_local, images_local, masks_local, configure=None):
    """Every rank passes the seam-scale warps of ITS frames (ndarrays or UMats, in the order of its global indices); returns
    (compensator fed with all images in global order, compensator with this rank's gains in local order)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = [i for i in range(len(owner)) if owner[i] == rank]
    if len(mine) != len(images_local):
        raise ValueError(f"rank {rank} owns {len(mine)} frames but passed {len(images_local)} seam-scale images")
    host = lambda a: a.get() if hasattr(a, "get") else np.asarray(a)  # noqa: E731
    payload = [(g, tuple(int(v) for v in c), host(im), host(mk)) for g, c, im, mk in zip(mine, corners_local, images_local, masks_local)]
    gathered = [None] * world
    dist.all_gather_object(gathered, payload)
    items = sorted((it for part in gathered for it in part), key=lambda it: it[0])
    if [it[0] for it in items] != list(range(len(owner))):
        raise ValueError("distributed_compensator: the ranks' frames do not cover the global index range")
    full = cv.detail.ExposureCompensator_createDefault(comp_type)
    if configure is not None:
        configure(full)                       # e.g. setBlockSize / setNrFeeds, identically on every rank
    full.feed(corners=[it[1] for it in items], images=[it[2] for it in items], masks=[it[3] for it in items])
    return full, subset_compensator(cv, full, mine)
