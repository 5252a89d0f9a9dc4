"""Synthetic star-field frames and camera rigs for the BASELINE.json configurations.

The reference stitches photographs of the night sky; its registration half (out of scope) produces the cameras.
For measurement and parity tests the hot path is fed instead with seeded synthetic inputs:

* one global star catalogue per configuration (directions uniform on the sphere, Pareto(1.5) fluxes so that about
  1 % of the stars saturate, Gaussian PSF sigma in U(1.0, 2.2) px, slight colour tint), projected into every camera,
  so overlapping frames really show the same sky;
* per frame: sky background N(12, 3^2) per channel with a vertical airglow gradient (+8 at the bottom), an exposure
  factor (config 3) and independent noise, ``rng = default_rng(1000 * config_id + frame_idx)``;
* pinhole cameras ``K = [[f,0,W/2],[0,f,H/2],[0,0,1]]``, ``f = (W/2)/tan(HFOV/2)``, ``R = R_y(yaw) @ R_x(pitch)``
  as float32, warper scale = f (compose_work_aspect = 1).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np


@dataclass
class Rig:
    name: str
    config_id: int
    width: int
    height: int
    hfov_deg: float
    yaws_deg: List[float]
    pitches_deg: List[float]
    warp: str
    blend: str                    # "multiband" | "feather" | "no"
    num_bands: int = 5
    blend_strength: float = 5.0   # feather: sharpness = 1/blend_width (sde.py:1808, :1819)
    expos_comp: int = 0           # cv.detail.ExposureCompensator_* code
    seam_megapix: float = 0.1
    dtype: str = "u8"             # "u8" | "f32"
    exposure_spread: Tuple[float, float] = (1.0, 1.0)
    Ks: List[np.ndarray] = field(default_factory=list)
    Rs: List[np.ndarray] = field(default_factory=list)

    @property
    def n(self) -> int:
        return len(self.yaws_deg)

    @property
    def focal(self) -> float:
        return (self.width / 2.0) / math.tan(math.radians(self.hfov_deg) / 2.0)

    @property
    def seam_scale(self) -> float:
        return min(1.0, math.sqrt(self.seam_megapix * 1e6 / (self.width * self.height)))

    @property
    def seam_size(self) -> Tuple[int, int]:
        s = self.seam_scale
        return (int(round(self.width * s)), int(round(self.height * s)))


def rot_y(a: float) -> np.ndarray:
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def rot_x(a: float) -> np.ndarray:
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def _finish(rig: Rig) -> Rig:
    f = rig.focal
    for yaw, pitch in zip(rig.yaws_deg, rig.pitches_deg):
        k = np.array([[f, 0, rig.width / 2.0], [0, f, rig.height / 2.0], [0, 0, 1]], dtype=np.float64).astype(np.float32)
        r = (rot_y(math.radians(yaw)) @ rot_x(math.radians(pitch))).astype(np.float32)
        rig.Ks.append(k)
        rig.Rs.append(r)
    return rig


def _ring(n: int, step: float) -> List[float]:
    return [(i - (n - 1) / 2.0) * step for i in range(n)]


def make_rig(config: int, scale_div: int = 1, n_override: Optional[int] = None, arc_step: Optional[float] = None) -> Rig:
    """BASELINE.json configs 1-5.  ``scale_div`` shrinks the frames (parity tests run the same rigs at 1/8 size); ``arc_step``: configs 3 / 4
    with another yaw step than SURVEY's 30 degrees (27: an open 357 degree arc)."""
    if config == 1:
        rig = Rig("cfg1: 3x1080p cylindrical + feather", 1, 1920 // scale_div, 1080 // scale_div, 60.0, [-40.0, 0.0, 40.0], [0.0] * 3,
                  "cylindrical", "feather", blend_strength=5.0)
    elif config == 2:
        n = n_override or 6
        rig = Rig("cfg2: 6x4K spherical + multiband(5)", 2, 3840 // scale_div, 2160 // scale_div, 60.0, _ring(n, 45.0), [0.0] * n, "spherical", "multiband", 5)
    elif config == 3:
        # the CLOSED ring, 12 frames at 30 degree steps.  The two frames at +-165 degrees straddle u = +-pi*scale: OpenCV's
        # by-border roi spans the full circle for them (the composer feeds their two live ends).  `arc_step` 27 gives the
        # 357 degree arc the rounds 1-3 were measured on (no straddling frame).
        n = n_override or 12
        rig = Rig("cfg3: 12x4K closed ring (30 deg yaw steps), spherical + gain blocks + multiband(5)", 3, 3840 // scale_div, 2160 // scale_div, 60.0, _ring(n, arc_step or 30.0),
                  [0.0] * n, "spherical", "multiband", 5, expos_comp=2, exposure_spread=(0.8, 1.25))
        if arc_step:
            rig.name = f"cfg3 as an open arc ({arc_step:g} deg yaw steps): 12x4K spherical + gain blocks + multiband(5)"
    elif config == 4:
        per_row = n_override or 12
        yaws, pitches = [], []
        for p in (-30.0, -10.0, 10.0, 30.0):
            yaws += _ring(per_row, arc_step or 30.0)
            pitches += [p] * per_row
        rig = Rig("cfg4: 48x4K (4 rows x 12) spherical + multiband(5)", 4, 3840 // scale_div, 2160 // scale_div, 60.0, yaws, pitches, "spherical", "multiband", 5)
    elif config == 5:
        per_row = n_override or 24
        yaws, pitches = [], []
        for p in (-24.0, -8.0, 8.0, 24.0):
            yaws += _ring(per_row, 14.5)
            pitches += [p] * per_row
        rig = Rig("cfg5: 96x8K f32 (4 rows x 24) spherical + multiband(7, float)", 5, 7680 // scale_div, 4320 // scale_div, 30.0, yaws, pitches, "spherical",
                  "multiband", 7, dtype="f32")
    else:
        raise ValueError(f"unknown config {config}")
    return _finish(rig)


def shard_rig(rig: Rig, rank: int, world: int) -> Rig:
    """Contiguous run of images for one GPU: rank r gets images [r*n/world, (r+1)*n/world)."""
    n = rig.n
    lo, hi = rank * n // world, (rank + 1) * n // world
    sub = Rig(rig.name + f" [rank {rank}/{world}]", rig.config_id, rig.width, rig.height, rig.hfov_deg, rig.yaws_deg[lo:hi], rig.pitches_deg[lo:hi], rig.warp,
              rig.blend, rig.num_bands, rig.blend_strength, rig.expos_comp, rig.seam_megapix, rig.dtype, rig.exposure_spread)
    sub.Ks = rig.Ks[lo:hi]
    sub.Rs = rig.Rs[lo:hi]
    return sub


class StarCatalogue:
    def __init__(self, config_id: int, density_per_sr: float):
        rng = np.random.default_rng(1000 * config_id)
        n = max(16, int(density_per_sr * 4 * math.pi))
        z = rng.uniform(-1, 1, n)
        phi = rng.uniform(0, 2 * math.pi, n)
        r = np.sqrt(1 - z * z)
        self.dirs = np.stack([r * np.cos(phi), z, r * np.sin(phi)], axis=1)
        flux = (rng.pareto(1.5, n) + 1.0)
        # scale so that ~1 % of the stars reach saturation at the PSF peak
        q99 = np.quantile(flux, 0.99)
        self.flux = flux / q99 * 255.0 * 2 * math.pi * 1.6 ** 2
        self.sigma = rng.uniform(1.0, 2.2, n)
        self.tint = np.clip(rng.normal(1.0, 0.08, (n, 3)), 0.75, 1.25)


def render_frame(rig: Rig, cat: StarCatalogue, idx: int, res_scale: float = 1.0) -> np.ndarray:
    """One frame as float32 HxWx3 (BGR), before quantisation."""
    w = int(round(rig.width * res_scale))
    h = int(round(rig.height * res_scale))
    rng = np.random.default_rng(1000 * rig.config_id + idx + (0 if res_scale == 1.0 else 500))
    lo, hi = rig.exposure_spread
    exposure = float(np.random.default_rng(7000 * rig.config_id + idx).uniform(lo, hi)) if hi > lo else 1.0
    img = rng.normal(12.0, 3.0, (h, w, 3)).astype(np.float32)
    img += (8.0 * np.arange(h, dtype=np.float32) / max(h - 1, 1))[:, None, None]
    K = rig.Ks[idx].astype(np.float64)
    R = rig.Rs[idx].astype(np.float64)
    cam = cat.dirs @ R  # R^T d for every star (rows)
    front = cam[:, 2] > 1e-6
    px = (K[0, 0] * cam[:, 0] / np.where(front, cam[:, 2], 1) + K[0, 2]) * res_scale
    py = (K[1, 1] * cam[:, 1] / np.where(front, cam[:, 2], 1) + K[1, 2]) * res_scale
    vis = front & (px > -8) & (px < w + 8) & (py > -8) & (py < h + 8)
    for s in np.nonzero(vis)[0]:
        sig = max(0.6, cat.sigma[s] * res_scale)
        rad = int(math.ceil(4 * sig))
        cx, cy = px[s], py[s]
        x0, x1 = max(0, int(cx) - rad), min(w, int(cx) + rad + 1)
        y0, y1 = max(0, int(cy) - rad), min(h, int(cy) + rad + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        gx = np.exp(-0.5 * ((np.arange(x0, x1) - cx) / sig) ** 2)
        gy = np.exp(-0.5 * ((np.arange(y0, y1) - cy) / sig) ** 2)
        amp = cat.flux[s] * res_scale ** 2 / (2 * math.pi * sig * sig)
        img[y0:y1, x0:x1, :] += (amp * gy[:, None] * gx[None, :])[:, :, None].astype(np.float32) * cat.tint[s][None, None, :].astype(np.float32)
    img *= np.float32(exposure)
    return np.clip(img, 0.0, 255.0)


def make_frames(rig: Rig, indices: Optional[List[int]] = None, want_seam: bool = False):
    """Full-resolution frames (uint8 or float32 BGR) and optionally the seam-scale frames (uint8)."""
    cat = StarCatalogue(rig.config_id, density_per_sr=2500.0 / _frame_solid_angle(rig))
    indices = list(range(rig.n)) if indices is None else indices
    frames, seams = [], []
    for i in indices:
        f = render_frame(rig, cat, i)
        frames.append(f.astype(np.float32) if rig.dtype == "f32" else np.rint(f).astype(np.uint8))
        if want_seam:
            sw, sh = rig.seam_size
            s = render_frame(rig, cat, i, res_scale=rig.seam_scale)
            seams.append(np.rint(s[:sh, :sw]).astype(np.uint8) if s.shape[0] >= sh and s.shape[1] >= sw else np.rint(np.pad(s, ((0, max(0, sh - s.shape[0])), (0, max(0, sw - s.shape[1])), (0, 0)), mode="edge")[:sh, :sw]).astype(np.uint8))
    return (frames, seams) if want_seam else frames


def _frame_solid_angle(rig: Rig) -> float:
    hf = math.radians(rig.hfov_deg)
    vf = 2 * math.atan((rig.height / 2.0) / rig.focal)
    return 4 * math.asin(math.sin(hf / 2) * math.sin(vf / 2))
