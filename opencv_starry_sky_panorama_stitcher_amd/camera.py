"""Caller-side camera preparation for the compose step (host logic, O(N) per panorama).

Mirrors what the reference does in Python/numpy before it reaches the warper
(stitching_detailed_enhanced.py):

* ``CameraParams``            cv2_pickleable/detail.py:62-110 (fields R, aspect, focal, ppx, ppy, t; ``K()``)
* ``median_focal``            sde.py:1373-1381
* ``wave_correct``            sde.py:1405-1411 -> cv.detail.waveCorrect (OpenCV motion_estimators.cpp)
* ``mirror_rotate``           sde.py:1413-1535
* ``scales``                  sde.py:750-752, :775-781, :1677-1681
* ``load_camera_params_json`` the ``*.CameraParams.json`` format written at sde.py:1122-1156
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

WAVE_CORRECT_HORIZ = 0
WAVE_CORRECT_VERT = 1
WAVE_CORRECT_AUTO = 2


@dataclass
class CameraParams:
    """cv.detail.CameraParams stand-in (same attribute names and ``K()``)."""

    focal: float = 1.0
    aspect: float = 1.0
    ppx: float = 0.0
    ppy: float = 0.0
    R: np.ndarray = field(default_factory=lambda: np.eye(3, dtype=np.float32))
    t: np.ndarray = field(default_factory=lambda: np.zeros((3, 1), dtype=np.float64))

    def K(self) -> np.ndarray:
        k = np.eye(3, dtype=np.float64)
        k[0, 0] = self.focal
        k[0, 2] = self.ppx
        k[1, 1] = self.focal * self.aspect
        k[1, 2] = self.ppy
        return k

    def clone(self) -> "CameraParams":
        return CameraParams(self.focal, self.aspect, self.ppx, self.ppy, np.copy(self.R), np.copy(self.t))


def load_camera_params_json(path: str) -> List[CameraParams]:
    """Read a ``*.CameraParams.json`` dump (sde.py:1122-1156)."""
    with open(path) as fh:
        doc = json.load(fh)
    cams = doc[doc.index("list_of_camera_params_for_disk_output:") + 1]
    return cameras_from_dicts(cams)


def cameras_from_dicts(cams: Sequence[dict]) -> List[CameraParams]:
    out = []
    for c in cams:
        out.append(
            CameraParams(
                focal=float(c["focal"]),
                aspect=float(c["aspect"]),
                ppx=float(c["ppx"]),
                ppy=float(c["ppy"]),
                R=np.asarray(c["R"], dtype=np.float32).reshape(3, 3),
                t=np.asarray(c.get("t", [[0.0], [0.0], [0.0]]), dtype=np.float64).reshape(3, 1),
            )
        )
    return out


def median_focal(cameras: Sequence[CameraParams]) -> float:
    """warped_image_scale (sde.py:1373-1381)."""
    focals = sorted(c.focal for c in cameras)
    n = len(focals)
    if n % 2 == 1:
        return focals[n // 2]
    return (focals[n // 2] + focals[n // 2 - 1]) / 2


def auto_detect_wave_correct_kind(rmats: Sequence[np.ndarray]) -> int:
    xs = [np.float32(r[0, 2]) / np.float32(r[2, 2]) for r in rmats]
    ys = [np.float32(r[1, 2]) / np.float32(r[2, 2]) for r in rmats]
    dx = float(max(xs)) - float(min(xs))
    dy = float(max(ys)) - float(min(ys))
    return WAVE_CORRECT_VERT if dy > dx else WAVE_CORRECT_HORIZ


def _hypot32(a, b):
    """cv::hypot(float, float): scaled form, every operation rounded to binary32."""
    f32 = np.float32
    a, b = abs(f32(a)), abs(f32(b))
    if a > b:
        b = f32(b / a)
        return f32(a * np.sqrt(f32(f32(1) + f32(b * b))))
    if b > 0:
        a = f32(a / b)
        return f32(b * np.sqrt(f32(f32(1) + f32(a * a))))
    return f32(0)


def eigen_symmetric_f32(mat):
    """cv::eigen on a CV_32F symmetric matrix = JacobiImpl_<float> (OpenCV core/src/lapack.cpp): cyclic-by-largest-pivot Jacobi
    rotations in binary32, eigenvalues descending, eigenvectors as ROWS.  Restated operation by operation: waveCorrect's result
    depends on the float32 round-off of this solver at the 1e-5 level (a float64 eigh gives rotations 1.5e-5 away, i.e. 0.01 px at
    the reference's focal lengths -- enough to flip the 1/32-px remap quantisation of a quarter of all samples;
    tests/test_real_images.py pins this against the reference's recorded lossless warps)."""
    f32 = np.float32
    A = np.array(mat, dtype=f32)
    n = A.shape[0]
    eps = np.finfo(f32).eps
    V = np.eye(n, dtype=f32)
    W = np.array([A[k, k] for k in range(n)], dtype=f32)
    ind_r, ind_c = [0] * n, [0] * n

    def row_max(k):      # column index of the largest |A[k, k+1:]|
        m, mv = k + 1, abs(A[k, k + 1])
        for i in range(k + 2, n):
            if mv < abs(A[k, i]):
                mv, m = abs(A[k, i]), i
        return m

    def col_max(k):      # row index of the largest |A[:k, k]|
        m, mv = 0, abs(A[0, k])
        for i in range(1, k):
            if mv < abs(A[i, k]):
                mv, m = abs(A[i, k]), i
        return m

    for k in range(n):
        if k < n - 1:
            ind_r[k] = row_max(k)
        if k > 0:
            ind_c[k] = col_max(k)
    for _ in range(n * n * 30 if n > 1 else 0):
        k, mv = 0, abs(A[0, ind_r[0]])
        for i in range(1, n - 1):
            if mv < abs(A[i, ind_r[i]]):
                mv, k = abs(A[i, ind_r[i]]), i
        l = ind_r[k]
        for i in range(1, n):
            if mv < abs(A[ind_c[i], i]):
                mv, k, l = abs(A[ind_c[i], i]), ind_c[i], i
        p = A[k, l]
        if abs(p) <= eps:
            break
        y = f32(f32(W[l] - W[k]) * f32(0.5))
        t = f32(abs(y) + _hypot32(p, y))
        s = _hypot32(p, t)
        c = f32(t / s)
        s = f32(p / s)
        t = f32(f32(p / t) * p)
        if y < 0:
            s, t = -s, -t
        A[k, l] = 0
        W[k] = f32(W[k] - t)
        W[l] = f32(W[l] + t)

        def rot(a0, b0):
            return f32(f32(a0 * c) - f32(b0 * s)), f32(f32(a0 * s) + f32(b0 * c))

        for i in range(0, k):
            A[i, k], A[i, l] = rot(A[i, k], A[i, l])
        for i in range(k + 1, l):
            A[k, i], A[i, l] = rot(A[k, i], A[i, l])
        for i in range(l + 1, n):
            A[k, i], A[l, i] = rot(A[k, i], A[l, i])
        for i in range(n):
            V[k, i], V[l, i] = rot(V[k, i], V[l, i])
        for idx in (k, l):
            if idx < n - 1:
                ind_r[idx] = row_max(idx)
            if idx > 0:
                ind_c[idx] = col_max(idx)
    for k in range(n - 1):
        m = k
        for i in range(k + 1, n):
            if W[m] < W[i]:
                m = i
        if k != m:
            W[[m, k]] = W[[k, m]]
            V[[m, k]] = V[[k, m]]
    return W, V


def _cross32(a, b):
    f32 = np.float32
    return np.array([f32(f32(a[1] * b[2]) - f32(a[2] * b[1])), f32(f32(a[2] * b[0]) - f32(a[0] * b[2])), f32(f32(a[0] * b[1]) - f32(a[1] * b[0]))], f32)


def wave_correct(rmats: Sequence[np.ndarray], kind: int) -> List[np.ndarray]:
    """cv.detail.waveCorrect (OpenCV stitching/src/motion_estimators.cpp) on float32 rotation matrices; returns new matrices.
    Float32 throughout as in OpenCV: moment matrix summed in binary32, cv::eigen's binary32 Jacobi solver, binary32 cross
    products; the 3x3 products go through cv::gemm, which accumulates a float product in double and rounds once."""
    rmats = [np.asarray(r, dtype=np.float32) for r in rmats]
    if len(rmats) <= 1:
        return list(rmats)
    if kind == WAVE_CORRECT_AUTO:
        kind = auto_detect_wave_correct_kind(rmats)
    moment = np.zeros((3, 3), np.float32)
    for r in rmats:
        col = r[:, 0].astype(np.float64)
        moment = (moment + (col[:, None] * col[None, :]).astype(np.float32)).astype(np.float32)
    _, evecs = eigen_symmetric_f32(moment)
    if kind == WAVE_CORRECT_HORIZ:
        rg1 = evecs[2].copy()
    elif kind == WAVE_CORRECT_VERT:
        rg1 = evecs[0].copy()
    else:
        raise ValueError("unsupported kind of wave correction")
    img_k = np.zeros(3, np.float32)
    for r in rmats:
        img_k = (img_k + r[:, 2]).astype(np.float32)
    rg0 = _cross32(rg1, img_k)
    rg0_norm = float(np.sqrt(np.sum(rg0.astype(np.float64) ** 2)))
    if rg0_norm <= np.finfo(np.float64).tiny:
        return list(rmats)
    rg0 = (rg0.astype(np.float64) / rg0_norm).astype(np.float32)
    rg2 = _cross32(rg0, rg1)
    conf = 0.0
    if kind == WAVE_CORRECT_HORIZ:
        for r in rmats:
            conf += float(np.dot(rg0.astype(np.float64), r[:, 0].astype(np.float64)))
    else:
        for r in rmats:
            conf -= float(np.dot(rg1.astype(np.float64), r[:, 0].astype(np.float64)))
    if conf < 0:
        rg0 = -rg0
        rg1 = -rg1
    rot = np.stack([rg0, rg1, rg2]).astype(np.float64)
    return [(rot @ r.astype(np.float64)).astype(np.float32) for r in rmats]


_MIRRORS = {
    "x": (-1, 1, 1),
    "y": (1, -1, 1),
    "z": (1, 1, -1),
    "x,y": (-1, -1, 1),
    "x,z": (-1, 1, -1),
    "y,z": (1, -1, -1),
    "x,y,z": (-1, -1, -1),
}


def mirror_rotate(R: np.ndarray, mirror_pano: Optional[str], rotate_pano_rad: float) -> np.ndarray:
    """R <- inv(M_mirror) @ inv(R_y(angle)) @ R, cast to float32 (sde.py:1413-1533)."""
    if not mirror_pano and rotate_pano_rad == 0:
        return R
    if rotate_pano_rad != 0:
        c, s = math.cos(rotate_pano_rad), math.sin(rotate_pano_rad)
        m_rot = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    else:
        m_rot = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]])
    if mirror_pano is not None:
        m_mirror = np.diag(_MIRRORS[mirror_pano])
    else:
        m_mirror = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]])
    return np.matmul(np.linalg.inv(m_mirror), np.matmul(np.linalg.inv(m_rot), R)).astype("float32")


def scale_for_megapix(megapix: float, full_w: int, full_h: int) -> float:
    """work/seam/compose scale rule (sde.py:751-752, :776-777, :1678-1679)."""
    if megapix <= 0:
        return 1.0
    return min(1.0, float(np.sqrt(megapix * 1e6 / (full_h * full_w))))


@dataclass
class ComposeGeometry:
    warper_scale: float
    compose_scale: float
    sizes: List[Tuple[int, int]]
    Ks: List[np.ndarray]
    Rs: List[np.ndarray]


def prepare_compose_cameras(
    cameras: Sequence[CameraParams],
    full_sizes: Sequence[Tuple[int, int]],
    work_scale: float,
    compose_megapix: float,
    wave_correct_kind: Optional[int] = None,
    mirror_pano: Optional[str] = None,
    rotate_pano_rad: float = 0.0,
) -> ComposeGeometry:
    """Everything between bundle adjustment and ``warper.warpRoi`` (sde.py:1373-1535, :1677-1695)."""
    cams = [c.clone() for c in cameras]
    scale = median_focal(cams)
    if wave_correct_kind is not None:
        rm = wave_correct([np.copy(c.R) for c in cams], wave_correct_kind)
        for c, r in zip(cams, rm):
            c.R = r
    if mirror_pano or rotate_pano_rad != 0:
        for c in cams:
            c.R = mirror_rotate(c.R, mirror_pano, rotate_pano_rad)
    w0, h0 = full_sizes[0]
    compose_scale = scale_for_megapix(compose_megapix, w0, h0) if compose_megapix > 0 else 1.0
    cwa = compose_scale / work_scale
    sizes, ks, rs = [], [], []
    for c, (fw, fh) in zip(cams, full_sizes):
        c.focal *= cwa
        c.ppx *= cwa
        c.ppy *= cwa
        sizes.append((int(round(fw * compose_scale)), int(round(fh * compose_scale))))
        ks.append(c.K().astype(np.float32))
        rs.append(np.asarray(c.R, dtype=np.float32))
    return ComposeGeometry(float(scale * cwa), compose_scale, sizes, ks, rs)
