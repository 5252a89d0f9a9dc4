"""Shared helpers for the test-suite: seeded star patches and small camera rigs."""
import math

import numpy as np


def star_patch(w, h, seed, cn=3, dtype=np.uint8, n_stars=None):
    """Small synthetic star field: noise background + gaussian blobs, high local gradients."""
    rng = np.random.default_rng(seed)
    img = rng.normal(14.0, 4.0, (h, w, cn)).astype(np.float32)
    n_stars = n_stars if n_stars is not None else max(4, w * h // 400)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for _ in range(n_stars):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        sig = rng.uniform(0.8, 2.0)
        amp = rng.pareto(1.5) * 60 + 30
        g = amp * np.exp(-0.5 * ((xx - cx) ** 2 + (yy - cy) ** 2) / sig ** 2)
        img += g[:, :, None] * rng.uniform(0.8, 1.2, cn).astype(np.float32)[None, None, :]
    img = np.clip(img, 0, 255)
    if cn == 1:
        img = img[:, :, 0]
    return np.rint(img).astype(np.uint8) if dtype == np.uint8 else img.astype(dtype)


def big_frame(w, h, seed):
    """Camera-resolution u8c3 frame in O(w h): a tiled star_patch plus per-pixel noise (star_patch itself is O(stars w h))."""
    rng = np.random.default_rng(seed)
    tw, th = 640, 480
    tile = star_patch(tw, th, seed, n_stars=150).astype(np.int16)
    img = np.tile(tile, ((h + th - 1) // th, (w + tw - 1) // tw, 1))[:h, :w]
    img = img + rng.integers(0, 12, size=img.shape, dtype=np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def camera(w, h, hfov_deg=60.0, yaw=0.0, pitch=0.0, roll=0.0):
    f = (w / 2.0) / math.tan(math.radians(hfov_deg) / 2.0)
    K = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1]]).astype(np.float32)
    R = (rot_y(math.radians(yaw)) @ rot_x(math.radians(pitch)) @ rot_z(math.radians(roll))).astype(np.float32)
    return K, R, f
