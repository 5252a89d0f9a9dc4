"""CPU tests: the C-ABI library loads and exports every declared symbol; no compute is possible without a GPU (loud
failure, no fallback); host-side logic (camera prep, rigs, plans)."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import opencv_starry_sky_panorama_stitcher_amd as cv
from opencv_starry_sky_panorama_stitcher_amd import camera as cam
from opencv_starry_sky_panorama_stitcher_amd import compose as cmp
from opencv_starry_sky_panorama_stitcher_amd import starfield

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(cv._lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = cv._lib.lib()
    syms = cv._lib.declared_symbols()
    assert len(syms) >= 70
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.ssp_version() >= 100


def test_header_cites_reference_call_sites():
    text = open(os.path.join(ROOT, "include", "ssp.h")).read()
    for site in ("sde.py:1696", "sde.py:1731", "sde.py:1613", "sde.py:1754", "sde.py:1886", "sde.py:1930"):
        assert site.split(":")[1] in text, site


def test_no_cpu_fallback_without_a_device():
    if cv.device_available():
        pytest.skip("a GPU is present: the loud-failure path is exercised on CPU-only hosts")
    K = np.eye(3, dtype=np.float32)
    with pytest.raises(cv.error, match="no CPU fallback|HIP"):
        cv.PyRotationWarper("spherical", 100.0).warpRoi((16, 16), K, K)
    with pytest.raises(cv.error):
        cv.UMat(np.zeros((4, 4, 3), np.uint8))
    with pytest.raises(cv.error):
        cv.detail_MultiBandBlender().prepare((0, 0, 8, 8))


def test_argument_errors_do_not_need_a_device():
    with pytest.raises(cv.error, match="unknown warper"):
        cv.PyRotationWarper("bogus", 1.0)
    with pytest.raises(cv.error):
        cv.PyRotationWarper("plane", 1.0).warpRoi((8, 8), np.eye(3), np.eye(3, dtype=np.float32))  # float64 K
    with pytest.raises(cv.error):
        cv.detail.Blender_createDefault(7)
    assert cv.detail.resultRoi(corners=[(0, 0), (-5, 3)], sizes=[(10, 10), (4, 20)]) == (-5, 0, 15, 23)


def test_level_strip_buffer_layout_is_host_arithmetic():
    """ssp_level_strip_buffer_bytes (the size both ends of an all-level strip agree on) needs no device: every level's image and weight
    planes with their aprons -- more than the payload, monotone in size and bands, float planes four times the 8-bit ones at level 0."""
    import ctypes as C
    L, chk = cv._lib.lib(), cv._lib.check
    def size(w, h, nb, flt=0):
        v = C.c_size_t()
        chk(L.ssp_level_strip_buffer_bytes(w, h, nb, flt, C.byref(v)))
        return v.value
    for nb in (2, 3, 5):
        m = 1 << nb
        w, h = 12 * m, 7 * m
        payload = 4 * w * h + sum((w >> l) * (h >> l) * 7 for l in range(1, nb + 1))
        assert payload < size(w, h, nb) < 2.5 * payload + 40000 * (nb + 1)
        assert size(w, h, nb) < size(w + m, h, nb) < size(w + m, h + m, nb)
        assert size(w, h, nb, 1) > size(w, h, nb) + 9 * w * h
        assert size(w, h, nb) % 256 == 0
    assert size(64, 64, 2) < size(64, 64, 3) < size(64, 64, 5)
    with pytest.raises(cv.error, match="not a multiple"):
        size(100, 64, 5)
    with pytest.raises(cv.error):
        size(64, 64, 99)


def test_wave_correct_makes_the_rig_horizontal():
    rng = np.random.default_rng(0)
    tilt = starfield.rot_x(0.2) @ starfield.rot_y(0.1)
    rmats = [(tilt @ starfield.rot_y(a) @ starfield.rot_x(rng.normal(0, 0.01))).astype(np.float32) for a in np.linspace(-1, 1, 7)]
    out = cam.wave_correct(rmats, cam.WAVE_CORRECT_HORIZ)
    for r in out:
        assert np.allclose(r @ r.T, np.eye(3), atol=1e-5)
    # after correction the camera x axes lie (almost) in the horizontal plane: their y components vanish
    assert max(abs(float(r[1, 0])) for r in out) < 0.02 < max(abs(float(r[1, 0])) for r in rmats)
    assert cam.auto_detect_wave_correct_kind(rmats) == cam.WAVE_CORRECT_HORIZ


def test_mirror_rotate_and_scales():
    R = starfield.rot_y(0.3).astype(np.float32)
    assert cam.mirror_rotate(R, None, 0) is R
    r2 = cam.mirror_rotate(R, "x,y", math.pi / 2)
    assert r2.dtype == np.float32 and np.allclose(r2 @ r2.T, np.eye(3), atol=1e-6)
    assert cam.scale_for_megapix(0.6, 5184, 3456) == pytest.approx(math.sqrt(0.6e6 / (5184 * 3456)))
    assert cam.scale_for_megapix(-1, 100, 100) == 1.0
    assert cam.median_focal([cam.CameraParams(focal=f) for f in (3.0, 1.0, 2.0)]) == 2.0
    assert cam.median_focal([cam.CameraParams(focal=f) for f in (4.0, 1.0, 2.0, 3.0)]) == 2.5
    assert cmp.num_bands_for(np.sqrt(2577 * 2557) * 42 / 100) == 9


def test_camera_params_json_roundtrip(tmp_path):
    doc = ["pinhole_focal_lengths_statistics:", {}, "pinhole_focal_lengths:", [1.0], "list_of_camera_params_for_disk_output:",
           [{"R": np.eye(3).tolist(), "aspect": 1.0, "focal": 1234.5, "ppx": 671.0, "ppy": 447.0, "t": [[0.0], [0.0], [0.0]]}]]
    p = tmp_path / "x.CameraParams.json"
    p.write_text(__import__("json").dumps(doc))
    cams = cam.load_camera_params_json(str(p))
    assert len(cams) == 1 and cams[0].focal == 1234.5 and cams[0].R.dtype == np.float32
    assert np.allclose(cams[0].K(), [[1234.5, 0, 671], [0, 1234.5, 447], [0, 0, 1]])


def test_rigs_and_frames_are_deterministic():
    for cfg, n in ((1, 3), (2, 6), (3, 12), (4, 48), (5, 96)):
        rig = starfield.make_rig(cfg, scale_div=16)
        assert rig.n == n and len(rig.Ks) == n and rig.Ks[0].dtype == np.float32
    rig = starfield.make_rig(2, scale_div=16, n_override=2)
    a = starfield.make_frames(rig)
    b = starfield.make_frames(rig)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and a[0].dtype == np.uint8 and a[0].shape == (135, 240, 3)
    sub = starfield.shard_rig(starfield.make_rig(4, scale_div=16), 3, 8)
    assert sub.n == 6


def test_composer_struct_matches_header_field_order():
    text = open(os.path.join(ROOT, "include", "ssp.h")).read()
    body = text[text.index("typedef struct {"):text.index("} ssp_compose_config;")]
    names = []
    for ln in body.splitlines():
        if ";" not in ln or ln.strip().startswith("/*"):
            continue
        decl = ln.split(";")[0].replace("const ", "").strip()
        first, *rest = [d.strip() for d in decl.split(",")]
        names.append(first.split()[-1].lstrip("*"))
        names += [d.lstrip("*") for d in rest]
    names = [n for n in names if n.isidentifier()]
    assert names == [f[0] for f in cmp._Cfg._fields_], (names, [f[0] for f in cmp._Cfg._fields_])
