"""Extended randomised sweep (not part of the suite): runs the parametrised fuzz tests of tests/test_gpu_parity.py with seeds outside
their committed ranges and reports every failing seed.   python tools/fuzz_sweep.py [first] [count]"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_gpu_parity as T  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
only = sys.argv[3] if len(sys.argv) > 3 else ""
bad = []
t0 = time.time()
for name in ("test_fuzz_multiband_layouts", "test_fuzz_warp_cameras", "test_fuzz_composer_rigs", "test_fuzz_strip_exchange", "test_fuzz_helpers", "test_fuzz_compensators", "test_fuzz_composer_float_rigs", "test_fuzz_composer_other_projections_and_rings", "test_fuzz_composer_with_gains",
             "test_fuzz_dp_seams", "test_fuzz_tall_frames_through_the_staged_pyramid_kernels"):
    if only and only not in name:
        continue
    fn = getattr(T, name)
    n = count if name in ("test_fuzz_multiband_layouts", "test_fuzz_warp_cameras") else max(1, count // (12 if "tall" in name else 4))
    for seed in range(first, first + n):
        try:
            fn(seed)
        except T.pytest.skip.Exception:
            continue
        except Exception as exc:  # noqa: BLE001
            bad.append((name, seed, repr(exc)[:200]))
            traceback.print_exc(limit=2)
    print(f"{name}: {n} seeds done, {len(bad)} failures so far, {time.time() - t0:.0f} s", flush=True)
# the batched non-separable projections (round 4: coordinate planes), seeds outside tests/test_generic_batched.py's range
if not only or only in "test_fuzz_generic_batched":
    import test_generic_batched as G  # noqa: E402
    n = max(1, count // 4)
    for seed in range(first, first + n):
        try:
            G.test_fuzz_generic_batched(seed)
        except T.pytest.skip.Exception:
            continue
        except Exception as exc:  # noqa: BLE001
            bad.append(("test_fuzz_generic_batched", seed, repr(exc)[:200]))
            traceback.print_exc(limit=2)
    print(f"test_fuzz_generic_batched: {n} seeds done, {len(bad)} failures so far, {time.time() - t0:.0f} s", flush=True)
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
