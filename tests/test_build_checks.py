"""Build-time checks of hand-placed waits: the LDS-staged pyrDown kernels read their row slots with inline-assembly LDS reads behind
`s_waitcnt vmcnt(2)`, which is only enough while the compiler keeps (at least) two vector-memory stores between the row copies
(buffer_load ... lds) and that wait and does not reorder them.  The test compiles csrc/ssp_multiband.hip to gfx950 assembly and checks, for every
such wait in the two kernels, what stands between it and the preceding LDS-DMA copy.  CPU only (hipcc cross-compiles)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "opencv_starry_sky_panorama_stitcher_amd", "csrc", "ssp_multiband.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_staged_pyrdown_waits_leave_exactly_the_stores_in_flight(tmp_path):
    asm = tmp_path / "mb.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only", "-o", str(asm), SRC]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    text = asm.read_text()
    checked = 0
    for name in ("k_pyr_down_strip_lds", "k_pyr_down_strip_lds_lv"):
        bodies = re.findall(r"^(_Z\d+" + name + r"ILi\d+E\w*):[^\n]*\n(.*?)\n\s*s_endpgm", text, flags=re.S | re.M)
        assert bodies, f"{name}: kernel not found in the assembly"
        for sym, body in bodies:
            lines = [ln.strip() for ln in body.splitlines()]
            waits = [i for i, ln in enumerate(lines) if ln.startswith("s_waitcnt") and "vmcnt(2)" in ln]
            assert waits, f"{sym}: no s_waitcnt vmcnt(2) -- the hand-placed wait is gone"
            for w in waits:
                # back to the closest LDS-DMA copy in front of the wait
                j = w - 1
                stores = 0
                while j >= 0 and not ("buffer_load" in lines[j] and " lds" in lines[j]):
                    if lines[j].startswith(("global_store", "buffer_store", "flat_store")):
                        stores += 1
                    j -= 1
                if j < 0:
                    continue            # (a wait at a loop head: its copies are at the loop's end, checked through that occurrence)
                assert stores >= 2, f"{sym}: only {stores} vector stores between the last row copy and s_waitcnt vmcnt(2): a row slot could be read before its copy lands"
                checked += 1
    assert checked >= 2
