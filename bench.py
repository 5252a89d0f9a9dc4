#!/usr/bin/env python3
"""bench.py -- MPix/s warped+blended into the final mosaic (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the hot path over one batch of synthetic star-field frames that are already resident in HBM:
warp(+mask) -> exposure apply -> mask prep -> pyramid build (blender.feed) for every frame, then blender.blend to the
8-bit mosaic (stitching_detailed_enhanced.py:1731-1938).
N = 1 (the default): BASELINE.json config 3, the one its end-to-end target is quoted on -- 12 4K frames in a CLOSED ring (30 degree yaw
steps, HFOV 60 degrees: SURVEY 8(d)'s rig; the two frames at +-165 degrees straddle u = +-pi*scale and are fed as their two live ends; the 357
degree arc of rounds 1-3 is carried beside it as `arc357`), spherical warp, GAIN_BLOCKS exposure compensation (fed once on the seam-scale warps outside the step as the
reference does, sde.py:1613; applied inside the step, fused into the warp), 5-band multiband blend.  The steps rotate through
3 distinct frame sets (0.9 GB), so no step finds its inputs in the 256 MiB Infinity Cache.  The line also carries `scale_base`: the
6-frame 2x3 block (no compensation) that one GPU handles in the N > 1 runs, measured in the same process.
N > 1 (launcher): the panorama has 6N frames (weak scaling, BASELINE config 4's 6 frames per GPU): every GPU owns a rectangle of the
panorama and receives, point-to-point over RCCL, the strips of its neighbours' pyramids (every level of the planes of their warped frames:
4 B/px at level 0, 7 B per sample above) that reach into it; it builds nothing for them and blends its rectangle bit-identically to a single
GPU (parallel.plan_strips).
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=str, default="3", help="3 (default: BASELINE config 3, 12x4K closed ring + gain blocks) | arc357 (the same frames as an open arc, 27 degree steps) | block (2x3 frames per GPU: what every rank runs when N > 1) | 2 | 5 "
                    "(SURVEY rigs; 5 = 8K float32 frames, 7 float bands).  With N > 1 the block rig is used whatever is given here (announced on stderr)")
    ap.add_argument("--warp", type=str, default="", help="projection instead of the rig's (e.g. fisheye, the reference's default, sde.py:237): a side measurement, "
                    "named in config.workload")
    ap.add_argument("--frame-sets", type=int, default=3, help="distinct input frame sets the steps rotate through (1 GPU; 3 x 299 MB defeats the 256 MiB Infinity Cache)")
    ap.add_argument("--pipeline", type=int, default=1, help="panoramas in flight on one GPU during the timed region (one composer + HIP stream each).  Default 1: "
                    "kernels run one after the other, so the per-kernel durations of the roofline object are the timed region's; the line also carries "
                    "the throughput with 2 in flight (in_flight_2)")
    ap.add_argument("--force-exchange", action="store_true", help="run the multi-GPU step (strip exchange over RCCL) even with one rank: a plumbing check")
    ap.add_argument("--frames", type=int, default=0, help="with --config N: number of frames (default: the rig's; 12 for config 5)")
    ap.add_argument("--serial-exchange", type=int, default=0, help="N>1: finish every panorama inside its own step instead of double buffering the strip exchange")
    ap.add_argument("--scale-div", type=int, default=1, help="shrink frames (debug only; invalid as a benchmark)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-frames", type=int, default=12)
    ap.add_argument("--no-scale-base", action="store_true", help="skip the 6-frame block measurement carried as scale_base")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel hipEvent pass")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes")
    ap.add_argument("--no-self-check", action="store_true", help="skip the eager-API recomposition (the profiled child passes: it launches the same kernels at other sizes)")
    ap.add_argument("--quick", type=int, default=0, help="1: the timed region and the per-kernel pass only (A/B runs of kernel variants, tools/ab_bench.sh)")
    return ap.parse_args()


# profile family (library side) -> kernel symbol fragments (rocprofv3 side)
KERNEL_OF = {"warp_fused": ("k_warp_strip_batch<", "k_warp_strip_planes<", "k_warp_f32_batch(", "k_warp_sep_f32c3("), "warp_rest": ("k_warp_rest_batch<",), "warp_prep": ("k_warp_prep_batch(",), "warp_cmap": ("k_warp_cmap_batch(",),
             "blend_level0": ("k_blend_oct<true", "k_blend_quad<true"),
             "blend_level": ("k_blend_oct<false", "k_blend_quad<false", "k_blend_level<"), "pyr_down_l0": ("k_pyr_down_strip_lds<", "k_pyr_down_strip<0", "k_pyr_down_2x2<0", "k_pyr_down_float<true"),
             "pyr_down": ("k_pyr_down_strip_lds_lv<", "k_pyr_down_strip<2", "k_pyr_down_strip<3", "k_pyr_down_2x2<2", "k_pyr_down_2x2<3", "k_pyr_down_float<false"), "border_l0": ("k_border0",),
             "pyr_apron": ("k_apron(",)}


def collect_pmc_traffic(args):
    """HBM bytes per launch from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
    rocprofv3 --pmc passes (no tracing domains besides --kernel-trace), FETCH_SIZE doubled (it tallies 128-B requests at 64 B
    on gfx950; calibrated in profiles/r01_pmc_calibration.txt), WRITE_SIZE exact.  The passes run as child processes BEFORE
    this process initialises the GPU, each under a timeout; any failure yields no traffic figure (null)."""
    import csv
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return {}
    res = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ssp_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--steps", "2",
               "--warmup", "1", "--no-cpu-baseline", "--no-profile", "--no-traffic", "--no-scale-base", "--no-self-check", "--frame-sets", "1", "--config", str(args.config), "--scale-div", str(args.scale_div)] + \
              (["--warp", args.warp] if args.warp else [])
        try:
            subprocess.run(cmd, timeout=180, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"), check=True)
            path = os.path.join(d, "p_counter_collection.csv")
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] != ctr:
                    continue
                for fam, syms in KERNEL_OF.items():
                    if any(sym in r["Kernel_Name"] for sym in syms):
                        res.setdefault(fam, {}).setdefault(ctr, []).append(float(r["Counter_Value"]))
        except Exception as exc:  # noqa: BLE001 -- measurement is optional
            print(f"pmc pass {ctr} failed: {exc}", file=sys.stderr)
            return {}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for fam, c in res.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
            w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
            out[fam] = {"read_bytes": 2.0 * f * 1024.0, "write_bytes": w * 1024.0}
    # third pass: what the kernels are bound by when it is not HBM -- vector-ALU instructions issued and busy cycles per launch
    # (SQ counters; own pass, --kernel-trace only).  Optional: a failure leaves the traffic figures intact.
    d = tempfile.mkdtemp(prefix="ssp_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
           os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-profile", "--no-traffic", "--no-scale-base", "--no-self-check", "--frame-sets", "1",
           "--config", str(args.config), "--scale-div", str(args.scale_div)] + (["--warp", args.warp] if args.warp else [])
    try:
        subprocess.run(cmd, timeout=180, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"), check=True)
        sq = {}
        for r in csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))):
            for fam, syms in KERNEL_OF.items():
                if any(sym in r["Kernel_Name"] for sym in syms):
                    sq.setdefault(fam, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for fam, c in sq.items():
            if all(k in c for k in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU")) and fam in out:
                n = len(c["SQ_WAVES"])
                out[fam]["waves"] = sum(c["SQ_WAVES"]) / n
                out[fam]["valu_insts_per_wave"] = sum(c["SQ_INSTS_VALU"]) / max(sum(c["SQ_WAVES"]), 1.0)
                out[fam]["valu_active_cycles"] = 4.0 * sum(c["SQ_ACTIVE_INST_VALU"]) / n      # a wave64 VALU instruction occupies its SIMD for 4 cycles
    except Exception as exc:  # noqa: BLE001
        print(f"pmc pass SQ failed: {exc}", file=sys.stderr)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def block_rig(starfield, world, rank, div, float8k=False):
    """6N frames: rows of pitch (-10, +10[, -30, +30]) x columns of 30 degree yaw steps; each GPU owns a 2x3 block.
    N = 8 is config 4's layout: 4 rows x 12 yaw positions, four closed rings -- the frames of the outer columns straddle u = +-pi*scale
    and are fed as their two live ends (parallel.feed_parts), so every GPU still warps about six frames' worth of pixels."""
    from opencv_starry_sky_panorama_stitcher_amd.starfield import Rig, _finish
    blocks_x = {1: 1, 2: 2, 4: 4, 8: 4}.get(world, world)
    blocks_y = max(1, world // blocks_x)
    cols, rows = 3 * blocks_x, 2 * blocks_y
    pitches_all = [(-10.0 - 20.0 * (rows // 2 - 1)) + 20.0 * r for r in range(rows)]
    yaws_all = [(c - (cols - 1) / 2.0) * 30.0 for c in range(cols)]
    bx, by = rank % blocks_x, rank // blocks_x
    yaws, pitches = [], []
    for r in range(2):
        for c in range(3):
            yaws.append(yaws_all[bx * 3 + c])
            pitches.append(pitches_all[by * 2 + r])
    if float8k:     # BASELINE config 5's frames and pyramids on the same block layout: 8K float32, 7 float bands
        rig = Rig(f"block 2x3 of {rows}x{cols} frames, 8K f32, spherical + multiband(7, float)", 5, 7680 // div, 4320 // div, 60.0, yaws, pitches, "spherical",
                  "multiband", 7, dtype="f32")
    else:
        rig = Rig(f"block 2x3 of {rows}x{cols} frames, 4K, spherical + multiband(5)", 4, 3840 // div, 2160 // div, 60.0, yaws, pitches, "spherical",
                  "multiband", 5)
    return _finish(rig), (rows, cols)


def main():
    args = parse()
    if args.quick:
        args.no_cpu_baseline = args.no_traffic = args.no_scale_base = True
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    # PMC traffic passes first: child processes, started before this process touches the GPU
    pmc = {}
    if world == 1 and not args.no_traffic and not args.no_profile:
        pmc = collect_pmc_traffic(args)
    import ctypes as C

    dist = None
    torch = None
    if world > 1 or args.force_exchange:
        # torch FIRST: it ships its own HIP runtime; once that is initialised libssp_hip.so shares it.  The other order (this
        # library's /opt/rocm runtime first) leaves torch without devices ("No HIP GPUs are available").
        import torch
        import torch.distributed as dist

        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517")):
            os.environ.setdefault(k, v)          # --force-exchange without a launcher: a world of one
        # SSP_DIST_BACKEND=gloo + SSP_SHARE_GPU=1: rehearsal of the N>1 path with all ranks on one GPU (timing meaningless)
        backend = os.environ.get("SSP_DIST_BACKEND", "nccl")
        if os.environ.get("SSP_SHARE_GPU"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        torch.zeros(1, device="cuda")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import opencv_starry_sky_panorama_stitcher_amd as cv
    from opencv_starry_sky_panorama_stitcher_amd import compose as cmp
    from opencv_starry_sky_panorama_stitcher_amd import starfield

    L = cv._lib.lib()
    cv._lib.check(L.ssp_init(local_rank))
    if torch is not None:
        # run the library on torch's current stream so that RCCL orders against our kernels
        cv._lib.check(L.ssp_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    # ---- workload -------------------------------------------------------------------------------------------------------
    multi = world > 1 or args.force_exchange
    if multi and str(args.config) not in ("block", "5"):
        if rank == 0:
            print(f"note: --config {args.config} is a 1-GPU workload; N > 1 runs the 2x3 block rig (6 frames per GPU)", file=sys.stderr)
        args.config = "block"

    def build_workload(config):
        """-> (rig, workload name, frame sets [list of UMat lists], host frames of set 0, host seam frames)"""
        if config == "block" or multi:
            f8k = str(config) == "5"
            rig, layout = block_rig(starfield, world, rank, args.scale_div, f8k)
            res = ("8K f32" if f8k else "4K") if args.scale_div == 1 else f"{rig.width}x{rig.height}"
            name = (f"{6 * world}x{res} star-field frames ({layout[0]} rows x {layout[1]} cols, 2x3 block per GPU), spherical warp + "
                    f"{rig.num_bands}-band {'float ' if f8k else ''}multiband blend")
        elif str(config) == "arc357":
            # rounds 1-3's workload: the same 12 frames at 27 degree steps, an open 357 degree arc without a straddling frame
            rig = starfield.make_rig(3, scale_div=args.scale_div, arc_step=27.0)
            name = rig.name
        else:
            cfg = int(config)
            if cfg == 5:   # one GPU's share of the 4 x 24 layout: consecutive 8K float32 frames of one row
                nfr = args.frames or 12
                rig = starfield.make_rig(5, scale_div=args.scale_div, n_override=nfr)
                rig.yaws_deg, rig.pitches_deg, rig.Ks, rig.Rs = rig.yaws_deg[:nfr], rig.pitches_deg[:nfr], rig.Ks[:nfr], rig.Rs[:nfr]
            else:
                rig = starfield.make_rig(cfg, scale_div=args.scale_div, n_override=(args.frames or None))
            name = rig.name
            if cfg == 3 and not args.frames:
                res = "4K" if args.scale_div == 1 else f"{rig.width}x{rig.height}"
                name = (f"BASELINE config 3: 12x{res} star-field frames (closed 360 deg ring, 30 deg yaw steps: SURVEY 8(d)), spherical warp + GAIN_BLOCKS exposure compensation "
                        f"(seam-scale feed outside the step, apply fused into the warp) + {rig.num_bands}-band multiband blend")
        if args.warp and args.warp != rig.warp:
            rig.warp = args.warp
            name = f"[--warp {args.warp}] " + name.replace("spherical warp", f"{args.warp} warp")
        host, seams = starfield.make_frames(rig, want_seam=True)
        # further frame sets: the same sky shifted sideways (distinct memory is what matters: a step must not find its inputs in the
        # Infinity Cache); the multi-GPU step keeps one set (its double buffering pins the frames of two panoramas)
        nsets = 1 if (multi or rig.dtype == "f32") else max(1, args.frame_sets)
        sets = [[cv.UMat(f) for f in host]]
        for k in range(1, nsets):
            sets.append([cv.UMat(np.ascontiguousarray(np.roll(f, 97 * k, axis=1))) for f in host])
        return rig, name, sets, host, seams

    t0 = time.time()
    rig, workload, frame_sets, frames_np, seams_np = build_workload(args.config)
    frames = frame_sets[0]
    gen_s = time.time() - t0
    mask_prep = True
    # --pipeline P (one GPU): P composers, each with its own HIP stream, take the steps round robin, so P panoramas are in flight and
    # the latency-bound small pyramid levels of one overlap the large kernels of another.  Every step is still one complete panorama.
    depth = max(1, args.pipeline) if not multi else 1

    def make_compensator(rig, seams_np):
        """sde.py:1543-1613: seam-scale warps of the frames and of their masks, compensator.feed -- once, outside the step."""
        if not rig.expos_comp:
            return None
        comp = cv.detail.ExposureCompensator_createDefault(rig.expos_comp)
        ws = cv.PyRotationWarper(rig.warp, rig.focal * rig.seam_scale)
        cs, ims, mks = [], [], []
        for i in range(rig.n):
            K = rig.Ks[i].copy()
            K[0, 0] *= rig.seam_scale; K[0, 2] *= rig.seam_scale; K[1, 1] *= rig.seam_scale; K[1, 2] *= rig.seam_scale
            cnr, im = ws.warp(seams_np[i], K, rig.Rs[i], cv.INTER_AREA, cv.BORDER_REFLECT)
            _, mk = ws.warp(255 * np.ones(seams_np[i].shape[:2], np.uint8), K, rig.Rs[i], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
            cs.append(cnr); ims.append(im); mks.append(mk)
        comp.feed(corners=cs, images=ims, masks=mks)
        return comp

    comp = make_compensator(rig, seams_np)

    def make_composer(own_stream, rig=rig, comp=comp):
        c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend=rig.blend, num_bands=rig.num_bands,
                         float_frames=(rig.dtype == "f32"), mask_prep=mask_prep, seam_size=rig.seam_size, seam_aspect=rig.seam_scale,
                         own_stream=own_stream)
        if comp is not None:
            c.set_compensator(comp)     # every composer the factory hands out carries the compensation
        return c
    composers = [make_composer(depth > 1) for _ in range(depth)]
    composer = composers[0]

    exchange, pipeline = None, None
    if world > 1 or args.force_exchange:
        from opencv_starry_sky_panorama_stitcher_amd import parallel
        # every rank derives the rois of ALL frames of the panorama (O(N) geometry) so that all ranks agree on the plan
        all_Ks, all_Rs, frame_owner = [], [], []
        for r in range(world):
            rr, _ = block_rig(starfield, world, r, args.scale_div, str(args.config) == "5")
            all_Ks += rr.Ks; all_Rs += rr.Rs; frame_owner += [r] * rr.n
        # what the composers feed: a frame's roi, or the two live ends of a frame that straddles u = +-pi*scale (N = 8: the outer columns)
        fparts = parallel.feed_parts(cv, rig.warp, rig.focal, (rig.width, rig.height), all_Ks, all_Rs, frame_owner, rig.num_bands)
        all_corners, all_sizes, owner = fparts.corners, fparts.sizes, fparts.owner
        # all-level strips (every pyramid level of the neighbours' planes, 4 B/px at level 0 + 7 B per sample above) go point-to-point to
        # the neighbours that need them; the receiver builds nothing for them (SSP_STRIP_LEVELS=0: level-0 strips, pyramids rebuilt).
        # Double buffered over two panoramas (parallel.HipStripPipeline): a step still launches one panorama's kernels in serial
        # order on one stream and completes one panorama, but the strips posted in it have until the next step to arrive.
        # --serial-exchange 1: warp -> pyramids -> export -> wait -> finish inside every step (the transfer is exposed).
        levels = os.environ.get("SSP_STRIP_LEVELS", "1") != "0"
        if args.serial_exchange:
            exchange = parallel.HipStripExchange(composer, dist, torch, all_corners, all_sizes, owner, rig.num_bands, levels=levels, pano_roi=fparts.pano_roi)
        else:
            spare = iter([composer])
            pipeline = parallel.HipStripPipeline(lambda: next(spare, None) or make_composer(False), dist, torch, all_corners, all_sizes, owner, rig.num_bands, levels=levels, pano_roi=fparts.pano_roi)
            exchange = pipeline.ex[0]

    counter = [0]

    def step():
        if pipeline is not None:
            pipeline.step(frames)
        elif exchange is not None:
            exchange.run(frames)
        else:
            composers[counter[0] % depth].run(frame_sets[counter[0] % len(frame_sets)])
            counter[0] += 1

    def sync():
        for cc in composers:
            cc.sync()
        cv._lib.check(L.ssp_use_stream(None))
        cv._lib.check(L.ssp_sync())
        if torch is not None:
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- timed region ---------------------------------------------------------------------------------------------------
    for _ in range(2 if pipeline is not None else depth):
        step()          # set-up, not a warm-up step: the first pass allocates the pool's blocks and (N>1) opens the point-to-point channels
    for _ in range(args.warmup):
        step()
    for ex in (pipeline.ex if pipeline is not None else ([exchange] if exchange is not None else [])):
        ex.time_waits(True)      # every panorama of the timed region: how long its collapse waited for the neighbours' strips
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync(); barrier(); sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    # ---- what the exchange moved and what it cost, per rank (N > 1): so that a scaling curve explains itself --------------------------------
    exchange_report = None
    if exchange is not None:
        exs = pipeline.ex if pipeline is not None else [exchange]
        ws = [ex.wait_stats() for ex in exs]
        for ex in exs:
            ex.time_waits(False)
        npan = sum(w["panoramas"] for w in ws)
        mine = dict(exchange.traffic(), rank=rank, feed_units=len(composer.parts()), owned=list(exchange.plan.owned[rank]), region=list(exchange.plan.region[rank]),
                    panoramas_timed=npan,
                    recv_wait_stream_ms=(round(sum(w["recv_wait_stream_ms"] * w["panoramas"] for w in ws if w["panoramas"]) / npan, 4) if npan else None),
                    recv_wait_host_ms=(round(sum(w["recv_wait_host_ms"] * w["panoramas"] for w in ws if w["panoramas"]) / npan, 4) if npan else None))
        gathered = [None] * world
        if dist is not None and world > 1:
            dist.all_gather_object(gathered, mine)
        else:
            gathered = [mine]
        exchange_report = {"protocol": mine["protocol"], "double_buffered": pipeline is not None, "backend": (dist.get_backend() if dist is not None else None),
                           "note": "sent_bytes / recv_bytes: strip buffers per neighbour and panorama; recv_wait_stream_ms: GPU time the collapse's stream waited for its "
                                   "receives (event pair around the wait; RCCL orders transfers against the stream), recv_wait_host_ms: the same on the host (gloo blocks there)",
                           "per_rank": gathered}
    # latency of ONE panorama (nothing else in flight), for the record next to the throughput figure
    latency_ms = ms_per_step
    if depth > 1:
        sync()
        t1 = time.perf_counter()
        for _ in range(max(3, min(args.steps, 10))):
            composer.run(frames)
        composer.sync()
        latency_ms = (time.perf_counter() - t1) / max(3, min(args.steps, 10)) * 1e3
    mpix_in = rig.n * world * rig.width * rig.height / 1e6
    value = mpix_in / (ms_per_step / 1e3)
    # the same steps with TWO panoramas in flight (second composer, one HIP stream each): throughput only, reported beside `value`
    in_flight_2 = None
    if depth == 1 and world == 1 and exchange is None and not args.no_profile and not args.quick:
        pair = [make_composer(True), make_composer(True)]
        for i in range(4):
            pair[i % 2].run(frame_sets[i % len(frame_sets)])
        for cc in pair:
            cc.sync()
        t2 = time.perf_counter()
        for i in range(args.steps):
            pair[i % 2].run(frame_sets[i % len(frame_sets)])
        for cc in pair:
            cc.sync()
        ms2 = (time.perf_counter() - t2) / args.steps * 1e3
        in_flight_2 = {"ms_per_step": round(ms2, 4), "value": round(mpix_in / (ms2 / 1e3), 1), "unit": "MPix/s"}
        del pair, cc        # (the loop variable would keep the second composer and its stream alive)
        cv._lib.check(L.ssp_use_stream(None))

    # the same step with the frames coming from host memory and the 8-bit mosaic going back ("also report with H2D/D2H
    # included"): never `value`, a side figure
    with_pcie = None
    if world == 1 and exchange is None and not args.no_profile and not args.quick:
        reps = max(3, min(args.steps, 5))
        t3 = time.perf_counter()
        for _ in range(reps):
            up = [cv.UMat(f) for f in frames_np]
            composer.run(up)
            host_mosaic = composer.result()[0].get()
        ms3 = (time.perf_counter() - t3) / reps * 1e3
        with_pcie = {"ms_per_step": round(ms3, 3), "value": round(mpix_in / (ms3 / 1e3), 1), "unit": "MPix/s",
                     "h2d_MB": round(sum(f.nbytes for f in frames_np) / 1e6, 1), "d2h_MB": round(host_mosaic.nbytes / 1e6, 1)}
        del up, host_mosaic

    # ---- dropin_umat: the reference's OWN call sequence (sde.py:1673-1930: warpRoi x n, then per image warp, warp(mask), compensator.apply,
    # astype(int16), dilate, resize, bitwise_and, blender.feed, then blender.blend), object by object through the cv2-shaped API on UMats --
    # compose.compose_panorama with this package as `cv` -- on the same 12 frames; the seam-scale stage (sde.py:1543-1624) outside the step,
    # as for `value`.  The calls return deferred arrays and blend() runs the recognised sequence as one plan (deferred.py).
    dropin_umat, dropin_mosaic = None, None
    if world == 1 and exchange is None and depth == 1 and not args.no_profile and not args.quick and rig.dtype == "u8":
        seam_state = cmp.seam_stage(cv, [cv.UMat(s) for s in seams_np], rig.Ks, rig.Rs, rig.warp, rig.focal, rig.seam_scale, rig.expos_comp)

        def dropin(fs):
            return cmp.compose_panorama(cv, fs, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend=rig.blend, num_bands=rig.num_bands,
                                        expos_comp=rig.expos_comp, seam_aspect=rig.seam_scale, mask_prep=mask_prep, seam_state=seam_state)
        from opencv_starry_sky_panorama_stitcher_amd import deferred as _dfr
        for i in range(3):
            res_d = dropin(frame_sets[i % len(frame_sets)])
        sync()
        planned0 = _dfr.stats["planned"]
        td = time.perf_counter()
        for i in range(args.steps):
            res_d = dropin(frame_sets[i % len(frame_sets)])
        sync()
        msd = (time.perf_counter() - td) / args.steps * 1e3
        res_d = dropin(frame_sets[0])
        dropin_mosaic = res_d.mosaic.get()
        dropin_umat = {"ms_per_step": round(msd, 4), "value": round(mpix_in / (msd / 1e3), 1), "unit": "MPix/s", "ratio_to_value_step": round(msd / ms_per_step, 3),
                       "blends_run_as_one_plan": _dfr.stats["planned"] - planned0 - 1, "steps": args.steps,
                       "what": "compose.compose_panorama(cv, UMat frames, ...): the reference's loop call for call (sde.py:1673-1930), int16 result + mask + 8-bit mosaic per panorama"}
        del res_d

    # ---- self_check (every mode, also --quick): the timed composer's mosaic of frame set 0 against the SAME panorama composed call by call through the
    # eager object API -- other kernels for the warp (k_warp_sep_u8c3 / k_warp_generic per frame), the gains, the mask preparation and the feed: an A/B
    # run of a kernel variant that computes something else shows here, not only in the parity tests
    self_check = None
    if world == 1 and exchange is None and rig.dtype == "u8" and not args.no_self_check:
        composer.run(frames)
        fused = [u.get() for u in composer.result()[:2]]
        os.environ["SSP_EAGER"] = "1"
        try:
            st = cmp.seam_stage(cv, [cv.UMat(s) for s in seams_np], rig.Ks, rig.Rs, rig.warp, rig.focal, rig.seam_scale, rig.expos_comp)
            eager = cmp.compose_panorama(cv, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend=rig.blend, num_bands=rig.num_bands,
                                         expos_comp=rig.expos_comp, seam_aspect=rig.seam_scale, mask_prep=mask_prep, seam_state=st)
            e_mosaic, e_mask = eager.mosaic.get(), eager.result_mask.get()
        finally:
            del os.environ["SSP_EAGER"]
        self_check = {"against": "the same frames through the eager object API (call by call, per-frame kernels)", "mosaic_identical": bool(np.array_equal(fused[0], e_mosaic)),
                      "mask_identical": bool(np.array_equal(fused[1], e_mask))}
        if not (self_check["mosaic_identical"] and self_check["mask_identical"]):
            print("bench.py: SELF CHECK FAILED -- the batched composer and the eager object API disagree on the mosaic of frame set 0", file=sys.stderr)
        del fused, eager, e_mosaic, e_mask, st

    # ---- scale_base: the 6-frame block one GPU handles in the N > 1 runs (no compensation), timed the same way in this process -------
    scale_base = None
    if world == 1 and exchange is None and str(args.config) == "3" and not args.no_scale_base and not args.no_profile:
        b_rig, b_name, b_sets, _, _ = build_workload("block")
        b_comp = make_composer(False, rig=b_rig, comp=None)
        for i in range(3):
            b_comp.run(b_sets[i % len(b_sets)])
        b_comp.sync(); cv._lib.check(L.ssp_sync())
        tb = time.perf_counter()
        for i in range(args.steps):
            b_comp.run(b_sets[i % len(b_sets)])
        b_comp.sync(); cv._lib.check(L.ssp_sync())
        msb = (time.perf_counter() - tb) / args.steps * 1e3
        scale_base = {"workload": b_name, "frames": b_rig.n, "ms_per_step": round(msb, 4), "value": round(b_rig.n * b_rig.width * b_rig.height / 1e6 / (msb / 1e3), 1),
                      "unit": "MPix/s", "note": "per-GPU work of the N > 1 lines; weak-scaling efficiency = value(N) / (N x this value)"}
        del b_comp, b_sets

    # ---- tables_rebuilt: the same steps with the composer's geometry knowledge dropped before every step, i.e. the prep launch (projection /
    # resize tables, dilated seam mask) and the rest-list launch in EVERY panorama -- like for like against cv2, which rebuilds its maps in every
    # warp call (sde.py:1731, :1740).  `value` reuses them per composer; this is the figure without that.
    tables_rebuilt = None
    if world == 1 and exchange is None and depth == 1 and not args.no_profile and not args.quick and rig.dtype == "u8":
        sync()
        tr = time.perf_counter()
        for i in range(args.steps):
            composer.forget_geometry()
            composer.run(frame_sets[i % len(frame_sets)])
        sync()
        msr = (time.perf_counter() - tr) / args.steps * 1e3
        tables_rebuilt = {"ms_per_step": round(msr, 4), "value": round(mpix_in / (msr / 1e3), 1), "unit": "MPix/s",
                          "note": "prep (tables) + rest-list launches in every step; forget_geometry() waits for the previous step's list read-back, so steps do not overlap on the host"}
        for _ in range(3):
            composer.run(frames)       # back to the steady state for the per-kernel pass
        sync()

    # ---- coordinate_planes: the same steps by a composer whose separable projection reads its map from coordinate planes (ssp_compose_config.coordinate_planes),
    # the choice for callers that compose many panoramas with the same cameras: 4 B per warped pixel and a plane-building launch per geometry for a warp
    # kernel with a third fewer instructions.  `value` stays on the default (tables: one panorama per camera set pays nothing up front).
    coordinate_planes = None
    if world == 1 and exchange is None and depth == 1 and not args.no_profile and not args.quick and rig.dtype == "u8" and rig.warp in ("spherical", "cylindrical", "mercator"):
        pc = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend=rig.blend, num_bands=rig.num_bands, mask_prep=mask_prep,
                          seam_size=rig.seam_size, seam_aspect=rig.seam_scale, coordinate_planes=True)
        if comp is not None:
            pc.set_compensator(comp)
        sync()
        tp = time.perf_counter()
        pc.run(frames)
        sync()
        first_ms = (time.perf_counter() - tp) * 1e3
        for _ in range(3):
            pc.run(frames)
        sync()
        tp = time.perf_counter()
        for i in range(args.steps):
            pc.run(frame_sets[i % len(frame_sets)])
        sync()
        msp = (time.perf_counter() - tp) / args.steps * 1e3
        p_mosaic = pc.result()[0].get()
        composer.run(frame_sets[(args.steps - 1) % len(frame_sets)])
        same = bool(np.array_equal(p_mosaic, composer.result()[0].get()))
        coordinate_planes = {"ms_per_step": round(msp, 4), "value": round(mpix_in / (msp / 1e3), 1), "unit": "MPix/s", "first_panorama_ms": round(first_ms, 2),
                             "planes_MB": round(4.0 * sum(roi[2] * roi[3] for _, roi in pc.parts()) / 1e6, 1),
                             "mosaic_identical_to_value_composer": same,
                             "note": "ssp_compose_config.coordinate_planes = 1: the map of every warped pixel read from planes built once per camera set instead of computed from tables"}
        del pc, p_mosaic
        for _ in range(3):
            composer.run(frames)
        sync()

    # ---- arc357: the workload of rounds 1-3 -- the same 12 frames at 27 degree steps, an open arc in which no frame straddles u = +-pi*scale.
    # Same input pixels per step as `value`; carried so that the rounds stay comparable.
    arc357 = None
    if world == 1 and exchange is None and str(args.config) == "3" and not args.frames and not args.no_scale_base and not args.no_profile:
        r_rig = starfield.make_rig(3, scale_div=args.scale_div, arc_step=27.0)
        r_host, r_seams = starfield.make_frames(r_rig, want_seam=True)
        r_sets = [[cv.UMat(f) for f in r_host], [cv.UMat(np.ascontiguousarray(np.roll(f, 97, axis=1))) for f in r_host]]
        r_comp = make_composer(False, rig=r_rig, comp=make_compensator(r_rig, r_seams))
        for i in range(4):
            r_comp.run(r_sets[i % 2])
        r_comp.sync(); cv._lib.check(L.ssp_sync())
        t4 = time.perf_counter()
        for i in range(args.steps):
            r_comp.run(r_sets[i % 2])
        r_comp.sync(); cv._lib.check(L.ssp_sync())
        ms4 = (time.perf_counter() - t4) / args.steps * 1e3
        arc357 = {"workload": r_rig.name, "ms_per_step": round(ms4, 4), "value": round(mpix_in / (ms4 / 1e3), 1), "unit": "MPix/s", "pano": list(r_comp.pano_roi()),
                  "warped_MPix": round(sum(p[1][2] * p[1][3] for p in r_comp.parts()) / 1e6, 1), "rest_tiles": r_comp.warp_rest_tiles()[1]}
        del r_comp, r_sets, r_host

    # ---- per-kernel durations (hipEvents on the launch stream) for the roofline object ----------------------------------------
    roofline, kernels = None, []
    if not args.no_profile:
        # EVERY rank runs these steps (a step of the N>1 path is an exchange with the neighbours); rank 0's durations are reported
        cv._lib.check(L.ssp_profile_reset())
        cv._lib.check(L.ssp_profile_enable(1))
        reps = max(4, min(args.steps, 10))
        for _ in range(reps):       # the same step() as the timed region: with --pipeline > 1 the durations include the co-running panorama
            step()
        sync()
        cv._lib.check(L.ssp_profile_enable(0))
        barrier()
    if not args.no_profile and rank == 0:
        n = C.c_int()
        cv._lib.check(L.ssp_profile_count(C.byref(n)))
        for i in range(n.value):
            name = C.create_string_buffer(64)
            launches, ms, ab = C.c_int(), C.c_float(), C.c_double()
            cv._lib.check(L.ssp_profile_get(i, name, 64, C.byref(launches), C.byref(ms), C.byref(ab)))
            if launches.value:
                kernels.append({"kernel": name.value.decode(), "launches_per_step": launches.value / reps, "avg_us": ms.value * 1e3 / launches.value,
                                "total_ms_per_step": ms.value / reps, "algo_bytes_per_launch": ab.value / launches.value,
                                "achieved_GBps": (ab.value / launches.value) / (ms.value * 1e-3 / launches.value) / 1e9 if ms.value > 0 else 0.0})
        for k in kernels:
            t = pmc.get(k["kernel"])
            k["hbm_traffic_bytes_per_launch"] = (t["read_bytes"] + t["write_bytes"]) if t else None
            k["hbm_read_bytes_per_launch"] = t["read_bytes"] if t else None
            k["hbm_write_bytes_per_launch"] = t["write_bytes"] if t else None
            # share of the vector-ALU issue capacity the kernel used over its own duration: 256 CUs x 4 SIMDs at 2.4 GHz
            k["valu_insts_per_wave"] = round(t["valu_insts_per_wave"], 1) if t and "valu_insts_per_wave" in t else None
            k["valu_busy"] = round(t["valu_active_cycles"] / (k["avg_us"] * 1e-6 * 2.4e9 * 256 * 4), 3) if t and "valu_active_cycles" in t and k["avg_us"] > 0 else None
        kernels.sort(key=lambda k: -k["total_ms_per_step"])
        dom = next((k for k in kernels if k["algo_bytes_per_launch"] > 0), None)
        if dom:
            peak = 8000.0  # MI355X HBM3E, GB/s (MI355X_MICROARCH.md: 8.0 TB/s spec; ~6.3 TB/s achievable)
            roofline = {"bound": "hbm", "kernel": dom["kernel"], "achieved": round(dom["achieved_GBps"], 1), "peak": peak, "unit": "GB/s",
                        "frac": round(dom["achieved_GBps"] / peak, 4), "traffic": dom["hbm_traffic_bytes_per_launch"], "avg_us": round(dom["avg_us"], 2),
                        "algo_bytes_per_launch": dom["algo_bytes_per_launch"], "valu_busy": dom.get("valu_busy"), "valu_insts_per_wave": dom.get("valu_insts_per_wave")}
            # what actually bounds this kernel is the vector ALU, not HBM: a wave64 VALU instruction occupies its SIMD for 4 cycles, so the
            # issue ceiling of the launch is VALU instructions per wave x waves x 4 cycles / (1024 SIMDs x 2.4 GHz); valu_frac = that time / avg
            t_dom = pmc.get(dom["kernel"])
            if t_dom and "waves" in t_dom and "valu_insts_per_wave" in t_dom:
                floor_us = t_dom["valu_insts_per_wave"] * t_dom["waves"] * 4.0 / (1024 * 2.4e9) * 1e6
                roofline["valu_issue_floor_us"] = round(floor_us, 1)
                roofline["valu_frac"] = round(floor_us / dom["avg_us"], 4)
            # the ceiling this box actually reaches with a plain device-to-device copy (read + write bytes): a 16-byte-per-lane copy
            # KERNEL of this library (what MI355X_MICROARCH.md's 6.29 TB/s was measured with), and the runtime's hipMemcpyAsync D2D beside it
            try:
                nbytes = 512 << 20
                bufs = [cv.UMat.empty(nbytes, 1, 1, np.uint8) for _ in range(2)]
                ptrs = [C.c_void_p(b.info()[5]) for b in bufs]
                rates = {}
                for key, fn in (("kernel", L.ssp_device_copy_kernel), ("memcpy", L.ssp_device_copy)):
                    for _ in range(2):
                        cv._lib.check(fn(ptrs[0], ptrs[1], C.c_size_t(nbytes)))
                    cv._lib.check(L.ssp_sync())
                    tc = time.perf_counter()
                    for _ in range(8):
                        cv._lib.check(fn(ptrs[0], ptrs[1], C.c_size_t(nbytes)))
                    cv._lib.check(L.ssp_sync())
                    rates[key] = 8 * 2.0 * nbytes / (time.perf_counter() - tc) / 1e9
                ceiling = rates["kernel"]
                roofline["copy_ceiling_GBps"] = round(ceiling, 1)
                roofline["copy_ceiling_kind"] = "own 16-B/lane copy kernel (k_copy16), 512 MiB, read + write bytes"
                roofline["hipMemcpyDtoD_GBps"] = round(rates["memcpy"], 1)
                if dom["hbm_traffic_bytes_per_launch"]:
                    roofline["traffic_GBps"] = round(dom["hbm_traffic_bytes_per_launch"] / (dom["avg_us"] * 1e-6) / 1e9, 1)
                    roofline["traffic_frac_of_copy_ceiling"] = round(roofline["traffic_GBps"] / ceiling, 4)
                del bufs
            except Exception as exc:  # noqa: BLE001 -- an extra, never fatal
                print(f"copy ceiling not measured: {exc}", file=sys.stderr)

    # ---- CPU baseline: the oracle (a scalar port of OpenCV's algorithm structure) on a bounded sample ----------------------------
    cpu_baseline, cpu_ref, parity = None, None, None
    if not args.no_cpu_baseline and rank == 0 and world == 1:   # the CPU baseline is an N=1 figure
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_cv as ocv

        nf = max(1, min(args.cpu_baseline_frames, rig.n))
        fr = [np.ascontiguousarray(f) for f in frames_np[:nf]]
        if rig.dtype != "f32":
            def cpu_pass():
                t0 = time.perf_counter()
                cmp.compose_panorama(ocv, fr, rig.Ks[:nf], rig.Rs[:nf], warp=rig.warp, warper_scale=rig.focal, blend=rig.blend, num_bands=rig.num_bands,
                                     seam_frames=seams_np[:nf], seam_aspect=rig.seam_scale, expos_comp=rig.expos_comp)
                return time.perf_counter() - t0
            # (a) one core: repeat the sample until about 8 s of CPU work are done (at most 4 passes); the fastest pass counts
            times = []
            while len(times) < 4 and sum(times) < 8.0:
                times.append(cpu_pass())
            dt1 = min(times)
            # (b) the same oracle with its row loops under OpenMP (liborc_omp.so, bit-identical): the box's CPU share for one GPU
            threads = max(1, min(16, os.cpu_count() or 1))
            threads = ocv.orc.use_openmp(True, threads)
            try:
                mt = [cpu_pass() for _ in range(3)]
                if nf == rig.n:      # the oracle's panorama of the very frames the timed composer ran on: the `parity` key
                    cpu_ref = cmp.compose_panorama(ocv, fr, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend=rig.blend, num_bands=rig.num_bands,
                                                   seam_frames=seams_np, seam_aspect=rig.seam_scale, expos_comp=rig.expos_comp)
            finally:
                ocv.orc.use_openmp(False)
            dtn = min(mt)
            mpix = nf * rig.width * rig.height / 1e6
            cpu_baseline = {"value": round(mpix / dtn, 3), "unit": "MPix/s", "cores": threads, "kind": "port",
                            "sample": f"{nf} of the {rig.n} frames of the same workload through the reference's call sequence (seam-scale warps + compensator feed, warp+mask, apply, mask prep, feed, blend); "
                                      f"OpenMP over rows, fastest of {len(mt)} passes ({dtn:.2f} s); one core: fastest of {len(times)} passes ({dt1:.1f} s); "
                                      f"{sum(times) + sum(mt):.0f} s of CPU time in all",
                            "single_core_value": round(mpix / dt1, 3), "host_cpus": os.cpu_count()}

    # ---- parity: the timed composer's mosaic of frame set 0 against the oracle's (the cpu_baseline leg computed it), and the drop-in sequence's
    if rank == 0 and world == 1 and exchange is None and cpu_ref is not None and rig.dtype == "u8":
        composer.run(frames)
        g_mosaic, g_mask, _ = [u.get() if u is not None else None for u in composer.result()]
        d = np.abs(g_mosaic.astype(np.int16) - cpu_ref.mosaic.astype(np.int16))
        parity = {"against": f"the oracle's panorama of the same {rig.n} frames through the reference's call sequence (cpu_baseline leg)", "pano": list(cpu_ref.pano_roi),
                  "pano_identical": tuple(cpu_ref.pano_roi) == tuple(composer.pano_roi()), "mask_identical": bool(np.array_equal(g_mask, cpu_ref.result_mask)),
                  "max_abs": int(d.max()), "frac_differing": float((d > 0).mean()),
                  "tolerance": "north_star: +-1 LSB (the GAIN_BLOCKS gains are double sums in another order: 1e-9 relative)",
                  "dropin_umat_identical_to_composer": (bool(np.array_equal(dropin_mosaic, g_mosaic)) if dropin_mosaic is not None else None)}
        del d, g_mosaic, g_mask
    if rank == 0:
        out = {
            "metric": "MPix/s warped+blended into final mosaic", "value": round(value, 1), "unit": "MPix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8" if rig.dtype == "u8" else "f32", "data": "synthetic",
            "config": {"workload": workload, "frames_per_gpu": rig.n, "frame": f"{rig.width}x{rig.height}", "warp": rig.warp, "blend": rig.blend,
                       "num_bands": rig.num_bands, "expos_comp": rig.expos_comp, "mask_prep": mask_prep, "pano": list(composer.pano_roi()),
                       "feed_units": len(composer.parts()), "warped_MPix": round(sum(p[1][2] * p[1][3] for p in composer.parts()) / 1e6, 1),
                       "roi_MPix": round(sum(composer.image_roi(i)[2] * composer.image_roi(i)[3] for i in range(rig.n)) / 1e6, 1),
                       "scale_div": args.scale_div, "input_gen_s": round(gen_s, 2), "frame_sets": len(frame_sets), "library": cv._lib.lib_identity(),
                       "frame_set_MB": round(sum(f.nbytes for f in frames_np) / 1e6, 1),
                       # input-independent tables (projection sines / cosines, resize coordinates, dilated seam mask) are a product of the
                       # cameras: built on the composer's first panoramas, reused while the geometry is unchanged.  Every
                       # pixel of every step is computed from that step's frames.  `tables_rebuilt` is the step that rebuilds them every time.
                       "geometry_tables": "per composer (reused across steps); see tables_rebuilt for the step that rebuilds them",
                       "exchange_bytes_rank0": (exchange.plan.bytes_sent(0, 13 if rig.dtype == "f32" else 4) if exchange is not None else 0),
                       "exchange_protocol": (("all-level strips" if exchange.plan.levels else "level-0 strips, pyramids rebuilt by the receiver") if exchange is not None else None)},
            "end_to_end_ms": round(latency_ms * (2 if pipeline is not None else 1), 4), "panoramas_in_flight": 2 if pipeline is not None else depth, "in_flight_2": in_flight_2, "with_pcie": with_pcie,
            "scale_base": scale_base, "tables_rebuilt": tables_rebuilt, "coordinate_planes": coordinate_planes, "arc357": arc357, "dropin_umat": dropin_umat, "parity": parity, "self_check": self_check, "exchange": exchange_report, "roofline": roofline, "cpu_baseline": cpu_baseline, "kernels": kernels,
        }
        print(json.dumps(out))
    if pipeline is not None:
        pipeline.drain()
        sync()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
