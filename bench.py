#!/usr/bin/env python3
"""bench.py -- frames/s of the RAW image chain on MI355X, with the roofline of its dominant kernel
and the CPU baseline beside it.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by
torch.distributed.run with one rank per GPU -- or called plainly, in which case it launches its N ranks itself
(`self_launch`).  It never prints a line whose n_gpus differs from --gpus: fewer devices than ranks (RCCL) or a
WORLD_SIZE that disagrees with --gpus is a non-zero exit.  A step is one pass of the hot path over one batch of
synthetic frames that are already resident in HBM.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): 2448x2048
bayer_rggb8, full chain = debayer + flip(180) + grey-world WB + colour calibration + gamma(k=0.8)
+ vignetting + fisheye undistortion, `--batch` frames per step (default 256).  Frames are
independent units, so ranks shard by camera stream with no data-path collective (weak scaling:
every rank runs its own batch); RCCL is used only for the barrier and the max-over-ranks time.

Other workloads (for development, not the driver's line): --workload chain (configs 2a: the fused
per-pixel kernel alone), config3 (1920x1200 gbrg8, ccc + enhancer), config5 (3840x2160 debayer +
remap).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)

# algorithmic bytes per pixel of each kernel class: compulsory traffic of the
# fused schedule
BYTES_PER_PX = {
    "stats": 1.0,   # Bayer read (grey-world / pca pre-pass)
    "ccc": 0.0,     # reads 4 x 360 x 270 taps per frame: O(1) per frame
    "chain": 4.0,   # 1 B Bayer read + 3 B BGR write
    # SURVEY's ledger prices the remap at 14 B/px (8 B float2 map + 3 B gather + 3 B write) per frame.  The
    # tiled kernel reads a compiled plan (4 B/px) once per LAUNCH instead of the map once per frame, so
    # the compulsory traffic of one launch is 3 + 3 + 4 / frames B/px; the roofline uses that (smaller)
    # figure -- pricing it at 14 would report more than the HBM peak.
    "remap": None,
}


FUSED = {"on": False}  # set by main(): no chain launch was recorded although the workload has a chain -> rip_fused.hip ran


def bytes_per_px(kernel_class, frames_per_launch):
    if kernel_class == "remap":
        # two kernels: 3 B gathered + 3 B written (+ plan).  Chain inside the remap's tiles (memory-rate stage sets, no tap:
        # csrc/rip_fused.hip): 1 B of Bayer read + 3 B written (+ plan) -- there is no intermediate image
        return (4.0 if FUSED["on"] else 6.0) + 4.0 / max(frames_per_launch, 1)
    return BYTES_PER_PX[kernel_class]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per step per GPU (BASELINE config 2: >= 64 resident frames)")
    ap.add_argument("--workload", default="config2", choices=["config2", "chain", "default_chain", "config3", "config5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-probe", action="store_true", help="skip the streaming copy/read microbenchmark")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-run this command under rocprofv3 --pmc for roofline.traffic "
                    "(the committed profiles/pmc_traffic.json figure is reported instead, labelled as such)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--batch-total", type=int, default=0, help="N > 1: frames of the ONE batch rank 0 owns in the end-to-end leg "
                    "(scatter -> every rank processes the shard it received -> checksums compared); 0 = --batch.  BASELINE config 5: 512")
    return ap.parse_args()


def make_frames(width, height, pattern, batch, rank, distinct=4):
    """`batch` synthetic Bayer frames: `distinct` seeded scenes (seed = 1000*camera + frame) plus
    even-offset rolls of them (keeps the Bayer phase, changes every pixel's neighbourhood)."""
    from raw_image_pipeline_amd import synth
    # RIP_BENCH_FRAMES (development only; the driver's line is always "scene"): "flat" = one colour per channel, every lane of a
    # wave looks up the same table entries (no LDS bank conflicts: what the fused chain costs in instructions alone);
    # "uniform" = iid bytes (the worst case for the table lookups)
    kind = os.environ.get("RIP_BENCH_FRAMES", "scene")
    if kind == "flat":
        bgr = np.empty((height, width, 3), np.uint8)
        bgr[...] = (66, 120, 84)
        base = [synth.mosaic(bgr, pattern)]
    else:
        base = [synth.gen_frame(width, height, pattern, seed=1000 * rank + i, kind=kind) for i in range(min(distinct, batch))]
    frames = []
    for i in range(batch):
        f = base[i % len(base)]
        k = i // len(base)
        frames.append(np.roll(f, (2 * k, 4 * k), axis=(0, 1)) if k else f)
    return np.stack(frames)


def configure(pipe, workload, width, height):
    from raw_image_pipeline_amd import synth
    pattern = "bayer_rggb8"
    if workload == "config2":
        synth.configure_full_chain(pipe, width, height, "grey_world")
        stages = "debayer+flip180+grey_world+color_calib+gamma+vignetting+undistort"
    elif workload == "chain":
        synth.configure_full_chain(pipe, width, height, "grey_world")
        pipe.set_white_balance(False)  # gains only matter through the statistics pre-pass
        pipe.set_undistortion(False)
        stages = "debayer+flip180+color_calib+gamma+vignetting (fused per-pixel kernel only)"
    elif workload == "default_chain":
        # the stage set of the reference's three default parameter sets plus the flip and the colour calibration of config 2:
        # everything of the full chain except the vignetting (off in every default: raw_image_pipeline.cpp:123,
        # pipeline_params_example.yaml:27) and the undistortion -- the variant of the fused kernel that runs at the memory rate
        synth.configure_full_chain(pipe, width, height, "grey_world")
        pipe.set_vignetting_correction(False)
        pipe.set_undistortion(False)
        stages = "debayer+flip180+grey_world+color_calib+gamma (no vignetting, no undistortion)"
    elif workload == "config3":
        pattern = "bayer_gbrg8"
        filt, bias = synth.ccc_model()
        pipe.set_ccc_model(filt, bias)
        pipe.set_flip(False)
        pipe.set_white_balance(True)
        pipe.set_white_balance_method("ccc")
        pipe.set_white_balance_saturation_threshold(0.8, 0.2)
        pipe.set_white_balance_temporal_consistency(True)
        pipe.set_color_calibration(False)
        pipe.set_gamma_correction(False)
        pipe.set_vignetting_correction(False)
        pipe.set_color_enhancer(True)
        pipe.set_color_enhancer_saturation_gain(1.2)
        pipe.set_undistortion(False)
        stages = "debayer+ccc(temporal)+color_enhancer"
    else:  # config5
        pipe.set_flip(False)
        pipe.set_white_balance(False)
        pipe.set_color_calibration(False)
        pipe.set_gamma_correction(False)
        pipe.set_vignetting_correction(False)
        pipe.set_color_enhancer(False)
        pipe.set_undistortion(True)
        synth.load_camera(pipe, synth.camera_model(width, height))
        stages = "debayer+undistort"
    return pattern, stages


def cpu_baseline(width, height, pattern, seconds):
    """The oracle (CPU restatement of the reference's OpenCV path) timed on this box's host cores -- SURVEY 8(d):
    two schedules x two thread counts.  `faithful` keeps the reference's passes (one pass per OpenCV call, 4 frame
    copies, vignetting mask rebuilt per frame because W != H); `tight` gives the same results without the redundant
    passes (mask cached).  Single thread: median of 20 frames.  All cores: one frame per thread (the reference's
    one-process-per-camera deployment), whole rounds of `cores` frames until the time budget is spent, median round.
    The headline `value` is the faithful schedule on all cores (what a user of the reference gets from this host)."""
    import statistics
    import threading
    import oracle as O
    from raw_image_pipeline_amd import synth
    O.build()
    cores = os.cpu_count() or 1
    cam = synth.camera_model(width, height)
    newK = O.fisheye_new_camera_matrix(cam["K"], cam["D"], (width, height), cam["R"], 0.0, None, 1.0)
    mx, my = O.fisheye_maps(cam["K"], cam["D"], cam["R"], newK, (width, height))
    mask = O.vignetting_mask(height, width, 1.5, 1e-3, 1e-6)
    frame = synth.gen_frame(width, height, pattern, seed=0, kind="scene")

    def params(faithful):
        prm = O.Params()
        prm.flip_enabled, prm.flip_angle = 1, 180
        prm.wb_enabled, prm.wb_method, prm.wb_bright_thr, prm.wb_dark_thr = 1, 1, 0.8, 0.2
        prm.cc_enabled, prm.cc_available = 1, 1
        for i, v in enumerate(synth.COLOR_MATRIX):
            prm.cc_matrix[i] = v
        prm.gamma_enabled, prm.gamma_k = 1, 0.8
        prm.vig_enabled, prm.vig_scale, prm.vig_a2, prm.vig_a4 = 1, 1.5, 1e-3, 1e-6
        prm.und_enabled = 1
        prm.map_x, prm.map_y = mx.ctypes.data, my.ctypes.data
        prm.map_rows, prm.map_cols = height, width
        prm.reference_schedule = 1 if faithful else 0
        if not faithful:
            prm.vig_mask = mask.ctypes.data
        return prm

    out = {}
    budget = max(seconds, 4.0) / 2.0  # per schedule
    total_frames, total_s = 0, 0.0
    for name, faithful in (("faithful", True), ("tight", False)):
        prm = params(faithful)
        work = lambda: O.pipeline(prm, frame, pattern)
        work()  # warm-up (page faults, table init)
        t_start = time.perf_counter()
        singles = []
        for _ in range(20):
            t1 = time.perf_counter()
            work()
            singles.append(time.perf_counter() - t1)
            if time.perf_counter() - t_start > budget * 0.4 and len(singles) >= 5:
                break
        rounds = []
        while True:
            th = [threading.Thread(target=work) for _ in range(cores)]
            t1 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            rounds.append(time.perf_counter() - t1)
            # at least 5 whole rounds per schedule whatever the budget says (SURVEY 8(d) asks for a distribution, not one
            # sample), at most 20
            if (time.perf_counter() - t_start >= budget and len(rounds) >= 5) or len(rounds) >= 20:
                break
        el = time.perf_counter() - t_start
        total_frames += len(singles) + len(rounds) * cores
        total_s += el
        pct = lambda v, q: float(np.percentile(np.asarray(v), q))
        out[name] = {"value": round(cores / statistics.median(rounds), 3), "single_thread_value": round(1.0 / statistics.median(singles), 3),
                     "cores": cores, "single_thread_reps": len(singles), "all_core_rounds": len(rounds),
                     # frames/s at the 10th / 90th percentile of the per-round (per-frame) times: p10 time = fast rounds
                     "all_core_p10_p90": [round(cores / pct(rounds, 90), 3), round(cores / pct(rounds, 10), 3)],
                     "single_thread_p10_p90": [round(1.0 / pct(singles, 90), 3), round(1.0 / pct(singles, 10), 3)]}
    return {"value": out["faithful"]["value"], "unit": "frames/s", "cores": cores, "kind": "port",
            "single_thread_value": out["faithful"]["single_thread_value"], "faithful": out["faithful"], "tight": out["tight"],
            "sample": "%d frames of the same %dx%d %s full chain through the oracle (CPU restatement of the OpenCV path; OpenCV "
                      "itself is not installable here), reference-faithful and tight schedules, 1 thread (median of <= 20 frames) "
                      "and %d threads x 1 frame (>= 5 rounds per schedule, median round; p10 / p90 beside it), %.1f s" % (total_frames, width, height, pattern, cores, total_s)}


def live_pmc_traffic(args, kernel_class):
    """HBM bytes of one launch of the dominant kernel class, measured NOW: this very command (same workload, same batch, 3
    steps) re-run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, KiB units, x2 on FETCH_SIZE for
    gfx950's 128-byte requests tallied at 64 B, as MI355X_MICROARCH.md prescribes (tools/collect_pmc.py documents the
    calibration) -- plus a third pass with SQ_INSTS_VALU and GRBM_GUI_ACTIVE for the measured VALU issue rate of the dominant
    kernel.  Returns (bytes, valu_record_or_None, None) or (None, None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    valu = None
    if shutil.which("rocprofv3") is None:
        return None, None, "rocprofv3 not on PATH"
    pat = {"stats": "stats_", "chain": "chain_", "remap": "remap_", "ccc": "ccc_"}[kernel_class]
    med = lambda v: sorted(v)[len(v) // 2] if v else None
    total = 0.0
    tmp = tempfile.mkdtemp(prefix="rip_pmc_", dir="/tmp")
    try:
        for counter, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0), ("SQ_INSTS_VALU GRBM_GUI_ACTIVE", None)):
            d = os.path.join(tmp, counter.split()[0])
            cmd = ["rocprofv3", "--pmc"] + counter.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-probe", "--no-pmc",
                   "--workload", args.workload, "--batch", str(args.batch)]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=240, check=False)
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if pat in row["Kernel_Name"] and row["Counter_Name"] in counter.split():
                        per.setdefault((row["Kernel_Name"], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
            if factor is None:
                # issue counters: VALU wave-instructions per SIMD per cycle.  GRBM_GUI_ACTIVE comes back summed over the 8
                # XCDs (tools/collect_pmc_sq.py); 1024 SIMDs; 0.5 = one wave64 instruction per 2 cycles, the SIMD-32 issue
                # ceiling.  Collected for EVERY kernel of the run (the counters are not filtered by class): the dominant one
                # goes into roofline.valu, all of them into roofline.per_kernel.
                every = {}
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if row["Counter_Name"] in counter.split():
                            every.setdefault((row["Kernel_Name"], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
                allk = {}
                for (kn, cn), v in every.items():
                    if cn != "GRBM_GUI_ACTIVE" or not med(v):
                        continue
                    insts = med(every.get((kn, "SQ_INSTS_VALU"), []))
                    cls = next((c for c, pt in (("stats", "stats_"), ("chain", "chain_"), ("remap", "remap_"), ("ccc", "ccc_")) if pt in kn), None)
                    if cls is None or insts is None:
                        continue
                    cycles = med(v) / 8.0
                    rec = {"kernel": kn.split("(")[0][-60:], "valu_wave_instructions": int(insts), "kernel_cycles": int(cycles),
                           "valu_instr_per_simd_cycle": round(insts / 1024.0 / cycles, 4)}
                    if cls not in allk or cycles > allk[cls]["kernel_cycles"]:  # the longest kernel of the class
                        allk[cls] = rec
                if kernel_class in allk:
                    valu = dict(allk[kernel_class], all_classes=allk)
                continue
            if not per:
                return None, None, "rocprofv3 --pmc %s produced no rows (rc %d)" % (counter, r.returncode)
            total += sum(med(v) for v in per.values()) * 1024.0 * factor  # one launch of every kernel of the class
        return int(total), valu, None
    except Exception as e:  # noqa: BLE001 -- the bench line must come out whatever the profiler does
        return None, None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def frame_checksums(images):
    """Two position-sensitive 63-bit sums per frame of a [n, ...] uint8 device tensor (no image leaves its GPU)."""
    import torch
    n = images.shape[0]
    flat = images.reshape(n, -1)
    w = (torch.arange(flat.shape[1], device=flat.device, dtype=torch.int32) % 65521) + 1
    out = []
    for i in range(n):
        x = flat[i].to(torch.int32)
        out.append([int(x.sum(dtype=torch.int64).item()), int((x * w).sum(dtype=torch.int64).item())])
    return out


def end_to_end_leg(args, pipe, frames, pattern, frame_shape, rank, world, backend, barrier):
    """N > 1: ONE batch that originates on rank 0 goes through the whole multi-GPU path (VERDICT round 4 missing-3; BASELINE
    configs 4 / 5): scatter_frames -> every rank processes THE SHARD IT RECEIVED -> per-frame checksums are gathered and
    compared with rank 0 processing the same frame ranges of its own copy.  Frames are independent on these workloads
    (no temporal state), so the two must agree bit for bit.  The per-camera constants are checked on the way: rank 0's
    undistortion maps are broadcast and every rank compares them with the maps its own device built.  Outside the timed
    steady state; its own times are reported."""
    import torch
    import torch.distributed as dist
    from raw_image_pipeline_amd import sharding
    if args.workload == "config3":
        return {"skipped": "the ccc filter carries state from frame to frame: a single-stream batch does not shard by frame ranges (cameras shard instead)"}
    total = args.batch_total if args.batch_total > 0 else args.batch
    sc_dev = "cuda" if backend == "nccl" else "cpu"
    whole = None
    if rank == 0:
        reps = (total + frames.shape[0] - 1) // frames.shape[0]
        whole = torch.cat([torch.roll(frames, shifts=(2 * k, 4 * k), dims=(1, 2)) if k else frames for k in range(reps)])[:total].contiguous()
    barrier()
    t0 = time.perf_counter()
    mine = sharding.scatter_frames(whole.to(sc_dev) if rank == 0 else None, frame_shape, device=sc_dev)
    barrier()
    t_scatter = sharding.max_over_ranks(time.perf_counter() - t0)
    mine = mine.cuda()
    t0 = time.perf_counter()
    got = pipe.apply_device(mine, pattern) if mine.shape[0] else None
    barrier()
    t_process = sharding.max_over_ranks(time.perf_counter() - t0)
    sums = frame_checksums(got) if got is not None else []
    del got, mine
    a, b = sharding.frame_range_of_rank(total, world, rank)
    assert len(sums) == b - a
    # every rank's checksums to rank 0 (all_gather of a padded int64 tensor: ranges differ by at most one frame)
    width_max = (total + world - 1) // world
    pad = torch.zeros((width_max, 2), dtype=torch.int64)
    if sums:
        pad[:len(sums)] = torch.tensor(sums, dtype=torch.int64)
    pad = pad.cuda() if backend == "nccl" else pad
    gathered = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(gathered, pad)
    # constants: rank 0's maps against every rank's own
    consts_equal = None
    if pipe.is_undistortion_enabled():
        mx, my = pipe.get_undistortion_maps()
        own = torch.from_numpy(np.stack([mx, my]))
        ref = own.clone()
        ref = ref.cuda() if backend == "nccl" else ref
        sharding.broadcast_constants(ref, src=0)
        same = 1.0 if torch.equal(ref.cpu(), own) else 0.0
        consts_equal = sharding.sum_over_ranks(same) == float(world)
    rec = None
    if rank == 0:
        mismatched, checked = [], 0
        for r in range(world):
            ra, rb = sharding.frame_range_of_rank(total, world, r)
            if rb <= ra:
                continue
            ref_out = pipe.apply_device(whole[ra:rb].contiguous(), pattern)
            ref_sums = frame_checksums(ref_out)
            del ref_out
            theirs = gathered[r][:rb - ra].cpu().tolist()
            for i, (x, y) in enumerate(zip(theirs, ref_sums)):
                checked += 1
                if list(x) != list(y):
                    mismatched.append(ra + i)
        rec = {"frames_total": total, "frames_per_rank": [sharding.frame_range_of_rank(total, world, r)[1] - sharding.frame_range_of_rank(total, world, r)[0] for r in range(world)],
               "scatter_s": round(t_scatter, 6), "process_s": round(t_process, 6),
               "frames_per_s_end_to_end": round(total / (t_scatter + t_process), 1),
               "frames_checked_against_rank0": checked, "mismatched_frames": mismatched[:16], "results_equal": not mismatched,
               "constants_equal_across_ranks": consts_equal}
    del whole
    return rec


def memory_rate_variant(args):
    """the stage set the reference ships by default (no vignetting) runs the variant of the fused
    kernel without the Lab round trip -- the one north_star's 80 % can physically apply to.  Measured here by running this
    script's `default_chain` workload (same geometry, same batch, its own PMC passes) and reported beside the headline's
    dominant kernel.  Never part of the timed region of the headline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "default_chain", "--steps", "10", "--warmup", "2", "--batch", str(args.batch),
           "--no-cpu-baseline", "--no-hbm-probe"]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600, check=False)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            return {"error": "default_chain run printed no line (rc %d)" % r.returncode}
        j = json.loads(line[-1])
        rf = j["roofline"]
        pk = (rf.get("per_kernel") or {}).get("chain", {})
        ms = rf["kernel_ms_per_step"].get("chain")
        alg = 4.0 * 2448 * 2048 * args.batch
        led = {}
        try:
            with open(os.path.join(ROOT, "profiles", "chain_ledger.json")) as f:
                led = json.load(f).get("default_chain") or {}
        except Exception:
            pass
        return {"workload": j["config"]["workload"], "kernel": "chain_fast_kernel<CC|GAMMA, Q8 gains, 256 threads>",
                "frames_per_s_stats_plus_chain": j["value"], "chain_ms_per_%d_frames" % args.batch: ms,
                "achieved": round(alg / (ms * 1e-3) / 1e9, 1) if ms else None, "unit": "GB/s",
                "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms else None,
                "traffic": rf.get("traffic") if rf.get("kernel") == "chain" else None,
                "valu_instr_per_simd_cycle": pk.get("valu_instr_per_simd_cycle"), "valu_floor_frac": pk.get("valu_floor_frac"),
                "ledger_valu_instr_per_item": led.get("valu_instr_per_item"), "ledger_issue_cycles_per_item": led.get("valu_cycles_per_item"),
                "stats_ms": rf["kernel_ms_per_step"].get("stats")}
    except Exception as e:  # noqa: BLE001 -- never lose the headline over the side measurement
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def remap_components(args):
    """roofline.per_kernel.remap.components: the ring remap with one part dropped at a time --
    timing-only builds of the same kernel (raw_image_pipeline_amd/variants/exp.so: -DRIP_EXPERIMENTS, wrong pixels), all in ONE
    process on one output allocation (tools/probes/remap_exp_probe.py), ms per launch of args.batch frames.  The parts do not
    add up to the whole and do not hide behind each other either: loads alone + stores alone ~ the complete kernel -- the memory
    system serves this kernel's read and write streams one after the other (EXPERIMENTS.md round 6)."""
    exp = os.path.join(ROOT, "raw_image_pipeline_amd", "variants", "exp.so")
    probe = os.path.join(ROOT, "tools", "probes", "remap_exp_probe.py")
    if not (os.path.exists(exp) and os.path.exists(probe)):
        return None
    names = {0: "complete", 8: "without_gather_reads_and_arithmetic", 16: "without_stores", 32: "without_source_loads",
             24: "source_loads_only", 40: "stores_only", 64: "without_plan_words"}
    try:
        r = subprocess.run([sys.executable, probe, "--masks", ",".join(str(m) for m in names), "--rounds", "2", "--steps", "4", "--batch", str(args.batch)],
                           env=dict(os.environ, RIP_LIBRARY=exp), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=240)
    except Exception:
        return None
    out = {}
    for l in r.stdout.splitlines():
        if not l.startswith("mask"):
            continue
        try:
            m = int(l.split()[1])
            ms = json.loads(l[l.index("{"):].replace("'", '"'))
            out[names[m]] = ms.get("remap")
        except Exception:
            continue
    if "complete" not in out:
        return None
    out["unit"] = "ms per launch of %d frames; timing-only builds of remap_ring_kernel (wrong pixels), one process, medians of 2 x 4 launches" % args.batch
    return out


def baseline_metric():
    """BASELINE.json's metric string, verbatim."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "frames/sec at 2448\u00d72048 bayer_rggb8 full chain; achieved HBM GB/s vs roofline"


def committed_single_gpu_value(workload):
    """The newest committed N = 1 figure of this workload: the driver's BENCH_rNN.json (config2 only) or the round's bench
    lines under profiles/.  {"value", "source"} or None."""
    import glob
    import re
    best = None
    if workload == "config2":
        for f in sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json"))):
            try:
                d = json.load(open(f)).get("parsed") or {}
                if d.get("n_gpus") == 1 and d.get("value"):
                    best = {"value": float(d["value"]), "source": os.path.basename(f)}
            except Exception:
                pass
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_lines.jsonl"))):
        try:
            for line in open(f):
                d = json.loads(line)
                if d.get("n_gpus") == 1 and d.get("config", {}).get("name") == workload and d.get("value"):
                    rnd = int(re.search(r"r(\d+)_", os.path.basename(f)).group(1))
                    if best is None or not best["source"].startswith("BENCH") or rnd >= int(re.search(r"r(\d+)", best["source"]).group(1)):
                        best = {"value": float(d["value"]), "source": "profiles/" + os.path.basename(f)}
        except Exception:
            pass
    return best


def hbm_probe(pipe, torch, nbytes=1 << 30, reps=10):
    """What this box's memory system delivers to plain streaming kernels, GB/s of bytes moved (read + written), best of
    `reps` launches each: the library's own hand-written kernels (rip_debug_hbm_probe, csrc/rip_probe.hip: 16-byte copy, read
    (four loads in flight per lane; eight with the non-temporal hint), fill, the chain's 1 : 3 expand with ordinary and non-temporal stores, a 12-byte-lane copy) -- and, for continuity with the
    round-1..3 lines, torch's elementwise copy_ and sum, which run 15-20 % below them."""
    res = {}
    for kind in ("copy", "read", "read_nt", "fill", "expand13", "expand13_nt", "expand13_wide", "expand13_wide_nt", "expand13_coalesced", "copy12"):
        try:
            res[kind + "_GBps"] = round(pipe.hbm_probe(kind, nbytes, reps), 1)
        except Exception as e:  # noqa: BLE001 -- the bench line must come out whatever a probe does
            res[kind + "_GBps"] = None
            res.setdefault("errors", []).append("%s: %s" % (kind, str(e)[:120]))
    src = torch.empty(nbytes // 4, dtype=torch.int32, device="cuda").fill_(1)
    dst = torch.empty_like(src)
    for name, fn, moved in (("torch_copy_GBps", lambda: dst.copy_(src), 2 * nbytes), ("torch_read_GBps", lambda: src.view(torch.int64).sum(), nbytes)):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = round(moved * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    res["source"] = "rip_debug_hbm_probe (hand-written, best of %d launches over %d MiB); torch_* = torch elementwise kernels, average of %d" % (reps, nbytes >> 20, reps)
    return res


def self_launch(n):
    """Re-runs this very command under `torch.distributed.run` with one process per GPU on this node (the launch line of
    the driver's contract; the reference's counterpart is one node per camera, raw_image_pipeline_node.launch:85).
    Returns the launcher's exit code; rank 0 of the children prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    # RIP_BENCH_BACKEND=gloo + fewer GPUs than ranks: rehearsal of the multi-rank path on a 1-GPU box (ranks share
    # the device); the driver's runs use one rank per GPU over RCCL
    backend = os.environ.get("RIP_BENCH_BACKEND", "nccl")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the pipeline has no CPU execution path")
    if backend == "nccl" and torch.cuda.device_count() < args.gpus:
        # never a silent single-GPU number under an N-GPU label
        raise SystemExit("--gpus %d but this node shows %d GPU(s): one rank per GPU over RCCL needs %d devices "
                         "(RIP_BENCH_BACKEND=gloo rehearses the multi-rank path with ranks sharing a device)"
                         % (args.gpus, torch.cuda.device_count(), args.gpus))
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` called plainly: launch the N ranks ourselves, exactly as the driver would
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the line's n_gpus must be the number of ranks that ran" % (args.gpus, world))
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    # RIP_DIST_FORCE=1 at N = 1: a single-rank communicator, and every multi-rank code path below runs over it (collectives on
    # device tensors, the scatter's broadcast, the end-to-end leg) -- the rehearsal of the RCCL path a 1-GPU box allows; its
    # line carries the `multi` / `scatter` records and is not a benchmark line
    forced = world == 1 and os.environ.get("RIP_DIST_FORCE", "") == "1"
    dist_on = world > 1 or forced
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group(backend, rank=rank, world_size=world)

    from raw_image_pipeline_amd import RawImagePipeline

    dims = {"config2": (2448, 2048), "chain": (2448, 2048), "default_chain": (2448, 2048), "config3": (1920, 1200), "config5": (3840, 2160)}
    width, height = dims[args.workload]
    pipe = RawImagePipeline(False, "", "", "", device=device_index)
    pipe.set_stream(torch.cuda.current_stream())
    pattern, stages = configure(pipe, args.workload, width, height)

    frames = torch.from_numpy(make_frames(width, height, pattern, args.batch, rank)).cuda()
    orows, ocols, ocn, _ = pipe.query_output(height, width, 1, pattern)
    out = torch.empty((args.batch, orows, ocols, ocn), dtype=torch.uint8, device="cuda")

    def step():
        pipe.apply_device(frames, pattern, out=out)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    pipe.profile_begin(64 * args.steps + 8)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = pipe.profile_end()
    from raw_image_pipeline_amd import sharding
    own_elapsed = elapsed
    elapsed = sharding.max_over_ranks(elapsed)  # the job is as slow as its slowest rank
    multi = None
    if dist_on:
        # self-verifying multi-GPU record: every rank's own rate, what the communicator says about itself, and (rank 0,
        # below) the share of the committed single-GPU rate each GPU retains
        multi = {"per_rank_frames_per_s": [round(v, 1) for v in sharding.gather_over_ranks(args.batch * args.steps / own_elapsed)],
                 "communicator": sharding.communicator_census()}
        multi["rccl_ranks"] = multi["communicator"]["ranks_counted"] if multi["communicator"]["backend"] == "nccl" else None

    total_frames = args.batch * args.steps * world
    fps = total_frames / elapsed

    # roofline of the dominant kernel class, from the HIP events recorded around its launches
    FUSED["on"] = prof.get("remap", (0, 0))[1] > 0 and prof.get("chain", (0, 0))[1] == 0
    px = width * height
    # the streaming class with the largest total; the ccc estimator (O(1) bytes per frame) is listed in
    # kernel_ms_per_step but has no HBM roofline
    dom = max((k for k in prof if bytes_per_px(k, args.batch) > 0), key=lambda k: prof[k][0])
    dom_ms, dom_n = prof[dom]
    out_px = orows * ocols if dom == "remap" else px
    per_launch_bytes = bytes_per_px(dom, args.batch) * out_px * args.batch
    avg_s = (dom_ms / max(dom_n, 1)) * 1e-3
    achieved = per_launch_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            with open(pmc_path) as f:
                pmc = json.load(f)
            traffic = pmc.get(args.workload, {}).get(dom)
            if traffic is not None:  # bytes of one launch at the batch size of the PMC run: scale to this run's
                traffic = int(traffic * args.batch / float(pmc.get("frames_per_launch", 64)))
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": "profiles/pmc_traffic.json: rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this command (tools/collect_pmc.py), rescaled to this batch" if traffic is not None else None,
                "algorithmic_bytes_per_launch": int(per_launch_bytes), "avg_launch_ms": round(avg_s * 1e3, 4),
                "launches": dom_n,
                "kernel_ms_per_step": {k: round(v[0] / max(args.steps, 1), 4) for k, v in prof.items()},
                # algorithmic GB/s of every streaming class (same definition as `achieved`), for the non-dominant kernels
                "kernel_GBps": {k: round(bytes_per_px(k, args.batch) * (orows * ocols if k == "remap" else px) * args.batch /
                                         (v[0] / max(v[1], 1) * 1e-3) / 1e9, 1)
                                for k, v in prof.items() if v[0] > 0 and bytes_per_px(k, args.batch) > 0}}

    # whole-step algorithmic traffic (SURVEY 8(d) ledger as this build schedules it; SURVEY's 19 B/px prices the
    # float2 map at 8 B/px per frame and an intermediate round trip the tiled remap does not make)
    ledger = {k: bytes_per_px(k, args.batch) * (orows * ocols if k == "remap" else px) for k, v in prof.items() if v[0] > 0}
    step_bytes = sum(ledger.values()) * args.batch
    roofline["step"] = {"algorithmic_bytes_per_px": {k: round(v / float(px), 3) for k, v in ledger.items()},
                        "algorithmic_bytes_per_step": int(step_bytes), "survey_bytes_per_px": 19.0 if args.workload == "config2" else None,
                        "achieved": round(step_bytes / (elapsed / args.steps) / 1e9, 1), "unit": "GB/s",
                        "frac": round(step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
    led_path = os.path.join(ROOT, "profiles", "chain_ledger.json")
    if dom == "chain" and os.path.exists(led_path):
        # The fused chain with the Lab round trip sits on VALU issue and LDS bank conflicts, not on HBM
        # (profiles/r02_pmc_sq_summary.txt): priced from the committed ISA ledger of the launched variant
        # (tools/chain_ledger.py: executed VALU issue cycles per 8-pixel item, measured per-class costs).
        try:
            with open(led_path) as f:
                led = json.load(f).get(args.workload)
            if led:
                items_per_s = px / 8.0 / 64.0 * args.batch / avg_s  # wave-items per second
                peak = 1024 * led["clock_GHz"] * 1e9              # SIMD issue cycles per second on 256 CUs
                ach = items_per_s * led["valu_cycles_per_item"]
                # The ledger is a MODEL (static instruction counts priced with per-class issue costs measured on dependent
                # pairs of one opcode): it over-prices a mixed stream by 10-35 % and is reported as such, uncapped.  The
                # MEASURED figure is `measured_frac` below, filled in from the live PMC pass (SQ_INSTS_VALU per SIMD per
                # cycle over the SIMD-32 ceiling of 0.5).
                roofline["valu"] = {"model": {"instr_per_wave_item": led["valu_instr_per_item"], "issue_cycles_per_wave_item": led["valu_cycles_per_item"],
                                              "lds_cycles_per_wave_item": led["lds_cycles_per_item"], "priced_G_issue_cycles_per_s": round(ach / 1e9, 1),
                                              "available_G_issue_cycles_per_s": round(peak / 1e9, 1), "priced_over_available": round(ach / peak, 4),
                                              "valu_instr_per_simd_cycle": round(items_per_s * led["valu_instr_per_item"] / peak, 4),
                                              "lds_frac": round(items_per_s * led["lds_cycles_per_item"] * 4 / peak, 4),
                                              "source": "profiles/chain_ledger.json (tools/chain_ledger.py)"},
                                    "measured_frac": None, "measured": None, "peak_instr_per_simd_cycle": 0.5}
                # `bound` stays "hbm": it names the roofline `peak` / `frac` are priced against (the contract's two values are
                # "hbm" and "mfma").  What actually limits this kernel goes into `limiter`, with `valu_floor_frac` as its ceiling.
                roofline["limiter"] = "valu+lds"
        except Exception:
            pass

    scatter = None
    if dist_on:
        # The only data movement between ranks on this path: the frame scatter when a batch originates on one rank.
        # Timed outside the steady state (it is bounded by the source GPU's xGMI egress) and reported apart.
        # It runs LAST and under a watchdog: the steady-state numbers above are complete at this point, so a point-to-point
        # transfer that does not come back within 240 s must not cost the run its line -- every rank then leaves on its own,
        # rank 0 after printing the line with the scatter marked as timed out.
        import threading
        state = {"done": False}

        def bail_out():
            if state["done"]:
                return
            if rank == 0:
                print(json.dumps(dict(result_base(), scatter={"ranks": world, "backend": backend, "error": "scatter / end-to-end leg did not complete within 240 s"})),
                      flush=True)
            os._exit(0)

        def result_base():
            return {"metric": baseline_metric(), "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                    "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                    "config": {"workload": "%dx%d %s, %s" % (width, height, pattern, stages), "frames_per_step_per_gpu": args.batch,
                               "sharding": "one camera stream per GPU, no data-path collective", "name": args.workload},
                    "roofline": roofline}

        timer = threading.Timer(240.0, bail_out)
        timer.daemon = True
        timer.start()
        try:
            sc_dev = "cuda" if backend == "nccl" else "cpu"  # gloo rehearsal: point-to-point needs host tensors
            src_batch = (frames if backend == "nccl" else frames.cpu()) if rank == 0 else None
            barrier()
            t_sc = time.perf_counter()
            mine = sharding.scatter_frames(src_batch, (height, width), device=sc_dev)
            barrier()
            t_sc = sharding.max_over_ranks(time.perf_counter() - t_sc)
            a, b = sharding.frame_range_of_rank(args.batch, world, 1 if world > 1 else 0)
            scatter = {"ranks": world, "backend": backend, "frames": args.batch, "frames_per_destination": b - a,
                       "bytes_per_destination": (b - a) * height * width, "seconds": round(t_sc, 6),
                       "GBps_per_destination": round((b - a) * height * width / t_sc / 1e9, 3),
                       "GBps_source_egress": round((args.batch - (b - a)) * height * width / t_sc / 1e9, 3)}
            del mine
            scatter["end_to_end"] = end_to_end_leg(args, pipe, frames, pattern, (height, width), rank, world, backend, barrier)
        except Exception as e:  # noqa: BLE001 -- report, never lose the line
            scatter = {"ranks": world, "backend": backend, "error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        state["done"] = True
        timer.cancel()
    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return
    if world == 1 and not args.no_pmc:
        live, valu_live, why = live_pmc_traffic(args, dom)
        if valu_live is not None:
            allk = valu_live.pop("all_classes", {})
            v = roofline.setdefault("valu", {"model": None, "peak_instr_per_simd_cycle": 0.5})
            v["measured"] = dict(valu_live, source="this run: rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE, median launch of the dominant kernel")
            v["measured_frac"] = round(valu_live["valu_instr_per_simd_cycle"] / 0.5, 4)
            # The instruction floor: the time the launch would take if every VALU instruction it
            # EXECUTES (counted by the hardware, not by a listing) issued at the SIMD's full rate of one wave64 instruction per
            # 2 cycles with nothing else in the way -- no quarter-rate opcodes, no LDS, no memory.  floor_ms = instructions /
            # (1024 SIMDs x 0.5) / clock, the clock taken from the same launch (kernel cycles / measured duration).
            # valu_floor_frac is what `frac` would be at that floor: the ceiling this instruction stream puts on the kernel's
            # HBM roofline fraction, below 1 whenever the kernel cannot be HBM-bound by construction.
            per_kernel = {}
            for cls, rec in allk.items():
                if cls not in prof or not prof[cls][1] or bytes_per_px(cls, args.batch) <= 0:
                    continue
                ms = prof[cls][0] / prof[cls][1]
                floor_ms = ms * (rec["valu_wave_instructions"] / 512.0) / rec["kernel_cycles"]
                alg = bytes_per_px(cls, args.batch) * (orows * ocols if cls == "remap" else px) * args.batch
                per_kernel[cls] = {"avg_launch_ms": round(ms, 4), "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "valu_instr_per_simd_cycle": rec["valu_instr_per_simd_cycle"],
                                   "valu_floor_ms": round(floor_ms, 4),
                                   "valu_floor_frac": round(min(1.0, alg / (floor_ms * 1e-3) / 1e9 / HBM_PEAK_GBS), 4),
                                   "clock_GHz": round(rec["kernel_cycles"] / (ms * 1e-3) / 1e9, 3)}
            roofline["per_kernel"] = per_kernel
            if dom in per_kernel:
                roofline["valu_floor_frac"] = per_kernel[dom]["valu_floor_frac"]
                roofline["valu_floor_ms"] = per_kernel[dom]["valu_floor_ms"]
        if live is not None:
            roofline["traffic"] = live
            roofline["traffic_source"] = ("measured in this run: the same command re-run under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE "
                                          "(separate passes, KiB, x2 on FETCH_SIZE for gfx950), median launch of the %s kernels" % dom)
        elif roofline.get("traffic_source"):
            roofline["traffic_source"] += "; live PMC pass unavailable (%s)" % why
    if world == 1 and not args.no_pmc and args.workload == "config2":
        roofline["memory_rate_variant"] = memory_rate_variant(args)
        comp = remap_components(args)
        if comp is not None:
            roofline.setdefault("per_kernel", {}).setdefault("remap", {})["components"] = comp
    if world == 1 and not args.no_hbm_probe:
        # what this box's HBM actually delivers to a plain streaming kernel, beside the 8 TB/s spec
        del out
        probe = hbm_probe(pipe, torch)
        # the access shape that bounds each class: the chain is a 1 : 3 expand (non-temporal stores when nothing reads the image
        # again), the remap gathers 3 B and writes 3 B per pixel (12-byte lanes), the statistics pre-pass only reads
        # (chain inside the remap's tiles: 1 B of Bayer in, 3 B out per pixel -- the expand again)
        shape = {"chain": "expand13_nt_GBps" if not pipe.is_undistortion_enabled() else "expand13_GBps",
                 "remap": "expand13_GBps" if FUSED["on"] else "copy12_GBps",
                 # the better of the two read kernels (four loads in flight per lane / eight non-temporal ones)
                 "stats": "read_nt_GBps" if (probe.get("read_nt_GBps") or 0) > (probe.get("read_GBps") or 0) else "read_GBps"}
        ceil = probe.get(shape.get(dom, "copy_GBps")) or probe.get("copy_GBps")
        roofline["empirical"] = dict(probe, shape_of_dominant_kernel=shape.get(dom, "copy_GBps"),
                                     frac_of_shape=round(achieved / ceil, 4) if ceil else None,
                                     frac_of_copy=round(achieved / probe["copy_GBps"], 4) if probe.get("copy_GBps") else None,
                                     frac_of_read=round(achieved / probe["read_GBps"], 4) if probe.get("read_GBps") else None)
        # every streaming class against its own shape's measured rate, on ACTUAL bytes where the PMC passes have them
        if roofline.get("kernel_GBps"):
            roofline["empirical"]["per_kernel_frac_of_shape"] = {
                k: round(v / probe[shape[k]], 4) for k, v in roofline["kernel_GBps"].items() if k in shape and probe.get(shape[k])}
    result = {
        "metric": baseline_metric(),
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%dx%d %s, %s" % (width, height, pattern, stages), "frames_per_step_per_gpu": args.batch,
                   "sharding": "one camera stream per GPU, no data-path collective", "name": args.workload,
                   "chain_inside_remap_tiles": FUSED["on"]},
        "roofline": roofline,
    }
    if scatter is not None:
        result["scatter"] = scatter
    if multi is not None:
        n1 = committed_single_gpu_value(args.workload)
        multi["single_gpu_reference"] = n1
        multi["retained_vs_n1"] = round(fps / world / n1["value"], 4) if n1 else None
        result.update(multi)
    if not args.no_cpu_baseline and world == 1 and args.workload == "config2":
        result["cpu_baseline"] = cpu_baseline(width, height, pattern, args.cpu_seconds)
    elif world == 1:
        result["cpu_baseline"] = None
    print(json.dumps(result))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
