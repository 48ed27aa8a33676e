"""BASELINE.json configs[3] and configs[4] on ONE GPU, and the multi-rank bench rehearsal.

configs[3] ("8-camera Alphasense rig, 8 concurrent 2448x2048 streams") shards cameras over GPUs with no data-path
collective, so its correctness content is: eight independent handles (own tables, maps, mask plane, ccc Kalman state),
each on its own HIP stream, running concurrently, every frame equal to the oracle.  That runs on one device.
configs[4] (3840x2160, 512-frame batch over 8 GPUs = 64 resident frames per GPU) is exercised at its per-GPU batch."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from raw_image_pipeline_amd import synth

from helpers import assert_images_equal, cfg, configure, oracle_run

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config4_rig_of_8_cameras_2448x2048_on_one_device(rip_lib, oracle):
    """Eight CameraStreams (the ROS node's counterpart: one handle + one HIP stream per camera, camera c -> device
    c mod n_devices, here n_devices = 1), full chain at 2448x2048, different parameters per camera, two of them on the
    ccc estimator with temporal consistency and different tint drifts.  All batches are enqueued before any
    synchronisation.  Every frame of every camera must equal the oracle run with that camera's parameters and -- for
    ccc -- that camera's own filter state: a leak of tables, mask, maps or Kalman state between handles shows up."""
    import torch
    from raw_image_pipeline_amd.frontend import CameraRig
    w, h, ncam, nframes = 2448, 2048, 8, 2
    filt, bias = synth.ccc_model()
    rig = CameraRig([{"output_prefix": "/alphasense/cam%d" % c} for c in range(ncam)], n_devices=1, ccc_model=(filt, bias))
    assert len({s.cuda_stream for s in rig.hip_streams}) == ncam
    cams = []
    for c in range(ncam):
        ccc = c in (2, 5)
        conf = cfg(flip=True, flip_angle=180 if c % 2 == 0 else 0, wb=True, wb_method="ccc" if ccc else ("grey_world" if c % 3 else "pca"),
                   wb_temporal=ccc, cc=True, gamma=True, gamma_k=0.7 + 0.05 * c, vig=True, vig_params=(1.5 - 0.1 * c, 1e-3, 1e-6),
                   undistort=True, cam=synth.camera_model(w, h), balance=0.1 * (c % 3))
        configure(rig.streams[c].pipe, conf)
        if ccc:
            rig.streams[c].pipe.set_ccc_kalman_model(1.0, 10.0)
            rig.streams[c].reset_white_balance()
        drift = 0.04 if c == 2 else -0.05
        frames = np.stack([synth.gen_frame(w, h, "bayer_rggb8", seed=1000 * c + i, kind="scene",
                                           tint=(0.70 + drift * i, 1.0, 0.55 + 0.01 * c)) for i in range(nframes)])
        cams.append((conf, frames, ccc))
    batches = [torch.from_numpy(f).cuda() for _, f, _ in cams]
    torch.cuda.synchronize()
    outs = rig.process_resident(batches, ["bayer_rggb8"] * ncam)  # asynchronous: eight streams in flight
    rig.synchronize()
    info = {c: rig.streams[c].pipe.get_white_balance_info(nframes) for c in (2, 5)}
    for c, (conf, frames, ccc) in enumerate(cams):
        occ = None
        if ccc:
            occ = oracle.CCC(filt, bias)
            occ.set_kalman_model(1.0, 10.0)
        got = outs[c].cpu().numpy()
        for i in range(nframes):
            ref, _ = oracle_run(oracle, conf, frames[i], "bayer_rggb8", ccc=occ)
            assert_images_equal(got[i], ref, "camera %d frame %d" % (c, i))
    # the two ccc cameras saw opposite drifts: their filtered (u, v) tracks must differ (state is per handle)
    assert not np.array_equal(info[2][:, 6:8], info[5][:, 6:8])


def test_config5_batch_of_64_resident_4k_frames(gpu_pipe, oracle):
    """BASELINE configs[4] at its per-GPU share: 64 resident 3840x2160 rggb8 frames through apply_device, debayer +
    undistortion.  Frame 0 is checked against the oracle; every frame must equal the single-frame result of its source
    frame (the batch kernels walk frames innermost and split the batch into groups: no cross-talk, no missed frame)."""
    import torch
    w, h, n, distinct = 3840, 2160, 64, 4
    c = cfg(undistort=True, cam=synth.camera_model(w, h))
    configure(gpu_pipe, c)
    base = [synth.gen_frame(w, h, "bayer_rggb8", seed=5 + i, kind="scene" if i % 2 == 0 else "uniform") for i in range(distinct)]
    singles = [gpu_pipe.process(f, "bayer_rggb8") for f in base]
    ref, _ = oracle_run(oracle, c, base[0], "bayer_rggb8")
    assert_images_equal(singles[0], ref, "config5 single frame vs oracle")
    dev = torch.from_numpy(np.stack(base)).cuda()
    order = (torch.arange(n, device="cuda") * 7 + 3) % distinct  # not a plain repeat pattern
    batch = dev[order].contiguous()
    out = gpu_pipe.apply_device(batch, "bayer_rggb8")
    torch.cuda.synchronize()
    assert out.shape == (n, h, w, 3)
    expect = torch.from_numpy(np.stack(singles)).cuda()
    order = order.cpu().numpy()
    for i in range(n):
        assert torch.equal(out[i], expect[order[i]]), "batch frame %d (source %d) differs from its single-frame result" % (i, order[i])


def test_config2_batch_of_64_resident_frames_full_chain(gpu_pipe, oracle):
    """BASELINE configs[1] with the 64 resident frames it names: every frame of the batch equals the single-frame
    result (frame 0 also the oracle), taps included."""
    import torch
    from helpers import cfg as _cfg
    w, h, n, distinct = 2448, 2048, 64, 4
    c = _cfg(flip=True, flip_angle=180, wb=True, wb_method="grey_world", cc=True, gamma=True, gamma_k=0.8, vig=True, undistort=True,
             cam=synth.camera_model(w, h))
    configure(gpu_pipe, c)
    base = [synth.gen_frame(w, h, "bayer_rggb8", seed=300 + i, kind="scene", tint=(0.6 + 0.05 * i, 1.0, 0.5)) for i in range(distinct)]
    singles = [gpu_pipe.process(f, "bayer_rggb8") for f in base]
    ref, _ = oracle_run(oracle, c, base[0], "bayer_rggb8")
    assert_images_equal(singles[0], ref, "config2 single frame vs oracle")
    dev = torch.from_numpy(np.stack(base)).cuda()
    order = (torch.arange(n, device="cuda") * 5 + 1) % distinct
    out = gpu_pipe.apply_device(dev[order].contiguous(), "bayer_rggb8")
    torch.cuda.synchronize()
    expect = torch.from_numpy(np.stack(singles)).cuda()
    order = order.cpu().numpy()
    for i in range(n):
        assert torch.equal(out[i], expect[order[i]]), "batch frame %d differs" % i


def test_bench_two_rank_rehearsal_prints_one_json_line():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), rehearsed on a 1-GPU box
    with RIP_BENCH_BACKEND=gloo (both ranks share device 0): exactly one JSON line, from rank 0, with n_gpus = 2, the
    whole-job frame count and the scatter record of the N > 1 path."""
    env = dict(os.environ, RIP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--no-cpu-baseline", "--no-hbm-probe", "--no-pmc"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak"
    assert j["config"]["frames_per_step_per_gpu"] == 8
    assert abs(j["value"] - 2 * 8 * 2 / (j["ms_per_step"] * 2 * 1e-3)) / j["value"] < 1e-3  # whole-job frames / max-over-ranks time
    assert j["scatter"]["ranks"] == 2 and j["scatter"]["frames_per_destination"] > 0 and j["scatter"]["GBps_per_destination"] > 0
    # the end-to-end leg: rank 0's batch scattered, every rank processed what it RECEIVED, checksums equal rank 0's own
    e = j["scatter"]["end_to_end"]
    assert e["frames_total"] == 8 and e["frames_per_rank"] == [4, 4] and e["frames_checked_against_rank0"] == 8
    assert e["results_equal"] is True and e["mismatched_frames"] == [] and e["constants_equal_across_ranks"] is True
    assert e["frames_per_s_end_to_end"] > 0


def test_bench_config5_end_to_end_leg_with_a_batch_total(rip_lib):
    """BASELINE configs[4] through the N > 1 path (gloo rehearsal, two ranks on one device): `--workload config5 --batch-total 12`
    -- rank 0 owns 12 frames of 3840x2160, each rank receives 6, processes them with the chain inside the remap's tiles and
    reports checksums that must equal rank 0 processing the same ranges."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(RIP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "config5", "--steps", "1", "--warmup", "1", "--batch", "4",
           "--batch-total", "12", "--no-cpu-baseline", "--no-hbm-probe", "--no-pmc"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    e = j["scatter"]["end_to_end"]
    assert e["frames_total"] == 12 and e["frames_per_rank"] == [6, 6] and e["results_equal"] is True and e["constants_equal_across_ranks"] is True


def test_bench_gpus_2_called_plainly_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no RANK in the environment re-runs itself under
    torch.distributed.run (gloo rehearsal: both ranks share device 0) and the line says what ran: n_gpus = 2, two ranks counted
    by a collective over the communicator, two per-rank rates."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(RIP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--no-cpu-baseline", "--no-hbm-probe", "--no-pmc"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2
    assert j["communicator"]["ranks_counted"] == 2 and j["communicator"]["world_size"] == 2
    assert len(j["per_rank_frames_per_s"]) == 2 and all(v > 0 for v in j["per_rank_frames_per_s"])


def test_bench_single_rank_rccl_communicator_runs_every_multi_rank_code_path():
    """every N > 1 rehearsal above goes over gloo with host tensors, so the RCCL path
    (device tensors through `nccl`) had never executed before the driver's 8-GPU run.  RIP_DIST_FORCE=1 makes
    `bench.py --gpus 1` open a ONE-rank `nccl` communicator and run the whole multi-rank code over it: barrier,
    max-over-ranks (all_reduce f64), per-rank rates (all_gather f64), communicator census (all_reduce int64), the scatter's
    size broadcast (int64) and own-slice copy of uint8 device frames, the end-to-end leg (processing the received shard,
    checksums through the padded int64 all_gather, maps through broadcast_constants on device tensors).  It cannot show
    scaling and does not reach isend / recv (one rank has nobody to send to); it does catch a dtype / device / API mistake."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RIP_BENCH_BACKEND")}
    env.update(RIP_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--no-cpu-baseline", "--no-hbm-probe", "--no-pmc"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1
    c = j["communicator"]
    assert c["backend"] == "nccl" and c["world_size"] == 1 and c["ranks_counted"] == 1 and c["devices"] == [0], c
    assert j["rccl_ranks"] == 1 and len(j["per_rank_frames_per_s"]) == 1 and j["per_rank_frames_per_s"][0] > 0
    sc = j["scatter"]
    assert "error" not in sc, sc
    assert sc["backend"] == "nccl" and sc["ranks"] == 1 and sc["frames_per_destination"] == 8
    e = sc["end_to_end"]
    assert e["frames_total"] == 8 and e["frames_per_rank"] == [8] and e["frames_checked_against_rank0"] == 8
    assert e["results_equal"] is True and e["constants_equal_across_ranks"] is True


def test_bench_refuses_more_gpus_than_the_node_has():
    """`--gpus N` over RCCL on a node with fewer than N devices exits non-zero and prints no line -- never a single-GPU
    number under an N-GPU label."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RIP_BENCH_BACKEND")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--batch", "2",
           "--no-cpu-baseline", "--no-hbm-probe", "--no-pmc"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], r.stdout
    assert "--gpus %d" % n in r.stderr
    # and a launcher environment that disagrees with --gpus is refused as well (here: WORLD_SIZE=1 under --gpus 2, gloo)
    env2 = dict(env, RIP_BENCH_BACKEND="gloo", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run(cmd[:3] + ["2"] + cmd[4:], env=env2, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_config3_sequence_of_64_frames_1920x1200_tracks_the_drifting_tint(gpu_pipe, oracle):
    """BASELINE configs[2] exactly as SURVEY 8(d) writes it: 64 frames of 1920x1200 bayer_gbrg8 whose tint drifts r 0.70 ->
    0.80, ccc white balance with temporal consistency, colour enhancer 1.2.  Once as ONE resident batch (the estimator's
    sequential Kalman step on the device) and once as 64 single calls; in both, for every frame, the RAW argmax of the
    response, the FILTERED (u, v) the gains are taken at and the gains themselves must equal the oracle's -- the sequence,
    not only the pixels (convolutional_color_constancy.cpp:273-381) -- and every output frame must equal the oracle's."""
    import torch
    w, h, n = 1920, 1200, 64
    filt, bias = synth.ccc_model()
    c = cfg(wb=True, wb_method="ccc", wb_bright=0.8, wb_dark=0.2, wb_temporal=True, ce=True, ce_sat=1.2)
    frames = np.stack([synth.gen_frame(w, h, "bayer_gbrg8", seed=6400 + i, kind="scene", tint=(0.70 + 0.10 * i / (n - 1), 1.0, 0.55))
                       for i in range(n)])
    # the oracle's track and images: one filter state walked through the 64 frames
    occ_track = oracle.CCC(filt, bias)
    occ_track.set_thresholds(0.8, 0.2)
    occ_track.set_temporal_consistency(True)
    occ_track.set_kalman_model(1.0, 10.0)
    occ_pixels = oracle.CCC(filt, bias)
    occ_pixels.set_kalman_model(1.0, 10.0)

    def oracle_pass():
        track, gains, refs = [], [], []
        for i in range(n):
            _, info, g = occ_track.balance(oracle.debayer(frames[i], "bayer_gbrg8"))
            track.append(info)
            gains.append(g)
            refs.append(oracle_run(oracle, c, frames[i], "bayer_gbrg8", ccc=occ_pixels)[0])
        return np.asarray(track, np.int32), np.asarray(gains, np.float32), refs

    track, gains, refs = oracle_pass()
    assert len({tuple(t[2:]) for t in track}) >= 3, "the filtered estimate must follow the drift (else the test shows nothing)"
    assert (track[:, :2] != track[:, 2:]).any(), "the Kalman filter must lag the raw argmax somewhere"

    gpu_pipe.set_ccc_model(filt, bias)
    gpu_pipe.set_ccc_kalman_model(1.0, 10.0)
    configure(gpu_pipe, c)
    # (a) one resident batch
    gpu_pipe.reset_white_balance_temporal_consistency()
    out = gpu_pipe.apply_device(torch.from_numpy(frames).cuda(), "bayer_gbrg8")
    torch.cuda.synchronize()
    got_track = gpu_pipe.get_ccc_track(n)
    got_info = gpu_pipe.get_white_balance_info(n)
    assert np.array_equal(got_track, track), "batch: (u, v) sequence differs first at frame %d" % int(np.argmax((got_track != track).any(axis=1)))
    assert np.array_equal(got_info[:, 0:3], gains), "batch: gains differ"
    out = out.cpu().numpy()
    for i in range(n):
        assert_images_equal(out[i], refs[i], "batch frame %d" % i)
    del out
    # (b) 64 single calls on the same stream (host frames through process()).  resetWhiteBalanceTemporalConsistency only re-arms
    # first_frame_ (convolutional_color_constancy.cpp:433-435): the error covariance the first pass left behind stays, so the
    # second pass filters differently from the first -- the oracle's two filters are walked on in the same way
    gpu_pipe.reset_white_balance_temporal_consistency()
    occ_track.reset()
    occ_pixels.reset()
    first_track = track
    track, gains, refs = oracle_pass()
    assert not np.array_equal(track, first_track), "the kept covariance must show in the second pass"
    for i in range(n):
        got = gpu_pipe.process(frames[i], "bayer_gbrg8")
        t = gpu_pipe.get_ccc_track(1)[0]
        assert np.array_equal(t, track[i]), "single calls: frame %d (u, v) %s, oracle %s" % (i, t, track[i])
        assert np.array_equal(gpu_pipe.get_white_balance_info(1)[0][0:3], gains[i])
        assert_images_equal(got, refs[i], "single call frame %d" % i)
