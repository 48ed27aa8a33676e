"""Race hunt inside the driver's `-m gpu` run (the long form is tools/probes/determinism_stress.py).

The grey-world / pca statistics kernel finishes its frames itself: every workgroup adds its partial sums to one of eight
per-XCD shards of the frame's record with returning agent-scope atomics, draws a ticket, and the workgroup with the last
ticket collects the shards, writes the gains and hands the record back ZEROED for the next batch -- no memset, no
finalisation launch (rip_stats.hip stat_flush).  That ordering leans on gfx950 behaviour, not on the HIP memory model, so it
is hammered here: a lost update changes the gains (every pixel of the frame changes), a record that does not come back
zeroed poisons the same frame slot of the NEXT batch.  Also covered by the same loop: the LDS-DMA ring of the remap and the
frame-group split of the chain.  white_balance.cpp:59-64 is the reference call the sums feed."""
import numpy as np
import pytest

from raw_image_pipeline_amd import synth

pytestmark = pytest.mark.gpu

W, H = 2448, 2048


@pytest.mark.parametrize("method", ["gray_world", "pca"])
def test_fused_statistics_finalisation_is_deterministic_across_batches(rip_lib, method):
    """Batches of 1 / 3 / 17 / 64 resident frames x statistics grids of 8 / 264 / 2048 workgroups (8 = one workgroup per
    shard, 264 = not a multiple of the frame count, 2048 = the default), each processed repeatedly, interleaved so that every
    launch starts from the records the previous, differently shaped launch left behind: every frame of every repetition
    equals the frame's single-frame result bit for bit."""
    import torch
    from raw_image_pipeline_amd import RawImagePipeline
    pipe = RawImagePipeline(False, "", "", "", device=0)
    pipe.set_stream(torch.cuda.current_stream())
    synth.configure_full_chain(pipe, W, H, method)
    distinct = 3
    base = [synth.gen_frame(W, H, "bayer_rggb8", seed=40 + i, kind="scene", tint=(0.55 + 0.1 * i, 1.0, 0.75 - 0.1 * i)) for i in range(distinct)]
    singles = torch.from_numpy(np.stack([pipe.process(b, "bayer_rggb8") for b in base])).cuda()
    assert not torch.equal(singles[0], singles[1])
    dev = torch.from_numpy(np.stack(base)).cuda()
    batches = {}
    for n in (1, 3, 17, 64):
        order = (torch.arange(n, device="cuda") * 2 + n) % distinct
        batches[n] = (dev[order].contiguous(), order)
    launches = 0
    for rep in range(6):
        for blocks in (8, 264, 2048):
            pipe.set_tunable("stats_blocks", blocks)
            for n, (frames, order) in batches.items():
                out = pipe.apply_device(frames, "bayer_rggb8")
                launches += 1
                if not torch.equal(out, singles[order]):
                    bad = (out != singles[order]).flatten(1).any(dim=1).nonzero().flatten().tolist()
                    pytest.fail("%s, repetition %d, %d statistics workgroups, batch of %d: frames %s differ from their "
                                "single-frame result" % (method, rep, blocks, n, bad[:8]))
    assert launches == 72


def test_full_step_is_bit_identical_over_repetitions_and_ring_settings(rip_lib):
    """The short form of tools/probes/determinism_stress.py: one resident 48-frame batch of the config-2 chain, 12 times
    under the default launch shape and 4 times each under three other ring / frame-group settings."""
    import torch
    from raw_image_pipeline_amd import RawImagePipeline
    pipe = RawImagePipeline(False, "", "", "", device=0)
    pipe.set_stream(torch.cuda.current_stream())
    synth.configure_full_chain(pipe, W, H, "grey_world")
    base = [synth.gen_frame(W, H, "bayer_rggb8", seed=60 + i, kind="scene", tint=(0.6 + 0.05 * i, 1.0, 0.55)) for i in range(3)]
    frames = torch.from_numpy(np.stack([base[i % 3] for i in range(48)])).cuda()
    first = pipe.apply_device(frames, "bayer_rggb8").clone()
    out = torch.empty_like(first)
    settings = [{}] * 12 + [{"remap_stages": 2}] * 4 + [{"remap_frames": 7, "chain_frames": 5}] * 4 + [{"remap_per_cu": 3, "chain_blocks": 1024}] * 4
    for i, st in enumerate(settings):
        for k in ("remap_stages", "remap_frames", "chain_frames", "remap_per_cu", "chain_blocks"):
            pipe.set_tunable(k, st.get(k, 0))
        pipe.apply_device(frames, "bayer_rggb8", out=out)
        assert torch.equal(out, first), "repetition %d (%s) differs from the first result" % (i, st or "defaults")
    single = pipe.process(base[1], "bayer_rggb8")
    assert np.array_equal(first[1].cpu().numpy(), single) and np.array_equal(first[46].cpu().numpy(), single)


@pytest.mark.gpu
def test_switching_streams_between_frame_calls_keeps_the_scratch_state_ordered(rip_lib):
    """The handle's scratch state crosses frame calls in stream order (statistics records and ccc histogram counters handed
    back zeroed by the previous call's kernels, the Kalman state).  rip_set_stream orders the new stream behind the work left
    on the old one, so a caller may alternate streams from call to call without synchronising: 40 resident single frames,
    ccc with temporal consistency and grey-world, alternating between three streams, equal the same sequence on one stream."""
    import torch
    from raw_image_pipeline_amd import RawImagePipeline, synth
    w, h, n = 1440, 1080, 40
    frames = torch.from_numpy(np.stack([synth.gen_frame(w, h, "bayer_gbrg8", seed=9300 + i, kind="scene", tint=(0.55 + 0.01 * i, 1.0, 0.6)) for i in range(n)])).cuda()
    streams = [torch.cuda.Stream() for _ in range(3)]
    filt, bias = synth.ccc_model()
    for method in ("ccc", "grey_world"):
        outs = []
        for alternate in (False, True):
            pipe = RawImagePipeline(False, "", "", "", device=0)
            synth.configure_full_chain(pipe, w, h, "grey_world")
            if method == "ccc":
                pipe.set_ccc_model(filt, bias)
                pipe.set_ccc_kalman_model(1.0, 10.0)
                pipe.set_white_balance_method("ccc")
                pipe.set_white_balance_temporal_consistency(True)
            out = torch.empty((n, h, w, 3), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            for i in range(n):
                st = streams[i % 3] if alternate else streams[0]
                pipe.set_stream(st)
                pipe.apply_device(frames[i:i + 1], "bayer_gbrg8", out=out[i:i + 1])
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy())
        assert np.array_equal(outs[0], outs[1]), "%s: alternating streams changed %d values" % (method, int((outs[0] != outs[1]).sum()))
