#!/bin/bash
# One entry point for the GPU calls of a round: `gpurun -- bash tools/run_batch.sh <experiment> [args]`.
# Results land under gpurun_out/r6_<experiment>/ (scratch); what is worth keeping is copied to profiles/ by hand.
# Rounds 4-5 used one script per call (tools/batches/, kept for the record); round 6 on: one case per experiment here.
# Library variants some cases name (raw_image_pipeline_amd/variants/, git-ignored) are built first with
#   python tools/ab_chain.py build exp=-DRIP_EXPERIMENTS [name=-DFLAG,...]      (the tree with extra flags)
#   git worktree add /tmp/wt <commit> && (cd /tmp/wt && python -c "from raw_image_pipeline_amd import build as b; b.build(force=True, out='<repo>/raw_image_pipeline_amd/variants/r5.so', tag='_r5')")   (an older tree: r5.so = round 5's 9e7d5bb, head.so)
set -u
exp=${1:?experiment name}; shift
out=gpurun_out/r6_$exp; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
V=raw_image_pipeline_amd/variants
case $exp in
  remap_exp)     # timing-only switches of the ring remap, one process / handle / output allocation (EXPERIMENTS.md round 6)
    RIP_LIBRARY=$V/exp.so python tools/probes/remap_exp_probe.py --masks "${1:-0,1,2,4,6,8,24,16,32,64}" --rounds "${2:-3}" ${3:+--tunable $3} 2>&1 | tee $out/probe${4:-}.log ;;
  remap_exp2)    # second look: stores only, combinations, residency, small (Infinity-Cache-resident) batches
    export RIP_LIBRARY=$V/exp.so
    python tools/probes/remap_exp_probe.py --masks 0,40,42,70,78,72 --rounds 3 2>&1 | tee $out/masks.log
    python tools/probes/remap_exp_probe.py --masks 0 --rounds 3 --tunable remap_per_cu=4,5,6 2>&1 | tee $out/per_cu.log
    for b in 4 8 16 32; do python tools/probes/remap_exp_probe.py --masks 0,8 --rounds 2 --steps 20 --batch $b 2>&1 | grep -v "^round" | tee $out/batch$b.log; done ;;
  remap_exp3)    # are the stores latency-bound behind the in-order counter, or slow as 12-byte lanes?
    export RIP_LIBRARY=$V/exp.so
    python tools/probes/remap_exp_probe.py --masks 0,128,40,168,298,552,512,258,640 --rounds 3 2>&1 | tee $out/masks.log ;;
  remap_exp4)    # occupancy / frames-per-visit scaling of the stores-only, loads-only and complete kernel
    export RIP_LIBRARY=$V/exp.so
    python tools/probes/remap_exp_probe.py --masks 40,24,0 --rounds 2 --tunable remap_per_cu=2,3,4,6 2>&1 | grep -v "^round" | tee $out/per_cu.log
    python tools/probes/remap_exp_probe.py --masks 40,24,0 --rounds 2 --tunable remap_frames=1,2,4,8,16 2>&1 | grep -v "^round" | tee $out/frames.log ;;
  remap_exp5)    # how the tiles are dealt to the XCDs (bits 12..15 of the mask): stores only, loads only, complete
    export RIP_LIBRARY=$V/exp.so
    python tools/probes/remap_exp_probe.py --masks 40,4136,8232,12328,16424,24,4120,8216,12312,16408,0,4096,8192,12288,16384 --rounds 2 2>&1 | grep -v "^round" | tee $out/masks.log ;;
  remap_exp6)    # run length of the round-robin deal, and frames per visit / residency under it
    export RIP_LIBRARY=$V/exp.so
    python tools/probes/remap_exp_probe.py --masks 0,8192,16384,20480,24576,36864,28672,32768 --rounds 3 2>&1 | grep -v "^round" | tee $out/masks.log
    python tools/probes/remap_exp_probe.py --masks 16384,24576 --rounds 2 --tunable remap_frames=3,4,5,6,8 2>&1 | grep -v "^round" | tee $out/frames.log
    python tools/probes/remap_exp_probe.py --masks 16384 --rounds 2 --tunable remap_per_cu=3,4,5,6 2>&1 | grep -v "^round" | tee $out/per_cu.log ;;
  remap_exp7)    # frames per visit x run length of the deal (and the contiguous deal beside them)
    export RIP_LIBRARY=$V/exp.so
    python tools/probes/remap_exp_probe.py --masks 0,8192,16384,24576 --rounds 2 --tunable remap_frames=4,6,8,10,12,16 2>&1 | grep -v "^round" | tee $out/frames.log
    python tools/probes/remap_exp_probe.py --masks 16384 --rounds 2 --tunable remap_stages=2,3,4 2>&1 | grep -v "^round" | tee $out/stages.log ;;
  remap_deal)    # the round-robin deal (RIP_REMAP_DEAL) x frames per visit on the four sensor geometries, two-kernel path and config 5's fused kernel
    for size in 2448x2048 1440x1080 1920x1200 3840x2160; do for deal in 0 1; do
      echo "== config2 $size deal=$deal"; RIP_REMAP_DEAL=$deal python tools/probes/remap_exp_probe.py --size $size --rounds 2 --tunable remap_frames=${2:-0,4,6,8,10,12,16} 2>&1 | grep "^mask"
    done; done | tee $out/config2.log
    for deal in 0 1 2; do
      echo "== config5 3840x2160 deal=$deal"; RIP_REMAP_DEAL=$deal python tools/probes/remap_exp_probe.py --workload config5 --size 3840x2160 --rounds 2 --tunable remap_frames=8,12,16 2>&1 | grep "^mask"
    done | tee $out/config5.log
    python -m pytest tests -m gpu -x -q -k "undist or remap or fused or config" 2>&1 | tail -5 | tee $out/pytest.log ;;
  asan)          # the host-frame ring, copy threads, rig, fork handling under ASan + UBSan on the GPU box (tools/run_asan.sh)
    tools/run_asan.sh python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "${1:-submit or collect or ring or rig or thread or fork or pageable or pool or frontend or facade or error or taps or abi}" > $out/pytest_full.log 2>&1
    grep -v "^  File\|^Loading" $out/pytest_full.log | head -c 20000 | tee $out/pytest.log | tail -30 ;;
  remap_exp8)    # stores: half-line segments? frame stride?  (timing only)
    export RIP_LIBRARY=$V/exp.so
    python tools/probes/remap_exp_probe.py --masks 0,1024,40,1064,42 --rounds 3 2>&1 | grep -v "^round" | tee $out/masks.log
    python tools/probes/remap_exp_probe.py --masks 106,104 --rounds 2 --tunable remap_frames=1,2,4,8,16 2>&1 | grep -v "^round" | tee $out/frames.log ;;
  remap_stream)  # the ring as one stream of units: parity of everything that gathers, then timing (and the components)
    python -m pytest tests -m gpu -x -q -k "undist or remap or fused or config or determinism or fuzz" 2>&1 | tail -5 | tee $out/pytest.log
    python tools/probes/remap_exp_probe.py --rounds 3 --tunable remap_frames=${1:-0,4,6,8,12} 2>&1 | grep "^mask" | tee $out/frames.log
    RIP_LIBRARY=$V/exp.so python tools/probes/remap_exp_probe.py --masks 0,8,16,32,24,40,64 --rounds 2 2>&1 | grep "^mask" | tee $out/components.log ;;
  remap_stream2) # per-visit cost of the stream kernel: stores only / loads only / complete against frames per visit
    export RIP_LIBRARY=$V/exp.so
    python tools/probes/remap_exp_probe.py --masks 104,88,0 --rounds 2 --tunable remap_frames=1,2,4,8,16 2>&1 | grep "^mask" | tee $out/frames.log
    python tools/probes/remap_exp_probe.py --masks 0,8,16,32,24,40,64 --rounds 2 2>&1 | grep "^mask" | tee $out/components.log ;;
  remap_abc)     # one process, one output allocation: HEAD's ring (per-visit), the stream ring with short-lived workgroups, with persistent ones
    python tools/probes/remap_exp_probe.py --libs head=$V/head.so,short=,pers=,glob= --set short:remap_persistent=0 --set pers:remap_persistent=1 --set glob:remap_persistent=2 --rounds 3 --tunable remap_frames=${1:-4,6,8} 2>&1 | grep "^mask" | tee $out/abc.log ;;
  mall)          # does the intermediate image survive in the Infinity Cache between chain and remap?  frame groups on two streams (overlap_groups), small enough for it
    for nt in -1 0; do for mode in 1 2; do
      echo "== RIP_CHAIN_NT=$nt overlap_mode=$mode"; RIP_CHAIN_NT=$nt RIP_OVERLAP_MODE=$mode python tools/probes/remap_exp_probe.py --rounds 2 --steps 4 --tunable overlap_groups=0,8,16,22,32,43,64 2>&1 | grep "^mask"
    done; done | tee $out/mall.log ;;
  hsv_tab)       # the enhancer's per-channel work tabulated (config 3): parity, then HEAD's library beside the tree's in one process
    python -m pytest tests -m gpu -x -q -k "enhancer or hsv or config3 or fuzz or taps or colour or color" 2>&1 | tail -4 | tee $out/pytest.log
    python tools/probes/remap_exp_probe.py --workload config3 --size 1920x1200 --libs head=$V/head.so,new= --rounds 3 2>&1 | grep "^mask" | tee $out/ab.log ;;
  chain_deal)    # runs of chunks dealt round-robin to the XCDs in the fused chain (compile-time variants), one process
    for wl in default_chain chain; do
      python tools/probes/remap_exp_probe.py --workload $wl --libs base=,d1=$V/deal1.so,d3=$V/deal3.so,d5=$V/deal5.so,d10=$V/deal10.so,d24=$V/deal24.so --rounds 3 2>&1 | grep "^mask" | sed "s/^/$wl /"
    done | tee $out/ab.log ;;
  chain_deal2)   # the runtime chain deal (default 3): parity, then frames per item visit under it
    python -m pytest tests -m gpu -x -q -k "not submit and not rig and not ring and not bench" 2>&1 | tail -4 | tee $out/pytest.log
    python tools/probes/remap_exp_probe.py --workload default_chain --rounds 2 --tunable chain_frames=2,4,6,8,16 2>&1 | grep "^mask" | sed "s/^/default_chain /" | tee $out/frames.log
    python tools/probes/remap_exp_probe.py --workload chain --rounds 2 --tunable chain_frames=8,16,32 2>&1 | grep "^mask" | sed "s/^/chain /" | tee -a $out/frames.log
    python tools/probes/remap_exp_probe.py --workload chain --rounds 2 --tunable chain_deal=0,1,2,3,4,6 2>&1 | grep "^mask" | sed "s/^/chain /" | tee -a $out/frames.log
    python tools/probes/remap_exp_probe.py --workload default_chain --rounds 2 --tunable chain_deal=0,1,2,3,4,6 2>&1 | grep "^mask" | sed "s/^/default_chain /" | tee -a $out/frames.log ;;
  chain_frames)  # frames per item visit of the VALU-bound stage sets under the deal, and the demosaic-only set
    python tools/probes/remap_exp_probe.py --workload chain --rounds 3 --tunable chain_frames=12,16,24,32,64 2>&1 | grep "^mask" | sed "s/^/chain /" | tee $out/frames.log
    python tools/probes/remap_exp_probe.py --workload config3 --size 1920x1200 --rounds 3 --tunable chain_frames=8,16,24,32,64 2>&1 | grep "^mask" | sed "s/^/config3 /" | tee -a $out/frames.log
    python tools/probes/remap_exp_probe.py --workload config2 --rounds 3 --tunable chain_frames=16,32 2>&1 | grep "^mask" | sed "s/^/config2 /" | tee -a $out/frames.log
    python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee $out/bench.log ;;
  step_ab)       # the whole config 2 step, round 5's library (variants/r5.so) beside the tree's, one process
    python tools/probes/remap_exp_probe.py --workload config2 --libs r5=$V/r5.so,new= --rounds 4 2>&1 | grep "^mask" | tee $out/ab.log ;;
  step_sweep)    # tunables inside the whole config 2 step (the chain behaves differently in front of the remap)
    python tools/probes/remap_exp_probe.py --workload config2 --rounds 3 --tunable chain_deal=0,1,2,3,4 2>&1 | grep "^mask" | tee $out/chain_deal.log
    python tools/probes/remap_exp_probe.py --workload config2 --rounds 3 --tunable remap_frames=4,6,8 2>&1 | grep "^mask" | tee $out/remap_frames.log
    python tools/probes/remap_exp_probe.py --workload config2 --rounds 3 --tunable remap_deal=0,1,2 2>&1 | grep "^mask" | tee $out/remap_deal.log ;;
  all_ab)        # every bench workload, round 5's library beside the tree's, one process each
    for w in config2:2448x2048 chain:2448x2048 default_chain:2448x2048 config3:1920x1200 config5:3840x2160 config2:1440x1080 config2:1920x1200 config2:3840x2160; do
      python tools/probes/remap_exp_probe.py --workload ${w%%:*} --size ${w##*:} --libs r5=$V/r5.so,new= --rounds 3 2>&1 | grep "^mask" | sed "s/^/$w /"
    done | tee $out/ab.log ;;
  tile_shape)    # 128 x 8 tiles (whole 128-byte lines per row segment) against 64 x 16, complete / stores only / loads only, one process
    python tools/probes/remap_exp_probe.py --workload config2 --libs t64x16=$V/exp.so,t128x8=$V/t128x8.so --masks 0,40,24 --rounds 3 2>&1 | grep "^mask" | tee $out/ab.log ;;
  remap_store_bits) # cache-policy bits of the remap's output stores under the deal (round 5, contiguous deal: nt +14 %, sc1 +10 %)
    RIP_LIBRARY=$V/exp.so python tools/probes/remap_exp_probe.py --masks 0,2048,4096,6144 --rounds 3 2>&1 | grep "^mask" | tee $out/masks.log ;;
  deal_traffic)  # what the deals do to the L2-level traffic (FETCH_SIZE / WRITE_SIZE per launch), and the time beside it
    for d in 0 1 2 4; do
      RIP_REMAP_DEAL=$d python tools/collect_pmc.py $out/remap_deal$d config2 > /dev/null 2>&1
      echo "== RIP_REMAP_DEAL=$d"; grep "remap_ring_kernel\|chain_fast" $out/remap_deal$d/pmc_summary.txt
    done | tee $out/remap.log
    for d in 0 3 6 12; do
      RIP_CHAIN_DEAL=$d python tools/collect_pmc.py $out/chain_deal$d chain default_chain > /dev/null 2>&1
      echo "== RIP_CHAIN_DEAL=$d"; grep "chain_fast" $out/chain_deal$d/pmc_summary.txt
    done | tee $out/chain.log
    rm -rf $out/*/pmc_*_fetch $out/*/pmc_*_write $out/*/pmc_*_rdsplit ;;
  deal_final)    # the final deal defaults: invariance test, then the run lengths once more in the step
    python -m pytest tests -m gpu -x -q -k "deals_of_tiles or undist or fused or config2 or flip" 2>&1 | tail -4 | tee $out/pytest.log
    python tools/probes/remap_exp_probe.py --workload config2 --rounds 3 --tunable remap_deal=1,2,4,6 2>&1 | grep "^mask" | tee $out/remap_deal.log
    python tools/probes/remap_exp_probe.py --workload config2 --rounds 3 --tunable chain_deal=0,2,3,4 2>&1 | grep "^mask" | tee $out/chain_deal.log
    python tools/probes/remap_exp_probe.py --workload default_chain --rounds 3 --tunable chain_deal=0,2,3,4,6 2>&1 | grep "^mask" | tee $out/chain_deal_default.log ;;
  step_order)    # is the chain slower in the step with the new library, or is it the handle's allocation?  new first, r5 second, new with chain_deal 0 third
    python tools/probes/remap_exp_probe.py --workload config2 --libs new=,r5=$V/r5.so,new0= --set new0:chain_deal=0 --rounds 4 2>&1 | grep "^mask" | tee $out/ab.log
    python tools/probes/remap_exp_probe.py --workload config2 --libs r5=$V/r5.so,new0=,new= --set new0:chain_deal=0 --rounds 4 2>&1 | grep "^mask" | tee -a $out/ab.log ;;
  chain_sched)   # LLVM machine-scheduler strategies for rip_chain.hip (RIP_CHAIN_SCHED builds), one process, the chain writing the shared output
    L=base=,ilp=$V/sched_iterative-ilp.so,minreg=$V/sched_iterative-minreg.so,maxocc=$V/sched_iterative-maxocc.so,memcl=$V/sched_max-memory-clause.so,dflt=$V/sched_default.so
    for wl in chain default_chain config2; do python tools/probes/remap_exp_probe.py --workload $wl --libs $L --rounds 4 2>&1 | grep "^mask" | sed "s/^/$wl /"; done | tee $out/ab.log
    python tools/probes/remap_exp_probe.py --workload config3 --size 1920x1200 --libs $L --rounds 4 2>&1 | grep "^mask" | sed "s/^/config3 /" | tee -a $out/ab.log ;;
  chain_sched2)  # max-ilp (the tree) against iterative-ilp once more, both orders, the chain writing the shared output
    for i in 1 2; do
      python tools/probes/remap_exp_probe.py --workload chain --libs ilp=$V/sched_iterative-ilp.so,base= --rounds 5 2>&1 | grep "^mask"
      python tools/probes/remap_exp_probe.py --workload chain --libs base=,ilp=$V/sched_iterative-ilp.so --rounds 5 2>&1 | grep "^mask"
    done | tee $out/ab.log ;;
  deal_check)    # the final remap deal (4 tile rows) on the other geometries and on config 5's fused kernel
    python tools/probes/remap_exp_probe.py --workload config5 --size 3840x2160 --rounds 3 --tunable remap_deal=0,1,2,4,8 2>&1 | grep "^mask" | sed "s/^/config5 /" | tee $out/ab.log
    for size in 1440x1080 1920x1200 3840x2160; do python tools/probes/remap_exp_probe.py --workload config2 --size $size --rounds 3 --tunable remap_deal=0,1,4 2>&1 | grep "^mask" | sed "s/^/config2:$size /"; done | tee -a $out/ab.log ;;
  act_model)     # is the remap bound by DRAM row activations?  row-segment patterns against tile-linear ones, one stream at a time and both
    RIP_LIBRARY=$V/exp.so python tools/probes/remap_exp_probe.py --masks 24,28,40,42,8,14,78 --rounds 3 2>&1 | grep "^mask" | tee $out/masks.log ;;
  survey)        # every reachable path (tools/path_survey.py), round 5's library and the tree's; latency probe
    RIP_LIBRARY=$PWD/$V/r5.so python tools/path_survey.py 2>&1 | tee $out/r5.log | tail -40
    python tools/path_survey.py 2>&1 | tee $out/new.log | tail -40
    python tools/latency_probe.py 2>&1 | tee $out/latency.log | tail -12
    RIP_LIBRARY=$PWD/$V/r5.so python tools/latency_probe.py 2>&1 | tee $out/latency_r5.log | tail -12 ;;
  soak)          # 2 000-case fuzz soak against the oracle + the determinism stress (the deals, the enhancer tables and the new defaults under them)
    RIP_FUZZ_CASES=${1:-2000} python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $out/fuzz.log
    python tools/probes/determinism_stress.py 2>&1 | tail -12 | tee $out/determinism.log ;;
  write_pattern) # the remap's write pattern alone (tools/probes/write_pattern_probe.hip): pattern or kernel structure?
    B=tools/probes/bin/write_pattern_probe
    { $B 64 16 6 4; $B 64 16 1 4; $B 64 16 16 4; $B 64 16 6 0; $B 64 16 12 4; $B 128 8 6 4; $B 256 4 6 4; } 2>&1 | tee $out/probe.log ;;
  write_stride)  # does the distance between the frames of a visit matter to the write pattern?  (frame stride 15 040 512 = 2^15 x 459)
    B=tools/probes/bin/write_pattern_probe
    for pad in 0 256 1024 4096 8192 12288 32768 65536 69632 1048576 2101248; do $B 64 16 6 4 256 $pad 3 | grep -v "mode 0"; done 2>&1 | tee $out/probe.log
    for f in 2 3 4; do $B 64 16 $f 4 256 0 3 | grep -v "mode 0"; done 2>&1 | tee -a $out/probe.log ;;
  write_lock)    # persistent workgroups held to one frame by a loose frame lock: does the write stream reach the one-frame rate?
    tools/probes/bin/write_pattern_probe 64 16 1 4 256 0 8 2>&1 | tee $out/probe.log ;;
  remap_locked)  # (needs tools/probes/remap_frame_locked_experiment.patch applied) the frame-locked persistent remap kernel: parity against the per-visit ring, then its time for several leads
    timeout 600 python -m pytest tests -m gpu -x -q -k "frame_locked" 2>&1 | tail -6 | tee $out/pytest.log
    timeout 300 python tools/probes/remap_exp_probe.py --workload config2 --rounds 3 --tunable remap_locked=${1:-0,1,2,3,4,8} 2>&1 | grep "^mask" | tee $out/lead.log ;;
  suite)         # whole GPU suite + smoke
    python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $out/pytest.log
    python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $out/smoke.log ;;
  bench)         # default line (+ any extra bench.py arguments)
    python bench.py "$@" 2>&1 | tee $out/bench.log ;;
  ab)            # tools/ab_chain.py run <args>: library variants in separate processes
    python tools/ab_chain.py run "$@" 2>&1 | tee $out/ab.log ;;
  sh)            # free-form: the rest of the line is the command
    bash -c "$*" 2>&1 | tee $out/sh.log ;;
  *) echo "unknown experiment $exp"; exit 2 ;;
esac
