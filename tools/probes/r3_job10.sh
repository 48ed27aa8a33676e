mkdir -p gpurun_out/r3
for v in nt1 nt0 nt1 nt0; do
  echo "$v"
  RIP_LIBRARY=raw_image_pipeline_amd/variants/$v.so python tools/perf_explore.py 2448x2048:256:d 2448x2048:256:wcg 3840x2160:256:d 1920x1200:256:wcgv 2>/dev/null | grep -v Loading
done > gpurun_out/r3/nt_stores.log 2>&1
python tools/ab_chain.py run --workload config5,config3,chain nt1 nt0 nt1 nt0 >> gpurun_out/r3/nt_stores.log 2>&1
