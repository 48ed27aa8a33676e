set -x
mkdir -p gpurun_out/r3
python tools/ab_chain.py run --workload config2,config5,config3,chain base nt2 base nt2 > gpurun_out/r3/ab_nt.log 2>&1
for v in base nt2; do RIP_LIBRARY=raw_image_pipeline_amd/variants/$v.so python tools/latency_probe.py > gpurun_out/r3/lat_$v.log 2>&1; done
