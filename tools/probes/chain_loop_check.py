#!/usr/bin/env python3
"""Frame-loop schedule of the fused chain variants bench.py launches: instructions, s_waitcnt and s_nop inside the loop, lane spills in
the kernel (tools/chain_ledger.py walker on a fresh -S listing of the tree; extra hipcc flags as arguments).  Run it before and after any
change near fast_chunks: a run-time divisor there once cost config 2 eleven waits per trip and 1.5 % (EXPERIMENTS.md round 6)."""
import sys,collections,subprocess,os,shutil
sys.path.insert(0,'/root/repo/tools')
import chain_ledger as cl
d='/tmp/cmp/cur'; os.makedirs(d,exist_ok=True)
for f in ('rip_chain.hip','rip_device.hpp','rip_chain_dev.hpp','rip_kernels.hpp','rip_tile.hpp'):
    shutil.copy('/root/repo/raw_image_pipeline_amd/csrc/'+f,d)
subprocess.run("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -fno-slp-vectorize -D__HIP_PLATFORM_AMD__ -mllvm -amdgpu-sched-strategy=max-ilp -x hip -S --cuda-device-only -o chain.s rip_chain.hip".split()+sys.argv[1:],cwd=d,stderr=subprocess.DEVNULL)
for kern in ("ILi7ELi1ELi512E","ILi7ELi0ELi512E","ILi3ELi1ELi256E","ILi8ELi2ELi256E"):
    lines=open(d+'/chain.s').read().splitlines()
    i0=[i for i,l in enumerate(lines) if l.startswith("_ZN3rip12_GLOBAL__N_117chain_fast_kernel"+kern)][0]
    i1=next(i for i in range(i0,len(lines)) if 's_endpgm' in lines[i])
    body=lines[i0:i1+1]
    h=collections.Counter()
    for b in cl.loop_blocks(body):
        for ins in b["ins"]: h[ins.split()[0]]+=1
    full=collections.Counter(l.split()[0] for l in body if l.startswith('\t') and l.strip() and l.strip()[0] not in '.;')
    print(kern,'loop total',sum(h.values()),'waitcnt',h['s_waitcnt'],'nop',h['s_nop'],'| kernel readlane',full['v_readlane_b32'],'writelane',full['v_writelane_b32'])
