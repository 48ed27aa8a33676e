"""A hazard hipcc cannot pad: the result of a dot instruction needs three wait states before another VALU instruction reads it
(gfx90a+), and the compiler does not look inside inline assembly.  tools/isa_hazard_check.py walks the -save-temps listing
of every kernel source and must find nothing (the remap's blend_rows is the one place that feeds dot results into inline
asm today).  CPU only: hipcc cross-compiles the listings; ~30 s."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_inline_asm_reads_a_fresh_dot_result():
    src = [os.path.join(ROOT, "raw_image_pipeline_amd", "csrc", f) for f in ("rip_remap.hip", "rip_stats.hip", "rip_chain.hip", "rip_ccc.hip", "rip_maps.hip")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hazard_check.py")] + src, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    for f in src:
        assert os.path.basename(f) + ": 0 finding(s)" in r.stdout, r.stdout
