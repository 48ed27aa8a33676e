"""A hazard hipcc cannot pad: the result of a dot instruction needs three wait states before another VALU instruction reads it
(gfx90a+), and the compiler does not look inside inline assembly.  tools/isa_hazard_check.py walks the -save-temps listing
of every kernel source and must find nothing (the remap's blend_rows is the one place that feeds dot results into inline
asm today).  CPU only: hipcc cross-compiles the listings; ~30 s."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_inline_asm_reads_a_fresh_dot_result():
    src = [os.path.join(ROOT, "raw_image_pipeline_amd", "csrc", f) for f in ("rip_remap.hip", "rip_stats.hip", "rip_chain.hip", "rip_ccc.hip", "rip_maps.hip", "rip_fused.hip")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hazard_check.py")] + src, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout + r.stderr
    for f in src:
        assert os.path.basename(f) + ": 0 finding(s)" in r.stdout, r.stdout
    # the translation units compiled a second time under the contracted model ship in the library too
    for f in ("rip_chain.hip", "rip_fused.hip"):
        assert f + " (RIP_FP_CONTRACT=1): 0 finding(s)" in r.stdout, r.stdout


def test_the_checker_sees_the_hazard_it_is_there_for(tmp_path):
    """A listing shaped like the round-4 bug (v_dot4 result read by the first instruction of an asm block one wait state
    later) is reported; the same with `s_nop 2` in front, or with three instructions in between, is not."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_hazard_check as hz
    head = "_ZN3rip4testEv:\n\tv_dot4_u32_u8 v36, v36, v17, 16\n\ts_and_b32 s1, s39, 0xffff\n"
    asm = "\t;;#ASMSTART\n%s\tv_mul_u32_u24 v38, v36, v7\n\tv_mad_u32_u24 v38, v37, v9, v38\n\t;;#ASMEND\n\ts_endpgm\n"
    cases = {"bad": head + asm % "", "nop": head + asm % "\ts_nop 2\n",
             "far": head + "\tv_add_u32_e32 v1, v2, v3\n\tv_add_u32_e32 v4, v2, v3\n" + asm % "",
             "overwritten": head.replace("s_and_b32 s1, s39, 0xffff", "v_mov_b32_e32 v36, v2") + asm % ""}
    found = {}
    for name, text in cases.items():
        f = tmp_path / (name + ".s")
        f.write_text(text)
        found[name] = hz.check_listing(str(f))
    assert len(found["bad"]) == 1 and "v36" in found["bad"][0] and "1 wait state" in found["bad"][0], found["bad"]
    assert found["nop"] == [] and found["far"] == [] and found["overwritten"] == []
