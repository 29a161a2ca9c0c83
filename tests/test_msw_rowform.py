"""CPU check of the algebra behind csrc/pair_kernels.hip: msw_reg_kernel -- the mate rescue's striped local alignment (ksw_align2's byte-mode kernel,
/root/reference/src/ksw.c:389-547) computed as plain rows: first sweep with F starting at 0 at every lane's first column, the F that flows in from the lanes to the
left as a max-plus scan, E and the row maximum from the first sweep's H.  scripts/proto/msw_rowform_fuzz.cpp holds that form in C++ and compares it with
csrc/local_sw.cpp, the host walk of the striped kernel (itself pinned to the compiled reference by tests/test_oracle.py), on random windows: substitutions, gaps,
repeats, N, six scorings, both exit flags.  The GPU test (test_mate_rescue_alignments_on_the_device) compares the kernel itself with local_sw.cpp."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_form_equals_the_striped_kernel_walk():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "msw_rowform_fuzz")
    src = [os.path.join(ROOT, "scripts", "proto", "msw_rowform_fuzz.cpp"), os.path.join(ROOT, "bwa-mem_gpu_amd", "csrc", "local_sw.cpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "bwa-mem_gpu_amd", "csrc"), *src, "-o", exe])
    for seed in (1, 2):
        r = subprocess.run([exe, "6000", str(seed)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]
