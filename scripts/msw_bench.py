"""The mate-rescue alignments alone (bmh_matesw_batch_device: csrc/pair_kernels.hip): n jobs of one mate against a window that holds it, as mem_matesw cuts them
(BASELINE configs[3]: 150 bp mates, windows of ~480 rows), both forms of the kernel.  usage: msw_bench.py [n_jobs] [read_len] [window]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip import fmindex, synth
n_jobs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
win = int(sys.argv[3]) if len(sys.argv) > 3 else 480
L = B.load_library()
g = synth.make_genome(4_000_000, seed=5, repeat_frac=0.3, repeat_len=(100, 400), repeat_copies=(5, 40), repeat_div=0.03) if os.environ.get("MSW_HARD") else synth.make_genome(4_000_000, seed=5)
idx = fmindex.build_fmd_index(g, device="cuda:0")
codes = np.concatenate([g, np.zeros((-len(g)) % 4, np.uint8)]).reshape(-1, 4)
pac = np.ascontiguousarray(((codes[:, 0] << 6) | (codes[:, 1] << 4) | (codes[:, 2] << 2) | codes[:, 3]).astype(np.uint8))
dindex = B.Index.upload(idx, pac=pac, l_pac=len(g))
rng = np.random.default_rng(3)


class Job(C.Structure):
    _fields_ = [("rb", C.c_int64), ("re", C.c_int64), ("read", C.c_uint32), ("l_ms", C.c_int32), ("is_rev", C.c_int32), ("xtra", C.c_int32), ("bl_off", C.c_uint32), ("pad", C.c_uint32)]


p0 = rng.integers(1000, len(g) - rl - 1000, size=n_jobs)
reads = g[p0[:, None] + np.arange(rl)[None, :]].copy()
m = rng.random(reads.shape) < 0.01
reads[m] = (reads[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
if os.environ.get("MSW_HARD"):          # a harder mix for the kernel-against-kernel comparison: diverged mates, gaps, N, windows without the mate, repeats inside the window
    k = np.nonzero(rng.random(n_jobs) < 0.25)[0]
    mm = rng.random((len(k), rl)) < 0.10
    sub = reads[k]; sub[mm] = (sub[mm] + rng.integers(1, 4, size=int(mm.sum()))) & 3; reads[k] = sub
    for i in np.nonzero(rng.random(n_jobs) < 0.15)[0]:
        at, L_ = int(rng.integers(15, rl - 15)), int(rng.integers(1, 9))
        row = reads[i]
        reads[i] = np.concatenate([row[:at], row[at + L_:], rng.integers(0, 4, size=L_).astype(np.uint8)]) if rng.random() < 0.5 else np.concatenate([row[:at], rng.integers(0, 4, size=L_).astype(np.uint8), row[at:]])[:rl]
    nn = rng.random(reads.shape) < 0.002
    reads[nn] = 4
    miss = np.nonzero(rng.random(n_jobs) < 0.1)[0]
    p0[miss] = rng.integers(1000, len(g) - rl - 1000, size=len(miss))
jobs = (Job * n_jobs)()
off = rng.integers(0, win - rl, size=n_jobs)
for i in range(n_jobs):
    jobs[i].rb, jobs[i].re, jobs[i].read, jobs[i].l_ms, jobs[i].is_rev = int(p0[i] - off[i]), int(p0[i] - off[i] + win), i, rl, 0
    jobs[i].xtra = 0x40000 | 0x80000 | (0x10000 if rl < 250 else 0) | 19
if os.environ.get("MSW_ONLY"):          # debug: only these jobs (a:b ranges or single ids, comma-separated), in this order
    ids = []
    for part in os.environ["MSW_ONLY"].split(","):
        ids += list(range(int(part.split(":")[0]), int(part.split(":")[1]))) if ":" in part else [int(part)]
    sub = (Job * len(ids))()
    for k, i in enumerate(ids):
        sub[k].rb, sub[k].re, sub[k].read, sub[k].l_ms, sub[k].is_rev, sub[k].xtra = jobs[i].rb, jobs[i].re, jobs[i].read, jobs[i].l_ms, jobs[i].is_rev, jobs[i].xtra
    jobs_all, jobs, n_jobs_all, n_jobs = jobs, sub, n_jobs, len(ids)
    reads_sel = ids
r = torch.from_numpy(synth.codes_to_ascii(reads.reshape(-1))).cuda()
o = (torch.arange(len(reads), dtype=torch.int64) * rl).to(torch.int32).cuda()
ep = B.ExtParams.default()
L.bmh_matesw_batch_device.restype = C.c_int
L.bmh_matesw_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
outs = {}
for knob in (1, 0, 1, 0):
    L.bmh_tune_set(b"MSW_REG", knob, 0)
    out = np.zeros((n_jobs, 7), np.int32)
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = L.bmh_matesw_batch_device(dindex.handle, r.data_ptr(), o.data_ptr(), C.byref(ep), C.byref(jobs), n_jobs, out.ctypes.data_as(C.c_void_p), None)
        ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0, L.bmh_last_error()
    outs[knob] = out
    cells = float(n_jobs) * rl * win
    print("MSW_REG=%d: %d jobs of %d x %d: %s ms -> best %.2f ms = %.2f TCUPS (first pass cells only)" % (knob, n_jobs, rl, win, [round(t, 2) for t in ts], min(ts), cells / min(ts) / 1e9), flush=True)
print("identical results:", bool(np.array_equal(outs[0], outs[1])), "| scores >= 100:", int((outs[1][:, 0] >= 100).sum()), "| with a second-best:", int((outs[1][:, 3] > 0).sum()), "| start found:", int((outs[1][:, 5] >= 0).sum()))
bad = np.nonzero((outs[0] != outs[1]).any(1))[0]
if bad.size:                            # which of the two kernels disagrees with the host walk of the striped kernel (csrc/local_sw.cpp)?
    L.bmh_local_sw_c.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    text = np.concatenate([g, 3 - g[::-1]])
    n_reg = n_lds = 0
    for i in bad[:2000]:
        q = np.ascontiguousarray(reads[jobs[i].read]); t = np.ascontiguousarray(text[jobs[i].rb:jobs[i].re])
        want = np.zeros(7, np.int32)
        L.bmh_local_sw_c(rl, q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), C.byref(ep), jobs[i].xtra, want.ctypes.data_as(C.c_void_p))
        r_ok, l_ok = np.array_equal(outs[1][i], want), np.array_equal(outs[0][i], want)
        n_reg += not r_ok; n_lds += not l_ok
        if n_reg + n_lds <= 6 and not (r_ok and l_ok):
            print("job", i, "rb", jobs[i].rb, "re", jobs[i].re, "register form", outs[1][i], "LDS form", outs[0][i], "host", want)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "msw_bad.npz"), ids=bad[:64], q=np.stack([reads[jobs[i].read] for i in bad[:64]]), t=np.stack([text[jobs[i].rb:jobs[i].re] for i in bad[:64]]),
             xtra=np.array([jobs[i].xtra for i in bad[:64]]), reg=outs[1][bad[:64]], lds=outs[0][bad[:64]])
    print("%d jobs differ between the kernels; of the first %d: register form wrong %d, LDS form wrong %d" % (bad.size, min(bad.size, 2000), n_reg, n_lds))
assert bad.size == 0
