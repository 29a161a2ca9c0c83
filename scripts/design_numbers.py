#!/usr/bin/env python3
"""The table of DESIGN.md section 5.1 from the round's bench lines: design_numbers.py [tag]  (profiles/<tag>_bench*.json)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
def ld(nm):
    p = os.path.join(ROOT, "profiles", nm)
    return json.loads([l for l in open(p) if l.startswith("{")][-1]) if os.path.exists(p) else None
rows = [("**configs[1]: 1 M x 150 bp single-end, the driver's command (20 timed steps)**", f"{tag}_bench_driver_cmd.json"), ("the same, default 40 timed steps", f"{tag}_bench.json"),
        ("configs[3]: `--paired`", f"{tag}_bench_paired.json"), ("configs[4]: `--read-len 300`", f"{tag}_bench_300bp.json")]
print("| | Mreads/s (`value`) | ms / step | incl. PCIe | `clock_mhz` | stages alone: seeding / chaining / extension (ms) | extension: executed lane-instr per reference cell / `roofline.frac` (of 78.6 T) / issue slots of the measured ceiling | `smem_backward_kernel`: ms / frac of 8 TB/s | reads -> SAM text (Mreads/s) |")
print("|---|---|---|---|---|---|---|---|---|")
cpu = []
for name, fn in rows:
    d = ld(fn)
    if not d:
        continue
    e = d.get("extension_stage", d["roofline"]); i = d["stage_ms_isolated"]; h = d["roofline_hbm_kernel"]
    sam = d.get("next_rows", {}).get("reads_to_sam_native", {}).get("Mreads_per_s")
    print(f"| {name} | **{d['value']:.2f}** | {d['ms_per_step']:.2f} | {d['incl_pcie_value']:.2f} | {d['clock_mhz']:.0f} | {i['total']:.2f} / {i['chain']:.2f} / {i['extend']:.2f} | "
          f"{e.get('executed_lane_instr_per_reference_cell')} / **{e['frac']:.3f}** / {e.get('issue_slot_frac_of_measured_ceiling')} | {h['avg_ms']:.2f} / {h['frac']:.3f} | {sam} |")
    c = d.get("cpu_baseline")
    if c:
        cpu.append((c["value"], c["cores"], c["one_thread"]["value"], c.get("reference_code_one_thread", {}).get("value"), d.get("speedup_vs_cpu_baseline")))
if cpu:
    v = [c[0] for c in cpu]
    print(f"| CPU: the C restatement on the box's {cpu[0][1]} usable threads (one thread: {cpu[0][2]:.4f}; the reference's compiled code on one thread: {cpu[0][3]}) | {min(v):.3f}-{max(v):.3f} (the GPU path is {min(c[4] for c in cpu):.0f}-{max(c[4] for c in cpu):.0f} x that) | | | | | | | |")
