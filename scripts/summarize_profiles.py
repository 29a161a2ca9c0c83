"""Condense rocprofv3 output directories into the small summaries committed under profiles/.
usage: summarize_profiles.py <tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> [<pmc_sq_dir>]"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = ("pack_reads", "smem_", "cand_", "per_read_counts", "expand_kernel", "locate_kernel", "extend16", "extend_wide", "ext_", "calib_gather",
        "chain_", "emit_kernel", "materialize_kernel", "merge_kernel")
tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
sq_dir = sys.argv[5] if len(sys.argv) > 5 else None

def find(d, suffix):
    r = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return r[0] if r else None

p = find(stats_dir, "kernel_stats.csv")
rows = list(csv.reader(open(p)))
keep = [rows[0]] + [r for r in rows[1:] if any(o in r[0] for o in OURS)]
with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_bench.csv"), "w", newline="") as f:
    csv.writer(f).writerows(keep)

def pmc(d, counter):
    p = find(d, "counter_collection.csv")
    acc = {}
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] != counter: continue
        k = r["Kernel_Name"].split("(")[0]
        if not any(o in k for o in OURS): continue
        a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    return {k: {"launches": v[0], "avg_per_launch_KB": round(v[1] / v[0], 1)} for k, v in acc.items()}

out = {"FETCH_SIZE": pmc(fetch_dir, "FETCH_SIZE"), "WRITE_SIZE": pmc(write_dir, "WRITE_SIZE")}
if sq_dir:
    # raw SQ counters per launch (sums over the XCDs / SEs as rocprofv3 reports them); SQ_*_CYCLES and SQ_ACTIVE_INST_* count
    # quad-cycles (MI355X_MICROARCH.md), GRBM_GUI_ACTIVE is summed over the 8 XCDs
    sq = {}
    for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
        try:
            for k, v in pmc(sq_dir, c).items():
                sq.setdefault(k, {"launches": v["launches"]})[c] = v["avg_per_launch_KB"]
        except Exception as e:
            print("no", c, e)
    json.dump(sq, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_sq.json"), "w"), indent=1)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_write.json"), "w"), indent=1)
print("kernels:", len(keep) - 1, "pmc kernels:", len(out["FETCH_SIZE"]))
