"""Condense rocprofv3 output directories into the small summaries committed under profiles/.

usage: summarize_profiles.py <tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <pmc_sq_dir>
Every directory holds the rocprofv3 output of one run of the SAME bench.py command plus that run's own bench line
(<dir>/bench.json), which carries `passes` (how often each stage ran in the process) and `config.workload_key`.
Writes  profiles/<tag>_kernel_stats_bench.csv  (the --kernel-trace --stats table, our kernels only) and
        profiles/<tag>_pmc.json                (per kernel and per kernel family: counters per pass of the stage).
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB; for this path's random-gather kernels FETCH_SIZE needs no
correction (63.9 B per 64-byte sector under bmh_calib_gather), so bytes = KB * 1024 as read.
"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = ("pack_reads", "smem_", "cand_", "per_read_counts", "expand_kernel", "locate_kernel", "extend16", "extend_wide", "ext_", "calib_",
        "chain_", "emit_kernel", "materialize_kernel", "merge_kernel", "merge2_kernel", "split_counts", "extpk", "reblock", "densify", "fin_")
# kernel family -> (substrings, which `passes` counter of the bench line divides its sums)
FAMILIES = {
    "forward": (("smem_forward_kernel",), "seed"),
    "backward": (("smem_backward_kernel", "cand_scatter"), "seed"),
    "locate": (("locate_kernel",), "seed"),
    "seed_other": (("pack_reads", "smem_filter", "per_read_counts", "expand_kernel", "cand_count"), "seed"),
    "extend": (("ext_closed_form", "ext_key", "ext_offsets", "ext_scatter", "extend16", "extend_wide", "extpk"), "extend"),
    "chain": (("chain_", "emit_kernel", "split_counts", "merge_kernel", "merge2_kernel"), "chain"),
}
SQ = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE")


def find(d, suffix):
    r = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return r[0] if r else None


def bench_line(d):
    try:
        for ln in open(os.path.join(d, "bench.json")):
            if ln.startswith("{"):
                return json.loads(ln)
    except Exception:
        pass
    return None


def short(k):
    return k.split("(")[0].replace("void ", "").strip()


def counter_sums(d, counters):
    """{kernel: {counter: [launches, sum]}} of one PMC run"""
    p = find(d, "counter_collection.csv")
    acc = {}
    if not p:
        return acc
    for r in csv.DictReader(open(p)):
        c = r["Counter_Name"]
        if c not in counters:
            continue
        k = short(r["Kernel_Name"])
        if not any(o in k for o in OURS):
            continue
        a = acc.setdefault(k, {}).setdefault(c, [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc


def main():
    tag, stats_dir, fetch_dir, write_dir, sq_dir = sys.argv[1:6]
    sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
    from bwamem_hip.lib import sources_sha16
    # the build the counters belong to: bench.py quotes a counter of this file only while the library's sources still hash to this
    out = {"tag": tag, "lib_sources_sha16": sources_sha16(), "kernels": {}, "families": {}}
    # ---- kernel stats of the --kernel-trace --stats run
    p = find(stats_dir, "kernel_stats.csv")
    rows = list(csv.reader(open(p)))
    hdr = rows[0]
    keep = [hdr] + [r for r in rows[1:] if any(o in r[0] for o in OURS)]
    with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_bench.csv"), "w", newline="") as f:
        csv.writer(f).writerows(keep)
    iname, icalls, itot = hdr.index("Name"), hdr.index("Calls"), hdr.index("TotalDurationNs")
    bl = bench_line(stats_dir)
    if bl:
        out["workload_key"] = bl["config"]["workload_key"]
        if bl.get("lib_sources_sha16") and bl["lib_sources_sha16"] != out["lib_sources_sha16"]:
            print("WARNING: the stats run's library sources differ from the tree's:", bl["lib_sources_sha16"], out["lib_sources_sha16"])
        out["stats_run"] = {"passes": bl.get("passes"), "value": bl["value"], "ms_per_step": bl["ms_per_step"], "steps": bl["steps"], "warmup": bl["warmup"]}
    for r in keep[1:]:
        out["kernels"].setdefault(short(r[iname]), {}).update(calls=int(r[icalls]), total_ms=round(float(r[itot]) / 1e6, 3),
                                                                avg_ms=round(float(r[itot]) / 1e6 / max(int(r[icalls]), 1), 4))
    # ---- counters
    runs = {"fetch": (fetch_dir, ("FETCH_SIZE",)), "write": (write_dir, ("WRITE_SIZE",)), "sq": (sq_dir, SQ)}
    sums, passes = {}, {}
    for name, (d, ctrs) in runs.items():
        b = bench_line(d)
        passes[name] = (b or {}).get("passes")
        if b and out.get("workload_key") and b["config"]["workload_key"] != out["workload_key"]:
            print("WARNING: workload differs in", d)
        sums[name] = counter_sums(d, ctrs)
        for k, cs in sums[name].items():
            for c, (n, s) in cs.items():
                out["kernels"].setdefault(k, {})[c] = {"launches": n, "sum": s, "avg_per_launch": round(s / n, 2)}
    out["pmc_passes"] = passes
    for fam, (subs, pk) in FAMILIES.items():
        o = {"kernels": sorted(k for k in out["kernels"] if any(s in k for s in subs)), "per": f"pass of the {pk} stage"}
        if bl and bl.get("passes"):
            tot = sum(out["kernels"][k].get("total_ms", 0.0) for k in o["kernels"])
            o["rocprof_ms_per_launch"] = round(tot / bl["passes"][pk], 4)
        for name, ctr, field, scale in (("fetch", "FETCH_SIZE", "fetch_bytes_per_launch", 1024.0), ("write", "WRITE_SIZE", "write_bytes_per_launch", 1024.0),
                                        ("sq", "SQ_INSTS_VALU", "valu_wave_instr_per_launch", 1.0), ("sq", "SQ_ACTIVE_INST_VALU", "active_inst_valu_per_launch", 1.0),
                                        ("sq", "SQ_WAVES", "waves_per_launch", 1.0)):
            if not passes.get(name):
                continue
            s = sum(cs[ctr][1] for k, cs in sums[name].items() if any(x in k for x in subs) and ctr in cs)
            o[field] = int(s * scale / passes[name][pk])
        if "fetch_bytes_per_launch" in o and "write_bytes_per_launch" in o:
            o["hbm_bytes_per_launch"] = o["fetch_bytes_per_launch"] + o["write_bytes_per_launch"]
        out["families"][fam] = o
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
    print("kernels:", len(keep) - 1, "families:", {k: v.get("hbm_bytes_per_launch") for k, v in out["families"].items()})


if __name__ == "__main__":
    main()
