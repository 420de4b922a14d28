#!/usr/bin/env python3
"""Condenses the CSVs written by tools/profile_bench.sh into one JSON + markdown summary.

usage: tools/summarize_prof.py <profdir> <out_prefix> [kernel-name-substring]
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_sha import git_head, kernel_sources_sha  # noqa: E402


def find(profdir, name, suffix):
    hits = glob.glob(os.path.join(profdir, name, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def main():
    profdir, out = sys.argv[1], sys.argv[2]
    pick = sys.argv[3] if len(sys.argv) > 3 else "k_pairs"
    # which tree the numbers belong to: bench.py quotes a profile's counters only while the kernel
    # sources still hash to this value (the GPU box has no .git: the head comes in as GRAFT_HEAD)
    summary = {"kernel_filter": pick, "kernel_sources_sha": kernel_sources_sha(),
               "git_head": os.environ.get("GRAFT_HEAD") or git_head()}
    # the bench line the kernel-trace pass itself printed: its ms_per_step next to the traced kernel
    # time tells how much of a difference to another box's bench line is the box (clock), and how
    # much the tracing
    log = os.path.join(profdir, "trace.log")
    if os.path.exists(log):
        for text in reversed(open(log).read().splitlines()):
            if text.startswith('{"metric"'):
                line = json.loads(text)
                summary["bench_line_of_trace_pass"] = {
                    "ms_per_step": line["ms_per_step"], "kernel_ms": line["roofline"]["kernel_ms"],
                    "prepare_ms": line["roofline"]["prepare_ms"], "value": line["value"], "steps": line["steps"],
                    "roofline_frac": line["roofline"]["frac"]}
                break
    stats = find(profdir, "trace", "_kernel_stats.csv")
    rows = list(csv.DictReader(open(stats)))
    summary["kernel_stats"] = [
        {"name": r["Name"][:110], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
         "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}
        for r in rows[:12]]
    dom = [r for r in summary["kernel_stats"] if pick in r["name"]]
    summary["dominant"] = dom[0] if dom else None
    counters = {}
    for name in ("pmc_valu", "pmc_wait", "pmc_mem", "pmc_fetch", "pmc_write"):
        f = find(profdir, name, "_counter_collection.csv")
        if not f:
            continue
        per_dispatch = {}
        for r in csv.DictReader(open(f)):
            if pick not in r["Kernel_Name"]:
                continue
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(r["Counter_Value"])
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            counters.setdefault(name + ":kernel_ms", dur)
        for (d, c), v in per_dispatch.items():
            counters[c] = v  # one dispatch per counter pass (steps=1, warmup=0)
    summary["counters_per_launch"] = counters
    d = {}
    if "SQ_INSTS_VALU" in counters:
        f64 = sum(counters.get(k, 0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
                                               "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
        d["valu_wave_insts"] = counters["SQ_INSTS_VALU"]
        d["fp64_wave_insts"] = f64
        d["fp64_share_of_valu_insts"] = f64 / counters["SQ_INSTS_VALU"]
        ms = counters.get("pmc_valu:kernel_ms")
        if ms and "GRBM_GUI_ACTIVE" in counters:
            clock_hz = counters["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3)
            d["shader_clock_ghz_from_grbm"] = clock_hz / 1e9
            # fp64 VALU pipe time: 4 cycles per add/mul/fma, 16 per transcendental, per SIMD
            pipe = 4 * (f64 - counters.get("SQ_INSTS_VALU_TRANS_F64", 0)) + \
                16 * counters.get("SQ_INSTS_VALU_TRANS_F64", 0)
            d["fp64_pipe_busy_frac"] = pipe / (1024 * counters["GRBM_GUI_ACTIVE"] / 8)
        if "SQ_ACTIVE_INST_VALU" in counters and "SQ_BUSY_CYCLES" in counters:
            d["valu_active_quadcycles_per_simd_busy"] = \
                4 * counters["SQ_ACTIVE_INST_VALU"] / (1024 * counters["GRBM_GUI_ACTIVE"] / 8)
    if "FETCH_SIZE" in counters:
        # gfx950: FETCH_SIZE under-counts wide coalesced reads by 2x; both raw and doubled given
        d["hbm_fetch_bytes_raw"] = counters["FETCH_SIZE"] * 1024
        d["hbm_fetch_bytes_x2"] = counters["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in counters:
        d["hbm_write_bytes"] = counters["WRITE_SIZE"] * 1024
    summary["derived"] = d
    json.dump(summary, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as f:
        f.write(f"# rocprofv3 summary ({os.path.basename(out)})\n\n")
        f.write(f"tree: git {summary['git_head']}, kernel sources sha {summary['kernel_sources_sha']}\n\n")
        b = summary.get("bench_line_of_trace_pass")
        if b:
            f.write(f"bench line of the kernel-trace pass on this box: {b['ms_per_step']:.3f} ms per step "
                    f"({b['steps']} steps), pair kernel {b['kernel_ms']:.3f} ms + binning {b['prepare_ms']:.3f} ms by "
                    f"the library's own HIP events, roofline.frac {b['roofline_frac']:.4f}\n\n")
        f.write("## kernel-trace --stats\n\n")
        f.write("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|\n")
        for r in summary["kernel_stats"]:
            f.write(f"| `{r['name']}` | {r['calls']} | {r['avg_ms']:.4f} | {r['total_ms']:.3f} | {r['pct']:.3f} |\n")
        f.write("\n## counters, one launch of the dominant kernel\n\n| counter | value |\n|---|---|\n")
        for k, v in counters.items():
            f.write(f"| {k} | {v:.6g} |\n")
        f.write("\n## derived\n\n| quantity | value |\n|---|---|\n")
        for k, v in d.items():
            f.write(f"| {k} | {v:.6g} |\n")
    print(json.dumps(summary["dominant"]), json.dumps(d))


if __name__ == "__main__":
    main()
