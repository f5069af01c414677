#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_round.sh (gpurun_out/<tag>_*) into the committed summaries
profiles/<tag>_bench_50kx30k_kernel_stats.csv and profiles/<tag>_bench_50kx30k_pmc.csv (library kernels only).

usage: python tools/summarize_profiles.py <tag> "<one-line description of the profiled build>"
"""
import csv
import glob
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, note = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
OUT = os.path.join(ROOT, "profiles")
ours = lambda name: "vcy::" in name or name.startswith("void k_") or " k_" in name.split("(")[0]


def find(sub, suffix):
    hits = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_{sub}", "**", f"*{suffix}"), recursive=True)
    if not hits:
        raise SystemExit(f"no {suffix} under gpurun_out/{tag}_{sub}")
    return max(hits, key=os.path.getmtime)          # gpurun merges into gpurun_out/: an older pass of the same tag may still be there


def condense(sfx, dtype_name):
    """One arithmetic type's passes -> (kernel-stats rows, per-(counter, kernel) averages)."""
    rows = list(csv.reader(open(find("stats" + sfx, "kernel_stats.csv"))))
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for sub in ("fetch", "write", "sq", "grbm"):
        try:
            path = find(sub + sfx, "counter_collection.csv")
        except SystemExit:
            if sub == "grbm":
                continue
            raise
        with open(path) as f:
            for r in csv.DictReader(f):
                if not ours(r["Kernel_Name"]):
                    continue
                a = acc[(r["Counter_Name"], r["Kernel_Name"])]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return rows, acc


import json
cells, genes, nrndm = 50000, 30000, 250                       # the default workload of bench.py, which profile_round.sh runs
counters = {}
stats_out, pmc_out = [], []
for sfx, dname, ctype, chunk in (("", "f32", "float", 1536), ("_f64", "f64", "double", 1024)):
    try:
        rows, acc = condense(sfx, dname)
    except SystemExit as e:
        print("skipping", dname, "-", e)
        continue
    if not stats_out:
        stats_out.append(rows[0] + ["bench_dtype"])
    stats_out += [r + [dname] for r in rows[1:] if ours(r[0])]
    pmc_out += [[cn, kn, n, v / n, d / n, dname] for (cn, kn), (n, v, d) in sorted(acc.items())]
    # ---- the dominant kernel's per-launch figures in the form bench.py reads (roofline.counters_from)
    tagk = f"k_cdc_partial_grouped<{ctype}"
    cdc = {cn: v / n for (cn, kn), (n, v, d) in acc.items() if tagk in kn}
    dur = {cn: d / n for (cn, kn), (n, v, d) in acc.items() if tagk in kn}
    if "SQ_INSTS_VALU" not in cdc:
        continue
    pair_chunks = cells * nrndm * ((genes + chunk - 1) // chunk)
    kname = next(kn for (cn, kn) in acc if tagk in kn and cn == "SQ_INSTS_VALU")
    rule = {"1": "partial (literal)", "2": "partial, pseudocount dropped (VCY_RULES_PARTIAL_NOPSC)", "0": "full"}.get(kname.split("<")[1].split(",")[2].strip(), "?")
    wc = cdc.get("SQ_WAVE_CYCLES")
    # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (each has its own GRBM; a single-XCD reading would be 18 "GHz"): cycles per ns per XCD
    ghz = cdc["GRBM_GUI_ACTIVE"] / dur["GRBM_GUI_ACTIVE"] / 8.0 if "GRBM_GUI_ACTIVE" in cdc else None
    counters[dname] = {
        "profile": f"profiles/{tag}_bench_50kx30k_pmc.csv", "kernel": kname.split("(")[0].replace("void vcy::", ""), "rules": int(kname.split("<")[1].split(",")[2]),
        "rule": rule,
        "workload": {"cells": cells, "genes": genes, "nrndm": nrndm, "genes_per_chunk": chunk, "pair_chunks_per_launch": pair_chunks},
        "SQ_INSTS_VALU_per_launch": cdc["SQ_INSTS_VALU"], "valu_insts_per_pair_chunk": cdc["SQ_INSTS_VALU"] / pair_chunks,
        "valu_insts_per_pair_gene": cdc["SQ_INSTS_VALU"] / (float(cells) * nrndm * genes),
        "FETCH_SIZE_KiB": cdc.get("FETCH_SIZE"), "WRITE_SIZE_KiB": cdc.get("WRITE_SIZE"),
        "hbm_bytes_per_launch": (2 * cdc["FETCH_SIZE"] + cdc["WRITE_SIZE"]) * 1024 if "FETCH_SIZE" in cdc and "WRITE_SIZE" in cdc else None,
        "profiled_launch_ms": dur["SQ_INSTS_VALU"] / 1e6,
        "GRBM_GUI_ACTIVE_per_launch": cdc.get("GRBM_GUI_ACTIVE"), "grbm_pass_launch_ms": dur["GRBM_GUI_ACTIVE"] / 1e6 if ghz else None,
        "effective_clock_ghz": ghz,
        # where a wave's time goes (fractions of SQ_WAVE_CYCLES; the three are disjoint, MI355X_MICROARCH.md counters table)
        "wave_time": {k: (cdc[c] / wc if wc and c in cdc else None) for k, c in
                      (("parked_at_waitcnt_or_barrier", "SQ_WAIT_ANY"), ("waiting_to_issue", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY"))},
        "note": "counters of ONE launch under rocprofv3 --pmc (separate passes for FETCH_SIZE, WRITE_SIZE, the SQ set and GRBM_GUI_ACTIVE); HBM-side read "
                "bytes = 2 x FETCH_SIZE x 1024 on gfx950 (MI355X_MICROARCH.md, HBM); effective_clock_ghz = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration "
                "of that launch (the chip clocks to its power budget: MI355X_MICROARCH.md, DVFS)"}

with open(os.path.join(OUT, f"{tag}_bench_50kx30k_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow([f"# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extra --dtype f32|f64 (50k cells x 30k genes, 1 warmup + 3 timed "
                f"steps each; last column = which run): {note}; library kernels only (torch's own elementwise/index kernels of the harness are omitted)"])
    w.writerows(stats_out)
with open(os.path.join(OUT, f"{tag}_bench_50kx30k_pmc.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow([f"# {note}. Separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_* set | GRBM_GUI_ACTIVE), each with --kernel-trace, of: python bench.py "
                "--no-cpu-baseline --no-extra --dtype f32|f64 --steps 1 --warmup 0 (tools/profile_round.sh). FETCH/WRITE unit = KiB per dispatch; gfx950 correction: "
                "HBM-side read bytes = 2*FETCH_SIZE*1024 (calibrated on k_velocity_chain in round-1 profiles). SQ_* summed over the dispatch; "
                "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count in units of 4 clocks; GRBM_GUI_ACTIVE in shader clocks."])
    w.writerow(["Counter", "Kernel", "Dispatches", "AvgCounterPerDispatch", "AvgDurationNs", "bench_dtype"])
    w.writerows(pmc_out)
if counters:
    with open(os.path.join(OUT, f"{tag}_cdc_counters.json"), "w") as f:
        json.dump(counters, f, indent=1)
src = os.path.join(ROOT, "gpurun_out", f"{tag}_bench_line.json")
if os.path.exists(src):
    shutil.copy(src, os.path.join(OUT, f"{tag}_bench_line.json"))
print("wrote profiles/%s_*" % tag)
