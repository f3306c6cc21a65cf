#!/usr/bin/env python3
"""Markdown rows for DESIGN.md section 7 from a round's filed bench lines:  tools/bench_table.py r04
(profiles/<round>_bench_default.json, profiles/<round>_bench_other_workloads.json; traffic from the lines themselves)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
lines = []
for name in (f"{rnd}_bench_default.json", f"{rnd}_bench_other_workloads.json"):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        lines += [json.loads(l) for l in f.read().splitlines() if l.startswith("{")]
print("| workload | kernel (the step's FIR kernel) | Gsamples/s | device ms per step (FIR kernel) | frac of 8 TB/s | traffic |")
print("|---|---|---|---|---|---|")
for d in lines:
    c, r = d["config"], d["roofline"]
    w = c["workload"].split(":")[0]
    mods = []
    if "32-bit grid" in c["workload"]: mods.append("32-bit taps")
    if ", level " in c["workload"]: mods.append("level " + c["workload"].split(", level ")[1].split(" dB")[0] + " dB")
    if c.get("as_rank"): mods.append("rank " + c["as_rank"]["rank_of"] + " of a channel split, alone")
    if c.get("files_per_gpu") not in (None, 64): mods.append(f"{c['files_per_gpu']} files")
    tr = r.get("traffic")
    alg = r.get("algorithmic_bytes_per_launch")
    trs = f"{tr / alg:.2f}x" if tr and alg and r.get("scope") == "kernel" else (f"{tr / 1e9:.1f} GB" if tr else "")
    print(f"| {w}{' (' + ', '.join(mods) + ')' if mods else ''} | `{c.get('kernel', r.get('kernel'))}` | {d['value'] / 1e3:.0f} | {r['step_kernels_ms']:.2f} ({r['fir_kernel_ms']:.2f}) | {r['frac']:.3f} ({r['scope']}) | {trs} |")
