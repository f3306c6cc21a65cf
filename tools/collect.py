#!/usr/bin/env python3
"""File one tools/prof.sh run under profiles/:  tools/collect.py <tag> <name>   (from the repo root, after gpurun merged gpurun_out/)
  profiles/<name>_summary.txt, <name>_kernel_stats.csv, <name>_bench_under_rocprof_trace.json, and an entry in profiles/pmc_traffic.json:
  HBM bytes per launch of every engine kernel of the step (FETCH_SIZE x 2 per MI355X_MICROARCH.md -- gfx950 tallies 128-B requests at
  64 B -- plus WRITE_SIZE, both in KiB), under the FIR kernel's name and under "step", stamped with the hash of the kernel sources."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
prof = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(prof, name + "_summary.txt"))
ks = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True))
if ks:
    shutil.copy(ks[-1], os.path.join(prof, name + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(prof, name + "_bench_under_rocprof_trace.json"))
line = [l for l in open(os.path.join(src, "bench_trace.json")).read().splitlines() if l.startswith("{")][-1]
bench = json.loads(line)
cfg = bench["config"]["workload"]
workload = cfg.split(":")[0]
# a bench line taken with modifiers gets its own key (bench.py only ever cites the plain workload names)
if "32-bit grid" in cfg:
    workload += "+taps32"
if ", level " in cfg:
    workload += "+level" + cfg.split(", level ")[1].split(" dB")[0]
fetch, write = defaultdict(list), defaultdict(list)
for i, dst, cname in ((3, fetch, "FETCH_SIZE"), (4, write, "WRITE_SIZE")):
    for f in glob.glob(os.path.join(src, f"pmc{i}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == cname and row["Kernel_Name"].startswith(("void d2d::", "d2d::")):
                dst[row["Kernel_Name"].replace("void ", "").split("(")[0]].append(float(row["Counter_Value"]))
from bench import kernel_source_hash
per_kernel = {}
for k in sorted(set(fetch) | set(write)):
    fk = sum(fetch[k]) / len(fetch[k]) if fetch[k] else 0.0
    wk = sum(write[k]) / len(write[k]) if write[k] else 0.0
    per_kernel[k] = {"fetch_size_kb": fk, "write_size_kb": wk, "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024)}
step_bytes = sum(v["hbm_bytes_per_launch"] for v in per_kernel.values())
common = {"files_per_gpu": bench["config"]["files_per_gpu"], "seconds_per_file": bench["config"]["seconds_per_file"],
          "kernel_src_sha16": kernel_source_hash(),
          "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/prof.sh), FETCH_SIZE x 2 per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)",
          "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"]}
path = os.path.join(prof, "pmc_traffic.json")
cur = json.load(open(path))
fir = bench["config"]["kernel"]
fir_key = next((k for k in per_kernel if k.replace("d2d::", "") == fir), None)
if fir_key:
    cur.setdefault(fir, {})[workload] = dict(per_kernel[fir_key], **common)
cur.setdefault("step", {})[workload] = dict(common, hbm_bytes_per_launch=step_bytes, kernels=per_kernel)
json.dump(cur, open(path, "w"), indent=1)
print(name, workload, "step traffic %.3f GB = %.2f x algorithmic" % (step_bytes / 1e9, step_bytes / common["algorithmic_bytes_per_launch"]))
for k, v in per_kernel.items():
    print("   %-70s %.3f GB" % (k, v["hbm_bytes_per_launch"] / 1e9))
