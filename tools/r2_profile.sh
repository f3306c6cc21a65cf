#!/bin/bash
# round-2 evidence run (on the GPU box): rocprofv3 trace + PMC passes of the default bench command, the default bench
# line itself (with CPU baseline and pcie_inclusive), the other workloads, the multi-rank rehearsal
bash tools/prof.sh r02 --steps 5 --warmup 2 --reps 1 --no-pcie > gpurun_out/prof_r02_console.txt 2>&1
python3 - <<'PY'
import csv, glob, json, os, sys
sys.path.insert(0, ".")
import bench
out = "gpurun_out/prof_r02"
vals = {}
for i in (3, 4):
    for f in glob.glob(os.path.join(out, f"pmc{i}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "d2d_fir" in row["Kernel_Name"]:
                vals.setdefault((row["Kernel_Name"], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
res = {}
for (k, c), v in vals.items():
    res.setdefault(k, {})[c] = sum(v) / len(v)
ent = {}
for k, cs in res.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        name = k.replace("void d2d::", "").split("(")[0]
        if name.startswith("d2d_fir_mfma2_kernel<") and name.count(",") == 3:      # the engine's name omits the epilogue flavour (4th argument)
            name = name[:name.rindex(",")] + ">"
        # FETCH_SIZE / WRITE_SIZE are in KiB... (rocprofv3 reports kilobytes); gfx950: FETCH_SIZE counts half of a wide streaming read
        hbm = int((2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024)
        ent[name] = {"dsd64_to_88k2_s24_stereo": {"hbm_bytes_per_launch": hbm, "fetch_size_kb": cs["FETCH_SIZE"], "write_size_kb": cs["WRITE_SIZE"],
                     "files_per_gpu": 64, "seconds_per_file": 60.0, "kernel_src_sha16": bench.kernel_source_hash(),
                     "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x 2 per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)"}}
json.dump(ent, open("gpurun_out/pmc_traffic_r02.json", "w"), indent=1)
print(json.dumps(ent, indent=1))
PY
timeout -k 10 600 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; echo "default rc=$?"
bash tools/r2_workloads.sh r02_workloads
bash tools/r2_bench_ranks.sh
