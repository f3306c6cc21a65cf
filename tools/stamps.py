"""diagnostic: where a wave of the MFMA kernel spends its cycles (D2D_DBG=16 build path)"""
import ctypes as C, os, sys, subprocess
os.environ["D2D_DBG"] = "16"
sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--distinct", "8"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import runpy
import dsd2dxd_amd as d
L = d.lib()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
out = (C.c_ulonglong * 8)()
L.d2d_debug_stamps(out)
names = ["head: wait prefetch + LDS writes", "prefetch issue + sync", "chains (all pairs)", "epilogue", "store"]
tot = sum(out[:5])
for n, v in zip(names, out[:5]):
    print("%-36s %6.1f %%  %.3e wave-cycles" % (n, 100.0 * v / max(tot, 1), v))
