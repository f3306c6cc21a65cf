"""diagnostic: where a wave of the two-group MFMA kernel spends its cycles (make DIAG=1; D2D_DBG gets bit 256)"""
import ctypes as C, os, sys
base = int(os.environ.get("D2D_DBG", "0"))
os.environ["D2D_DBG"] = str(base | 256)
sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--distinct", "4"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import runpy
import dsd2dxd_amd as d
L = d.lib()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
out = (C.c_ulonglong * 8)()
L.d2d_debug_stamps2(out)
names = ["issue next loads + sync", "chain", "epilogue", "wait loads + LDS writes", "flush stores"]
tot = sum(out[:5])
units = 5 * 64 * 2 * 5292032 / 512      # launches x units
for n, v in zip(names, out[:5]):
    print("dbg %d %-44s %6.1f %%  %.0f cycles per unit and wave" % (base, n, 100.0 * v / max(tot, 1), v / units))
