"""diagnostic: per-wave lifetimes, phase shares and in-kernel clock of the pipelined kernels (a stamps build from tools/ab_build.sh <name> mx -DD2D_MX_STAMPS=1 or ... mfma3 -DD2D_M3_STAMPS=1, selected with D2D_AMD_LIB)"""
import ctypes as C, os, sys
base = int(os.environ.get("D2D_DBG", "0"))
os.environ["D2D_DBG"] = str(base | 256)
sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--reps", "1", "--no-cpu-baseline", "--no-pcie", "--sustain", "0"] + sys.argv[1:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import runpy
import dsd2dxd_amd as d
L = d.lib()
out = (C.c_ulonglong * 8)()
L.d2d_debug_stamps3(out)     # reset
try:
    runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
except SystemExit:
    pass
L.d2d_debug_stamps3(out)
mn, mx, sm, n = out[0], out[1], out[2], out[3]
tot = max(sm, 1)
print("share of wave time: staging %.1f %%  regions %.1f %%  after-region %.1f %%" % (100.0 * out[4] / tot, 100.0 * out[5] / tot, 100.0 * out[6] / tot))
print("core clock while the waves ran: %.3f GHz (s_memtime / s_memrealtime)" % (0.1 * sm / max(out[7], 1)))
print("waves %d  lifetime ticks: min %d  avg %.0f  max %d   avg/max %.3f  min/max %.3f" % (n, mn, sm / max(n, 1), mx, sm / max(n, 1) / max(mx, 1), mn / max(mx, 1)))
