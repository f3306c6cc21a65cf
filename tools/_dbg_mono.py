import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import dsd2dxd_amd as d
from helpers import random_bytes, pack_layout, synth
sys.path.insert(0, '/root/repo/oracle')
import oracle as om
nbytes = 4096 * 3 + 52
for dither, bits, src in (("T", 24, "rnd"), ("X", 24, "rnd"), ("R", 16, "rnd"), ("T", 24, "sine"), ("X", 32, "rnd")):
    x = random_bytes(nbytes, 100) if src == "rnd" else synth("sine", nbytes, seed=3)
    kw = dict(dsd_rate=1, output_rate=88200, channels=1, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=bits, dither=dither, seed=5)
    e = d.Engine(kernel=2, **kw); o = om.Oracle(**kw)
    sb = o.frame_bytes
    for a, b in ((0, 4096), (4096, 4096 * 3)):
        g, gf = e.translate(x[a:b]); r, rf = o.translate(x[a:b])
        r = r[:rf * sb]
        bad = np.nonzero(g != r)[0]
        smp = sorted(set((bad // sb).tolist()))
        print(dither, bits, src, a, b, e.kernel_name(), gf, rf, len(smp), smp[:10], [int.from_bytes(bytes(g[s*sb:(s+1)*sb]), 'little') for s in smp[:4]], [int.from_bytes(bytes(r[s*sb:(s+1)*sb]), 'little') for s in smp[:4]], e.peak(0) == o.peak(0))
