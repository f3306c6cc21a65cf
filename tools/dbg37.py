import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import dsd2dxd_amd as d
from oracle import oracle as O
from helpers import pack_layout, random_bytes
import test_gpu_fuzz as t
kw, total, cuts = t._case(37)
chans = [random_bytes(total, 1000 * 37 + c) for c in range(kw["channels"])]
bufs = [pack_layout([ch[a:b] for ch in chans], kw["fmt"], kw["block_size"]) if b > a else np.zeros(0, np.uint8) for a, b in zip(cuts[:-1], cuts[1:])]
for blk in (7, 4096):
    kw2 = dict(kw); kw2["block_size"] = blk
    bufs = [pack_layout([ch[a:b] for ch in chans], kw2["fmt"], blk) if b > a else np.zeros(0, np.uint8) for a, b in zip(cuts[:-1], cuts[1:])]
    o = O.Oracle(**kw2); e = d.Engine(kernel=2, **kw2)
    for b in bufs:
        r, rf = o.translate(b); g, gf = e.translate(b)
        print(blk, len(b), rf, gf, np.array_equal(r[:rf*6], g[:gf*6]), o.peak_dbfs(), e.peak_dbfs(), e.kernel_name())
