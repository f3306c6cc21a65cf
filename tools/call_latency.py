"""Per-call cost of the host-pointer entry point at different block sizes (what a caller that feeds the engine block by block, like the
reference's translate() loop, would see): pageable buffers (staged through device memory) and pinned ones (addressed by the kernels)."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import dsd2dxd_amd as d

kw = dict(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="T", seed=1)
for pinned in (False, True):
    for blocks in (1, 4, 16, 64, 256, 1024):
        e = d.Engine(n_files=1, kernel=d.KERNEL_AUTO, **kw)
        buf = torch.from_numpy(np.random.default_rng(0).integers(0, 256, size=4096 * blocks * 2, dtype=np.uint8))
        out = torch.zeros(e.next_frames(4096 * blocks) * e.frame_bytes + 64, dtype=torch.uint8)
        if pinned:
            buf, out = buf.pin_memory(), out.pin_memory()
        for _ in range(5):
            e.translate_into(buf.data_ptr(), 4096 * blocks, out.data_ptr(), out.numel())
        n = max(5, 2000 // blocks)
        t = time.perf_counter()
        for _ in range(n):
            e.translate_into(buf.data_ptr(), 4096 * blocks, out.data_ptr(), out.numel())
        dt = (time.perf_counter() - t) / n
        audio = 4096 * blocks * 8 / 2822400.0
        print("%-8s blocks/call %5d  %8.1f us/call  %9.0fx real time  %7.1f Msamples/s" % ("pinned" if pinned else "pageable", blocks, dt * 1e6, audio / dt, 4096 * blocks * 8 / 32 * 2 / dt / 1e6), flush=True)
