#!/usr/bin/env python3
"""Can the FIR kernel read its DSD from, and write its frames to, PINNED HOST memory directly (no staging copies)?
   tools/zero_copy_probe.py [files] [seconds]   (GPU box; prints ms per step of the bench shape for the four placements)
A pinned host allocation is device-addressable under HIP's unified addressing, so d2d_translate_batch_device takes it as it is."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dsd2dxd_amd as d
from bench import make_files, DSD64

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
blocks = max(1, int(round(seconds * DSD64 / 8 / 4096)))
bpc = blocks * 4096
kw = dict(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="T", seed=206)
files = make_files(n_files, bpc, 1, n_files, 0, 16)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
ref = None
for in_host, out_host in ((False, False), (True, False), (False, True), (True, True)):
    eng = d.Engine(n_files=n_files, kernel=d.KERNEL_AUTO, device=0, **kw)
    frames = eng.next_frames(bpc)
    fb = eng.frame_bytes
    ins = [torch.from_numpy(b).pin_memory() if in_host else torch.from_numpy(b).to(dev) for b in files]
    out = torch.empty((n_files, (frames * fb + 31) // 16 * 16), dtype=torch.uint8)
    out = out.pin_memory() if out_host else out.to(dev)
    ios = (d.FileIO * n_files)()
    for f in range(n_files):
        ios[f].dsd = ins[f].data_ptr()
        ios[f].bytes_per_channel = bpc
        ios[f].pcm = out[f].data_ptr()
        ios[f].pcm_capacity_bytes = frames * fb
    eng.translate_batch_device(ios, stream)
    torch.cuda.synchronize()
    got = out.cpu()
    if ref is None:
        ref = got
    same = bool(torch.equal(ref, got))
    eng.reset()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        eng.reset()
        eng.translate_batch_device(ios, stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    up, down = n_files * bpc * 2, n_files * frames * fb
    link = (up if in_host else 0) + (down if out_host else 0)
    print("DSD in %-6s frames in %-6s  %8.3f ms per step  %6.1f GB/s over the link  same frames: %s" %
          ("host" if in_host else "HBM", "host" if out_host else "HBM", dt * 1e3, link / dt / 1e9, same), flush=True)
    del eng, ins, out
