#!/usr/bin/env python3
"""Stage B of the 48k cascade (DSD256 / DSD512 input) alone: device time per output for a few shapes (GPU box, repo root)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dsd2dxd_amd as d

def run(dsd_rate, out_rate, channels, files, seconds, debug=0):
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=channels, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="T", seed=1)
    e = d.Engine(n_files=files, kernel=2, debug=debug, **kw)
    n = int(seconds * 2822400 * dsd_rate / 8 / 4096) * 4096
    buf = torch.randint(0, 256, (n * channels,), dtype=torch.uint8, device="cuda")
    frames = e.next_frames(n)
    out = torch.zeros((files, (frames * e.frame_bytes + 31) // 16 * 16), dtype=torch.uint8, device="cuda")
    ios = (d.FileIO * files)()
    for i in range(files):
        ios[i].dsd = buf.data_ptr(); ios[i].bytes_per_channel = n
        ios[i].pcm = out[i].data_ptr(); ios[i].pcm_capacity_bytes = frames * e.frame_bytes
    for it in range(3):
        e.reset(); e.profile_enable(True)
        e.translate_batch_device(ios); torch.cuda.synchronize()
        fir, step, _ = e.profile_read_all()
    outs = frames * channels * files
    print(f"DSD{64*dsd_rate}->{out_rate} {channels}ch x{files} files {seconds}s debug={debug}: stage A {fir:.2f} ms, rest {step-fir:.2f} ms, {(step-fir)*1e6/outs:.2f} ps per output, kernel {e.kernel_name()}")

for args in [(8, 96000, 2, 64, 30), (8, 96000, 8, 16, 30), (4, 96000, 2, 64, 30), (4, 192000, 2, 64, 30), (4, 96000, 8, 16, 30), (8, 96000, 4, 32, 30)]:
    run(*args)
