#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: output PCM Msamples/s (+ achieved HBM GB/s) for
DSD64 -> 88.2 kHz stereo, one process per GPU.

A "step" = one pass of the hot path (unpack -> FIR decimate -> dither -> 24-bit pack) over this rank's
shard of synthetic DSD files, device-resident in, device-resident out.  Weak scaling: every rank
converts `--files` files of `--seconds` seconds (default 64 x 60 s = config 4's 512-file batch cut
8 ways); files are independent, so there is no data-path collective -- RCCL only broadcasts the
filter tables once, before the timed region.

`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself (child
processes, one per GPU, rendezvous on 127.0.0.1) before anything touches the GPU; under
`torch.distributed.run` the ranks already exist and it just joins them.

Prints ONE JSON line (rank 0).  The timed loop of K steps is repeated `--reps` times; `value` and
`ms_per_step` are the median repetition, `repetitions` lists them all.  `roofline` is timed with HIP
events on the launch stream inside the timed region: around the FIR launch when that kernel is the
whole step, around every kernel of the step otherwise (`roofline.scope`).  `cpu_baseline` is the repo's
own CPU restatement (oracle/, "port" -- the reference's Rust core is absent from the checkout and cannot
be built here) on a bounded sample; `pcie_inclusive` is the same batch from pinned host memory.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec
DSD64 = 2822400

WORKLOADS = {
    # name: (dsd_rate, out_rate, bit_depth, dither, channels, algorithmic bytes per output sample)
    "dsd64_to_88k2_s24_stereo": (1, 88200, 24, "T", 2, 32 / 8 + 3),
    "dsd64_to_88k2_s24_stereo_nodither": (1, 88200, 24, "X", 2, 32 / 8 + 3),
    "dsd64_to_88k2_s24_6ch": (1, 88200, 24, "T", 6, 32 / 8 + 3),          # a 5.1 stream, planar
    "dsd64_to_88k2_s24_4ch": (1, 88200, 24, "T", 4, 32 / 8 + 3),          # a quad stream, planar
    "dsd512_to_352k8_s24_8ch": (8, 352800, 24, "T", 8, 64 / 8 + 3),       # DSD512's one documented output rate (main.rs:90), eight channels byte-interleaved MSB-first
    "dsd64_to_88k2_s24_mono": (1, 88200, 24, "T", 1, 32 / 8 + 3),         # six of the reference's eleven fixtures are mono (test/1kHz_mono_p.dsf ...)
    "dsd64_to_352k8_s24_mono": (1, 352800, 24, "T", 1, 8 / 8 + 3),        # run_all_tests.sh:8 / build_test_mono.sh convert mono at +4 dB (bench.py --level 4)
    "dsd64_to_88k2_s16_stereo": (1, 88200, 16, "T", 2, 32 / 8 + 2),
    "dsd64_to_88k2_f32_stereo": (1, 88200, 32, "X", 2, 32 / 8 + 4),
    "dsd64_to_176k4_s24_stereo": (1, 176400, 24, "T", 2, 16 / 8 + 3),
    "dsd64_to_352k8_s24_stereo": (1, 352800, 24, "T", 2, 8 / 8 + 3),
    "dsd64_to_352k8_f32_stereo": (1, 352800, 32, "X", 2, 8 / 8 + 4),
    "dsd128_to_88k2_s24_stereo": (2, 88200, 24, "T", 2, 64 / 8 + 3),
    "dsd256_to_88k2_s24_stereo": (4, 88200, 24, "T", 2, 128 / 8 + 3),      # M = 128 (test_all_44k_mults.sh converts 1kHz_stereo_256.dsf to 88.2 kHz)
    "dsd256_to_176k4_s24_stereo": (4, 176400, 24, "T", 2, 64 / 8 + 3),
    "dsd128_to_88k2_s24_stereo_ns": (2, 88200, 24, "N", 2, 64 / 8 + 3),   # BASELINE config 3's noise-shaped variant (an extension)
    "dsd64_to_96k_s24_stereo": (1, 96000, 24, "T", 2, 29.4 / 8 + 3),
    "dsd64_to_96k_s24_6ch": (1, 96000, 24, "T", 6, 29.4 / 8 + 3),          # a 5.1 stream into the 48k family
    "dsd64_to_192k_s24_stereo": (1, 192000, 24, "T", 2, 14.7 / 8 + 3),
    "dsd128_to_384k_s24_stereo": (2, 384000, 24, "T", 2, 14.7 / 8 + 3),
    "dsd128_to_96k_s24_stereo": (2, 96000, 24, "T", 2, 58.8 / 8 + 3),
    "dsd256_to_192k_s24_stereo": (4, 192000, 24, "T", 2, 58.8 / 8 + 3),     # the reference README's other recommended setting (README.md:223-228)
    "dsd512_to_96k_s24_8ch": (8, 96000, 24, "T", 8, 235.2 / 8 + 3),      # config 5: byte-interleaved MSB-first
    "dsd64_to_88k2_s24_stereo_dff": (1, 88200, 24, "T", 2, 32 / 8 + 3),   # the reference CLI's default input format (-f I; DFF files): byte-interleaved MSB-first
    "dsd64_to_352k8_s24_stereo_dff": (1, 352800, 24, "T", 2, 8 / 8 + 3),  # ... at the CLI's default output rate
    "dsd64_to_96k_s24_stereo_dff": (1, 96000, 24, "T", 2, 29.4 / 8 + 3),  # ... into the 48k cascade
}
LAYOUTS = {"dsd512_to_96k_s24_8ch": ("I", "M", 1), "dsd512_to_352k8_s24_8ch": ("I", "M", 1), "dsd64_to_88k2_s24_stereo_dff": ("I", "M", 1),
           "dsd64_to_352k8_s24_stereo_dff": ("I", "M", 1), "dsd64_to_96k_s24_stereo_dff": ("I", "M", 1)}                # default: planar 4096 LSB-first


REF_SCREENSHOT_MSAMPLES_PER_WORKER = 17.6   # /root/reference/asset/progress.jpg: 50x realtime, stereo 176.4 kHz


def kernel_source_hash():
    """sha256[:16] over the kernel sources: pmc_traffic.json entries are only cited for the code they were taken on"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dsd2dxd_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name == "d2d_engine.cpp":
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def spawn_ranks(n, timeout_s=None):
    """--gpus N without a launcher: start N copies of this script, one per GPU, from a parent that never
    touches the GPU (a process that has initialised HIP must not be replaced or forked on this pool).  Rank 0's
    JSON line is the parent's output.  The parent watches ALL children: the first one that exits non-zero ends the
    run -- the others are terminated (they are fresh child processes; nothing is re-executed) -- and its code is the
    parent's; so does an overall timeout (D2D_BENCH_TIMEOUT seconds, default 1500).  The stderr tail of every failed
    rank is repeated on the parent's stderr."""
    import tempfile
    if timeout_s is None:
        timeout_s = float(os.environ.get("D2D_BENCH_TIMEOUT", "1500"))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, logs = [], []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        log = tempfile.TemporaryFile()
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=log))
    t0 = time.monotonic()
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = max(abs(c) for c in bad)
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() - t0 > timeout_s:
            sys.stderr.write(f"bench.py: ranks still running after {timeout_s:.0f} s, giving up\n")
            rc = 124
            break
        time.sleep(0.05)
    if rc:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    for r, (p, log) in enumerate(zip(procs, logs)):
        log.seek(0)
        text = log.read().decode(errors="replace")
        if p.returncode not in (0, None) or (rc and r == 0):
            sys.stderr.write(f"---- rank {r} (exit {p.returncode}) stderr tail ----\n" + "\n".join(text.splitlines()[-15:]) + "\n")
        elif r == 0:
            sys.stderr.write(text)
    out0.seek(0)
    for line in out0.read().decode(errors="replace").splitlines():      # rank 0's JSON line only (gloo chats on stdout)
        if line.startswith("{") and not rc:
            print(line)
    sys.stdout.flush()
    return rc


def stub_rank(args):
    """D2D_BENCH_STUB=1: the rank protocol without a GPU (tests/test_sharding_gloo.py drives spawn_ranks with it): rendezvous over gloo,
    a sleep as the step, MAX over ranks, one JSON line from rank 0.  D2D_BENCH_STUB_FAIL_RANK=r makes rank r die before the rendezvous."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("D2D_BENCH_STUB_FAIL_RANK", "") == str(rank):
        sys.stderr.write(f"stub rank {rank}: failing on request\n")
        sys.exit(7)
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (rank + 1))
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print("gloo-like chatter that is not JSON")
        print(json.dumps({"metric": "stub", "value": args.steps / dt, "n_gpus": world, "steps": args.steps, "ms_per_step": dt / args.steps * 1e3}))
    if world > 1:
        dist.destroy_process_group()


def usable_cpus():
    """logical CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands each
    lease a share of the host, e.g. 16 of 256)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return max(1, n)


def make_files(n_files, bytes_per_channel, dsd_rate, distinct, rank, threads, channels=2, fmt="P", endian="L", block=4096):
    """Synthetic planar-4096 LSB-first stereo files: half 1 kHz-family sines at 0.352 FS, half pink
    noise at ~0.098 RMS.  `distinct` different files are generated and tiled."""
    from helpers import pack_layout, synth
    distinct = min(distinct, n_files)

    msb = endian == "M"

    def one(i):
        seed = 1000 * rank + i
        if i % 2 == 0:
            ch = [synth("sine", bytes_per_channel, seed=seed + 500 * c, freq=1000.0 + 7 * i, phase=0.1 * i + 0.5 * c, dsd_rate=dsd_rate, msb_first=msb)
                  for c in range(channels)]
        else:
            ch = [synth("pink", bytes_per_channel, seed=seed + 500 * c, amp=0.098, dsd_rate=dsd_rate, msb_first=msb) for c in range(channels)]
        return pack_layout(ch, fmt, block)

    with ThreadPoolExecutor(max_workers=threads) as ex:
        base = list(ex.map(one, range(distinct)))
    return [base[i % distinct] for i in range(n_files)]


def cpu_baseline(kw, files, threads, budget_s):
    """The CPU restatement on the host cores: one file per thread (the reference's Rayon policy,
    src/main.rs:148-155,280-300: threads = logical cores / 2), fed in chunks like its block loop, through the oracle
    library's streaming organisation (orc_translate_stream: buffers kept between calls, integer byte tables, no bit
    reversal -- the same bytes as the oracle, tests/test_oracle_kat.py)."""
    from oracle import oracle as O
    O.use_native()          # -O3 -march=native, built on this box
    chunk_blocks = 64
    C_ = kw["channels"]

    def work(buf):
        o = O.Oracle(**kw)
        step = 4096 * C_ * chunk_blocks          # (a multiple of every layout's block group)
        out = None
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s:        # the file again and again until the budget is spent
            for a in range(0, len(buf), step):
                out, fr = o.translate_stream(buf[a:a + step], out)
                n += fr * C_
                if time.perf_counter() - t0 > budget_s:
                    break
        return n

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        counts = list(ex.map(work, [files[i % len(files)] for i in range(threads)]))
    dt = time.perf_counter() - t0
    return sum(counts) / dt / 1e6, sum(counts), dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--files", type=int, default=64, help="files per GPU")
    ap.add_argument("--seconds", type=float, default=60.0, help="audio seconds per file")
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic files generated per rank (0 = all of them distinct; fewer are tiled to --files)")
    ap.add_argument("--workload", default="dsd64_to_88k2_s24_stereo", choices=sorted(WORKLOADS))
    ap.add_argument("--kernel", default="auto", choices=["auto", "lut", "mfma"])
    ap.add_argument("--shard", default="files", choices=["files", "channels"],
                    help="files: every rank converts its own files (weak scaling, the default).  channels: every rank holds the SAME "
                         "multichannel files and converts its channel range (BASELINE config 5: one stream split by channel; strong scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="--shard files only.  weak (default): --files files on EVERY rank.  strong: --files-total files split over the ranks "
                         "(SURVEY.md 8d S3: the 512-file batch on 1/2/4/8 GPUs with the same total)")
    ap.add_argument("--files-total", type=int, default=512, help="--scaling strong: files of the whole job")
    ap.add_argument("--sustain", type=float, default=2.0, help="seconds of back-to-back steps timed once more after the repetitions (0 = skip)")
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed K-step loop (median reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-resident batch (pinned host in/out, upload/convert/download overlapped) that is reported as the extra pcie_inclusive object, never as value")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU-baseline work")
    ap.add_argument("--dither", default="", help="override the workload's dither (T, R, X, F, N)")
    ap.add_argument("--level", type=float, default=0.0, help="volume in dB (the reference's -l; its own test scripts use +-4)")
    ap.add_argument("--tap-bits", type=int, default=24, choices=(24, 32), help="tap grid (32: the optional 32-bit taps, two FIR passes and a combining pass; 44.1k-family workloads with dither T/R/F/X)")
    ap.add_argument("--as-rank", default="", help="--shard channels on ONE GPU: convert the channel range rank r of an N-way split would take, written r/N (e.g. 0/8)")
    ap.add_argument("--force-dist", action="store_true", help="one rank, but through the multi-GPU protocol: process group over RCCL, table blob broadcast and imported, barriers and the MAX-reduce around the timed region")
    ap.add_argument("--debug", type=lambda v: int(v, 0), default=0, help="d2d_params.debug_flags (include/dsd2dxd_amd.h: D2D_DBG_*), e.g. 2 = the older kernels for levels other than 0 dB; the line then names the flags and cites no counter traffic")
    ap.add_argument("--pcie-slice", type=int, default=0, help="bytes per channel per slice of the host-resident batch (0 = the library's default)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if os.environ.get("D2D_BENCH_STUB"):
        return stub_rank(args)

    import torch
    import torch.distributed as dist
    import dsd2dxd_amd as d

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one process per GPU over RCCL ("nccl" IS RCCL on ROCm).  D2D_BENCH_BACKEND=gloo is a rehearsal
    # mode for boxes with fewer GPUs than ranks: same protocol, collectives on host tensors, ranks
    # folded onto the GPUs that exist.
    backend = os.environ.get("D2D_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node")
    local_dev = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    cdev = dev if backend == "nccl" else torch.device("cpu")      # where collective payloads live
    # --force-dist: the collective path of a multi-GPU run with ONE rank (init, table-blob broadcast and import, barriers, MAX-reduce of
    # the time), so that it has executed on a real GPU before an 8-GPU node sees it; no scaling number follows from it
    dist_on = world > 1 or args.force_dist
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if dist_on:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    dsd_rate, out_rate, bits, dither, channels, bytes_per_sample = WORKLOADS[args.workload]
    if args.dither:
        dither = args.dither.upper()
    M = DSD64 * dsd_rate / out_rate
    blocks = max(1, int(round(args.seconds * DSD64 * dsd_rate / 8 / 4096)))
    bpc = blocks * 4096                                   # bytes per channel per file
    fmt, endian, block = LAYOUTS.get(args.workload, ("P", "L", 4096))
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=channels, fmt=fmt, endianness=endian,
              block_size=block, filter="E", bit_depth=bits, dither=dither, seed=206)
    if args.tap_bits == 32:
        kw["tap_bits"] = 32
    if args.level != 0.0:
        kw["level_db"] = args.level
    kernel = {"auto": d.KERNEL_AUTO, "lut": d.KERNEL_LUT, "mfma": d.KERNEL_MFMA}[args.kernel]
    ncpu = usable_cpus()
    gen_threads = max(1, min(32, ncpu // max(1, min(world, 8))))

    if args.scaling == "strong":
        if args.shard != "files":
            raise SystemExit("--scaling strong goes with --shard files (a channel shard is strong scaling already)")
        from dsd2dxd_amd.shard import shard_range
        lo, hi = shard_range(args.files_total, world, rank)
        args.files = hi - lo
        if args.files < 1:
            raise SystemExit(f"--scaling strong: rank {rank} of {world} gets no file of {args.files_total}")
    if args.distinct <= 0:
        args.distinct = args.files
    ch_first, ch_count = 0, channels
    if args.shard == "channels":
        from dsd2dxd_amd.shard import shard_channels
        if world > channels:
            raise SystemExit(f"--shard channels: {world} ranks but only {channels} channels")
        ch_first, ch_count = shard_channels(channels, world, rank)
        if args.as_rank:
            # one GPU converts exactly the share rank r of an N-way channel split would convert (BASELINE config 5: one channel, or a pair,
            # of an 8-channel stream per GPU): that rank's cost on record from a one-GPU lease
            r_, n_ = (int(v) for v in args.as_rank.split("/"))
            if world != 1 or not (0 <= r_ < n_ <= channels):
                raise SystemExit("--as-rank r/N: one process, 0 <= r < N <= channels")
            ch_first, ch_count = shard_channels(channels, n_, r_)
        kw.update(channel_first=ch_first, channel_count=ch_count)
    # file shards differ per rank; a channel shard reads the same files on every rank
    files = make_files(args.files, bpc, dsd_rate, args.distinct, rank if args.shard == "files" else 0, gen_threads, channels, fmt, endian, block)
    eng = d.Engine(n_files=args.files, kernel=kernel, device=local_dev, debug=args.debug, **kw)
    stream = torch.cuda.current_stream().cuda_stream

    # the shared filter tables: rank 0's copy is broadcast over RCCL and adopted by the others
    if dist_on:
        nb_t = torch.tensor([eng.tables_bytes()], dtype=torch.int64, device=cdev)      # rank 0's size (another variant may differ)
        dist.broadcast(nb_t, src=0)
        nb = int(nb_t.item())
        blob = torch.empty(nb, dtype=torch.uint8, device=dev)
        if rank == 0:
            eng.tables_export_device(blob.data_ptr(), nb, stream)
        torch.cuda.synchronize()
        wire = blob.to(cdev)
        dist.broadcast(wire, src=0)
        blob.copy_(wire)
        torch.cuda.synchronize()
        if rank != 0 or args.force_dist:
            # a rank whose conversion differs from rank 0's (an uneven channel shard: other channel count, hence another kernel and
            # table variant) is refused by the blob's header and keeps the tables it built itself
            try:
                eng.tables_import_device(blob.data_ptr(), nb, stream)
            except d.D2DError as ex:
                sys.stderr.write(f"rank {rank}: keeping its own filter tables ({ex})\n")

    # device-resident inputs and outputs (distinct files uploaded once, tiled by pointer)
    uniq = {}
    d_in = []
    for b in files:
        if id(b) not in uniq:
            uniq[id(b)] = torch.from_numpy(b).to(dev)
        d_in.append(uniq[id(b)])
    frames = eng.next_frames(bpc)
    fb = eng.frame_bytes
    d_out = torch.empty((args.files, (frames * fb + 31) // 16 * 16), dtype=torch.uint8, device=dev)
    ios = (d.FileIO * args.files)()
    for f in range(args.files):
        ios[f].dsd = d_in[f].data_ptr()
        ios[f].bytes_per_channel = bpc
        ios[f].pcm = d_out[f].data_ptr()
        ios[f].pcm_capacity_bytes = frames * fb
    assert d_out.stride(0) % 16 == 0

    def step():
        eng.translate_batch_device(ios, stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.profile_read_all()
    eng.profile_enable(True)
    dts = []
    for _ in range(max(1, args.reps)):
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        dts.append(dt)
    fir_ms, step_ms, launches = eng.profile_read_all()
    eng.profile_enable(False)
    # does the rate hold for seconds (the chip sets its clock by the load)?  One more timed stretch of back-to-back steps
    sustained = None
    if args.sustain > 0:
        n_sus = max(args.steps, int(args.sustain / max(min(dts) / args.steps, 1e-6)) + 1)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            step()
        torch.cuda.synchronize()
        dts_ = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dts_], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts_ = float(t.item())
        sustained = {"steps": n_sus, "seconds": round(dts_, 3), "ms_per_step": round(dts_ / n_sus * 1e3, 4)}
    dts_sorted = sorted(dts)
    dt = dts_sorted[len(dts_sorted) // 2]                 # the median repetition is the reported one

    samples_per_step_rank = frames * ch_count * args.files
    if world > 1 and (args.shard == "channels" or args.scaling == "strong"):
        tot = torch.tensor([samples_per_step_rank], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_samples = int(tot.item()) * args.steps
    else:
        total_samples = samples_per_step_rank * args.steps * world
    value = total_samples / dt / 1e6
    fir_s = fir_ms / 1e3 / max(1, launches)               # the FIR kernel alone, per launch (mean over all repetitions)
    step_s = step_ms / 1e3 / max(1, launches)             # every kernel of a step
    # the FIR kernel is "the dominant kernel" only when it IS the step; cascades, the noise-shaping pass and the
    # de-interleave pre-pass are priced with the whole step's device time
    scope = "kernel" if fir_s >= 0.95 * step_s else "step"
    roof_s = fir_s if scope == "kernel" else step_s
    alg_bytes = samples_per_step_rank * bytes_per_sample      # per launch (one launch = one step of one rank)
    achieved = alg_bytes / roof_s / 1e9 if roof_s > 0 else 0.0

    out = {
        "metric": "output PCM Msamples/s, DSD64->88.2k stereo" if args.workload.startswith("dsd64_to_88k2") else f"output PCM Msamples/s, {args.workload}",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
        "repetitions": {"n": len(dts), "ms_per_step_min": round(dts_sorted[0] / args.steps * 1e3, 4), "ms_per_step_max": round(dts_sorted[-1] / args.steps * 1e3, 4),
                        "ms_per_step_all": [round(x / args.steps * 1e3, 4) for x in dts]}, "scaling": "weak" if args.shard == "files" and args.scaling == "weak" else "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic (2nd-order 1-bit modulator: 1 kHz-family sines at 0.352 FS and pink noise at ~0.1 RMS; %d distinct files per rank%s)" % (min(args.distinct, args.files), "" if args.distinct >= args.files else " tiled to %d" % args.files),
        "config": {"workload": f"{args.workload}: {args.files} files/GPU x {blocks * 4096 * 8 / (DSD64 * dsd_rate):.1f} s, {'planar 4096-B LSB-first' if fmt == 'P' else 'byte-interleaved MSB-first'} {channels} ch -> {bits}-bit {out_rate} Hz, dither {dither}{', level %g dB' % args.level if args.level else ''}, filter E ({eng.info()['ntaps']} taps{', 32-bit grid' if args.tap_bits == 32 else ''}, M={M:g})",
                   "files_per_gpu": args.files, "seconds_per_file": round(blocks * 4096 * 8 / (DSD64 * dsd_rate), 3),
                   "parallelism": (f"files sharded over {world} GPU(s), no data-path collective" if args.shard == "files" else
                                   f"channels of every file split over {world} GPU(s) ({ch_count} of {channels} on rank 0), no data-path collective"), "kernel": eng.kernel_name()},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                     "scope": scope, "kernel": eng.kernel_name() if scope == "kernel" else "every kernel of a step (FIR: %s)" % eng.kernel_name(),
                     "kernel_ms": round(roof_s * 1e3, 4), "fir_kernel_ms": round(fir_s * 1e3, 4), "step_kernels_ms": round(step_s * 1e3, 4),
                     "algorithmic_bytes_per_launch": int(alg_bytes), "bytes_per_output_sample": bytes_per_sample},
    }
    if sustained:
        sustained["value"] = round(total_samples / args.steps * sustained["steps"] / sustained["seconds"] / 1e6, 3)
        out["sustained"] = sustained
    # a development library (D2D_AMD_LIB, tools/ab_*.sh) is named in the line, and no counter traffic is cited for it
    if args.debug:
        out["config"]["debug_flags"] = args.debug
    if args.as_rank:
        # the rank's algorithmic bytes count ITS channels only; what it must fetch is the whole byte-interleaved stream
        out["config"]["as_rank"] = {"rank_of": args.as_rank, "channels": [ch_first, ch_count], "input_bytes_it_must_fetch_per_launch": int(args.files * bpc * channels) if fmt == "I" else int(args.files * bpc * ch_count),
                                    "note": "one rank's share of a channel split, timed alone on one GPU; value and roofline count this rank's channels only"}
    if dist_on:
        out["config"]["collectives"] = {"backend": backend, "world": world, "table_blob_bytes": int(nb)}
    alt_lib = os.environ.get("D2D_AMD_LIB")
    if alt_lib:
        with open(alt_lib, "rb") as fl:
            out["config"]["library"] = {"path": alt_lib, "sha16": hashlib.sha256(fl.read()).hexdigest()[:16]}
    # PMC traffic is collected in separate rocprofv3 --pmc passes (tools/prof.sh -> profiles/); bench.py only cites an
    # entry taken on exactly these kernel sources and this workload size, otherwise traffic stays null
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fjs:
            pmc = json.load(fjs)
        # (a line taken with modifiers has its own key, as tools/collect.py files it: never the plain workload's entry)
        pmc_key = args.workload + ("+taps32" if args.tap_bits == 32 else "") + ("+level%g" % args.level if args.level else "")
        ent = pmc.get(eng.kernel_name() if scope == "kernel" else "step", {}).get(pmc_key)      # "step": every kernel of the step, summed
        if (ent and not alt_lib and not args.debug and not args.as_rank and ent.get("kernel_src_sha16") == kernel_source_hash() and ent.get("files_per_gpu") == args.files
                and abs(ent.get("seconds_per_file", 0) - out["config"]["seconds_per_file"]) < 1e-3):
            out["roofline"]["traffic"] = ent["hbm_bytes_per_launch"]
    except Exception:
        pass

    if not args.no_pcie:
        # end-to-end from pinned host memory: d2d_translate_batch_host (three streams, double-buffered
        # staging), every rank at once on its own GPU and link.  A separate engine so the timed engine's
        # state is untouched.
        e2 = d.Engine(n_files=args.files, kernel=kernel, device=local_dev, debug=args.debug, **kw)
        e3 = d.Engine(n_files=args.files, kernel=kernel, device=local_dev, debug=args.debug | d.DBG_HOST_STAGED, **kw)    # the same through the staged pipeline
        h_in = {}
        for b in files:
            if id(b) not in h_in:
                h_in[id(b)] = torch.from_numpy(b).pin_memory()
        h_out = torch.empty((args.files, (frames * fb + 31) // 16 * 16), dtype=torch.uint8).pin_memory()
        hios = (d.FileIO * args.files)()
        for f in range(args.files):
            hios[f].dsd = h_in[id(files[f])].data_ptr()
            hios[f].bytes_per_channel = bpc
            hios[f].pcm = h_out[f].data_ptr()
            hios[f].pcm_capacity_bytes = frames * fb
        def host_pass(e2):
            e2.reset()
            e2.translate_batch_host(hios, args.pcie_slice)       # warm-up (allocates what the route needs)
            reps = 2
            if dist_on:
                dist.barrier()
            t1 = time.perf_counter()
            for _ in range(reps):
                e2.translate_batch_host(hios, args.pcie_slice)
            dth = (time.perf_counter() - t1) / reps
            if dist_on:
                t = torch.tensor([dth], dtype=torch.float64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dth = float(t.item())
            return dth
        hb = args.files * (bpc * channels + frames * fb)
        up_b, down_b = args.files * bpc * channels, args.files * frames * fb
        forced = bool(args.debug & d.DBG_HOST_STAGED)
        dth = host_pass(e2)                                      # pinned buffers: the kernels address them directly (one pass, no staging)
        dts_ = host_pass(e3)                                     # the sliced upload / convert / download pipeline (what pageable buffers get)
        out["pcie_inclusive"] = {"value": round(total_samples / args.steps / dth / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(dth * 1e3, 3),
                                 "host_bytes_per_step_per_gpu": int(hb), "link_GBps_both_ways_per_gpu": round(hb / dth / 1e9, 2),
                                 "link_GBps_up_per_gpu": round(up_b / dth / 1e9, 2), "link_GBps_down_per_gpu": round(down_b / dth / 1e9, 2),
                                 "route": "staged pipeline (D2D_DBG_HOST_STAGED)" if forced else "kernels read and write the pinned host buffers in place",
                                 "staged_pipeline": {"ms_per_step": round(dts_ * 1e3, 3), "link_GBps_both_ways_per_gpu": round(hb / dts_ / 1e9, 2),
                                                     "link_GBps_up_per_gpu": round(up_b / dts_ / 1e9, 2), "link_GBps_down_per_gpu": round(down_b / dts_ / 1e9, 2)},
                                 "note": "pinned host buffers -> pinned host buffers through d2d_translate_batch_host, all %d GPU(s) at once; never the reported value" % world}
        del e2, e3

    if rank == 0 and not args.no_cpu_baseline:
        threads = max(1, ncpu // 2)
        v, n, secs = cpu_baseline(kw, files, threads, args.cpu_budget)
        out["cpu_baseline"] = {"value": round(v, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
                               "per_thread": round(v / threads, 3),
                               "reference_screenshot_per_worker": REF_SCREENSHOT_MSAMPLES_PER_WORKER,
                               "host_cpus_usable": ncpu,
                               "sample": f"{threads} streams of the same workload (one file per thread, threads = usable logical cores/2 as src/main.rs:148-155), {n} output samples in {secs:.1f} s; oracle/d2d_oracle.c orc_translate_stream (integer byte tables, f64 epilogue), gcc -O3 -march=native"}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
