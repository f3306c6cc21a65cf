"""One rank of the two-process engine test (tests/test_gpu_sharding.py): builds its share of a deterministic
batch, adopts rank 0's filter-table blob over gloo (the bench's RCCL broadcast, rehearsed on one GPU) and converts
it with a real Engine on cuda:0.  usage: shard_worker.py <mode files|channels> <rank> <world> <port> <out.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_FILES, NBYTES, CHN = 5, 4096 * 6 + 200, 6
KW_FILES = dict(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="T", seed=206)
# (D2D_SHARD_RATE: the uneven-shard test runs an 88.2 kHz stream, whose one- and two-channel engines hold different table variants;
# at 96 kHz one composed table serves every channel count)
KW_CHANNELS = dict(dsd_rate=1, output_rate=int(os.environ.get("D2D_SHARD_RATE", "96000")), channels=CHN, fmt="I", endianness="M", block_size=1, filter="E", bit_depth=24, dither="T", seed=9)


def file_bytes(f):
    from helpers import pack_layout, synth
    return pack_layout([synth("sine", NBYTES, seed=10 + f), synth("pink", NBYTES, seed=20 + f, amp=0.098)], "P", 4096)


def stream_bytes():
    from helpers import pack_layout, random_bytes
    return pack_layout([random_bytes(NBYTES, 40 + c) for c in range(CHN)], "I", 1)


def main():
    mode, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import dsd2dxd_amd as d
    from dsd2dxd_amd.shard import shard_channels, shard_range
    backend = os.environ.get("D2D_SHARD_BACKEND", "gloo")      # "nccl": RCCL, collective payloads on the device (one rank per GPU)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cdev = dev if backend == "nccl" else torch.device("cpu")
    if mode == "files":
        b, e = shard_range(N_FILES, world, rank)
        kw, nf = KW_FILES, e - b
        bufs = [file_bytes(f) for f in range(b, e)]
    else:
        first, count = shard_channels(CHN, world, rank)
        kw, nf = dict(KW_CHANNELS, channel_first=first, channel_count=count), 1
        bufs = [stream_bytes()]
    eng = d.Engine(n_files=max(nf, 1), kernel=d.KERNEL_MFMA, device=0, **kw)
    # rank 0's tables, broadcast and adopted: what bench.py does before its timed region
    # (a rank whose conversion differs from rank 0's -- an uneven channel shard: another channel count, hence another kernel and table
    # variant -- is refused by the blob's header and keeps the tables it built itself)
    nb_t = torch.tensor([eng.tables_bytes()], dtype=torch.int64, device=cdev)
    dist.broadcast(nb_t, src=0)
    nb = int(nb_t.item())
    blob = torch.zeros(nb, dtype=torch.uint8, device=dev)
    if rank == 0:
        eng.tables_export_device(blob.data_ptr(), nb)
    torch.cuda.synchronize()
    wire = blob.to(cdev)
    dist.broadcast(wire, src=0)
    adopted = rank == 0
    if rank != 0 or os.environ.get("D2D_SHARD_IMPORT_OWN"):
        blob.copy_(wire)
        torch.cuda.synchronize()
        try:
            eng.tables_import_device(blob.data_ptr(), nb)
            adopted = True
        except d.D2DError:
            adopted = False
    # the conversion: device-resident batch in two calls (state carried), like the bench's step
    res = {}
    if nf:
        bpc = NBYTES
        cut = 4096 * 4
        d_in = [torch.from_numpy(x).to(dev) for x in bufs]
        fb = eng.frame_bytes
        outs = [[] for _ in bufs]
        chin = kw["channels"]
        for a, z in ((0, cut), (cut, bpc)):
            L = z - a
            frames = eng.next_frames(L)
            d_out = torch.zeros((len(bufs), (frames * fb + 31) // 16 * 16), dtype=torch.uint8, device=dev)
            ios = (d.FileIO * len(bufs))()
            keep = []
            for i, x in enumerate(bufs):
                # planar 4096 blocks: bytes [a, z) per channel are whole block groups (the cut is a block multiple, the ragged
                # tail is one short group at the end); byte-interleaved: the same byte range times the channel count
                piece = torch.from_numpy(np.ascontiguousarray(x[a * chin:z * chin])).to(dev)
                keep.append(piece)
                ios[i].dsd = piece.data_ptr(); ios[i].bytes_per_channel = L
                ios[i].pcm = d_out[i].data_ptr(); ios[i].pcm_capacity_bytes = frames * fb
            eng.translate_batch_device(ios)
            torch.cuda.synchronize()
            for i in range(len(bufs)):
                outs[i].append(d_out[i, :ios[i].frames_out * fb].cpu().numpy().copy())
        for i in range(len(bufs)):
            res["pcm%d" % i] = np.concatenate(outs[i])
        res["peaks"] = np.array([[eng.peak(c, file=i) for c in range(eng.out_channels)] for i in range(len(bufs))])
    res["kernel"] = np.array(eng.kernel_name())
    res["adopted"] = np.array(adopted)
    np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
