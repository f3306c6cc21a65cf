import torch, time
up = torch.empty(2_700_000_000, dtype=torch.uint8).pin_memory(); dn = torch.empty(2_030_000_000, dtype=torch.uint8).pin_memory()
dup = torch.empty_like(up, device="cuda"); ddn = torch.empty(dn.numel(), dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(do_up, do_dn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if do_up:
        with torch.cuda.stream(s1): dup.copy_(up, non_blocking=True)
    if do_dn:
        with torch.cuda.stream(s2): dn.copy_(ddn, non_blocking=True)
    torch.cuda.synchronize(); return time.perf_counter() - t0
for _ in range(2): run(True, True)
tu = run(True, False); td = run(False, True); tb = run(True, True)
print("up alone %.1f GB/s, down alone %.1f GB/s, both at once: %.1f ms -> up %.1f + down %.1f = %.1f GB/s" % (2.7/tu, 2.03/td, tb*1e3, 2.7/tb, 2.03/tb, 4.73/tb))
# chunked like the engine: 64 x 8 MiB pieces per direction per slice
