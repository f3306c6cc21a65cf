import os, sys, subprocess, time, numpy as np
sys.path.insert(0, "tests")
from helpers import write_dsf, random_bytes
n = 4096 * 25840   # 300 s of DSD64 per channel (105.8 MB)
chans = [random_bytes(n, 1), random_bytes(n, 2)]
os.makedirs("/tmp/cli", exist_ok=True)
write_dsf("/tmp/cli/big.dsf", chans)
for out in ("w", "f"):
    t = time.time()
    p = subprocess.run(["dsd2dxd_amd/dsd2dxd_amd_cli", "-o", out, "-r", "88200", "-b", "24", "-p", "/tmp/cli", "/tmp/cli/big.dsf"], stderr=subprocess.PIPE)
    print(out, "wall %.2f s" % (time.time() - t), p.stderr.decode().strip())
