#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing: tools/isa_blocks.py file.s <kernel-name-substring>"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (l.startswith("_Z") and key in l and ": " in l and "; @" in l))
blocks, cur = [], ["entry", {}]
def cls(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"): return "mfma"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_"): return "salu"
    return "other"
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end"): break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = [m.group(1), {}]; continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    op = t.split()[0]
    c = cls(op); cur[1][c] = cur[1].get(c, 0) + 1
    if c == "br": cur[1].setdefault("to", []).append(t.split()[-1])
blocks.append(cur)
for name, d in blocks:
    tot = sum(v for k, v in d.items() if k != "to")
    if tot < int(sys.argv[3]) if len(sys.argv) > 3 else 0: continue
    print("%-12s" % name, " ".join("%s=%s" % (k, d[k]) for k in ("mfma", "valu", "lane", "lds", "vmem", "salu", "smem", "wait") if k in d), "->", ",".join(d.get("to", [])))
