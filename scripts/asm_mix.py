#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel (substring match on the mangled name) in a .hip file."""
import re, subprocess, sys
from collections import Counter
src, pat = sys.argv[1], sys.argv[2]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", "/tmp/_mix.s", src] + sys.argv[3:], check=True, stderr=subprocess.DEVNULL)
s = open("/tmp/_mix.s").read()
m = re.search(r"^(\S*" + pat + r"\S*):", s, re.M)
start = m.start(); end = s.index(".Lfunc_end", start)
cur = "entry"; order = [cur]; blocks = {}
for l in s[start:end].splitlines():
    mm = re.match(r"^(\.LBB\d+_\d+):", l)
    if mm: cur = mm.group(1); order.append(cur); continue
    x = l.strip()
    if not x or x[0] in ";." : continue
    blocks.setdefault(cur, []).append(x)
def cls(op):
    if "mfma" in op: return "mfma"
    if op.startswith("v_exp"): return "exp"
    if op.startswith("ds_read"): return "ds_rd"
    if op.startswith("ds_write"): return "ds_wr"
    if op.startswith("v_pk_"): return "v_pk"
    if op.startswith("v_cvt"): return "cvt"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return op
print(m.group(1))
for k in order:
    v = blocks.get(k, [])
    if len(v) < 8: continue
    c = Counter(cls(x.split()[0]) for x in v)
    print(f"{k:10s} n={len(v):4d} ", dict(sorted(c.items())))
