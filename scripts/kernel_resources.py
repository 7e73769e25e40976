#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPR/AGPR/SGPR/spill/LDS/occupancy) from hipcc remarks.
usage: scripts/kernel_resources.py pigeon_amd/csrc/attention.hip [extra hipcc flags]"""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: .*?(Function Name|Name): (\S+)", line)
    if m:
        cur = {"name": m.group(2)}; rows.append(cur); continue
    m = re.search(r"remark: \s*([A-Za-z \[\]/]+): (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()[:90]
    print(f"{name:92s} vgpr={r.get('VGPRs','?'):>4} agpr={r.get('AGPRs','?'):>3} sgpr={r.get('TotalSGPRs','?'):>4} "
          f"spill={r.get('VGPRs Spill','?')}/{r.get('SGPRs Spill','?')} scratch={r.get('ScratchSize [bytes/lane]','?')} "
          f"lds={r.get('LDS Size [bytes/block]','?')} occ={r.get('Occupancy [waves/SIMD]','?')}")
