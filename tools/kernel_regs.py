#!/usr/bin/env python
"""Per-kernel register / spill / LDS figures from the AMDGPU metadata of a saved assembly
(hipcc -save-temps): in each kernel's YAML block the keys are sorted, so .vgpr_count follows .name.

    python tools/kernel_regs.py /tmp/gram-hip-amdgcn-amd-amdhsa-gfx950.s [filter]
"""
import re
import sys

keys = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
        ".group_segment_fixed_size", ".private_segment_fixed_size")
cur, rows = None, []
for line in open(sys.argv[1]):
    m = re.match(r"\s+-?\s*\.name:\s+(\S+)", line)
    if m and m.group(1).startswith("_Z"):
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.match(r"\s+(\.\w+):\s+(\d+)", line)
    if m and cur is not None and m.group(1) in keys:
        cur[m.group(1)] = int(m.group(2))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    if flt in r["name"]:
        print(r["name"][:70], " ".join(f"{k[1:]}={r.get(k)}" for k in keys if k in r))
