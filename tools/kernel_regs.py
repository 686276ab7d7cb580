#!/usr/bin/env python
"""Per-kernel register / spill / LDS figures from the AMDGPU metadata of a saved assembly
(hipcc -save-temps).  Each kernel is one item of the `amdhsa.kernels:` list ("  - .agpr_count: ..."
opens it); its keys are sorted, so .name sits in the MIDDLE of the item -- reading "the lines
around the name" attributes half of the figures to the neighbouring kernel.

    python tools/kernel_regs.py /tmp/gram-hip-amdgcn-amd-amdhsa-gfx950.s [filter]
"""
import re
import sys

keys = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
        ".group_segment_fixed_size", ".private_segment_fixed_size")
cur, rows, inside = None, [], False
for line in open(sys.argv[1]):
    if line.startswith("amdhsa.kernels:"):
        inside = True
        continue
    if not inside:
        continue
    if re.match(r"^\S", line):           # next top-level key ends the list
        inside = False
        continue
    if re.match(r"^  - \.", line):       # a new kernel item
        cur = {}
        rows.append(cur)
    m = re.match(r"^  [- ] (\.\w+):\s+(\S+)", line)
    if m and cur is not None:
        k, v = m.group(1), m.group(2)
        if k == ".name":
            cur["name"] = v
        elif k in keys:
            cur[k] = int(v)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    if flt in r.get("name", ""):
        print(r["name"][:70], " ".join(f"{k[1:]}={r.get(k)}" for k in keys if k in r))
