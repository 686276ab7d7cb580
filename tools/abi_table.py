#!/usr/bin/env python
"""Markdown table of every entry point of include/hydra_hip.h with the reference lines its
comment cites (INTEGRATION.md section 2.1 is this script's output)."""
import os
import re

HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "hydra_hip.h")
text = open(HDR).read()
# (comment, declaration) pairs: a declaration without its own comment inherits the previous one
pat = re.compile(r"(/\*.*?\*/)|((?:int|size_t|const char\*)\s+(hydra_\w+)\s*\([^;]*\);)", re.S)
rows, comment = [], ""
for m in pat.finditer(text):
    if m.group(1):
        comment = m.group(1)
        continue
    name = m.group(3)
    cites = re.findall(r"[\w/]+\.py:\s*\d+(?:-\d+)?(?:,\s*\d+(?:-\d+)?)*", comment)
    cites = [re.sub(r"\s+", "", c) for c in cites]
    body = re.sub(r"^/\*+|\*+/$", "", comment)
    body = re.sub(r"\n\s*\*", " ", body)
    body = re.sub(r"\s+", " ", body).strip(" -")
    if name in ("hydra_last_error", "hydra_abi_version"):
        cites = []
    rows.append((name, ", ".join(dict.fromkeys(cites)) or "none (error text, version, profiling or test hook)"))
print("| Entry point | Reference lines it replaces |")
print("|---|---|")
for name, cites in rows:
    print(f"| `{name}` | {cites} |")
