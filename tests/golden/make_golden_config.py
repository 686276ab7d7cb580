#!/usr/bin/env python
"""
The reference's flag set as data: every ``add_argument`` call of hydra/config.py:5-589 (flag,
dest, type, nargs, action and the VALUE of its default) written to config_flags.json.  The file
is parsed, not executed, except for the default expressions, which are literals or numpy
arithmetic.  Run in the build container only.

Usage:  python tests/golden/make_golden_config.py
"""
import ast
import json
import os

import numpy as np

REF = "/root/reference/hydra/config.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_flags.json")


def main():
    rows = []
    for node in ast.walk(ast.parse(open(REF).read())):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) \
                and node.func.attr == "add_argument":
            kw = {k.arg: k.value for k in node.keywords}
            default = eval(ast.unparse(kw["default"]), {"np": np}) if "default" in kw else False
            if isinstance(default, tuple):
                default = list(default)
            rows.append({
                "flag": node.args[0].value,
                "dest": kw["dest"].value,
                "type": ast.unparse(kw["type"]) if "type" in kw else None,
                "nargs": ast.literal_eval(kw["nargs"]) if "nargs" in kw else None,
                "action": kw["action"].value,
                "default": default,
            })
    with open(OUT, "w") as fh:
        json.dump(rows, fh, indent=1)
    print("wrote", OUT, len(rows), "flags")


if __name__ == "__main__":
    main()
