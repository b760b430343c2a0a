#!/usr/bin/env python3
"""Tabulate the arms of a tools/lutm_ab.py log ('== <arm>' header lines followed by the JSON line of that arm): us per layer per shape_M."""
import json
import sys

rows, cur = {}, None
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("=="):
        cur = line[3:]
    elif line.startswith("{"):
        for k, v in json.loads(line).items():
            if isinstance(v, dict) and "us" in v:
                rows.setdefault(k, {}).setdefault(cur, []).append(v["us"])
arms = []
for v in rows.values():
    for a in v:
        if a not in arms:
            arms.append(a)
print("shape_M".ljust(18) + "".join(a.rjust(22) for a in arms))
for k, v in rows.items():
    print(k.ljust(18) + "".join("/".join(f"{x:.2f}" for x in v.get(a, [])).rjust(22) for a in arms))
