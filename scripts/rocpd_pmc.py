#!/usr/bin/env python3
"""Per-kernel mean of every collected PMC counter from a rocprofv3 rocpd .db.
usage: rocpd_pmc.py results.db [out.csv]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"speck::", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<.*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    rows = cur.execute(f"select {namecol}, counter_name, value, dispatch_id from counters_collection").fetchall()
    agg = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for kn, cn, v, did in rows:
        agg[short(kn)][cn][did] += v
    counters = sorted({c for k in agg for c in agg[k]})
    lines = ["kernel,dispatches," + ",".join(counters)]
    for k in sorted(agg):
        nd = max(len(agg[k][c]) for c in agg[k])
        vals = [sum(agg[k][c].values()) / max(len(agg[k][c]), 1) if c in agg[k] else 0 for c in counters]
        lines.append(f"\"{k}\",{nd}," + ",".join(f"{v:.0f}" for v in vals))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
