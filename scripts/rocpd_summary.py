#!/usr/bin/env python3
"""Per-kernel summary (calls, avg/min/max us, share) from a rocprofv3 rocpd .db file.
usage: rocpd_summary.py results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"speck::", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<.*>)?\(", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {namecol}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [])
        a.append((e - s) / 1e3)
    total = sum(sum(v) for v in agg.values())
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,pct"]
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"\"{k}\",{len(v)},{sum(v):.1f},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{100*sum(v)/total:.1f}")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
