#!/usr/bin/env python3
"""Timeline of the LAST `n` kernel dispatches in a rocprofv3 rocpd db: start offset, duration.
usage: rocpd_timeline.py results.db [n=40]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"speck::", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<.*>)?\(", name)
    return (m.group(1) + (m.group(2) or ""))[:70] if m else name[:70]


db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else "kernel_name"
extra = [c for c in ("queue_id", "stream_id") if c in cols]
rows = cur.execute(f"select {namecol}, start, end {''.join(', ' + e for e in extra)} from kernels order by start").fetchall()
rows = rows[-n:]
t0 = rows[0][1]
for r in rows:
    print(f"{(r[1]-t0)/1e3:9.1f} us  +{(r[2]-r[1])/1e3:8.1f} us  end {(r[2]-t0)/1e3:9.1f}  {short(r[0])}  {r[3:] if extra else ''}")
