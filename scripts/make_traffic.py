#!/usr/bin/env python3
"""Merge the FETCH_SIZE and WRITE_SIZE passes of rocprofv3 (csv from rocpd_pmc.py) into the per-kernel
HBM traffic table that bench.py's roofline.traffic reads.
usage: make_traffic.py fetch.csv write.csv out.csv traffic.json workload
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE under-reports coalesced 4/8/16-byte
per-lane streams by exactly 2x on gfx950, WRITE_SIZE is exact (profiles/r01_fetch_write_calibration.txt)."""
import csv
import json
import os
import re
import sys

KEYS = [
    (r"nf_dense_kernel", "num_numeric_first"), (r"nf_copy_kernel", "num_nfcopy"),
    (r"num_light_kernel", "num_light"), (r"num_tiny_kernel", "num_tiny"), (r"sym_light_kernel", "sym_light"),
    (r"num_hash_kernel<Block<512>", "num_block8k"), (r"num_hash_kernel<Block<256>", "num_block2k"),
    (r"num_hash_kernel<SubWave<64>, \w+, 512u", "num_wave512"),
    (r"num_hash_kernel<SubWave<64>, \w+, 128u", "num_wave128"), (r"num_hash_kernel<SubWave<16>", "num_g16"),
    (r"num_direct_kernel", "num_direct"), (r"num_dense_kernel<\w+, 4096u", "num_dense4k"),
    (r"num_dense_kernel<\w+, 16384u", "num_dense16k"), (r"num_global_kernel", "num_global"),
    (r"analysis_kernel", "analysis"), (r"sym_scatter_kernel", "sym_scatter"),
    (r"num_count_kernel", "num_count"), (r"num_apply_kernel", "num_apply"),
]


def load(path, col):
    out = {}
    for row in csv.DictReader(open(path)):
        out[row["kernel"]] = float(row[col])
    return out


def main():
    fpath, wpath, out_csv, tjson, workload = sys.argv[1:6]
    f, w = load(fpath, "FETCH_SIZE"), load(wpath, "WRITE_SIZE")
    lines = ["kernel,FETCH_SIZE_KB_raw_avg,WRITE_SIZE_KB_avg,hbm_bytes_per_launch=(2*FETCH+WRITE)*1024"]
    traffic = json.load(open(tjson)) if os.path.exists(tjson) else {}
    traffic = {k: v for k, v in traffic.items() if not k.startswith(workload + ":")}
    traffic["_source"] = os.path.dirname(out_csv) + "/" + os.path.basename(out_csv).split("_")[0] + "_pmc_*_fetch_write.csv"
    for k in sorted(set(f) | set(w)):
        b = int((2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024)
        lines.append(f"\"{k}\",{f.get(k, 0.0):.1f},{w.get(k, 0.0):.1f},{b}")
        for pat, key in KEYS:
            if re.search(pat, k):
                traffic[f"{workload}:{key}"] = b
                break
    open(out_csv, "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(tjson, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
