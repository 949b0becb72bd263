#!/usr/bin/env python3
"""Merge separate rocprofv3 --pmc passes (csv files written by rocpd_pmc.py: one row per kernel, mean per dispatch)
into ONE per-kernel table and into the two json files bench.py reads:
  profiles/traffic.json   workload:launch -> HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024
                          (FETCH_SIZE tallies 128-B fabric requests at 64 B on gfx950, WRITE_SIZE is exact:
                          profiles/r01_fetch_write_calibration.txt, r02_fetch_calibration_gather.txt)
  profiles/counters.json  workload:launch -> {counter: mean per launch} for the ceilings next to the HBM roofline:
                          SQ_INSTS_LDS_ATOMIC (LDS atomic wave-instructions), TCP_TCC_READ/WRITE_REQ (L1 -> L2
                          requests), SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, LDS bank conflicts ...
usage: make_counters.py <workload> <out.csv> <traffic.json> <counters.json> pass1.csv [pass2.csv ...]"""
import csv
import json
import os
import re
import sys

KEYS = [
    # canonical keys = the kernels of the COMPLETE call (what bench.py's line quotes); the kernels only the structure-reuse
    # mode runs carry names of their own and *_reuse keys
    (r"walk_kernel", "num_walk"),
    (r"nf_dense_kernel<\w+, \d+, true>", "num_numeric_first_reuse"), (r"nf_dense_kernel", "num_numeric_first"), (r"nf_copy_kernel", "num_nfcopy"),
    (r"num_light_kernel<\w+, true", "num_light"), (r"num_light_kernel<\w+, false, true>", "num_light_reuse_verify"),
    (r"num_light_kernel", "num_light_reuse"),
    (r"analysis_kernel<\d+, \d+u, true>", "analysis_verify_reuse"), (r"verify_inputs_kernel", "verify_inputs_reuse"),
    (r"snapshot_inputs_kernel", "snapshot_inputs_reuse"),
    (r"sym_light_fused_kernel", "sym_light_fused_reuse"), (r"sym_light_kernel", "sym_light"),
    (r"num_hash_kernel<Block<512>", "num_block8k"), (r"num_hash_kernel<Block<256>", "num_block2k"),
    (r"num_dense_kernel<\w+, 16384u", "num_dense16k"),
    (r"num_spill_scatter_kernel", "num_global_scatter"), (r"num_spill_count_kernel", "num_global_count"),
    (r"num_spill_reduce_kernel<\w+, 2048u", "num_global_reduce"), (r"num_spill_reduce_kernel<\w+, 8192u", "num_global_reduce_big"),
    (r"num_spill_copy_kernel", "num_global_copy"),
    (r"sym_hash_kernel", "sym_hash"), (r"sym_bitmap_kernel", "sym_bitmap"), (r"sym_global_hash_kernel", "sym_global_hash"),
    (r"validate_b_kernel", "validate_b"),
    (r"analysis_kernel", "analysis"), (r"sym_scatter_kernel", "sym_scatter"), (r"scan_kernel", "scan"),
    (r"num_apply_pred_kernel", "num_apply_pred_reuse"),
    (r"num_count_kernel", "num_count"), (r"num_apply_kernel", "num_apply"), (r"done_kernel", "done"),
]


def main():
    workload, out_csv, tjson, cjson = sys.argv[1:5]
    table = {}
    for path in sys.argv[5:]:
        if not os.path.exists(path):
            continue
        for row in csv.DictReader(open(path)):
            k = row.pop("kernel")
            row.pop("dispatches", None)
            table.setdefault(k, {}).update({c: float(v) for c, v in row.items()})
    cols = sorted({c for k in table for c in table[k]})
    with open(out_csv, "w") as f:
        f.write("kernel," + ",".join(cols) + ",hbm_bytes_per_launch=(2*FETCH+WRITE)*1024\n")
        for k in sorted(table):
            hbm = int((2 * table[k].get("FETCH_SIZE", 0.0) + table[k].get("WRITE_SIZE", 0.0)) * 1024)
            f.write(f"\"{k}\"," + ",".join(f"{table[k].get(c, 0):.0f}" for c in cols) + f",{hbm}\n")
    traffic = json.load(open(tjson)) if os.path.exists(tjson) else {}
    counters = json.load(open(cjson)) if os.path.exists(cjson) else {}
    traffic = {k: v for k, v in traffic.items() if not k.startswith(workload + ":")}
    counters = {k: v for k, v in counters.items() if not k.startswith(workload + ":")}
    src = os.path.basename(out_csv).split("_")[0]
    traffic["_source"] = f"profiles/{src}_pmc_*_counters.csv"
    counters["_source"] = f"profiles/{src}_pmc_*_counters.csv"
    for k in table:
        for pat, key in KEYS:
            if re.search(pat, k):
                if "FETCH_SIZE" in table[k] or "WRITE_SIZE" in table[k]:
                    traffic[f"{workload}:{key}"] = int((2 * table[k].get("FETCH_SIZE", 0.0) + table[k].get("WRITE_SIZE", 0.0)) * 1024)
                counters[f"{workload}:{key}"] = {c: round(v) for c, v in table[k].items()}
                break
    json.dump(traffic, open(tjson, "w"), indent=1, sort_keys=True)
    json.dump(counters, open(cjson, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
