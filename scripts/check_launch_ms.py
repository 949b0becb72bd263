#!/usr/bin/env python3
"""The bench line's launch durations against the rocprofv3 kernel trace of the SAME command.

    python scripts/check_launch_ms.py profiles/r04_bench_scircuit.json profiles/r04_bench_scircuit_kernel_stats.csv [tol=0.05]

(the LINE of the plain run against the TRACE of the same command under rocprofv3: a line printed under the profiler
carries ~5 us of profiler overhead in every HIP-event duration)

bench.py times the launches of the replayed sequence in an untimed pre-pass (HIP events on the launch's own stream);
`rocprofv3 --kernel-trace --stats` of the same command averages every dispatch of a kernel.  The kernels of the replayed
sequence carry names of their own (sym_light_fused_kernel; num_light_kernel<T, false, false>: no register-class bodies,
num_light_kernel<T, false, true>: ... and verifying bodies;
analysis_kernel<.., true>: the verifier; num_apply_pred_kernel), so the trace's average IS the replay launch's.  Exits 1
if a launch of `roofline.launches` differs from the trace by more than tol."""
import csv
import json
import sys

KERNEL_OF = {  # bench launch name -> kernel name in the trace (fp64 legs)
    "fused_light": "sym_light_fused_kernel<double>",
    "light": "num_light_kernel<double, false, false>",
    "numeric_first": "nf_dense_kernel<double, 256, true>",
}


def main():
    # (the bench line: the last line of the file that is a JSON object -- the profiler logs behind it)
    line = json.loads([ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")][-1])
    stats = {r["kernel"]: r for r in csv.DictReader(open(sys.argv[2]))}
    tol = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
    bad = 0
    headline = line["roofline"]["kernel"].split(":")[-1]  # the launch `roofline.achieved / avg_launch_ms` are quoted on
    for launch in line["roofline"]["launches"]:
        k = KERNEL_OF.get(launch["name"])
        if not k:
            continue
        # (a sequence whose numeric launch verifies the row lengths itself -- no symbolic pass, nlpkkt stand-in -- runs the
        #  third form of that kernel; its first replay, and the eager pre-pass, run the others)
        if launch["name"] == "light" and "num_light_kernel<double, false, true>" in stats:
            k = "num_light_kernel<double, false, true>"
        if k not in stats and launch["name"] == "light":      # a sequence that is not fused: the eager kernel's name
            k = "num_light_kernel<double, true, false>"
        if k not in stats:
            print(f"{launch['name']:14s} {k}: not in the trace")
            bad += 1
            continue
        avg_ms = float(stats[k]["avg_us"]) * 1e-3
        rel = abs(launch["ms"] - avg_ms) / avg_ms
        ok = rel <= tol
        is_head = launch["name"] == headline
        print(f"{launch['name']:14s} bench {launch['ms']*1e3:8.2f} us   trace avg {avg_ms*1e3:8.2f} us ({stats[k]['calls']} calls)   "
              f"{rel*100:5.1f} %  {'ok' if ok else 'DIFFERS'}{'   <- roofline.kernel' if is_head else ''}")
        # (only the headline launch decides: a launch that shares the chip with side-stream launches of its phase -- the
        #  fused light launch of the webbase stand-in next to the heavy symbolic classes -- runs beside other work in the
        #  event-timed pre-pass than in the graph)
        bad += 0 if (ok or not is_head) else 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
