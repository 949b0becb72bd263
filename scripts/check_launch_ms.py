#!/usr/bin/env python3
"""The bench line's launch durations against the rocprofv3 kernel trace of the SAME command.

    python scripts/check_launch_ms.py profiles/r05_bench_scircuit_detail.json profiles/r05_bench_scircuit_kernel_stats.csv [tol=0.05]

(the DETAIL file of the plain run against the TRACE of the same command under rocprofv3: a line printed under the
profiler carries ~5 us of profiler overhead in every HIP-event duration)

bench.py times the numeric launches of the COMPLETE call in an untimed pre-pass (HIP events on the launch's own stream;
the light launches and the numeric-first one carry kernel-exact begin / end stamps); `rocprofv3 --kernel-trace --stats`
of the same command averages every dispatch of a kernel.  The kernels only the structure-reuse mode runs carry names of
their own (sym_light_fused_kernel, num_light_kernel<T, false, ..>, nf_dense_kernel<T, N, true>), so the trace's average
of an eager kernel name IS the complete call's launch.  Exits 1 if the launch `roofline.kernel` names differs from the
trace by more than tol."""
import csv
import json
import sys

KERNEL_OF = {  # bench launch name -> kernel name in the trace
    "light": "num_light_kernel<{T}, true, false>",
    "numeric_first": "nf_dense_kernel<{T}, 256, false>",
    "nfcopy": "nf_copy_kernel<{T}>",
}


def main():
    d = json.load(open(sys.argv[1]))
    head = d["headline"]
    T = "float" if head["dtype"] == "f32" else "double"
    stats = {r["kernel"]: r for r in csv.DictReader(open(sys.argv[2]))}
    tol = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
    bad = 0
    headline = head["roofline"]["kernel"].split(":")[-1]  # the launch `roofline.achieved / avg_launch_ms` are quoted on
    for launch in head["roofline"]["launches"]:
        k = KERNEL_OF.get(launch["name"], "").format(T=T)
        if not k:
            continue
        if k not in stats:
            print(f"{launch['name']:14s} {k}: not in the trace")
            bad += launch["name"] == headline
            continue
        avg_ms = float(stats[k]["avg_us"]) * 1e-3
        rel = abs(launch["ms"] - avg_ms) / avg_ms
        ok = rel <= tol
        is_head = launch["name"] == headline
        print(f"{launch['name']:14s} bench {launch['ms']*1e3:8.2f} us   trace avg {avg_ms*1e3:8.2f} us ({stats[k]['calls']} calls)   "
              f"{rel*100:5.1f} %  {'ok' if ok else 'DIFFERS'}{'   <- roofline.kernel' if is_head else ''}")
        bad += 0 if (ok or not is_head) else 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
