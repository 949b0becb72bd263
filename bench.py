#!/usr/bin/env python3
"""bench.py -- SpGEMM GFLOP/s (2 * intermediate products / s) for A*A, the metric of BASELINE.json.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload scircuit|...] [--scaling weak|strong]

One "step" = one COMPLETE MultiplyspECK call with A and B resident in HBM and the output matrix reused across
steps (the reference's benchmark loop, source/Executor.cpp:43-72): analysis -> binning -> symbolic -> scan ->
allocation check -> numeric (+ in-kernel sort), every stage inside the timed region, as the reference runs them in
every iteration (source/GPU/Multiply.cu:488-575, 835-1043).  `value` / `ms_per_step` are THAT call (library option
reuse = 0).  The structure-reuse mode of a repeated identical call (the replayed sequence) is
reported beside it as `value_reuse` / `ms_reuse` at N = 1 and is never the metric.

stdout carries ONE JSON line (<= 4 KB, no prose).  Everything else -- every launch with bytes / duration / ceilings,
rows per class, the verification details of every leg -- goes to --detail (default bench_detail.json next to this
file; scripts/check_launch_ms.py and the profiles/ regeneration read that).

N > 1 (torch.distributed.run, or started here when no launcher did): rows of A are sharded by the analysis pass'
product counts, B is replicated, every step ends with ONE exchange: the gatherv of the C shards to rank 0
(speck_gather_* of the C ABI over RCCL).  `value` includes the exchange, `multiply_only` is the same steps with C
left row-sharded.  --scaling weak (default): N x the rows; --scaling strong: the same matrix at every N.
Unless --no-config5 the line carries `config5`: the nlpkkt160 stand-in, strong scaling, at this N.
At N = 1 the line carries `configs`: the other single-GPU configurations of BASELINE.json, one short object each.

The output of the LAST timed step is downloaded and checked (the reference compares after every iteration,
source/Executor.cpp:51-55, 67-71): against the CPU oracle in full (indices bit-exact, values within
1e-12 * sum|a*b|), the nlpkkt leg through size-independent properties on the device plus the oracle on sampled row
blocks (oracle/verify.py).  `verified` is false -- and the exit code 3 -- if any check of any leg fails.

Inputs: $SPECK_MTX_DIR/<name>.mtx (scircuit, webbase-1M, mac_econ_fwd500, cant, nlpkkt160) is used when present
("data": "suitesparse"); SuiteSparse files do not exist offline, so the default is the stand-in of SURVEY.md 8d
fitted to the original's n / nnz / P / nnz(C) ("data": "synthetic").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: the HIP runtime torch bundles is the one the process shares)
import torch.distributed as dist  # noqa: E402

import speck_amd as sa  # noqa: E402
from speck_amd.api import NUM_CLASS_NAMES  # noqa: E402
from speck_amd.sharding import GatherPlan, NativeComm, NativeGatherPlan, TRANSPORT_HOSTMEM, TRANSPORT_RCCL  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
XGMI_LINK_GBS = 153.0  # one xGMI link, per direction (7 links per GPU, point to point)
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md: aggregate L2 bandwidth of the 8 XCDs
# LDS atomic issue ceiling, wave-instructions per second for the whole chip: ds_add_f64 takes 20.6 cycles per
# wave-instruction and CU (scripts/ubench/lds_atomics.hip); 256 CUs at 2.4 GHz
LDS_ATOMIC_PEAK_GWPS = 256 * 2.4 / 20.6
# ... and for the MIX a hash kernel issues (round 6, profiles/r06_lds_atomics_ubench.txt): cycles per wave-instruction and CU
# at a given share of ds_add_f64 among returning ds_cmpst_b32 (random slots): pure compare-and-swap 11.2, 7:1 11.9,
# 3:1 13.0, 1:1 15.4, pure add 20.7 -- a launch is priced against the rate of ITS mix (piecewise linear in the share)
LDS_MIX_CYCLES = ((0.0, 11.2), (0.125, 11.9), (0.25, 13.0), (0.5, 15.4), (1.0, 20.7))


def lds_atomic_peak_gwps(add_share):
    a = min(max(add_share, 0.0), 1.0)
    for (x0, y0), (x1, y1) in zip(LDS_MIX_CYCLES, LDS_MIX_CYCLES[1:]):
        if a <= x1:
            return 256 * 2.4 / (y0 + (y1 - y0) * (a - x0) / (x1 - x0))
    return LDS_ATOMIC_PEAK_GWPS
# VALU issue ceiling: a wave64 instruction occupies its 16-lane SIMD for 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz
VALU_PEAK_GWPS = 256 * 4 * 2.4 / 4
SUITESPARSE_FILES = {"scircuit": "scircuit", "webbase": "webbase-1M", "mac_econ": "mac_econ_fwd500",
                     "cant": "cant", "nlpkkt": "nlpkkt160"}
# numeric launches of the COMPLETE call as bench.py names them -> key in profiles/{traffic,counters}.json
# (scripts/make_counters.py; the kernels of the reuse mode carry names of their own and other keys)
COUNTER_KEYS = {"light": "num_light", "numeric_first": "num_numeric_first", "nfcopy": "num_nfcopy", "block8k": "num_block8k",
                "dense16k": "num_dense16k", "global": "num_global_reduce"}
# the classes of the merged 256-thread numeric launch (num_light_kernel)
LIGHT = ("dense4k", "block2k", "wave512", "wave256", "r64", "r32", "wave128", "g16", "g8", "direct")


class _DevArray:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr=typestr, data=(int(ptr), False),
                                             version=2, strides=None)


def shard_tensors(dC):
    n, rows = dC.nnz, dC.rows
    ro = torch.as_tensor(_DevArray(dC._c.row_offsets, rows + 1, "<i4"), device="cuda")
    col = torch.as_tensor(_DevArray(dC._c.col_ids, max(n, 1), "<i4"), device="cuda")[:n]
    val = torch.as_tensor(_DevArray(dC._c.data, max(n, 1), "<f4" if dC.dtype == np.float32 else "<f8"), device="cuda")[:n]
    return ro, col, val


def find_suitesparse(workload):
    """$SPECK_MTX_DIR/<name>.mtx or $SPECK_MTX_DIR/<name>/<name>.mtx (the layout of the collection's tarballs)."""
    d, name = os.environ.get("SPECK_MTX_DIR"), SUITESPARSE_FILES.get(workload)
    if not d or not name:
        return None
    for p in (os.path.join(d, name + ".mtx"), os.path.join(d, name, name + ".mtx")):
        if os.path.exists(p):
            return p
    return None


def load_workload(workload, scale, seed, mtx=None):
    path = mtx or find_suitesparse(workload)
    if path:
        return sa.load_matrix(path, write_cache=False), "suitesparse", os.path.basename(path)
    A = sa.gen_matrix(workload, scale, seed, signed=True)
    return A, "synthetic", f"{workload}-like A*A (SURVEY 8d stand-in)"


class Env:
    """Process-wide state: ranks, device, options."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # plumbing check of the N > 1 path on a box with fewer GPUs than ranks: every rank on GPU 0,
        # gloo instead of RCCL (the exchange is staged through host memory) -- never a measurement
        self.shared_gpu = os.environ.get("SPECK_BENCH_SHARED_GPU") == "1"
        if self.shared_gpu:
            self.local_rank = 0
        if self.world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(self.local_rank)
            if self.shared_gpu:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
        else:
            torch.cuda.set_device(0)
        # (main() re-launches itself as N ranks when --gpus N arrives without a launcher: a line never claims fewer
        #  GPUs than it was asked for)
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}")
        self.dev = torch.device("cuda", self.local_rank)
        self.opts = [o.split("=") for o in args.opt]
        # the exchange itself: the library's own RCCL gatherv (C ABI); ranks that share one GPU cannot form an
        # RCCL communicator and take the library's host-staged transport -- same plan / displacement code
        self.exchange = args.exchange
        self.comm = None
        self.exchange_note = None
        if self.world > 1 and self.exchange == "native":
            # The library's RCCL transport has only ever met ranks that share one GPU in this build environment (no
            # multi-GPU box): it is brought up and SELF-TESTED here -- a tiny gatherv whose result rank 0 checks -- and
            # if any rank fails, all ranks fall back to the torch.distributed exchange together (and the line says so).
            ok, why = 1, ""
            try:
                if os.environ.get("SPECK_BENCH_FAIL_NATIVE") == "1":   # (test hook: the bring-up fails on every rank --
                    raise RuntimeError("forced by SPECK_BENCH_FAIL_NATIVE")  #  a failure of ONE rank inside a collective
                                                                             #  cannot be recovered from: the others wait)
                self.comm = NativeComm(self.local_rank, TRANSPORT_HOSTMEM if self.shared_gpu else TRANSPORT_RCCL)
                rows = 3 + self.rank
                ro = np.arange(rows + 1, dtype=np.uint32) * 2
                col = np.tile(np.array([1, 5], dtype=np.uint32), rows)
                val = np.full(2 * rows, float(self.rank + 1))
                shard = sa.dCSR.from_host(sa.HostCSR(rows, 8, ro, col, val))
                full = self.comm.gatherv(shard, 8, root=0)
                if self.rank == 0:
                    h = full.to_host()
                    want_rows = sum(3 + r for r in range(self.world))
                    want_val = np.concatenate([np.full(2 * (3 + r), float(r + 1)) for r in range(self.world)])
                    good = (h.rows == want_rows and h.nnz == 2 * want_rows and
                            (h.row_offsets == np.arange(want_rows + 1, dtype=np.uint32) * 2).all() and
                            (h.col_ids == np.tile(np.array([1, 5], dtype=np.uint32), want_rows)).all() and
                            (h.data == want_val).all())
                    if not good:
                        ok, why = 0, "self-test gatherv returned a wrong matrix"
            except Exception as e:  # noqa: BLE001
                ok, why = 0, repr(e)
            flag = torch.tensor([ok], dtype=torch.int64, device=self.dev if not self.shared_gpu else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if self.comm is not None:
                    try:
                        self.comm.close()
                    except Exception:  # noqa: BLE001
                        pass
                self.comm = None
                self.exchange = "torch"
                self.exchange_note = "native exchange failed its self-test on some rank (" + (why or "another rank") + \
                                     "): fell back to torch.distributed"

    def new_config(self):
        cfg = sa.spECKConfig.initialize(self.local_rank)
        for name, value in self.opts:
            cfg.set_option(name, int(value))
        return cfg

    def barrier(self):
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, *ints):
        if self.world == 1:
            return ints
        t = torch.tensor(list(ints), dtype=torch.int64, device=self.dev)
        dist.all_reduce(t)
        return tuple(int(x) for x in t.tolist())


class Job:
    """A (resident in HBM) x A on this rank's row shard, with the optional pipelined gatherv."""

    def __init__(self, env, A, gather):
        self.env, self.A = env, A
        dev = env.dev
        self.t_ro = torch.from_numpy(A.row_offsets.view(np.int32)).to(dev)
        self.t_col = torch.from_numpy(A.col_ids.view(np.int32)).to(dev)
        self.t_val = torch.from_numpy(A.data).to(dev)
        self.dA = sa.dCSR.from_device(A.rows, A.cols, A.nnz, self.t_ro.data_ptr(), self.t_col.data_ptr(),
                                      self.t_val.data_ptr(), dtype=A.data.dtype, keep=(self.t_ro, self.t_col, self.t_val),
                                      host_row_offsets=A.row_offsets)
        self.cfg = env.new_config()
        if env.world > 1:
            bounds = sa.partition_rows(self.dA, self.dA, self.cfg, env.world)
            self.mine = self.dA.row_view(bounds[env.rank], bounds[env.rank + 1])
        else:
            self.mine = self.dA
        self.gather = gather and env.world > 1
        # N > 1: two output matrices (each with its own config) alternate, so that a shard can be sent while the
        # next one is computed
        self.slots = [(self.cfg, sa.dCSR(A.data.dtype))]
        if self.gather:
            self.slots.append((env.new_config(), sa.dCSR(A.data.dtype)))
        self.plan = None
        self.n_step = 0
        self.last = None  # (config, output matrix) of the last step
        self.bounds = (0, A.rows) if env.world == 1 else (bounds[env.rank], bounds[env.rank + 1])

    def set_reuse(self, on):
        """on: a repeated identical call may run the structure-reuse sequence (library option reuse);
        off: every call is the complete pipeline."""
        for scfg, _ in self.slots:
            scfg.set_option("reuse", int(on))

    def step(self, exchange=True):
        slot = self.n_step % len(self.slots)
        self.n_step += 1
        scfg, sC = self.slots[slot]
        self.last = (scfg, sC)
        if self.plan is not None:
            self.plan.wait(slot)  # the exchange that still reads this slot's output matrix
        sa.MultiplyspECK(self.mine, self.dA, sC, scfg)  # returns with C complete in HBM
        if self.gather and exchange:
            if self.env.comm is not None:
                if self.plan is None:
                    self.plan = NativeGatherPlan(self.env.comm, self.mine.rows, self.dA.cols, sC.nnz,
                                                 self.A.data.dtype.itemsize, root=0,
                                                 slots=len(self.slots))
                self.plan.start(slot, sC)
            else:
                ro, col, val = shard_tensors(sC)
                if self.plan is None:
                    self.plan = GatherPlan(sC.rows, sC.nnz, col.dtype, val.dtype, self.env.dev, root=0,
                                           slots=len(self.slots), stage_on_host=self.env.shared_gpu)
                self.plan.start(slot, ro[1:] - ro[:-1], col, val)

    def drain(self):
        if self.plan is not None:
            self.plan.wait_all()

    def barrier(self):
        self.drain()
        self.env.barrier()

    def timed(self, steps, exchange=True):
        """K steps between barriers; MAX over ranks, seconds."""
        self.barrier()
        if len(self.slots) == 1 and self.plan is None:
            # one output matrix, nothing to exchange: the reference's loop (source/Executor.cpp:59-72) -- the same
            # MultiplyspECK call K times, its arguments bound once (speck_amd.BoundMultiply)
            scfg, sC = self.slots[0]
            call = sa.BoundMultiply(self.mine, self.dA, sC, scfg)
            t0 = time.perf_counter()
            for _ in range(steps):
                call()
            self.n_step += steps
            self.last = (scfg, sC)
        else:
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(exchange)
        self.barrier()
        return self.env.max_over_ranks(time.perf_counter() - t0)

    def close(self):
        self.drain()
        if isinstance(self.plan, NativeGatherPlan):
            self.plan.close()
        for scfg, _ in self.slots:
            scfg.cleanup()


def profile_prepass(job, prof_steps=9):
    """Untimed COMPLETE calls with HIP events, each recorded on the stream the launch runs on (the light launches and
    the numeric-first one carry kernel-exact begin / end stamps).  One call collects the algorithmic bytes per class;
    prof_steps calls with events around every launch -- the MEDIAN per launch (round 6: the first profiled calls start
    on a chip the event records have drained and measured up to 6 % above the kernel-trace average of the same command);
    prof_steps calls with one event pair per phase only."""
    cfg = job.cfg
    job.step()                           # (allocates C; the next calls are sized from this one like the timed ones)
    cfg.profile_kernels(1)
    cfg.set_option("collect_bytes", 1)   # per-class algorithmic bytes (per row): one call is enough
    job.step()
    torch.cuda.synchronize()
    st = cfg.last_stats()
    cfg.set_option("collect_bytes", 0)
    kernel_s = {k: [] for k in list(NUM_CLASS_NAMES) + ["light", "numeric_first"]}
    stage_s = {"analysis_binning": [], "scan": [], "sym_light": []}
    for _ in range(prof_steps):
        job.step()
        s = cfg.last_stats()
        for k in NUM_CLASS_NAMES:
            kernel_s[k].append(s["num_bin_ms"][k])
        kernel_s["light"].append(s["num_light_ms"] + s["num_tiny_ms"])
        # numeric-first rows: their NUMERIC kernel runs inside the symbolic phase -- a numeric launch
        kernel_s["numeric_first"].append(s["sym_bin_ms"]["numeric_first"] if s["sym_bin_rows"]["numeric_first"] else 0.0)
        stage_s["analysis_binning"].append(s["analysis_ms"])
        stage_s["scan"].append(s["scan_ms"])
        stage_s["sym_light"].append(s["sym_light_ms"] + s["sym_tiny_ms"])
    kernel_ms = {k: float(np.median(v)) for k, v in kernel_s.items()}
    stage_ms = {k: float(np.median(v)) for k, v in stage_s.items()}
    # the phases, WITHOUT an event between their class launches (mode 2: one event pair per phase)
    cfg.profile_kernels(2)
    sym_l, num_l = [], []
    for _ in range(prof_steps):
        job.step()
        s = cfg.last_stats()
        # symbolic = analysis + binning + symbolic launches + scan (the reference's countProducts + loadBalanceCounting
        # + globalMapsCounting + spGEMMCounting); the numeric-first kernel is numeric work inside it
        sym_l.append(s["analysis_ms"] + s["sym_phase_ms"] + s["scan_ms"] - kernel_ms["numeric_first"])
        num_l.append(s["num_phase_ms"] + kernel_ms["numeric_first"])
    sym_ms, num_ms = float(np.median(sym_l)), float(np.median(num_l))
    cfg.profile_kernels(0)
    kernel_bytes = dict(st["num_bin_bytes"])
    # the algorithmic bytes of the numeric-first rows belong to the launch that computes them; the copy of the
    # finished rows into C is extra traffic outside the model (listed by time only)
    kernel_bytes["numeric_first"] = kernel_bytes.pop("nfcopy")
    kernel_bytes["light"] = sum(kernel_bytes.pop(k) for k in LIGHT)
    st["num_bin_bytes"] = kernel_bytes
    return st, kernel_ms, stage_ms, sym_ms, num_ms


def ceilings_for(counters, workload, launch_name, ms, products=0):
    """Secondary ceilings of one launch from the committed rocprofv3 --pmc passes of the same command
    (profiles/counters.json; NOT measured in this run -- only the duration is): HBM-side bytes, L1 -> L2 requests,
    LDS atomic wave-instructions, VALU wave-instructions -- each as a fraction of its peak over `ms`.  The LDS-atomic
    peak is the one of the launch's own MIX: one ds_add_f64 per product (`products` / 64 wave-instructions), the rest
    returning compare-and-swaps (lds_atomic_peak_gwps).  Nothing is dropped: a fraction above 1 is printed as it is."""
    c = counters.get(f"{workload}:{COUNTER_KEYS.get(launch_name, 'num_' + launch_name)}")
    if not c or ms <= 0:
        return None
    sec = ms * 1e-3
    out = {}
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
        out["hbm_measured_frac"] = hbm / sec / 1e9 / HBM_PEAK_GBS
    if "TCP_TCC_READ_REQ_sum" in c:
        l2 = (c.get("TCP_TCC_READ_REQ_sum", 0) + c.get("TCP_TCC_WRITE_REQ_sum", 0)) * 64
        out["l2_frac"] = l2 / sec / 1e9 / L2_PEAK_GBS
    if c.get("SQ_INSTS_LDS_ATOMIC"):
        share = min(1.0, products / 64.0 / c["SQ_INSTS_LDS_ATOMIC"]) if products else 1.0
        out["lds_atomic_frac"] = c["SQ_INSTS_LDS_ATOMIC"] / sec / 1e9 / lds_atomic_peak_gwps(share)
        out["lds_add_share"] = share
    if "SQ_INSTS_VALU" in c:
        out["valu_frac"] = c["SQ_INSTS_VALU"] / sec / 1e9 / VALU_PEAK_GWPS
    out = {k: round(v, 4) for k, v in out.items()}
    if c.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_ratio"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 3)
    return out or None


def b_num_bytes(rows, nnz_a, products, nnz_c, vsize=8):
    """the algorithmic bytes of the numeric phase of one multiply."""
    return 4 * (rows + 1) + (4 + vsize) * nnz_a + 8 * nnz_a + (4 + vsize) * products + 4 * (rows + 1) + (4 + vsize) * nnz_c


def _load_json(name):
    p = os.path.join(ROOT, "profiles", name)
    return json.load(open(p)) if os.path.exists(p) else {}


def roofline_blocks(workload, st, kernel_ms, num_ms, b_num):
    """(compact object for the line, full object for the detail file).  Every numeric launch of the complete call with
    its algorithmic bytes (SURVEY 8d's per-row model summed over the rows of its classes, computed on the device),
    its duration and the fraction of the HBM peak; the line's kernel is the launch with the LONGEST duration (it
    bounds the phase).  `bound` = the measured ceiling with the highest fraction of its peak."""
    launches = []
    for name, b in st["num_bin_bytes"].items():
        ms = kernel_ms.get(name, 0.0)
        if b <= 0 or ms <= 0:
            continue
        gbs = b / (ms * 1e-3) / 1e9
        launches.append({"name": name, "bytes": int(b), "ms": round(ms, 5), "GBps": round(gbs, 1),
                         "frac": round(gbs / HBM_PEAK_GBS, 4)})
    launches.sort(key=lambda x: -x["ms"])
    if not launches:
        return None, None
    traffic_tab, counters = _load_json("traffic.json"), _load_json("counters.json")
    all_bytes = sum(x["bytes"] for x in launches) or 1
    for x in launches:
        # (products of a launch ~ its share of the numeric bytes: 12 of the ~13-26 bytes per product are the product's)
        x["ceilings"] = ceilings_for(counters, workload, x["name"], x["ms"], st["sum_products"] * x["bytes"] / all_bytes)
        x["traffic"] = traffic_tab.get(f"{workload}:{COUNTER_KEYS.get(x['name'], 'num_' + x['name'])}")
    dom = launches[0]
    ceil = dom["ceilings"] or {}
    named = {"hbm": ceil.get("hbm_measured_frac"), "l2": ceil.get("l2_frac"), "lds_atomic": ceil.get("lds_atomic_frac"),
             "valu": ceil.get("valu_frac")}
    named = {k: v for k, v in named.items() if v is not None}
    # the ceiling with the highest measured fraction; when none reaches a quarter of its peak the launch is bound by
    # LATENCY x occupancy (chains of dependent trips), not by any pipe -- said so instead of naming the largest small number
    bound = (max(named, key=named.get) if max(named.values()) >= 0.25 else "latency") if named else "hbm"
    phase_frac = round(b_num / max(num_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS, 4)
    compact = {
        "bound": bound, "kernel": f"numeric:{dom['name']}", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": dom["frac"], "traffic": dom["traffic"], "bytes": dom["bytes"], "avg_launch_ms": dom["ms"],
        "hbm_measured_frac": ceil.get("hbm_measured_frac"), "l2_frac": ceil.get("l2_frac"),
        "lds_atomic_frac": ceil.get("lds_atomic_frac"), "valu_frac": ceil.get("valu_frac"),
        "numeric_phase_frac": phase_frac,
    }
    full = dict(compact, launches=launches, numeric_phase_bytes=int(b_num), numeric_phase_ms=round(num_ms, 5),
                counters_source=counters.get("_source"), traffic_source=traffic_tab.get("_source"),
                peaks={"hbm_GBps": HBM_PEAK_GBS, "l2_GBps": L2_PEAK_GBS, "lds_atomic_Gwaveinst_per_s": round(LDS_ATOMIC_PEAK_GWPS, 2),
                       "valu_Gwaveinst_per_s": VALU_PEAK_GWPS})
    return compact, full


def verify_last_output(env, job, A, mode, cache):
    """Check the output matrix of the last step of `job` (this rank's row shard).  mode "oracle": the whole
    shard against the CPU oracle; "properties": device-side properties + the oracle on sampled row blocks.
    `cache` keeps the oracle's product between the two checks of a leg (complete call, reuse mode)."""
    from oracle import verify as ov
    scfg, sC = job.last
    r0, r1 = job.bounds
    st = scfg.last_stats()
    info = {"mode": mode, "replayed": st["replayed"], "pred_stages": st["pred_stages"], "eager_speculated": st["eager_speculated"]}
    try:
        if mode == "oracle":
            got = sC.to_host()
            from oracle import pyoracle as po
            if "ref" not in cache:
                H = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data)
                cache["ref"] = po.spgemm_f64_of(H.row_slice(r0, r1) if (r0, r1) != (0, A.rows) else H, H)
            ok, d = ov.compare_with_reference(cache["ref"], got.row_offsets, got.col_ids, got.data,
                                              ov.TOL32 if A.data.dtype == np.float32 else ov.TOL64)
        else:
            ro, col, val = shard_tensors(sC)
            ok, d = ov.device_properties(torch, job.t_ro, job.t_col, job.t_val, r0, r1, ro, col, val, A.cols)
            if ok and (r0, r1) == (0, A.rows):
                ok2, d2 = ov.sampled_blocks(torch, A, ro, col, val, blocks=3)
                ok, d = ok and ok2, dict(d, **d2)
        info.update(d)
    except Exception as e:  # a failed check must not hide the measurement; it fails the run instead
        ok = False
        info["error"] = repr(e)
    info["ok"] = bool(ok)
    info["ok_all_ranks"] = env.sum_over_ranks(0 if ok else 1)[0] == 0
    return info


def measure(env, A, steps, warmup, gather, profile, verify=None, reuse=False):
    """One leg.  Returns a result dict on every rank: the COMPLETE call timed (`ms_per_step`), and with `reuse` the
    structure-reuse mode of the same repeated call beside it (`ms_reuse`)."""
    job = Job(env, A, gather)
    job.set_reuse(False)
    out = {}
    if profile:
        st, kernel_ms, stage_ms, sym_ms, num_ms = profile_prepass(job)
        out.update(st=st, kernel_ms=kernel_ms, stage_ms=stage_ms, sym_ms=sym_ms, num_ms=num_ms)
    else:
        job.step()
        out["st"] = job.cfg.last_stats()
    P_local, nnzc_local = out["st"]["sum_products"], out["st"]["nnz_c"]
    for _ in range(max(warmup, len(job.slots) + 1)):
        job.step()
    elapsed = job.timed(steps)
    out["speculated"] = job.cfg.last_stats()["eager_speculated"]
    out["through"] = job.cfg.last_stats().get("eager_through", 0)  # 1: the timed complete calls were enqueued as one batch
    cache = {}
    out["verify"] = None
    if verify:
        job.drain()
        out["verify"] = verify_last_output(env, job, A, verify, cache)
    out["P"], out["nnzC"] = env.sum_over_ranks(P_local, nnzc_local)
    # the floor the exchange puts under a step at N > 1: every peer -> root transfer rides ONE xGMI link, so a step
    # cannot be shorter than the largest peer shard (row counts + column ids + values) over that link's peak
    out["exchange_floor_ms"] = None
    if env.world > 1:
        mine_bytes = 4 * (job.bounds[1] - job.bounds[0]) + 12 * nnzc_local
        t = torch.zeros(env.world, dtype=torch.int64, device=env.dev if not env.shared_gpu else "cpu")
        t[env.rank] = mine_bytes
        dist.all_reduce(t)
        peers = [int(x) for i, x in enumerate(t.tolist()) if i != 0]
        out["exchange_floor_ms"] = round(max(peers) / XGMI_LINK_GBS / 1e9 * 1e3, 4) if peers else 0.0
    out["ms_per_step"] = elapsed * 1e3 / steps
    out["gflops"] = 2.0 * out["P"] / (elapsed / steps) / 1e9
    # N > 1, reported next to `value` (never instead of it): the same K steps without the exchange,
    # i.e. what the row-sharded multiply alone sustains while C stays distributed like A
    out["multiply_only"] = None
    if job.gather:
        e2 = job.timed(steps, exchange=False)
        out["multiply_only"] = {"value": round(2.0 * out["P"] / (e2 / steps) / 1e9, 3), "ms_per_step": round(e2 * 1e3 / steps, 4)}
    # the structure-reuse mode of the same repeated call: reported beside the metric, never as it
    out["ms_reuse"] = out["verify_reuse"] = None
    out["replays"] = 0
    if reuse:
        job.set_reuse(True)
        for _ in range(6):               # (capture, first replay, the sequence without a scan from the second replay on)
            job.step(exchange=False)
        e3 = job.timed(steps, exchange=False)
        out["ms_reuse"] = e3 * 1e3 / steps
        out["replays"] = job.cfg.last_stats()["graph_replays"]
        if verify:
            out["verify_reuse"] = verify_last_output(env, job, A, verify, cache)
    job.close()
    return out


def leg_ok(res):
    checks = [v["ok_all_ranks"] for v in (res["verify"], res["verify_reuse"]) if v]
    return all(checks) if checks else None


def config_objects(workload, wl_name, data_label, A, res, steps):
    """(short object for the line, full object for the detail file) of one single-GPU leg."""
    st = res["st"]
    vsize = 4 if A.data.dtype == np.float32 else 8
    key = workload if vsize == 8 else workload + "_f32"
    compact_roof, full_roof = roofline_blocks(key, st, res["kernel_ms"], res["num_ms"],
                                              b_num_bytes(A.rows, A.nnz, res["P"], res["nnzC"], vsize))
    ms, ms_reuse = res["ms_per_step"], res["ms_reuse"]
    short = {
        "name": workload, "dtype": "f32" if vsize == 4 else "f64", "ms_per_step": round(ms, 4),
        "ms_reuse": round(ms_reuse, 4) if ms_reuse else None, "value": round(res["gflops"], 2),
        "value_reuse": round(2.0 * res["P"] / (ms_reuse * 1e-3) / 1e9, 2) if ms_reuse else None,
        "roofline_frac": compact_roof["frac"] if compact_roof else None,
        "numeric_phase_frac": compact_roof["numeric_phase_frac"] if compact_roof else None,
        "bound": compact_roof["bound"] if compact_roof else None, "verified": leg_ok(res),
    }
    full = dict(short, workload=wl_name, data=data_label, rows=A.rows, nnzA=A.nnz, products=res["P"], nnzC=res["nnzC"],
                steps=steps, phases_ms={"symbolic": round(res["sym_ms"], 4), "numeric": round(res["num_ms"], 4)},
                stages_ms={k: round(v, 5) for k, v in res["stage_ms"].items()}, roofline=full_roof,
                kernels_ms={k: round(v, 5) for k, v in res["kernel_ms"].items() if v > 0},
                rows_per_class={k: v for k, v in st["num_bin_rows"].items() if v},
                sym_rows_per_class={k: v for k, v in st["sym_bin_rows"].items() if v},
                eager_speculated=res["speculated"], eager_through=res.get("through", 0), graph_replays=res["replays"],
                verify=res["verify"],
                verify_reuse=res["verify_reuse"])
    return short, full, compact_roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="scircuit")
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64", help="value type of the headline workload")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--mtx", default=None, help="real MatrixMarket file instead of the stand-in")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lib-baseline", action="store_true", help="skip the rocSPARSE SpGEMM timing (apps/runspECK --time-library)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the gatherv exchange")
    ap.add_argument("--exchange", choices=("native", "torch"), default="native",
                    help="N>1: speck_gather_* of the C ABI (RCCL inside the library) or torch.distributed")
    ap.add_argument("--no-config5", action="store_true", help="skip the nlpkkt160 strong-scaling leg")
    ap.add_argument("--no-configs", action="store_true", help="N=1: skip the other single-GPU configurations")
    ap.add_argument("--no-verify", action="store_true", help="do not check the output of the last timed step")
    ap.add_argument("--no-f32", action="store_true", help="N=1: skip the fp32 legs (mac_econ, cant)")
    ap.add_argument("--no-reuse", action="store_true", help="N=1: skip the structure-reuse mode beside the metric")
    ap.add_argument("--configs-steps", type=int, default=20)
    ap.add_argument("--config5-scale", type=float, default=1.0)
    ap.add_argument("--config5-steps", type=int, default=5)
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="file for everything the line omits")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (tuning)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    env = Env(args)
    n_gpus, rank = env.world, env.rank
    reuse = n_gpus == 1 and not args.no_reuse

    # ---- workload (same on every rank: deterministic generator / same file, B replicated)
    scale = args.scale * (n_gpus if args.scaling == "weak" else 1)
    A, data_label, wl_name = load_workload(args.workload, scale, args.seed, args.mtx)
    if args.dtype == "f32":
        A = sa.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data.astype(np.float32))
    if data_label == "suitesparse" and args.scaling == "weak" and n_gpus > 1:
        args.scaling = "strong"  # a file has one size
    assert A.rows == A.cols, "A*A needs a square matrix (use the transpose for rectangular inputs)"
    # oracle in full wherever it finishes in seconds; the big stencil through properties + sampled blocks
    def verify_mode(w, A_):
        if args.no_verify:
            return None
        return "properties" if (w == "nlpkkt" and A_.rows > 2_000_000) or A_.nnz > 40_000_000 else "oracle"

    res = measure(env, A, args.steps, args.warmup, gather=not args.no_gather, profile=True,
                  verify=verify_mode(args.workload, A), reuse=reuse)
    verdicts = []
    out, detail = None, {}
    if rank == 0:
        short, full, roof = config_objects(args.workload, wl_name, data_label, A, res, args.steps)
        detail["headline"] = full
        out = {
            "metric": "SpGEMM GFLOP/s (2*flops_intermediate/s), A*A",
            "value": short["value"], "unit": "GFLOP/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": short["ms_per_step"], "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": short["dtype"],
            "data": data_label if not env.shared_gpu else data_label + " (ranks share one GPU, gloo: plumbing check only)",
            "config": {"workload": wl_name, "rows": A.rows, "nnzA": A.nnz, "products": res["P"], "nnzC": res["nnzC"],
                       "parallelism": f"rows{n_gpus}" if n_gpus > 1 else "single",
                       "timed_call": "complete (analysis+binning+symbolic+scan+numeric+sort)"},
            "value_reuse": short["value_reuse"], "ms_reuse": short["ms_reuse"],
            "verified": None,
            "phases_ms": full["phases_ms"],
            "roofline": roof,
        }
        if n_gpus > 1:
            out["config"].update(gather=not args.no_gather, exchange_floor_ms=res["exchange_floor_ms"],
                                 exchange=(env.exchange + (":hostmem" if env.shared_gpu else ":rccl") if env.exchange == "native"
                                           else "torch") if not args.no_gather else None)
            if env.exchange_note:
                detail["exchange_note"] = out["config"]["exchange_note"] = env.exchange_note[:160]
            out["multiply_only"] = res["multiply_only"]
        verdicts.append(short["verified"])
        if n_gpus == 1 and not args.no_lib_baseline:
            lb = lib_baseline(args.workload, args.scale, args.seed, args.mtx)
            out["lib_baseline"] = {k: lb[k] for k in ("kind", "ms")} if lb else None
            if lb:
                out["lib_baseline"]["value"] = round(2.0 * res["P"] / (lb["ms"] * 1e-3) / 1e9, 2)
            detail["lib_baseline"] = lb
    del res

    # ---- N = 1: every other single-GPU configuration of BASELINE.json (configs[1..3]), one short object each
    if n_gpus == 1 and not args.no_configs:
        entries, full_entries = [], []
        for w in ("scircuit", "webbase", "mac_econ", "cant"):
            if w == args.workload and args.scale == 1.0 and not args.mtx and args.dtype == "f64":
                continue
            Aw, label_w, name_w = load_workload(w, 1.0, args.seed)
            rw = measure(env, Aw, args.configs_steps, 3, gather=False, profile=True, verify=verify_mode(w, Aw), reuse=reuse)
            s, f, _ = config_objects(w, name_w, label_w, Aw, rw, args.configs_steps)
            if not args.no_lib_baseline:
                lb = lib_baseline(w, 1.0, args.seed)
                s["lib_ms"] = lb["ms"] if lb else None      # rocSPARSE SpGEMM, same input and protocol (lib_baseline)
                f["lib_baseline"] = lb
            entries.append(s)
            full_entries.append(f)
            verdicts.append(s["verified"])
            del Aw, rw
        # the <float, ...> instantiation (source/GPU/Multiply.cu:1130) on the two mid-density inputs: 8 bytes per
        # product in the byte model; checked against the product in fp64 (4 eps32 * sum|a*b|)
        if not args.no_f32:
            for w in ("mac_econ", "cant"):
                Aw, label_w, name_w = load_workload(w, 1.0, args.seed)
                Aw = sa.HostCSR(Aw.rows, Aw.cols, Aw.row_offsets, Aw.col_ids, Aw.data.astype(np.float32))
                rw = measure(env, Aw, args.configs_steps, 3, gather=False, profile=True, verify=verify_mode(w, Aw), reuse=reuse)
                s, f, _ = config_objects(w, name_w + " (fp32 values)", label_w, Aw, rw, args.configs_steps)
                entries.append(s)
                full_entries.append(f)
                verdicts.append(s["verified"])
                del Aw, rw
        out["configs"] = entries
        detail["configs"] = full_entries

    # ---- BASELINE.json configs[4]: the nlpkkt160 stand-in, STRONG scaling, at this N
    if not args.no_config5 and not (args.workload == "nlpkkt" and args.scaling == "strong"):
        A5, label5, name5 = load_workload("nlpkkt", args.config5_scale, args.seed)
        r5 = measure(env, A5, args.config5_steps, 2, gather=not args.no_gather, profile=n_gpus == 1,
                     verify=verify_mode("nlpkkt", A5), reuse=reuse)
        roof5 = full5 = None
        if rank == 0 and n_gpus == 1:
            _, full5, roof5 = config_objects("nlpkkt", name5, label5, A5, r5, args.config5_steps)
        if rank == 0:
            out["config5"] = {
                "name": "nlpkkt", "scaling": "strong", "n_gpus": n_gpus, "rows": A5.rows, "products": r5["P"],
                "steps": args.config5_steps, "value": round(r5["gflops"], 2), "ms_per_step": round(r5["ms_per_step"], 4),
                "value_reuse": round(2.0 * r5["P"] / (r5["ms_reuse"] * 1e-3) / 1e9, 2) if r5["ms_reuse"] else None,
                "ms_reuse": round(r5["ms_reuse"], 4) if r5["ms_reuse"] else None,
                "multiply_only": r5["multiply_only"], "exchange_floor_ms": r5["exchange_floor_ms"], "verified": leg_ok(r5),
            }
            if roof5:  # (N = 1: the one leg that is HBM-scale carries its fractions too)
                out["config5"].update(roofline_frac=roof5["frac"], numeric_phase_frac=roof5["numeric_phase_frac"], bound=roof5["bound"],
                                      kernel=roof5["kernel"])
                detail["config5_full"] = full5
            if n_gpus == 1 and not args.no_lib_baseline and args.config5_scale <= 1.0:
                lb = lib_baseline("nlpkkt", args.config5_scale, args.seed, timeout=420)
                out["config5"]["lib_ms"] = lb["ms"] if lb else None
                detail["config5_lib_baseline"] = lb
            detail["config5"] = dict(out["config5"], workload=name5, data=label5, nnzA=A5.nnz, nnzC=r5["nnzC"],
                                     verify=r5["verify"], verify_reuse=r5["verify_reuse"])
            verdicts.append(out["config5"]["verified"])
        del A5, r5

    if rank == 0:
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(A, out["config"]["products"])
        checked = [v for v in verdicts if v is not None]
        out["verified"] = all(checked) if checked else None
        line = json.dumps(out, separators=(",", ":"))
        detail["line"] = out
        detail["line_bytes"] = len(line)
        try:
            with open(args.detail, "w") as f:
                json.dump(detail, f, indent=1)
        except OSError as e:
            print(f"bench.py: could not write {args.detail}: {e}", file=sys.stderr)
        print(line, flush=True)
    failed = rank == 0 and out["verified"] is False
    if env.comm is not None:
        env.comm.close()
    if n_gpus > 1:
        dist.destroy_process_group()
    if failed:
        sys.exit(3)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (what torch.distributed.run would do on one
    node: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, rendezvous on 127.0.0.1), pass rank 0's output
    through and return the worst exit code.  Refuses to run N ranks on fewer GPUs unless SPECK_BENCH_SHARED_GPU=1 (the
    plumbing check of the tests): a line that says "n_gpus": N was measured on N GPUs."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("SPECK_BENCH_SHARED_GPU") != "1":
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s); nothing measured "
              f"(SPECK_BENCH_SHARED_GPU=1 runs the ranks on GPU 0 as a plumbing check)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst = 0
    try:
        for pr in procs:
            rc = pr.wait()
            worst = worst or rc
    finally:
        for pr in procs:          # a rank that died must not leave the others waiting in a collective forever
            if pr.poll() is None:
                pr.terminate()
    return worst


def lib_baseline(workload, scale, seed, mtx=None, timeout=240):
    """The same-box LIBRARY baseline: rocSPARSE generic SpGEMM on the same input with the reference's protocol (10 + 10
    products, buffers reused), timed by apps/runspECK --time-library in a process of its own (this one carries torch's HIP
    runtime; the driver links /opt/rocm's rocSPARSE).  Role of the cuSPARSE product the reference's driver computes beside
    its own (source/Executor.cpp:29-40).  None if the driver is missing or fails."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "apps", "runspECK")
    if not os.path.exists(exe):
        return None
    spec = mtx or find_suitesparse(workload) or f"gen:{workload}:{scale}:{seed}"
    try:
        with tempfile.NamedTemporaryFile("w", suffix=".ini", delete=False) as f:
            f.write("IterationsWarmUp=10\nIterationsExecution=10\nTrackCompleteTimes=true\nCompareResult=false\n")
            ini = f.name
        r = subprocess.run([exe, spec, ini, "--time-library"], capture_output=True, text=True, timeout=timeout)
        os.unlink(ini)
        ms = nnz = None
        for ln in r.stdout.splitlines():
            if "rocSPARSE SpGEMM:" in ln:
                ms = float(ln.split(":")[1].split()[0])
            if "rocSPARSE -> NNZ:" in ln:
                nnz = int(ln.split(":")[1])
        if ms is None:
            return None
        return {"kind": "rocsparse_spgemm", "ms": round(ms, 4), "nnzC": nnz, "protocol": "10+10, nnz+compute stages, buffers reused, rows unsorted"}
    except Exception:
        return None


def cpu_baseline(A, P):
    """The oracle (row-parallel Gustavson, symbolic + numeric) timed on the same workload with the
    thread count that serves it best on this host (a short sweep: the per-thread dense
    accumulators make it slower again beyond a few dozen threads); ~10-30 s of CPU work."""
    from oracle import pyoracle as po
    H = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data)
    avail = min(po.max_threads(), len(os.sched_getaffinity(0)))
    sample = "full workload"
    if P > 4e9:  # keep the CPU leg bounded: a leading row block of the same matrix
        rows = max(1, int(A.rows * 4e9 / P))
        Hs = H.row_slice(0, rows)
        sample = f"first {rows} of {A.rows} rows"
    else:
        Hs = H
    Cs, _ = po.spgemm(Hs, H, threads=min(avail, 16), with_abs=False)  # warm-up; its arrays are reused
    best, cores = None, 1
    for th in sorted({t for t in (4, 8, 16, 32, 64, 128, avail) if t <= avail}):
        t0 = time.perf_counter()
        n = 0
        while n < 1 or (time.perf_counter() - t0 < 0.5 and n < 10):
            po.spgemm(Hs, H, threads=th, with_abs=False, out=Cs)
            n += 1
        dt = (time.perf_counter() - t0) / n
        if best is None or dt < best:
            best, cores = dt, th
    reps, t_total, Ps = 0, 0.0, po.analysis(Hs, H)["sum_products"]
    while reps < 3 or (t_total < 5.0 and reps < 50):
        t0 = time.perf_counter()
        po.spgemm(Hs, H, threads=cores, with_abs=False, out=Cs)
        t_total += time.perf_counter() - t0
        reps += 1
    return {"value": round(2.0 * Ps * reps / t_total / 1e9, 3), "unit": "GFLOP/s", "cores": cores,
            "kind": "port", "sample": f"{sample}, {reps} runs, symbolic+numeric"}


if __name__ == "__main__":
    main()
