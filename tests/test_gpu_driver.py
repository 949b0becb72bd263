"""The runspECK driver (reference source/Executor.cpp:13-81) end to end on the GPU, including its
CompareResult path -- here against rocSPARSE SpGEMM, the stand-in for the reference's cuSPARSE check."""
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "apps", "runspECK")


def run(args, cwd):
    assert os.path.exists(EXE), "apps/runspECK missing: run `make` / __graft_entry__.build()"
    p = subprocess.run([EXE] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    return p.returncode, p.stdout.decode()


def write_ini(path, **kv):
    with open(path, "w") as f:
        for k, v in kv.items():
            f.write(f"{k}={v}\n")


# The one third-party pin of parity (the reference ships no golden vectors): rocSPARSE
# SpGEMM, the stand-in for the reference's cuSPARSE compare (source/Executor.cpp:29-40).  Structure
# bit-exact AND values within 1e-12 * sum|a*b| per entry -- the driver exits non-zero on either.  Same
# inputs and scales as tests/test_gpu_parity.py::test_suitesparse_standins_full_parity, plus config #1.
@pytest.mark.parametrize("spec", ["gen:uniform:1.0:42", "gen:scircuit:1.0:1", "gen:mac_econ:1.0:1",
                                  "gen:cant:0.25:1", "gen:webbase:0.1:1", "gen:nlpkkt:0.002:1",
                                  "gen:scircuit:0.1:3", "gen:webbase:0.02:3"])
def test_driver_matches_rocsparse(tmp_path, spec):
    ini = tmp_path / "config.ini"
    write_ini(ini, TrackCompleteTimes="true", TrackIndividualTimes="false", CompareResult="true",
              IterationsWarmUp=2, IterationsExecution=3)
    rc, out = run([spec, str(ini)], tmp_path)
    assert rc == 0, out
    assert "compare vs rocSPARSE: ok" in out, out
    assert "Error:" not in out
    m = re.search(r"var-SpGEMM -> NNZ: (\d+)", out)
    assert m and int(m.group(1)) > 0
    m = re.search(r"var-SpGEMM SpGEMM: ([0-9.eE+-]+) ms", out)
    assert m and float(m.group(1)) > 0


def test_driver_mtx_cache_and_rectangular(tmp_path):
    src = os.path.join(ROOT, "tests", "golden", "formats", "general_real.mtx")   # 4 x 5: B = A^T
    mtx = tmp_path / "m.mtx"
    shutil.copy(src, mtx)
    ini = tmp_path / "config.ini"
    write_ini(ini, CompareResult="true", IterationsWarmUp=1, IterationsExecution=2)
    rc, out = run([str(mtx), str(ini)], tmp_path)
    assert rc == 0, out
    assert "Matrix: 4x5: 7 nonzeros" in out
    assert os.path.exists(str(mtx) + "d_.hicsr")                  # DataLoader.cpp cache file name
    assert "compare vs rocSPARSE: ok" in out
    rc, out2 = run([str(mtx), str(ini)], tmp_path)                # second run is served by the cache
    assert rc == 0 and "successfully loaded: " in out2


def test_driver_individual_times_table(tmp_path):
    ini = tmp_path / "config.ini"
    write_ini(ini, TrackIndividualTimes="true", IterationsWarmUp=1, IterationsExecution=1)
    rc, out = run(["gen:mac_econ:0.05:1", str(ini)], tmp_path)
    assert rc == 0, out
    assert "spECK      numeric kernel = " in out and "spECK     counting kernel = " in out
    # the stage fields the reference fills (Timings.h:7-18) are non-zero: analysis (+ symbolic binning: one kernel since
    # round 5, so "load-balancer" reports 0), symbolic, scan + numeric binning, numeric (sorting is fused into the numeric
    # kernels and reports 0)
    vals = {k.strip(): float(v) for k, v in re.findall(r"spECK\s+([A-Za-z -]+?) = ([0-9.eE+-]+) ms", out)}
    for k in ("count computations", "counting kernel", "num load-balancer", "numeric kernel"):
        assert vals[k] > 0, (k, vals)


def _bench_line(out):
    """The ONE JSON line bench.py prints: <= 4 KB, the driver's contract fields, no prose."""
    import json
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    assert len(lines[0]) <= 4096, len(lines[0])
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "verified", "phases_ms", "roofline"):
        assert k in d, k
    assert d["config"]["timed_call"].startswith("complete")
    return d


def test_bench_two_ranks_share_the_gpu(tmp_path):
    """Plumbing of bench.py's N > 1 path (row shards, two alternating output matrices, pipelined
    gatherv) with both ranks on GPU 0 and gloo instead of RCCL -- not a measurement."""
    import json
    import sys
    env = dict(os.environ, SPECK_BENCH_SHARED_GPU="1")
    port = 29600 + os.getpid() % 300
    detail = tmp_path / "detail.json"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--scale", "0.25", "--config5-scale", "0.004", "--config5-steps", "3", "--detail", str(detail)],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-2000:]
    d = _bench_line(out)
    assert d["n_gpus"] == 2 and d["config"]["gather"] and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "rows2" and d["config"]["exchange"] == "native:hostmem"
    assert d["multiply_only"]["value"] > 0 and d["config"]["exchange_floor_ms"] > 0
    # every rank checked its row shard of the last step against the oracle
    full = json.load(open(detail))
    assert d["verified"] is True and full["headline"]["verify"]["ok_all_ranks"] and full["headline"]["verify"]["mode"] == "oracle"
    # (N > 1: every step is the complete call -- the structure-reuse mode is an N = 1 side figure)
    assert d["value_reuse"] is None and full["headline"]["verify"]["replayed"] is False
    # the BASELINE.json configs[4] leg: nlpkkt stand-in, strong scaling, with and without the exchange
    c5 = d["config5"]
    assert c5["scaling"] == "strong" and c5["n_gpus"] == 2 and c5["value"] > 0 and c5["multiply_only"]["value"] > 0
    assert c5["verified"] is True


def test_bench_launches_its_own_ranks_when_called_without_a_launcher(tmp_path):
    """`python bench.py --gpus N` as the driver calls it -- no torch.distributed.run in front: bench.py starts the N
    ranks itself and rank 0 prints the (one, compact) line with n_gpus = N.  On a box with fewer GPUs it refuses (exit
    code 2, no line) instead of printing a one-GPU number, unless the ranks may share GPU 0 (plumbing check)."""
    import sys
    import torch
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--scale", "0.1",
           "--no-config5", "--no-cpu-baseline", "--detail", str(tmp_path / "detail.json")]
    p = subprocess.run(cmd, cwd=ROOT, env=dict(base, SPECK_BENCH_SHARED_GPU="1"), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-2000:]
    d = _bench_line(out)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "rows2" and d["verified"] is True
    assert d["config"]["exchange_floor_ms"] > 0 and d["multiply_only"]["value"] > 0
    if torch.cuda.device_count() < 2:
        p = subprocess.run(cmd, cwd=ROOT, env=base, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 2 and not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]


def test_bench_default_line_is_compact_and_times_the_complete_call(tmp_path):
    """The N = 1 line as the driver parses it (small scale here): `value` is the COMPLETE call -- every stage inside the
    timed region, so its symbolic phase is not a few microseconds of verifier --, the structure-reuse mode of the repeated
    call stands beside it as `value_reuse`, every other configuration is one short object, the nlpkkt leg carries the
    same pair, and everything else is in the detail file."""
    import json
    import sys
    detail = tmp_path / "detail.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--configs-steps", "3",
                        "--config5-scale", "0.01", "--config5-steps", "2", "--no-cpu-baseline", "--detail", str(detail)],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-2000:]
    d = _bench_line(out)
    assert d["n_gpus"] == 1 and d["verified"] is True and d["dtype"] == "f64"
    assert d["ms_reuse"] < d["ms_per_step"] and d["value_reuse"] > d["value"] > 0
    full = json.load(open(detail))
    head = full["headline"]
    # the timed call ran the symbolic launches: its symbolic phase is longer than the symbolic light launch alone
    assert d["phases_ms"]["symbolic"] > head["stages_ms"]["sym_light"] > 0.005
    assert head["verify"]["replayed"] is False and head["verify_reuse"]["replayed"] is True
    r = d["roofline"]
    assert r["kernel"].startswith("numeric:") and 0 < r["frac"] <= 1 and r["avg_launch_ms"] > 0 and r["bytes"] > 0
    for k in ("hbm_measured_frac", "l2_frac", "lds_atomic_frac", "valu_frac"):
        assert r[k] is None or r[k] >= 0, (k, r[k])          # (round 6: nothing is dropped -- a fraction above 1 is printed)
    assert r["bound"] in ("hbm", "l2", "lds_atomic", "valu", "latency")
    # the same-box library baseline beside the metric: rocSPARSE SpGEMM, same input and protocol (apps/runspECK --time-library)
    assert d["lib_baseline"]["kind"] == "rocsparse_spgemm" and d["lib_baseline"]["ms"] > 0
    names = [c["name"] + ":" + c["dtype"] for c in d["configs"]]
    assert names == ["webbase:f64", "mac_econ:f64", "cant:f64", "mac_econ:f32", "cant:f32"]
    for c in d["configs"]:
        assert set(c) - {"lib_ms"} == {"name", "dtype", "ms_per_step", "ms_reuse", "value", "value_reuse", "roofline_frac",
                                       "numeric_phase_frac", "bound", "verified"}
        assert c["verified"] is True and c["ms_per_step"] > 0 and c["ms_reuse"] > 0
        assert c["dtype"] == "f32" or c["lib_ms"] > 0
    c5 = d["config5"]
    assert c5["verified"] is True and c5["ms_per_step"] > 0 and c5["ms_reuse"] > 0 and c5["value_reuse"] > 0
    # ... and the one HBM-scale leg carries its fractions too (its pre-pass is profiled since round 6)
    assert 0 < c5["roofline_frac"] and 0 < c5["numeric_phase_frac"] and c5["bound"] and c5["lib_ms"] > 0
    assert full["line_bytes"] <= 4096 and len(full["headline"]["roofline"]["launches"]) >= 1


def test_bench_falls_back_to_the_torch_exchange_when_the_native_one_fails_its_self_test():
    """bench.py brings the library's exchange up with a tiny self-checked gatherv; if that fails (here: forced on every
    rank) all ranks switch to the torch.distributed exchange together, and the line says so."""
    import json
    import sys
    env = dict(os.environ, SPECK_BENCH_SHARED_GPU="1", SPECK_BENCH_FAIL_NATIVE="1")
    port = 29000 + os.getpid() % 300
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--scale", "0.1", "--no-config5", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-2000:]
    d = _bench_line(out)
    assert d["n_gpus"] == 2 and d["verified"] is True and d["value"] > 0
    assert "torch" in d["config"]["exchange"] and "fell back" in d["config"]["exchange_note"]


def test_bench_strong_scaling_mode_two_ranks_share_the_gpu():
    """`--workload nlpkkt --scaling strong`: the same matrix at every N, sharded by products."""
    import json
    import sys
    env = dict(os.environ, SPECK_BENCH_SHARED_GPU="1")
    port = 29300 + os.getpid() % 300
    sizes = {}
    for n in (1, 2):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] if n == 1 else \
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
             "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")]
        p = subprocess.run(cmd + ["--gpus", str(n), "--steps", "3", "--warmup", "2", "--workload", "nlpkkt", "--scale",
                                  "0.004", "--scaling", "strong", "--no-cpu-baseline", "--no-configs"],
                           cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        out = p.stdout.decode()
        assert p.returncode == 0, out[-2000:]
        d = _bench_line(out)
        assert d["scaling"] == "strong" and d["n_gpus"] == n and d["value"] > 0 and "config5" not in d
        assert d["verified"] is True
        sizes[n] = (d["config"]["rows"], d["config"]["products"], d["config"]["nnzC"])
    assert sizes[1] == sizes[2]          # strong scaling: the same job at every N


@pytest.mark.parametrize("world,kind,scale", [(2, "scircuit", 0.1), (3, "webbase", 0.02)])
def test_native_gatherv_ranks_share_the_gpu(world, kind, scale):
    """The library's own exchange (C ABI: speck_comm_*, speck_gatherv_csr, speck_gather_plan with two slots) on
    2 / 3 ranks that share GPU 0 -- RCCL rejects duplicate devices, so the host-staged transport carries the
    bytes; plan, displacements and the rebase kernel are the ones the RCCL transport uses."""
    import sys
    env = dict(os.environ, SPECK_SHARED_GPU="1")
    port = 29000 + os.getpid() % 250 + world
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "tools", "native_gather_check.py"), kind, str(scale)],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "NATIVE_GATHER_OK" in out, out[-3000:]


def test_native_gatherv_rccl_single_rank():
    """RCCL itself (dlopen'ed by the library): a communicator of one rank on GPU 0, the gatherv degenerates to the
    root's own device-to-device copy + rebase.  (More ranks need more GPUs: the driver's scaling run.)"""
    import sys
    env = dict(os.environ, SPECK_SHARED_GPU="0")
    port = 29500 + os.getpid() % 250
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "tools", "native_gather_check.py"), "mac_econ", "0.05"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "NATIVE_GATHER_OK" in out, out[-3000:]


def test_driver_row_sharded_ranks_share_the_gpu(tmp_path):
    """runspECK --gpus 2 --shared-gpu: the driver re-launches itself as two rank processes (both on GPU 0, the
    library's host-staged transport), each multiplies its row shard, rank 0 receives the concatenation through
    speck_gather_plan and compares it with rocSPARSE."""
    ini = tmp_path / "config.ini"
    write_ini(ini, CompareResult="true", IterationsWarmUp=2, IterationsExecution=3)
    rc, out = run(["gen:scircuit:0.2:3", str(ini), "--gpus", "2", "--shared-gpu"], tmp_path)
    assert rc == 0, out
    assert "compare vs rocSPARSE: ok" in out and "row shards: 2 ranks" in out
    nnz_sharded = int(re.search(r"var-SpGEMM -> NNZ: (\d+)", out).group(1))
    rc, out1 = run(["gen:scircuit:0.2:3", str(ini)], tmp_path)
    assert rc == 0 and int(re.search(r"var-SpGEMM -> NNZ: (\d+)", out1).group(1)) == nnz_sharded


def test_bench_reads_a_symmetric_suitesparse_file(tmp_path):
    """$SPECK_MTX_DIR/nlpkkt160.mtx (here: the stand-in at scale 0.05 written as a `symmetric` lower-triangle file,
    the way SuiteSparse ships the original) goes through the parallel MatrixMarket reader into bench.py:
    "data": "suitesparse", mirrored entry count, output verified against the oracle."""
    import json
    import sys
    import numpy as np
    import speck_amd as sa
    A = sa.gen_matrix("nlpkkt", 0.05, 3, signed=True)
    sa.store_mtx(A, tmp_path / "nlpkkt160.mtx", symmetric_lower=True)
    ro = A.row_offsets.astype(np.int64)
    rows_of = np.repeat(np.arange(A.rows), np.diff(ro))
    n_lower = int((A.col_ids <= rows_of).sum())
    n_diag = int((A.col_ids == rows_of).sum())
    env = dict(os.environ, SPECK_MTX_DIR=str(tmp_path))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "nlpkkt", "--scaling", "strong",
                        "--steps", "3", "--warmup", "2", "--no-configs", "--no-cpu-baseline", "--detail",
                        str(tmp_path / "detail.json")],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-2000:]
    d = _bench_line(out)
    assert d["data"] == "suitesparse" and d["config"]["rows"] == A.rows
    assert d["config"]["nnzA"] == 2 * n_lower - n_diag
    assert d["verified"] is True and json.load(open(tmp_path / "detail.json"))["headline"]["verify"]["mode"] == "oracle"


def test_symmetric_file_with_both_triangles_is_rejected_end_to_end(tmp_path):
    """A `symmetric` file that lists both (i, j) and (j, i): the loader mirrors without deduplication (as the
    reference does), every off-diagonal column appears twice in its row, and the multiply refuses the input
    (SPECK_ERR_UNSORTED) instead of computing garbage."""
    import speck_amd as sa
    p = tmp_path / "both.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real symmetric\n3 3 5\n1 1 1.0\n2 1 2.0\n1 2 3.0\n3 3 4.0\n3 2 5.0\n")
    A = sa.load_mtx(p)
    assert A.nnz == 8
    cfg = sa.spECKConfig.initialize(0)
    try:
        dA, dC = sa.dCSR.from_host(A), sa.dCSR()
        with pytest.raises(sa.SpeckError) as e:
            sa.MultiplyspECK(dA, dA, dC, cfg)
        assert e.value.status == 8 and dC.nnz == 0
    finally:
        cfg.cleanup()
    rc, out = run([str(p)], tmp_path)
    assert "ERROR" in out and "strictly ascending" in out


def test_declarations_only_caller_runs(tmp_path):
    """tests/cpp/caller_decl_only.cpp (g++, SPECK_DECLARATIONS_ONLY) against the exported instantiations: float and
    double products, six live streams and four events in the public spECKConfig fields."""
    from test_host import _build_decl_only_caller
    exe = str(tmp_path / "caller")
    _build_decl_only_caller(exe)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert p.returncode == 0 and "decl-only caller ok" in p.stdout.decode(), p.stdout.decode()


def test_device_to_device_convert_makes_no_host_round_trip(tmp_path):
    """convert(dCSR&, const dCSR&, padding) (reference source/dCSR.cpp:81-89) through speck_dcsr_copy: own buffers,
    identical contents, a row-range view rebased to 0, the padding honoured -- and no device-to-host copy: under
    rocprofv3 --memory-copy-trace the K = 1 and the K = 9 run of tests/cpp/convert_d2d.cpp show the SAME number of
    device-to-host copies (those of the final download that checks the contents)."""
    import csv
    import glob
    exe = str(tmp_path / "convert_d2d")
    subprocess.check_call(["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                           "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "convert_d2d.cpp"),
                           "-L", os.path.join(ROOT, "speck_amd"), "-lspeck_amd", "-L", "/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.join(ROOT, "speck_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert p.returncode == 0 and "convert d2d ok" in p.stdout.decode(), p.stdout.decode()
    counts = {}
    for k in (1, 9):
        out = tmp_path / f"trace{k}"
        env = dict(os.environ, TMPDIR=str(tmp_path))
        q = subprocess.run(["rocprofv3", "--memory-copy-trace", "--output-format", "csv", "-d", str(out), "-o", "r", "--",
                            exe, str(k)], cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=300)
        assert q.returncode == 0 and "convert d2d ok" in q.stdout.decode(), q.stdout.decode()[-1500:]
        files = glob.glob(str(out / "**" / "*memory_copy_trace.csv"), recursive=True)
        assert files, q.stdout.decode()[-1500:]
        rows = list(csv.DictReader(open(files[0])))
        counts[k] = sum(1 for r in rows if "DEVICE_TO_HOST" in (r.get("Direction") or "").upper().replace(" ", "_"))
        assert len(rows) > 0
    assert counts[1] == counts[9] and counts[1] > 0, counts
