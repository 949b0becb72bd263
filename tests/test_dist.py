"""N > 1 path on CPU: world_size-2 gloo processes run the row-shard partitioning + the gatherv
exchange of speck_amd/sharding.py (the same code bench.py runs over RCCL).  Each rank's shard
product is computed by the oracle here (there is no GPU in the build container); the GPU
kernels themselves are covered by tests/test_gpu_parity.py::test_row_shards_concatenate."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as po
        from speck_amd.sharding import balanced_bounds, gatherv_csr

        A = po.gen_uniform(600, 7, kmin=2, kspan=9, signed=True)
        an = po.analysis(A, A)
        bounds = balanced_bounds(an["row_ops"], world)
        S, _ = po.spgemm(A.row_slice(bounds[rank], bounds[rank + 1]), A)
        cnt = torch.from_numpy(np.diff(S.row_offsets.astype(np.int64)))
        col = torch.from_numpy(S.col_ids.view(np.int32).copy())
        val = torch.from_numpy(S.data.copy())
        out = gatherv_csr(cnt, col, val, root=0)
        if rank == 0:
            C, _ = po.spgemm(A, A)
            ro, c, v = out
            ok = (ro.numpy() == C.row_offsets.astype(np.int64)).all() and \
                 (c.numpy().view(np.uint32) == C.col_ids).all() and (v.numpy() == C.data).all()
            q.put(("ok" if ok else "mismatch", bounds))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def _plan_worker(rank, world, port, q):
    """Pipelined exchange: three multiplies of different values on the same pattern, two slots."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as po
        from speck_amd.sharding import GatherPlan, balanced_bounds

        A = po.gen_uniform(500, 11, kmin=1, kspan=8, signed=True)
        bounds = balanced_bounds(po.analysis(A, A)["row_ops"], world)
        plan, results, ok = None, {}, True
        for step in range(3):
            Ak = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data * (step + 1))
            S, _ = po.spgemm(Ak.row_slice(bounds[rank], bounds[rank + 1]), Ak)
            cnt = torch.from_numpy(np.diff(S.row_offsets.astype(np.int64)))
            col = torch.from_numpy(S.col_ids.view(np.int32).copy())
            val = torch.from_numpy(S.data.copy())
            if plan is None:
                plan = GatherPlan(cnt.numel(), col.numel(), col.dtype, val.dtype, col.device, root=0)
            slot = step % 2
            done = plan.wait(slot)           # the exchange posted two steps ago
            if done is not None:
                results[step - 2] = [t.clone() for t in done]
            plan.start(slot, cnt, col, val)
        for slot, done in enumerate(plan.wait_all()):
            if done is not None:
                results[[s for s in (1, 2) if s % 2 == slot][0]] = [t.clone() for t in done]
        if rank == 0:
            for step in range(3):
                Ak = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data * (step + 1))
                C, _ = po.spgemm(Ak, Ak)
                ro, c, v = results[step]
                ok = ok and (ro.numpy() == C.row_offsets.astype(np.int64)).all() and \
                    (c.numpy().view(np.uint32) == C.col_ids).all() and (v.numpy() == C.data).all()
            q.put("ok" if ok else "mismatch")
        # a shard of another size is refused instead of corrupting the exchange
        try:
            plan.start(0, cnt[:-1], col, val)
            q.put("size change accepted")
        except ValueError:
            pass
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_plan_two_slots():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) == "ok"
    assert q.empty()


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_gatherv_matches_unsharded(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    status, bounds = q.get(timeout=10)
    assert status == "ok"
    assert bounds[0] == 0 and bounds[-1] == 600 and bounds == sorted(bounds)


def test_balanced_bounds_balances_products():
    sys.path.insert(0, ROOT)
    from speck_amd.sharding import balanced_bounds
    rng = np.random.default_rng(0)
    ops = rng.zipf(1.5, size=5000).clip(max=5000).astype(np.uint32)
    b = balanced_bounds(ops, 8)
    cost = ops.astype(np.int64) + 1
    shares = [cost[b[i]:b[i + 1]].sum() for i in range(8)]
    assert max(shares) <= cost.sum() / 8 + cost.max()
    assert balanced_bounds(np.zeros(0, dtype=np.uint32), 4) == [0, 0, 0, 0, 0]
