"""Row-sharded multiply + the library's own gatherv (speck_comm_* / speck_gather_* of the C ABI) on N ranks.
Launched by torch.distributed.run (tests/test_gpu_driver.py); with SPECK_SHARED_GPU=1 every rank uses GPU 0
and the library's host-staged transport, otherwise rank r uses GPU r and RCCL.  torch.distributed (gloo) is
only the launcher: it carries the unique id.  Rank 0 checks the concatenation against the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import torch.distributed as dist  # noqa: E402

import speck_amd as sa  # noqa: E402
from speck_amd.sharding import NativeComm, NativeGatherPlan, TRANSPORT_HOSTMEM, TRANSPORT_RCCL  # noqa: E402


def main():
    shared = os.environ.get("SPECK_SHARED_GPU") == "1"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = 0 if shared else rank
    kind, scale = sys.argv[1], float(sys.argv[2])
    A = sa.gen_matrix(kind, scale, 5, signed=True)
    cfg = sa.spECKConfig.initialize(dev)
    dA = sa.dCSR.from_host(A)
    bounds = sa.partition_rows(dA, dA, cfg, world)
    mine = dA.row_view(bounds[rank], bounds[rank + 1])
    comm = NativeComm(dev, TRANSPORT_HOSTMEM if shared else TRANSPORT_RCCL)
    # one-shot
    dC = sa.dCSR()
    sa.MultiplyspECK(mine, dA, dC, cfg)
    full = comm.gatherv(dC, A.cols, root=0)
    ok = True
    if rank == 0:
        from oracle import pyoracle as po
        H = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data)
        R, ab = po.spgemm(H, H)

        def same(m, tag):
            got = m.to_host()
            good = (got.rows == R.rows and got.nnz == R.nnz and (got.row_offsets == R.row_offsets).all() and
                    (got.col_ids == R.col_ids).all() and (np.abs(got.data - R.data) <= 1e-12 * ab + 1e-300).all())
            if not good:
                print("MISMATCH", tag, got.rows, got.nnz, R.rows, R.nnz, flush=True)
            return good
        ok &= same(full, "one-shot")
    # repeated: two slots, the exchange of step k under the multiply of step k + 1
    outs = [sa.dCSR(), sa.dCSR()]
    cfgs = [cfg, sa.spECKConfig.initialize(dev)]
    plan = None
    for step in range(5):
        slot = step % 2
        if plan is not None:
            v = plan.wait(slot)
            if rank == 0 and v is not None:
                ok &= same(v, f"slot {slot} before step {step}")
        sa.MultiplyspECK(mine, dA, outs[slot], cfgs[slot])
        if plan is None:
            plan = NativeGatherPlan(comm, outs[slot].rows, A.cols, outs[slot].nnz, 8, root=0, slots=2)
            if rank == 0:
                assert plan.r_off == bounds, (plan.r_off, bounds)
        plan.start(slot, outs[slot])
    for v in plan.wait_all():
        if rank == 0 and v is not None:
            ok &= same(v, "drain")
    plan.close()
    comm.close()
    for c in cfgs:
        c.cleanup()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("NATIVE_GATHER_OK" if int(flag.item()) == 1 else "NATIVE_GATHER_FAILED", world, kind, flush=True)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
