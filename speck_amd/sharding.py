"""Row-sharded SpGEMM across the GPUs of one node (new functionality; the reference is
single-GPU -- SURVEY.md 8e).

Rows of C are independent: rank p multiplies the contiguous row range [b_p, b_{p+1}) of A
(balanced by the analysis pass' per-row product counts) with a replicated B, then ONE
exchange step concatenates the shards: an all_gather of the shard sizes followed by a
gatherv of col_ids / data / per-row nnz to the root.  RCCL has no gatherv
(rccl.h: ncclGather/ncclAllGather are equal-count), so it is a batch of point-to-point
send/recv -- each peer->root transfer rides exactly one xGMI link.

Works with any torch.distributed backend ("nccl" == RCCL on ROCm, "gloo" on CPU for tests).
"""
import numpy as np
import torch
import torch.distributed as dist


def balanced_bounds(row_ops, parts):
    """Contiguous row ranges with ~equal sum(row_ops + 1). Same rule as speck_partition_rows."""
    m = len(row_ops)
    cost = np.asarray(row_ops, dtype=np.uint64) + np.uint64(1)
    total = int(cost.sum())
    run = np.cumsum(cost, dtype=np.uint64)
    bounds = [0]
    for p in range(1, parts):
        # first row count i such that run[i-1] * parts >= total * p
        target = -(-total * p // parts)
        i = int(np.searchsorted(run, target, side="left")) + 1
        bounds.append(min(max(i, bounds[-1]), m))
    bounds.append(m)
    return bounds


def gatherv_csr(row_nnz, col_ids, data, root=0, group=None):
    """Concatenate per-rank CSR shards on `root`.

    row_nnz : int64/uint32 tensor [rows_p]   nnz of each local C row
    col_ids : int32 tensor [nnz_p] (u32 bit pattern), data : float tensor [nnz_p]
    Returns (row_offsets[int64, rows+1], col_ids, data) on root, None elsewhere.
    """
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    dev = col_ids.device
    sizes = torch.tensor([row_nnz.numel(), col_ids.numel()], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = torch.stack(all_sizes).cpu().numpy()
    rows_p, nnz_p = all_sizes[:, 0], all_sizes[:, 1]
    row_nnz = row_nnz.to(torch.int32).contiguous()

    if rank != root:
        ops = []
        for t in (row_nnz, col_ids, data):
            if t.numel():
                ops.append(dist.P2POp(dist.isend, t, root, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return None

    total_rows, total_nnz = int(rows_p.sum()), int(nnz_p.sum())
    out_cnt = torch.empty(total_rows, dtype=torch.int32, device=dev)
    out_col = torch.empty(total_nnz, dtype=col_ids.dtype, device=dev)
    out_val = torch.empty(total_nnz, dtype=data.dtype, device=dev)
    r_off = np.concatenate([[0], np.cumsum(rows_p)])
    n_off = np.concatenate([[0], np.cumsum(nnz_p)])
    ops = []
    for p in range(world):
        rs, ns = slice(int(r_off[p]), int(r_off[p + 1])), slice(int(n_off[p]), int(n_off[p + 1]))
        if p == root:
            out_cnt[rs] = row_nnz
            out_col[ns] = col_ids
            out_val[ns] = data
            continue
        for dst, sl in ((out_cnt, rs), (out_col, ns), (out_val, ns)):
            if sl.stop > sl.start:
                ops.append(dist.P2POp(dist.irecv, dst[sl], p, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    row_offsets = torch.zeros(total_rows + 1, dtype=torch.int64, device=dev)
    torch.cumsum(out_cnt.to(torch.int64), 0, out=row_offsets[1:])
    return row_offsets, out_col, out_val


class GatherPlan:
    """gatherv_csr for a REPEATED exchange (the benchmark loop, an iterative method multiplying
    the same pattern): the shard sizes are exchanged once, the root's output buffers are
    allocated once per slot, and start() only posts the point-to-point transfers and returns --
    they run on the communication stream while the next multiply computes.  wait() completes a
    slot; a slot's source tensors must stay untouched until then (alternate two output matrices).

    Every rank must call start()/wait() in the same order, and every shard must keep the size
    it had when the plan was made (start() checks the local shard).
    """

    def __init__(self, rows_local, nnz_local, col_dtype, val_dtype, device, root=0, group=None, slots=2,
                 stage_on_host=False):
        self.group, self.root = group, root
        # gloo cannot send device tensors point to point: the transfers then go through host copies
        # (tests and the shared-GPU plumbing check of bench.py; RCCL sends device memory directly)
        self.stage_on_host = stage_on_host
        self.device = device
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.rows_local, self.nnz_local = int(rows_local), int(nnz_local)
        meta_dev = "cpu" if stage_on_host else device
        sizes = torch.tensor([self.rows_local, self.nnz_local], dtype=torch.int64, device=meta_dev)
        all_sizes = [torch.zeros(2, dtype=torch.int64, device=meta_dev) for _ in range(self.world)]
        dist.all_gather(all_sizes, sizes, group=group)
        all_sizes = torch.stack(all_sizes).cpu().numpy()
        self.r_off = np.concatenate([[0], np.cumsum(all_sizes[:, 0])]).astype(np.int64)
        self.n_off = np.concatenate([[0], np.cumsum(all_sizes[:, 1])]).astype(np.int64)
        self.pending = [None] * slots
        self.out = [None] * slots
        if self.rank == root:
            rows, nnz = int(self.r_off[-1]), int(self.n_off[-1])
            out_dev = "cpu" if stage_on_host else device
            self.out = [(torch.empty(rows, dtype=torch.int32, device=out_dev),
                         torch.empty(nnz, dtype=col_dtype, device=out_dev),
                         torch.empty(nnz, dtype=val_dtype, device=out_dev)) for _ in range(slots)]

    def start(self, slot, row_nnz, col_ids, data):
        assert self.pending[slot] is None, "slot still in flight: wait() first"
        if row_nnz.numel() != self.rows_local or col_ids.numel() != self.nnz_local:
            raise ValueError("shard size changed since the plan was made: build a new GatherPlan")
        row_nnz = row_nnz.to(torch.int32).contiguous()
        if self.stage_on_host:
            row_nnz, col_ids, data = row_nnz.cpu(), col_ids.cpu(), data.cpu()
        ops, keep = [], (row_nnz, col_ids, data)
        if self.rank != self.root:
            for t in keep:
                if t.numel():
                    ops.append(dist.P2POp(dist.isend, t, self.root, self.group))
        else:
            cnt, col, val = self.out[slot]
            for p in range(self.world):
                rs = slice(int(self.r_off[p]), int(self.r_off[p + 1]))
                ns = slice(int(self.n_off[p]), int(self.n_off[p + 1]))
                if p == self.root:
                    cnt[rs].copy_(row_nnz)
                    col[ns].copy_(col_ids)
                    val[ns].copy_(data)
                    continue
                for dst, sl in ((cnt, rs), (col, ns), (val, ns)):
                    if sl.stop > sl.start:
                        ops.append(dist.P2POp(dist.irecv, dst[sl], p, self.group))
        works = dist.batch_isend_irecv(ops) if ops else []
        self.pending[slot] = (works, keep)

    def wait(self, slot):
        """Complete the exchange of `slot`.  Returns (row_offsets, col_ids, data) on the root
        (views of the slot's buffers, valid until the slot is started again), None elsewhere."""
        if self.pending[slot] is None:
            return None
        works, _keep = self.pending[slot]
        for w in works:
            w.wait()
        if _keep[1].is_cuda:
            # RCCL work.wait() only orders the current stream behind the transfer; the caller is about to
            # overwrite the sources from another stream, so complete it on the host -- with an EVENT recorded
            # behind the transfer, not a synchronisation of everything else queued on the stream
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            ev.synchronize()
        self.pending[slot] = None
        if self.rank != self.root:
            return None
        cnt, col, val = self.out[slot]
        if self.stage_on_host and torch.device(self.device).type != "cpu":
            cnt, col, val = cnt.to(self.device), col.to(self.device), val.to(self.device)
        row_offsets = torch.zeros(cnt.numel() + 1, dtype=torch.int64, device=cnt.device)
        torch.cumsum(cnt.to(torch.int64), 0, out=row_offsets[1:])
        return row_offsets, col, val

    def wait_all(self):
        return [self.wait(s) for s in range(len(self.pending))]


# ---------------------------------------------------------------------------------------------------------------
# The exchange through the library's own C ABI (speck_comm_* / speck_gather_*: RCCL resolved inside
# libspeck_amd.so, or its host-staged transport when the ranks share one GPU).  torch.distributed is only the
# LAUNCHER here: it carries the 128-byte unique id from rank 0 to the others.
TRANSPORT_RCCL, TRANSPORT_HOSTMEM = 0, 1


class NativeComm:
    """speck_comm over all ranks of the default (or given) process group."""

    def __init__(self, device_index, transport=TRANSPORT_RCCL, group=None):
        import ctypes as C
        from . import _lib
        from .api import _check
        self._lib = _lib.load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        ident = (C.c_ubyte * 128)()
        if self.rank == 0:
            _check(self._lib.speck_comm_unique_id(int(transport), ident), "speck_comm_unique_id")
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0, group=group)   # launcher duty: hand the id to every rank
        ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
        self._h = C.c_void_p()
        _check(self._lib.speck_comm_init(int(device_index), self.world, self.rank, int(transport), ident,
                                         C.byref(self._h)), "speck_comm_init")
        self.transport = transport

    def gatherv(self, shard, cols, root=0):
        """One-shot speck_gatherv_csr: returns the concatenated dCSR on root, None elsewhere."""
        import ctypes as C
        from .api import _check, dCSR
        full = dCSR(shard.dtype)
        _check(self._lib.speck_gatherv_csr(self._h, root, C.byref(shard._c), int(cols), shard.dtype.itemsize,
                                           C.byref(full._c)), "speck_gatherv_csr")
        return full if self.rank == root else None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.speck_comm_destroy(self._h)
            self._h = None


class NativeGatherPlan:
    """speck_gather_plan: same start / wait / wait_all protocol as GatherPlan, on dCSR shards."""

    def __init__(self, comm, rows_local, cols, nnz_local, value_size=8, root=0, slots=2):
        import ctypes as C
        from .api import _check
        self.comm, self.root, self.slots = comm, root, slots
        self._dtype = np.float64 if value_size == 8 else np.float32
        self._lib = comm._lib
        self._h = C.c_void_p()
        _check(self._lib.speck_gather_plan_create(comm._h, root, int(rows_local), int(cols), int(nnz_local),
                                                  int(value_size), int(slots), C.byref(self._h)),
               "speck_gather_plan_create")
        self._keep = [None] * slots
        r = (C.c_uint64 * (comm.world + 1))()
        n = (C.c_uint64 * (comm.world + 1))()
        _check(self._lib.speck_gather_plan_layout(self._h, r, n))
        self.r_off, self.n_off = [int(x) for x in r], [int(x) for x in n]

    def start(self, slot, shard):
        import ctypes as C
        from .api import _check
        _check(self._lib.speck_gather_start(self._h, int(slot), C.byref(shard._c)), "speck_gather_start")
        self._keep[slot] = shard

    def wait(self, slot):
        """Completes the slot; on the root returns a non-owning dCSR view of the concatenated matrix."""
        import ctypes as C
        from .api import _check, dCSR
        from ._lib import DCsr
        view = DCsr()
        _check(self._lib.speck_gather_wait(self._h, int(slot), C.byref(view)), "speck_gather_wait")
        self._keep[slot] = None
        if self.comm.rank != self.root or not view.row_offsets:
            return None
        v = dCSR(self._dtype)
        v._owner = False
        v._keep = self
        v._c = view
        return v

    def wait_all(self):
        return [self.wait(s) for s in range(self.slots)]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.speck_gather_plan_destroy(self._h)
            self._h = None
