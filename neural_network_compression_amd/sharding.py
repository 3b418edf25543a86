"""Sharding of one flattened weight vector over the GPUs of a node (one process per GPU).

The vector is cut into contiguous shards, one per rank, each starting on a multiple of
8192 elements (NumPy's float32 reduction chunk) so that every rank's per-chunk partial sums
are exactly the chunk sums of the unsharded vector.  The data path needs only small
collectives, all over RCCL (backend "nccl" on ROCm) when the tensors are on GPUs:

  * sigma / mean / var : all-gather of the per-chunk float32 sums (N/8192 floats), then the
    same left-to-right fold on every rank;
  * min / max          : all-reduce MIN / MAX of two floats;
  * histogram, per-cluster fixed-point sums and counts: all-reduce SUM of int64 (exact).

These helpers only move torch tensors, so they run unchanged on CPU tensors over gloo
(that is how tests/test_sharding_gloo.py exercises them without a GPU).
"""
from __future__ import annotations

import torch

CHUNK = 8192


def shard_bounds(n_total: int, world: int, rank: int):
    """[start, stop) of `rank`'s shard: equal numbers of whole 8192-chunks, remainder to the last ranks."""
    nchunks = (n_total + CHUNK - 1) // CHUNK
    base, extra = divmod(nchunks, world)
    # the first (world - extra) ranks get `base` chunks, the rest `base + 1`, so the ragged
    # tail chunk always belongs to the last rank
    first_big = world - extra
    c0 = rank * base + max(0, rank - first_big)
    c1 = c0 + base + (1 if rank >= first_big else 0)
    return min(c0 * CHUNK, n_total), min(c1 * CHUNK, n_total)


def _world(group):
    import torch.distributed as dist

    return dist.get_world_size(group)


def gather_chunks(chunks: torch.Tensor, group) -> torch.Tensor:
    """Concatenate every rank's per-chunk sums in rank order (ragged all-gather)."""
    import torch.distributed as dist

    world = _world(group)
    n_local = torch.tensor([chunks.numel()], dtype=torch.int64, device=chunks.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    padded = torch.zeros(mx, dtype=chunks.dtype, device=chunks.device)
    padded[: chunks.numel()] = chunks
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])


def allreduce_minmax(mm: torch.Tensor, group) -> torch.Tensor:
    """mm = float32[2] {min, max} of this shard (+-inf for an empty shard) -> global."""
    import torch.distributed as dist

    mn, mx = mm[0:1].clone(), mm[1:2].clone()
    dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    return torch.cat([mn, mx])


def allreduce_sum_(t: torch.Tensor, group) -> torch.Tensor:
    import torch.distributed as dist

    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def total_count(n_local: int, device, group) -> int:
    t = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    return int(allreduce_sum_(t, group).item())


class RcclComm:
    """The library's own RCCL communicator (include/nnc.h, nnc_comm_*): with it the per-iteration exchange of a sharded
    fit -- streaming pass, all-reduce of the 2K int64 sums / counts, finalize -- is enqueued by ONE call into the C
    library per batch of iterations (nnc_kmeans_iterate_sharded), like the single-GPU loop, instead of three
    Python-issued calls per iteration.  Built from a torch.distributed group of one rank per GPU, which only
    carries the 128-byte id to the other ranks."""

    def __init__(self, group, device: torch.device):
        import ctypes

        import torch.distributed as dist

        from . import _native as nat

        self.L = nat.load()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        torch.cuda.set_device(device)
        self.handle = None
        backend = dist.get_backend(group)
        cdev = device if backend == "nccl" else "cpu"
        # ncclCommInitRank blocks until every rank has entered it: the ranks first agree, over the caller's group, that all of them
        # can bind librccl and that rank 0 has an id to hand out -- a rank that cannot raises here, together with all the others,
        # instead of leaving them inside the init for ever
        buf = (ctypes.c_ubyte * 128)()
        ok = 1 if self.L.nnc_comm_available() == 0 else 0
        why = "" if ok else self.L.nnc_last_error().decode("utf-8", "replace")
        if ok and self.rank == 0 and self.L.nnc_comm_unique_id(buf, 128) != 0:
            ok, why = 0, self.L.nnc_last_error().decode("utf-8", "replace")
        flag = torch.tensor([ok], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            raise nat.NativeLibraryError("the library's RCCL communicator cannot be created on every rank" + (f" (rank {self.rank}: {why})" if why else ""))
        idt = torch.tensor(list(buf), dtype=torch.uint8, device=cdev)
        dist.broadcast(idt, src=dist.get_global_rank(group, 0), group=group)
        raw = bytes(idt.cpu().tolist())
        idbuf = ctypes.create_string_buffer(raw, 128)
        handle = ctypes.c_void_p()
        nat.check(self.L.nnc_comm_init(ctypes.byref(handle), idbuf, 128, self.rank, self.world))
        self.handle = handle

    def allreduce_(self, t: torch.Tensor, op: str = "sum") -> torch.Tensor:
        from . import _native as nat

        dt = {torch.int64: 0, torch.int32: 1, torch.float32: 2}[t.dtype]
        nat.check(self.L.nnc_comm_allreduce(self.handle, t.data_ptr(), t.numel(), dt, {"sum": 0, "max": 1, "min": 2}[op],
                                            torch.cuda.current_stream(t.device).cuda_stream))
        return t

    def close(self):
        if getattr(self, "handle", None):
            self.L.nnc_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover - interpreter teardown order
        try:
            self.close()
        except Exception:
            pass
