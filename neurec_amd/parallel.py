"""One process per GPU over RCCL (torch.distributed backend "nccl" on ROCm); gloo on CPU
for the multi-process tests.

How the hot path shards:
  * sampler    — every rank owns a contiguous slice of each epoch's triplet stream
                 (units are independent; no exchange).
  * training   — tables replicated; the step's one exchange is either
                 (a) "triplets" (default): every rank samples B triplets, the 12·B bytes of ids
                     are all-gathered (prefetched one step ahead, so the collective hides behind
                     the previous step) and every rank runs the step on the world·B global batch —
                     bit-identical tables on all ranks, no gradient traffic; or
                 (b) "allreduce": each rank back-propagates its own batch and the dense dL/dE0
                     ([N][d] fp32) is summed with one all-reduce before Adam.
                 Both equal the reference run with batch_size = world * B.  The graph
                 propagation itself is replicated (the whole gowalla-size problem is 0.03 % of
                 one GPU's HBM): its cost is per step, not per triplet, so more ranks amortise
                 it over more triplets — DESIGN.md states what that does and does not buy.
  * evaluation — test users are split across ranks (independent units); the metric column
                 sums are all-reduced once at the end.
"""
import os

import torch
import torch.distributed as dist


class Comm:
    def __init__(self, rank=0, world=1, local_rank=0, backend=None):
        self.rank, self.world, self.local_rank, self.backend = rank, world, local_rank, backend

    @property
    def active(self):
        return self.world > 1

    def barrier(self):
        if self.active:
            dist.barrier()

    def allreduce_sum_(self, t):
        """In-place SUM all-reduce (RCCL on GPU tensors, gloo on CPU tensors)."""
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def allgather_cat_start(self, parts):
        """Begin all-gathering equal-length 1-D tensors (e.g. a batch's users/pos/neg ids).
        Returns a token for allgather_cat_finish; the collective runs asynchronously."""
        if not self.active:
            return (list(parts), None, None)
        packed = torch.stack([p.contiguous() for p in parts]).contiguous()        # [P][k]
        out = torch.empty((self.world,) + tuple(packed.shape), dtype=packed.dtype,
                          device=packed.device)
        if self.backend == "nccl":
            work = dist.all_gather_into_tensor(out, packed, async_op=True)
            return (None, out, work)
        # gloo (tests: CPU tensors, or two ranks sharing one GPU) gathers host tensors only
        host_in = packed.cpu()
        host_out = torch.empty((self.world,) + tuple(packed.shape), dtype=packed.dtype)
        work = dist.all_gather([host_out[r] for r in range(self.world)], host_in, async_op=True)
        return (packed.device, host_out, work)

    def allgather_cat_finish(self, token):
        """-> list of P tensors, each the rank-major concatenation [world * k]."""
        parts, out, work = token
        if work is None:
            return parts
        work.wait()
        if parts is not None:                      # gloo path: `parts` holds the target device
            out = out.to(parts)
        return [out[:, i, :].reshape(-1).contiguous() for i in range(out.shape[1])]

    def max_float(self, x):
        if not self.active:
            return float(x)
        dev = "cuda" if (self.backend == "nccl" and torch.cuda.is_available()) else "cpu"
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def shutdown(self):
        if self.active and dist.is_initialized():
            dist.destroy_process_group()


def init_from_env(backend=None):
    """Read RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (set by torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = os.environ.get("NEUREC_DIST_BACKEND") or \
            ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        # one process per GPU; with fewer GPUs than ranks (tests: 2 ranks on one GPU over gloo)
        # ranks share devices round-robin
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return Comm(rank, world, local_rank, backend)


def partition(n, rank, world):
    """Contiguous block partition of range(n): rank r owns [lo, hi)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def shard_users(user_ids, rank, world):
    lo, hi = partition(len(user_ids), rank, world)
    return user_ids[lo:hi]
