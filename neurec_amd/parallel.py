"""One process per GPU over RCCL (torch.distributed backend "nccl" on ROCm); gloo on CPU
for the multi-process tests.

How the hot path shards:
  * sampler    — every rank owns a contiguous slice of each epoch's triplet stream
                 (units are independent; no exchange).
  * training   — tables replicated; per step either
                 (a) "replicated" (default): NO exchange — the sampler is a counter-based generator
                     (seed, epoch, position), so every rank produces the same global epoch stream
                     itself (47 us per 814 k-triplet epoch) and steps on the same world·B global
                     batch, whose batch plans come from the sampler a whole epoch ahead;
                 (b) "triplets": every rank samples B triplets of its own slice, the 12·B bytes of
                     ids are all-gathered (prefetched one step ahead, so the collective hides behind
                     the previous step) and every rank runs the step on the world·B global batch —
                     bit-identical tables on all ranks, no gradient traffic (the plan of the
                     global batch is then sorted inside the step); or
                 (c) "allreduce": each rank back-propagates its own batch and the dense dL/dE0
                     ([N][d] fp32) is summed with one all-reduce before Adam.
                 All equal the reference run with batch_size = world * B.  The graph
                 propagation itself is replicated (the whole gowalla-size problem is 0.03 % of
                 one GPU's HBM): its cost is per step, not per triplet, so more ranks amortise
                 it over more triplets — DESIGN.md states what that does and does not buy.
  * evaluation — test users are split across ranks (independent units); the metric column
                 sums are all-reduced once at the end.
"""
import os

import torch
import torch.distributed as dist


class _RcclDirect:
    """ncclAllGather issued on torch's CURRENT stream through librccl's C API (r06, VERDICT r5 #7b).

    ProcessGroupNCCL runs a collective on its own stream: an event recorded on the compute stream, a wait on the
    collective stream, the collective, an event back — measured at +34 us on the device timeline of the column-sharded
    step before a byte moves (profiles/r05_exp_colshard_nccl_overhead.txt), 17 % of that step.  The step's ONE exchange
    (12 B per triplet and rank) has nothing to overlap with, so it is issued where the kernels around it are: one more
    launch on the compute stream, ordered like them.  A second RCCL communicator next to torch's, created from a unique
    id that rank 0 broadcasts through the process group.  Anything that fails here leaves the torch path in charge."""

    DTYPES = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6, torch.float32: 7,
              torch.float64: 8}

    def __init__(self, rank, world):
        """loads librccl and, on rank 0, draws the unique id: LOCAL work only — connect() holds the collective part, so
        that Comm._rccl_direct can make every rank agree on the outcome of each half (a rank that fell back to the
        process group while the others issue on this communicator would hang the job)"""
        import ctypes as C

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        self._C, self.rank, self.world, self.comm = C, int(rank), int(world), None
        self.lib = lib = C.CDLL(os.environ.get("NEUREC_RCCL_LIB", "librccl.so"))
        lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        self._uid = UniqueId()
        if self.rank == 0 and lib.ncclGetUniqueId(C.byref(self._uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")

    def uid_bytes(self):
        return bytes(self._uid)

    def connect(self, uid_bytes):
        """collective: every rank calls it with rank 0's id"""
        C = self._C
        C.memmove(C.byref(self._uid), uid_bytes, 128)
        self.comm = C.c_void_p()
        rc = self.lib.ncclCommInitRank(C.byref(self.comm), self.world, self._uid, self.rank)
        if rc != 0:
            self.comm = None
            raise RuntimeError("ncclCommInitRank failed with %d" % rc)

    def self_check(self):
        """one tiny call of each entry point against the answer it must give (the dtype codes, the argument order and
        the in-place form are this file's reading of rccl.h: a library that disagrees shows here, not in a table)"""
        w, r, dev = self.world, self.rank, torch.device("cuda", torch.cuda.current_device())
        got = torch.empty(w, dtype=torch.int32, device=dev)
        self.all_gather(torch.full((1,), r, dtype=torch.int32, device=dev), got)
        s64 = torch.full((3,), float(r + 1), dtype=torch.float64, device=dev)
        self.all_reduce_sum(s64)
        send = (torch.arange(w, dtype=torch.float32, device=dev) + r * w).reshape(w, 1).repeat(1, 4).contiguous()
        recv = torch.empty_like(send)
        self.all_to_all_rows(send, [1] * w, recv, [1] * w)
        torch.cuda.current_stream().synchronize()
        want = torch.arange(w, dtype=torch.float32) * w + r
        if not (got.cpu().tolist() == list(range(w)) and s64.cpu().tolist() == [w * (w + 1) / 2.0] * 3
                and torch.equal(recv.cpu(), want.reshape(w, 1).repeat(1, 4))):
            raise RuntimeError("the direct communicator's self-check returned wrong values")

    def all_gather(self, local, out):
        C = self._C
        rc = self.lib.ncclAllGather(C.c_void_p(local.data_ptr()), C.c_void_p(out.data_ptr()), C.c_size_t(local.numel()),
                                    self.DTYPES[local.dtype], self.comm, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError("ncclAllGather failed with %d" % rc)

    def all_reduce_sum(self, t):
        C = self._C
        rc = self.lib.ncclAllReduce(C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), C.c_size_t(t.numel()),
                                    self.DTYPES[t.dtype], 0, self.comm,                     # 0 = ncclSum; in place
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError("ncclAllReduce failed with %d" % rc)

    def all_to_all_rows(self, send, send_counts, recv, recv_counts):
        """rows [n][...] ordered by destination -> rows ordered by source: one group of ncclSend / ncclRecv pairs"""
        C, lib = self._C, self.lib
        row = 1
        for x in send.shape[1:]:
            row *= int(x)
        esz, dt, st = send.element_size(), self.DTYPES[send.dtype], C.c_void_p(torch.cuda.current_stream().cuda_stream)
        sp, rp, so, ro = send.data_ptr(), recv.data_ptr(), 0, 0
        rc = lib.ncclGroupStart()
        for r in range(len(send_counts)):
            if send_counts[r] and rc == 0:
                rc = lib.ncclSend(C.c_void_p(sp + so * row * esz), C.c_size_t(send_counts[r] * row), dt, r, self.comm, st)
            if recv_counts[r] and rc == 0:
                rc = lib.ncclRecv(C.c_void_p(rp + ro * row * esz), C.c_size_t(recv_counts[r] * row), dt, r, self.comm, st)
            so += send_counts[r]
            ro += recv_counts[r]
        rc2 = lib.ncclGroupEnd()
        if rc != 0 or rc2 != 0:
            raise RuntimeError("ncclSend / ncclRecv group failed with %d / %d" % (rc, rc2))

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


class Comm:
    """Collectives of the hot path.  STREAM CONTRACT (backend "nccl" = RCCL): every kernel of this package is
    launched through ctypes on torch's CURRENT stream (engine._stream()); ProcessGroupNCCL runs a collective on its
    own stream, ordered behind the work enqueued on the current stream when the collective is issued.  Every
    method below therefore returns only after `work.wait()` has made the CURRENT stream wait for the collective
    (a stream-side dependency, no host synchronisation) — or hands out the work object as a token whose
    `*_finish` does that — so the next ctypes launch reads what the collective wrote.  NEUREC_DIST_DEBUG_SYNC=1
    brackets every collective with torch.cuda.synchronize(): a run must give the same bits with and without it
    (tests/test_rccl_gpu.py), which is how a missing dependency would show."""

    def __init__(self, rank=0, world=1, local_rank=0, backend=None, force=False):
        self.rank, self.world, self.local_rank, self.backend = rank, world, local_rank, backend
        # force: run the collectives through the process group even at world size 1 (the one-GPU RCCL test: the
        # library loads, every call signature / dtype / split argument is accepted)
        self.force = bool(force)
        self.debug_sync = os.environ.get("NEUREC_DIST_DEBUG_SYNC", "") == "1"
        self.calls = {}                                      # collective name -> times issued (tests, bench line)
        self._direct, self._direct_tried = None, False       # the RCCL communicator of all_gather_rows (made on first use)

    def _rccl_direct(self):
        """librccl on the compute stream for the blocking collectives (backend nccl; NEUREC_RCCL_DIRECT=0: torch's).
        COLLECTIVE on first use: after each half of the set-up (library loaded / communicator connected and its
        self-check passed) the ranks take the minimum of their outcomes through the process group, so that either every
        rank issues on the direct communicator or none does."""
        if not self._direct_tried:
            self._direct_tried = True
            if self.backend == "nccl" and os.environ.get("NEUREC_RCCL_DIRECT", "1") != "0" and dist.is_initialized():
                import sys
                dev = torch.device("cuda", torch.cuda.current_device())

                def agreed(ok):
                    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    return bool(flag.item())
                d, why = None, ""
                try:
                    d = _RcclDirect(self.rank, self.world)
                except Exception as e:
                    why = "%s: %s" % (type(e).__name__, e)
                if agreed(d is not None):
                    raw = torch.frombuffer(bytearray(d.uid_bytes()), dtype=torch.uint8).clone().to(dev)
                    dist.broadcast(raw, src=0)
                    ok = True
                    try:
                        d.connect(raw.cpu().numpy().tobytes())
                        d.self_check()
                    except Exception as e:
                        ok, why = False, "%s: %s" % (type(e).__name__, e)
                    if agreed(ok):
                        self._direct = d
                    else:
                        d.close()
                if self._direct is None:                     # torch.distributed stays in charge
                    sys.stderr.write("neurec_amd.parallel: direct RCCL collectives unavailable (%s), using "
                                     "torch.distributed\n" % (why or "another rank could not set them up"))
        return self._direct

    @property
    def active(self):
        """more than one rank: what the ENGINES ask when they choose a mode"""
        return self.world > 1

    @property
    def live(self):
        """collectives go through torch.distributed (more than one rank, or forced at one)"""
        return self.world > 1 or self.force

    def _enter(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1
        if self.debug_sync and torch.cuda.is_available():
            torch.cuda.synchronize()

    def _done(self, work=None):
        """the current stream waits for the collective (see the class docstring)"""
        if work is not None:
            work.wait()
        if self.debug_sync and torch.cuda.is_available():
            torch.cuda.synchronize()

    def barrier(self):
        if self.live:
            if self.backend == "nccl":
                dist.barrier(device_ids=[torch.cuda.current_device()])
            else:
                dist.barrier()

    def allreduce_sum_(self, t):
        """In-place SUM all-reduce (RCCL on GPU tensors, gloo on CPU tensors)."""
        if self.live:
            self._enter("all_reduce")
            if self.backend != "nccl" and t.is_cuda:      # gloo (tests: ranks sharing one GPU): host-staged
                host = t.detach().cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                t.copy_(host)
                self._done()
            else:
                direct = self._rccl_direct() if t.is_cuda else None
                if direct is not None and t.dtype in direct.DTYPES and t.is_contiguous():
                    direct.all_reduce_sum(t)              # on the current stream, as all_gather_rows
                    self._done()
                else:
                    self._done(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))
        return t

    def allgather_cat_start(self, parts):
        """Begin all-gathering equal-length 1-D tensors (e.g. a batch's users/pos/neg ids).
        Returns a token for allgather_cat_finish; the collective runs asynchronously."""
        if not self.live:
            return (list(parts), None, None)
        self._enter("all_gather_ids")
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
        self._done(work)
        if parts is not None:                      # gloo path: `parts` holds the target device
            out = out.to(parts)
        return [out[:, i, :].reshape(-1).contiguous() for i in range(out.shape[1])]

    # ---- row-sharded tables (BASELINE config 4) ---------------------------------------------
    def all_gather_rows(self, local, out):
        """out[world·n][...] = rank-major concatenation of every rank's `local` [n][...]
        (identical shape on all ranks).  RCCL all-gather; gloo stages through the host."""
        if not self.live:
            out.copy_(local)
            return out
        self._enter("all_gather")
        if self.backend == "nccl":
            direct = self._rccl_direct()
            if direct is not None and local.dtype in direct.DTYPES and out.is_contiguous():
                # on the CURRENT stream: ordered behind the kernels already enqueued and before the next ones
                direct.all_gather(local.contiguous(), out)
                self._done()
                return out
            self._done(dist.all_gather_into_tensor(out, local.contiguous(), async_op=True))
            return out
        host = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype)
        dist.all_gather([host[r] for r in range(self.world)], local.detach().cpu().contiguous())
        out.copy_(host.reshape(out.shape))
        self._done()
        return out

    def all_gather_rows_start(self, local, out):
        """all_gather_rows without the wait: -> token for all_gather_rows_finish.  RCCL: the collective runs on the
        process group's stream behind what the current stream holds NOW; kernels enqueued after this call overlap it
        (the column-sliced hop of neurec_amd/sharded.py).  gloo: done when this returns."""
        if not self.live:
            out.copy_(local)
            return None
        self._enter("all_gather_async")
        if self.backend == "nccl":
            return dist.all_gather_into_tensor(out, local.contiguous(), async_op=True)
        host = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype)
        dist.all_gather([host[r] for r in range(self.world)], local.detach().cpu().contiguous())
        out.copy_(host.reshape(out.shape))
        return None

    def all_gather_rows_finish(self, token):
        self._done(token)

    def bcast_rows_start(self, buf, src):
        """Begin broadcasting rank `src`'s `buf` into every other rank's `buf` (one chunk of a chunked all-gather:
        neurec_amd/sharded.py).  RCCL: asynchronous on the collective stream, ordered behind the work already
        enqueued on the current stream (so a receive slot is not overwritten under the kernel that still reads it);
        returns a token for bcast_rows_finish.  gloo: host-staged, done when this returns."""
        if not self.live:
            return None
        self._enter("broadcast")
        if self.backend == "nccl":
            return dist.broadcast(buf, src=src, async_op=True)
        host = buf.detach().cpu().contiguous() if self.rank == src else torch.empty(tuple(buf.shape), dtype=buf.dtype)
        dist.broadcast(host, src=src)
        if self.rank != src:
            buf.copy_(host)
        return None

    def bcast_rows_finish(self, token):
        """the current stream waits for the chunk (no host synchronisation)"""
        self._done(token)

    def all_to_all_rows(self, send, send_counts, recv_counts=None):
        """Variable all-to-all of rows: `send` [n][...] is ordered by destination rank,
        `send_counts[r]` rows go to rank r.  Returns (recv [m][...], recv_counts) ordered by
        source rank.  With `recv_counts` given (e.g. from RowRouter.plan_epoch) nothing is
        exchanged or copied to the host besides the rows themselves; without, the counts are
        exchanged first (one host synchronisation).  RCCL all_to_all_single over xGMI; gloo:
        host-staged point-to-point."""
        send_counts = [int(c) for c in send_counts]
        if not self.live:
            return send, send_counts
        self._enter("all_to_all")
        tail = tuple(send.shape[1:])
        if self.backend == "nccl":
            if recv_counts is None:
                sc = torch.tensor(send_counts, dtype=torch.int64, device=send.device)
                rc = torch.empty_like(sc)
                self._done(dist.all_to_all_single(rc, sc, async_op=True))
                recv_counts = [int(c) for c in rc.cpu()]
            recv_counts = [int(c) for c in recv_counts]
            recv = torch.empty((sum(recv_counts),) + tail, dtype=send.dtype, device=send.device)
            direct = self._rccl_direct()
            if direct is not None and send.dtype in direct.DTYPES:
                # on the current stream (see all_gather_rows): the row-sharded step's three exchanges are blocking ones
                direct.all_to_all_rows(send.contiguous(), send_counts, recv, recv_counts)
                self._done()
                return recv, recv_counts
            self._done(dist.all_to_all_single(recv, send.contiguous(), recv_counts, send_counts, async_op=True))
            return recv, recv_counts
        if recv_counts is None:
            mat = torch.empty((self.world, self.world), dtype=torch.int64)
            dist.all_gather([mat[r] for r in range(self.world)], torch.tensor(send_counts, dtype=torch.int64))
            recv_counts = [int(mat[r, self.rank]) for r in range(self.world)]
        recv_counts = [int(c) for c in recv_counts]
        host_send = send.detach().cpu().contiguous()
        host_recv = torch.empty((sum(recv_counts),) + tail, dtype=send.dtype)
        reqs, so, ro = [], 0, 0
        for r in range(self.world):
            s_chunk = host_send[so:so + send_counts[r]]
            r_chunk = host_recv[ro:ro + recv_counts[r]]
            if r == self.rank:
                r_chunk.copy_(s_chunk)
            else:
                if send_counts[r]:
                    reqs.append(dist.isend(s_chunk.contiguous(), r))
                if recv_counts[r]:
                    reqs.append(dist.irecv(r_chunk, r))
            so += send_counts[r]
            ro += recv_counts[r]
        for q in reqs:
            q.wait()
        self._done()
        return host_recv.to(send.device), recv_counts

    def all_to_all_equal_start(self, send, recv):
        """Equal-split all-to-all of row blocks: send [world·n][...] holds n rows for every rank (destination-major),
        recv [world·n][...] receives n rows from every rank (source-major).  -> token for all_to_all_equal_finish.
        RCCL: asynchronous on the process group's stream (every link carries one block at the same time); gloo:
        host-staged point-to-point, done when this returns."""
        if not self.live:
            recv.copy_(send)
            return None
        self._enter("all_to_all_equal")
        if self.backend == "nccl":
            return dist.all_to_all_single(recv, send, async_op=True)
        n = send.shape[0] // self.world
        got, _ = self.all_to_all_rows(send, [n] * self.world, [n] * self.world)
        recv.copy_(got.reshape(recv.shape))
        return None

    def all_to_all_equal_finish(self, token):
        self._done(token)

    def broadcast_(self, t, src=0):
        """In-place broadcast from rank `src` (used to re-align replicas whose scatter atomics
        summed in different orders)."""
        if self.live:
            self._enter("broadcast")
            if self.backend != "nccl" and t.is_cuda:
                host = t.detach().cpu()
                dist.broadcast(host, src=src)
                t.copy_(host)
                self._done()
            else:
                self._done(dist.broadcast(t, src=src, async_op=True))
        return t

    def max_float(self, x):
        if not self.live:
            return float(x)
        dev = "cuda" if (self.backend == "nccl" and torch.cuda.is_available()) else "cpu"
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def shutdown(self):
        if self._direct is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self._direct.close()
            self._direct = None
        if self.live and dist.is_initialized():
            dist.destroy_process_group()


def init_from_env(backend=None, force=None):
    """Read RANK / WORLD_SIZE / LOCAL_RANK / LOCAL_WORLD_SIZE / MASTER_* (set by torch.distributed.run, or by bench.py
    when it starts its own ranks).  Backend: NEUREC_DIST_BACKEND, else "nccl" (= RCCL) when every rank OF THIS NODE has a GPU of its own, else
    "gloo" — RCCL cannot put two ranks on one device, so on a box with fewer GPUs than ranks the ranks share devices
    round-robin and exchange through the host (the line then says dist_backend gloo, rccl_ranks 0).
    force (or NEUREC_DIST_FORCE_GROUP=1): create the process group even at world size 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if force is None:
        force = os.environ.get("NEUREC_DIST_FORCE_GROUP", "") == "1"
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    # ranks on THIS node (torch.distributed.run sets LOCAL_WORLD_SIZE; a 2 x 8 launch has world 16, 8 per node)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if backend is None:
        backend = os.environ.get("NEUREC_DIST_BACKEND") or ("nccl" if n_dev >= local_world else "gloo")
    if n_dev:
        torch.cuda.set_device(local_rank % n_dev)
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not dist.is_initialized():
            if backend == "gloo":
                # gloo's C++ side announces "[Gloo] Rank r is connected to n peer ranks" on STDOUT when the mesh
                # connects; rank 0's stdout is the bench's one JSON line, so that chatter goes to stderr
                import sys
                sys.stdout.flush()
                saved = os.dup(1)
                os.dup2(2, 1)
                try:
                    dist.init_process_group(backend=backend, rank=rank, world_size=world, **_rendezvous())
                    dist.barrier()
                finally:
                    sys.stdout.flush()
                    os.dup2(saved, 1)
                    os.close(saved)
            else:
                dist.init_process_group(backend=backend, rank=rank, world_size=world, **_rendezvous())
    return Comm(rank, world, local_rank, backend, force=force)


def _rendezvous():
    """NEUREC_DIST_INIT_FILE (bench.py's self-started ranks): a file store instead of MASTER_ADDR / MASTER_PORT — no
    port to pick before the ranks exist, hence none to lose to another process in between."""
    path = os.environ.get("NEUREC_DIST_INIT_FILE")
    return {"init_method": "file://" + path} if path else {}


class BipartitePartition:
    """Row partition of the N = U + I nodes of a bipartite user-item graph over `world` ranks.

    A contiguous block partition of [users; items] would hand the LAST rank every item row — half of
    all non-zeros of the adjacency at config 4 (10^7 users, 10^6 items) — so every rank owns a slice
    of the users AND a slice of the items: rank r holds users [r·bu, (r+1)·bu) then items
    [r·bi, (r+1)·bi), padded to b = bu + bi rows.  An all-gather of the padded blocks lays the table
    out rank-major; `position(node)` is a node's row in that gathered layout (the local CSR blocks
    carry these positions as column indices; their STORAGE order stays ascending node id, which is
    the order the row sums run in)."""

    def __init__(self, n_users, n_items, world):
        self.U, self.I, self.world = int(n_users), int(n_items), int(world)
        self.bu = (self.U + self.world - 1) // self.world
        self.bi = (self.I + self.world - 1) // self.world
        self.b = self.bu + self.bi
        self.n_pad = self.b * self.world

    def users_of(self, rank):
        return min(rank * self.bu, self.U), min((rank + 1) * self.bu, self.U)

    def items_of(self, rank):
        return min(rank * self.bi, self.I), min((rank + 1) * self.bi, self.I)

    def owner_local(self, nodes):
        """global node ids (users < U <= U + item) -> (owner rank, row in the owner's block); numpy
        arrays or torch tensors."""
        is_item = nodes >= self.U
        idx = nodes - self.U * is_item
        if isinstance(nodes, torch.Tensor):
            per = torch.where(is_item, torch.full_like(nodes, self.bi), torch.full_like(nodes, self.bu))
            owner = torch.div(idx, per, rounding_mode="floor")
        else:
            import numpy as np
            per = np.where(is_item, self.bi, self.bu)
            owner = idx // per
        local = idx - owner * per + self.bu * is_item
        return owner, local

    def position(self, nodes):
        owner, local = self.owner_local(nodes)
        return owner * self.b + local

    def gathered_index(self, device=None):
        """index tensors (users, items) into the gathered [n_pad][d] layout, in natural id order"""
        u = torch.arange(self.U, dtype=torch.int64, device=device)
        i = torch.arange(self.I, dtype=torch.int64, device=device) + self.U
        return self.position(u), self.position(i)


def block_size(n, world):
    """Rows per rank of the padded block partition used for row-sharded tables: rank r owns
    global rows [r·b, min((r+1)·b, n)), so an all-gather of the padded blocks is the table."""
    return (n + world - 1) // world


def partition(n, rank, world):
    """Contiguous block partition of range(n): rank r owns [lo, hi)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def shard_users(user_ids, rank, world):
    lo, hi = partition(len(user_ids), rank, world)
    return user_ids[lo:hi]


_process_comm = None


def get_comm():
    """The process's communicator: created from the launcher's environment on first use (world size 1 without one:
    an inactive Comm, every collective a no-op).  The drop-in entry point (`python -m torch.distributed.run
    --nproc-per-node N -m neurec_amd.main ...`) and the plugins ask here whether they are one rank of several."""
    global _process_comm
    if _process_comm is None:
        _process_comm = init_from_env()
    return _process_comm


def gather_rows_by_user(comm, users_local, rows_local, device):
    """Every rank's (user ids int32 [n_r], per-user rows float32 [n_r][w]) -> (users [n], rows [n][w]) on every rank,
    rank-major.  The sharded drop-in evaluation gathers the per-user METRIC ROWS (M·K floats per user) so that the
    mean over users is the reference's float32 np.mean over the same rows in the same order
    (uni_evaluator.py:150-151), whatever the number of ranks."""
    import numpy as np
    n = torch.tensor([int(users_local.numel())], dtype=torch.int64, device=device)
    counts = torch.zeros(comm.world, dtype=torch.int64, device=device)
    comm.all_gather_rows(n, counts)
    counts = [int(c) for c in counts.cpu()]
    cap, w = max(counts + [1]), int(rows_local.shape[1])
    pad_u = torch.zeros(cap, dtype=torch.int32, device=device)
    pad_r = torch.zeros((cap, w), dtype=torch.float32, device=device)
    pad_u[:users_local.numel()] = users_local
    pad_r[:rows_local.shape[0]] = rows_local
    all_u = torch.empty(comm.world * cap, dtype=torch.int32, device=device)
    all_r = torch.empty((comm.world * cap, w), dtype=torch.float32, device=device)
    comm.all_gather_rows(pad_u, all_u)
    comm.all_gather_rows(pad_r, all_r)
    keep = np.concatenate([np.arange(r * cap, r * cap + c) for r, c in enumerate(counts)]) if sum(counts) else np.zeros(0, np.int64)
    keep = torch.from_numpy(keep.astype(np.int64)).to(device)
    return all_u[keep], all_r[keep]
