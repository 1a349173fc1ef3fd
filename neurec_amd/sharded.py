"""Row-sharded LightGCN / BPR-MF over the GPUs of one node (BASELINE config 4: tables that are not
meant to be replicated — U = 10⁷, I = 10⁶, d = 128 is 5.6 GB per [N][d] buffer and the step keeps
a dozen of them).

Partition (`parallel.BipartitePartition`): every rank owns a slice of the users AND a slice of the items
— rank r holds users [r·bu, (r+1)·bu) then items [r·bi, (r+1)·bi), padded to b = bu + bi rows — of the
embedding table E0, of its Adam moments and of every layer buffer, plus the CSR rows of Â (and of Âᵀ
when Â is not symmetric) for those nodes.  (A contiguous cut of [users; items] would give the last
rank every item row: 52 % of the non-zeros at config 4.)  An all-gather of the padded blocks lays a
table out rank-major; the local CSR blocks carry positions in that layout as column indices, in
ascending-node-id storage order (the order the row sums run in).  A rank can be built from its own
rows alone (`local_rows=`): no rank needs the whole graph.

One LightGCN step (LightGCN.py:132-166,178) then has exactly these exchange points:
  * per propagation hop (L forward, L backward): ONE all-gather of the [b][d] blocks into the
    [world·b][d] operand of the local SpMM  (RCCL all-gather over xGMI; N·d·4 bytes per hop);
  * BPR head: the 3·B rows a rank's triplets touch live on their owners — ids go out with an
    all-to-all, owners answer with the rows (Esum and E0 side by side), the head runs locally on
    the compact [3B][d] block, and the 3·B gradient rows return by the reverse all-to-all;
  * the owners add the rows they receive IN THE ORDER OF THE GLOBAL BATCH (class, rank, position
    — what TF's unsorted_segment_sum over the concatenated batch does, and what the single-GPU
    head does): keys (row, global position) are sorted (nrhip_sort_u64) and every row's run is
    summed in key order (nrhip_rows_sum_sorted).  No atomics: the sharded step is bit-identical to
    the single-process step on the concatenated batch (tests assert array_equal);
  * Adam is owner-local (dense TF-Adam on the rank's block): no exchange.
Routing needs the per-destination row counts on the host (all_to_all_single takes Python lists).
`plan_epoch` computes them for every batch of an epoch stream in one pass and ONE device→host
copy — and, with them, the routing itself of every batch (requests in owner order, their inverse,
what each owner is asked for, the owner-side sorted keys: csrc/route.hip, one all-to-all of the
epoch's ids) — so a planned step runs without host synchronisation and without a single routing
launch; a step on a batch that was not planned routes and counts on the spot (one sync).
The per-hop all-gather is PIPELINED with the local SpMM without changing a single rounding (`ChunkedHop`,
r04).  Splitting a row into (local columns) + (remote columns) would re-associate its sum; but with this
partition a user row's columns are items and an item row's are users, so ascending column id IS ascending
owner rank: the operand is received in rank order, one chunk per rank (a broadcast each, two receive slots),
and launch r adds the non-zeros owned by rank r to the row accumulators carried from launch r-1 — the same
ascending chain, cut where the owners change — while chunk r+1 is on the links.  (`norm`'s self loop is the
first term of a user row and the last of an item row, and always local.)  It also drops the [world·b][d]
gathered operand: a hop holds two [b][d] receive slots.  NEUREC_ROWSHARD_PIPELINE=0 keeps the one-all-gather
form (bit-identical; tests/test_sharded_gpu.py runs both).
All arithmetic is the same HIP kernels as the replicated engine; the collectives are
torch.distributed plumbing (`parallel.Comm`).
"""
import numpy as np
import torch

import os

from . import engine as E
from . import parallel

# NRHIP_ATOMIC_SCATTER=1 (A/B knob, read once per process by the library): the one-wave-per-triplet heads
# accumulate with fp32 atomics instead of storing ordered sums, so their compact output rows must start from zero
_ATOMIC_HEADS = os.environ.get("NRHIP_ATOMIC_SCATTER", "") == "1"


def _compact_plan(engine, B):
    """The batch plan of the compact head block (users = rows 0..B-1, positives B..2B-1, negatives 2B..3B-1: every
    occurrence has its own row) depends on B alone: built once per batch length instead of once per step."""
    cache = engine.__dict__.setdefault("_compact_plans", {})
    if B not in cache:
        dev = torch.device("cuda", torch.cuda.current_device())
        ar = torch.arange(B, dtype=torch.int32, device=dev)
        cache[B] = E.bpr_plan(ar, ar, (ar + B).contiguous(), B, B).clone()
    return cache[B]


def _compact_neg(engine, B):
    """ids B .. 2B-1: the negatives' rows of the compact head block, per batch length"""
    cache = engine.__dict__.setdefault("_compact_negs", {})
    if B not in cache:
        cache[B] = torch.arange(B, 2 * B, dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
    return cache[B]


class RowRouter:
    """Requests for table rows by global node id -> owners and back, for block-partitioned tables.  The index
    bookkeeping of a step is two native calls (csrc/route.hip): `request` (requests in owner order + the inverse
    permutation) and `ordered_keys` (the owner-side keys of the global batch order)."""

    CODE = 1 << 24                       # occurrence code = class * CODE + position in the rank's batch

    def __init__(self, comm, part, max_batch):
        self.comm, self.part = comm, part
        self.world, self.rank = comm.world, comm.rank
        self._epoch = None               # per-epoch routing counts (plan_epoch)
        self._tables = None              # ... and the routing itself, every batch (plan_epoch -> _build_tables)
        dev = torch.device("cuda", torch.cuda.current_device())
        n = 3 * int(max_batch)
        self._keys = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        self._packed = torch.empty((max(n, 1), 2), dtype=torch.int32, device=dev)
        self._order = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        self._inv = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        self._counts = torch.zeros(self.world, dtype=torch.int32, device=dev)

    # ---- per-epoch routing tables: no host sync inside the steps
    def plan_epoch(self, classes, offsets, batch, tables=True):
        """classes: list of int32 device tensors (one per id class: users, pos, neg) holding this
        rank's slice of the epoch stream; offsets: global-id offset per class.  One pass + one
        count exchange + one device->host copy for the whole epoch."""
        n = classes[0].numel()
        nb = (n + batch - 1) // batch
        dev = classes[0].device
        which = torch.arange(n, device=dev) // batch
        cnt = torch.zeros(nb * self.world, dtype=torch.int64, device=dev)
        for ids, off in zip(classes, offsets):
            owner, _ = self.part.owner_local(ids.long() + off)
            cnt += torch.bincount(which * self.world + owner, minlength=nb * self.world)
        send = cnt.view(nb, self.world)
        sizes = torch.full((nb,), batch, dtype=torch.int64, device=dev)
        if nb:
            sizes[-1] = n - (nb - 1) * batch
        recv, all_sizes = self._exchange_counts(send, sizes)
        # what the owner-side key kernel reads, for every batch of the epoch, resident on the device
        zero = torch.zeros((nb, 1), dtype=torch.int64, device=dev)
        recv_prefix = torch.cat([zero, torch.cumsum(recv, 1)], 1).to(torch.int32).contiguous()
        size_off = torch.cat([zero, torch.cumsum(all_sizes, 1)[:, :-1]], 1).to(torch.int32).contiguous()
        self._epoch = (int(batch), send.cpu().tolist(), recv.cpu().tolist(), all_sizes.cpu().tolist(),
                       recv_prefix, size_off)
        self._tables = None
        if tables and nb and len(classes) == 3 and tuple(offsets[1:]) == (offsets[1], offsets[1]) and offsets[0] == 0:
            self._build_tables(classes, int(offsets[1]), int(batch), sizes, send, recv, all_sizes, recv_prefix, size_off)

    def _build_tables(self, classes, n_users, batch, sizes, send, recv, all_sizes, recv_prefix, size_off):
        """The routing of EVERY batch of the planned stream, built by a handful of launches that work on all batches
        at once (csrc/route.hip: nrhip_route_epoch, nrhip_route_epoch_owner_keys; one all-to-all of the whole
        epoch's (row, code) pairs): afterwards a planned step issues NO routing launch — `planned_route(k)` hands out
        slices.  Per batch the tables equal what `request` + `ordered_keys` compute (tests/test_sharded_gpu.py)."""
        users, pos, neg = classes
        dev, W, B = users.device, self.world, batch
        n, nb = users.numel(), send.shape[0]
        max_asked = int(recv.sum(1).max())
        if self.comm.live:                                 # the decision below must be the same on every rank
            max_asked = int(self.comm.max_float(float(max_asked)))
        if 3 * B > 16384 or max_asked > 16384:            # one sorting workgroup per batch: routed per step instead
            return
        i32 = lambda *shape: torch.empty(*shape, dtype=torch.int32, device=dev)
        seg_off = (torch.arange(nb, device=dev, dtype=torch.int64) * (3 * B)).contiguous()
        seg_len = (3 * sizes).to(torch.int32).contiguous()
        keys = torch.empty(3 * nb * B, dtype=torch.int64, device=dev)
        packed, order, inv = i32((3 * nb * B, 2)), i32(3 * nb * B), i32(3 * nb * B)       # (counts: plan_epoch has them)
        E.call("nrhip_route_epoch", E._ptr(users, torch.int32), E._ptr(pos, torch.int32), E._ptr(neg, torch.int32), n, B,
               n_users, self.part.bu, self.part.bi, self.CODE, W, E._ptr(keys), E._ptr(packed), E._ptr(order),
               E._ptr(inv), None, E._ptr(seg_off), E._ptr(seg_len), E._stream())
        mine = packed[:3 * n]                                 # only the LAST batch can be short: the valid slots are a prefix
        if W == 1:
            asked = mine
        else:
            # to the owners: destination-major, batches in order inside a destination
            q = torch.arange(3 * n, device=dev, dtype=torch.int64)
            k, o = q // (3 * B), keys[:3 * n] >> 32
            soff = torch.cumsum(send, 1) - send               # [nb][W] first routed index of owner o inside batch k
            kpre = torch.cumsum(send, 0) - send               # requests to o in the batches before k
            tot_s, tot_r = send.sum(0), recv.sum(0)
            dbase = torch.cumsum(tot_s, 0) - tot_s
            dst = dbase[o] + kpre[k, o] + (q - 3 * B * k) - soff[k, o]
            out = torch.empty_like(mine)
            out[dst] = mine
            got, _ = self.comm.all_to_all_rows(out, tot_s.cpu().tolist(), tot_r.cpu().tolist())
            # from the sources: source-major -> [batch][source]
            rpre = torch.cumsum(recv, 0) - recv
            sbase = torch.cumsum(tot_r, 0) - tot_r
            spre = torch.cumsum(recv, 1) - recv
            per_batch = recv.sum(1)
            aoff = torch.cumsum(per_batch, 0) - per_batch
            src_start = (sbase.view(1, W) + rpre).reshape(-1)
            dst_start = (aoff.view(nb, 1) + spre).reshape(-1)
            lens = recv.reshape(-1)
            total = int(lens.sum())
            idx = torch.repeat_interleave(src_start - dst_start, lens) + torch.arange(total, device=dev)
            asked = got[idx]
        per_batch = recv.sum(1)
        asked_off = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
        asked_off[1:] = torch.cumsum(per_batch, 0)
        asked_len = per_batch.to(torch.int32).contiguous()
        total = int(asked.shape[0])
        batch_of = torch.repeat_interleave(torch.arange(nb, device=dev, dtype=torch.int32), per_batch).contiguous()
        glen = all_sizes.sum(1).to(torch.int32).contiguous()
        stride = 3 * int(glen.max())
        rows, codes = asked[:, 0].contiguous(), asked[:, 1].contiguous()
        okeys = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
        iop = i32(max(nb * stride, 1))
        E.call("nrhip_route_epoch_owner_keys", E._ptr(rows), E._ptr(codes), E._ptr(batch_of), total,
               E._ptr(asked_off), E._ptr(asked_len), nb, max_asked, E._ptr(recv_prefix), E._ptr(size_off),
               E._ptr(glen), W, self.CODE, stride, E._ptr(okeys), E._ptr(iop), E._stream())
        self._tables = (B, order, inv, rows, codes, asked_off.cpu().tolist(), okeys, iop, stride)

    def planned_route(self, k, batch_len):
        """The Route of planned batch k (slices of the epoch tables, no launch) with its owner-side keys attached,
        or None when that batch has no tables."""
        t, planned = self._tables, self.epoch_counts(k, batch_len)
        if t is None or planned is None:
            return None
        B, order, inv, rows, codes, aoff, okeys, iop, stride = t
        send_counts, recv_counts, sizes, recv_prefix, size_off = planned
        s0, n3, a0, a1, G = 3 * B * k, 3 * batch_len, aoff[k], aoff[k + 1], int(sum(sizes))
        rt = Route(order[s0:s0 + n3], inv[s0:s0 + n3], rows[a0:a1], codes[a0:a1], send_counts, recv_counts, G,
                   recv_prefix, size_off)
        rt.keys, rt.index_of_pos = okeys[a0:a1], iop[k * stride:k * stride + 3 * G]
        return rt

    def _exchange_counts(self, send, sizes):
        """send [nb][world] (what I send to r in batch k) -> recv [nb][world] (what r sends me);
        sizes [nb] -> [nb][world] batch length of every rank."""
        if not self.comm.live:
            return send.clone(), sizes.view(-1, 1).clone()
        import torch.distributed as dist
        nb = send.shape[0]
        host = self.comm.backend != "nccl"
        s = send.t().contiguous()                                   # [world][nb]: row r goes to rank r
        r = torch.empty_like(s)
        g = torch.empty((self.world, nb), dtype=torch.int64, device=s.device)
        if host:
            s, r, g, sizes = s.cpu(), r.cpu(), g.cpu(), sizes.cpu()
            mat = torch.empty((self.world,) + tuple(s.shape), dtype=torch.int64)     # gloo: no all_to_all
            dist.all_gather(list(mat.unbind(0)), s)
            r = mat[:, self.rank, :].contiguous()
            dist.all_gather(list(g.unbind(0)), sizes.contiguous())
            r, g = r.to(send.device), g.to(send.device)
        else:
            dist.all_to_all_single(r, s)
            dist.all_gather_into_tensor(g, sizes.contiguous())
        return r.t().contiguous(), g.t().contiguous()

    def epoch_counts(self, k, batch_len):
        ep = self._epoch
        if ep is None or k is None or k >= len(ep[1]) or ep[3][k][self.rank] != batch_len:
            return None
        return ep[1][k], ep[2][k], ep[3][k], ep[4][k], ep[5][k]

    # ---- one batch
    def request(self, users, pos, neg, n_users, planned=None):
        """ids of one batch -> a Route: the requests in owner order went out, `asked` / `asked_code` are what this
        rank was asked for.  planned = epoch_counts(k, B) or None (then counted now: one host sync)."""
        B = users.numel()
        n = 3 * B
        counts = None if planned is not None else self._counts
        E.route_batch(users, pos, neg, n_users, self.part.bu, self.part.bi, self.CODE, self.world,
                      self._keys[:n], self._packed[:n], self._order[:n], self._inv[:n], counts)
        if planned is None:
            send_counts = self._counts.cpu().tolist()
            dev = users.device
            sizes = torch.tensor([B], dtype=torch.int64, device=dev)
            recv, all_sizes = self._exchange_counts(torch.tensor([send_counts], dtype=torch.int64, device=dev), sizes)
            recv_counts, sizes = recv[0].cpu().tolist(), all_sizes[0].cpu().tolist()
            recv_prefix = torch.tensor(np.concatenate([[0], np.cumsum(recv_counts)]), dtype=torch.int32, device=dev)
            size_off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)[:-1]]), dtype=torch.int32, device=dev)
        else:
            send_counts, recv_counts, sizes, recv_prefix, size_off = planned
        asked, _ = self.comm.all_to_all_rows(self._packed[:n], send_counts, recv_counts)
        return Route(self._order[:n], self._inv[:n], asked[:, 0].contiguous(), asked[:, 1].contiguous(),
                     send_counts, recv_counts, int(sum(sizes)), recv_prefix, size_off)

    def ordered_keys(self, route):
        """Sorted keys (local row << 32 | global position) of the rows received from the other ranks and the
        index_of_pos table: global position = class * G + (offset of the source rank) + b, G = the global batch
        length — the occurrence order of the single-process head on the concatenated batch."""
        if route.keys is not None:                                  # planned with the epoch
            return route.keys, route.index_of_pos
        dev = route.asked.device
        keys = torch.empty(route.asked.numel(), dtype=torch.int64, device=dev)
        index_of_pos = torch.empty(max(3 * route.G, 1), dtype=torch.int32, device=dev)
        E.route_owner_keys(route.asked, route.asked_code, route.recv_prefix, route.size_off, self.world, route.G,
                           self.CODE, keys, index_of_pos)
        return keys, index_of_pos


class Route:
    """one batch's routing: order[i] = the request (class * B + b) that travels at position i, inv = its inverse;
    asked / asked_code = the local rows (and occurrence codes) this rank was asked for, by source rank"""
    __slots__ = ("order", "inv", "asked", "asked_code", "send_counts", "recv_counts", "G", "recv_prefix", "size_off",
                 "keys", "index_of_pos")

    def __init__(self, *a):
        self.keys = self.index_of_pos = None
        for k, v in zip(self.__slots__, a):
            setattr(self, k, v)


class ChunkedHop:
    """A rank's row block of Â re-cut for the chunked hop: CSR r holds, for every VIRTUAL row, the non-zeros whose
    column is owned by rank r (in storage order), columns addressed as (own block row | b + row of the received
    chunk).  Virtual rows: a real row of more than `seg_len` non-zeros is cut by position into seg_len-segments, each
    summed as a row of its own and added in segment order at the end — the association of the one-launch kernels
    (seg_len = SpmmCSR.exact_row_nnz(d): 256 for the work-item kernel, 64 where the lane-group schedule runs)."""

    def __init__(self, part, rank, indptr, cols_global, vals, seg_len, device):
        t = lambda a, dt: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(device, dt)
        ip, cg, va = t(indptr, torch.int64), t(cols_global, torch.int64), t(vals, torch.float32)
        self.world, self.rank, self.b = part.world, int(rank), part.b
        b, W, nnz = part.b, part.world, int(cg.numel())
        assert ip.numel() == b + 1
        lens = ip[1:] - ip[:-1]
        n_v = torch.clamp((lens + seg_len - 1) // seg_len, min=1)                # virtual rows of every real row
        first = torch.zeros(b + 1, dtype=torch.int64, device=device)
        first[1:] = torch.cumsum(n_v, 0)
        self.n_virtual = int(first[-1])
        self.first_vrow = first.to(torch.int32).contiguous()
        row_of = torch.repeat_interleave(torch.arange(b, device=device), lens, output_size=nnz)
        pos = torch.arange(nnz, device=device) - ip[row_of]
        vrow = first[row_of] + pos // seg_len
        owner, local = part.owner_local(cg)
        # launch of every non-zero = the owner of its column; a column on the ROW'S OWN side of the graph (`norm`'s self
        # loop) is local and sits first in a user row, last in an item row
        user_row = row_of < part.bu
        same_side = (cg < part.U) == user_row
        if bool((same_side & (owner != self.rank)).any()):
            raise NotImplementedError("chunked hop: a same-side edge to another rank (not a bipartite graph + self loops)")
        launch = torch.where(same_side, torch.where(user_row, 0, W - 1), owner)
        inside = row_of[1:] == row_of[:-1]
        if bool((inside & (launch[1:] < launch[:-1])).any()):
            raise NotImplementedError("chunked hop: a row's owners are not ascending in storage order")
        col = torch.where(owner == self.rank, local, b + local).to(torch.int32)
        self.chunks = []
        for r in range(W):
            sel = launch == r
            cnt = torch.bincount(vrow[sel], minlength=self.n_virtual)
            ipr = torch.zeros(self.n_virtual + 1, dtype=torch.int64, device=device)
            ipr[1:] = torch.cumsum(cnt, 0)
            self.chunks.append(E.SpmmCSR(ipr, col[sel], va[sel], n_cols=2 * b))
            assert self.chunks[-1].n_split_rows == 0
        self.real_of_vrow = torch.repeat_interleave(torch.arange(b, device=device), n_v,
                                                    output_size=self.n_virtual).to(torch.int32)
        self._Yv = {}
        self._slots = {}
        self._vmask = torch.zeros(self.n_virtual, dtype=torch.uint8, device=device)

    def buffers(self, d, device):
        if d not in self._Yv:
            self._Yv[d] = torch.empty((self.n_virtual, d), dtype=torch.float32, device=device)
            self._slots[d] = [torch.empty((self.b, d), dtype=torch.float32, device=device) for _ in range(2)]
        return self._Yv[d], self._slots[d]

    def matmul(self, comm, local, out=None, addend=None, sum_in=None, sum_out=None, y_row_wanted=None):
        """out = Â_block · (all ranks' `local` blocks) (+ addend); sum_out = sum_in + out — nrhip_spmm_csr's contract,
        with the other ranks' blocks arriving chunk by chunk under the launches."""
        d, W = local.shape[1], self.world
        Yv, slots = self.buffers(d, local.device)
        vmask = None
        if y_row_wanted is not None:
            E.gather_u8(y_row_wanted, self.real_of_vrow, self._vmask)
            vmask = self._vmask
        bufs = [local if r == self.rank else slots[r & 1] for r in range(W)]
        tokens = [None] * W
        for r in range(min(2, W)):
            tokens[r] = comm.bcast_rows_start(bufs[r], r)
        for r in range(W):
            comm.bcast_rows_finish(tokens[r])
            ch = self.chunks[r]
            E.call("nrhip_spmm_csr_carry", ch.plan, E._ptr(ch.indptr), E._ptr(ch.indices), E._ptr(ch.vals),
                   E._ptr(local, torch.float32), E._ptr(bufs[r], torch.float32), self.b, d, E._ptr(Yv),
                   1 if r > 0 else 0, E._ptr(vmask, allow_none=True), E._stream())
            if r + 2 < W:                                  # its slot is free once launch r has been enqueued
                tokens[r + 2] = comm.bcast_rows_start(bufs[r + 2], r + 2)
        E.call("nrhip_spmm_chunks_finish", E._ptr(self.first_vrow), self.b, E._ptr(Yv), d,
               E._ptr(out, torch.float32, allow_none=True), E._ptr(addend, allow_none=True),
               E._ptr(sum_in, allow_none=True), E._ptr(sum_out, allow_none=True),
               E._ptr(y_row_wanted, allow_none=True), E._stream())
        return out


class ShardedLightGCN:
    def __init__(self, comm, adj_csr, n_users, n_items, embed, n_layers, lr, reg, max_batch,
                 symmetric=None, local_rows=None, local_rows_t=None, pipeline=None):
        """adj_csr: the full [N][N] scipy adjacency (each rank slices its block), or None with
        local_rows = (indptr, indices, vals) of this rank's row block only (global column ids;
        local_rows_t for Âᵀ when Â is not symmetric).  embed: the full [N][d] table or just this
        rank's [n_loc][d] rows.  pipeline: the chunked hop (None: on when there is more than one rank,
        unless NEUREC_ROWSHARD_PIPELINE=0)."""
        dev = E.require_gpu()
        self.comm, self.rank, self.world = comm, comm.rank, comm.world
        if pipeline is None:
            pipeline = self.world > 1 and os.environ.get("NEUREC_ROWSHARD_PIPELINE", "1") != "0"
        self.pipeline = bool(pipeline)
        self._d_hint = int(embed.shape[1])
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.N = self.n_users + self.n_items
        self.L, self.reg, self.max_batch = int(n_layers), float(reg), int(max_batch)
        self.part = parallel.BipartitePartition(self.n_users, self.n_items, self.world)
        self.b, self.Npad = self.part.b, self.part.n_pad
        self.ulo, self.uhi = self.part.users_of(self.rank)
        self.ilo, self.ihi = self.part.items_of(self.rank)
        self.nu, self.ni = self.uhi - self.ulo, self.ihi - self.ilo
        if local_rows is not None:
            self.A = self._from_block(*local_rows)
            self.At = self.A if local_rows_t is None else self._from_block(*local_rows_t)
        else:
            a = adj_csr.tocsr().astype(np.float32)
            a.sort_indices()
            if symmetric is None:
                symmetric = (a != a.T).nnz == 0
            self.A = self._local_rows(a)
            self.At = self.A if symmetric else self._local_rows(a.T.tocsr())
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        if isinstance(embed, torch.Tensor):             # this rank's rows (users then items) or the full table
            self.d = embed.shape[1]
            src_u, src_i = ((embed[:self.nu], embed[self.nu:self.nu + self.ni])
                            if embed.shape[0] == self.nu + self.ni and embed.shape[0] != self.N else
                            (embed[self.ulo:self.uhi], embed[self.n_users + self.ilo:self.n_users + self.ihi]))
            src_u, src_i = src_u.to(dev), src_i.to(dev)
        else:
            embed = np.asarray(embed, dtype=np.float32)
            self.d = embed.shape[1]
            if embed.shape[0] == self.nu + self.ni and embed.shape[0] != self.N:
                hu, hi_ = embed[:self.nu], embed[self.nu:]
            else:
                hu, hi_ = embed[self.ulo:self.uhi], embed[self.n_users + self.ilo:self.n_users + self.ihi]
            src_u = torch.from_numpy(np.ascontiguousarray(hu)).to(dev)
            src_i = torch.from_numpy(np.ascontiguousarray(hi_)).to(dev)
        self.E0 = z(self.b, self.d)
        self.E0[:self.nu] = src_u
        self.E0[self.part.bu:self.part.bu + self.ni] = src_i
        self.m, self.v = z(self.b, self.d), z(self.b, self.d)
        # gathered operand of the one-all-gather hop; the chunked hop holds two [b][d] receive slots instead
        self._X = None
        self.Ya, self.Yb, self.Esum = (z(self.b, self.d) for _ in range(3))
        self.H, self.Greg, self.Ga, self.Gb = (z(self.b, self.d) for _ in range(4))
        B3 = 3 * self.max_batch
        self.req_rows = z(B3, 2 * self.d)                   # [Esum | E0] rows of my triplets
        self.gc_star, self.gc_reg = z(B3, self.d), z(B3, self.d)
        self.terms = z(8 * self.max_batch)
        self.adam = E.AdamState(lr)
        self._cu = torch.arange(self.max_batch, dtype=torch.int32, device=dev)
        self._cp = self._cu.clone()
        self.router = RowRouter(comm, self.part, self.max_batch)
        self._offsets = (0, self.n_users, self.n_users)
        self._gidx = None
        # rows this rank was asked for in the current step (= rows of E* the loss reads = rows that receive gradient):
        # the last forward hop produces only those, the first backward hop skips operand rows outside them
        self.flag = torch.zeros(self.b, dtype=torch.uint8, device=dev)
        self.flagX = self.flag if (not comm.live or self.pipeline) else \
            torch.zeros(self.Npad, dtype=torch.uint8, device=dev)
        self.es_buf, self.e0_buf = z(B3, self.d), z(B3, self.d)
        self._pow2 = ((self.L + 1) & self.L) == 0
        self._Gs = None

    def _from_block(self, indptr, indices, vals):
        """CSR of this rank's nu user rows followed by its ni item rows, GLOBAL node ids as columns
        (numpy or device tensors) -> the padded [b]-row local block whose columns are positions in the
        gathered layout; a row keeps its storage order (ascending node id)."""
        bu = self.part.bu
        if isinstance(indices, torch.Tensor):
            ip = torch.zeros(self.b + 1, dtype=torch.int64, device=indices.device)
            ip[1:self.nu + 1] = indptr[1:self.nu + 1]
            ip[self.nu + 1:bu + 1] = indptr[self.nu]
            ip[bu + 1:bu + self.ni + 1] = indptr[self.nu + 1:self.nu + self.ni + 1]
            ip[bu + self.ni + 1:] = indptr[self.nu + self.ni]
            cols = self.part.position(indices.long()).to(torch.int32)
            return self._with_chunks(E.SpmmCSR(ip, cols, vals, n_cols=self.Npad), ip, indices, vals)
        indptr = np.asarray(indptr, dtype=np.int64)
        ip = np.zeros(self.b + 1, dtype=np.int64)
        ip[1:self.nu + 1] = indptr[1:self.nu + 1]
        ip[self.nu + 1:bu + 1] = indptr[self.nu]
        ip[bu + 1:bu + self.ni + 1] = indptr[self.nu + 1:self.nu + self.ni + 1]
        ip[bu + self.ni + 1:] = indptr[self.nu + self.ni]
        cols = self.part.position(np.asarray(indices, dtype=np.int64)).astype(np.int32)
        return self._with_chunks(E.SpmmCSR(ip, cols, np.asarray(vals, np.float32), n_cols=self.Npad), ip, indices, vals)

    def _with_chunks(self, A, ip, cols_global, vals):
        """attach the chunked (pipelined) form of the block (ChunkedHop) where the hop is pipelined"""
        A.chunked = None
        if self.pipeline and self._d_hint in (64, 128, 256):
            A.chunked = ChunkedHop(self.part, self.rank, ip, cols_global, vals, A.exact_row_nnz(self._d_hint),
                                   A.indices.device)
        return A

    def _local_rows(self, a):
        import scipy.sparse as sp
        blk = sp.vstack([a[self.ulo:self.uhi], a[self.n_users + self.ilo:self.n_users + self.ihi]]).tocsr()
        return self._from_block(blk.indptr, blk.indices, blk.data)

    @property
    def X(self):
        if self._X is None:
            self._X = torch.zeros((self.Npad, self.d), dtype=torch.float32, device=self.E0.device)
        return self._X

    def natural(self, gathered):
        """[n_pad][d] gathered (rank-major) table -> (user rows [U][d], item rows [I][d]) in id order"""
        if self._gidx is None:
            self._gidx = self.part.gathered_index(gathered.device)
        return gathered[self._gidx[0]], gathered[self._gidx[1]]

    def plan_epoch(self, users, pos, neg, batch):
        """Routing counts of every batch of this rank's epoch stream (see RowRouter.plan_epoch);
        afterwards step(..., batch_index=k) runs without host synchronisation."""
        self.router.plan_epoch([users, pos, neg], self._offsets, batch)

    # ------------------------------------------------------------------ propagation
    def _operand(self, local):
        """the gathered [n_pad][d] operand of a hop: one all-gather (at one rank the block itself)"""
        if not self.comm.live:
            return local
        self.comm.all_gather_rows(local, self.X)                        # exchange: one all-gather per hop
        return self.X

    def propagate(self, wanted=None):
        """Esum (this rank's rows) = Σ_k E^k; returns it (E* = Esum / (L+1)).  wanted: uint8 [b] — the last hop
        then produces only the flagged rows (the rows a step's loss reads; other rows of Esum are stale)."""
        if self.L == 0:
            self.Esum.copy_(self.E0)
            return self.Esum
        src, acc_in = self.E0, self.E0
        ping = (self.Ya, self.Yb)
        for k in range(self.L):
            last = k == self.L - 1
            out = None if last else ping[k & 1]                         # the last layer is only needed in the sum
            if self.A.chunked is not None:
                self.A.chunked.matmul(self.comm, src, out=out, sum_in=acc_in, sum_out=self.Esum,
                                      y_row_wanted=wanted if last else None)
            else:
                self.A.matmul(self._operand(src), out=out, sum_in=acc_in, sum_out=self.Esum,
                              y_row_wanted=wanted if last else None)
            src, acc_in = out, self.Esum
        return self.Esum

    def final_embeddings(self):
        """Full (user, item) tables on every rank (one all-gather; evaluation entrance)."""
        loc = torch.zeros_like(self.Esum)
        E.div_scalar(self.propagate(), float(self.L + 1), loc)
        full = self.X if self.comm.live else torch.empty_like(loc)
        self.comm.all_gather_rows(loc, full)
        return self.natural(full)

    def step(self, users, pos, neg, loss_out=None, batch_index=None):
        """One optimiser step on this rank's B triplets (global batch = all ranks' triplets)."""
        B, d, dev = users.numel(), self.d, self.E0.device
        if B > self.max_batch:
            raise ValueError("batch larger than max_batch")
        # --- routing: ids -> owners (native: requests in owner order, one all-to-all of (row, code) pairs)
        rt = self.router.planned_route(batch_index, B)
        if rt is None:
            rt = self.router.request(users, pos, neg, self.n_users, self.router.epoch_counts(batch_index, B))
        n_asked = rt.asked.numel()
        # --- forward: the last hop only on the rows somebody asked for
        E.mark_rows(rt.asked, self.flag)
        esum = self.propagate(wanted=self.flag)
        # --- lookup answers: [Esum | E0] rows back to the askers, unscattered into request order by the gather
        rows = torch.empty((n_asked, 2 * d), dtype=torch.float32, device=dev)
        E.rows_gather2(rt.asked, esum, self.E0, rows[:, :d], rows[:, d:])
        got, _ = self.comm.all_to_all_rows(rows, rt.recv_counts, rt.send_counts)
        es, e0 = self.es_buf[:3 * B], self.e0_buf[:3 * B]
        E.rows_gather2(rt.inv, got[:, :d], got[:, d:], es, e0)          # es[p], e0[p] = answer to request p
        # --- BPR head on the compact block: "users" = rows [0,B), "items" = rows [B,3B); every occurrence has its
        #     own row there, so gs / gr hold one gradient row per occurrence (gs already divided by L+1 when that
        #     is exact: L+1 a power of two)
        gs, gr = self.gc_star[:3 * B], self.gc_reg[:3 * B]
        if _ATOMIC_HEADS:                  # the A/B knob's heads ADD into their rows (the ordered heads store them)
            gs.zero_()
            gr.zero_()
        E.lightgcn_bpr_grad(es, e0, B, self.L, self._cu[:B], self._cp[:B], _compact_neg(self, B),
                            self.reg, gs, gr, self.terms, loss_out, divided=self._pow2, plan=_compact_plan(self, B))
        # --- gradient rows back to the owners (routed order), added there in the order of the global batch
        back = torch.empty((3 * B, 2 * d), dtype=torch.float32, device=dev)
        E.rows_gather2(rt.order, gs, gr, back[:, :d], back[:, d:])
        mine, _ = self.comm.all_to_all_rows(back, rt.send_counts, rt.recv_counts)
        keys, index_of_pos = self.router.ordered_keys(rt)
        if self._pow2:
            E.rows_sum_sorted2(keys, index_of_pos, mine[:, :d], self.H, mine[:, d:], self.Greg)
        else:
            if self._Gs is None:                                         # zero outside the rows of a step
                self._Gs = torch.zeros_like(self.H)
            E.rows_sum_sorted2(keys, index_of_pos, mine[:, :d], self._Gs, mine[:, d:], self.Greg)
            E.rows_div(rt.asked, self._Gs, float(self.L + 1), self.H)
            E.rows_clear(rt.asked, d, (self._Gs,))
        # --- backward hops: G_k = H + Aᵀ G_{k+1}; H is non-zero on the asked rows only (first hop skips the rest),
        #     the last hop carries ApplyAdam as its epilogue where the lane-group schedule exists
        chunked = self.At.chunked is not None
        if self.comm.live and not chunked:
            self.comm.all_gather_rows(self.flag, self.flagX)
        g = self.H
        ping = (self.Ga, self.Gb)
        applied = False
        for k in range(self.L):
            if chunked:                                 # the operand arrives chunk by chunk under the launches
                out = ping[(k + 1) & 1]
                self.At.chunked.matmul(self.comm, g, out=out, addend=self.H)
                g = out
                continue
            X = self._operand(g)
            if k == self.L - 1 and self.L >= 2:
                applied = self.At.matmul_adam(X, self.H, self.Greg, self.E0, self.m, self.v, self.adam,
                                              row_flag=self.flag)
                if applied:
                    break
            out = ping[(k + 1) & 1]
            self.At.matmul(X, out=out, addend=self.H, x_row_nonzero=self.flagX if k == 0 else None)
            g = out
        if not applied:
            E.adam_dense2(self.E0, self.m, self.v, g, self.Greg, self.adam)
            E.rows_clear(rt.asked, d, (self.H, self.Greg), self.flag)   # rows the ordered sums stored, and the flags
        self.adam.advance()
        return loss_out


class ShardedMF:
    """BPR-MF with both tables row-sharded (SURVEY §8e "BPR-MF step"): rank r owns the padded block
    [r·b, (r+1)·b) of the node rows (users first, then items) with its Adam moments.  One step
    (MF.py:54-76,101) = ids to the owners (all-to-all) -> the requested rows back (all-to-all) ->
    the BPR head on the compact [3B][d] block -> gradient rows to the owners (all-to-all), added
    there in the order of the global batch (sorted keys, no atomics) -> owner-local TF-sparse Adam,
    which sweeps every local row.  Bit-identical to the single-process step on the
    concatenated global batch."""

    def __init__(self, comm, user_table, item_table, lr, reg, max_batch):
        dev = E.require_gpu()
        self.comm, self.rank, self.world = comm, comm.rank, comm.world
        ut = np.asarray(user_table, dtype=np.float32)
        it = np.asarray(item_table, dtype=np.float32)
        self.n_users, self.n_items, self.d = ut.shape[0], it.shape[0], ut.shape[1]
        self.N = self.n_users + self.n_items
        self.part = parallel.BipartitePartition(self.n_users, self.n_items, self.world)
        self.b = self.part.b
        ulo, uhi = self.part.users_of(self.rank)
        ilo, ihi = self.part.items_of(self.rank)
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        self.T = z(self.b, self.d)                                   # my user rows, then my item rows
        self.T[:uhi - ulo] = torch.from_numpy(np.ascontiguousarray(ut[ulo:uhi])).to(dev)
        self.T[self.part.bu:self.part.bu + ihi - ilo] = torch.from_numpy(np.ascontiguousarray(it[ilo:ihi])).to(dev)
        self.m, self.v, self.G = z(self.b, self.d), z(self.b, self.d), z(self.b, self.d)
        self.reg, self.max_batch = float(reg), int(max_batch)
        self.adam = E.AdamState(lr)
        B3 = 3 * self.max_batch
        self.req = z(B3, self.d)
        self._gcat = z(B3, self.d)               # per-occurrence gradient rows: users' [B], then items' [2B]
        self.terms = z(8 * self.max_batch)
        self._ar = torch.arange(2 * self.max_batch, dtype=torch.int32, device=dev)
        self.router = RowRouter(comm, self.part, self.max_batch)
        self._offsets = (0, self.n_users, self.n_users)
        self._gidx = None

    def plan_epoch(self, users, pos, neg, batch):
        self.router.plan_epoch([users, pos, neg], self._offsets, batch)

    def step(self, users, pos, neg, loss_out, batch_index=None):
        B, d = users.numel(), self.d
        if B > self.max_batch:
            raise ValueError("batch larger than max_batch")
        rt = self.router.planned_route(batch_index, B)                          # exchange 1 (ids) done with the epoch's
        if rt is None:
            rt = self.router.request(users, pos, neg, self.n_users, self.router.epoch_counts(batch_index, B))
        rows = torch.empty((rt.asked.numel(), d), dtype=torch.float32, device=self.T.device)
        E.rows_gather(rt.asked, self.T, rows)
        got, _ = self.comm.all_to_all_rows(rows, rt.recv_counts, rt.send_counts)   # exchange 2: rows
        req = self.req[:3 * B]
        E.rows_gather(rt.inv, got, req)                                        # req[p] = answer to request p
        # compact tables: P' = rows [0,B) (one per triplet), Q' = rows [B,3B) (pos then neg)
        P, Q = req[:B], req[B:3 * B]
        gP, gQ = self._gcat[:B], self._gcat[B:3 * B]
        if _ATOMIC_HEADS:
            self._gcat[:3 * B].zero_()
        E.bpr_mf_grad(P, Q, self._ar[:B], self._ar[:B], _compact_neg(self, B), self.reg,
                      gP, gQ, self.terms, loss_out, _compact_plan(self, B))
        back = torch.empty((3 * B, d), dtype=torch.float32, device=self.T.device)
        E.rows_gather(rt.order, self._gcat[:3 * B], back)                       # routed order
        mine, _ = self.comm.all_to_all_rows(back, rt.send_counts, rt.recv_counts)  # exchange 3: gradients
        keys, index_of_pos = self.router.ordered_keys(rt)
        E.rows_sum_sorted(keys, index_of_pos, mine, self.G)
        E.adam_sparse(self.T, self.m, self.v, self.G, self.adam)                # clears G
        self.adam.advance()

    def tables(self):
        """Full (P, Q) on every rank (one all-gather; evaluation entrance)."""
        full = torch.empty(self.b * self.world, self.d, dtype=torch.float32, device=self.T.device)
        self.comm.all_gather_rows(self.T, full)
        if self._gidx is None:
            self._gidx = self.part.gathered_index(full.device)
        return full[self._gidx[0]], full[self._gidx[1]]
