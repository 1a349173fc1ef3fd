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
    """hop (how a propagation hop gets the other ranks' rows; NEUREC_ROWSHARD_HOP, default "sliced" where there is
    something to exchange and the width allows it, else "allgather"):
      "allgather"  one all-gather of the [b][d] blocks, then the one-launch SpMM (exact);
      "sliced"     the table lives as `col_slices` COLUMN SLABS [S][b][d/S]; SpMM column c depends on operand column c
                   only, so slab s+1 is all-gathered (all_gather_into_tensor: every xGMI link busy) while the
                   one-launch kernel runs on slab s, across hop boundaries too — no carry kernel, no rank order, and
                   not a single sum re-associated: every output element is the same chain over the same non-zeros
                   (exact; a d >= 128 table's 64-column slabs keep the work-item kernel and its 256-non-zero
                   segments, the single-GPU engine's association at that d);
      "chunked"    r04's form, kept for A/B: the operand in W rank-ordered broadcasts under W carry launches (exact;
                   whether RCCL's one-source broadcasts keep more than one link busy is not known);
      "reduce"     the reduced-exchange hop: user rows need item rows (all-gather of the item blocks only), item
                   rows are formed as per-rank PARTIALS over each rank's OWN user rows (the transpose of its user-row
                   block times its own block, no operand exchange) which an equal-split all-to-all delivers to the
                   owners, added there in rank order (nrhip_partials_sum_rows).  At config 4 a rank receives 0.9 GB
                   per hop instead of 4.9 GB, and both exchanges run under a local product.  NOT bitwise the single
                   engine (a sum of W per-rank partial sums) — deterministic, within fp32 rounding (tests: 1e-5)."""

    HOPS = ("allgather", "sliced", "chunked", "reduce")

    def __init__(self, comm, adj_csr, n_users, n_items, embed, n_layers, lr, reg, max_batch,
                 symmetric=None, local_rows=None, local_rows_t=None, hop=None, col_slices=None):
        """adj_csr: the full [N][N] scipy adjacency (each rank slices its block), or None with
        local_rows = (indptr, indices, vals) of this rank's row block only (global column ids;
        local_rows_t for Âᵀ when Â is not symmetric).  embed: the full [N][d] table or just this
        rank's [n_loc][d] rows."""
        dev = E.require_gpu()
        self.comm, self.rank, self.world = comm, comm.rank, comm.world
        self.d = d = int(embed.shape[1])
        if hop is None:
            hop = os.environ.get("NEUREC_ROWSHARD_HOP") or ("sliced" if comm.live and d in (64, 128, 256) else "allgather")
        if hop not in self.HOPS:
            raise ValueError("hop must be one of %s" % (self.HOPS,))
        if hop == "chunked" and d not in (64, 128, 256):
            raise NotImplementedError("chunked hop: widths 64 / 128 / 256")
        if hop == "reduce" and d not in (64, 128, 256):
            raise NotImplementedError("reduce hop: widths 64 / 128 / 256 (masked one-launch kernels)")
        self.hop = hop
        if col_slices is None:
            col_slices = int(os.environ.get("NEUREC_ROWSHARD_SLICES", "2")) if hop == "sliced" else 1
        self.S = S = int(col_slices)
        if hop != "sliced" and S != 1:
            raise ValueError("col_slices > 1 needs hop='sliced'")
        if d % S or (d // S) not in (16, 32, 64, 128, 256):
            raise ValueError("embed_size %d in %d column slices: slab width %s is not a built SpMM width" % (d, S, d / S))
        self.w = w = d // S
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.N = self.n_users + self.n_items
        self.L, self.reg, self.max_batch = int(n_layers), float(reg), int(max_batch)
        self.part = parallel.BipartitePartition(self.n_users, self.n_items, self.world)
        self.b, self.Npad = self.part.b, self.part.n_pad
        self.ulo, self.uhi = self.part.users_of(self.rank)
        self.ilo, self.ihi = self.part.items_of(self.rank)
        self.nu, self.ni = self.uhi - self.ulo, self.ihi - self.ilo
        if local_rows is not None:
            blk = self._block_arrays(*local_rows)
            blk_t = blk if local_rows_t is None else self._block_arrays(*local_rows_t)
        else:
            a = adj_csr.tocsr().astype(np.float32)
            a.sort_indices()
            if symmetric is None:
                symmetric = (a != a.T).nnz == 0
            blk = self._block_arrays(*self._slice_rows(a))
            blk_t = blk if symmetric else self._block_arrays(*self._slice_rows(a.T.tocsr()))
        self.A = self._gathered_matrix(*blk)
        self.At = self.A if blk_t is blk else self._gathered_matrix(*blk_t)
        self.R = self.Rt = None                            # (user-row matrix, partial matrix) of the reduce hop
        if hop == "reduce":
            self.R = self._reduce_matrices(blk, blk_t)
            self.Rt = self.R if blk_t is blk else self._reduce_matrices(blk_t, blk)
        del blk, blk_t
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        if isinstance(embed, torch.Tensor):             # this rank's rows (users then items) or the full table
            src_u, src_i = ((embed[:self.nu], embed[self.nu:self.nu + self.ni])
                            if embed.shape[0] == self.nu + self.ni and embed.shape[0] != self.N else
                            (embed[self.ulo:self.uhi], embed[self.n_users + self.ilo:self.n_users + self.ihi]))
            src_u, src_i = src_u.to(dev), src_i.to(dev)
        else:
            embed = np.asarray(embed, dtype=np.float32)
            if embed.shape[0] == self.nu + self.ni and embed.shape[0] != self.N:
                hu, hi_ = embed[:self.nu], embed[self.nu:]
            else:
                hu, hi_ = embed[self.ulo:self.uhi], embed[self.n_users + self.ilo:self.n_users + self.ihi]
            src_u = torch.from_numpy(np.ascontiguousarray(hu)).to(dev)
            src_i = torch.from_numpy(np.ascontiguousarray(hi_)).to(dev)
        # every [b][d] resident is S column slabs [S][b][w] (S = 1: the row-major block itself)
        b, bu = self.b, self.part.bu
        self.E0 = z(S, b, w)
        for s in range(S):
            self.E0[s, :self.nu] = src_u[:, s * w:(s + 1) * w]
            self.E0[s, bu:bu + self.ni] = src_i[:, s * w:(s + 1) * w]
        self.m, self.v = z(S, b, w), z(S, b, w)
        self.Ya, self.Yb, self.Esum = (z(S, b, w) for _ in range(3))
        self.H, self.Greg, self.Ga, self.Gb = (z(S, b, w) for _ in range(4))
        self._Xs = []                                       # receive buffers [n_pad][w] of the all-gathered operand
        B3 = 3 * self.max_batch
        self.gc_star, self.gc_reg = z(B3, d), z(B3, d)
        self.terms = z(8 * self.max_batch)
        self.adam = E.AdamState(lr)
        self._cu = torch.arange(self.max_batch, dtype=torch.int32, device=dev)
        self._cp = self._cu.clone()
        self.router = RowRouter(comm, self.part, self.max_batch)
        self._offsets = (0, self.n_users, self.n_users)
        self._gidx = None
        # rows this rank was asked for in the current step (= rows of E* the loss reads = rows that receive gradient):
        # the last forward hop produces only those, the first backward hop skips operand rows outside them
        self.flag = torch.zeros(b, dtype=torch.uint8, device=dev)
        self.flagX = self.flag if (not comm.live or hop in ("chunked", "reduce")) else \
            torch.zeros(self.Npad, dtype=torch.uint8, device=dev)
        self.es_buf, self.e0_buf = z(B3, d), z(B3, d)
        self._pow2 = ((self.L + 1) & self.L) == 0
        self._Gs = None
        # the one-launch kernels skip work by row flags at every width they have a lane-group schedule for, and at
        # 64 columns and more without one
        self._masks_ok = w >= 64 or (self.A.ensure_schedule(w) and self.At.ensure_schedule(w))
        if hop == "reduce":
            self._ipad = self.part.bi * self.world
            self._Z = z(bu + self._ipad, d)                 # operand of the user rows: [own users ; every item row]
            self._P = z(self._ipad, d)                      # my partial of every item row
            self._Rv = z(self._ipad, d)                     # the W partials of MY item rows, rank-major
            self.flagZ = torch.zeros(bu + self._ipad, dtype=torch.uint8, device=dev)

    # ------------------------------------------------------------------ this rank's rows of the adjacency
    def _slice_rows(self, a):
        import scipy.sparse as sp
        blk = sp.vstack([a[self.ulo:self.uhi], a[self.n_users + self.ilo:self.n_users + self.ihi]]).tocsr()
        return blk.indptr, blk.indices, blk.data

    def _block_arrays(self, indptr, indices, vals):
        """CSR of this rank's nu user rows followed by its ni item rows, GLOBAL node ids as columns (numpy or device
        tensors) -> device tensors (indptr padded to the [b]-row block, global columns int64, values); a row keeps
        its storage order (ascending node id)."""
        dev = torch.device("cuda", torch.cuda.current_device())
        t = lambda a_, dt: (a_ if isinstance(a_, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a_))).to(dev, dt)
        indptr, cg, va = t(indptr, torch.int64), t(indices, torch.int64), t(vals, torch.float32)
        bu = self.part.bu
        ip = torch.zeros(self.b + 1, dtype=torch.int64, device=dev)
        ip[1:self.nu + 1] = indptr[1:self.nu + 1]
        ip[self.nu + 1:bu + 1] = indptr[self.nu]
        ip[bu + 1:bu + self.ni + 1] = indptr[self.nu + 1:self.nu + self.ni + 1]
        ip[bu + self.ni + 1:] = indptr[self.nu + self.ni]
        return ip, cg, va

    def _gathered_matrix(self, ip, cg, va):
        """the block as the one-launch hop multiplies it: columns = positions in the rank-major gathered layout"""
        A = E.SpmmCSR(ip, self.part.position(cg).to(torch.int32), va, n_cols=self.Npad)
        # column slabs of a d >= 128 table run at 64 columns: keep the work-item kernel (256-non-zero segments) so the
        # association stays the single-GPU engine's at that d
        A.lane_group = not (self.S > 1 and self.d >= 128)
        A.chunked = None
        if self.hop == "chunked":
            A.chunked = ChunkedHop(self.part, self.rank, ip, cg, va, A.exact_row_nnz(self.d), A.indices.device)
        return A

    def _reduce_matrices(self, blk, blk_t):
        """(Mu, Mp) of the reduced-exchange hop Y = M·X for the matrix M whose rows of this rank are `blk` and whose
        TRANSPOSE's rows of this rank are `blk_t` (the same arrays when M is symmetric).
          Mu [bu] x [bu + I_pad]: my user rows; a user column (norm's self loop: my own row) -> that row of the
             operand's first bu rows, item i -> row bu + i (the all-gathered item blocks ARE the item table in id order);
          Mp [I_pad] x [b]: row i holds M[i][u] for MY users u (= the transpose of my user rows of Mᵀ, users ascending)
             and, for my own items, the same-side entries of the row (the self loop, last): the partial a rank forms
             from its own block alone."""
        ip, cg, va = blk
        ip_t, cg_t, va_t = blk_t
        part, U, W = self.part, self.n_users, self.world
        bu, bi, b, dev = part.bu, part.bi, self.b, cg.device
        ipad = bi * W
        n_u = int(ip[bu])
        cu = cg[:n_u]
        same = cu < U
        if bool((same & ((cu < self.ulo) | (cu >= self.uhi))).any()):
            raise NotImplementedError("reduce hop: a user-user edge to another rank (not a bipartite graph + self loops)")
        col_u = torch.where(same, cu - self.ulo, bu + (cu - U)).to(torch.int32)
        Mu = E.SpmmCSR(ip[:bu + 1].clone(), col_u, va[:n_u].clone(), n_cols=bu + ipad)
        n_t = int(ip_t[bu])
        ct, vt = cg_t[:n_t], va_t[:n_t]
        row_u = torch.repeat_interleave(torch.arange(bu, device=dev), ip_t[1:bu + 1] - ip_t[:bu], output_size=n_t)
        keep = ct >= U
        i0, i1 = int(ip[bu]), int(ip[b])
        ci, vi = cg[i0:i1], va[i0:i1]
        row_i = torch.repeat_interleave(torch.arange(bi, device=dev), ip[bu + 1:b + 1] - ip[bu:b], output_size=i1 - i0)
        ss = ci >= U
        loc = ci[ss] - U - self.ilo
        if bool(((loc < 0) | (loc >= bi)).any()):
            raise NotImplementedError("reduce hop: an item-item edge to another rank (not a bipartite graph + self loops)")
        rows = torch.cat([ct[keep] - U, self.ilo + row_i[ss]])
        cols = torch.cat([row_u[keep], bu + loc])
        vals = torch.cat([vt[keep], vi[ss]])
        order = torch.sort(rows, stable=True).indices        # inside an item: my users ascending, then the self loop
        ipp = torch.zeros(ipad + 1, dtype=torch.int64, device=dev)
        ipp[1:] = torch.cumsum(torch.bincount(rows, minlength=ipad), 0)
        Mp = E.SpmmCSR(ipp, cols[order].to(torch.int32), vals[order], n_cols=b)
        return Mu, Mp

    # ------------------------------------------------------------------ layouts
    def _xbuf(self, k):
        while len(self._Xs) <= k:
            self._Xs.append(torch.zeros((self.Npad, self.w), dtype=torch.float32, device=self.E0.device))
        return self._Xs[k]

    @property
    def X(self):
        """the gathered [n_pad][w] operand buffer of a hop (w = d with one slab)"""
        return self._xbuf(0)

    def rows_of(self, slabs):
        """[S][b][w] column slabs -> the [b][d] row-major block (a view when S = 1)"""
        return slabs[0] if self.S == 1 else torch.cat([slabs[s] for s in range(self.S)], dim=1)

    def table_rows(self):
        """this rank's [b][d] block of E0 (users first, then items), row-major"""
        return self.rows_of(self.E0)

    def natural(self, gathered):
        """[n_pad][*] gathered (rank-major) table -> (user rows [U][*], item rows [I][*]) in id order"""
        if self._gidx is None:
            self._gidx = self.part.gathered_index(gathered.device)
        return gathered[self._gidx[0]], gathered[self._gidx[1]]

    def plan_epoch(self, users, pos, neg, batch):
        """Routing counts of every batch of this rank's epoch stream (see RowRouter.plan_epoch);
        afterwards step(..., batch_index=k) runs without host synchronisation."""
        self.router.plan_epoch([users, pos, neg], self._offsets, batch)

    # ------------------------------------------------------------------ propagation
    def _hops(self, mats, hops):
        """Run a chain of hops Y_h = M·X_h (+ epilogue).  mats = (M in gathered-column form, its reduce pair);
        hops: dicts with src (slabs), out / addend / sum_in / sum_out (slabs or None), wanted (produce flagged rows
        only), nonzero (operand is zero outside flagged rows), adam (last backward hop: ApplyAdam as the epilogue
        where the d = 64 lane-group schedule exists).  Returns True when the last hop applied Adam itself."""
        M, red = mats
        comm, S = self.comm, self.S
        sl = lambda t, s: None if t is None else t[s]
        if self.hop == "reduce":
            for hp in hops:
                self._hop_reduce(red, hp)
            return False
        if M.chunked is not None:
            for hp in hops:
                M.chunked.matmul(comm, hp["src"][0], out=sl(hp.get("out"), 0), addend=sl(hp.get("addend"), 0),
                                 sum_in=sl(hp.get("sum_in"), 0), sum_out=sl(hp.get("sum_out"), 0),
                                 y_row_wanted=self.flag if hp.get("wanted") else None)
            return False
        tasks = [(h, s) for h in range(len(hops)) for s in range(S)]
        live = comm.live
        # task j's all-gather may be issued once the launches that produce its source slab (task j - S) and that
        # still read its receive buffer (task j - lag) have been enqueued: with two slabs or more the gather of the
        # next slab — of the next hop, too — is on the links while the current launch runs
        lag = min(S, 2)
        tok = {}

        def start(j):
            h, s = tasks[j]
            tok[j] = comm.all_gather_rows_start(hops[h]["src"][s], self._xbuf(j % lag))
        if live:
            for j in range(min(lag, len(tasks))):
                start(j)
        applied = False
        for j, (h, s) in enumerate(tasks):
            hp = hops[h]
            if live:
                comm.all_gather_rows_finish(tok.pop(j))
                X = self._xbuf(j % lag)
            else:
                X = hp["src"][s]
            masks = self._masks_ok
            if hp.get("adam") and S == 1 and M.matmul_adam(X, self.H[0], self.Greg[0], self.E0[0], self.m[0], self.v[0],
                                                           self.adam, row_flag=self.flag):
                applied = True
            else:
                M.matmul(X, out=sl(hp.get("out"), s), addend=sl(hp.get("addend"), s), sum_in=sl(hp.get("sum_in"), s),
                         sum_out=sl(hp.get("sum_out"), s),
                         x_row_nonzero=self.flagX if (hp.get("nonzero") and masks) else None,
                         y_row_wanted=self.flag if (hp.get("wanted") and masks) else None)
            if live and j + lag < len(tasks):
                start(j + lag)
        return applied

    def _hop_reduce(self, red, hp):
        """One reduced-exchange hop (class docstring).  Order of issue: all-gather of the item blocks ‖ my partial of
        every item row (own block only) -> all-to-all of the partials ‖ my user rows (gathered items) -> the owners'
        rank-ordered sums with the hop's epilogue."""
        Mu, Mp = red
        comm, bu, b, W = self.comm, self.part.bu, self.b, self.world
        src = hp["src"][0]
        row = lambda key, lo, hi: None if hp.get(key) is None else hp[key][0][lo:hi]
        wanted, nonzero = hp.get("wanted"), hp.get("nonzero")
        tok = comm.all_gather_rows_start(src[bu:b], self._Z[bu:])
        Mp.matmul(src, out=self._P, y_row_wanted=self.flagZ[bu:] if wanted else None,
                  x_row_nonzero=self.flag if nonzero else None)
        comm.all_gather_rows_finish(tok)
        tok = comm.all_to_all_equal_start(self._P, self._Rv)
        if self._same_side_users(Mu):
            E.copy2d(src[:bu], self._Z[:bu])
        if Mu.n_rows:
            Mu.matmul(self._Z, out=row("out", 0, bu), addend=row("addend", 0, bu), sum_in=row("sum_in", 0, bu),
                      sum_out=row("sum_out", 0, bu), y_row_wanted=self.flag[:bu] if wanted else None,
                      x_row_nonzero=self.flagZ if nonzero else None)
        comm.all_to_all_equal_finish(tok)
        E.partials_sum_rows(self._Rv.view(W, b - bu, self.d), W, out=row("out", bu, b), addend=row("addend", bu, b),
                            sum_in=row("sum_in", bu, b), sum_out=row("sum_out", bu, b),
                            row_mask=self.flag[bu:] if wanted else None)

    def _same_side_users(self, Mu):
        """does a user row of this matrix read a user row (norm's self loop)?  Decided once per matrix."""
        if not hasattr(Mu, "_same_side"):
            Mu._same_side = bool((Mu.indices[:Mu.nnz] < self.part.bu).any()) if Mu.nnz else False
        return Mu._same_side

    def _publish_flags(self):
        """the step's row flags where the masked hops of this form read them"""
        if self.hop == "reduce":
            bu = self.part.bu
            self.flagZ[:bu].copy_(self.flag[:bu])
            self.comm.all_gather_rows(self.flag[bu:], self.flagZ[bu:])
        elif self.flagX is not self.flag:
            self.comm.all_gather_rows(self.flag, self.flagX)

    def propagate(self, wanted=None):
        """Esum (this rank's rows, column slabs) = Σ_k E^k; returns it (E* = Esum / (L+1)).  wanted: True — the last
        hop then produces only the rows flagged in self.flag (the rows a step's loss reads; other rows of Esum are
        stale)."""
        if self.L == 0:
            self.Esum.copy_(self.E0)
            return self.Esum
        hops, src, acc_in = [], self.E0, self.E0
        ping = (self.Ya, self.Yb)
        for k in range(self.L):
            last = k == self.L - 1
            out = None if last else ping[k & 1]                         # the last layer is only needed in the sum
            hops.append(dict(src=src, out=out, sum_in=acc_in, sum_out=self.Esum, wanted=bool(wanted) and last))
            src, acc_in = out, self.Esum
        self._hops((self.A, self.R), hops)
        return self.Esum

    def local_pass(self, k=0):
        """one LOCAL full pass over this rank's rows (no collective): what bench.py times as the hop's kernel"""
        out = (self.Ya, self.Yb)[k & 1]
        if self.hop == "reduce":
            Mu, Mp = self.R
            Mp.matmul(self.E0[0], out=self._P)
            Mu.matmul(self._Z, out=out[0][:self.part.bu], addend=self.H[0][:self.part.bu])
            E.partials_sum_rows(self._Rv.view(self.world, self.b - self.part.bu, self.d), self.world,
                                out=out[0][self.part.bu:], addend=self.H[0][self.part.bu:])
            return
        for s in range(self.S):
            self.A.matmul(self._xbuf(0) if self.comm.live else self.E0[s], out=out[s], addend=self.H[s])

    def eval_factors(self):
        """The evaluation entrance of a row-sharded run: (this rank's user rows of E* [nu][d], the WHOLE item table of
        E* [I][d]).  uni_evaluator.py:101-157 scores a batch of users against every item, so a rank needs all item rows
        but only ITS users: the item blocks are all-gathered (config 4: 0.5 GB arrive per rank), the 5.1 GB user
        table never moves.  Same bits as the rows final_embeddings() returns."""
        esum = self.propagate()
        dev, d, w = esum.device, self.d, self.w
        bu, bi = self.part.bu, self.part.bi
        users = torch.empty((self.nu, d), dtype=torch.float32, device=dev)
        block = torch.zeros((bi, d), dtype=torch.float32, device=dev)        # my items, padded to the block
        loc = torch.empty((self.b, w), dtype=torch.float32, device=dev)
        for s in range(self.S):
            E.div_scalar(esum[s], float(self.L + 1), loc)
            users[:, s * w:(s + 1) * w] = loc[:self.nu]
            block[:self.ni, s * w:(s + 1) * w] = loc[bu:bu + self.ni]
        del loc
        items = torch.empty((self.world * bi, d), dtype=torch.float32, device=dev)
        self.comm.all_gather_rows(block, items)      # item blocks in rank order ARE the item table in id order
        return users, items[:self.n_items]

    def final_embeddings(self):
        """Full (user, item) tables on every rank (one all-gather per slab).  Small models and tests; an evaluation
        takes eval_factors() — only the item table travels."""
        esum = self.propagate()
        loc = torch.empty((self.b, self.w), dtype=torch.float32, device=esum.device)
        us, its = [], []
        for s in range(self.S):
            E.div_scalar(esum[s], float(self.L + 1), loc)
            full = self._xbuf(0) if self.comm.live else torch.empty_like(loc)
            self.comm.all_gather_rows(loc, full)
            u, i = self.natural(full)
            us.append(u)
            its.append(i)
        return (us[0], its[0]) if self.S == 1 else (torch.cat(us, 1), torch.cat(its, 1))

    def step(self, users, pos, neg, loss_out=None, batch_index=None):
        """One optimiser step on this rank's B triplets (global batch = all ranks' triplets)."""
        B, d, w, S, dev = users.numel(), self.d, self.w, self.S, self.E0.device
        if B > self.max_batch:
            raise ValueError("batch larger than max_batch")
        cols = lambda t, s: t[:, s * w:(s + 1) * w]
        # --- routing: ids -> owners (native: requests in owner order, one all-to-all of (row, code) pairs)
        rt = self.router.planned_route(batch_index, B)
        if rt is None:
            rt = self.router.request(users, pos, neg, self.n_users, self.router.epoch_counts(batch_index, B))
        n_asked = rt.asked.numel()
        # --- forward: the last hop only on the rows somebody asked for
        E.mark_rows(rt.asked, self.flag)
        if self.hop == "reduce":
            self._publish_flags()                          # its item rows are produced where they are NOT owned
        esum = self.propagate(wanted=True)
        # --- lookup answers: [Esum | E0] rows back to the askers, unscattered into request order by the gather
        rows = torch.empty((n_asked, 2 * d), dtype=torch.float32, device=dev)
        for s in range(S):
            E.rows_gather2(rt.asked, esum[s], self.E0[s], cols(rows, s), cols(rows[:, d:], s))
        got, _ = self.comm.all_to_all_rows(rows, rt.recv_counts, rt.send_counts)
        es, e0 = self.es_buf[:3 * B], self.e0_buf[:3 * B]
        E.rows_gather2(rt.inv, got[:, :d], got[:, d:], es, e0)          # es[p], e0[p] = answer to request p
        # --- BPR head on the compact block: "users" = rows [0,B), "items" = rows [B,3B); every occurrence has its
        #     own row there, so gs / gr hold one gradient row per occurrence (gs already divided by L+1 when that
        #     is exact: L+1 a power of two)
        gs, gr = self.gc_star[:3 * B], self.gc_reg[:3 * B]
        E.lightgcn_bpr_grad(es, e0, B, self.L, self._cu[:B], self._cp[:B], _compact_neg(self, B),
                            self.reg, gs, gr, self.terms, loss_out, divided=self._pow2, plan=_compact_plan(self, B))
        # --- gradient rows back to the owners (routed order), added there in the order of the global batch
        back = torch.empty((3 * B, 2 * d), dtype=torch.float32, device=dev)
        E.rows_gather2(rt.order, gs, gr, back[:, :d], back[:, d:])
        mine, _ = self.comm.all_to_all_rows(back, rt.send_counts, rt.recv_counts)
        keys, index_of_pos = self.router.ordered_keys(rt)
        if not self._pow2 and self._Gs is None:                         # zero outside the rows of a step
            self._Gs = torch.zeros_like(self.H)
        for s in range(S):
            if self._pow2:
                E.rows_sum_sorted2(keys, index_of_pos, cols(mine, s), self.H[s], cols(mine[:, d:], s), self.Greg[s])
            else:
                E.rows_sum_sorted2(keys, index_of_pos, cols(mine, s), self._Gs[s], cols(mine[:, d:], s), self.Greg[s])
                E.rows_div(rt.asked, self._Gs[s], float(self.L + 1), self.H[s])
                E.rows_clear(rt.asked, w, (self._Gs[s],))
        # --- backward hops: G_k = H + Aᵀ G_{k+1}; H is non-zero on the asked rows only (first hop skips the rest),
        #     the last hop carries ApplyAdam as its epilogue where the lane-group schedule exists
        if self.hop != "reduce":
            self._publish_flags()
        hops, g = [], self.H
        ping = (self.Ga, self.Gb)
        for k in range(self.L):
            out = ping[(k + 1) & 1]
            hops.append(dict(src=g, out=out, addend=self.H, nonzero=k == 0,
                             adam=(k == self.L - 1 and self.L >= 2)))
            g = out
        applied = self._hops((self.At, self.Rt), hops)
        if not applied:
            E.adam_dense2(self.E0, self.m, self.v, g, self.Greg, self.adam)
            for s in range(S):                                          # rows the ordered sums stored, and the flags
                E.rows_clear(rt.asked, w, (self.H[s], self.Greg[s]), self.flag if s == S - 1 else None)
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
        E.bpr_mf_grad(P, Q, self._ar[:B], self._ar[:B], _compact_neg(self, B), self.reg,
                      gP, gQ, self.terms, loss_out, _compact_plan(self, B))
        back = torch.empty((3 * B, d), dtype=torch.float32, device=self.T.device)
        E.rows_gather(rt.order, self._gcat[:3 * B], back)                       # routed order
        mine, _ = self.comm.all_to_all_rows(back, rt.send_counts, rt.recv_counts)  # exchange 3: gradients
        keys, index_of_pos = self.router.ordered_keys(rt)
        E.rows_sum_sorted(keys, index_of_pos, mine, self.G)
        E.adam_sparse(self.T, self.m, self.v, self.G, self.adam)                # clears G
        self.adam.advance()

    def eval_factors(self):
        """The evaluation entrance of a row-sharded run: (this rank's user rows [nu][d], the WHOLE item table [I][d]) —
        only the item blocks are all-gathered (cf. ShardedLightGCN.eval_factors)."""
        bu, bi = self.part.bu, self.part.bi
        ulo, uhi = self.part.users_of(self.rank)
        items = torch.empty((self.world * bi, self.d), dtype=torch.float32, device=self.T.device)
        self.comm.all_gather_rows(self.T[bu:bu + bi].contiguous(), items)
        return self.T[:uhi - ulo], items[:self.n_items]

    def tables(self):
        """Full (P, Q) on every rank (one all-gather; tests and the plugin's predict())."""
        full = torch.empty(self.b * self.world, self.d, dtype=torch.float32, device=self.T.device)
        self.comm.all_gather_rows(self.T, full)
        if self._gidx is None:
            self._gidx = self.part.gathered_index(full.device)
        return full[self._gidx[0]], full[self._gidx[1]]


class ShardedEvaluator:
    """Full-rank evaluation of a row-sharded model (SURVEY 8e "Evaluator"; uni_evaluator.py:101-157 on one process):
    users are independent units, so every rank ranks ITS users — rows [ulo, uhi) of the BipartitePartition — against
    the whole item table and the M·K metric sums are added over the ranks once (one all-reduce of M·K + 1 doubles).
    Nothing but the item table (ShardedLightGCN.eval_factors) and those sums is exchanged.

    train_rows / test_rows: DeviceCSR of this rank's users ONLY (row r = user ulo + r, global item ids) — e.g.
    DeviceCSR.rows(ulo, uhi) of the whole matrices; the strike plan and the user -> row table of the pruned path
    are then built over the rank's users, not over all 10^7."""

    def __init__(self, comm, train_rows, test_rows, metric_ids, top_k, batch_rows=32768, **evaluator_args):
        # batch_rows: the tile maxima of a batch are 4 B x 2·ceil(I / 64) per user (10^6 items: 4 GB at 32,768 users);
        # large batches matter at large I — the planned strikes and level 2's tile buckets work in 32-pair chunks of
        # ONE tile, which a batch fills only if it brings >= 32 pairs per tile (config 4: 2.04 M users/s at 8,192,
        # 2.38 M at 65,536)
        from .trainer import FullRankEvaluator
        if train_rows.n_rows != test_rows.n_rows:
            raise ValueError("train and test rows of a rank cover the same users")
        self.comm = comm
        self.ev = FullRankEvaluator(train_rows, test_rows, metric_ids, top_k, batch_rows=batch_rows, **evaluator_args)
        dev = test_rows.indptr.device
        has = (test_rows.indptr[1:] - test_rows.indptr[:-1]) > 0          # tool.py:63: users without test items are skipped
        self.users = torch.nonzero(has, as_tuple=False).flatten().to(torch.int32)
        self.n_local = int(self.users.numel())
        self.n_out = len(self.ev.metric_ids) * self.ev.top_k
        t = torch.tensor([float(self.n_local)], dtype=torch.float64, device=dev)
        comm.allreduce_sum_(t)
        self.n_total = int(t.item())

    def evaluate_factors(self, user_rows, item_table):
        """user_rows [nu][d]: this rank's rows of the user factors; item_table [I][d].  -> fp64 means [M·K] over ALL
        ranks' test users (the same numbers on every rank)."""
        dev = item_table.device
        both = torch.zeros(self.n_out + 1, dtype=torch.float64, device=dev)
        if self.n_local:
            sums = self.ev.evaluate_factors(user_rows, item_table, self.users, column_sums=True)
            both[:self.n_out] = torch.from_numpy(np.asarray(sums, np.float64)).to(dev)
            both[self.n_out] = float(self.ev.n_flagged)
        self.comm.allreduce_sum_(both)
        host = both.cpu().numpy()
        self.rows_redone = int(host[-1])
        return host[:-1] / max(self.n_total, 1)

    def evaluate(self, engine):
        """engine: ShardedLightGCN (anything with eval_factors())."""
        eu, items = engine.eval_factors()
        return self.evaluate_factors(eu, items)
