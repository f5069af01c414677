"""Cell-sharded execution over the GPUs of one node (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference is single-process: there is nothing to mirror, so the
partitioning follows the data: a rank owns a contiguous block of CELLS (one slab of the
cells-major matrices).  Exchange steps of the hot path:

  A  pooling     none   -- S_sz/U_sz are replicated inputs (any cell can be a neighbour)
  B  fit_slope   all-reduce(sum) of the per-gene moments, 3*G fp64 (720 KB at 30k genes)
  C  velocity    none   -- row-local
  D  colDeltaCor every rank needs the rows of `e` = Sx_sz that its cells' sampled neighbours live in.
                 Two exchanges are provided:
                   * HaloPlan (default when cells are sharded in embedding order): neighbours are near in the
                     embedding, so a spatially coherent shard needs only a HALO of remote rows; each rank sends
                     exactly the rows another rank's neighbour lists reference - one all_to_all_single with
                     uneven splits per pass (xGMI is point-to-point: every pair of GPUs has its own link, an
                     all-to-all keeps all 7 links busy with 1/10th of the all-gather volume);
                   * all_gather_rows: the whole matrix to everybody, ONE collective on the full shard
                     (largest possible message; xGMI rings are per-link bound).
                 Then the all-gather of the compact correlation rows (C x nrndm) for whoever wants them whole.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


import os

# VCY_FORCE_COLLECTIVES=1 issues the collectives even at world size 1 (single-GPU smoke test of the RCCL path)
FORCE = os.environ.get("VCY_FORCE_COLLECTIVES", "0") == "1"


def active() -> bool:
    """True when the exchange steps must be issued (more than one rank, or forced for testing)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE)


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced split of range(n): the first n % world ranks get one extra row."""
    base, extra = divmod(n, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_shard_bounds(n: int, world_size: int) -> List[Tuple[int, int]]:
    return [shard_bounds(n, world_size, r) for r in range(world_size)]


def _host_staged(t: torch.Tensor, group=None) -> bool:
    """gloo moves host memory: device tensors go through a host copy (multi-process tests on one GPU; RCCL never)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    if active():
        if _host_staged(t, group):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_reduce_abs_stats(st: torch.Tensor, group=None) -> torch.Tensor:
    """ops.abs_stats vectors [sum |e|, smallest non-zero |e|, non-zero count] of the ranks' shards -> the whole matrix's (sum,
    min, sum), in place: every rank then takes the same ops.partial_rules_for decision, whatever the sharding."""
    if active():
        mn = st[1:2].clone()
        all_reduce_sum(st, group)
        if _host_staged(mn, group):
            h = mn.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MIN, group=group)
            mn.copy_(h)
        else:
            dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
        st[1:2] = mn
    return st


def all_reduce_max(t: torch.Tensor, group=None) -> torch.Tensor:
    """MAX over the ranks, in place (a fact every rank must judge alike before it raises - e.g. the f64 sqrt element's domain)."""
    if active():
        if _host_staged(t, group):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def all_gather_rows(local: torch.Tensor, n_total: int, out: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """Reassemble a row-sharded (shard_bounds) tensor: local (n_loc, ...) -> (n_total, ...).
    Equal shards use one all_gather_into_tensor straight into `out`; ragged shards pad to the
    largest shard and trim."""
    rank, ws = world()
    if not active():
        if out is None:
            return local
        if out.data_ptr() != local.data_ptr():
            out.copy_(local)
        return out
    bounds = all_shard_bounds(n_total, ws)
    assert local.shape[0] == bounds[rank][1] - bounds[rank][0], "local shard does not match shard_bounds"
    tail = tuple(local.shape[1:])
    if out is None:
        out = torch.empty((n_total,) + tail, dtype=local.dtype, device=local.device)
    sizes = [b - a for a, b in bounds]
    if len(set(sizes)) == 1 and local.is_contiguous() and out.is_contiguous() and dist.get_backend(group) != "gloo":
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    mx = max(sizes)
    dev = torch.device("cpu") if _host_staged(local, group) else local.device
    buf = torch.zeros((mx,) + tail, dtype=local.dtype, device=dev)
    buf[: local.shape[0]] = local
    pieces = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(pieces, buf, group=group)
    for (a, b), p in zip(bounds, pieces):
        out[a:b] = p[: b - a]
    return out


def all_to_all_uneven(send: torch.Tensor, send_splits: Sequence[int], recv_splits: Sequence[int], group=None) -> torch.Tensor:
    """One all_to_all_single with uneven splits along dim 0 (RCCL); gloo (CPU tests, one-GPU logic tests) goes through
    point-to-point transfers of host copies.  Returns the received rows, peers in rank order."""
    rank, ws = world()
    tail = tuple(send.shape[1:])
    recv = torch.empty((int(sum(recv_splits)),) + tail, dtype=send.dtype, device=send.device)
    if not active():
        return recv
    if dist.get_backend(group) != "gloo":
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=list(recv_splits), input_split_sizes=list(send_splits), group=group)
        return recv
    send_h = send.cpu().contiguous()
    recv_h = torch.empty(recv.shape, dtype=recv.dtype)
    outs, ins = list(recv_h.split(list(recv_splits))), list(send_h.split(list(send_splits)))
    reqs = []
    for peer in range(ws):
        if peer == rank:
            if ins[peer].numel():
                outs[peer].copy_(ins[peer])
            continue
        if ins[peer].numel():
            reqs.append(dist.isend(ins[peer].contiguous(), peer, group=group))
        if outs[peer].numel():
            reqs.append(dist.irecv(outs[peer], peer, group=group))
    for r in reqs:
        r.wait()
    recv.copy_(recv_h)
    return recv


class HaloPlan:
    """Exchange of exactly the remote rows a rank needs (built once per neighbour graph, reused every pass).

    `need` is this rank's boolean mask over all n_total rows (rows its local work references).  All masks are
    all-gathered once; from them both sides of every (src -> dst) transfer derive the same ascending row list,
    so no indices travel in the data path.  exchange() packs the rows to send, runs ONE all_to_all_single with
    uneven splits and scatters what arrives into a full-height buffer at the rows' global positions (so kernels
    keep indexing rows by global cell id); rows nobody asked for are simply never written."""

    def __init__(self, need: torch.Tensor, n_total: int, group=None):
        self.rank, self.ws = world()
        self.n, self.group = int(n_total), group
        dev = need.device
        bounds = all_shard_bounds(self.n, self.ws)
        self.c0, self.c1 = bounds[self.rank]
        need = need.to(torch.uint8).contiguous()
        assert need.numel() == self.n
        if active():
            if dist.get_backend(group) != "gloo":
                masks = torch.empty((self.ws, self.n), dtype=torch.uint8, device=dev)
                dist.all_gather_into_tensor(masks, need, group=group)
            else:
                pieces = [torch.empty(self.n, dtype=torch.uint8) for _ in range(self.ws)]
                dist.all_gather(pieces, need.cpu(), group=group)
                masks = torch.stack(pieces).to(dev)
        else:
            masks = need[None, :]
        masks = masks.bool()
        send_idx, self.send_splits, recv_idx, self.recv_splits = [], [], [], []
        for peer, (a, b) in enumerate(bounds):
            if peer == self.rank:
                self.send_splits.append(0)
                self.recv_splits.append(0)
                continue
            mine_for_peer = torch.nonzero(masks[peer, self.c0:self.c1], as_tuple=False).ravel()          # local row numbers, ascending
            theirs_for_me = torch.nonzero(masks[self.rank, a:b], as_tuple=False).ravel() + a             # global row numbers, ascending
            send_idx.append(mine_for_peer)
            recv_idx.append(theirs_for_me)
            self.send_splits.append(int(mine_for_peer.numel()))
            self.recv_splits.append(int(theirs_for_me.numel()))
        cat = lambda xs: torch.cat(xs) if xs else torch.empty(0, dtype=torch.int64, device=dev)
        self.send_idx, self.recv_idx = cat(send_idx), cat(recv_idx)
        self.n_send, self.n_recv = int(self.send_idx.numel()), int(self.recv_idx.numel())
        self._send = self._recv = self._cur = None

    def localize(self, ixs: torch.Tensor) -> torch.Tensor:
        """Global row numbers -> row numbers of the COMPACT buffer a rank keeps when `e` is sharded: its own rows first
        (global c0..c1-1 -> 0..n_loc-1), then the received halo rows in the order they arrive (ascending global number ->
        n_loc, n_loc+1, ...).  Every index must be an own row or a row named in this rank's `need` mask."""
        n_loc = self.c1 - self.c0
        g = ixs.long()
        own = (g >= self.c0) & (g < self.c1)
        pos = torch.searchsorted(self.recv_idx, g.reshape(-1)).reshape(g.shape) if self.n_recv else torch.zeros_like(g)
        if self.n_recv:
            hit = self.recv_idx[pos.clamp(max=self.n_recv - 1)] == g
            assert bool((own | hit).all()), "localize: an index is neither an own row nor a halo row of this plan"
        else:
            assert bool(own.all()), "localize: an index is not an own row and the plan has no halo"
        return torch.where(own, g - self.c0, n_loc + pos).to(torch.int32).contiguous()

    def fetch(self, local: torch.Tensor) -> torch.Tensor:
        """The halo rows of a small per-row tensor (graph rows, size factors, ...): (n_recv, ...) in arrival order =
        ascending global row number, the order localize() assumes."""
        local = local.contiguous()
        recv = torch.empty((self.n_recv,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        h = self.begin(local, recv_out=recv)
        self.end(h, recv, row0=0)
        return recv

    def fetch_ragged(self, lens: torch.Tensor, *flat: torch.Tensor):
        """Halo rows of RAGGED per-row data (the CSR rows of a count layer): `lens` (n_loc) row lengths, every tensor of
        `flat` the rows' elements back to back.  Returns (lens of the n_recv halo rows, their elements back to back for each
        tensor of `flat`), rows in the order of fetch()."""
        lens = lens.to(torch.int64).contiguous()
        lens_halo = self.fetch(lens.reshape(-1, 1)).reshape(-1)
        ptr = torch.zeros(lens.numel() + 1, dtype=torch.int64, device=lens.device)
        torch.cumsum(lens, 0, out=ptr[1:])
        send_lens = lens[self.send_idx]
        # elements each peer gets / sends: row lengths summed over that peer's segment of the row lists
        seg = lambda v, splits: [int(x.sum()) for x in v.split(list(splits))] if v.numel() else [0] * len(splits)
        e_send, e_recv = seg(send_lens, self.send_splits), seg(lens_halo, self.recv_splits)
        total = int(send_lens.sum()) if send_lens.numel() else 0
        if total:
            sp = torch.zeros(send_lens.numel() + 1, dtype=torch.int64, device=lens.device)
            torch.cumsum(send_lens, 0, out=sp[1:])
            r = torch.repeat_interleave(torch.arange(send_lens.numel(), device=lens.device), send_lens)
            src = ptr[self.send_idx][r] + (torch.arange(total, device=lens.device) - sp[r])
        outs = []
        for f in flat:
            packed = f[src] if total else f[:0]
            outs.append(all_to_all_uneven(packed, e_send, e_recv, group=self.group))
        return (lens_halo, *outs)

    def exchange(self, local: torch.Tensor, out_full: torch.Tensor) -> torch.Tensor:
        """local: (c1-c0, ld) rows this rank owns; out_full: (n_total, ld).  Afterwards out_full holds the
        rank's own rows and every remote row its mask asked for (full-height, replicated-size buffer)."""
        return self.end(self.begin(local, out_full), out_full)

    def begin(self, local: torch.Tensor, out_full: Optional[torch.Tensor] = None, recv_out: Optional[torch.Tensor] = None):
        """First half of the exchange: rows to send packed, the all-to-all STARTED (RCCL: asynchronous on its own stream,
        so kernels launched next - work that needs no remote row - overlap with the transfer).  With `out_full` the rank's
        own rows are put in place in the full-height buffer; with `recv_out` (n_recv rows, e.g. the tail of the compact
        own+halo buffer) the received rows land there directly.  Returns the handle end() takes."""
        assert local.shape[0] == self.c1 - self.c0
        if out_full is not None:
            assert out_full.shape[0] == self.n and local.shape[1:] == out_full.shape[1:]
            if local.data_ptr() != out_full[self.c0:self.c1].data_ptr():      # callers may keep their rows inside out_full already
                out_full[self.c0:self.c1].copy_(local)
        if not active():
            return None                      # (forced collectives at world size 1 run the empty all-to-all: API smoke test)
        tail = tuple(local.shape[1:])
        if self._send is None or self._send.dtype != local.dtype or tuple(self._send.shape[1:]) != tail:
            self._send = torch.empty((self.n_send,) + tail, dtype=local.dtype, device=local.device)
            self._recv = None
        if recv_out is not None:
            assert recv_out.shape[0] == self.n_recv and tuple(recv_out.shape[1:]) == tail and recv_out.is_contiguous()
            recv = recv_out
        else:
            if self._recv is None or self._recv.dtype != local.dtype or tuple(self._recv.shape[1:]) != tail:
                self._recv = torch.empty((self.n_recv,) + tail, dtype=local.dtype, device=local.device)
            recv = self._recv
        self._cur = recv
        torch.index_select(local, 0, self.send_idx, out=self._send)
        if _host_staged(local, self.group):
            # gloo (one-GPU logic tests): point-to-point exchange of host copies, completed here
            send_h = self._send.cpu()
            recv_h = torch.empty(recv.shape, dtype=recv.dtype)
            outs = list(recv_h.split(self.recv_splits)) if self.n_recv else [recv_h[:0] for _ in range(self.ws)]
            ins = list(send_h.split(self.send_splits)) if self.n_send else [send_h[:0] for _ in range(self.ws)]
            reqs = []
            for peer in range(self.ws):
                if peer == self.rank:
                    continue
                if ins[peer].numel():
                    reqs.append(dist.isend(ins[peer].contiguous(), peer, group=self.group))
                if outs[peer].numel():
                    reqs.append(dist.irecv(outs[peer], peer, group=self.group))
            for r in reqs:
                r.wait()
            recv.copy_(recv_h)
            return "done"
        return dist.all_to_all_single(recv, self._send, output_split_sizes=self.recv_splits, input_split_sizes=self.send_splits,
                                      group=self.group, async_op=True)

    def end(self, handle, out: Optional[torch.Tensor] = None, row0: Optional[int] = None) -> Optional[torch.Tensor]:
        """Second half of the exchange: wait for the transfer (the current stream waits, not the host).  row0 None: scatter
        the received rows to their global positions in the full-height `out`; row0 given: the rows belong at out[row0:
        row0 + n_recv] (compact own+halo buffer) - nothing to do when begin() received straight into that slice."""
        if handle is None:
            return out
        if handle != "done":
            handle.wait()
        recv = self._cur
        if row0 is None:
            out.index_copy_(0, self.recv_idx, recv)
        elif out is not None and recv.data_ptr() != out[row0:row0 + self.n_recv].data_ptr():
            out[row0:row0 + self.n_recv].copy_(recv)
        return out


def overlap_schedules(base: torch.Tensor, interior: torch.Tensor, cells_per_round: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Two stage-D schedules for the halo overlap: (cells that run WHILE the halo rows move, cells that run after).

    `base`: the rank's cells in schedule order (int64 local cell numbers); `interior`: bool per local cell, True when every
    sampled neighbour is rank-local.  Only interior cells may run before the halo has landed, but not all of them have to:
    the grouped stage-D kernel keeps one workgroup per CU, every group of 8 cells costs the same, so a launch costs whole
    ROUNDS of `cells_per_round` (= CUs x cells per group) cells plus one tiled tail - and two launches pay two tails.  The
    first launch therefore takes the largest whole number of rounds the interior cells fill (at least one round covers the
    transfer of a halo: ~3 ms of work against ~2 ms of xGMI time at 8 ranks), everything else - the remaining interior cells
    and the cells with remote neighbours, in schedule order - goes to the second launch, which then holds the only tail.
    Measured on 6 250-cell shards (tools/shard_model.py): interior + remote as they fall 12.4 ms, round-aligned 9.5 ms, one
    launch 9.8 ms.  With less than one round of interior cells there is nothing to overlap: ([], all)."""
    inter_sched = base[interior[base]]
    n1 = (int(inter_sched.numel()) // cells_per_round) * cells_per_round
    if n1 == 0:
        return base[:0].to(torch.int32).contiguous(), base.to(torch.int32).contiguous()
    first = inter_sched[:n1]
    taken = torch.zeros(interior.numel(), dtype=torch.bool, device=base.device)
    taken[first] = True
    return first.to(torch.int32).contiguous(), base[~taken[base]].to(torch.int32).contiguous()


SELF_CHECKS = ("all_reduce_sum", "all_reduce_min", "all_to_all_uneven", "all_to_all_halo_sizes", "all_gather_rows_equal", "all_gather_rows_ragged", "all_gather_masks")

# Halo rows rank r RECEIVES from peer p in the 8-rank run of cfg3 (50 000 cells x 30 000 genes, Hilbert-ordered shards, nrndm 250):
# profiles/r04_shard_model.json, worlds["8"].ranks[r].halo_bytes_per_peer / (30 016 x 8 B).  The start-up check replays exactly these
# split vectors - empty segments, 7-row and 3 000-row segments side by side - at the real row width where the device has the memory
# (up to 723 MB to one peer: the size-dependent paths of the transport are built and used once before real data moves).
CFG3_HALO_ROWS_8 = ((0, 2119, 0, 0, 0, 0, 0, 0), (3013, 0, 1389, 7, 0, 0, 0, 0), (0, 202, 0, 375, 228, 0, 0, 0), (0, 108, 1413, 0, 1568, 0, 106, 0),
                    (0, 0, 329, 1446, 0, 956, 1373, 38), (0, 0, 0, 0, 580, 0, 1833, 0), (0, 0, 0, 302, 947, 1732, 0, 1524), (0, 0, 0, 0, 57, 0, 1435, 0))


def self_check(device: torch.device, group=None) -> dict:
    """Tiny instances of every collective SHAPE the sharded path issues, each checked against the values it must deliver, before
    any real data moves: the all-reduce of the fit moments (3 x G fp64, SUM) and of the branch-rule facts (MIN), the halo
    exchange (ONE all_to_all_single with uneven splits, zero-length segments included - the self segment always, some peers
    too), the all-gather of equal and of ragged row shards (correlation rows, Sx shards) and the uint8 mask all-gather HaloPlan
    is built from.  Returns {check: "ok" | "<error>"} plus "agreed": the per-check verdict AND-ed over the ranks (a MIN
    all-reduce), so that every rank takes the same fallback decision (bench.py: a failed uneven all-to-all switches the run to
    --exchange allgather).  Raises only when the plain all-reduce itself does not work - then no
    sharded run is possible and the message says which call failed."""
    rank, ws = world()
    res = {}
    if not active():
        return {"agreed": {}, "skipped": "one rank, collectives not forced"}

    inject = os.environ.get("VCY_SELF_CHECK_FAIL", "")       # tests: "name" or "name@rank" makes that check fail (on that rank only)

    def attempt(name, fn):
        try:
            fn()
            if inject and inject.split("@")[0] == name and (("@" not in inject) or int(inject.split("@")[1]) == rank):
                raise RuntimeError("failure injected by VCY_SELF_CHECK_FAIL")
            if device.type == "cuda":
                torch.cuda.synchronize(device)
            res[name] = "ok"
        except Exception as e:                                                  # noqa: BLE001 (the report IS the point)
            res[name] = f"{type(e).__name__}: {e}"[:240]

    def ar_sum():
        t = torch.full((3, 64), float(rank + 1), dtype=torch.float64, device=device)
        all_reduce_sum(t, group)
        assert float(t[2, 63]) == ws * (ws + 1) / 2, f"sum over ranks = {float(t[2, 63])}"

    def ar_min():
        st = torch.tensor([1.0, 10.0 + rank, 2.0], dtype=torch.float64, device=device)
        all_reduce_abs_stats(st, group)
        assert st.tolist() == [float(ws), 10.0, 2.0 * ws], st.tolist()

    def a2a():
        # rank r sends (r + 2 p) % 3 rows to peer p: lengths 1 and 2 always occur, empty segments between different ranks from 4 ranks up
        # (0 -> 3, 1 -> 4, ...), the self segment is always empty
        n_to = lambda src, dst: 0 if src == dst else (src + 2 * dst) % 3
        send_splits = [n_to(rank, p) for p in range(ws)]
        recv_splits = [n_to(p, rank) for p in range(ws)]
        send = torch.cat([torch.full((n, 4), 1000.0 * rank + p, dtype=torch.float32, device=device) for p, n in enumerate(send_splits)] +
                         [torch.empty((0, 4), dtype=torch.float32, device=device)])
        recv = all_to_all_uneven(send, send_splits, recv_splits, group)
        want = torch.cat([torch.full((n, 4), 1000.0 * p + rank, dtype=torch.float32, device=device) for p, n in enumerate(recv_splits)] +
                         [torch.empty((0, 4), dtype=torch.float32, device=device)])
        assert recv.shape == want.shape and torch.equal(recv, want), "rows arrived in the wrong place"

    def a2a_sizes():
        # the halo exchange at the SIZES of a real run: at 8 ranks cfg3's own table and row width (f64, 30 016 columns: 240 KB per row) when the
        # device has room for it; any other world size (and CPU tensors): the same kind of table - empty, short and long segments - on narrow rows
        if ws == 8:
            table = CFG3_HALO_ROWS_8
        else:
            table = tuple(tuple(0 if (src == dst or (7 * src + 3 * dst) % 4 == 0) else 5 + 997 * ((src + 2 * dst) % 5) for src in range(ws)) for dst in range(ws))
        recv_splits = list(table[rank])
        send_splits = [table[p][rank] for p in range(ws)]
        # Row width and element type are ONE decision for the whole group: free memory is a rank-local fact (and a shared, moving one
        # when several ranks sit on one device), and an all_to_all_single whose ranks disagree on the element size hangs or corrupts.
        # Every rank states whether the wide buffers fit where it is; the MIN over the ranks (the all-reduce verified just above) decides.
        fits = 0.0
        if ws == 8 and device.type == "cuda":
            free, _ = torch.cuda.mem_get_info(device)
            fits = 1.0 if free > 3 * (sum(recv_splits) + sum(send_splits)) * 30016 * 8 else 0.0
        claim = os.environ.get("VCY_SELF_CHECK_FITS")            # tests: "r0,r1,..." = the ranks whose local view says the wide buffers fit
        if claim is not None:
            fits = 1.0 if str(rank) in claim.split(",") else 0.0
        agreed = torch.tensor([fits], dtype=torch.float64, device=device)
        if _host_staged(agreed, group):
            h = agreed.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MIN, group=group)
            agreed = h
        else:
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN, group=group)
        width = 30016 if float(agreed[0]) > 0.5 else 4
        dt = torch.float64 if width > 4 else torch.float32
        send = torch.empty((sum(send_splits), width), dtype=dt, device=device)
        o = 0
        for p, n in enumerate(send_splits):
            send[o:o + n] = 1000.0 * rank + p
            o += n
        recv = all_to_all_uneven(send, send_splits, recv_splits, group)
        assert recv.shape == (sum(recv_splits), width), f"received {tuple(recv.shape)}"
        o = 0
        for p, n in enumerate(recv_splits):
            if n:
                seg = recv[o:o + n]
                assert float(seg.min()) == float(seg.max()) == 1000.0 * p + rank, f"segment of peer {p} holds {float(seg.min())} .. {float(seg.max())}"
            o += n

    def gather(n_total):
        def run():
            a, b = shard_bounds(n_total, ws, rank)
            local = torch.arange(a, b, dtype=torch.float32, device=device)[:, None].repeat(1, 3).contiguous()
            out = all_gather_rows(local, n_total, group=group)
            assert torch.equal(out[:, 1], torch.arange(n_total, dtype=torch.float32, device=device)), "rows out of order"
        return run

    def masks():
        need = torch.zeros(4 * ws, dtype=torch.bool, device=device)
        need[rank::ws] = True
        HaloPlan(need, 4 * ws, group)

    attempt("all_reduce_sum", ar_sum)
    if res["all_reduce_sum"] != "ok":
        raise RuntimeError(f"rank {rank}: the basic all-reduce of the '{dist.get_backend(group)}' backend failed, no sharded run is possible: {res['all_reduce_sum']}")
    attempt("all_reduce_min", ar_min)
    attempt("all_to_all_uneven", a2a)
    attempt("all_to_all_halo_sizes", a2a_sizes)
    attempt("all_gather_rows_equal", gather(2 * ws))
    attempt("all_gather_rows_ragged", gather(2 * ws + 1) if ws > 1 else gather(2))
    attempt("all_gather_masks", masks)
    flags = torch.tensor([1.0 if res[n] == "ok" else 0.0 for n in SELF_CHECKS], dtype=torch.float64, device=device)
    if _host_staged(flags, group):
        h = flags.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MIN, group=group)
        flags = h
    else:
        dist.all_reduce(flags, op=dist.ReduceOp.MIN, group=group)
    res["agreed"] = {n: bool(v > 0.5) for n, v in zip(SELF_CHECKS, flags.tolist())}
    return res
