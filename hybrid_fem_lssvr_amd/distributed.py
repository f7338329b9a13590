"""Multi-GPU layer: contiguous element shards, one RCCL all-gather to stitch.

One process per GPU (``torch.distributed``, backend ``"nccl"`` = RCCL on ROCm).
Element ``i`` needs only ``x_i, x_{i+1}, u_i, u_{i+1}`` (Dual.py:144-147), so the
per-element path shards embarrassingly: rank ``r`` owns the contiguous element range
``[s_r, s_{r+1})`` and holds the ``s_{r+1}-s_r+1`` nodes it touches (one halo node).
Nothing is exchanged before or during the kernels; the only collective is the
all-gather of the coefficient rows ``W`` (or of sampled ``u``) afterwards.

xGMI is point-to-point (7 links per GPU), so the gather is issued in a few large
chunks on a side stream while the next chunk is still being computed; chunk
boundaries are element-aligned and identical on every rank.

The stitching code is device-agnostic (it only uses ``torch.distributed``
collectives), which lets the CPU test-suite exercise it with ``gloo``.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class ShardPlan:
    """Contiguous partition of ``ne`` elements over ``world`` ranks; sizes differ by <= 1."""

    ne: int
    world: int

    def bounds(self, rank):
        base, rem = divmod(self.ne, self.world)
        s0 = rank * base + min(rank, rem)
        return s0, s0 + base + (1 if rank < rem else 0)

    def size(self, rank):
        s0, s1 = self.bounds(rank)
        return s1 - s0

    @property
    def max_size(self):
        return -(-self.ne // self.world)

    def node_slice(self, rank):
        """Nodes (and nodal values) rank needs: its elements' end points."""
        s0, s1 = self.bounds(rank)
        return slice(s0, s1 + 1)

    def owner_of_element(self, e):
        base, rem = divmod(self.ne, self.world)
        cut = rem * (base + 1)
        if e < cut:
            return e // (base + 1)
        return rem + (e - cut) // max(base, 1)


ALLGATHER_ALGOS = ("collective", "pairs")

# xGMI on an 8-GPU MI355X node: 7 point-to-point links per GPU, ~76.8 GB/s per direction each
XGMI_LINKS = 7
XGMI_LINK_GBPS_PER_DIRECTION = 76.8


def stitch_traffic_model(ne_global, world, bytes_per_element):
    """Bytes one rank RECEIVES per stitched step (equal padded blocks, rank-major) and the xGMI time
    floor of the two algorithms -- the model of DESIGN.md section 8, used by bench.py for the byte
    counts it reports and for the predictions the measured N > 1 line is compared with:
      direct all-pairs: every link carries one peer's block, all 7 at once  -> block / link rate
      ring            : world-1 steps of one block over one link           -> (world-1) blocks / link rate
    """
    plan = ShardPlan(int(ne_global), int(world))
    block = plan.max_size * int(bytes_per_element)
    recv = (int(world) - 1) * block
    link = XGMI_LINK_GBPS_PER_DIRECTION * 1e9
    waves = -(-(int(world) - 1) // XGMI_LINKS) if world > 1 else 0          # more peers than links: several rounds
    return {"block_bytes": block, "bytes_received_per_rank_per_step": recv,
            "direct_all_pairs_floor_s": waves * block / link, "ring_floor_s": (int(world) - 1) * block / link,
            "inbound_GBps_at_direct_floor": (recv / (waves * block / link) / 1e9) if world > 1 else 0.0}


def allgather_flat(out, src, rank, world, *, algo="collective", group=None):
    """Every rank contributes ``src`` (1-D, equal length on all ranks) and receives all of them,
    rank-major, in ``out`` (1-D, ``world * src.numel()``).

    ``collective``: the backend's all-gather (RCCL ``ncclAllGather`` on MI355X; on a fully
    connected xGMI node RCCL picks its own ring / direct algorithm).
    ``pairs``: direct all-pairs exchange -- one batched group of point-to-point sends and
    receives, a different peer on every one of the 7 xGMI links at once, each block landing in
    its final place (a ring is bound by one link, the direct exchange drives
    all of them).  Both forms are asynchronous on the current stream for device tensors.
    """
    n = src.numel()
    if out.numel() != world * n:
        raise ValueError(f"out must hold world*n = {world * n} elements, got {out.numel()}")
    if world == 1:
        out.copy_(src)
        return out
    if algo == "collective":
        dist.all_gather_into_tensor(out, src, group=group)
        return out
    if algo != "pairs":
        raise ValueError(f"unknown all-gather algorithm {algo!r} (choose from {ALLGATHER_ALGOS})")
    ops_ = []
    for off in range(1, world):
        to, frm = (rank + off) % world, (rank - off) % world
        ops_.append(dist.P2POp(dist.isend, src, to, group=group))
        ops_.append(dist.P2POp(dist.irecv, out[frm * n:(frm + 1) * n], frm, group=group))
    out[rank * n:(rank + 1) * n].copy_(src)
    for req in dist.batch_isend_irecv(ops_):
        req.wait()
    return out


def allgather_rows(local, plan, rank, group=None, chunks=1, out=None, compute_chunk=None,
                   algo="collective", stats=None):
    """Stitch per-rank row blocks into the global array, on every rank -- every block lands in ``out`` where
    it belongs, with as few extra passes over HBM as the algorithm allows (round 3 staged every chunk and then
    issued ``world`` copy kernels per chunk: each received byte crossed HBM twice more than needed):

    * ``algo="pairs"``: point-to-point receives straight into ``out``'s row ranges (shards of any size, any
      chunking): no staging buffer, no copy of received bytes;
    * ``algo="collective"``, equal shards, one chunk: ``all_gather_into_tensor`` straight into ``out``;
    * ``algo="collective"`` otherwise (chunked, or shard sizes differing by one): the backend's all-gather
      needs one contiguous equal-sized destination, so the chunk is gathered into a staging buffer and moved
      by ONE strided copy (equal shards) or ONE ``index_copy_`` (ragged) -- not ``world`` copies.

    This rank's own rows: when ``compute_chunk`` produces them and no ``local`` buffer is given, they are
    computed directly inside ``out`` (no copy at all); otherwise one copy per chunk.

    local: [max_size or size(rank), C] tensor (rows beyond ``plan.size(rank)`` ignored)
           -- or None when ``compute_chunk`` produces it piecewise.
    compute_chunk(r0, r1, dst): optional; fills ``dst`` (a [r1-r0, C] view of the local
           block) for local rows [r0, r1) on the current stream.  With ``chunks > 1`` the
           gather of chunk i overlaps the computation of chunk i+1 (CUDA tensors only).
    stats: optional dict, filled with ``copy_calls`` / ``bytes_copied`` (device copies of row data issued
           here), ``staged_bytes`` and ``collectives`` / ``p2p_ops`` -- what the tests count.
    Returns ``out`` [plan.ne, C].
    """
    world = plan.world
    s0, s1 = plan.bounds(rank)
    n_loc = s1 - s0
    pad = plan.max_size
    if algo not in ALLGATHER_ALGOS:
        raise ValueError(f"unknown all-gather algorithm {algo!r} (choose from {ALLGATHER_ALGOS})")
    if local is None and compute_chunk is None:
        raise ValueError("need local rows or a compute_chunk callback")
    ref = local if local is not None else out
    if ref is None:
        raise ValueError("pass `out` when rows are produced by compute_chunk")
    C = ref.shape[1]
    device, dtype = ref.device, ref.dtype
    esize = torch.empty((), dtype=dtype).element_size()
    if out is None:
        out = torch.empty((plan.ne, C), dtype=dtype, device=device)
    st = stats if stats is not None else {}
    for key in ("copy_calls", "bytes_copied", "staged_bytes", "collectives", "p2p_ops"):
        st.setdefault(key, 0)

    def copied(rows):
        st["copy_calls"] += 1
        st["bytes_copied"] += int(rows) * C * esize

    equal = plan.ne % world == 0
    in_place = local is None                       # compute this rank's rows inside `out`
    own = out[s0:s1] if in_place else local
    chunks = max(1, min(int(chunks), pad)) if pad > 0 else 1
    step = -(-pad // chunks) if pad > 0 else 0
    direct = algo == "pairs" or world == 1 or (equal and chunks == 1)
    stage = send = None
    if not direct:
        stage = torch.empty((world, max(step, 1), C), dtype=dtype, device=device)
        st["staged_bytes"] = stage.numel() * esize
        if in_place or own.shape[0] < pad:
            # the collective ships equal blocks: ragged shards send a zero-padded copy of each chunk
            send = torch.zeros((max(step, 1), C), dtype=dtype, device=device)
    use_streams = device.type == "cuda"
    comm = torch.cuda.Stream(device=device) if use_streams else None
    main = torch.cuda.current_stream(device) if use_streams else None

    def exchange(r0, r1):
        lo, hi = min(r0, n_loc), min(r1, n_loc)        # this rank's rows of the chunk
        if direct:
            if not in_place and hi > lo:
                out[s0 + lo:s0 + hi].copy_(own[lo:hi])
                copied(hi - lo)
            if world == 1:
                return
            if algo == "collective":                    # equal shards, one chunk: rank-major IS the global order
                dist.all_gather_into_tensor(out.view(-1), out[s0:s1].reshape(-1), group=group)
                st["collectives"] += 1
                return
            reqs = []
            for off in range(1, world):
                to, frm = (rank + off) % world, (rank - off) % world
                f0, f1 = plan.bounds(frm)
                flo, fhi = min(r0, f1 - f0), min(r1, f1 - f0)
                if hi > lo:
                    reqs.append(dist.P2POp(dist.isend, out[s0 + lo:s0 + hi], to, group=group))
                if fhi > flo:
                    reqs.append(dist.P2POp(dist.irecv, out[f0 + flo:f0 + fhi], frm, group=group))
            st["p2p_ops"] += len(reqs)
            if reqs:
                for req in dist.batch_isend_irecv(reqs):
                    req.wait()
            return
        # collective through the staging buffer: one gather, ONE move
        nrow = r1 - r0
        if send is not None:
            src = send[:nrow]
            if hi - lo < nrow:
                src[max(hi - lo, 0):].zero_()
            if hi > lo:
                src[:hi - lo].copy_(own[lo:hi])
                copied(hi - lo)
        else:
            src = own[r0:r1]
        dst = stage[:, :nrow]
        if not dst.is_contiguous():
            dst = torch.empty((world, nrow, C), dtype=dtype, device=device)
        dist.all_gather_into_tensor(dst.view(-1), src.reshape(-1), group=group)
        st["collectives"] += 1
        if equal:
            out.view(world, pad, C)[:, r0:r1].copy_(dst)                 # one strided copy
            copied(world * nrow)
        else:
            idx = _ragged_index(plan, r0, r1, nrow, device)
            out.index_copy_(0, idx[0], dst.reshape(world * nrow, C).index_select(0, idx[1]))
            copied(idx[0].numel())

    for ci in range(chunks):
        r0 = ci * step
        r1 = min(pad, r0 + step)
        if r1 <= r0:
            break
        if compute_chunk is not None:
            lo, hi = min(r0, n_loc), min(r1, n_loc)
            if hi > lo:
                compute_chunk(lo, hi, own[lo:hi])
        if use_streams:
            ready = torch.cuda.Event()
            ready.record(main)
            comm.wait_event(ready)
            with torch.cuda.stream(comm):
                exchange(r0, r1)
                out.record_stream(comm)
                if stage is not None:
                    stage.record_stream(comm)
        else:
            exchange(r0, r1)
    if use_streams:
        main.wait_stream(comm)
    return out


_RAGGED_IDX = {}


def _ragged_index(plan, r0, r1, nrow, device):
    """(destination rows in the global array, source rows in the [world * nrow] staged chunk) of the rows
    [r0, r1) of every rank's block, for shard sizes that differ by one; cached per plan / chunk / device."""
    key = (plan.ne, plan.world, r0, r1, str(device))
    hit = _RAGGED_IDX.get(key)
    if hit is None:
        dst, src = [], []
        for r in range(plan.world):
            f0, f1 = plan.bounds(r)
            lo, hi = min(r0, f1 - f0), min(r1, f1 - f0)
            if hi > lo:
                dst.append(torch.arange(f0 + lo, f0 + hi))
                src.append(torch.arange(r * nrow + lo - r0, r * nrow + hi - r0))
        hit = (torch.cat(dst).to(device), torch.cat(src).to(device))
        if len(_RAGGED_IDX) > 64:
            _RAGGED_IDX.clear()
        _RAGGED_IDX[key] = hit
    return hit


def enhance_sharded(x_local, u_local, plan, rank, M, gamma, n_colloc=12, *, global_domain,
                    rhs=None, bc=(0.0, 0.0), group=None, chunks=4, gather=True, out=None,
                    algo="collective", stats=None):
    """Rank-local enhancement of this rank's shard + (optionally) the stitched global W.

    x_local/u_local: float64 device tensors of the shard's nodes (``plan.node_slice(rank)``).
    Returns (W_local [size, M], status [size], W_global [ne, M] or None).
    """
    from . import ops

    s0, s1 = plan.bounds(rank)
    n_loc = s1 - s0
    if x_local.numel() != n_loc + 1:
        raise ValueError(f"rank {rank} owns {n_loc} elements, expected {n_loc + 1} nodes")
    dev = x_local.device
    status = torch.empty((n_loc,), dtype=torch.int32, device=dev)
    kw = {} if rhs is None else {"rhs": rhs}

    def compute(lo, hi, dst):
        ops.enhance(x_local[lo:hi + 1], u_local[lo:hi + 1], M, gamma, n_colloc,
                    elem_offset=s0 + lo, ne_global=plan.ne, global_domain=global_domain, bc=bc,
                    out=dst, status=status[lo:hi], **kw)

    if not gather:
        W_loc = torch.empty((n_loc, M), dtype=torch.float64, device=dev)
        if n_loc:
            compute(0, n_loc, W_loc)
        return W_loc, status, None
    if out is None:
        out = torch.empty((plan.ne, M), dtype=torch.float64, device=dev)
    # the shard's rows are computed where they belong in the global array; the gather fills in the rest
    Wg = allgather_rows(None, plan, rank, group=group, chunks=chunks, out=out,
                        compute_chunk=compute, algo=algo, stats=stats)
    return Wg[s0:s1], status, Wg


# --------------------------------------------------------------------------
# sharded P1 solve: assemble locally (one halo node on the left), one 24-byte all-gather
# --------------------------------------------------------------------------
def combine_flux(a, b):
    """(a1,r1,g1) o (a2,r2,g2) = (a1+a2, r1+r2, g1+g2 + r2*a1): the associative operator of
    csrc/flux_solve.hip on [..., 3] tensors (a first, then b)."""
    return torch.stack([a[..., 0] + b[..., 0], a[..., 1] + b[..., 1],
                        (a[..., 2] + b[..., 2]) + b[..., 1] * a[..., 0]], dim=-1)


def solve_fem_sharded(x_ext, plan, rank, *, nquad=2, rhs=None, u0=0.0, u1=0.0, group=None):
    """``solve_fem`` (Dual.py:110-137) on a sharded mesh without gathering it anywhere.

    x_ext: this rank's nodes ``plan.node_slice(rank)`` PLUS one halo node on the left when
    rank > 0 (the load of the shard's first node needs the element left of it).  The global
    Dirichlet solve is the flux prefix scan: every shard contributes one (alpha, rho, gamma)
    aggregate, all-gathered (24 bytes per rank), combined in rank order.
    Returns (u_local float64[size+1] for the shard's own nodes, bands of the extended shard).
    """
    from . import ops

    s0, s1 = plan.bounds(rank)
    halo = 1 if s0 > 0 else 0
    n_loc = s1 - s0
    if x_ext.numel() != n_loc + 1 + halo:
        raise ValueError(f"rank {rank}: expected {n_loc + 1 + halo} nodes (shard + left halo)")
    kw = {} if rhs is None else {"rhs": rhs}
    bands = ops.p1_assemble(x_ext, nquad, want_local=True, **kw)
    kloc = bands["kloc"][halo:]
    load = bands["load"][halo:]
    agg, work = ops.p1_flux_aggregate(kloc, load, first_global=(s0 == 0))
    world = plan.world
    if world > 1:
        allagg = torch.empty((world, 3), dtype=torch.float64, device=agg.device)
        dist.all_gather_into_tensor(allagg.view(-1), agg, group=group)
    else:
        allagg = agg.view(1, 3)
    prefix = torch.zeros(3, dtype=torch.float64, device=agg.device)
    grand = torch.zeros(3, dtype=torch.float64, device=agg.device)
    for r in range(world):
        if r == rank:
            prefix = grand.clone()
        grand = combine_flux(grand, allagg[r])
    u = ops.p1_flux_finish(kloc, load, work, first_global=(s0 == 0), last_global=(s1 == plan.ne),
                           prefix=prefix, grand=grand, u0=u0, u1=u1)
    return u, bands


def solve_sharded(x_ext, plan, rank, M, gamma, n_colloc=12, *, global_domain, nquad=2, rhs=None,
                  bc=(0.0, 0.0), group=None, chunks=4, gather=True, algo="collective"):
    """Solve-then-enhance (Dual.py:171-174) on a sharded mesh: sharded P1 solve, rank-local
    enhancement, optional all-gather of W.  Returns (u_local, W_local, status, W_global|None)."""
    s0, _ = plan.bounds(rank)
    halo = 1 if s0 > 0 else 0
    u, _ = solve_fem_sharded(x_ext, plan, rank, nquad=nquad, rhs=rhs, u0=bc[0], u1=bc[1], group=group)
    Wl, st, Wg = enhance_sharded(x_ext[halo:], u, plan, rank, M, gamma, n_colloc,
                                 global_domain=global_domain, rhs=rhs, bc=bc, group=group,
                                 chunks=chunks, gather=gather, algo=algo)
    return u, Wl, st, Wg
