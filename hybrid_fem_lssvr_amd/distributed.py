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


def allgather_rows(local, plan, rank, group=None, chunks=1, out=None, compute_chunk=None):
    """Stitch per-rank row blocks into the global array, on every rank.

    local: [max_size or size(rank), C] tensor (rows beyond ``plan.size(rank)`` ignored)
           -- or None when ``compute_chunk`` produces it piecewise.
    compute_chunk(r0, r1, dst): optional; fills ``dst`` (a [r1-r0, C] view of the local
           block) for local rows [r0, r1) on the current stream.  With ``chunks > 1`` the
           gather of chunk i overlaps the computation of chunk i+1 (CUDA tensors only).
    Returns ``out`` [plan.ne, C].
    """
    world = plan.world
    n_loc = plan.size(rank)
    pad = plan.max_size
    if local is None and compute_chunk is None:
        raise ValueError("need local rows or a compute_chunk callback")
    ref = local if local is not None else out
    if ref is None:
        raise ValueError("pass `out` when rows are produced by compute_chunk")
    C = ref.shape[1]
    device, dtype = ref.device, ref.dtype
    if out is None:
        out = torch.empty((plan.ne, C), dtype=dtype, device=device)
    if local is None:
        local = torch.empty((pad, C), dtype=dtype, device=device)
    elif local.shape[0] < pad:
        grown = torch.zeros((pad, C), dtype=dtype, device=device)
        grown[:local.shape[0]] = local
        local = grown
    chunks = max(1, min(int(chunks), pad)) if pad > 0 else 1
    step = -(-pad // chunks) if pad > 0 else 0
    stage = torch.empty((world, max(step, 1), C), dtype=dtype, device=device)
    use_streams = device.type == "cuda"
    comm = torch.cuda.Stream(device=device) if use_streams else None
    main = torch.cuda.current_stream(device) if use_streams else None

    for ci in range(chunks):
        r0 = ci * step
        r1 = min(pad, r0 + step)
        if r1 <= r0:
            break
        if compute_chunk is not None:
            lo, hi = min(r0, n_loc), min(r1, n_loc)
            if hi > lo:
                compute_chunk(lo, hi, local[lo:hi])
        src = local[r0:r1]
        dst = stage[:, : r1 - r0]
        if use_streams:
            ready = torch.cuda.Event()
            ready.record(main)
            comm.wait_event(ready)
            with torch.cuda.stream(comm):
                _gather_chunk(dst, src, world, group)
                _scatter_rows(out, dst, plan, r0, r1)
                stage.record_stream(comm)
                out.record_stream(comm)
        else:
            _gather_chunk(dst, src, world, group)
            _scatter_rows(out, dst, plan, r0, r1)
    if use_streams:
        main.wait_stream(comm)
    return out


def _gather_chunk(dst, src, world, group):
    if world == 1:
        dst[0].copy_(src)
        return
    if dst.is_contiguous():
        dist.all_gather_into_tensor(dst.view(-1), src.contiguous().view(-1), group=group)
    else:
        tmp = torch.empty((world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(tmp.view(-1), src.contiguous().view(-1), group=group)
        dst.copy_(tmp)


def _scatter_rows(out, staged, plan, r0, r1):
    """staged[r, j] = row r0+j of rank r's block -> its global position."""
    for r in range(plan.world):
        s0, s1 = plan.bounds(r)
        lo, hi = min(r0, s1 - s0), min(r1, s1 - s0)
        if hi > lo:
            out[s0 + lo:s0 + hi].copy_(staged[r, lo - r0:hi - r0])


def enhance_sharded(x_local, u_local, plan, rank, M, gamma, n_colloc=12, *, global_domain,
                    rhs=None, bc=(0.0, 0.0), group=None, chunks=4, gather=True, out=None):
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
    pad = plan.max_size
    W_buf = torch.empty((pad, M), dtype=torch.float64, device=dev)
    status = torch.empty((n_loc,), dtype=torch.int32, device=dev)
    kw = {} if rhs is None else {"rhs": rhs}

    def compute(lo, hi, dst):
        ops.enhance(x_local[lo:hi + 1], u_local[lo:hi + 1], M, gamma, n_colloc,
                    elem_offset=s0 + lo, ne_global=plan.ne, global_domain=global_domain, bc=bc,
                    out=dst, status=status[lo:hi], **kw)

    if not gather:
        if n_loc:
            compute(0, n_loc, W_buf[:n_loc])
        return W_buf[:n_loc], status, None
    if out is None:
        out = torch.empty((plan.ne, M), dtype=torch.float64, device=dev)
    if pad > n_loc:
        W_buf[n_loc:].zero_()      # the collective never ships uninitialised memory
    Wg = allgather_rows(W_buf, plan, rank, group=group, chunks=chunks, out=out,
                        compute_chunk=compute)
    return W_buf[:n_loc], status, Wg
