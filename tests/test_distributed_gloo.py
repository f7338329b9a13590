"""CPU, world_size 2, gloo: the N > 1 path -- shard plan, boundary flags through
elem_offset / ne_global, and the chunked all-gather that stitches W.  The per-shard
compute is the oracle here (no GPU in this container); the stitching code is the
product's (hybrid_fem_lssvr_amd.distributed), backend-agnostic by construction."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import lssvr_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ne, M, n, chunks, q, algo="collective"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hybrid_fem_lssvr_amd.distributed import ShardPlan, allgather_rows
        nodes = np.linspace(-1, 1, ne + 1)
        values = np.cos(2 * nodes)                       # non-zero at the ends on purpose
        plan = ShardPlan(ne, world)
        s0, s1 = plan.bounds(rank)
        sl = plan.node_slice(rank)

        def compute(lo, hi, dst):
            # stand-in for ops.enhance(x[lo:hi+1], u[lo:hi+1], elem_offset=s0+lo, ne_global=ne)
            W = np.zeros((hi - lo, M))
            xs, us = nodes[sl][lo:hi + 1], values[sl][lo:hi + 1]
            for k in range(hi - lo):
                gl, gr = orc.boundary_values(s0 + lo + k, ne, xs[k], xs[k + 1], us[k], us[k + 1], (-1.0, 1.0))
                W[k] = orc.solve_bc_eliminated(orc.element_system(xs[k], xs[k + 1], gl, gr, M, 1e4, n))
            dst.copy_(torch.from_numpy(W))

        local = torch.zeros((plan.max_size, M), dtype=torch.float64)
        Wg = allgather_rows(local, plan, rank, chunks=chunks, compute_chunk=compute, algo=algo)
        q.put((rank, Wg.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ne,chunks,algo", [(11, 1, "collective"), (12, 3, "collective"), (7, 4, "collective"),
                                            (11, 2, "pairs"), (7, 4, "pairs")])
def test_two_rank_gloo_stitch(ne, chunks, algo):
    M, n, world = 6, 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ne, M, n, chunks, q, algo)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    nodes = np.linspace(-1, 1, ne + 1)
    Wref, _ = orc.enhance_all(nodes, np.cos(2 * nodes), M, 1e4, n, global_domain=(-1.0, 1.0))
    for r in range(world):
        assert np.array_equal(got[r], Wref), f"rank {r}"       # every rank holds the global W
    # only the two global-boundary elements saw the Dirichlet value (Dual.py:150-151)
    assert abs(orc.clenshaw(-1.0, Wref[0])) < 1e-13
    assert abs(orc.clenshaw(1.0, Wref[0]) - np.cos(2 * nodes[1])) < 1e-12


def _count_worker(rank, world, port, ne, C, chunks, algo, in_place, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hybrid_fem_lssvr_amd.distributed import ShardPlan, allgather_rows
        plan = ShardPlan(ne, world)
        s0, s1 = plan.bounds(rank)

        def compute(lo, hi, dst):
            dst.copy_((torch.arange(s0 + lo, s0 + hi, dtype=torch.float64)[:, None] * 10.0
                       + torch.arange(C, dtype=torch.float64)[None, :]))

        stats = {}
        if in_place:
            out = torch.full((ne, C), -1.0, dtype=torch.float64)
            Wg = allgather_rows(None, plan, rank, chunks=chunks, out=out, compute_chunk=compute, algo=algo,
                                stats=stats)
        else:
            local = torch.zeros((plan.size(rank), C), dtype=torch.float64)
            compute(0, s1 - s0, local)
            Wg = allgather_rows(local, plan, rank, chunks=chunks, algo=algo, stats=stats)
        q.put((rank, Wg.numpy(), stats))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ne,chunks,algo,in_place", [
    (2, 12, 1, "collective", True), (2, 12, 1, "collective", False), (2, 12, 3, "collective", True),
    (3, 11, 1, "collective", True), (3, 11, 2, "collective", False),
    (2, 12, 1, "pairs", True), (3, 11, 3, "pairs", True), (3, 11, 2, "pairs", False)])
def test_allgather_rows_lands_blocks_in_place(world, ne, chunks, algo, in_place):
    """Round-3 review: the receive side staged every chunk and then issued `world` copy kernels per chunk.
    Now: the direct exchange and the unchunked equal-shard all-gather receive straight into the global
    array (no staging, no copy of received bytes; zero copies at all when the shard is computed in place),
    and the staged collective moves a chunk with ONE copy.  Counted through the `stats` hook."""
    C = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_count_worker, args=(r, world, port, ne, C, chunks, algo, in_place, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, W, st = q.get(timeout=120)
        got[r] = (W, st)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = np.arange(ne, dtype=np.float64)[:, None] * 10.0 + np.arange(C, dtype=np.float64)[None, :]
    from hybrid_fem_lssvr_amd.distributed import ShardPlan
    plan = ShardPlan(ne, world)
    for r in range(world):
        W, st = got[r]
        assert np.array_equal(W, want), (r, st)
        own_bytes = plan.size(r) * C * 8
        equal = ne % world == 0
        nchunks = min(chunks, plan.max_size)
        if algo == "pairs":
            assert st["staged_bytes"] == 0 and st["collectives"] == 0 and st["p2p_ops"] > 0
            assert st["bytes_copied"] == (0 if in_place else own_bytes)          # never a received byte
            assert st["copy_calls"] <= (0 if in_place else nchunks)
        elif equal and chunks == 1:
            assert st["staged_bytes"] == 0 and st["collectives"] == 1
            assert st["bytes_copied"] == (0 if in_place else own_bytes)
        else:
            assert st["collectives"] == nchunks and st["staged_bytes"] > 0
            # ONE move per chunk on the receive side (round 3: `world` per chunk) + at most one staging copy of
            # this rank's own rows per chunk (ragged or computed-in-place shards)
            assert st["copy_calls"] <= 2 * nchunks
            assert st["bytes_copied"] <= ne * C * 8 + own_bytes


def test_single_rank_allgather_is_identity():
    from hybrid_fem_lssvr_amd.distributed import ShardPlan, allgather_rows
    plan = ShardPlan(9, 1)
    local = torch.arange(27, dtype=torch.float64).reshape(9, 3)
    out = allgather_rows(local, plan, 0, chunks=2)
    assert torch.equal(out, local)


def _flat_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hybrid_fem_lssvr_amd.distributed import ALLGATHER_ALGOS, allgather_flat
        src = torch.arange(n, dtype=torch.float64) + 1000.0 * rank
        res = {}
        for algo in ALLGATHER_ALGOS:
            out = torch.full((world * n,), -1.0, dtype=torch.float64)
            allgather_flat(out, src, rank, world, algo=algo)
            res[algo] = out.numpy().copy()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_allgather_flat_algorithms_agree(world):
    """The direct all-pairs exchange (one peer per xGMI link on MI355X) and the backend's
    all-gather deliver the same rank-major array on every rank."""
    n = 37
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = np.concatenate([np.arange(n) + 1000.0 * r for r in range(world)])
    for r in range(world):
        for algo, arr in got[r].items():
            assert np.array_equal(arr, want), (r, algo)


def test_bench_refuses_wrong_rank_count():
    """bench.py --gpus N must never run another rank count: WORLD_SIZE != N exits non-zero before
    anything touches a GPU (and WORLD_SIZE unset makes it spawn its own N ranks -- covered by the
    one-GPU gloo rehearsal in tests/test_gpu_sharded.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "refusing" in r.stderr and r.stdout == ""


def test_stitch_traffic_model_matches_design():
    """The byte counts bench.py reports for the stitch and DESIGN.md section 8's predictions come from
    one function: BASELINE config 3 on 8 ranks, u at 2 / 1 points per element and W."""
    from hybrid_fem_lssvr_amd.distributed import ShardPlan, stitch_traffic_model
    ne, world = 10000008, 8
    assert ShardPlan(ne, world).max_size == 1250001
    m2 = stitch_traffic_model(ne, world, 16)
    assert m2["bytes_received_per_rank_per_step"] == 7 * 1250001 * 16 == 140000112      # "140 MB"
    assert abs(m2["direct_all_pairs_floor_s"] - 20000016 / 76.8e9) < 1e-12              # 0.26 ms at 100 %, 0.33 ms at 80 %
    assert abs(m2["ring_floor_s"] - 7 * 20000016 / 76.8e9) < 1e-12                      # 1.8 ms
    assert abs(m2["inbound_GBps_at_direct_floor"] - 7 * 76.8) < 1e-6                    # 538 GB/s inbound
    m1 = stitch_traffic_model(ne, world, 8)
    assert m1["bytes_received_per_rank_per_step"] * 2 == m2["bytes_received_per_rank_per_step"]
    mw = stitch_traffic_model(ne, world, 72)
    assert mw["bytes_received_per_rank_per_step"] == 7 * 1250001 * 72                   # "630 MB"
    # 6x the N = 1 line (1e10 el/s) = 6e10 el/s needs ne / 6e10 s per step: the 2-point stitch cannot
    # (0.26 ms floor > 0.167 ms), the 1-point stitch can (0.13 ms) -- DESIGN.md section 8
    assert m2["direct_all_pairs_floor_s"] > ne / 6e10 > m1["direct_all_pairs_floor_s"]
    assert stitch_traffic_model(1000, 1, 16)["bytes_received_per_rank_per_step"] == 0


def _bytes_worker(rank, world, port, ne, width, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hybrid_fem_lssvr_amd.distributed import ShardPlan, allgather_flat, stitch_traffic_model
        plan = ShardPlan(ne, world)
        pad = plan.max_size
        # exactly the buffers bench.py::timed_stitch gathers: equal padded blocks, rank-major
        send = torch.full((pad * width,), float(rank + 1), dtype=torch.float64)
        res = {}
        for algo in ("collective", "pairs"):
            recv = torch.zeros(world * pad * width, dtype=torch.float64)
            allgather_flat(recv, send, rank, world, algo=algo)
            foreign = int((recv != float(rank + 1)).sum().item()) * 8        # bytes that came from other ranks
            res[algo] = foreign
        q.put((rank, res, stitch_traffic_model(ne, world, 8 * width)["bytes_received_per_rank_per_step"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ne,width", [(2, 1001, 2), (3, 100, 1), (2, 64, 9)])
def test_stitch_bytes_equal_the_model(world, ne, width):
    """What a rank receives in one stitched step (every other rank's padded block) is what
    stitch_traffic_model -- and therefore the bench line's bytes_received_per_rank_per_step -- says."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bytes_worker, args=(r, world, port, ne, width, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, res, model in got:
        assert res["collective"] == res["pairs"] == model, (rank, res, model)


def test_bench_self_launch_ends_siblings_on_first_failure(tmp_path):
    """bench.py's own launcher polls its ranks: when one exits non-zero the others are terminated
    instead of waiting in a collective for a peer that is gone."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fake_rank.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(3)\n"
                      "time.sleep(120)\n")
    sys.path.insert(0, root)
    import importlib
    bench = importlib.import_module("bench")
    old = bench.__file__
    try:
        bench.__file__ = str(script)           # self_launch starts `python <bench.__file__> args` per rank
        t0 = time.time()
        rc = bench.self_launch(3, [])
    finally:
        bench.__file__ = old
    assert rc == 3 and time.time() - t0 < 30.0
