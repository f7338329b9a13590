"""GPU: the sharded solve-then-enhance pipeline (sharded P1 flux solve with one 24-byte
all-gather, rank-local enhancement, stitch) -- two ranks rehearsed on ONE GPU over gloo
(RCCL refuses two ranks on the same device), compared with the single-rank result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ne, M, n, q, backend="gloo"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hybrid_fem_lssvr_amd.distributed import ShardPlan, solve_sharded
        dev = torch.device("cuda:0")
        nodes = np.linspace(-1.0, 1.0, ne + 1)
        plan = ShardPlan(ne, world)
        s0, s1 = plan.bounds(rank)
        lo = s0 - 1 if s0 > 0 else s0
        x_ext = torch.as_tensor(nodes[lo:s1 + 1].copy(), device=dev)
        u, Wl, st, Wg = solve_sharded(x_ext, plan, rank, M, 1e4, n, global_domain=(-1.0, 1.0),
                                      chunks=2)
        torch.cuda.synchronize()
        q.put((rank, u.cpu().numpy(), Wg.cpu().numpy(), int(st.sum().item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_sharded_solve_matches_single_rank(dev, world):
    import hybrid_fem_lssvr_amd as pkg
    ne, M, n = 10001, 9, 16
    ref = pkg.FEMLSSVRPrimalSolver(ne + 1, lssvr_M=M, lssvr_gamma=1e4, n_colloc=n, fem_solver="flux")
    ref.solve()
    W_ref = ref.enhanced.W.cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ne, M, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, u, Wg, nbad = q.get(timeout=300)
        got[r] = (u, Wg, nbad)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    from hybrid_fem_lssvr_amd import ShardPlan
    plan = ShardPlan(ne, world)
    for r in range(world):
        u, Wg, nbad = got[r]
        s0, s1 = plan.bounds(r)
        assert nbad == 0
        # nodal values of the shard: same scan, different association of the partial sums
        assert np.max(np.abs(u - ref.fem_values[s0:s1 + 1])) <= 1e-14
        # every rank holds the stitched global W; it differs from the single-rank run only
        # through those last-bit differences in the nodal values
        assert np.max(np.abs(Wg - W_ref)) <= 1e-13
    assert np.array_equal(got[0][1], got[world - 1][1])


def test_sharded_pipeline_over_rccl_single_rank(dev):
    """The same pipeline with backend "nccl" (= RCCL) and ONE rank: what a one-GPU box can
    exercise of the production backend -- communicator bring-up, the 24-byte all-gather of the
    flux aggregates and the chunk-overlapped all-gather of W on a side stream."""
    import hybrid_fem_lssvr_amd as pkg
    ne, M, n = 10001, 9, 16
    ref = pkg.FEMLSSVRPrimalSolver(ne + 1, lssvr_M=M, lssvr_gamma=1e4, n_colloc=n, fem_solver="flux")
    ref.solve()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), ne, M, n, q, "nccl"))
    p.start()
    r, u, Wg, nbad = q.get(timeout=300)
    p.join(120)
    assert p.exitcode == 0 and nbad == 0
    assert np.max(np.abs(u - ref.fem_values)) <= 1e-14
    assert np.max(np.abs(Wg - ref.enhanced.W.cpu().numpy())) <= 1e-13
