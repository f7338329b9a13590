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


def _worker(rank, world, port, ne, M, n, q, backend="gloo", algo="collective"):
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
                                      chunks=2, algo=algo)
        torch.cuda.synchronize()
        q.put((rank, u.cpu().numpy(), Wg.cpu().numpy(), int(st.sum().item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,algo", [(1, "collective"), (2, "collective"), (3, "collective"), (2, "pairs"), (3, "pairs")])
def test_sharded_solve_matches_single_rank(dev, world, algo):
    import hybrid_fem_lssvr_amd as pkg
    ne, M, n = 10001, 9, 16
    ref = pkg.FEMLSSVRPrimalSolver(ne + 1, lssvr_M=M, lssvr_gamma=1e4, n_colloc=n, fem_solver="flux")
    ref.solve()
    W_ref = ref.enhanced.W.cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ne, M, n, q, "gloo", algo)) for r in range(world)]
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


@pytest.mark.parametrize("algo", ["collective", "pairs"])
def test_sharded_pipeline_over_rccl_single_rank(dev, algo):
    """The same pipeline with backend "nccl" (= RCCL) and ONE rank: what a one-GPU box can
    exercise of the production backend -- communicator bring-up, the 24-byte all-gather of the
    flux aggregates and the chunk-overlapped stitch of W on a side stream, with the backend's
    all-gather and with the direct all-pairs form (`algo="pairs"`, driven through enhance_sharded)."""
    import hybrid_fem_lssvr_amd as pkg
    ne, M, n = 10001, 9, 16
    ref = pkg.FEMLSSVRPrimalSolver(ne + 1, lssvr_M=M, lssvr_gamma=1e4, n_colloc=n, fem_solver="flux")
    ref.solve()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), ne, M, n, q, "nccl", algo))
    p.start()
    r, u, Wg, nbad = q.get(timeout=300)
    p.join(120)
    assert p.exitcode == 0 and nbad == 0
    assert np.max(np.abs(u - ref.fem_values)) <= 1e-14
    assert np.max(np.abs(Wg - ref.enhanced.W.cpu().numpy())) <= 1e-13


def _run_bench(extra_args, env_extra, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra_args, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]          # exactly ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_self_launches_its_ranks(dev):
    """`python bench.py --gpus 2` with no launcher must start two ranks itself (rehearsed over
    gloo on the one GPU of this box: RCCL refuses two ranks on one device), run BASELINE config
    3's strong-scaling mode (here on a smaller total) and report the stitched rate with both
    all-gather algorithms."""
    out = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--elements", "200008"],
                     {"LSSVR_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["ranks_in_process_group"] == 2
    assert out["scaling"] == "strong" and out["config"]["elements_total"] == 200008
    assert out["config"]["elements_per_gpu"] == 100004 and out["config"]["fallback_elements"] == 0
    assert out["value"] > 0 and out["value_with_allgather"] > 0
    for key, nbytes in (("stitch_u", 16), ("stitch_W", 72)):
        st = out[key]
        assert st["picked"] in ("collective", "pairs")
        for algo in ("collective", "pairs"):
            a = st["algorithms"][algo]
            assert "error" not in a, a
            assert a["own_block_intact"] and a["bytes_per_element"] == nbytes
            assert a["bytes_received_per_rank_per_step"] == 100004 * nbytes
    assert out["weak_scaling"]["value"] > 0 and out["one_rank_same_workload"]["value"] > 0
    # the N > 1 line is judged like the N = 1 line: aggregate roofline (all shards over the slowest rank's
    # step, against N x the one-GPU peaks) and the CPU baseline timed by rank 0 in the same run
    rf = out["roofline"]
    assert rf["peak"] == 2 * 78.6 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1.5
    assert abs(rf["kernel_us_avg"] - out["ms_per_step"] * 1e3) < 1e-6 * rf["kernel_us_avg"]
    assert abs(rf["achieved"] - rf["flops_per_element"] * 200008 / (rf["kernel_us_avg"] * 1e-6) / 1e12) < 1e-9 * rf["achieved"]
    assert out["roofline_hbm"]["peak"] == 2 * 8000.0 and 0 < out["roofline_hbm"]["frac"] < 1
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] >= 1
    assert out["timed_brackets"] >= 10 and out["ms_per_step_min"] <= out["ms_per_step"] <= out["ms_per_step_max"]


def test_bench_single_rank_line(dev):
    """The N = 1 line the driver records: metric / config of BASELINE config 2, roofline and
    cpu_baseline objects, event-timed value consistent with the host clock."""
    out = _run_bench(["--steps", "50", "--warmup", "5"], {})
    assert out["n_gpus"] == 1 and out["config"]["elements_total"] == 100008
    assert out["dtype"] == "f64" and out["unit"] == "elements/s" and out["vs_baseline"] is None
    rf = out["roofline"]
    assert rf["bound"] == "fp64-valu" and 0 < rf["frac"] < 1.5 and rf["kernel_us_avg"] > 1
    assert abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-12
    # the roofline prices the kernel the timed region launches with the region's own per-step time
    assert rf["kernel"].startswith("step_small_kernel<M=9>")
    assert abs(rf["kernel_us_avg"] - out["ms_per_step"] * 1e3) < 1e-6 * rf["kernel_us_avg"]
    assert rf["enhancement_only"]["kernel_us_in_sequence_avg"] > 1
    assert out["timed_brackets"] >= 10 and out["ms_per_step_min"] <= out["ms_per_step"] <= out["ms_per_step_max"]
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
    assert out["ms_per_step"] <= out["host_wall_ms_per_step"] * 1.05
    assert out["accuracy"]["rel_l2_vs_float64_kkt_oracle"] < 1e-13
    assert out["config"]["fallback_elements"] == 0
    # BASELINE configs 4 and 5 ride along in the default line (compact objects, same contract)
    for key, bound in (("config4", "fp64-valu"), ("config5", "hbm")):
        c = out[key]
        assert "error" not in c, c
        assert c["value"] >= 1.0e6 and c["timed_brackets"] >= 10
        assert c["roofline"]["bound"] == bound and 0 < c["roofline"]["frac"] < 1.5
        assert abs(c["roofline"]["kernel_us_avg"] - c["ms_per_step"] * 1e3) < 1e-6 * c["roofline"]["kernel_us_avg"]
        assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["value"] > 0
        assert c["accuracy"]["rel_l2_vs_60_digit_minimiser"] < 1e-13 and c["config"]["fallback_elements"] == 0
    assert out["config4"]["config"]["elements_total"] == 100008 and out["config5"]["config"]["elements_total"] == 1000008
