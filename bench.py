#!/usr/bin/env python3
"""Benchmark of the per-element LSSVR enhancement hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON
line on rank 0.  For N > 1 it is launched through ``torch.distributed.run`` with one
rank per GPU (RCCL).

Metric  : LSSVR-enhanced elements/s (BASELINE.json ``metric``).
Step    : one pass of the hot path over one batch of elements that is already
          resident in HBM: element-local P1 stiffness/load assembly (Dual.py:117-128)
          + the per-element Gram + solve (Dual.py:139-169), one fused launch per rank.
          The path shards with no data-path collective (the metric is
          ne / (t_assemble_local + t_enhance)); the RCCL all-gather that stitches the
          coefficient rows afterwards is timed in a second region and reported under
          "stitch" (t_allgather), never folded into ``value``.
Workload: BASELINE config 2 -- degree 8 (M = 9), 16 collocation points, gamma = 1e4,
          1e5 P1 elements per GPU -- on the wide domain [-N, N] with h = 1/12
          (100 008 elements per GPU; SURVEY.md finding 5: on [-1,1] the reference's own
          SLSQP loop stops converging above ~5e4 elements, so the CPU baseline could
          not be timed on it; kernel cost does not depend on the domain).
          ``--domain narrow`` runs exactly 1e5 elements on [-1,1] instead.
Scaling : weak (per-GPU elements fixed).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M_DEG8, N_COLLOC, GAMMA = 9, 16, 1.0e4
NE_WIDE, HALF_WIDE = 100008, 4167.0        # h = 1/12 exactly
NE_NARROW = 100000
FP64_PEAK_TFLOPS = 78.6                    # MI355X vector = matrix FP64 peak (SURVEY.md 8(d):
                                           # 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz); probe below
HBM_PEAK_GBS = 8000.0


def algorithmic_flops(M, n):
    """SURVEY.md 8(d), primal form: Gram M(M+1)n + A^T f 2Mn + KKT-LU 2/3 (M+2)^3 + 2 (M+2)^2."""
    return M * (M + 1) * n + 2 * M * n + (2.0 / 3.0) * (M + 2) ** 3 + 2 * (M + 2) ** 2


def algorithmic_bytes(M):
    """node coordinate 8 B + nodal value 8 B + 8 M B of coefficients."""
    return 16 + 8 * M


# ----------------------------------------------------------------------------------------
# CPU baseline: the reference's own per-element SLSQP loop (oracle restatement, "port")
# ----------------------------------------------------------------------------------------
def _cpu_worker(args):
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = "1"
    import numpy as np
    from oracle import lssvr_oracle as orc
    nodes, values, elems, ne, gd, seed = args
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    ok = 0
    for i in elems:
        j = int(i)
        _, s = orc.slsqp_element(orc.poisson_rhs, nodes[j], nodes[j + 1], values[j], values[j + 1],
                                 M_DEG8, GAMMA, N_COLLOC, left=(j == 0), right=(j == ne - 1),
                                 global_domain=gd, rng=rng)
        ok += int(s)
    return len(elems), ok, time.perf_counter() - t0


def cpu_baseline(nodes_host, values_host, gd, per_core=16):
    """Times the SLSQP loop on a bounded sample of the same mesh with every host core the
    box gives us (one process per core, like N copies of the single-threaded reference)."""
    import multiprocessing as mp
    import numpy as np
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    ne = len(nodes_host) - 1
    sample = np.linspace(0, ne - 1, cores * per_core).astype(np.int64)
    parts = np.array_split(sample, cores)
    jobs = [(nodes_host, values_host, p, ne, gd, 1000 + k) for k, p in enumerate(parts)]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    conv = sum(r[1] for r in res)
    busy = sum(r[2] for r in res)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": done / wall,
        "unit": "elements/s",
        "cores": cores,
        "cpu_model": model,
        "kind": "port",
        "sample": f"{done} elements evenly spaced through the same {ne}-element mesh, "
                  f"per-element scipy SLSQP loop (oracle/lssvr_oracle.py::slsqp_element = "
                  f"Dual.py:20-98), one process per core; {conv}/{done} converged",
        "single_core_value": done / busy,
    }


# ----------------------------------------------------------------------------------------
def main():
    # Only the result line may reach stdout (the driver reads ONE JSON line there): RCCL prints
    # a version banner to stdout when the process group comes up, libraries may print warnings.
    # Everything written to fd 1 from here on goes to stderr; the JSON is written to the saved fd.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.write(real_stdout, (line + "\n").encode())

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--domain", choices=["wide", "narrow"], default="wide")
    ap.add_argument("--degree", type=int, default=8)
    ap.add_argument("--colloc", type=int, default=N_COLLOC)
    ap.add_argument("--elements", type=int, default=0, help="elements per GPU (0 = config default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the stitch measurement")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from hybrid_fem_lssvr_amd import ops
    from hybrid_fem_lssvr_amd.distributed import ShardPlan

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    M = args.degree + 1
    n = args.colloc
    if args.domain == "wide":
        ne_loc = args.elements or NE_WIDE
        half = ne_loc * world / 24.0                      # h = 1/12
        lo, hi = -half, half
    else:
        ne_loc = args.elements or NE_NARROW
        lo, hi = -1.0, 1.0
    plan = ShardPlan(ne_loc * world, world)
    ne_glob = plan.ne
    s0, s1 = plan.bounds(rank)

    # synthetic input, resident in HBM before the timed region ----------------------------
    # nodes of this rank's shard exactly as np.linspace(lo, hi, ne_glob+1) gives them
    step = (hi - lo) / ne_glob
    idx = np.arange(s0, s1 + 1, dtype=np.float64)
    nodes_h = idx * step + lo
    if s1 == ne_glob:
        nodes_h[-1] = hi
    values_h = np.sin(np.pi * nodes_h)                    # nodal values of the exact solution
    if s0 == 0:
        values_h[0] = 0.0
    if s1 == ne_glob:
        values_h[-1] = 0.0
    gd = (lo, hi)

    # CPU baseline first: its worker processes are forked before this process touches the
    # GPU (a forked child of a GPU-initialised process is best avoided on this pool)
    cpu_res = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and M == M_DEG8 and n == N_COLLOC:
        cpu_res = cpu_baseline(nodes_h, values_h, gd)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one rank per GPU; LSSVR_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse
    # the N > 1 code path on a single-GPU box (RCCL refuses duplicate devices)
    backend = os.environ.get("LSSVR_BENCH_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # LSSVR_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, stitch regions) with a
    # single rank -- lets a one-GPU box exercise RCCL initialisation and the collectives' stream logic
    use_dist = world > 1 or bool(os.environ.get("LSSVR_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    x = torch.as_tensor(nodes_h, device=dev)
    u = torch.as_tensor(values_h, device=dev)
    W = torch.empty((plan.max_size, M), dtype=torch.float64, device=dev)
    status = torch.empty(ne_loc, dtype=torch.int32, device=dev)
    bands = ops.p1_assemble(x, 2)
    Wg = torch.empty((ne_glob, M), dtype=torch.float64, device=dev) if use_dist and not args.no_gather else None
    gather = use_dist and not args.no_gather

    # N = 1: the whole step (assembly + enhancement) is one fused launch bound once
    fused = ops.StepPlan(x, u, M, GAMMA, n, elem_offset=s0, ne_global=ne_glob, global_domain=gd,
                         bands=bands, out=W[:ne_loc], status=status)
    st = torch.cuda.current_stream().cuda_stream

    def one_step(i=None):
        fused.launch(st)

    bar_t = torch.zeros(1, dtype=torch.float32, device=dev) if use_dist else None

    def barrier():
        # an all-reduce of one preallocated element; the torch.cuda.synchronize() that follows every
        # call completes it (dist.barrier() allocates and synchronises by itself: ~10x the cost)
        if use_dist:
            if backend == "nccl":
                dist.all_reduce(bar_t)
            else:
                dist.barrier()

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    barrier()                      # (the first collective also brings the communicator up)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_fallback = int(status.sum().item())

    # stitch: every rank ends up with the global W.  Second timed region, same K steps:
    # step i computes into W[i%2] on the main stream while the all-gather of step i-1 runs on
    # a side stream (double-buffered, equal shards: the gather lands directly in Wg, no copies)
    stitch = None
    if gather:
        try:
            Wb = [W[:ne_loc], torch.empty((ne_loc, M), dtype=torch.float64, device=dev)]
            Wgb = [Wg, torch.empty_like(Wg)]
            plans = [ops.StepPlan(x, u, M, GAMMA, n, elem_offset=s0, ne_global=ne_glob, global_domain=gd,
                                  bands=bands, out=Wb[k], status=status) for k in range(2)]
            comm = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            done = [None, None]

            def stitched_step(i):
                k = i & 1
                if done[k] is not None:
                    main.wait_event(done[k])            # the gather that read W[k] has finished
                plans[k].launch(main.cuda_stream)
                ready = torch.cuda.Event()
                ready.record(main)
                comm.wait_event(ready)
                with torch.cuda.stream(comm):
                    dist.all_gather_into_tensor(Wgb[k].view(-1), Wb[k].view(-1))
                    done[k] = torch.cuda.Event()
                    done[k].record(comm)

            for i in range(max(args.warmup, 2)):
                stitched_step(i)
            torch.cuda.synchronize()
            barrier()
            t1 = time.perf_counter()
            for i in range(args.steps):
                stitched_step(i)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t1
            t = torch.tensor([el2], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
            # correctness of the stitch: rank r's block of the gathered array equals what rank r holds
            mine = Wgb[(args.steps - 1) & 1][s0:s1]
            same = bool(torch.equal(mine, Wb[(args.steps - 1) & 1]))
            stitch = {
                "what": "RCCL all-gather of W over xGMI, overlapped with the next step's kernel",
                "value_with_allgather": ne_glob * args.steps / el2,
                "ms_per_step": el2 / args.steps * 1e3,
                "bytes_received_per_rank_per_step": (world - 1) * ne_loc * M * 8,
                "recv_GBps_per_rank": (world - 1) * ne_loc * M * 8 / (el2 / args.steps) / 1e9,
                "own_block_intact": same,
            }
        except Exception as exc:  # pragma: no cover
            stitch = {"error": repr(exc)}

    # the same stitch for the sampled solution instead of the coefficients (north_star: "all-gather
    # ... to stitch the global enhanced solution vector"): every rank evaluates its own shard at
    # two interior points per element (lssvr_eval, rank-local) and gathers 16 B per element
    # instead of 72; third timed region, same K steps, same double buffering
    stitch_u = None
    if gather:
        try:
            h_el = x[1:] - x[:-1]
            xq = torch.stack([x[:-1] + 0.25 * h_el, x[:-1] + 0.75 * h_el], dim=1).reshape(-1).contiguous()
            P_loc = xq.numel()
            ub = [torch.empty(P_loc, dtype=torch.float64, device=dev) for _ in range(2)]
            ug = [torch.empty(P_loc * world, dtype=torch.float64, device=dev) for _ in range(2)]
            done_u = [None, None]

            def stitched_u_step(i):
                k = i & 1
                if done_u[k] is not None:
                    main.wait_event(done_u[k])
                plans[k].launch(main.cuda_stream)
                ops.evaluate(x, Wb[k], xq, want_elem=False, out=ub[k], stream=main.cuda_stream)
                ready = torch.cuda.Event()
                ready.record(main)
                comm.wait_event(ready)
                with torch.cuda.stream(comm):
                    dist.all_gather_into_tensor(ug[k], ub[k])
                    done_u[k] = torch.cuda.Event()
                    done_u[k].record(comm)

            for i in range(max(args.warmup, 2)):
                stitched_u_step(i)
            torch.cuda.synchronize()
            barrier()
            t2 = time.perf_counter()
            for i in range(args.steps):
                stitched_u_step(i)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            el3 = time.perf_counter() - t2
            t = torch.tensor([el3], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el3 = float(t.item())
            kk = (args.steps - 1) & 1
            own = bool(torch.equal(ug[kk][rank * P_loc:(rank + 1) * P_loc], ub[kk]))
            stitch_u = {
                "what": "rank-local lssvr_eval at 2 interior points per element + RCCL all-gather of u "
                        "(16 B per element), overlapped with the next step's kernels",
                "value_with_allgather": ne_glob * args.steps / el3,
                "ms_per_step": el3 / args.steps * 1e3,
                "points_total": P_loc * world,
                "bytes_received_per_rank_per_step": (world - 1) * P_loc * 8,
                "recv_GBps_per_rank": (world - 1) * P_loc * 8 / (el3 / args.steps) / 1e9,
                "own_block_intact": own,
            }
        except Exception as exc:  # pragma: no cover
            stitch_u = {"error": repr(exc)}

    # dominant kernel (the per-element enhancement): launch duration from HIP events that
    # hipExtLaunchKernelGGL stamps with the dispatch's own begin / end times -- the quantity
    # rocprofv3 --kernel-trace reports -- over the same launch as in the timed region,
    # right after it (a plain hipEventRecord pair adds ~4 us of dispatch latency)
    k_s = sorted(ops.enhance_profiled(x, u, M, GAMMA, n, elem_offset=s0, ne_global=ne_glob,
                                      global_domain=gd, out=W[:ne_loc], status=status)
                 for _ in range(min(args.steps, 100)))
    k_avg = sum(k_s) / len(k_s)
    k_med = k_s[len(k_s) // 2]

    # the same step on exactly 1e5 elements of [-1, 1] (BASELINE.json's wording of config 2),
    # where the CPU baseline cannot be timed (SURVEY.md finding 5): shows that the kernel cost
    # does not depend on the domain
    narrow = None
    if world == 1 and args.domain == "wide" and not args.elements:
        nn = NE_NARROW
        xn_h = np.arange(nn + 1, dtype=np.float64) * (2.0 / nn) - 1.0
        xn_h[-1] = 1.0
        un_h = np.sin(np.pi * xn_h)
        un_h[0] = un_h[-1] = 0.0
        xn, un = torch.as_tensor(xn_h, device=dev), torch.as_tensor(un_h, device=dev)
        pl = ops.StepPlan(xn, un, M, GAMMA, n, global_domain=(-1.0, 1.0))
        for _ in range(args.warmup):
            pl.launch(st)
        torch.cuda.synchronize()
        tn = time.perf_counter()
        for _ in range(args.steps):
            pl.launch(st)
        torch.cuda.synchronize()
        tn = time.perf_counter() - tn
        narrow = {"workload": "%d elements on [-1, 1], same step" % nn,
                  "value": nn * args.steps / tn, "ms_per_step": tn / args.steps * 1e3,
                  "fallback_elements": int(pl.status.sum().item())}

    # the same K steps issued round-robin on two HIP streams with separate output buffers: at
    # 1e5 elements one launch fills only ~60 % of the chip's wave slots, so independent batches
    # overlap.  Reported beside `value` (which stays the strictly sequential single-stream rate).
    pipelined = None
    if world == 1 and rank == 0:
        try:
            nstream = 2
            streams = [torch.cuda.Stream(device=dev) for _ in range(nstream)]
            Wp = [torch.empty((ne_loc, M), dtype=torch.float64, device=dev) for _ in range(nstream)]
            stp = [torch.empty(ne_loc, dtype=torch.int32, device=dev) for _ in range(nstream)]
            bp = [ops.p1_assemble(x, 2) for _ in range(nstream)]
            pls = [ops.StepPlan(x, u, M, GAMMA, n, elem_offset=s0, ne_global=ne_glob, global_domain=gd,
                                bands=bp[k], out=Wp[k], status=stp[k]) for k in range(nstream)]
            torch.cuda.synchronize()
            for i in range(args.warmup):
                pls[i % nstream].launch(streams[i % nstream].cuda_stream)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for i in range(args.steps):
                pls[i % nstream].launch(streams[i % nstream].cuda_stream)
            torch.cuda.synchronize()
            tp = time.perf_counter() - tp
            pipelined = {"what": "same K steps, round-robin on %d streams, separate W buffers" % nstream,
                         "streams": nstream, "value": ne_loc * args.steps / tp, "unit": "elements/s",
                         "ms_per_step": tp / args.steps * 1e3,
                         "results_equal": bool(torch.equal(Wp[0], W[:ne_loc]) and torch.equal(Wp[1], W[:ne_loc]))}
        except Exception as exc:  # pragma: no cover
            pipelined = {"error": repr(exc)}

    # the uniform-mesh shortcut (lssvr_enhance_shared; SURVEY.md 8(d): "reported as a separate line
    # if built"): one shared operator applied per element.  Never part of `value`; its own
    # roofline is HBM (88 B per element against ~8 TB/s).
    shared = None
    if rank == 0 and M <= 33:
        try:
            op = ops.build_shared_operator((hi - lo) / ne_glob, M, GAMMA, n, device=dev)
            Ws = torch.empty((ne_loc, M), dtype=torch.float64, device=dev)
            ts = sorted(ops.enhance_shared(x, u, op, M, n, elem_offset=s0, ne_global=ne_glob,
                                           global_domain=gd, out=Ws, status=status, profiled=True)
                        for _ in range(min(args.steps, 50)))
            t_sh = sum(ts) / len(ts)
            diff = (Ws - W[:ne_loc]).double()
            rel = float((diff.pow(2).sum(1).sqrt() / W[:ne_loc].pow(2).sum(1).sqrt().clamp_min(1e-300)).max().item())
            shared = {
                "what": "lssvr_enhance_shared: uniform mesh, one (n+2) x M operator (built by the general "
                        "kernel) applied per element; same inputs, same W layout",
                "value": ne_loc / t_sh, "unit": "elements/s", "kernel_us_avg": t_sh * 1e6,
                "kernel_us_median": ts[len(ts) // 2] * 1e6,
                "roofline": {"bound": "hbm", "achieved": algorithmic_bytes(M) * ne_loc / t_sh / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": algorithmic_bytes(M) * ne_loc / t_sh / 1e9 / HBM_PEAK_GBS,
                             "bytes_per_element": algorithmic_bytes(M)},
                "max_rel_coef_diff_vs_general_kernel": rel,
            }
        except Exception as exc:  # pragma: no cover
            shared = {"error": repr(exc)}

    # the stages around the hot path, each timed on its own (t_global_solve,
    # t_eval, H2D/D2H are reported separately and never enter `value`)
    stages = None
    if rank == 0:
        def med_us(fn, reps=20):
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                a0 = torch.cuda.Event(enable_timing=True)
                a1 = torch.cuda.Event(enable_timing=True)
                a0.record()
                fn()
                a1.record()
                a1.synchronize()
                ts.append(a0.elapsed_time(a1) * 1e3)
            return sorted(ts)[len(ts) // 2]
        try:
            bl = ops.p1_assemble(x, 2, want_local=True)
            uq_pts = torch.linspace(float(nodes_h[0]), float(nodes_h[-1]), 2 * ne_loc + 1,
                                    dtype=torch.float64, device=dev)
            x_pin = torch.as_tensor(nodes_h).pin_memory()
            W_pin = torch.empty((ne_loc, M), dtype=torch.float64).pin_memory()
            stages = {
                "note": "median of 20, microseconds, event pairs on the launch stream (each includes "
                        "~4 us of event + dispatch latency); this rank's shard",
                "p1_assemble_us": med_us(lambda: ops.p1_assemble(x, 2, out=bl)),
                "dirichlet_solve_bands_us": med_us(lambda: ops.tridiag_dirichlet_solve(bl["diag"], bl["off"], bl["load"])),
                "dirichlet_solve_flux_us": med_us(lambda: ops.p1_flux_solve(bl["kloc"], bl["load"])),
                "enhance_us": med_us(lambda: ops.enhance(x, u, M, GAMMA, n, elem_offset=s0, ne_global=ne_glob,
                                                          global_domain=gd, out=W[:ne_loc], status=status)),
                "evaluate_2ne_points_us": med_us(lambda: ops.evaluate(x, W[:ne_loc], uq_pts, want_elem=False)),
                "h2d_nodes_and_values_us": med_us(lambda: (x.copy_(x_pin, non_blocking=True),
                                                          u.copy_(x_pin, non_blocking=True))),
                "d2h_coefficients_us": med_us(lambda: W_pin.copy_(W[:ne_loc], non_blocking=True)),
            }
            x.copy_(torch.as_tensor(nodes_h))
            u.copy_(torch.as_tensor(values_h))
            fused.launch(st)
            torch.cuda.synchronize()
            hd = stages["h2d_nodes_and_values_us"] + stages["d2h_coefficients_us"]
            stages["pcie_inclusive_elements_per_s"] = ne_loc / ((elapsed / args.steps) + hd * 1e-6)
        except Exception as exc:  # pragma: no cover
            stages = {"error": repr(exc)}

    # accuracy of what was just timed (reported with every timing), rank 0's
    # shard: sampled elements against the float64 KKT oracle (and the 60-digit minimiser when
    # mpmath is present), and the stitched u(x) against sin(pi x) on a probe grid
    accuracy = None
    if rank == 0:
        try:
            from oracle import lssvr_oracle as orc
            from oracle import closed_form_mp as cf
            W_h = W[:ne_loc].cpu().numpy()
            sel = np.unique(np.linspace(0, ne_loc - 1, 9).astype(np.int64))
            Wo = np.array([orc.solve_primal_kkt(orc.element_system(
                nodes_h[i], nodes_h[i + 1],
                *orc.boundary_values(s0 + int(i), ne_glob, nodes_h[i], nodes_h[i + 1], values_h[i],
                                     values_h[i + 1], gd), M, GAMMA, n)) for i in sel])
            accuracy = {"sampled_elements": int(len(sel)),
                        "rel_l2_vs_float64_kkt_oracle": float(orc.rel_l2_coef(W_h[sel], Wo).max())}
            if cf.HAVE_MP:
                tr = np.array([cf.solve_truth(orc.element_system(
                    nodes_h[i], nodes_h[i + 1],
                    *orc.boundary_values(s0 + int(i), ne_glob, nodes_h[i], nodes_h[i + 1], values_h[i],
                                         values_h[i + 1], gd), M, GAMMA, n)) for i in sel[:5]])
                accuracy["rel_l2_vs_60_digit_minimiser"] = float(orc.rel_l2_coef(W_h[sel[:5]], tr).max())
            xq_h = np.linspace(nodes_h[0], nodes_h[-1], 20001)
            norms = ops.eval_error(x, W[:ne_loc], torch.as_tensor(xq_h, device=dev)).cpu().numpy()
            accuracy["rel_l2_vs_sin_pi_x_on_20001_probes"] = float(np.sqrt(norms[0] / norms[1]))
            accuracy["max_abs_err_vs_sin_pi_x"] = float(norms[2])
            accuracy["note"] = ("nodal values are sin(pi x_i) here (device-resident synthetic input), so the "
                                "last figure is the enhancement's own error, not the P1 nodal error")
        except Exception as exc:  # pragma: no cover
            accuracy = {"error": repr(exc)}

    if rank == 0:
        total = ne_glob * args.steps
        flops = algorithmic_flops(M, n)
        byts = algorithmic_bytes(M)
        k_dur = max(k_avg, 1e-9)
        ach_tflops = flops * ne_loc / k_dur / 1e12
        kernel_name = "enhance_small_kernel<M=%d>" % M if M <= 22 else "enhance_large_kernel"
        out = {
            "metric": "LSSVR-enhanced elements/sec, 1D Poisson deg-%d/%d-pt" % (args.degree, n),
            "value": total / elapsed,
            "unit": "elements/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("1D Poisson, %d P1 elements per GPU on [%g, %g] (h = %s), Legendre degree %d "
                             "(M = %d), %d collocation points, gamma = 1e4, f = pi^2 sin(pi x) in-kernel; "
                             "step = element-local P1 assembly + per-element Gram + solve, one fused "
                             "launch per rank, no data-path collective"
                             % (ne_loc, lo, hi, "1/12" if args.domain == "wide" else "2/ne", args.degree,
                                M, n)),
                "elements_per_gpu": ne_loc,
                "elements_total": ne_glob,
                "parallelism": "elements sharded contiguously, %d rank(s)" % world,
                "solver": "primal, BC-eliminated SPD (M-2), Cholesky",
                "fallback_elements": n_fallback,
            },
            "roofline": {
                "bound": "mfma",
                "pipe": "FP64: vector FMA at M <= 22, f64 MFMA (4x4x4 blocks) Gram + DPP-broadcast LDL^T "
                        "above; on gfx950 the FP64 vector and matrix peaks are the same 78.6 TFLOP/s and "
                        "the two share one pipe (DESIGN.md section 3)",
                "kernel": kernel_name,
                "achieved": ach_tflops,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": ach_tflops / FP64_PEAK_TFLOPS,
                "flops_per_element": flops,
                "elements_per_launch": ne_loc,
                "kernel_us_avg": k_dur * 1e6,
                "kernel_us_median": k_med * 1e6,
                "traffic": None,
            },
            "roofline_hbm": {
                "bound": "hbm",
                "achieved": byts * ne_loc / k_dur / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": byts * ne_loc / k_dur / 1e9 / HBM_PEAK_GBS,
                "bytes_per_element": byts,
                "traffic": None,
            },
        }
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                tr = json.load(open(tf)).get("M%d_n%d_ne%d" % (M, n, ne_loc))
                if tr:
                    out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                    out["roofline_hbm"]["traffic"] = tr["hbm_bytes_per_launch"]
            except Exception:
                pass
        try:
            out["roofline"]["fp64_fma_probe_tflops"] = round(ops.fp64_probe(8192, 4096, False), 2)
            if M > 22:
                out["roofline"]["fp64_mfma_4x4x4_probe_tflops"] = round(ops.fp64_probe(8192, 2048, 3), 2)
                out["roofline"]["fp64_mfma_16x16x4_probe_tflops"] = round(ops.fp64_probe(8192, 1024, 1), 2)
        except Exception as exc:  # pragma: no cover
            out["roofline"]["fp64_fma_probe_tflops"] = "failed: %s" % exc
        if stages is not None:
            out["stages"] = stages
        if narrow is not None:
            out["narrow_domain"] = narrow
        if shared is not None:
            out["shared_operator"] = shared
        if pipelined is not None:
            out["pipelined"] = pipelined
        if accuracy is not None:
            out["accuracy"] = accuracy
        if cpu_res is not None:
            out["cpu_baseline"] = cpu_res
        if stitch is not None:
            out["stitch"] = stitch
        if stitch_u is not None:
            out["stitch_u"] = stitch_u
        emit(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
