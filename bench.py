#!/usr/bin/env python3
"""Benchmark of the per-element LSSVR enhancement hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on
rank 0.  For N > 1 the driver launches it through ``torch.distributed.run`` with one rank per GPU
(RCCL); started WITHOUT a launcher (``WORLD_SIZE`` unset) ``--gpus N`` spawns its N rank processes
itself, before anything touches a GPU -- it never silently runs one rank.  ``WORLD_SIZE`` set to
anything but N is an error.

Metric  : LSSVR-enhanced elements/s (BASELINE.json ``metric``).
Step    : one pass of the hot path over one batch of elements already resident in HBM:
          element-local P1 stiffness/load assembly (Dual.py:117-128) + the per-element Gram +
          solve (Dual.py:139-169), one fused launch per rank.
N = 1   : BASELINE config 2 -- degree 8 (M = 9), 16 collocation points, gamma = 1e4, 1e5 P1
          elements -- on the wide domain [-N, N] with h = 1/12 (100 008 elements; SURVEY.md
          finding 5: on [-1,1] the reference's own SLSQP loop stops converging above ~5e4
          elements, so the CPU baseline could not be timed on it; the kernel cost does not depend
          on the domain -- ``narrow_domain`` repeats the step on exactly 1e5 elements of [-1,1]).
N > 1   : BASELINE config 3 -- 10 000 008 elements IN TOTAL (h = 1/12), ne/N per rank
          (``"scaling": "strong"``), stitched by an all-gather of the sampled solution u over
          xGMI.  ``value`` is the compute-only rate (ne / (t_assemble_local +
          t_enhance); no collective and no host synchronisation inside the timed region),
          ``value_with_allgather`` the rate of the whole stitched step (kernel + rank-local
          evaluation of u + all-gather, the gather of step i overlapping the kernel of step
          i+1) -- the number BASELINE's ">= 6x at 8 GPUs" is judged on, since config 3 names the
          all-gather of u as part of the 8-GPU job.  ``--scaling weak`` runs 100 008 elements per
          GPU instead; the other mode is always reported as a second, shorter line
          (``"weak_scaling"`` / ``"strong_scaling"``).
Timing  : each rank brackets its K launches with HIP events on the launch stream after a
          barrier + device synchronisation; the maximum over ranks is taken AFTER the region.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

# Kernel arguments in device memory (the HIP runtime reads this when it initialises; this image's
# default already): with them in host memory every launch of the 10 us kernel pays 2 us more
# (measured 11.0 against 9.0 us).
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M_DEG8, N_COLLOC, GAMMA = 9, 16, 1.0e4
NE_WIDE = 100008                           # h = 1/12 exactly on [-4167, 4167]
NE_NARROW = 100000
NE_CONFIG3 = 10000008                      # BASELINE config 3, h = 1/12 on [-416667, 416667]
FP64_PEAK_TFLOPS = 78.6                    # MI355X vector = matrix FP64 peak (SURVEY.md 8(d):
                                           # 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz); probe below
HBM_PEAK_GBS = 8000.0
PREWARM_SECONDS = float(os.environ.get("LSSVR_BENCH_PREWARM", "0.05"))
PREWARM_DONE = 0
# The timed region: the K steps are captured ONCE in a hipGraph (untimed, like the warm-up) and the replay is
# timed -- the loop is launch-bound at BASELINE's size (the host needs ~7 us per call by call launch, the step
# takes ~8), which is the case hipGraphs are for.  LSSVR_BENCH_TIMED=eager: K launches issued call by call.
# The other mode is measured as well and reported beside `value` (`eager_loop` / `graph_replay`).
TIMED_MODE = os.environ.get("LSSVR_BENCH_TIMED", "graph")
TIMED_USED = []            # mode each timed_compute call really used ("graph" / "eager (<why>)"), in call order
SAMPLES_PER_ELEMENT = 2                    # u is stitched on the uniform grid of spacing h/2 (--stitch-samples)
NE_C5_WIDE = 1000008                       # BASELINE config 5 on [-41667, 41667], h = 1/12 exactly
NE_C5_NARROW = 1000000                     # ... and on [-1, 1] as BASELINE words it
HBM_ACHIEVABLE_GBS = 6290.0                # measured stream rate (MI355X_MICROARCH.md), quoted beside the 8 TB/s spec


def algorithmic_flops(M, n):
    """SURVEY.md 8(d), primal form: Gram M(M+1)n + A^T f 2Mn + KKT-LU 2/3 (M+2)^3 + 2 (M+2)^2."""
    return M * (M + 1) * n + 2 * M * n + (2.0 / 3.0) * (M + 2) ** 3 + 2 * (M + 2) ** 2


def algorithmic_flops_dual(M, n):
    """SURVEY.md 8(d), dual form: Gram (n+2)(n+3)M + LDL^T 1/3 (n+2)^3 + 2 (n+2)^2 + w 2M(n+2)."""
    return (n + 2) * (n + 3) * M + (1.0 / 3.0) * (n + 2) ** 3 + 2 * (n + 2) ** 2 + 2 * M * (n + 2)


def executed_fraction(key, ne, kernel_s, fallback_key=None):
    """Issue-slot utilisation of the FP64 pipe by what the kernel actually EXECUTES (the nominal
    `frac` prices SURVEY's direct-Gram flop count, which the Chebyshev-moment kernels do not execute): VALU
    wave-instructions per launch (profiles/instruction_counts.json: rocprofv3 SQ_INSTS_VALU of this kernel) x 4
    cycles each / (SIMDs x 2.4 GHz x measured duration).  <= 1 by construction; None when the kernel has no count."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "instruction_counts.json")))
        ent = tab.get(key) or tab[fallback_key]
    except Exception:
        return None
    per_el = ent["valu_per_element"] if "valu_per_element" in ent else ent["valu_per_wave"] / 64.0
    cycles = per_el * ne * 4.0 + ent.get("mfma_4x4x4_per_element", 0.0) * ne * 16.0     # (a 4x4x4 MFMA: ~20 cycles)
    return {"issue_slot_frac": cycles / (1024 * 2.4e9 * kernel_s), "valu_instructions_per_element": per_el,
            "source": ent["source"],
            "meaning": "fraction of the chip's FP64 issue slots (1024 SIMDs x 2.4 GHz / 4 cycles per wave-instruction) "
                       "the kernel's executed vector instructions fill during the measured launch"}


def algorithmic_bytes(M):
    """node coordinate 8 B + nodal value 8 B + 8 M B of coefficients."""
    return 16 + 8 * M


# ----------------------------------------------------------------------------------------
# self-launch: --gpus N without a launcher
# ----------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv):
    """Start the N rank processes (fresh children; this parent never touches a GPU), pass rank
    0's JSON line through, return the worst exit code."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LSSVR_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=(None if r == 0 else subprocess.DEVNULL)))
    # poll all ranks: the first non-zero exit ends the others (a rank that died would otherwise leave
    # its siblings blocked in a collective, holding their GPUs until the RCCL timeout)
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0:
                rc = max(rc, abs(r))
                for q in live:
                    q.terminate()
                deadline = time.time() + 10.0
                for q in live:
                    try:
                        q.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
                live = []
                break
    return rc


# ----------------------------------------------------------------------------------------
# CPU baseline: the reference's own per-element SLSQP loop (oracle restatement, "port")
# ----------------------------------------------------------------------------------------
def _cpu_worker(args):
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = "1"
    import numpy as np
    from oracle import lssvr_oracle as orc
    xa, xb, ua, ub, j, ne, gd, seed, varcoef, M, n = args        # (the element's own data: no mesh-sized pickles)
    rng = np.random.default_rng(seed)
    rhs, kw = orc.poisson_rhs, {}
    if varcoef:
        a, da, f = orc.varcoef_functions(*orc.varcoef_params())
        rhs, kw = f, {"coef_a": a, "coef_da": da}
    t0 = time.perf_counter()
    _, ok = orc.slsqp_element(rhs, xa, xb, ua, ub,
                              M, GAMMA, n, left=(j == 0), right=(j == ne - 1),
                              global_domain=gd, rng=rng, **kw)
    return int(ok), time.perf_counter() - t0


def wide_mesh(ne):
    """Host nodes / nodal values of the reference-valid synthetic mesh: h = 1/12 on [-ne/24, ne/24]
    (SURVEY.md 8(d) S-wide), u_i = sin(pi x_i) with the Dirichlet ends."""
    import numpy as np
    half = ne / 24.0
    nodes = np.arange(ne + 1, dtype=np.float64) * ((2.0 * half) / ne) - half
    nodes[-1] = half
    values = np.sin(np.pi * nodes)
    values[0] = values[-1] = 0.0
    return nodes, values, (-half, half)


def cpu_baseline(nodes_host, values_host, gd, per_core=16, varcoef=False, M=M_DEG8, n=N_COLLOC, note="",
                 central=False, budget_s=45.0):
    """Times the SLSQP loop on a bounded sample of the same mesh with every host core the
    box gives us (one process per core, like N copies of the single-threaded reference; one
    element per job).  ``varcoef``: the residual of BASELINE config 5 (an extension of Dual.py:43-44).
    ``central``: the sample is the elements nearest the origin of the mesh instead of an even spread -- degree
    32 / 64 points: far from the origin (|x|/h = 5e4) single elements run into SLSQP's iteration limit after
    ~200 s, near it every one converges in ~8-15 s (7.9 s).  ``budget_s``: jobs still running
    after that long are abandoned and NOT counted (the rate is finished elements / wall time)."""
    import multiprocessing as mp
    import numpy as np
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    ne = len(nodes_host) - 1
    count = min(ne, cores * per_core)
    if central:
        sample = np.arange(ne // 2 - count // 2, ne // 2 - count // 2 + count, dtype=np.int64)
    else:
        sample = np.linspace(0, ne - 1, count).astype(np.int64)
    jobs = [(float(nodes_host[j]), float(nodes_host[j + 1]), float(values_host[j]), float(values_host[j + 1]), int(j),
             ne, gd, 1000 + k, varcoef, M, n) for k, j in enumerate(sample)]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    res, abandoned = [], 0
    pool = ctx.Pool(cores)
    try:
        it = pool.imap_unordered(_cpu_worker, jobs)
        for _ in jobs:
            left = budget_s - (time.perf_counter() - t0)
            try:
                res.append(it.next(timeout=max(left, 0.01)))
            except mp.TimeoutError:
                abandoned = len(jobs) - len(res)
                break
    finally:
        pool.terminate()
        pool.join()
    wall = time.perf_counter() - t0
    done = len(res)
    conv = sum(r[0] for r in res)
    busy = sum(r[1] for r in res)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    where = ("the %d elements nearest the origin of" % count) if central else ("%d elements evenly spaced through" % count)
    return {
        "value": done / wall,
        "unit": "elements/s",
        "cores": cores,
        "cpu_model": model,
        "kind": "port",
        "sample": f"{where} the same {ne}-element mesh, degree {M - 1} / {n} points, "
                  f"per-element scipy SLSQP loop (oracle/lssvr_oracle.py::slsqp_element = "
                  f"Dual.py:20-98" + (" with the variable-coefficient residual -a u'' - a' u' - f, an "
                                      "extension of Dual.py:43-44" if varcoef else "")
                  + f"), one process per core; {done} finished, {conv} converged"
                  + (f", {abandoned} abandoned at the {budget_s:.0f} s budget (not counted)" if abandoned else "") + note,
        "single_core_value": (done / busy) if busy > 0 else None,
        "wall_s": wall,
    }


# ----------------------------------------------------------------------------------------
# one sharded workload resident in HBM + its timed regions
# ----------------------------------------------------------------------------------------
class Workload:
    """This rank's shard of ``ne_glob`` elements on [lo, hi]: nodes exactly as
    ``np.linspace(lo, hi, ne_glob + 1)`` gives them, nodal values sin(pi x) with the Dirichlet
    ends, output buffers, and the fused step bound once (``ops.StepPlan``)."""

    def __init__(self, ne_glob, lo, hi, M, n, rank, world, dev, nbuf=1):
        import numpy as np
        import torch
        from hybrid_fem_lssvr_amd import ops
        from hybrid_fem_lssvr_amd.distributed import ShardPlan
        self.M, self.n, self.rank, self.world, self.dev = M, n, rank, world, dev
        self.plan = ShardPlan(ne_glob, world)
        self.ne_glob = ne_glob
        self.s0, self.s1 = self.plan.bounds(rank)
        self.ne_loc = self.s1 - self.s0
        self.gd = (lo, hi)
        step = (hi - lo) / ne_glob
        nodes = np.arange(self.s0, self.s1 + 1, dtype=np.float64) * step + lo
        if self.s1 == ne_glob:
            nodes[-1] = hi
        values = np.sin(np.pi * nodes)
        if self.s0 == 0:
            values[0] = 0.0
        if self.s1 == ne_glob:
            values[-1] = 0.0
        self.nodes_h, self.values_h = nodes, values
        self.x = torch.as_tensor(nodes, device=dev)
        self.u = torch.as_tensor(values, device=dev)
        self.status = torch.empty(self.ne_loc, dtype=torch.int32, device=dev)
        self.bands = ops.p1_assemble(self.x, 2)
        # coefficient rows live at the head of a buffer padded to the largest shard, so that the
        # all-gather of W sends them from where the kernel wrote them (equal counts on every rank)
        self.Wflat = [torch.zeros(self.plan.max_size * M, dtype=torch.float64, device=dev) for _ in range(nbuf)]
        self.W = [f[:self.ne_loc * M].view(self.ne_loc, M) for f in self.Wflat]
        self.plans = [ops.StepPlan(self.x, self.u, M, GAMMA, n, elem_offset=self.s0, ne_global=ne_glob,
                                   global_domain=self.gd, bands=self.bands, out=w, status=self.status)
                      for w in self.W]

    def describe(self, degree):
        return ("1D Poisson, %d P1 elements in total on [%g, %g] (h = %.6g), %d per rank, Legendre degree %d "
                "(M = %d), %d collocation points, gamma = 1e4, f = pi^2 sin(pi x) in-kernel; step = "
                "element-local P1 assembly + per-element Gram + solve, one fused launch per rank"
                % (self.ne_glob, self.gd[0], self.gd[1], (self.gd[1] - self.gd[0]) / self.ne_glob,
                   self.plan.max_size, degree, self.M, self.n))


class Dist:
    """Process-group plumbing: barrier and max-over-ranks that cost nothing at world == 1."""

    def __init__(self, use_dist, backend, dev):
        import torch
        self.use, self.backend, self.dev = use_dist, backend, dev
        self._t = torch.zeros(1, dtype=torch.float32, device=dev) if use_dist else None

    def barrier(self):
        import torch
        import torch.distributed as dist
        if self.use:
            if self.backend == "nccl":
                dist.all_reduce(self._t)     # one preallocated element; synchronised below
            else:
                dist.barrier()
        torch.cuda.synchronize()

    def max(self, v):
        import torch
        import torch.distributed as dist
        if not self.use:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(self, ok):
        """True iff EVERY rank passes True: ranks agree on a failure before any of them enters the
        collectives of a timed region (a rank that raised on its own would leave the others blocked)."""
        return self.max(0.0 if ok else 1.0) == 0.0


def prewarm(launch, st):
    """Steady state before the W warm-up steps: the chip idles while the CPU baseline runs, and W steps of a few
    microseconds do not bring it back (measured: 0.61 against 0.51 ms per step at 1e7 elements, 8.6 against 8.0 us
    at 1e5) -- at least PREWARM_SECONDS of the same launches, untimed.  LSSVR_BENCH_PREWARM=0 switches it off; the
    count of the first (headline) measurement is reported as `prewarm_steps`."""
    import torch
    global PREWARM_DONE
    if PREWARM_SECONDS <= 0:
        return 0
    launch(st)                                     # (first call: lazy initialisation, not an estimate)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        launch(st)
    torch.cuda.synchronize()
    est = max((time.perf_counter() - t0) / 8, 1e-6)
    pre = int(min(40000, max(0, PREWARM_SECONDS / est)))
    for _ in range(pre):
        launch(st)
    torch.cuda.synchronize()
    if PREWARM_DONE == 0:
        PREWARM_DONE = pre + 9
    return pre + 9


REPLAYS = max(1, int(os.environ.get("LSSVR_BENCH_REPLAYS", "11")))
LSSVR_PICK_FASTER_MODE = os.environ.get("LSSVR_BENCH_PICK", "1") != "0"      # 0: `value` = TIMED_MODE's line, whatever the other reads


class Timed(dict):
    """Result of timed_compute: seconds per bracket of K steps -- ``s`` the median over the brackets (what
    ``value`` divides by), ``s_min`` / ``s_max``, ``brackets``, ``wall_s`` (host clock around the median's
    neighbourhood: the mean host time per bracket), ``mode``."""
    __getattr__ = dict.__getitem__


def timed_compute(wl, D, steps, warmup, mode=None, replays=None):
    """EXACTLY K steps of the fused step on the current stream between two HIP events, barrier + device
    synchronisation on both sides, nothing but the launches (mode "eager") or the replay of the K captured
    launches (mode "graph", see TIMED_MODE) inside -- and that bracket ``replays`` times (the
    median of >= 10 timed repetitions; min and max are reported beside it).  Every bracket is reduced to the
    maximum over ranks AFTER its region.  The mode really used is appended to TIMED_USED."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    st = torch.cuda.current_stream().cuda_stream
    plan = wl.plans[0]
    mode = mode or TIMED_MODE
    replays = replays or REPLAYS
    prewarm(plan.launch, st)
    for _ in range(warmup):
        plan.launch(st)
    graph, used = None, "eager"
    if mode == "graph":
        try:
            torch.cuda.synchronize()
            graph = ops.StepGraph(plan, steps=steps)
            graph.replay()                             # (untimed: the first replay uploads the graph)
            torch.cuda.synchronize()
            used = "graph"
        except Exception as exc:  # pragma: no cover
            graph, used = None, "eager (hipGraph capture failed: %r)" % (exc,)
    TIMED_USED.append(used)
    D.barrier()
    dev, wall = [], []
    for _ in range(replays):
        D.barrier()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        if graph is not None:
            graph.replay()
        else:
            for _ in range(steps):
                plan.launch(st)
        e1.record()
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)
        D.barrier()
        dev.append(e0.elapsed_time(e1) * 1e-3)
    dev = [D.max(v) for v in dev]
    srt = sorted(dev)
    return Timed(s=srt[len(srt) // 2], s_min=srt[0], s_max=srt[-1], brackets=len(srt),
                 wall_s=D.max(sum(wall) / len(wall)), mode=used)


def timing_fields(t, steps, ne_total):
    """The contract's value / ms_per_step (median bracket) with the spread beside them."""
    return {"value": ne_total * steps / t.s, "ms_per_step": t.s / steps * 1e3,
            "timed_brackets": t.brackets,
            "ms_per_step_min": t.s_min / steps * 1e3, "ms_per_step_max": t.s_max / steps * 1e3,
            "value_is": "elements of the K steps / the MEDIAN of %d event-bracketed repetitions of exactly K steps "
                        "(barrier + device synchronisation on both sides of every one; SURVEY.md 8(d))" % t.brackets,
            "host_wall_ms_per_step": t.wall_s / steps * 1e3}


def region_text(used):
    return ("the K steps captured once in a hipGraph (untimed, like the warm-up) and replayed"
            if used.startswith("graph") else "K launches issued call by call: " + used)


def timed_stitch(wl, D, steps, warmup, what, algo, samples=SAMPLES_PER_ELEMENT):
    """The stitched step, K times: step i computes into buffer i%2 on the main stream (and, for
    what == "u", evaluates the shard's solution at ``samples`` points per element, rank-local),
    then its all-gather runs on a side stream while step i+1 computes into the other buffer.
    Event-bracketed like timed_compute; the closing event waits for the last gather.
    Buffers are allocated first and the ranks AGREE that all of them succeeded before anyone enters
    a collective (RuntimeError on every rank otherwise).  Returns dict(seconds, bytes, intact)."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    from hybrid_fem_lssvr_amd.distributed import allgather_flat, stitch_traffic_model
    dev, world, rank = wl.dev, wl.world, wl.rank
    pad = wl.plan.max_size
    main = torch.cuda.current_stream(dev)
    width = samples if what == "u" else wl.M
    setup_error = None
    try:
        comm = torch.cuda.Stream(device=dev)
        if what == "u":
            h_el = wl.x[1:] - wl.x[:-1]
            fr = [(2 * k + 1) / (2.0 * samples) for k in range(samples)]
            xq = torch.stack([wl.x[:-1] + f * h_el for f in fr], dim=1).reshape(-1).contiguous()
        else:
            xq = None
        # padded send buffers (equal on every rank), gathered rank-major
        send = (wl.Wflat if what == "W" else
                [torch.zeros(pad * width, dtype=torch.float64, device=dev) for _ in range(2)])
        recv = [torch.empty(world * pad * width, dtype=torch.float64, device=dev) for _ in range(2)]
    except Exception as exc:  # pragma: no cover  (out of memory on one rank)
        setup_error = exc
    if not D.all_ok(setup_error is None):
        raise RuntimeError("stitch %s/%s: buffer setup failed on at least one rank (%r here)"
                           % (what, algo, setup_error))
    n_own = wl.ne_loc * width
    done = [None, None]

    def step(i):
        k = i & 1
        if done[k] is not None:
            main.wait_event(done[k])                     # the gather that read send[k] has finished
        wl.plans[k].launch(main.cuda_stream)
        if what == "u":
            ops.evaluate(wl.x, wl.W[k], xq, want_elem=False, out=send[k][:n_own], stream=main.cuda_stream)
        ready = torch.cuda.Event()
        ready.record(main)
        comm.wait_event(ready)
        with torch.cuda.stream(comm):
            allgather_flat(recv[k], send[k], rank, world, algo=algo)
            done[k] = torch.cuda.Event()
            done[k].record(comm)

    for i in range(max(warmup, 2)):
        step(i)
    D.barrier()
    D.barrier()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for i in range(steps):
        step(i)
    main.wait_stream(comm)
    e1.record(main)
    torch.cuda.synchronize()
    D.barrier()
    sec = D.max(e0.elapsed_time(e1) * 1e-3)
    kk = (steps - 1) & 1
    own = recv[kk][rank * pad * width: rank * pad * width + n_own]
    intact = bool(torch.equal(own, send[kk][:n_own]))
    model = stitch_traffic_model(wl.ne_glob, world, 8 * width)
    return {"seconds": sec, "bytes_received_per_rank_per_step": (world - 1) * pad * width * 8,
            "own_block_intact": intact, "bytes_per_element": 8 * width,
            "model": {"bytes_received_per_rank_per_step": model["bytes_received_per_rank_per_step"],
                      "xgmi_floor_ms_direct_all_pairs": model["direct_all_pairs_floor_s"] * 1e3,
                      "xgmi_floor_ms_ring": model["ring_floor_s"] * 1e3}}


def measure_multi(wl, D, steps, warmup, algos, samples_list=(2, 1)):
    """Compute-only region + the stitches with every all-gather algorithm: u at every sample count of
    ``samples_list`` (the first one is the headline ``stitch_u``), then W.  A failure inside a timed
    collective region is NOT caught per rank (the others would block in the collective): it ends the
    job, and bench.py's self-launcher / torchrun ends the sibling ranks."""
    t = timed_compute(wl, D, steps, warmup)
    total = wl.ne_glob * steps
    res = {"elements_total": wl.ne_glob, "elements_per_rank": wl.plan.max_size,
           **timing_fields(t, steps, wl.ne_glob), "step_s": t.s / steps,
           "timed_region": region_text(t.mode) + " (one graph per rank)"}
    jobs = [("u", sp) for sp in samples_list] + [("W", None)]
    for what, sp in jobs:
        by_algo = {}
        for algo in algos:
            r = timed_stitch(wl, D, steps, warmup, what, algo, **({"samples": sp} if sp else {}))
            r["value_with_allgather"] = total / r["seconds"]
            r["ms_per_step"] = r.pop("seconds") / steps * 1e3
            r["recv_GBps_per_rank"] = r["bytes_received_per_rank_per_step"] / (r["ms_per_step"] * 1e-3) / 1e9
            by_algo[algo] = r
        best = max(by_algo, key=lambda a: by_algo[a]["value_with_allgather"]) if by_algo else None
        key = "stitch_W" if what == "W" else ("stitch_u" if sp == samples_list[0] else "stitch_u_%d_sample" % sp)
        res[key] = {
            "what": ("rank-local lssvr_eval at %d point%s per element + all-gather of u (%d B per element)"
                     % (sp, "" if sp == 1 else "s", 8 * sp)) if what == "u" else
                    "all-gather of the coefficient rows W (%d B per element)" % (8 * wl.M),
            "samples_per_element": sp,
            "overlap": "gather of step i on a side stream under the kernels of step i+1 (double-buffered)",
            "algorithms": by_algo, "picked": best,
            "value_with_allgather": by_algo[best]["value_with_allgather"] if best else None,
            "ms_per_step": by_algo[best]["ms_per_step"] if best else None,
        }
    return res


# ----------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--domain", choices=["wide", "narrow"], default="wide")
    ap.add_argument("--degree", type=int, default=8)
    ap.add_argument("--colloc", type=int, default=N_COLLOC)
    ap.add_argument("--elements", type=int, default=0,
                    help="N = 1 / weak scaling: elements per GPU; strong scaling: elements in total (0 = config default)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1 only: strong = BASELINE config 3 (1e7 elements in total), weak = 1e5 per GPU")
    ap.add_argument("--solver", choices=["primal", "dual"], default="primal",
                    help="dual: time LSSVR_SOLVER_DUAL (north_star's Gram form) instead of the default solver")
    ap.add_argument("--config", type=int, choices=[2, 4, 5], default=2,
                    help="BASELINE config: 2 = degree 8 / 16 points (default; the flags above refine it), "
                         "4 = degree 32 / 64 points on 100 008 elements (h = 1/12; --domain narrow: 1e5 on [-1,1]), 5 = variable coefficient "
                         "-(a u')' = f, 1e6 elements, degree 8 / 16 points, tabulated a, a', f")
    ap.add_argument("--table-layout", choices=["point", "element"], default="point",
                    help="--config 5: layout of the tabulated a, a', f (point-major t[k, e] is what the lane kernel "
                         "reads at full HBM rate; the other layout is reported beside it)")
    ap.add_argument("--stitch-samples", type=int, default=0,
                    help="N > 1: points per element of the stitched u (0 = report both 1 and 2; value_with_allgather "
                         "is the 2-point line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-embedded", action="store_true",
                    help="default N = 1 run: skip the compact config 4 / config 5 lines it carries")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the stitch measurements")
    ap.add_argument("--no-second-line", action="store_true", help="N > 1: skip the other scaling mode")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # no launcher: be the launcher (decided before any GPU call; children are fresh processes)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to run a "
                         f"different rank count than asked for\n")
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # Only the result line may reach stdout (the driver reads ONE JSON line there): RCCL prints
    # a version banner to stdout when the process group comes up, libraries may print warnings.
    # Everything written to fd 1 from here on goes to stderr; the JSON is written to the saved fd.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.write(real_stdout, (line + "\n").encode())

    import numpy as np

    if args.config == 4:
        args.degree, args.colloc = 32, 64
    if args.config == 5:
        args.degree, args.colloc = 8, N_COLLOC
    M = args.degree + 1
    n = args.colloc
    multi = world > 1
    if args.config == 5 and multi:
        raise SystemExit("bench.py --config 5 is a one-GPU line (N > 1 runs BASELINE config 3)")
    if multi:
        if args.scaling == "strong":
            ne_glob = args.elements or NE_CONFIG3
        else:
            ne_glob = (args.elements or NE_WIDE) * world
    elif args.config == 5:
        ne_glob = args.elements or (NE_C5_WIDE if args.domain == "wide" else NE_C5_NARROW)
    else:
        ne_glob = args.elements or (NE_WIDE if args.domain == "wide" else NE_NARROW)
    if args.domain == "wide":
        half = ne_glob / 24.0                             # h = 1/12
        lo, hi = -half, half
    else:
        lo, hi = -1.0, 1.0
    # the default N = 1 run also carries compact lines of BASELINE configs 4 and 5 (`config4`, `config5`), so that
    # one driver run on a fresh box times all three
    embed = (not multi and args.config == 2 and args.degree == 8 and n == N_COLLOC and args.domain == "wide"
             and not args.elements and args.solver == "primal" and not args.no_embedded)

    # CPU baselines first: the worker processes are forked before this process touches the GPU (a forked child
    # of a GPU-initialised process is best avoided on this pool).  Rank 0 only; wide domain only (on [-1, 1] the
    # SLSQP loop stops converging above ~5e4 elements, SURVEY.md finding 5: for a narrow-domain line the sample
    # is taken from the h = 1/12 mesh of the same element count, and says so).
    cpu_res, cpu4, cpu5 = None, None, None
    if rank == 0 and not args.no_cpu_baseline and args.solver == "primal" and M <= 33:
        note = ""
        if args.domain == "wide":
            step_h = (hi - lo) / ne_glob
            nodes_h = np.arange(ne_glob + 1, dtype=np.float64) * step_h + lo
            nodes_h[-1] = hi
            values_h = np.sin(np.pi * nodes_h)
            values_h[0] = values_h[-1] = 0.0
            gd_h = (lo, hi)
        else:
            nodes_h, values_h, gd_h = wide_mesh(ne_glob + (-ne_glob) % 24)
            note = ("; sampled from the h = 1/12 mesh of (nearly) the same element count: on [-1, 1] the reference's "
                    "SLSQP stops converging above ~5e4 elements (SURVEY.md finding 5)")
        # ~10-30 s of CPU work: 16 elements per core at degree 8 (0.2 s each), ONE per core at degree 32 (8 s each)
        per_core = 1 if M > 22 else (8 if args.config == 5 else 16)
        cpu_res = cpu_baseline(nodes_h, values_h, gd_h, varcoef=(args.config == 5), per_core=per_core, M=M, n=n,
                               note=note, central=(M > 22))
        if embed:
            nd, vl, gd4 = wide_mesh(NE_WIDE)
            cpu4 = cpu_baseline(nd, vl, gd4, per_core=1, M=33, n=64, central=True)
            nd, vl, gd5 = wide_mesh(NE_C5_WIDE)
            cpu5 = cpu_baseline(nd, vl, gd5, varcoef=True, per_core=8)
            del nd, vl

    import torch
    import torch.distributed as dist
    from hybrid_fem_lssvr_amd import ops

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one rank per GPU; LSSVR_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse
    # the N > 1 code path on a single-GPU box (RCCL refuses duplicate devices)
    backend = os.environ.get("LSSVR_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU "
                             f"({torch.cuda.device_count()} visible); RCCL needs one GPU per rank")
        local_dev = local_rank
    else:
        local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # LSSVR_BENCH_FORCE_DIST=1: take the N > 1 plumbing (process group, collectives) with a single
    # rank -- lets a one-GPU box exercise RCCL initialisation and the collectives' stream logic
    use_dist = multi or bool(os.environ.get("LSSVR_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != world:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, expected {world}")
    D = Dist(use_dist, backend, dev)

    if multi:
        out = run_multi(args, D, M, n, ne_glob, lo, hi, rank, world, dev, backend, cpu_res)
    elif args.config == 5:
        out = measure_config5(D, M, n, ne_glob, lo, hi, dev, args.steps, args.warmup,
                              pm=(args.table_layout == "point"), full=True, cpu_res=cpu_res,
                              narrow_too=(args.domain == "wide" and not args.elements))
    else:
        default_run = args.domain == "wide" and not args.elements and args.solver == "primal"
        label = ("; BASELINE config %d" % args.config) if default_run and (M, n) in ((9, 16), (33, 64)) else ""
        out = measure_poisson(D, M, n, ne_glob, lo, hi, dev, args.steps, args.warmup, solver=args.solver, full=True,
                              cpu_res=cpu_res, use_dist=use_dist, label=label)
        if default_run:
            # the same step on exactly 1e5 elements of [-1, 1] (BASELINE.json's wording), where the CPU baseline
            # cannot be timed (SURVEY.md finding 5): shows that the kernel cost does not depend on the domain
            try:
                wn = Workload(NE_NARROW, -1.0, 1.0, M, n, 0, 1, dev)
                tn = timed_compute(wn, D, args.steps, args.warmup, replays=5)
                out["narrow_domain"] = {"workload": "%d elements on [-1, 1], same step" % NE_NARROW,
                                        "value": NE_NARROW * args.steps / tn.s, "ms_per_step": tn.s / args.steps * 1e3,
                                        "fallback_elements": int(wn.status.sum().item())}
                del wn
            except Exception as exc:  # pragma: no cover
                out["narrow_domain"] = {"error": repr(exc)}
        if embed:
            torch.cuda.empty_cache()
            for key, fn in (("config4", lambda: measure_poisson(D, 33, 64, NE_WIDE, -NE_WIDE / 24.0, NE_WIDE / 24.0, dev,
                                                                 args.steps, min(args.warmup, 5), full=False,
                                                                 cpu_res=cpu4, label="; BASELINE config 4")),
                            ("config5", lambda: measure_config5(D, M_DEG8, N_COLLOC, NE_C5_WIDE, -NE_C5_WIDE / 24.0,
                                                                 NE_C5_WIDE / 24.0, dev, args.steps,
                                                                 min(args.warmup, 5), pm=True, full=False,
                                                                 cpu_res=cpu5))):
                try:
                    out[key] = fn()
                except Exception as exc:  # pragma: no cover
                    out[key] = {"error": repr(exc)}
                torch.cuda.empty_cache()
    if rank == 0:
        emit(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------
# N > 1
# ----------------------------------------------------------------------------------------
def run_multi(args, D, M, n, ne_glob, lo, hi, rank, world, dev, backend, cpu_res=None):
    import torch
    import torch.distributed as dist
    from hybrid_fem_lssvr_amd.distributed import ALLGATHER_ALGOS
    algos = [] if args.no_gather else list(ALLGATHER_ALGOS)
    samples_list = (2, 1) if args.stitch_samples == 0 else (args.stitch_samples,)
    wl = Workload(ne_glob, lo, hi, M, n, rank, world, dev, nbuf=2)
    res = measure_multi(wl, D, args.steps, args.warmup, algos, samples_list)
    n_fallback = D.max(float(wl.status.sum().item()))

    # the other scaling mode, shorter (same code path, other sizes)
    second = None
    if not args.no_second_line:
        try:
            if args.scaling == "strong":
                ne2, lbl = NE_WIDE * world, "weak_scaling"
            else:
                ne2, lbl = NE_CONFIG3, "strong_scaling"
            half2 = ne2 / 24.0
            l2, h2 = (-half2, half2) if args.domain == "wide" else (lo, hi)
            wl2 = Workload(ne2, l2, h2, M, n, rank, world, dev, nbuf=2)
            r2 = measure_multi(wl2, D, max(args.steps // 2, 5), min(args.warmup, 5), algos, samples_list[:1])
            second = (lbl, {"workload": wl2.describe(args.degree), "value": r2["value"],
                            "ms_per_step": r2["ms_per_step"],
                            "value_with_allgather": (r2.get("stitch_u") or {}).get("value_with_allgather"),
                            "stitch_u_picked": (r2.get("stitch_u") or {}).get("picked"),
                            "value_with_allgather_of_W": (r2.get("stitch_W") or {}).get("value_with_allgather")})
            del wl2
        except Exception as exc:  # pragma: no cover
            second = ("second_line", {"error": repr(exc)})

    # the same TOTAL workload on rank 0 alone (the other ranks wait): what one GPU does with it
    one_rank = None
    if args.scaling == "strong" and not args.no_second_line:
        try:
            if rank == 0:
                w1 = Workload(ne_glob, lo, hi, M, n, 0, 1, dev)
                d1 = timed_compute(w1, Dist(False, backend, dev), max(args.steps // 4, 3), 2, replays=3).s
                one_rank = {"what": "the same %d elements on rank 0 alone, compute only" % ne_glob,
                            "value": ne_glob * max(args.steps // 4, 3) / d1,
                            "ms_per_step": d1 / max(args.steps // 4, 3) * 1e3}
                del w1
            D.barrier()
        except Exception as exc:  # pragma: no cover
            one_rank = {"error": repr(exc)}

    # one rank alone on ITS share of the job (the per-rank size): the denominator of the scaling efficiency
    per_rank = None
    try:
        if rank == 0:
            wr = Workload(wl.plan.max_size, lo, lo + (hi - lo) * wl.plan.max_size / ne_glob, M, n, 0, 1, dev)
            k3 = max(args.steps // 2, 5)
            dr = timed_compute(wr, Dist(False, backend, dev), k3, 3, replays=5).s
            per_rank = {"what": "%d elements (one rank's share) on rank 0 alone, compute only" % wl.plan.max_size,
                        "value": wl.plan.max_size * k3 / dr, "ms_per_step": dr / k3 * 1e3}
            del wr
        D.barrier()
    except Exception as exc:  # pragma: no cover
        per_rank = {"error": repr(exc)}

    if rank != 0:
        return None
    su = res.get("stitch_u") or {}
    out = {
        "metric": "LSSVR-enhanced elements/sec, 1D Poisson deg-%d/%d-pt" % (args.degree, n),
        "value": res["value"],
        "value_with_allgather": su.get("value_with_allgather"),
        "value_definition": (
            "value = compute only (assembly + per-element Gram + solve on every rank, no collective in the "
            "timed region; SURVEY.md 8(d)); value_with_allgather = the whole stitched step (kernel + rank-local "
            "evaluation of u at %d points per element + all-gather of u over xGMI, gather of step i under the "
            "kernel of step i+1). BASELINE's '>= 6x at 8 GPUs' is judged on value_with_allgather: config 3 "
            "names the all-gather of u as part of the 8-GPU job." % samples_list[0]),
        "unit": "elements/s",
        "n_gpus": world,
        "ranks_in_process_group": dist.get_world_size(),
        "backend": "RCCL (torch.distributed nccl)" if backend == "nccl" else backend,
        "steps": args.steps,
        "warmup": args.warmup,
        "prewarm_steps": PREWARM_DONE,
        "ms_per_step": res["ms_per_step"],
        "ms_per_step_min": res["ms_per_step_min"],
        "ms_per_step_max": res["ms_per_step_max"],
        "timed_brackets": res["timed_brackets"],
        "value_is": res["value_is"],
        "ms_per_step_with_allgather": su.get("ms_per_step"),
        "host_wall_ms_per_step": res["host_wall_ms_per_step"],
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": wl.describe(args.degree) + (", BASELINE config 3" if ne_glob == NE_CONFIG3 else ""),
            "elements_total": ne_glob,
            "elements_per_gpu": wl.plan.max_size,
            "parallelism": "elements sharded contiguously over %d ranks (one per GPU), one halo node, "
                           "no data-path collective before or during the kernels" % world,
            "solver": "primal, BC-eliminated SPD (M-2), Chebyshev-moment Gram, LDL^T",
            "fallback_elements": int(n_fallback),
            "timing": "per-rank HIP events around the K steps (after barrier + device sync), max over ranks taken after the region",
            "timed_region": res.get("timed_region"),
        },
        "stitch_u": res.get("stitch_u"),
        "stitch_W": res.get("stitch_W"),
    }
    # AGGREGATE roofline of the compute-only region: every rank launches the same fused step on its shard; the
    # region lasts as long as the slowest rank's K launches (max over ranks, taken after the region), so the job's
    # achieved rate is the algorithmic work of ALL shards over that time, against N x the one-GPU peaks
    flops, byts = algorithmic_flops(M, n), algorithmic_bytes(M)
    step_kernel, _, pipe, _, count_key = poisson_labels(M, n, False)
    ach = flops * ne_glob / res["step_s"] / 1e12
    gbs = byts * ne_glob / res["step_s"] / 1e9
    out["roofline"] = {
        "bound": "fp64-valu", "pipe": pipe, "kernel": step_kernel + ", one per rank",
        "achieved": ach, "peak": FP64_PEAK_TFLOPS * world, "unit": "TFLOP/s", "frac": ach / (FP64_PEAK_TFLOPS * world),
        "peak_is": "%d GPUs x %.1f TFLOP/s" % (world, FP64_PEAK_TFLOPS),
        "flops_per_element": flops, "elements_per_launch": wl.plan.max_size, "elements_per_step_all_ranks": ne_glob,
        "kernel_us_avg": res["step_s"] * 1e6,
        "kernel_us_is": "launch-to-launch duration of the step's kernel(s) inside the timed region on the SLOWEST rank "
                        "(median bracket / K, max over ranks per bracket)",
        "achieved_is": "direct-Gram-EQUIVALENT TFLOP/s of the whole job (SURVEY.md 8(d) flops x all elements / kernel_us_avg)",
        "executed": executed_fraction(count_key, wl.plan.max_size, res["step_s"]),
        "traffic": None,
    }
    out["roofline_hbm"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                           "frac": gbs / (HBM_PEAK_GBS * world), "bytes_per_element": byts, "traffic": None}
    if su.get("ms_per_step"):
        out["roofline_with_allgather"] = {
            "what": "the same algorithmic work over the stitched step's time (kernel + evaluation + all-gather of u)",
            "frac_fp64": flops * ne_glob / (su["ms_per_step"] * 1e-3) / 1e12 / (FP64_PEAK_TFLOPS * world),
            "xgmi_recv_GBps_per_rank": (su.get("algorithms") or {}).get(su.get("picked"), {}).get("recv_GBps_per_rank"),
            "xgmi_peak_GBps_per_rank_inbound": 7 * 76.8}
    if cpu_res is not None:
        out["cpu_baseline"] = cpu_res
    for key in res:
        if key.startswith("stitch_u_"):
            out[key] = res[key]
    if per_rank is not None:
        out["one_rank_per_rank_size"] = per_rank
        if "value" in per_rank:
            ideal = world * per_rank["value"]
            out["scaling_efficiency"] = {
                "definition": "rate of the N-rank job / (N x the rate of one rank alone on the per-rank element count)",
                "compute_only": res["value"] / ideal,
                "with_allgather_of_u": (su.get("value_with_allgather") or 0.0) / ideal,
                **{("with_allgather_of_u_%d_sample" % res[k]["samples_per_element"]):
                   (res[k].get("value_with_allgather") or 0.0) / ideal for k in res if k.startswith("stitch_u_")},
            }
    if second is not None:
        out[second[0]] = second[1]
    if one_rank is not None:
        out["one_rank_same_workload"] = one_rank
    return out


# ----------------------------------------------------------------------------------------
# N = 1, BASELINE config 5: -(a u')' = f with a tabulated smooth random a(x)
# ----------------------------------------------------------------------------------------
C5_BYTES_PER_ELEMENT = 16 + 8 * M_DEG8 + 3 * 8 * N_COLLOC       # x, u, W + rows of f, a, a' (SURVEY.md 8(d)): 472


def c5_flops(M, n):
    """SURVEY.md 8(d) primal form + the row combination rho = a L'' + (a'/scl)(L' - C1): 2 FMAs per
    row entry and point, (M-2) entries."""
    return algorithmic_flops(M, n) + 4 * (M - 2) * n


def _varcoef_device_tables(xc):
    """a, a', f at the points ``xc`` (device float64 tensor of any shape), on the DEVICE with torch
    (synthetic input generation only -- not part of any timed region, not the product): the same
    formulas as oracle/lssvr_oracle.py::varcoef_functions with SURVEY.md 8(d)'s seed; the accuracy
    block recomputes the sampled rows with numpy and the kernel is checked against those."""
    import numpy as np
    import torch
    rng = np.random.default_rng(20260130)
    c = rng.uniform(-1.0, 1.0, 8)
    phi = rng.uniform(0.0, 2.0 * np.pi, 8)
    a = torch.ones_like(xc)
    da = torch.zeros_like(xc)
    for k in range(1, 9):
        ang = (k * np.pi) * xc + phi[k - 1]
        a += (0.5 * c[k - 1] / k) * torch.sin(ang)
        da += (0.5 * np.pi * c[k - 1]) * torch.cos(ang)
    f = -da * np.pi * torch.cos(np.pi * xc) + a * np.pi ** 2 * torch.sin(np.pi * xc)
    return a, da, f


def measure_config5(D, M, n, ne, lo, hi, dev, steps, warmup, pm=True, full=True, cpu_res=None, narrow_too=False):
    """BASELINE config 5 (one GPU): the timed region (the fused variable-coefficient step), its roofline priced
    with the region's per-step time -- HBM is the binding roof -- the enhancement kernel's stamps beside it,
    accuracy of what was timed; ``full`` adds the other table layout, the eager loop and the narrow domain."""
    import numpy as np
    import torch
    from hybrid_fem_lssvr_amd import ops

    def build(ne_, lo_, hi_, pm_=True):
        step = (hi_ - lo_) / ne_
        nodes = np.arange(ne_ + 1, dtype=np.float64) * step + lo_
        nodes[-1] = hi_
        values = np.sin(np.pi * nodes)
        values[0] = values[-1] = 0.0
        x = torch.as_tensor(nodes, device=dev)
        u = torch.as_tensor(values, device=dev)
        xc = ops.colloc_points(x, n, point_major=pm_)
        a_c, da_c, f_c = _varcoef_device_tables(xc)
        xq = ops.quad_points(x, 2)
        a_q, _, f_q = _varcoef_device_tables(xq)
        del xc, xq
        W = torch.empty((ne_, M), dtype=torch.float64, device=dev)
        st = torch.empty(ne_, dtype=torch.int32, device=dev)
        bands = ops.p1_assemble(x, 2, rhs_quad=f_q, a_quad=a_q)
        return dict(nodes=nodes, values=values, x=x, u=u, a=a_c, da=da_c, f=f_c, aq=a_q, fq=f_q, W=W, st=st,
                    bands=bands, gd=(lo_, hi_), ne=ne_, pm=pm_)

    def enh(w, **kw):
        return ops.enhance_varcoef(w["x"], w["u"], M, GAMMA, n, w["a"], w["da"], w["f"], global_domain=w["gd"],
                                   out=w["W"], status=w["st"], point_major=w["pm"], **kw)

    def step_plan(w):
        return ops.StepPlanVarcoef(w["x"], w["u"], M, GAMMA, n, w["a"], w["da"], w["f"], w["fq"], w["aq"],
                                   nquad=2, point_major=w["pm"], global_domain=w["gd"], bands=w["bands"],
                                   out=w["W"], status=w["st"])

    def timed(w, steps, warmup, mode=None, replays=None):
        class _One:                                  # (timed_compute's view of a workload: .plans[0])
            plans = [step_plan(w)]
        return timed_compute(_One, D, steps, warmup, mode=mode, replays=replays)

    w = build(ne, lo, hi, pm)
    t = timed(w, steps, warmup)
    step_s = t.s / steps
    eager_loop = None
    if full and t.mode.startswith("graph"):
        try:
            te = timed(w, steps, warmup, mode="eager")
            eager_loop = {"what": "the same K steps issued call by call from Python (one stream)", "unit": "elements/s",
                          **timing_fields(te, steps, ne)}
            eager_loop.pop("value_is")
        except Exception as exc:  # pragma: no cover
            eager_loop = {"error": repr(exc)}
    n_fallback = int(w["st"].sum().item())
    # the enhancement kernel ALONE, stamped inside a running sequence and in isolation: secondary figures
    nprof = min(max(steps, 20), 50)
    enh(w, profiled=True, repeats=nprof)                                        # (untimed: steady state)
    k_s = sorted(enh(w, profiled=True, repeats=nprof))
    k_avg, k_med = sum(k_s) / len(k_s), k_s[len(k_s) // 2]
    k_iso = sorted(enh(w, profiled=True) for _ in range(20 if full else 8))
    step_plan(w).launch()                        # (leave the step's result in W for the accuracy block)
    torch.cuda.synchronize()

    # the other table layout on the same mesh (same values transposed; bit-equal W expected)
    other = None
    if full:
      try:
        W_first = w["W"].clone()
        wo = dict(w, a=w["a"].t().contiguous(), da=w["da"].t().contiguous(), f=w["f"].t().contiguous(), pm=not pm)
        do = timed(wo, max(steps // 2, 5), min(warmup, 5), replays=3).s
        ko = sorted(enh(wo, profiled=True) for _ in range(20))
        torch.cuda.synchronize()
        other = {"table_layout": "element-major t[e, k]" if pm else "point-major t[k, e]",
                 "value": ne * max(steps // 2, 5) / do, "ms_per_step": do / max(steps // 2, 5) * 1e3,
                 "kernel_us_avg": sum(ko) / len(ko) * 1e6,
                 "hbm_GBps": C5_BYTES_PER_ELEMENT * ne / (sum(ko) / len(ko)) / 1e9,
                 "W_bit_equal_to_primary_layout": bool(torch.equal(W_first, wo["W"]))}
        del wo, W_first
        step_plan(w).launch()                    # leave the primary layout's result in W for the accuracy block
        torch.cuda.synchronize()
      except Exception as exc:  # pragma: no cover
        other = {"error": repr(exc)}

    # accuracy of what was just timed: sampled elements against the 60-digit minimiser of the QP
    # built from numpy-tabulated a, a', f at np.linspace's points (the device tables above are
    # torch's sin / cos: they differ from numpy's by an ulp or two, which is an INPUT difference;
    # so the sampled elements are recomputed by the kernel on numpy-tabulated rows)
    try:
        from oracle import lssvr_oracle as orc
        from oracle import closed_form_mp as cf
        a_f, da_f, f_f = orc.varcoef_functions(*orc.varcoef_params())
        sel = np.unique(np.linspace(0, ne - 1, 7 if full else 4).astype(np.int64))
        idx = torch.as_tensor(sel, device=dev)
        xs = torch.stack([w["x"][idx], w["x"][idx + 1]], 1).cpu().numpy()
        accuracy = {"sampled_elements": int(len(sel))}
        Wk = []
        for k, i in enumerate(sel):
            nd = np.array([xs[k, 0], xs[k, 1]])
            xk = np.linspace(nd[0], nd[1], n)[None, :]
            Wi, sti = ops.enhance_varcoef(torch.as_tensor(nd, device=dev),   # (one element: either layout)
                                          torch.as_tensor(w["values"][i:i + 2].copy(), device=dev), M, GAMMA, n,
                                          torch.as_tensor(a_f(xk), device=dev), torch.as_tensor(da_f(xk), device=dev),
                                          torch.as_tensor(f_f(xk), device=dev), elem_offset=int(i), ne_global=ne,
                                          global_domain=w["gd"])
            Wk.append(Wi.cpu().numpy()[0])
        Wk = np.array(Wk)
        if cf.HAVE_MP:
            tr = cf.truth_all(w["nodes"], w["values"], M, GAMMA, n, f_f, w["gd"], sel, coef_a=a_f, coef_da=da_f)
            accuracy["rel_l2_vs_60_digit_minimiser"] = float(orc.rel_l2_coef(Wk, tr).max())
            accuracy["rel_l2_bubble_vs_60_digit_minimiser"] = float(orc.rel_l2_bubble(Wk, tr).max())
        W_run = w["W"][idx].cpu().numpy()
        accuracy["rel_l2_timed_rows_vs_numpy_tabulated_rows"] = float(orc.rel_l2_coef(W_run, Wk).max())
        xq_h = np.linspace(w["nodes"][0], w["nodes"][-1], 20001)
        norms = ops.eval_error(w["x"], w["W"], torch.as_tensor(xq_h, device=dev)).cpu().numpy()
        accuracy["rel_l2_vs_sin_pi_x_on_20001_probes"] = float(np.sqrt(norms[0] / norms[1]))
        accuracy["note"] = ("60-digit minimiser of the QP with a, a', f tabulated by numpy at np.linspace's points; the "
                            "timed launch reads torch-tabulated rows (an ulp or two away in a, a', f): third figure")
    except Exception as exc:  # pragma: no cover
        accuracy = {"error": repr(exc)}

    narrow = None
    if narrow_too:
        wn = build(NE_C5_NARROW, -1.0, 1.0, pm)
        dn = timed(wn, max(steps // 2, 5), min(warmup, 5), replays=3).s
        kn = sorted(enh(wn, profiled=True) for _ in range(20))
        narrow = {"workload": "%d elements on [-1, 1] (BASELINE's wording; the SLSQP baseline does not converge "
                              "there), same step" % NE_C5_NARROW,
                  "value": NE_C5_NARROW * max(steps // 2, 5) / dn, "ms_per_step": dn / max(steps // 2, 5) * 1e3,
                  "kernel_us_avg": sum(kn) / len(kn) * 1e6, "fallback_elements": int(wn["st"].sum().item())}
        del wn

    flops = c5_flops(M, n)
    byts = C5_BYTES_PER_ELEMENT
    gbs = byts * ne / step_s / 1e9
    tfl = flops * ne / step_s / 1e12
    degree = M - 1
    out = {
        "metric": "LSSVR-enhanced elements/sec, variable-coefficient -(a u')'=f deg-%d/%d-pt" % (degree, n),
        **timing_fields(t, steps, ne),
        "unit": "elements/s",
        "n_gpus": 1,
        "steps": steps,
        "warmup": warmup,
        "prewarm_steps": PREWARM_DONE,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": ("BASELINE config 5: -(a u')' = f, a(x) = 1 + 0.5 sum_k c_k sin(k pi x + phi_k)/k (SURVEY.md 8(d), "
                         "seed 20260130), manufactured u = sin(pi x); %d P1 elements on [%g, %g] (h = %.6g), Legendre "
                         "degree %d (M = %d), %d collocation points, gamma = 1e4; a, a', f tabulated per element and "
                         "point (3 x %d doubles per element, resident in HBM, %s); step = element-local P1 assembly "
                         "(a-weighted stiffness, 2-point Gauss) + per-element Gram + solve: one fused launch (lssvr_step_varcoef)"
                         % (ne, lo, hi, (hi - lo) / ne, degree, M, n, n,
                            "point-major t[k, e]" if pm else "element-major t[e, k]")),
            "elements_per_gpu": ne,
            "elements_total": ne,
            "parallelism": "one rank",
            "solver": "primal, BC-eliminated SPD (M-2), direct Gram of the weighted rows, LDL^T (lane per element)",
            "fallback_elements": n_fallback,
            "timing": "HIP events around exactly K steps on the launch stream, barrier + device synchronisation on "
                      "both sides, repeated; median reported",
            "timed_region": region_text(t.mode),
        },
        "roofline": {
            "bound": "hbm",
            "why": ("arithmetic intensity %.1f flop/B is below the ridge (78.6 TFLOP/s / 8 TB/s = 9.8), so HBM is the "
                    "roofline that prices it; measured apart (profiles/r03_pm_pattern.txt): this table pattern alone "
                    "streams at 5.9 TB/s (80 us for 472 B x 1e6), the kernel's 2 200 vector instructions per element "
                    "take 77 us of issue time, and the two overlap only partly -- see roofline_fp64.executed"
                    % (flops / byts)),
            "kernel": "step_small_vc_kernel<M=%d, %s> (ONE launch per step: the variable-coefficient enhancement "
                      "blocks + the a-weighted P1 assembly blocks of the same grid)" % (M, "RHS_ARRAY_PM" if pm else "RHS_ARRAY"),
            "achieved": gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS,
            "frac_of_achievable_6.29TBs": gbs / HBM_ACHIEVABLE_GBS,
            "bytes_per_element": byts,
            "elements_per_launch": ne,
            "kernel_us_avg": step_s * 1e6,
            "kernel_us_is": "launch-to-launch duration of the step's kernel INSIDE THE TIMED REGION = the median bracket / K "
                            "(HIP events on the launch stream); rocprofv3 --kernel-trace --stats of the same command "
                            "(profiles/) reads the kernel's begin -> end stamps, which must agree to a few %",
            "enhancement_only": {
                "kernel": "enhance_small_kernel<M=%d, %s, varcoef>" % (M, "RHS_ARRAY_PM" if pm else "RHS_ARRAY"),
                "what": "the enhancement WITHOUT the assembly blocks, begin -> end stamps of the dispatch",
                "kernel_us_in_sequence_avg": k_avg * 1e6,
                "kernel_us_in_sequence_median": k_med * 1e6,
                "kernel_us_isolated_avg": sum(k_iso) / len(k_iso) * 1e6,
                "launches": nprof,
                "frac_in_sequence": byts * ne / k_avg / 1e9 / HBM_PEAK_GBS,
            },
            "traffic": None,
            "traffic_source": None,
        },
        "roofline_fp64": {
            "bound": "fp64-valu",
            "executed": executed_fraction("c5_point_major_M%d_n%d" % (M, n) if pm else "c5_element_major_M%d_n%d" % (M, n),
                                          ne, step_s),
            "achieved": tfl,
            "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": tfl / FP64_PEAK_TFLOPS,
            "flops_per_element": flops,
            "flops_formula": "SURVEY.md 8(d) primal form + 4 (M-2) n for the weighted row combination",
        },
        "accuracy": accuracy,
    }
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            tr_ = tj.get("c5_M%d_n%d_ne%d" % (M, n, ne))
            if tr_:
                out["roofline"]["traffic"] = tr_["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = ("profiles/traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                     "passes of this kernel at this size, calibrated with "
                                                     "lssvr_stream_probe; NOT measured in this run"
                                                     % tj.get("_round", "committed profile"))
        except Exception:
            pass
    if eager_loop is not None:
        out["eager_loop"] = eager_loop
    if other is not None:
        out["other_table_layout"] = other
    if narrow is not None:
        out["narrow_domain"] = narrow
    if cpu_res is not None:
        out["cpu_baseline"] = cpu_res
    return out


# ----------------------------------------------------------------------------------------
# N = 1, Poisson rows (BASELINE configs 2 and 4, the 1e7-element mesh, the dual solver)
# ----------------------------------------------------------------------------------------
def poisson_labels(M, n, dual):
    """(kernel of the timed step, enhancement kernel, pipe text, solver text, instruction-count key)."""
    if dual:
        big = max(M, n) > 32
        enh = "enhance_dual_w64_kernel" if big else "enhance_dual_kernel"
        return ("p1_assemble_kernel + %s (two launches per step)" % enh, enh,
                "FP64 vector FMA only (row per lane: Gram by scalar FMAs, partial-pivot LU with the pivot row "
                + ("read out of the pivot lane by v_readfirstlane, two waves per SIMD" if big else "through LDS")
                + ", <= 3 safeguarded refinement steps); no MFMA is issued",
                "dual Gram form (K + I/gamma) alpha = y: boundary block pivot, Jacobi equilibration, partial-pivot LU",
                "dual_M%d_n%d" % (M, n))
    if M <= 22:
        return ("step_small_kernel<M=%d> (ONE launch per step: the per-element enhancement blocks + the P1 assembly "
                "blocks of the same grid)" % M, "enhance_small_kernel<M=%d, RHS_SIN>" % M,
                "FP64 vector FMA only (lane per element, no MFMA issued); the FP64 vector and matrix peaks of "
                "gfx950 are the same 78.6 TFLOP/s and share one pipe (DESIGN.md section 3)",
                "primal, BC-eliminated SPD (M-2), Chebyshev-moment Gram, LDL^T", "step_small_M%d_n%d" % (M, n))
    second = "solve4_parity_kernel" if n >= 2 * (M - 2) else "solve4_kernel (+ refinement kernels when n <= M + 12)"
    return ("p1_assemble_kernel + moments_kernel + %s (three launches per step)" % second,
            "moments_kernel + %s (the pair, gap included)" % second,
            "FP64 vector pipe: Chebyshev moments (lane per element) + parity-split four-systems-per-wave "
            "DPP-broadcast LDL^T (persistent waves); it executes about a quarter of the flops the formula "
            "prices, so frac can exceed what a direct Gram could reach; the f64-MFMA Gram kernel "
            "(LSSVR_SOLVER_PRIMAL_WAVE) is 2x slower (DESIGN.md section 3.8); vector and matrix FP64 share one "
            "pipe at the same 78.6 TFLOP/s peak",
            "primal, BC-eliminated SPD (M-2), Chebyshev-moment Gram, parity-split LDL^T + coupling "
            "iteration (two kernels, workspace)", "step_large_M%d_n%d" % (M, n))


def poisson_accuracy(wl, W, M, n, dev, nsel=9, ntruth=5):
    """Accuracy of what was just timed (reported with every timing): sampled elements against
    the float64 KKT oracle and the 60-digit minimiser (whole polynomial and the enhancement alone), and the
    stitched u(x) against sin(pi x) on a probe grid."""
    import numpy as np
    import torch
    from hybrid_fem_lssvr_amd import ops
    try:
        from oracle import lssvr_oracle as orc
        from oracle import closed_form_mp as cf
        nodes_h, values_h, ne = wl.nodes_h, wl.values_h, wl.ne_loc
        sel = np.unique(np.linspace(0, ne - 1, nsel).astype(np.int64))
        W_sel = W[torch.as_tensor(sel, device=dev)].cpu().numpy()

        def system(i):
            return orc.element_system(nodes_h[i], nodes_h[i + 1],
                                      *orc.boundary_values(int(i), wl.ne_glob, nodes_h[i], nodes_h[i + 1],
                                                           values_h[i], values_h[i + 1], wl.gd), M, GAMMA, n)
        Wo = np.array([orc.solve_primal_kkt(system(i)) for i in sel])
        acc = {"sampled_elements": int(len(sel)),
               "rel_l2_vs_float64_kkt_oracle": float(orc.rel_l2_coef(W_sel, Wo).max())}
        if cf.HAVE_MP:
            tr = np.array([cf.solve_truth(system(i)) for i in sel[:ntruth]])
            acc["rel_l2_vs_60_digit_minimiser"] = float(orc.rel_l2_coef(W_sel[:ntruth], tr).max())
            acc["rel_l2_bubble_vs_60_digit_minimiser"] = float(orc.rel_l2_bubble(W_sel[:ntruth], tr).max())
        xq_h = np.linspace(nodes_h[0], nodes_h[-1], 20001)
        norms = ops.eval_error(wl.x, W, torch.as_tensor(xq_h, device=dev)).cpu().numpy()
        acc["rel_l2_vs_sin_pi_x_on_20001_probes"] = float(np.sqrt(norms[0] / norms[1]))
        acc["max_abs_err_vs_sin_pi_x"] = float(norms[2])
        acc["note"] = ("nodal values are sin(pi x_i) here (device-resident synthetic input), so the "
                       "last figure is the enhancement's own error, not the P1 nodal error")
        return acc
    except Exception as exc:  # pragma: no cover
        return {"error": repr(exc)}


def measure_poisson(D, M, n, ne_glob, lo, hi, dev, steps, warmup, solver="primal", full=True, cpu_res=None,
                    use_dist=False, label=""):
    """One N = 1 line of the Poisson path: the timed region (``timed_compute``), the roofline of the KERNEL(S) THE
    TIMED REGION LAUNCHES priced with the region's own per-step time, the enhancement kernel's stamped durations
    beside it, accuracy of what was timed; ``full`` adds the secondary measurements of the headline line."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    degree = M - 1
    wl = Workload(ne_glob, lo, hi, M, n, 0, 1, dev)
    ne_loc = wl.ne_loc
    x, u, W, status, gd = wl.x, wl.u, wl.W[0], wl.status, wl.gd
    dual = solver == "dual"
    solver_id = ops.SOLVER_DUAL if dual else ops.SOLVER_PRIMAL
    st = torch.cuda.current_stream().cuda_stream
    if dual:
        # the dual Gram solver has no fused step: a step is assembly + enhancement, two launches
        class _TwoLaunch:
            def launch(self, s=None):
                ops.p1_assemble(x, 2, out=wl.bands, stream=s)
                ops.enhance(x, u, M, GAMMA, n, global_domain=gd, solver=solver_id, out=W, status=status, stream=s)
        _TwoLaunch.W, _TwoLaunch.status = W, status
        wl.plans = [_TwoLaunch()]
    t = timed_compute(wl, D, steps, warmup)
    n_fallback = int(status.sum().item())
    # The same K steps issued the OTHER way (call by call from Python if the graph replay was timed, and vice versa),
    # same plan, same buffers.  `value` takes the mode with the lower median: a replay pays ~12 us once per
    # hipGraphLaunch (0.6 us per step at K = 20, nothing at K = 200; profiles/r04_graph_floor.txt), the Python loop
    # needs ~7 us of host time per call and starves the GPU on a slow host moment -- which one is faster depends
    # on K and on the box, both are reported.
    t_other, other_equal = None, None
    if full and LSSVR_PICK_FASTER_MODE:
        try:
            W_ref = W.clone()
            t_other = timed_compute(wl, D, steps, warmup, mode=("eager" if t.mode.startswith("graph") else "graph"))
            other_equal = bool(torch.equal(W_ref, W))
            del W_ref
            if t_other.s < t.s:
                t, t_other = t_other, t
        except Exception as exc:  # pragma: no cover
            t_other = None
            sys.stderr.write("bench.py: the other timed mode failed: %r\n" % (exc,))
    step_s = t.s / steps

    # the per-element enhancement kernel ALONE (no assembly blocks), from HIP events that hipExtLaunchKernelGGL
    # stamps with the dispatch's own begin / end times -- rocprofv3's kernel duration -- inside a running sequence
    # of the same launches and in isolation: SECONDARY figures; `roofline.frac` is priced with the timed region
    nprof = min(max(steps, 20), 100)
    ops.enhance_profiled(x, u, M, GAMMA, n, global_domain=gd, out=W, status=status, solver=solver_id,
                         repeats=nprof)                                           # (untimed: steady state)
    k_s = sorted(ops.enhance_profiled(x, u, M, GAMMA, n, global_domain=gd, out=W, status=status,
                                      solver=solver_id, repeats=nprof))
    k_avg, k_med = sum(k_s) / len(k_s), k_s[len(k_s) // 2]
    k_iso = sorted(ops.enhance_profiled(x, u, M, GAMMA, n, global_domain=gd, out=W, status=status, solver=solver_id)
                   for _ in range(20 if full else 8))
    wl.plans[0].launch(st)                       # (leave the step's result in W for the accuracy block)
    torch.cuda.synchronize()

    flops = algorithmic_flops_dual(M, n) if dual else algorithmic_flops(M, n)
    byts = algorithmic_bytes(M)
    step_kernel, enh_kernel, pipe, solver_lbl, count_key = poisson_labels(M, n, dual)
    ach = flops * ne_loc / step_s / 1e12
    out = {
        "metric": "LSSVR-enhanced elements/sec, 1D Poisson deg-%d/%d-pt" % (degree, n),
        **timing_fields(t, steps, ne_glob),
        "unit": "elements/s",
        "n_gpus": 1,
        "steps": steps,
        "warmup": warmup,
        "prewarm_steps": PREWARM_DONE,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": wl.describe(degree) + label,
            "elements_per_gpu": ne_loc,
            "elements_total": ne_glob,
            "parallelism": "one rank",
            "solver": solver_lbl,
            "fallback_elements": n_fallback,
            "timing": "HIP events around exactly K steps on the launch stream, barrier + device synchronisation on "
                      "both sides, repeated; median reported",
            "timed_region": region_text(t.mode),
        },
        "roofline": {
            "bound": "fp64-valu",
            "pipe": pipe,
            "kernel": step_kernel,
            "achieved": ach,
            "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": ach / FP64_PEAK_TFLOPS,
            "flops_per_element": flops,
            "flops_formula": "SURVEY.md 8(d) " + ("dual" if dual else "primal") + " form (direct Gram); the kernel's "
                             "Chebyshev-moment Gram executes fewer (DESIGN.md section 2b)",
            "achieved_is": "direct-Gram-EQUIVALENT TFLOP/s (algorithmic flops of SURVEY.md 8(d) x elements per launch / "
                           "kernel_us_avg), not executed flops: see `executed`",
            "elements_per_launch": ne_loc,
            "kernel_us_avg": step_s * 1e6,
            "kernel_us_is": "launch-to-launch duration of the step's kernel(s) INSIDE THE TIMED REGION = the median "
                            "bracket / K (HIP events on the launch stream): the kernel's own duration plus the "
                            "boundary to the next dependent launch; rocprofv3 --kernel-trace --stats of the same "
                            "command (profiles/) reads the kernel's begin -> end stamps, which must agree to a few %",
            "executed": executed_fraction(count_key, ne_loc, step_s,
                                          fallback_key=("small_M%d_n%d_sin" % (M, n)) if M <= 22 else "large_pair_M%d_n%d" % (M, n)),
            "enhancement_only": {
                "kernel": enh_kernel,
                "what": "the enhancement WITHOUT the assembly blocks, begin -> end stamps of the dispatch "
                        "(hipExtLaunchKernelGGL events = rocprofv3's kernel duration)",
                "kernel_us_in_sequence_avg": k_avg * 1e6,
                "kernel_us_in_sequence_median": k_med * 1e6,
                "kernel_us_isolated_avg": sum(k_iso) / len(k_iso) * 1e6,
                "launches": nprof,
                "frac_in_sequence": flops * ne_loc / k_avg / 1e12 / FP64_PEAK_TFLOPS,
            },
            "traffic": None,
            "traffic_source": None,
        },
        "roofline_hbm": {
            "bound": "hbm",
            "achieved": byts * ne_loc / step_s / 1e9,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": byts * ne_loc / step_s / 1e9 / HBM_PEAK_GBS,
            "bytes_per_element": byts,
            "traffic": None,
        },
    }
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf) and not dual:
        try:
            tj = json.load(open(tf))
            key = "step_M%d_n%d_ne%d" % (M, n, ne_loc)
            tr = tj.get(key) or tj.get("M%d_n%d_ne%d" % (M, n, ne_loc))
            rnd = (tj.get("_round_r04") if key in tj else None) or tj.get("_round", "committed profile")
            if tr:
                out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                out["roofline_hbm"]["traffic"] = tr["hbm_bytes_per_launch"]
                if "algorithmic_bytes_per_launch_with_the_p1_bands" in tr:
                    out["roofline_hbm"]["algorithmic_bytes_per_launch_with_the_p1_bands"] = \
                        tr["algorithmic_bytes_per_launch_with_the_p1_bands"]
                out["roofline"]["traffic_source"] = ("profiles/traffic.json (%s): %s; NOT measured in this run"
                                                     % (rnd,
                                                        tr.get("what", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                                                       "at this size, calibrated")))
        except Exception:
            pass
    out["accuracy"] = poisson_accuracy(wl, W, M, n, dev, nsel=9 if full else 5, ntruth=5 if (full and M <= 22) else 3)
    if cpu_res is not None:
        out["cpu_baseline"] = cpu_res
    if not full:
        return out

    try:
        out["roofline"]["fp64_fma_probe_tflops"] = round(ops.fp64_probe(8192, 4096, False), 2)
        if M > 22 or dual:
            out["roofline"]["fp64_mfma_4x4x4_probe_tflops"] = round(ops.fp64_probe(8192, 2048, 3), 2)
            out["roofline"]["fp64_mfma_16x16x4_probe_tflops"] = round(ops.fp64_probe(8192, 1024, 1), 2)
    except Exception as exc:  # pragma: no cover
        out["roofline"]["fp64_fma_probe_tflops"] = "failed: %s" % exc

    # RCCL bring-up with one rank (LSSVR_BENCH_FORCE_DIST=1): both stitches once
    if use_dist:
        try:
            wl2 = Workload(ne_glob, lo, hi, M, n, 0, 1, dev, nbuf=2)
            out["forced_dist_single_rank"] = {w: timed_stitch(wl2, D, 5, 2, w, a) for w in ("u", "W")
                                              for a in ("collective", "pairs")}
            del wl2
        except Exception as exc:  # pragma: no cover
            out["forced_dist_single_rank"] = {"error": repr(exc)}

    if t_other is not None:
        eager = t_other.mode.startswith("eager")
        other = {"what": ("the same K steps issued call by call from Python (one stream)" if eager
                          else "the same K steps captured once in a hipGraph (one stream) and replayed"),
                 "mode_used": t_other.mode, **timing_fields(t_other, steps, ne_glob), "unit": "elements/s",
                 "results_equal": other_equal}
        other.pop("value_is")
        out["eager_loop" if eager else "graph_replay"] = other
        out["config"]["timed_region"] += ("; the faster of the two ways of issuing the K steps (lower median), the other "
                                         "one is `%s`" % ("eager_loop" if eager else "graph_replay"))
    if dual:
        return out

    # the same K steps issued round-robin on two HIP streams with separate output buffers: at
    # 1e5 elements one launch fills only ~60 % of the chip's wave slots, so independent batches
    # overlap.  Reported beside `value` (which is strictly sequential on one stream).
    try:
        nstream = 2
        streams = [torch.cuda.Stream(device=dev) for _ in range(nstream)]
        wp = [Workload(ne_glob, lo, hi, M, n, 0, 1, dev) for _ in range(nstream)]
        torch.cuda.synchronize()
        kp = max(steps, 200)                  # (host clock: enough launches for the closing synchronisation not to count)
        for i in range(max(warmup, 20)):
            wp[i % nstream].plans[0].launch(streams[i % nstream].cuda_stream)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for i in range(kp):
            wp[i % nstream].plans[0].launch(streams[i % nstream].cuda_stream)
        torch.cuda.synchronize()
        tp = time.perf_counter() - tp
        out["pipelined"] = {"what": "%d steps, round-robin on %d streams, separate W buffers (host clock)" % (kp, nstream),
                            "streams": nstream, "value": ne_loc * kp / tp, "unit": "elements/s",
                            "ms_per_step": tp / kp * 1e3,
                            "results_equal": bool(torch.equal(wp[0].W[0], W) and torch.equal(wp[1].W[0], W))}
        del wp
    except Exception as exc:  # pragma: no cover
        out["pipelined"] = {"error": repr(exc)}

    # the uniform-mesh shortcut (lssvr_enhance_shared; SURVEY.md 8(d): "reported as a separate line
    # if built"): one shared operator applied per element.  Never part of `value`; its own
    # roofline is HBM (88 B per element against ~8 TB/s).
    try:
        op = ops.build_shared_operator((hi - lo) / ne_glob, M, GAMMA, n, device=dev)
        Ws = torch.empty((ne_loc, M), dtype=torch.float64, device=dev)
        ts = sorted(ops.enhance_shared(x, u, op, M, n, global_domain=gd, out=Ws, status=status, profiled=True)
                    for _ in range(min(max(steps, 20), 50)))
        t_sh = sum(ts) / len(ts)
        diff = (Ws - W).double()
        rel = float((diff.pow(2).sum(1).sqrt() / W.pow(2).sum(1).sqrt().clamp_min(1e-300)).max().item())
        out["shared_operator"] = {
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
        wl.plans[0].launch(st)
        torch.cuda.synchronize()
    except Exception as exc:  # pragma: no cover
        out["shared_operator"] = {"error": repr(exc)}

    # the stages around the hot path, each timed on its own (t_global_solve,
    # t_eval, H2D/D2H are reported separately and never enter `value`)
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
        uq_pts = torch.linspace(float(wl.nodes_h[0]), float(wl.nodes_h[-1]), 2 * ne_loc + 1,
                                dtype=torch.float64, device=dev)
        x_pin = torch.as_tensor(wl.nodes_h).pin_memory()
        W_pin = torch.empty((ne_loc, M), dtype=torch.float64).pin_memory()
        stages = {
            "note": "median of 20, microseconds, event pairs on the launch stream (each includes "
                    "~4 us of event + dispatch latency)",
            "p1_assemble_us": med_us(lambda: ops.p1_assemble(x, 2, out=bl)),
            "dirichlet_solve_bands_us": med_us(lambda: ops.tridiag_dirichlet_solve(bl["diag"], bl["off"], bl["load"])),
            "dirichlet_solve_flux_us": med_us(lambda: ops.p1_flux_solve(bl["kloc"], bl["load"])),
            "enhance_us": med_us(lambda: ops.enhance(x, u, M, GAMMA, n, global_domain=gd, out=W, status=status,
                                                      solver=solver_id)),
            "evaluate_2ne_points_us": med_us(lambda: ops.evaluate(x, W, uq_pts, want_elem=False)),
            "h2d_nodes_and_values_us": med_us(lambda: (x.copy_(x_pin, non_blocking=True),
                                                      u.copy_(x_pin, non_blocking=True))),
            "d2h_coefficients_us": med_us(lambda: W_pin.copy_(W, non_blocking=True)),
        }
        x.copy_(torch.as_tensor(wl.nodes_h))
        u.copy_(torch.as_tensor(wl.values_h))
        wl.plans[0].launch(st)
        torch.cuda.synchronize()
        hd = stages["h2d_nodes_and_values_us"] + stages["d2h_coefficients_us"]
        stages["pcie_inclusive_elements_per_s"] = ne_loc / (step_s + hd * 1e-6)
        out["stages"] = stages
    except Exception as exc:  # pragma: no cover
        out["stages"] = {"error": repr(exc)}
    return out


if __name__ == "__main__":
    main()
