"""Device operators: thin, typed wrappers over the C ABI.

PyTorch-ROCm tensors are used as device buffers only (``data_ptr()`` + the
current HIP stream); all arithmetic happens in the hand-written gfx950 kernels
behind ``liblssvr_hip.so``.  Every function requires float64 CUDA tensors and
raises otherwise -- there is no CPU path.
"""
from __future__ import annotations

import math
import threading

import torch

from . import _capi
from ._capi import (RHS_ARRAY, RHS_ARRAY_PM, RHS_SIN, SOLVER_DUAL, SOLVER_PRIMAL,  # noqa: F401
                    SOLVER_PRIMAL_MOMENT, SOLVER_PRIMAL_WAVE, TABLE_ELEMENT_MAJOR, TABLE_POINT_MAJOR)

SOLVER_SHARED = 100      # facade-level choice for UNIFORM meshes: routed to lssvr_enhance_shared

POISSON_AMP = float(math.pi ** 2)     # Dual.py:12  np.pi**2
POISSON_OMEGA = float(math.pi)


def _dev(t, name, dtype=torch.float64):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor device buffer, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: must live in MI355X device memory (got {t.device}); "
                           "the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _check_buffers(ne, M, n_colloc, x, *, out=None, status=None, fail_count=None, rhs_values=None,
                   n_rhs_rows=None):
    """Caller-supplied buffers become raw device pointers: a wrong-sized tensor would be an
    out-of-bounds device access, so sizes are checked here.  Returns (out, status), allocated
    when not given."""
    if out is None:
        out = torch.empty((ne, M), dtype=torch.float64, device=x.device)
    else:
        _dev(out, "out")
        if out.numel() != ne * M:
            raise ValueError(f"out must hold ne*M = {ne * M} doubles, got {out.numel()}")
    if status is None:
        status = torch.empty((ne,), dtype=torch.int32, device=x.device)
    else:
        _dev(status, "status", torch.int32)
        if status.numel() != ne:
            raise ValueError(f"status must hold ne = {ne} int32, got {status.numel()}")
    if fail_count is not None:
        _dev(fail_count, "fail_count", torch.int32)
        if fail_count.numel() < 1:
            raise ValueError("fail_count must hold one int32")
    if rhs_values is not None:
        _dev(rhs_values, "rhs_values")
        rows = ne if n_rhs_rows is None else n_rhs_rows
        if rhs_values.numel() != rows * n_colloc:
            raise ValueError(f"rhs_values must hold {rows}*n_colloc = {rows * n_colloc} doubles, "
                             f"got {rhs_values.numel()}")
    return out, status


def _stream(stream):
    if stream is None:
        return torch.cuda.current_stream().cuda_stream
    if isinstance(stream, torch.cuda.Stream):
        return stream.cuda_stream
    return int(stream)


_WORK = {}                     # device index -> {"buf", "event", "stream"}: ONE shared scratch buffer per device
_WORK_LOCK = threading.Lock()


def _torch_stream(handle, device):
    """torch view of a raw hipStream_t handle (0 = the device's default stream)."""
    if not handle:
        return torch.cuda.default_stream(device)
    return torch.cuda.ExternalStream(int(handle), device=device)


class workspace:
    """Context manager around ONE launch that needs the scratch of ``lssvr_enhance_ws``
    (``lssvr_enhance_work_bytes``; None inside the ``with`` when no workspace is needed).

    One buffer per DEVICE, grown on demand, shared by every stream: a launch on another stream than the previous
    user first waits (on the device, ``hipStreamWaitEvent``) for the event recorded after that user's launch, so
    launches above M = 22 that go through this default workspace are serialised ACROSS streams (pass your own
    ``work=`` buffers, as :class:`StepPlan` does, for concurrent streams).  Memory: 96-128 doubles per element of
    the largest launch so far (0.8-1 GB per 1e6 elements), once per device -- round 3 kept one such buffer per
    (device, stream handle) for the life of the process, and a destroyed-and-reused handle could alias another
    stream's buffer.  The lock is held from hand-out to the recorded event, so host threads cannot
    interleave between the two."""

    def __init__(self, lib, device, ne, M, n_colloc, solver, stream=None):
        self.nbytes = int(lib.lssvr_enhance_work_bytes(int(ne), int(M), int(n_colloc), int(solver)))
        self.device, self.handle, self.locked = device, _stream(stream), False

    def __enter__(self):
        if self.nbytes <= 0:
            return None
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _WORK_LOCK.acquire()
        self.locked = True
        ent = _WORK.get(idx)
        self.ts = _torch_stream(self.handle, self.device)
        if ent is not None and ent["event"] is not None and ent["stream"] != self.handle:
            self.ts.wait_event(ent["event"])            # the previous user's kernels are done before ours start
        if ent is None or ent["buf"].numel() * 8 < self.nbytes:
            # (a replaced buffer's last use is ordered before this stream's work by the wait above)
            if ent is not None:
                ent["buf"].record_stream(self.ts)
            ent = {"buf": torch.empty((self.nbytes + 7) // 8, dtype=torch.float64, device=self.device), "event": None}
            _WORK[idx] = ent
        ent["stream"] = self.handle
        self.ent = ent
        return ent["buf"]

    def __exit__(self, *exc):
        if self.locked:
            try:
                ev = torch.cuda.Event()
                ev.record(self.ts)
                self.ent["event"] = ev
            finally:
                self.locked = False
                _WORK_LOCK.release()
        return False


def release_workspaces():
    """Drop every cached workspace (one per device otherwise, for the life of the process).  The device is
    synchronised first: no stream handle that may have been destroyed meanwhile is touched."""
    with _WORK_LOCK:
        if _WORK:
            torch.cuda.synchronize()
        _WORK.clear()


def _work_arg(work, lib, x, ne, M, n_colloc, solver, stream):
    """(context manager yielding the workspace tensor or None) for the ``work=`` convention of the wrappers:
    None = the shared per-device buffer, False = no workspace, a tensor = the caller's own."""
    import contextlib
    if work is None:
        return workspace(lib, x.device, ne, M, n_colloc, solver, stream)
    if work is False:
        return contextlib.nullcontext(None)
    _dev(work, "work")
    return contextlib.nullcontext(work)


def enhance(x, u, M, gamma, n_colloc=12, *, rhs=(POISSON_AMP, POISSON_OMEGA), rhs_values=None,
            elem_offset=0, ne_global=None, global_domain=None, bc=(0.0, 0.0),
            solver=SOLVER_PRIMAL, out=None, status=None, fail_count=None, stream=None, work=None,
            point_major=False):
    """``solve_lssvr_subproblems`` (Dual.py:139-169) for the shard (x, u).

    x, u: float64[ne+1] device tensors.  Returns (W float64[ne, M], status int32[ne]).
    ``rhs`` = (amp, omega) evaluates f = amp*sin(omega*x) in-kernel; ``rhs_values``
    float64[ne, n_colloc] (f tabulated at ``colloc_points``) overrides it;
    ``point_major=True``: ``rhs_values`` is float64[n_colloc, ne] instead (f tabulated at
    ``colloc_points(..., point_major=True)``) -- the layout the lane kernels (M <= 22) read at
    full HBM rate.
    ``work``: device scratch for the two-kernel path above M = 22 (default: the buffer
    :func:`workspace` keeps for this device AND stream; ``work=False`` runs the workspace-free
    kernels -- above M = 22 the single f64-MFMA kernel, about half the speed).  A caller-supplied
    ``work`` must not be shared by launches that can run concurrently.
    """
    lib = _capi.load()
    _dev(x, "x")
    _dev(u, "u")
    if x.numel() != u.numel() or x.dim() != 1:
        raise ValueError("x and u must be 1-D with equal length ne+1")
    ne = x.numel() - 1
    if ne < 0:
        raise ValueError("need at least one node")
    if ne_global is None:
        ne_global = elem_offset + ne
    if global_domain is None:
        if ne == 0:
            global_domain = (0.0, 0.0)
        else:
            ends = torch.stack([x[0], x[-1]]).cpu()
            global_domain = (float(ends[0]), float(ends[1]))
    out, status = _check_buffers(ne, M, n_colloc, x, out=out, status=status, fail_count=fail_count,
                                 rhs_values=rhs_values)
    if rhs_values is not None:
        rhs_id, params = (RHS_ARRAY_PM if point_major else RHS_ARRAY), None
    else:
        rhs_id, params = RHS_SIN, _capi.rhs_params(*rhs)
    with _work_arg(work, lib, x, ne, M, n_colloc, solver, stream) as wk:
        rc = lib.lssvr_enhance_ws(_ptr(x), _ptr(u), ne, int(elem_offset), int(ne_global),
                                  float(global_domain[0]), float(global_domain[1]),
                                  float(bc[0]), float(bc[1]), int(M), int(n_colloc), float(gamma),
                                  rhs_id, params, _ptr(rhs_values), int(solver),
                                  _ptr(out), _ptr(status), _ptr(fail_count),
                                  _ptr(wk), 0 if wk is None else wk.numel() * 8, _stream(stream), None)
    _capi.check(rc, "lssvr_enhance_ws")
    return out, status


def enhance_subset(x, u, M, gamma, n_colloc, W, *, elem_ids=None, gamma_values=None,
                   rhs=(POISSON_AMP, POISSON_OMEGA), rhs_values=None, elem_offset=0, ne_global=None,
                   global_domain, bc=(0.0, 0.0), status=None, fail_count=None, stream=None,
                   point_major=False, work=None):
    """``lssvr_enhance_subset_ws``: the elements ``elem_ids`` (int64 device tensor of mesh indices;
    None = all) of the shard (x, u) with one (M, n_colloc); per-element ``gamma_values``
    (float64[ne], indexed by mesh element) optional.  Rows go to ``W[id, :M]`` of the caller's
    float64[ne, ldw] array (ldw = W.shape[1] >= M; zero it first when ldw > M).  ``work``: as in
    :func:`enhance` (above M = 22 the group runs as the moment / solve kernel pair; ``False``: the
    single f64-MFMA kernel)."""
    lib = _capi.load()
    _dev(x, "x")
    _dev(u, "u")
    _dev(W, "W")
    ne = x.numel() - 1
    if W.dim() != 2 or W.shape[0] != ne or W.shape[1] < M or not W.is_contiguous():
        raise ValueError("W must be a contiguous float64[ne, ldw >= M] tensor")
    if elem_ids is not None:
        _dev(elem_ids, "elem_ids", torch.int64)
        nsub = elem_ids.numel()
    else:
        nsub = ne
    if gamma_values is not None:
        _dev(gamma_values, "gamma_values")
        if gamma_values.numel() != ne:
            raise ValueError("gamma_values must hold one value per mesh element")
    if status is not None:
        _dev(status, "status", torch.int32)
        if status.numel() != ne:
            raise ValueError("status is indexed by mesh element: int32[ne]")
    if fail_count is not None:
        _dev(fail_count, "fail_count", torch.int32)
    if ne_global is None:
        ne_global = elem_offset + ne
    if rhs_values is not None:
        _dev(rhs_values, "rhs_values")
        if rhs_values.numel() != nsub * n_colloc:
            raise ValueError("rhs_values must hold nsub*n_colloc doubles (indexed by position in elem_ids)")
        rhs_id, params = (RHS_ARRAY_PM if point_major else RHS_ARRAY), None
    else:
        rhs_id, params = RHS_SIN, _capi.rhs_params(*rhs)
    with _work_arg(work, lib, x, nsub, M, n_colloc, SOLVER_PRIMAL, stream) as wk:
        rc = lib.lssvr_enhance_subset_ws(_ptr(x), _ptr(u), ne, _ptr(elem_ids), int(nsub), int(elem_offset),
                                         int(ne_global), float(global_domain[0]), float(global_domain[1]),
                                         float(bc[0]), float(bc[1]), int(M), int(n_colloc), float(gamma),
                                         _ptr(gamma_values), rhs_id, params, _ptr(rhs_values),
                                         _ptr(W), int(W.shape[1]), _ptr(status), _ptr(fail_count),
                                         _ptr(wk), 0 if wk is None else wk.numel() * 8, _stream(stream))
    _capi.check(rc, "lssvr_enhance_subset_ws")
    return W


def build_shared_operator(h, M, gamma, n_colloc, *, device="cuda:0", stream=None):
    """Response table of ONE canonical element of length ``h`` for :func:`enhance_shared`:
    float64[n_colloc + 2, M], built by the general per-element kernel (``lssvr_enhance``) on a
    (n_colloc + 2)-element auxiliary mesh of the same spacing centred on 0 -- unit right-hand
    sides for rows k < n, unit boundary values for the last two rows."""
    n = int(n_colloc)
    h = float(h)
    nel = n + 2
    nodes = (torch.arange(nel + 1, dtype=torch.float64) - 0.5 * nel) * h
    scl = 2.0 / float(nodes[1] - nodes[0])
    f = torch.zeros((nel, n), dtype=torch.float64)
    f[torch.arange(n), torch.arange(n)] = scl * scl          # f~ = f / scl^2 = e_k
    uu = torch.zeros(nel + 1, dtype=torch.float64)
    uu[n + 1] = 1.0                # element n: (g_l, g_r) = (0, 1); element n+1: (1, 0)
    x = nodes.to(device)
    W, st = enhance(x, uu.to(device), int(M), float(gamma), n, rhs_values=f.to(device),
                    elem_offset=1, ne_global=nel + 2, global_domain=(float(nodes[0]) - h, float(nodes[-1]) + h),
                    stream=stream)
    op = torch.empty((nel, M), dtype=torch.float64, device=device)
    op[:n] = W[:n]
    op[n] = W[n + 1]               # response to g_l
    op[n + 1] = W[n]               # response to g_r
    if int(st.sum().item()) != 0:
        raise _capi.LssvrHipError("build_shared_operator: the canonical element's factorisation broke down")
    return op


def enhance_shared(x, u, op, M, n_colloc, *, rhs=(POISSON_AMP, POISSON_OMEGA), rhs_values=None,
                   elem_offset=0, ne_global=None, global_domain, bc=(0.0, 0.0), out=None, status=None,
                   fail_count=None, stream=None, profiled=False, point_major=False):
    """``lssvr_enhance_shared`` (uniform meshes; the caller vouches for uniformity).  Returns
    (W, status), or the kernel duration in seconds when ``profiled``."""
    import ctypes
    lib = _capi.load()
    _dev(x, "x")
    _dev(u, "u")
    _dev(op, "op")
    ne = x.numel() - 1
    if x.numel() != u.numel() or x.dim() != 1:
        raise ValueError("x and u must be 1-D with equal length ne+1")
    if op.dim() != 2 or op.shape[0] != n_colloc + 2 or op.shape[1] != M or not op.is_contiguous():
        raise ValueError("op must be a contiguous float64[(n_colloc+2), M] tensor")
    if ne_global is None:
        ne_global = elem_offset + ne
    out, status = _check_buffers(ne, M, n_colloc, x, out=out, status=status, fail_count=fail_count,
                                 rhs_values=rhs_values)
    if rhs_values is not None:
        rhs_id, params = (RHS_ARRAY_PM if point_major else RHS_ARRAY), None
    else:
        rhs_id, params = RHS_SIN, _capi.rhs_params(*rhs)
    ms = ctypes.c_float(0.0)
    rc = lib.lssvr_enhance_shared(_ptr(x), _ptr(u), ne, int(elem_offset), int(ne_global),
                                  float(global_domain[0]), float(global_domain[1]),
                                  float(bc[0]), float(bc[1]), int(M), int(n_colloc),
                                  rhs_id, params, _ptr(rhs_values), _ptr(op),
                                  _ptr(out), _ptr(status), _ptr(fail_count), _stream(stream),
                                  ctypes.byref(ms) if profiled else None)
    _capi.check(rc, "lssvr_enhance_shared")
    if profiled:
        return ms.value * 1e-3
    return out, status


def enhance_profiled(x, u, M, gamma, n_colloc=12, *, rhs=(POISSON_AMP, POISSON_OMEGA),
                     elem_offset=0, ne_global=None, global_domain, bc=(0.0, 0.0),
                     solver=SOLVER_PRIMAL, out=None, status=None, stream=None, work=None, repeats=None):
    """Same launch as :func:`enhance` but BLOCKING and stamped with the dispatch's own
    begin/end timestamps; returns the kernel duration in seconds (roofline measurement; on the
    two-kernel path above M = 22 the duration of the pair, gap included).  ``repeats=k``: k launches
    back to back, one synchronisation at the end (``lssvr_enhance_ws_sequence``): the list of the k
    durations inside a running sequence instead of one duration in isolation."""
    import ctypes
    lib = _capi.load()
    _dev(x, "x")
    _dev(u, "u")
    ne = x.numel() - 1
    if x.numel() != u.numel() or x.dim() != 1:
        raise ValueError("x and u must be 1-D with equal length ne+1")
    if ne_global is None:
        ne_global = elem_offset + ne
    out, status = _check_buffers(ne, M, n_colloc, x, out=out, status=status)
    ms = ctypes.c_float(0.0)
    with _work_arg(work, lib, x, ne, M, n_colloc, solver, stream) as wk:
        return _enhance_profiled(lib, x, u, ne, elem_offset, ne_global, global_domain, bc, M, n_colloc, gamma, rhs,
                                 solver, out, status, wk, stream, repeats, ms)


def _enhance_profiled(lib, x, u, ne, elem_offset, ne_global, global_domain, bc, M, n_colloc, gamma, rhs, solver,
                      out, status, work, stream, repeats, ms):
    import ctypes
    if repeats is not None:
        arr = (ctypes.c_float * int(repeats))()
        rc = lib.lssvr_enhance_ws_sequence(_ptr(x), _ptr(u), ne, int(elem_offset), int(ne_global),
                                           float(global_domain[0]), float(global_domain[1]),
                                           float(bc[0]), float(bc[1]), int(M), int(n_colloc), float(gamma),
                                           RHS_SIN, _capi.rhs_params(*rhs), None, int(solver),
                                           _ptr(out), _ptr(status), None,
                                           _ptr(work), 0 if work is None else work.numel() * 8, _stream(stream),
                                           int(repeats), arr)
        _capi.check(rc, "lssvr_enhance_ws_sequence")
        return [v * 1e-3 for v in arr]
    rc = lib.lssvr_enhance_ws(_ptr(x), _ptr(u), ne, int(elem_offset), int(ne_global),
                              float(global_domain[0]), float(global_domain[1]),
                              float(bc[0]), float(bc[1]), int(M), int(n_colloc), float(gamma),
                              RHS_SIN, _capi.rhs_params(*rhs), None, int(solver),
                              _ptr(out), _ptr(status), None,
                              _ptr(work), 0 if work is None else work.numel() * 8, _stream(stream),
                              ctypes.byref(ms))
    _capi.check(rc, "lssvr_enhance_ws(profiled)")
    return ms.value * 1e-3


def _bind(lib, name, args):
    """(bound foreign function, arguments pre-converted to the ctypes of its signature): a plan is launched many
    times, and converting ~20 Python values per call is a third of the host's ~7 us per launch of an ~8 us step."""
    import ctypes
    argtypes = _capi.SIGNATURES[name][1]
    ready = (ctypes.Array, ctypes._SimpleCData, ctypes._Pointer, type(ctypes.byref(ctypes.c_int())))
    conv = tuple(a if (a is None or isinstance(a, ready)) else tp(a) for tp, a in zip(argtypes, args))
    return getattr(lib, name), conv


class StepPlan:
    """One step of the hot path (element-local P1 assembly + per-element enhancement of
    the same resident mesh shard) bound once, launched many times: the argument tuple of
    ``lssvr_step`` is built at construction so that a launch is a single ctypes call."""

    def __init__(self, x, u, M, gamma, n_colloc=12, *, rhs=(POISSON_AMP, POISSON_OMEGA), nquad=2,
                 elem_offset=0, ne_global=None, global_domain, bc=(0.0, 0.0), bands=None,
                 out=None, status=None, fail_count=None):
        self.lib = _capi.load()
        _dev(x, "x")
        _dev(u, "u")
        ne = x.numel() - 1
        dev = x.device
        if ne_global is None:
            ne_global = elem_offset + ne
        self.bands = bands or {
            "diag": torch.empty(ne + 1, dtype=torch.float64, device=dev),
            "off": torch.empty(ne, dtype=torch.float64, device=dev),
            "load": torch.empty(ne + 1, dtype=torch.float64, device=dev),
        }
        if x.numel() != u.numel() or x.dim() != 1:
            raise ValueError("x and u must be 1-D with equal length ne+1")
        for k, cnt in (("diag", ne + 1), ("off", ne), ("load", ne + 1)):
            _dev(self.bands[k], k)
            if self.bands[k].numel() != cnt:
                raise ValueError(f"bands[{k!r}] must hold {cnt} doubles")
        self.W, self.status = _check_buffers(ne, M, n_colloc, x, out=out, status=status,
                                             fail_count=fail_count)
        self.fail_count = fail_count
        # above M = 22 lssvr_step is assembly + the workspace-free enhancement kernel (two launches);
        # with a workspace the enhancement runs as the faster moment / solve pair: three launches
        # (a private buffer: plans may run concurrently on different streams)
        nb = int(self.lib.lssvr_enhance_work_bytes(ne, int(M), int(n_colloc), SOLVER_PRIMAL))
        self._work = torch.empty((nb + 7) // 8, dtype=torch.float64, device=dev) if nb > 0 else None
        self._keep = (x, u, _capi.rhs_params(*rhs))
        self._args = (_ptr(x), _ptr(u), ne, int(elem_offset), int(ne_global),
                      float(global_domain[0]), float(global_domain[1]), float(bc[0]), float(bc[1]),
                      int(M), int(n_colloc), float(gamma), self._keep[2], int(nquad),
                      _ptr(self.bands["diag"]), _ptr(self.bands["off"]), _ptr(self.bands["load"]),
                      _ptr(self.W), _ptr(self.status), _ptr(fail_count))
        self._step, self._cargs = _bind(self.lib, "lssvr_step", self._args)
        # the library's own plan (lssvr_step_plan_*): the 20 arguments validated and bound once on the C side, a
        # launch is a two-argument call (host cost 4.4 -> ~2 us per launch; with the stream handle passed in, since
        # torch.cuda.current_stream() alone costs 3 us)
        import ctypes
        self._handle = ctypes.c_void_p()
        create, cargs = _bind(self.lib, "lssvr_step_plan_create", (ctypes.byref(self._handle),) + self._args)
        rc = create(*cargs)
        if rc < 0:
            self._handle = None
            _capi.check(rc, "lssvr_step_plan_create")
        self._launch = self.lib.lssvr_step_plan_launch

        if self._work is not None:
            self._asm_args = (_ptr(x), ne, int(nquad), RHS_SIN, self._keep[2], None, None,
                              _ptr(self.bands["diag"]), _ptr(self.bands["off"]), _ptr(self.bands["load"]),
                              None, None)
            self._enh_args = (_ptr(x), _ptr(u), ne, int(elem_offset), int(ne_global),
                              float(global_domain[0]), float(global_domain[1]), float(bc[0]), float(bc[1]),
                              int(M), int(n_colloc), float(gamma), RHS_SIN, self._keep[2], None,
                              SOLVER_PRIMAL, _ptr(self.W), _ptr(self.status), _ptr(fail_count),
                              _ptr(self._work), self._work.numel() * 8)

    def launch(self, stream=None):
        st = _stream(stream)
        if self._work is not None:
            rc = self.lib.lssvr_p1_assemble(*self._asm_args, st)
            if rc >= 0:
                rc = self.lib.lssvr_enhance_ws(*self._enh_args, st, None)
        else:
            rc = self._launch(self._handle, st)
        if rc < 0:
            _capi.check(rc, "lssvr_step")
        return self.W, self.status

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                self.lib.lssvr_step_plan_destroy(h)
            except Exception:
                pass
            self._handle = None


def enhance_varcoef(x, u, M, gamma, n_colloc, a_values, da_values, rhs_values, *, elem_offset=0,
                    ne_global=None, global_domain=None, bc=(0.0, 0.0), out=None, status=None,
                    fail_count=None, stream=None, profiled=False, point_major=False, repeats=None):
    """BASELINE config 5: rows -a (2/h)^2 L'' - a' (2/h) L' (no reference counterpart; the
    operator it generalises is Dual.py:43-44).  ``profiled``: BLOCKING, returns the launch
    duration in seconds (the dispatch's own begin / end stamps) instead of (W, status); with
    ``repeats=k`` the list of the durations of k launches back to back, one synchronisation at the end.
    ``point_major``: the three tables are float64[n_colloc, ne] (``t[k, e]``) instead of
    float64[ne, n_colloc] -- see :func:`colloc_points`; the fast layout for M <= 22."""
    import ctypes
    lib = _capi.load()
    _dev(x, "x")
    _dev(u, "u")
    ne = x.numel() - 1
    for t, nm in ((a_values, "a_values"), (da_values, "da_values"), (rhs_values, "rhs_values")):
        _dev(t, nm)
        if t.numel() != ne * n_colloc:
            raise ValueError(f"{nm} must hold ne*n_colloc doubles")
    if ne_global is None:
        ne_global = elem_offset + ne
    if global_domain is None:
        ends = torch.stack([x[0], x[-1]]).cpu()
        global_domain = (float(ends[0]), float(ends[1]))
    if x.numel() != u.numel() or x.dim() != 1:
        raise ValueError("x and u must be 1-D with equal length ne+1")
    out, status = _check_buffers(ne, M, n_colloc, x, out=out, status=status, fail_count=fail_count)
    ms = ctypes.c_float(0.0)
    if profiled and repeats is not None:
        arr = (ctypes.c_float * int(repeats))()
        rc = lib.lssvr_enhance_varcoef_ws_sequence(
            _ptr(x), _ptr(u), ne, int(elem_offset), int(ne_global), float(global_domain[0]), float(global_domain[1]),
            float(bc[0]), float(bc[1]), int(M), int(n_colloc), float(gamma), _ptr(a_values), _ptr(da_values),
            _ptr(rhs_values), TABLE_POINT_MAJOR if point_major else TABLE_ELEMENT_MAJOR, _ptr(out), _ptr(status),
            _ptr(fail_count), None, 0, _stream(stream), int(repeats), arr)
        _capi.check(rc, "lssvr_enhance_varcoef_ws_sequence")
        return [v * 1e-3 for v in arr]
    rc = lib.lssvr_enhance_varcoef_ws(_ptr(x), _ptr(u), ne, int(elem_offset), int(ne_global),
                                      float(global_domain[0]), float(global_domain[1]),
                                      float(bc[0]), float(bc[1]), int(M), int(n_colloc), float(gamma),
                                      _ptr(a_values), _ptr(da_values), _ptr(rhs_values),
                                      TABLE_POINT_MAJOR if point_major else TABLE_ELEMENT_MAJOR,
                                      _ptr(out), _ptr(status), _ptr(fail_count), None, 0,
                                      _stream(stream), ctypes.byref(ms) if profiled else None)
    _capi.check(rc, "lssvr_enhance_varcoef_ws")
    if profiled:
        return ms.value * 1e-3
    return out, status


class StepGraph:
    """``steps`` launches of a bound plan (:class:`StepPlan`, :class:`StepPlanVarcoef`) captured ONCE in a
    hipGraph and replayed: a loop of steps on fixed buffers without the host in it.  On the MI355X a replayed
    step of BASELINE config 2 takes 7.7-8.0 us against 8.7 us issued call by call (``bench.py``'s
    ``graph_replay``).  The plan's buffers (x, u, W, status, bands) are the graph's:
    write new nodal values into ``u`` in place (``u.copy_(...)``) and replay."""

    def __init__(self, plan, steps=1):
        self.plan = plan
        self.steps = int(steps)
        side = torch.cuda.Stream(device=plan.W.device)
        side.wait_stream(torch.cuda.current_stream(plan.W.device))
        with torch.cuda.stream(side):
            plan.launch()                               # warm-up outside the capture
        torch.cuda.current_stream(plan.W.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (torch.distributed's watchdog polls events) may keep
        # calling the runtime while this thread captures
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            for _ in range(self.steps):
                plan.launch()

    def replay(self):
        """Enqueue the captured steps on the current stream; returns the plan's (W, status)."""
        self.graph.replay()
        return self.plan.W, self.plan.status


class StepPlanVarcoef:
    """One step of BASELINE config 5 (a-weighted P1 assembly from tabulated quadrature values +
    variable-coefficient enhancement from tabulated a, a', f) bound once, launched many times:
    ``lssvr_step_varcoef`` -- one launch for M <= 12."""

    def __init__(self, x, u, M, gamma, n_colloc, a_values, da_values, rhs_values, rhs_quad, a_quad, *,
                 nquad=2, point_major=False, elem_offset=0, ne_global=None, global_domain, bc=(0.0, 0.0),
                 bands=None, out=None, status=None, fail_count=None):
        self.lib = _capi.load()
        _dev(x, "x")
        _dev(u, "u")
        ne = x.numel() - 1
        dev = x.device
        if x.numel() != u.numel() or x.dim() != 1:
            raise ValueError("x and u must be 1-D with equal length ne+1")
        for t, nm, cnt in ((a_values, "a_values", ne * n_colloc), (da_values, "da_values", ne * n_colloc),
                           (rhs_values, "rhs_values", ne * n_colloc), (rhs_quad, "rhs_quad", ne * nquad),
                           (a_quad, "a_quad", ne * nquad)):
            _dev(t, nm)
            if t.numel() != cnt:
                raise ValueError(f"{nm} must hold {cnt} doubles")
        if ne_global is None:
            ne_global = elem_offset + ne
        self.bands = bands or {
            "diag": torch.empty(ne + 1, dtype=torch.float64, device=dev),
            "off": torch.empty(ne, dtype=torch.float64, device=dev),
            "load": torch.empty(ne + 1, dtype=torch.float64, device=dev),
        }
        for k, cnt in (("diag", ne + 1), ("off", ne), ("load", ne + 1)):
            _dev(self.bands[k], k)
            if self.bands[k].numel() != cnt:
                raise ValueError(f"bands[{k!r}] must hold {cnt} doubles")
        self.W, self.status = _check_buffers(ne, M, n_colloc, x, out=out, status=status, fail_count=fail_count)
        self._keep = (x, u, a_values, da_values, rhs_values, rhs_quad, a_quad, fail_count)
        self._args = (_ptr(x), _ptr(u), ne, int(elem_offset), int(ne_global),
                      float(global_domain[0]), float(global_domain[1]), float(bc[0]), float(bc[1]),
                      int(M), int(n_colloc), float(gamma), _ptr(a_values), _ptr(da_values), _ptr(rhs_values),
                      TABLE_POINT_MAJOR if point_major else TABLE_ELEMENT_MAJOR, int(nquad),
                      _ptr(rhs_quad), _ptr(a_quad),
                      _ptr(self.bands["diag"]), _ptr(self.bands["off"]), _ptr(self.bands["load"]),
                      _ptr(self.W), _ptr(self.status), _ptr(fail_count))
        self._step, self._cargs = _bind(self.lib, "lssvr_step_varcoef", self._args)

    def launch(self, stream=None):
        rc = self._step(*self._cargs, _stream(stream))
        if rc < 0:
            _capi.check(rc, "lssvr_step_varcoef")
        return self.W, self.status


def colloc_points(x, n_colloc, *, stream=None, point_major=False):
    """``np.linspace(x[e], x[e+1], n)`` for every element (Dual.py:40) -> float64[ne, n], or
    float64[n, ne] with ``point_major`` (the same values transposed: tabulate ``rhs_func`` / a / a'
    on it and pass the tables with ``point_major=True``)."""
    lib = _capi.load()
    _dev(x, "x")
    ne = x.numel() - 1
    if point_major:
        xc = torch.empty((n_colloc, ne), dtype=torch.float64, device=x.device)
        _capi.check(lib.lssvr_colloc_points_pm(_ptr(x), ne, int(n_colloc), _ptr(xc), _stream(stream)),
                    "lssvr_colloc_points_pm")
        return xc
    xc = torch.empty((ne, n_colloc), dtype=torch.float64, device=x.device)
    _capi.check(lib.lssvr_colloc_points(_ptr(x), ne, int(n_colloc), _ptr(xc), _stream(stream)),
                "lssvr_colloc_points")
    return xc


def quad_points(x, nquad=2, *, stream=None):
    lib = _capi.load()
    _dev(x, "x")
    ne = x.numel() - 1
    xq = torch.empty((ne, nquad), dtype=torch.float64, device=x.device)
    _capi.check(lib.lssvr_quad_points(_ptr(x), ne, int(nquad), _ptr(xq), _stream(stream)),
                "lssvr_quad_points")
    return xq


def p1_assemble(x, nquad=2, *, rhs=(POISSON_AMP, POISSON_OMEGA), rhs_quad=None, a_quad=None,
                want_local=False, out=None, stream=None):
    """Element-local P1 stiffness/load and the assembled tridiagonal bands
    (Dual.py:117-128).  Returns dict(diag[ne+1], off[ne], load[ne+1][, kloc, floc])."""
    lib = _capi.load()
    _dev(x, "x")
    ne = x.numel() - 1
    dev = x.device
    if out is None:
        out = {
            "diag": torch.empty(ne + 1, dtype=torch.float64, device=dev),
            "off": torch.empty(ne, dtype=torch.float64, device=dev),
            "load": torch.empty(ne + 1, dtype=torch.float64, device=dev),
        }
        if want_local:
            out["kloc"] = torch.empty(ne, dtype=torch.float64, device=dev)
            out["floc"] = torch.empty((ne, 2), dtype=torch.float64, device=dev)
    if rhs_quad is not None:
        _dev(rhs_quad, "rhs_quad")
        if rhs_quad.numel() != ne * nquad:
            raise ValueError("rhs_quad must hold ne*nquad doubles")
        rhs_id, params = RHS_ARRAY, None
    else:
        rhs_id, params = RHS_SIN, _capi.rhs_params(*rhs)
    if a_quad is not None:
        _dev(a_quad, "a_quad")
    rc = lib.lssvr_p1_assemble(_ptr(x), ne, int(nquad), rhs_id, params, _ptr(rhs_quad),
                               _ptr(a_quad), _ptr(out["diag"]), _ptr(out["off"]),
                               _ptr(out["load"]), _ptr(out.get("kloc")), _ptr(out.get("floc")),
                               _stream(stream))
    _capi.check(rc, "lssvr_p1_assemble")
    return out


def tridiag_dirichlet_solve(diag, off, load, u0=0.0, u1=0.0, *, out=None, work=None, stream=None):
    """``enforce`` + ``solve`` (Dual.py:129-130) on the assembled bands -> u[ne+1]."""
    lib = _capi.load()
    _dev(diag, "diag")
    _dev(off, "off")
    _dev(load, "load")
    ne = off.numel()
    if diag.numel() != ne + 1 or load.numel() != ne + 1:
        raise ValueError("band lengths must be ne+1, ne, ne+1")
    if out is None:
        out = torch.empty(ne + 1, dtype=torch.float64, device=diag.device)
    nbytes = lib.lssvr_tridiag_work_bytes(ne)
    if work is None or work.numel() * work.element_size() < nbytes:
        work = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=diag.device)
    rc = lib.lssvr_tridiag_dirichlet_solve(_ptr(diag), _ptr(off), _ptr(load), ne, float(u0),
                                           float(u1), _ptr(out), _ptr(work), _stream(stream))
    _capi.check(rc, "lssvr_tridiag_dirichlet_solve")
    return out


def p1_flux_solve(kloc, load, u0=0.0, u1=0.0, *, out=None, work=None, stream=None):
    """``enforce`` + ``solve`` (Dual.py:129-130) for the assembled P1 system through the
    element-flux prefix scan (A = D^T K D); kloc[ne], load[ne+1] from :func:`p1_assemble`."""
    lib = _capi.load()
    _dev(kloc, "kloc")
    _dev(load, "load")
    ne = kloc.numel()
    if load.numel() != ne + 1:
        raise ValueError("load must have ne+1 entries")
    if out is None:
        out = torch.empty(ne + 1, dtype=torch.float64, device=kloc.device)
    nbytes = lib.lssvr_p1_flux_work_bytes(ne)
    if work is None or work.numel() * work.element_size() < nbytes:
        work = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=kloc.device)
    rc = lib.lssvr_p1_flux_solve(_ptr(kloc), _ptr(load), ne, float(u0), float(u1), _ptr(out),
                                 _ptr(work), _stream(stream))
    _capi.check(rc, "lssvr_p1_flux_solve")
    return out


def p1_flux_aggregate(kloc, load, *, first_global, work=None, stream=None):
    """Shard summary of the flux scan: returns (agg3 device double[3], work) -- see
    ``lssvr_p1_flux_aggregate``; ``work`` must be passed on to :func:`p1_flux_finish`."""
    lib = _capi.load()
    _dev(kloc, "kloc")
    _dev(load, "load")
    ne = kloc.numel()
    if load.numel() < ne + 1:
        raise ValueError("load must have at least ne+1 entries")
    nbytes = lib.lssvr_p1_flux_work_bytes(ne)
    if work is None or work.numel() * work.element_size() < nbytes:
        work = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=kloc.device)
    agg = torch.empty(3, dtype=torch.float64, device=kloc.device)
    rc = lib.lssvr_p1_flux_aggregate(_ptr(kloc), _ptr(load), ne, int(bool(first_global)), _ptr(work),
                                     _ptr(agg), _stream(stream))
    _capi.check(rc, "lssvr_p1_flux_aggregate")
    return agg, work


def p1_flux_finish(kloc, load, work, *, first_global, last_global, prefix=None, grand=None,
                   u0=0.0, u1=0.0, out=None, stream=None):
    """Second half of the sharded flux solve: nodal values u[ne+1] of this shard."""
    lib = _capi.load()
    ne = kloc.numel()
    if out is None:
        out = torch.empty(ne + 1, dtype=torch.float64, device=kloc.device)
    for t, nm in ((prefix, "prefix"), (grand, "grand")):
        if t is not None:
            _dev(t, nm)
    rc = lib.lssvr_p1_flux_finish(_ptr(kloc), _ptr(load), ne, int(bool(first_global)),
                                  int(bool(last_global)), _ptr(work), _ptr(prefix), _ptr(grand),
                                  float(u0), float(u1), _ptr(out), _stream(stream))
    _capi.check(rc, "lssvr_p1_flux_finish")
    return out


def evaluate(x, W, xq, *, want_elem=True, out=None, stream=None):
    """``evaluate_solution`` (Dual.py:176-203) -> (u float64[P], elem int64[P] | None).
    ``out``: optional preallocated float64[P] for u."""
    lib = _capi.load()
    _dev(x, "x")
    _dev(W, "W")
    _dev(xq, "xq")
    ne = x.numel() - 1
    if W.dim() != 2 or W.shape[0] != ne:
        raise ValueError("W must be [ne, M]")
    M = W.shape[1]
    P = xq.numel()
    if out is None:
        uq = torch.empty(P, dtype=torch.float64, device=x.device)
    else:
        uq = _dev(out, "out")
        if uq.numel() != P or not uq.is_contiguous():
            raise ValueError("out must be a contiguous float64[P] tensor")
    elem = torch.empty(P, dtype=torch.int64, device=x.device) if want_elem else None
    rc = lib.lssvr_eval(_ptr(x), _ptr(W), ne, int(M), _ptr(xq), P, _ptr(uq), _ptr(elem),
                        _stream(stream))
    _capi.check(rc, "lssvr_eval")
    return uq, elem


def eval_error(x, W, xq, *, exact=(1.0, math.pi), out=None, stream=None):
    """Device-side error norms of the hybrid solution vs exact(x) = amp*sin(omega*x) on xq
    (Dual.py:216-217 + 8-9): returns a device double[3] = [sum (u-ex)^2, sum ex^2, max |u-ex|]
    (accumulated into ``out`` when given, e.g. across shards)."""
    lib = _capi.load()
    _dev(x, "x")
    _dev(W, "W")
    _dev(xq, "xq")
    ne = x.numel() - 1
    if W.dim() != 2 or W.shape[0] != ne:
        raise ValueError("W must be [ne, M]")
    if out is None:
        out = torch.zeros(3, dtype=torch.float64, device=x.device)
    else:
        _dev(out, "out")
    rc = lib.lssvr_eval_error(_ptr(x), _ptr(W), ne, int(W.shape[1]), _ptr(xq), xq.numel(),
                              _capi.rhs_params(*exact), _ptr(out), _stream(stream))
    _capi.check(rc, "lssvr_eval_error")
    return out


def fp64_probe(blocks=4096, iters=4096, use_mfma=False, *, device="cuda:0", reps=5):
    """Measured FP64 FMA (use_mfma False), v_mfma_f64_16x16x4 (True / 1) or v_mfma_f64_4x4x4_4b (3)
    rate in TFLOP/s -- what the roofline's 78.6 TFLOP/s peak sustains in a pure loop."""
    lib = _capi.load()
    out = torch.empty(blocks * 256, dtype=torch.float64, device=device)
    st = _stream(None)
    _capi.check(lib.lssvr_fp64_probe(_ptr(out), blocks, iters, int(use_mfma), st), "probe")
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    best = float("inf")
    for _ in range(reps):
        e0.record()
        _capi.check(lib.lssvr_fp64_probe(_ptr(out), blocks, iters, int(use_mfma), st), "probe")
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    if int(use_mfma) == 3:
        flops = 8.0 * 512.0 * iters * blocks * 4      # 8 MFMAs/iter/wave, 4 blocks x 4x4x4 x 2 flop
    elif use_mfma:
        flops = 4.0 * 2048.0 * iters * blocks * 4     # 4 MFMAs/iter/wave, 16x16x4x2 flop, 4 waves/block
    else:
        flops = 2.0 * 8.0 * iters * blocks * 256
    return flops / best / 1e12
