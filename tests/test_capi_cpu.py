"""CPU: the C-ABI library loads without a GPU and exports every symbol that
include/lssvr_hip.h declares; argument errors are reported before any HIP call."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lssvr_hip.h")
BENCH_HEADER = os.path.join(ROOT, "include", "lssvr_hip_bench.h")      # measurement entries of the same library
MEASUREMENT_ONLY = ("lssvr_enhance_profiled", "lssvr_enhance_ws_sequence", "lssvr_enhance_varcoef_ws_sequence",
                    "lssvr_fp64_probe", "lssvr_stream_probe", "lssvr_row_chunk_probe")


def _declared_in(path):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lssvr_[a-z0-9_]+)\s*\(", txt)))


def _declared():
    return sorted(set(_declared_in(HEADER)) | set(_declared_in(BENCH_HEADER)))


def test_product_header_holds_no_measurement_entries():
    """Round-3 review: the product ABI header exported the measurement-only entries beside it.  They live in
    include/lssvr_hip_bench.h now (same library); the two headers are disjoint."""
    prod, bench = set(_declared_in(HEADER)), set(_declared_in(BENCH_HEADER))
    assert bench == set(MEASUREMENT_ONLY) and not (prod & bench)
    assert not any(("probe" in n or "profiled" in n or "sequence" in n) for n in prod)


def test_header_symbols_exported_and_bound():
    from hybrid_fem_lssvr_amd import _capi
    lib = _capi.load()
    names = _declared()
    assert "lssvr_enhance" in names and "lssvr_eval" in names and len(names) >= 11
    for nm in names:
        assert hasattr(lib, nm), f"{nm} declared in the header but not exported"
        assert nm in _capi.SIGNATURES, f"{nm} has no ctypes signature"
    assert sorted(_capi.SIGNATURES) == names
    assert lib.lssvr_version() == _capi.ABI_VERSION
    hdr_ver = int(re.search(r"#define LSSVR_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert hdr_ver == _capi.ABI_VERSION


def test_header_constants_match_binding():
    from hybrid_fem_lssvr_amd import _capi
    txt = open(HEADER).read()
    for c_name, py_val in (("LSSVR_RHS_ARRAY", _capi.RHS_ARRAY), ("LSSVR_RHS_SIN", _capi.RHS_SIN),
                           ("LSSVR_SOLVER_PRIMAL", _capi.SOLVER_PRIMAL),
                           ("LSSVR_SOLVER_DUAL", _capi.SOLVER_DUAL),
                           ("LSSVR_SOLVER_PRIMAL_WAVE", _capi.SOLVER_PRIMAL_WAVE),
                           ("LSSVR_ST_OK", _capi.ST_OK), ("LSSVR_ST_FALLBACK", _capi.ST_FALLBACK)):
        m = re.search(r"#define %s\s+(\d+)" % c_name, txt)
        assert m and int(m.group(1)) == py_val, c_name


def test_argument_errors_without_gpu():
    """Validation happens on the host, before any launch: safe to call on a CPU-only box."""
    from hybrid_fem_lssvr_amd import _capi
    lib = _capi.load()
    p = _capi.rhs_params(1.0, 1.0)
    fake = ctypes.c_void_p(4096)
    rc = lib.lssvr_enhance(fake, fake, -1, 0, 0, 0.0, 1.0, 0.0, 0.0, 9, 16, 1e4, 1, p, None, 0,
                           fake, None, None, None)
    assert rc == -2 and b"ne" in lib.lssvr_last_error()
    rc = lib.lssvr_enhance(fake, fake, 10, 0, 10, 0.0, 1.0, 0.0, 0.0, 99, 16, 1e4, 1, p, None, 0,
                           fake, None, None, None)
    assert rc == -3 and b"M = 99" in lib.lssvr_last_error()
    rc = lib.lssvr_enhance(fake, fake, 10, 0, 10, 0.0, 1.0, 0.0, 0.0, 9, 16, 1e4, 7, p, None, 0,
                           fake, None, None, None)
    assert rc == -4
    rc = lib.lssvr_enhance(fake, fake, 10, 0, 10, 0.0, 1.0, 0.0, 0.0, 9, 16, 1e4, 0, None, None, 0,
                           fake, None, None, None)
    assert rc == -4 and b"rhs_values" in lib.lssvr_last_error()
    rc = lib.lssvr_enhance(None, fake, 10, 0, 10, 0.0, 1.0, 0.0, 0.0, 9, 16, 1e4, 1, p, None, 0,
                           fake, None, None, None)
    assert rc == -1
    rc = lib.lssvr_enhance(fake, fake, 10, 5, 12, 0.0, 1.0, 0.0, 0.0, 9, 16, 1e4, 1, p, None, 0,
                           fake, None, None, None)
    assert rc == -2 and b"shard" in lib.lssvr_last_error()
    rc = lib.lssvr_enhance(fake, fake, 10, 0, 10, 0.0, 1.0, 0.0, 0.0, 9, 16, 1e4, 1, p, None, 9,
                           fake, None, None, None)
    assert rc == -5
    # empty shard: success without touching the device
    rc = lib.lssvr_enhance(None, None, 0, 0, 0, 0.0, 1.0, 0.0, 0.0, 9, 16, 1e4, 1, p, None, 0,
                           None, None, None, None)
    assert rc == 0
    assert lib.lssvr_p1_assemble(fake, 10, 9, 1, p, None, None, fake, fake, fake, None, None, None) == -7
    assert lib.lssvr_eval(fake, fake, 0, 9, fake, 1, fake, None, None) == -2
    assert lib.lssvr_tridiag_work_bytes(100000) > 8 * 100000
    with pytest.raises(_capi.LssvrHipError):
        _capi.check(-3, "demo")


def test_ops_reject_host_tensors():
    import torch
    from hybrid_fem_lssvr_amd import ops
    x = torch.linspace(0, 1, 5, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.enhance(x, x, 5, 1e4, 12, global_domain=(0.0, 1.0))
    with pytest.raises(TypeError):
        ops.enhance([0.0, 1.0], x, 5, 1e4, 12)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the package may import it."""
    pkg = os.path.join(ROOT, "hybrid_fem_lssvr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_enhance_work_bytes_is_host_arithmetic():
    """lssvr_enhance_work_bytes touches no device: the workspace of the kernel sequence above M = 22
    (96 doubles per element; 32 more where refinement steps follow: n_colloc - (M-2) <= 14), nothing
    for the lane kernel, the sequence on request for any M (LSSVR_SOLVER_PRIMAL_MOMENT)."""
    from hybrid_fem_lssvr_amd import _capi
    lib = _capi.load()
    wb = lib.lssvr_enhance_work_bytes
    assert wb(1000, 33, 64, _capi.SOLVER_PRIMAL) == 1000 * 96 * 8            # BASELINE config 4's shape
    assert wb(1000, 33, 45, _capi.SOLVER_PRIMAL) == 1000 * 128 * 8           # excess 14: one refinement step
    assert wb(1000, 33, 46, _capi.SOLVER_PRIMAL) == 1000 * 96 * 8
    assert wb(1000, 23, 21, _capi.SOLVER_PRIMAL) == 1000 * 128 * 8
    assert wb(1000, 22, 40, _capi.SOLVER_PRIMAL) == 0                        # lane kernel: no workspace
    assert wb(1000, 9, 16, _capi.SOLVER_PRIMAL) == 0
    assert wb(1000, 9, 16, _capi.SOLVER_PRIMAL_MOMENT) == 1000 * 96 * 8
    assert wb(1000, 33, 64, _capi.SOLVER_DUAL) == 0 and wb(1000, 33, 64, _capi.SOLVER_PRIMAL_WAVE) == 0
    assert wb(0, 33, 64, _capi.SOLVER_PRIMAL) == 0


def test_step_plan_binds_and_validates_without_gpu():
    """lssvr_step_plan_create is host arithmetic (validation + a small host allocation, no HIP call): it runs here.
    Same checks as lssvr_step; a failed create leaves a NULL handle; destroy accepts NULL."""
    from hybrid_fem_lssvr_amd import _capi
    lib = _capi.load()
    rhs = _capi.rhs_params(9.869604401089358, 3.141592653589793)
    fake = [0x10000 * (i + 1) for i in range(8)]            # never dereferenced: the plan is not launched

    def create(ne=100, M=9, n=16, nquad=2, diag=fake[2]):
        h = ctypes.c_void_p()
        rc = lib.lssvr_step_plan_create(ctypes.byref(h), fake[0], fake[1], ne, 0, ne, -1.0, 1.0, 0.0, 0.0, M, n, 1e4,
                                        rhs, nquad, diag, fake[3], fake[4], fake[5], fake[6], None)
        return rc, h

    rc, h = create()
    assert rc == 0 and h.value
    assert lib.lssvr_step_plan_destroy(h) == 0
    for kw in ({"nquad": 9}, {"ne": 0}, {"M": 40}, {"n": 3}, {"diag": None}):
        rc, h = create(**kw)
        assert rc < 0 and not h.value, kw
        assert lib.lssvr_last_error().decode()
    assert lib.lssvr_step_plan_destroy(None) == 0
    assert lib.lssvr_step_plan_launch(None, None) < 0         # NULL plan: an argument error, no HIP call
