"""CPU: libfluid_amd.so loads and exports every symbol include/fluid_amd.h
declares (no compute without a GPU), and the ctypes table matches the header."""
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "fluid_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|size_t|char)\s*\*?\s*(\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_symbols_exported():
    import ctypes
    import __graft_entry__ as g
    g.build()
    from fluidsimulationcuda_amd._build import LIB
    L = ctypes.CDLL(LIB)
    names = header_functions()
    assert "step" in names and "step_src" in names and len(names) >= 30
    for name in names:
        assert hasattr(L, name), "symbol %s declared in fluid_amd.h but not exported" % name


def test_binding_matches_header():
    from fluidsimulationcuda_amd import capi
    bound = set(capi.SIGNATURES) | set(capi.OTHER_SYMBOLS)
    assert bound == set(header_functions())


def test_argument_validation_without_gpu():
    """Checks that do not need a device: bad N / null pointers give FLUID_E_INVALID
    and a message, never a crash."""
    import ctypes as C
    from fluidsimulationcuda_amd import capi
    L = capi.lib()
    assert L.fluid_arena_bytes(0) == 0
    assert L.fluid_layout(0, None, None, None) == capi.E_INVALID
    assert b"N must be" in L.fluid_last_error()
    h = C.c_void_p()
    assert L.fluid_create(-3, C.byref(h)) == capi.E_INVALID and not h.value
    assert L.fluid_synchronize(None) == capi.E_INVALID
    pitch, xoff, ff = C.c_int(), C.c_int(), C.c_size_t()
    assert L.fluid_layout(4094, C.byref(pitch), C.byref(xoff), C.byref(ff)) == capi.OK
    assert pitch.value % 64 == 0 and pitch.value >= 4096 + xoff.value and (xoff.value + 1) % 64 == 0
    assert ff.value == 4096 * pitch.value
    assert L.fluid_arena_bytes(4094) == ff.value * 4 * capi.NFIELDS + 256      # fields + control block
    a, b = C.c_float(), C.c_float()
    assert L.fluid_coefficients(126, 0.016, 0.1, C.byref(a), C.byref(b)) == capi.OK


def test_coefficients_match_oracle(oracle):
    from fluidsimulationcuda_amd import coefficients
    for n in (1, 14, 126, 1022, 4094, 8190, 16382):
        for coef in (0.0025, 0.1, 1e-7, 3.0):
            assert coefficients(n, 0.016, coef) == oracle.coefficients(n, 0.016, coef)


def test_no_gpu_means_loud_failure():
    """The product has no CPU path: without a device, create fails with a HIP
    error (and with a device this test is moot)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ctypes as C
    from fluidsimulationcuda_amd import capi
    h = C.c_void_p()
    rc = capi.lib().fluid_create(30, C.byref(h))
    assert rc in (capi.E_HIP, capi.E_NOMEM) and not h.value


def test_product_never_touches_oracle():
    """The oracle is test infrastructure: nothing under the package may name it."""
    pkg = os.path.join(ROOT, "fluidsimulationcuda_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f


def test_planned_launch_depths_without_gpu():
    """fluid_plan_sweeps: the host logic that decides how many sweeps each launch of a solve fuses (pick_sweeps in
    fluid_solver.hip) needs no device.  Against conftest.tb_schedule (the mirror the GPU tests assert launch counts
    with) and against the size rules: deep launches from 8 M cells, 16 for the general form only once a field
    outgrows 96 MiB and never on slabs under 3000 rows, fp16 storage always the greedy 8 / 4 / 2."""
    import ctypes as C
    from conftest import tb_schedule
    from fluidsimulationcuda_amd import capi
    L = capi.lib()

    def plan(n, rows, storage, pressure, iters, max_t=16, t16=-1):
        buf, cnt = (C.c_int * 64)(), C.c_int()
        assert L.fluid_plan_sweeps(n, rows, storage, pressure, iters, max_t, t16, buf, 64, C.byref(cnt)) == capi.OK
        return list(buf[:cnt.value])

    for iters in (0, 2, 6, 8, 12, 20, 22, 28, 32, 40, 48, 100):
        for max_t in (16, 12, 8, 4, 2):
            assert plan(300, 300, 0, 1, iters, max_t, 0) == tb_schedule(iters, max_t), (iters, max_t)
            assert sum(plan(300, 300, 0, 0, iters, max_t, 0)) == iters
    assert plan(4094, 4094, 0, 1, 40) == [16, 12, 12]            # BASELINE config 2: pressure form at 4096^2
    assert plan(4094, 4094, 0, 0, 40) == [12, 12, 8, 8]          # general form: fields still inside the Infinity Cache
    assert plan(8190, 8190, 0, 0, 40) == [16, 12, 12]            # 8192^2: both forms
    assert plan(8190, 2047, 0, 1, 40) == [12, 12, 8, 8]          # one slab of four: 16 M cells, but too few rows for 16
    assert plan(8190, 1023, 0, 1, 40) == [8, 8, 8, 8, 8]         # one slab of eight: just under 8 M cells
    assert plan(1022, 1022, 0, 1, 40) == [8, 8, 8, 8, 8]         # small grids: 8 at a time
    assert plan(16382, 16382, 1, 1, 40) == [8, 8, 8, 8, 8]       # fp16 storage: the schedule is part of the result
    assert plan(16382, 16382, 1, 0, 20) == [8, 8, 4]
    buf, cnt = (C.c_int * 2)(), C.c_int()
    assert L.fluid_plan_sweeps(300, 300, 0, 1, 40, 8, -1, buf, 2, C.byref(cnt)) == capi.OK and cnt.value == 5   # reports the count past capacity
    assert L.fluid_plan_sweeps(300, 300, 0, 1, 7, 8, -1, buf, 2, C.byref(cnt)) == capi.E_INVALID               # odd sweep count
    assert L.fluid_plan_sweeps(300, 301, 0, 1, 8, 8, -1, buf, 2, C.byref(cnt)) == capi.E_INVALID
