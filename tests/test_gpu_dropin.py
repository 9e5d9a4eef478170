"""GPU: examples/drop_in.c -- a plain C99 program shaped like the reference's main() (FluidSequential.c:273-334)
with the loop body's calls replaced by step_src() / step() as INTEGRATION.md section 1 describes -- compiled with
gcc against include/fluid_amd.h, linked to libfluid_amd.so, run as its own process, and checked against the
reference's golden snapshot (tests/golden/step_n126_k40.npz).  No ctypes, no Python in the path under test."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_bit_equal, load_golden

pytestmark = pytest.mark.gpu


def build(tmp_path):
    exe = str(tmp_path / "drop_in")
    lib = os.path.join(ROOT, "fluidsimulationcuda_amd")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "drop_in.c"), "-o", exe, "-L" + lib, "-lfluid_amd",
           "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_drop_in_reproduces_the_references_steps(tmp_path):
    n = 126
    g = load_golden("step_n%d_k40.npz" % n)
    exe = build(tmp_path)
    pre = str(tmp_path / "in")
    for name in ("u_prev", "v_prev", "dens_prev"):
        g["init_" + name].astype("<f4").tofile("%s_%s.f32" % (pre, name))
    for steps in (1, 2, 5):
        out = str(tmp_path / ("out%d" % steps))
        p = subprocess.run([exe, str(n), str(steps), pre, out], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        assert p.stdout.startswith("Tot: ")
        for name in ("u", "v", "dens"):
            got = np.fromfile("%s_%s.f32" % (out, name), dtype="<f4").reshape(n + 2, n + 2)
            assert_bit_equal(got, g["s%d_%s" % (steps, name)], "%s after %d steps of the C drop-in" % (name, steps))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_drop_in_reports_errors_instead_of_exiting(tmp_path):
    """The reference's CUDA variants exit(EXIT_FAILURE) inside their CHECK macro; this ABI returns codes."""
    exe = build(tmp_path)
    p = subprocess.run([exe, "0", "1"], capture_output=True, text=True, timeout=120)       # N = 0 is rejected
    assert p.returncode == 1 and "libfluid_amd error 1" in p.stderr
