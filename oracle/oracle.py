"""ctypes view of oracle/liboracle.so (the CPU restatement) and of the
reference builds under oracle/_ref/.

TEST INFRASTRUCTURE, NOT PRODUCT: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_F = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")

# constants of the reference's harness (FluidSequential.c:7-9,91)
DT, VISC, DIFF, ITERS = 0.016, 0.0025, 0.1, 40


def build(force=False):
    """Compile liboracle.so; returns its path."""
    so = os.path.join(HERE, "liboracle.so")
    src = os.path.join(HERE, "fluid_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def build_ref(pairs=()):
    """Compile the reference itself into oracle/_ref (no-op off this container)."""
    subprocess.check_call([os.path.join(HERE, "build_ref.sh"), *pairs], stdout=subprocess.DEVNULL)


def field(n, fill=0.0):
    return np.full((n + 2, n + 2), fill, dtype=np.float32)


class Oracle:
    """The restatement.  All fields are (n+2, n+2) float32 C-contiguous arrays
    indexed [row i, col j]; operations are in place, as in the reference."""

    def __init__(self):
        L = self.lib = C.CDLL(build())
        i, f, u64 = C.c_int, C.c_float, C.c_ulonglong
        sig = {
            "fo_set_bnd": (None, [i, i, _F]),
            "fo_add_source": (None, [i, f, _F, _F]),
            "fo_jacobi_sweep": (None, [i, i, _F, _F, _F, f, f]),
            "fo_diffuse": (i, [i, i, _F, _F, f, f, i]),
            "fo_jacobi_rows": (None, [i, _F, _F, _F, f, f, i, i]),
            "fo_diffuse_mt": (i, [i, i, _F, _F, f, f, i, i]),
            "fo_advect": (None, [i, i, f, _F, _F, _F, _F]),
            "fo_divergence": (None, [i, _F, _F, _F, _F]),
            "fo_subtract_gradient": (None, [i, _F, _F, _F]),
            "fo_coefficients": (None, [i, f, f, C.POINTER(f), C.POINTER(f)]),
            "fo_vel_step": (i, [i, f, f, i, _F, _F, _F, _F]),
            "fo_dens_step": (i, [i, f, f, i, _F, _F, _F, _F]),
            "fo_step_src": (i, [i, f, f, f, i] + [_F] * 6),
            "fo_step": (i, [i, f, f, f, i] + [_F] * 6),
            "fo_initialize_glibc": (None, [i, C.c_uint] + [_F] * 6),
            "fo_initialize_portable": (None, [i, u64] + [_F] * 6),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args

    @staticmethod
    def _n(x):
        assert x.ndim == 2 and x.shape[0] == x.shape[1] and x.dtype == np.float32
        return x.shape[0] - 2

    def set_bnd(self, b, x):
        self.lib.fo_set_bnd(self._n(x), b, x)

    def add_source(self, x, s, dt=DT):
        self.lib.fo_add_source(self._n(x), dt, x, s)

    def jacobi_sweep(self, b, x, x0, out, alpha, beta):
        self.lib.fo_jacobi_sweep(self._n(x), b, x, x0, out, alpha, beta)

    def diffuse_threaded(self, b, x, x0, alpha, beta, iters, threads):
        """The same solve with every sweep split into row bands over `threads`
        POSIX threads (fo_diffuse_mt); bit-identical to diffuse()."""
        assert self.lib.fo_diffuse_mt(self._n(x), b, x, x0, alpha, beta, iters, threads) == 0

    def diffuse(self, b, x, x0, alpha, beta, iters=ITERS):
        assert self.lib.fo_diffuse(self._n(x), b, x, x0, alpha, beta, iters) == 0

    def advect(self, b, d, d0, u, v, dt=DT):
        assert d is not d0
        self.lib.fo_advect(self._n(d), b, dt, d, d0, u, v)

    def divergence(self, u, v, p, div):
        self.lib.fo_divergence(self._n(u), u, v, p, div)

    def subtract_gradient(self, u, v, p):
        self.lib.fo_subtract_gradient(self._n(u), u, v, p)

    def coefficients(self, n, dt, coef):
        a, b = C.c_float(), C.c_float()
        self.lib.fo_coefficients(n, dt, coef, C.byref(a), C.byref(b))
        return a.value, b.value

    def vel_step(self, u, v, u0, v0, visc=VISC, dt=DT, iters=ITERS):
        assert self.lib.fo_vel_step(self._n(u), dt, visc, iters, u, v, u0, v0) == 0

    def dens_step(self, x, x0, u, v, diff=DIFF, dt=DT, iters=ITERS):
        assert self.lib.fo_dens_step(self._n(x), dt, diff, iters, x, x0, u, v) == 0

    def step_src(self, u, v, dens, u0, v0, dens0, dt=DT, diff=DIFF, visc=VISC, iters=ITERS):
        assert self.lib.fo_step_src(self._n(u), dt, diff, visc, iters, u, v, dens, u0, v0, dens0) == 0

    def step(self, u, v, dens, u0, v0, dens0, dt=DT, diff=DIFF, visc=VISC, iters=ITERS):
        assert self.lib.fo_step(self._n(u), dt, diff, visc, iters, u, v, dens, u0, v0, dens0) == 0

    def initialize_glibc(self, n, seed=1):
        """(dens, dens_prev, u, u_prev, v, v_prev) as the reference's
        initializeParameters draws them (glibc rand, default seed 1)."""
        fs = [field(n) for _ in range(6)]
        self.lib.fo_initialize_glibc(n, seed, *fs)
        return fs

    def initialize_portable(self, n, seed=1):
        fs = [field(n) for _ in range(6)]
        self.lib.fo_initialize_portable(n, seed, *fs)
        return fs


def ref_path(n, iters=ITERS):
    return os.path.join(HERE, "_ref", "libfluidref_n%d_k%d.so" % (n, iters))


def have_ref(n, iters=ITERS):
    return os.path.exists(ref_path(n, iters))


def can_build_ref():
    return os.path.exists("/root/reference/project/sequential/FluidSequential.c")


class Reference:
    """project/sequential/FluidSequential.c itself, compiled by build_ref.sh
    for one (N, sweeps).  dt/visc/diff are the reference's macros; they are
    not arguments here because the reference has none."""

    def __init__(self, n, iters=ITERS):
        if not have_ref(n, iters):
            build_ref(["%d:%d" % (n, iters)])
        if not have_ref(n, iters):
            raise FileNotFoundError(ref_path(n, iters))
        self.n, self.iters = n, iters
        # every build exports the same names: keep each handle local
        L = self.lib = C.CDLL(ref_path(n, iters), mode=os.RTLD_LOCAL)
        i, f = C.c_int, C.c_float
        sig = {
            "set_bnd": [i, _F],                                # FluidSequential.c:62
            "add_source": [_F, _F],                            # :78
            "diffuse": [i, _F, _F, f, f],                      # :85
            "advect": [i, _F, _F, _F, _F],                     # :107
            "computeDivergenceAndPressure": [_F, _F, _F, _F],  # :143
            "lastProject": [_F, _F, _F, _F],                   # :161
            "dens_step": [_F, _F, _F, _F, f],                  # :176
            "vel_step": [_F, _F, _F, _F, f, i],                # :189
            "initializeParameters": [_F] * 6,                  # :244
        }
        for name, args in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = None, args
        self._srand = C.CDLL(None).srand

    def _chk(self, *xs):
        for x in xs:
            assert x.shape == (self.n + 2, self.n + 2) and x.dtype == np.float32

    def set_bnd(self, b, x):
        self._chk(x)
        self.lib.set_bnd(b, x)

    def add_source(self, x, s):
        self._chk(x, s)
        self.lib.add_source(x, s)

    def diffuse(self, b, x, x0, alpha, beta):
        """The reference swaps pointers internally; with an even sweep count
        the result is in x (FluidSequential.c:100-103)."""
        assert self.iters % 2 == 0
        self._chk(x, x0)
        self.lib.diffuse(b, x, x0, alpha, beta)

    def advect(self, b, d, d0, u, v):
        self._chk(d, d0, u, v)
        self.lib.advect(b, d, d0, u, v)

    def divergence(self, u, v, p, div):
        self._chk(u, v, p, div)
        self.lib.computeDivergenceAndPressure(u, v, p, div)

    def subtract_gradient(self, u, v, p):
        self._chk(u, v, p)
        self.lib.lastProject(u, v, p, p)

    def vel_step(self, u, v, u0, v0, visc=VISC):
        self._chk(u, v, u0, v0)
        self.lib.vel_step(u, v, u0, v0, visc, 0)

    def dens_step(self, x, x0, u, v, diff=DIFF):
        self._chk(x, x0, u, v)
        self.lib.dens_step(x, x0, u, v, diff)

    def step_src(self, u, v, dens, u0, v0, dens0):
        self.vel_step(u, v, u0, v0)
        self.dens_step(dens, dens0, u, v)

    def step(self, u, v, dens, u0, v0, dens0):
        for a in (u0, v0, dens0):
            a[...] = 0.0
        self.step_src(u, v, dens, u0, v0, dens0)

    def initialize(self, seed=1):
        fs = [field(self.n) for _ in range(6)]
        self._srand(seed)
        self.lib.initializeParameters(*fs)
        return fs
