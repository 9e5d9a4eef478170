"""Host-side mirror of the reference's step interface on top of the C ABI.

The reference (project/sequential/FluidSequential.c) exposes its hot path as
plain functions on six (N+2)^2 float arrays: set_bnd, add_source, diffuse,
advect, computeDivergenceAndPressure, lastProject, dens_step, vel_step
(:62-241).  FluidSolver keeps the same names, argument order and meaning, with
the arrays resident on the MI355X and addressed by name.
"""
import ctypes as C

import numpy as np

from . import capi

# the reference's harness constants (FluidSequential.c:7-9,91)
DT, VIS, DIFF, ITERS = 0.016, 0.0025, 0.1, 40

_ID = {name: k for k, name in enumerate(capi.FIELD_NAMES)}


def _fid(f):
    if isinstance(f, str):
        return _ID[f]
    return int(f)


def _host(a, n):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.shape != (n + 2, n + 2):
        raise ValueError("field must have shape (%d, %d), got %s" % (n + 2, n + 2, a.shape))
    return a


class FluidSolver:
    """Six resident fields + scratch on one GPU (or one row slab of several)."""

    def __init__(self, n, rank=0, nranks=1, halo=0, jacobi=capi.JACOBI_TB, stream=None,
                 arena_ptr=None, arena_bytes=0, params=None, storage=capi.STORAGE_F32):
        self._h = C.c_void_p()
        self.n = int(n)
        self.storage = storage
        cfg = capi.Config(n=self.n, rank=rank, nranks=nranks, halo=halo, jacobi_variant=jacobi,
                          stream=stream, arena=arena_ptr, arena_bytes=arena_bytes, storage=storage)
        capi.check(capi.lib().fluid_create_ex(C.byref(cfg), C.byref(self._h)))
        lo, hi = C.c_int(), C.c_int()
        capi.check(capi.lib().fluid_owned_rows(self._h, C.byref(lo), C.byref(hi)))
        self.owned_rows = (lo.value, hi.value)
        self.rank, self.nranks = rank, nranks
        self._cb = None
        for key, value in (params or {}).items():     # capi.PARAM_* tuning knobs
            self.set_param(key, value)

    # -- lifetime
    def close(self):
        if self._h:
            capi.lib().fluid_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- data movement
    def upload(self, **fields):
        for name, arr in fields.items():
            capi.check(capi.lib().fluid_upload(self._h, _fid(name), _host(arr, self.n)))

    def upload_rows(self, field, arr, row_lo, row_hi):
        capi.check(capi.lib().fluid_upload_rows(self._h, _fid(field), _host(arr, self.n), row_lo, row_hi))

    def download(self, field, out=None):
        if out is None:
            out = np.empty((self.n + 2, self.n + 2), dtype=np.float32)
        capi.check(capi.lib().fluid_download(self._h, _fid(field), _host(out, self.n)))
        return out

    def download_rows(self, field, out, row_lo, row_hi):
        capi.check(capi.lib().fluid_download_rows(self._h, _fid(field), out, row_lo, row_hi))
        return out

    def fill(self, field, value=0.0):
        capi.check(capi.lib().fluid_fill(self._h, _fid(field), value))

    def synchronize(self):
        capi.check(capi.lib().fluid_synchronize(self._h))

    def field_ptr(self, field):
        p = C.c_void_p()
        capi.check(capi.lib().fluid_field_ptr(self._h, _fid(field), C.byref(p)))
        return p.value

    def scalar_ptr(self):
        p = C.c_void_p()
        capi.check(capi.lib().fluid_scalar_ptr(self._h, C.byref(p)))
        return p.value

    # -- the reference's operators (same names / argument order)
    def set_bnd(self, b, x):
        capi.check(capi.lib().fluid_op_set_bnd(self._h, b, _fid(x)))

    def add_source(self, x, s, dt=DT):
        capi.check(capi.lib().fluid_op_add_source(self._h, _fid(x), _fid(s), dt))

    def jacobi_sweep(self, b, x, x0, out, alpha, beta):
        capi.check(capi.lib().fluid_op_jacobi_sweep(self._h, b, _fid(x), _fid(x0), _fid(out), alpha, beta))

    def diffuse(self, b, x, x0, alpha, beta, iters=ITERS):
        capi.check(capi.lib().fluid_op_diffuse(self._h, b, _fid(x), _fid(x0), alpha, beta, iters))

    def advect(self, b, d, d0, u, v, dt=DT):
        capi.check(capi.lib().fluid_op_advect(self._h, b, _fid(d), _fid(d0), _fid(u), _fid(v), dt))

    def computeDivergenceAndPressure(self, u, v, p, div):
        capi.check(capi.lib().fluid_op_divergence(self._h, _fid(u), _fid(v), _fid(p), _fid(div)))

    def lastProject(self, u, v, p, div=None):
        capi.check(capi.lib().fluid_op_subtract_gradient(self._h, _fid(u), _fid(v), _fid(p)))

    def vel_step(self, visc=VIS, dt=DT, iters=ITERS):
        """vel_step(u, v, u_prev, v_prev, visc) on the resident fields."""
        capi.check(capi.lib().fluid_vel_step(self._h, dt, visc, iters))

    def dens_step(self, diff=DIFF, dt=DT, iters=ITERS):
        """dens_step(dens, dens_prev, u, v, diff) on the resident fields."""
        capi.check(capi.lib().fluid_dens_step(self._h, dt, diff, iters))

    def step(self, nsteps=1, use_sources=False, dt=DT, diff=DIFF, visc=VIS, iters=ITERS):
        """nsteps bodies of the reference's main loop (FluidSequential.c:289-312)."""
        capi.check(capi.lib().fluid_step(self._h, dt, diff, visc, iters, nsteps, 1 if use_sources else 0))

    # -- diagnostics / tuning
    def residual(self, x, x0, alpha, beta):
        out = C.c_float()
        capi.check(capi.lib().fluid_residual(self._h, _fid(x), _fid(x0), alpha, beta, C.byref(out)))
        return out.value

    def absmax_velocity(self, u="u", v="v"):
        out = C.c_float()
        capi.check(capi.lib().fluid_absmax_velocity(self._h, _fid(u), _fid(v), C.byref(out)))
        return out.value

    def set_jacobi_variant(self, variant):
        capi.check(capi.lib().fluid_set_jacobi_variant(self._h, variant))

    def division_mode(self, alpha, beta):
        """0 true division, 2 double reciprocal, 3 two-term reciprocal (tile-proved), 4 exact reciprocal,
        5 float reciprocal + scaled residual correction."""
        m = C.c_int()
        capi.check(capi.lib().fluid_division_mode(self._h, alpha, beta, C.byref(m)))
        return m.value

    def autotune_pending(self):
        """Launch shapes whose strip height the run-time tuner is still measuring (process-wide)."""
        m = C.c_int()
        capi.check(capi.lib().fluid_autotune_pending(self._h, C.byref(m)))
        return m.value

    def set_param(self, key, value):
        capi.check(capi.lib().fluid_set_param(self._h, key, value))

    def timing_enable(self, on=True):
        capi.check(capi.lib().fluid_timing_enable(self._h, 1 if on else 0))

    def timing_read(self, reset=True):
        t = capi.Timing()
        capi.check(capi.lib().fluid_timing_read(self._h, C.byref(t), 1 if reset else 0))
        out = {"jacobi_ms": t.jacobi_ms, "sweeps": t.sweeps, "solves": t.solves,
               "jacobi_launches": t.jacobi_launches, "jacobi_field_launches": t.jacobi_field_launches,
               "pressure_ms": t.pressure_ms, "pressure_sweeps": t.pressure_sweeps}
        for k, name in enumerate(capi.TIMING_CATEGORIES):
            out[name + "_ms"] = t.category_ms[k]
            out[name + "_calls"] = t.category_calls[k]
        return out

    def diffuse_tol(self, b, x, x0, alpha, beta, tol, max_iters=10000, check_every=8):
        """Opt-in, not the reference's behaviour: sweep until the residual <= tol.
        Returns (sweeps done, final residual)."""
        it, res = C.c_int(), C.c_float()
        capi.check(capi.lib().fluid_op_diffuse_tol(self._h, b, _fid(x), _fid(x0), alpha, beta, tol, max_iters,
                                                   check_every, C.byref(it), C.byref(res)))
        return it.value, res.value

    def split_launches(self):
        """Jacobi launches that ran as interior + edge strips around an exchange in flight (row slabs)."""
        m = C.c_longlong()
        capi.check(capi.lib().fluid_split_launches(self._h, C.byref(m)))
        return m.value

    def exchange_stream(self):
        """hipStream_t (as an integer) an exchange callback should enqueue on right now."""
        p = C.c_void_p()
        capi.check(capi.lib().fluid_exchange_stream(self._h, C.byref(p)))
        return p.value or 0

    def set_exchange(self, fn):
        """fn(kind, fields, depth, scalar_or_None) -> new scalar or None; raises on failure."""
        if fn is None:
            self._cb = None
            capi.check(capi.lib().fluid_set_exchange(self._h, C.cast(None, capi.EXCHANGE_FN), None))
            return

        def tramp(_user, kind, fields, nfields, depth, scalar):
            try:
                ids = [fields[k] for k in range(nfields)]
                if kind in (capi.XCHG_MAX, capi.XCHG_MAX_END):
                    scalar[0] = float(fn(kind, ids, depth, float(scalar[0])))
                else:
                    fn(kind, ids, depth, None)
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        self._cb = capi.EXCHANGE_FN(tramp)
        capi.check(capi.lib().fluid_set_exchange(self._h, self._cb, None))


def coefficients(n, dt, coef):
    a, b = C.c_float(), C.c_float()
    capi.check(capi.lib().fluid_coefficients(n, dt, coef, C.byref(a), C.byref(b)))
    return a.value, b.value


def step(N, dt, diff, visc, u, v, dens):
    """The drop-in: one loop body of the reference's main for z > 0, in place
    on host arrays (C ABI `step`, include/fluid_amd.h)."""
    for a in (u, v, dens):
        if a.dtype != np.float32 or not a.flags.c_contiguous or a.shape != (N + 2, N + 2):
            raise ValueError("fields must be C-contiguous float32 of shape (N+2, N+2)")
    capi.check(capi.lib().step(N, dt, diff, visc, u, v, dens))


def step_src(N, dt, diff, visc, iters, u, v, dens, u_prev, v_prev, dens_prev):
    for a in (u, v, dens, u_prev, v_prev, dens_prev):
        if a.dtype != np.float32 or not a.flags.c_contiguous or a.shape != (N + 2, N + 2):
            raise ValueError("fields must be C-contiguous float32 of shape (N+2, N+2)")
    capi.check(capi.lib().step_src(N, dt, diff, visc, iters, u, v, dens, u_prev, v_prev, dens_prev))
