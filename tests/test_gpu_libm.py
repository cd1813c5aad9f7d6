"""The device side of csrc/gpis_libm.hpp: exp / log / logf / sin / cos / sincos / pow as the GPU computes them equal, bit for bit,
what the libm of this host returns (glibc, through ctypes — numpy's own vectorised exp / log are a different implementation and
not the comparator).  tests/test_libm_replica_cpu.py checks the same header compiled for the host on 10^7 arguments; here 2*10^5
per function go through the device."""
import ctypes
import ctypes.util

import numpy as np
import pytest

import _gpis_pkg

pytestmark = pytest.mark.gpu
pkg = _gpis_pkg.load_package()

N = 200_000


def _libm():
    m = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    for name in ("exp", "log", "sin", "cos"):
        getattr(m, name).restype = ctypes.c_double
        getattr(m, name).argtypes = [ctypes.c_double]
    m.pow.restype = ctypes.c_double
    m.pow.argtypes = [ctypes.c_double, ctypes.c_double]
    m.logf.restype = ctypes.c_float
    m.logf.argtypes = [ctypes.c_float]
    m.sincosf.restype = None
    m.sincosf.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    m.sincos.restype = None
    m.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    return m


def _same(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))


def _args(rng, kind):
    u = rng.random(N)
    if kind == "exp":
        return np.concatenate([-u[:N // 4] * 40, -u[N // 4:N // 2] * 800, (u[N // 2:3 * N // 4] - 0.5) * 1400, -u[3 * N // 4:] * 1e-3])
    if kind == "log":
        return np.concatenate([1 - u[:N // 4].astype(np.float32).astype(np.float64) * 0.999, u[N // 4:N // 2] * 4 + 1e-9, 0.9 + u[N // 2:3 * N // 4] * 0.2,
                               np.exp(u[3 * N // 4:] * 1400 - 700)])
    if kind == "logf":
        return np.concatenate([u[:N // 2] * 4 + 1e-3, np.exp(u[N // 2:] * 40 - 20)]).astype(np.float32).astype(np.float64)
    if kind == "trig":
        return np.concatenate([u[:N // 4] * 2 * np.pi, (u[N // 4:N // 2].astype(np.float32).astype(np.float64)) * float(np.float32(2 * 3.1415926536)),
                               (u[N // 2:3 * N // 4] - 0.5) * 2e4, (u[3 * N // 4:] - 0.5) * 2e8])
    raise KeyError(kind)


@pytest.mark.parametrize("fn,kind", [("exp", "exp"), ("log", "log"), ("logf", "logf"), ("sin", "trig"), ("cos", "trig")])
def test_unary_functions_equal_the_host_libm(fn, kind):
    m = _libm()
    x = _args(np.random.default_rng(11), kind)
    got = pkg.libm_eval(fn, x)
    if fn == "logf":
        want = np.array([m.logf(float(v)) for v in x], dtype=np.float64)
    else:
        f = getattr(m, fn)
        want = np.array([f(float(v)) for v in x], dtype=np.float64)
    bad = ~_same(got, want)
    assert not bad.any(), (fn, int(bad.sum()), x[bad][:4], got[bad][:4], want[bad][:4])


def test_sincos_equals_the_host_sincos_and_differs_from_sin_cos_somewhere():
    m = _libm()
    x = _args(np.random.default_rng(12), "trig")
    s, c = pkg.libm_eval("sincos", x)
    ws, wc = np.empty(N), np.empty(N)
    ps, pc = ctypes.c_double(), ctypes.c_double()
    for i, v in enumerate(x):
        m.sincos(float(v), ctypes.byref(ps), ctypes.byref(pc))
        ws[i], wc[i] = ps.value, pc.value
    assert _same(s, ws).all() and _same(c, wc).all()
    # the reason both exist: this libm's sincos (no FMA variant) is not bit-identical to its sin / cos
    assert (~_same(s, pkg.libm_eval("sin", x))).sum() > 0


def test_pow_equals_the_host_pow_on_the_cubes_and_squares_the_path_takes():
    m = _libm()
    rng = np.random.default_rng(13)
    u = rng.random(N)
    x = np.concatenate([u[:N // 4] * 4 + 1e-6, np.exp(u[N // 4:N // 2] * 60 - 30), (u[N // 2:3 * N // 4] * 2.5 + 0.01).astype(np.float32).astype(np.float64),
                        np.exp(u[3 * N // 4:] * 200 - 100)])
    y = np.concatenate([np.full(N // 2, 3.0), np.full(N // 4, 3.0), rng.integers(-6, 7, N // 4) + 0.5])
    got = pkg.libm_eval("pow", x, y)
    want = np.array([m.pow(float(a), float(b)) for a, b in zip(x, y)], dtype=np.float64)
    bad = ~_same(got, want)
    assert not bad.any(), (int(bad.sum()), x[bad][:4], y[bad][:4], got[bad][:4], want[bad][:4])
    assert pkg.libm_eval("pow", np.array([0.0]), np.array([3.0]))[0] == 0.0


def test_sincosf_equals_the_host_sincosf():
    m = _libm()
    u = np.random.default_rng(14).random(N)
    x = np.concatenate([u[:N // 2] * (np.pi / 2), (u[N // 2:3 * N // 4] - 0.5) * 12, (u[3 * N // 4:] - 0.5) * 238]).astype(np.float32)
    s, c = pkg.libm_eval("sincosf", x.astype(np.float64))
    ws, wc = np.empty(N, np.float32), np.empty(N, np.float32)
    ps, pc = ctypes.c_float(), ctypes.c_float()
    for i, v in enumerate(x):
        m.sincosf(float(v), ctypes.byref(ps), ctypes.byref(pc))
        ws[i], wc[i] = ps.value, pc.value
    assert np.array_equal(s.astype(np.float32).view(np.uint32), ws.view(np.uint32)) and np.array_equal(c.astype(np.float32).view(np.uint32), wc.view(np.uint32))
