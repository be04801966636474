"""Accuracy sweep of the engine's range-limited sin/cos/atan2/floormod (csrc/fastmath.hpp), host-compiled."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
F = np.float32


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("shim") / "libfastmath_shim.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-shared", "-fPIC",
                    os.path.join(HERE, "fastmath_shim.cpp"), "-o", out], check=True)
    return ctypes.CDLL(out)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _ulps(got, ref64):
    ref32 = ref64.astype(F)
    ulp = np.spacing(np.maximum(np.abs(ref32), F(1e-30))).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / ulp


def test_sincos_accuracy(shim):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-8, 8, 2_000_000), np.linspace(-8, 8, 200_001),
                        np.arange(-5, 6) * (np.pi / 2), rng.uniform(-1e-4, 1e-4, 1000), [0.0, -0.0, 8.0, -8.0]]).astype(F)
    s, c = np.empty_like(x), np.empty_like(x)
    shim.shim_sincos(_p(x), x.size, _p(s), _p(c))
    es, ec = _ulps(s, np.sin(x.astype(np.float64))), _ulps(c, np.cos(x.astype(np.float64)))
    # near multiples of pi/2 the true value is ~1e-8 and relative error is meaningless below fp32 resolution
    big_s, big_c = np.abs(s) > 1e-6, np.abs(c) > 1e-6
    assert es[big_s].max() < 1.7 and ec[big_c].max() < 1.7
    assert es[big_s].mean() < 0.4 and ec[big_c].mean() < 0.4
    assert np.abs(s[~big_s].astype(np.float64) - np.sin(x[~big_s].astype(np.float64))).max(initial=0) < 1e-7


def test_sin_on_principal_angles(shim):
    rng = np.random.default_rng(5)
    pi32 = float(F(np.pi))
    x = np.concatenate([rng.uniform(-np.pi, np.pi, 2_000_000), np.linspace(-pi32, pi32, 200_001),
                        [pi32, -pi32, np.pi / 2, -np.pi / 2, 0.0, 1e-20, -1e-20]]).astype(F)
    o = np.empty_like(x)
    shim.shim_sin_pi(_p(x), x.size, _p(o))
    ref = np.sin(x.astype(np.float64))
    e = _ulps(o, ref)
    big = np.abs(ref) > 1e-6
    assert e[big].max() < 2.0 and e[big].mean() < 0.4
    assert np.abs(o[~big].astype(np.float64) - ref[~big]).max() < 2e-8
    assert np.array_equal(np.sign(o[big]), np.sign(ref[big]))
    nan = np.array([np.nan], F)
    shim.shim_sin_pi(_p(nan), 1, _p(nan))
    assert np.isnan(nan[0])


def test_sin_folded_on_0_2pi(shim):
    rng = np.random.default_rng(6)
    x = np.concatenate([rng.uniform(0, 2 * np.pi, 2_000_000), np.linspace(0, 2 * np.pi, 200_001),
                        [0.0, np.pi / 2, np.pi, 1.5 * np.pi, 2 * np.pi, float(F(2 * np.pi)), -2e-7, 6.2831860]]).astype(F)
    o = np.empty_like(x)
    shim.shim_sin_fold(_p(x), x.size, _p(o))
    ref = np.sin(x.astype(np.float64))
    e = _ulps(o, ref)
    big = np.abs(ref) > 1e-6
    assert e[big].max() < 2.0 and e[big].mean() < 0.4
    assert np.abs(o[~big].astype(np.float64) - ref[~big]).max() < 2e-8


def test_sincos_falls_back_outside_fast_domain(shim):
    x = np.array([8.5, -100.0, 1e6, np.inf, np.nan, 3.0e38], F)
    s, c = np.empty_like(x), np.empty_like(x)
    shim.shim_sincos(_p(x), x.size, _p(s), _p(c))
    with np.errstate(invalid="ignore"):
        np.testing.assert_allclose(s, np.sin(x.astype(np.float64)).astype(F), rtol=2e-7, atol=1e-7)
        np.testing.assert_allclose(c, np.cos(x.astype(np.float64)).astype(F), rtol=2e-7, atol=1e-7)


def test_atan2_accuracy_and_quadrants(shim):
    rng = np.random.default_rng(1)
    th = rng.uniform(-np.pi, np.pi, 2_000_000)
    rad = np.exp(rng.uniform(-3, 3, th.size))
    y = (rad * np.sin(th)).astype(F)
    x = (rad * np.cos(th)).astype(F)
    o = np.empty_like(x)
    shim.shim_atan2(_p(y), _p(x), x.size, _p(o))
    e = _ulps(o, np.arctan2(y.astype(np.float64), x.astype(np.float64)))
    ok = np.abs(o) > 1e-6
    assert e[ok].max() < 2.1 and e[ok].mean() < 0.5
    # axes, signed zeros, specials follow libm
    yy = np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 0.0, 1.0, np.inf, np.nan, 1e-38, 3.0], F)
    xx = np.array([1.0, 1.0, -1.0, -1.0, 0.0, 0.0, 0.0, np.inf, 1.0, 1.0, 1e-38, -4.0], F)
    o = np.empty_like(xx)
    shim.shim_atan2(_p(yy), _p(xx), xx.size, _p(o))
    ref = np.arctan2(yy.astype(np.float64), xx.astype(np.float64)).astype(F)
    np.testing.assert_allclose(o, ref, rtol=3e-7, atol=0, equal_nan=True)
    assert np.array_equal(np.signbit(o[:4]), np.signbit(ref[:4]))


def test_floormod_is_exact(shim):
    rng = np.random.default_rng(2)
    y = F(2 * np.pi)
    x = np.concatenate([rng.uniform(0, 2 * np.pi, 500_000), rng.uniform(-50, 50, 100_000),
                        [0.0, float(y), float(F(np.pi)), 2 * float(y), -1e-30, np.nan]]).astype(F)
    o = np.empty_like(x)
    shim.shim_floormod(_p(x), ctypes.c_float(float(y)), x.size, _p(o))
    from oracle.oracle_np import floormod32
    np.testing.assert_array_equal(o, floormod32(x, y))
