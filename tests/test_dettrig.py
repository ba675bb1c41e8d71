"""include/mas_dettrig.h: accuracy against glibc and CPU-side reproducibility facts."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dt(tmp_path_factory):
    d = tmp_path_factory.mktemp("dettrig")
    src = d / "dt.c"
    src.write_text('#include "include/mas_dettrig.h"\n'
                   "double f_sin(double x){return mas_dt_sin(x);} double f_cos(double x){return mas_dt_cos(x);}\n"
                   "double f_tan(double x){return mas_dt_tan(x);}\n"
                   "void f_sincos(double x,double*s,double*c){mas_dt_sincos(x,s,c);}\n"
                   "void f_sin_v(const double*x,double*y,int n){for(int i=0;i<n;++i)y[i]=mas_dt_sin(x[i]);}\n"
                   "void f_cos_v(const double*x,double*y,int n){for(int i=0;i<n;++i)y[i]=mas_dt_cos(x[i]);}\n"
                   "void f_tan_v(const double*x,double*y,int n){for(int i=0;i<n;++i)y[i]=mas_dt_tan(x[i]);}\n"
                   "void f_tan1_v(const double*x,double*y,int n){for(int i=0;i<n;++i)y[i]=mas_dt_tan_onediv(x[i]);}\n")
    so = d / "libdt.so"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", f"-I{ROOT}", str(src), "-o", str(so), "-lm"])
    lib = C.CDLL(str(so))
    for n in ("f_sin", "f_cos", "f_tan"):
        getattr(lib, n).restype = C.c_double
        getattr(lib, n).argtypes = [C.c_double]
    return lib


def _vec(lib, name, x):
    y = np.empty_like(x)
    getattr(lib, name)(x.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(x.size))
    return y


@pytest.mark.parametrize("name,ref,limit,ranges", [("sin", np.sin, 1.0, (1, 10, 400, 1e5)), ("cos", np.cos, 1.0, (1, 10, 400, 1e5)),
                                                   ("tan", np.tan, 3.0, (0.7, 10, 400))])
def test_accuracy_in_ulps_against_glibc(dt, name, ref, limit, ranges):
    rng = np.random.default_rng(0)
    for R in ranges:
        x = rng.uniform(-R, R, 300000)
        got, want = _vec(dt, f"f_{name}_v", x), ref(x)
        ulps = np.abs(got - want) / np.spacing(np.abs(want))
        assert ulps.max() <= limit, (name, R, ulps.max())


def test_special_values_and_symmetry(dt):
    assert dt.f_sin(0.0) == 0.0 and dt.f_cos(0.0) == 1.0 and dt.f_tan(0.0) == 0.0
    for x in (0.3, 1.57, 2.0, 100.25, 54321.125):
        assert dt.f_sin(-x) == -dt.f_sin(x)
        assert dt.f_cos(-x) == dt.f_cos(x)
        assert dt.f_tan(-x) == -dt.f_tan(x)
    assert math.isnan(dt.f_sin(float("nan"))) and math.isnan(dt.f_cos(float("inf")))
    assert math.isnan(dt.f_sin(1e9)) and math.isnan(dt.f_tan(-1e7))   # outside the reduction range: NaN, never libm


def test_sincos_equals_separate_calls_bitwise(dt):
    s, c = C.c_double(), C.c_double()
    rng = np.random.default_rng(1)
    for x in rng.uniform(-500, 500, 20000):
        dt.f_sincos(C.c_double(x), C.byref(s), C.byref(c))
        assert s.value == dt.f_sin(x) and c.value == dt.f_cos(x)


def test_tan_with_one_division_is_tan_bitwise(dt):
    """mas_dt_tan_onediv ((-c) / s in odd quadrants, one division of selected operands) against mas_dt_tan (-(c / s), a branch
    between two divisions): the same bits in every quadrant, at both zeros, for subnormals and out-of-range arguments."""
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-10, 10, 2_000_000), rng.uniform(-1e5, 1e5, 200_000), rng.uniform(-1e-300, 1e-300, 1000),
                         np.array([0.0, -0.0, 5e-324, -5e-324, np.pi / 4, -np.pi / 4, np.pi / 2, -np.pi / 2, 1.5707963267948966, 3.0, -3.0, 0.5, -0.5,
                                   np.nan, np.inf, -np.inf, 1e9])])
    a, b = _vec(dt, "f_tan_v", xs), _vec(dt, "f_tan1_v", xs)
    fin = ~np.isnan(a)
    assert np.array_equal(a.view(np.uint64)[fin], b.view(np.uint64)[fin])     # bit patterns, signs of zero included
    assert np.array_equal(np.isnan(a), np.isnan(b))
