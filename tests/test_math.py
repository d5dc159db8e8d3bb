"""CPU: include/nova_math.h (the backend's bit-reproducible elementary functions) against float64,
on the ranges the path uses.  Bounds are ulps of the float32 result; all are inside OpenCL 1.2's
full-profile limits (sin/cos/asin/acos 4, atan 5, atan2 6, exp 3, pow 16)."""
import numpy as np
import pytest

from oracle import bind as O

N = 400000


def ulp_err(y, ref):
    ref32 = ref.astype(np.float32)
    sp = np.spacing(np.abs(ref32)).astype(np.float64)
    return float(np.nanmax(np.abs(y.astype(np.float64) - ref) / sp))


@pytest.mark.parametrize("name,lo,hi,fn,bound", [
    ("sin", 0.0, 2 * np.pi, np.sin, 2.0),
    ("cos", 0.0, 2 * np.pi, np.cos, 2.0),
    ("asin", -1.0, 1.0, np.arcsin, 3.0),
    ("acos", -1.0, 1.0, np.arccos, 2.0),
    ("exp", -30.0, 10.0, np.exp, 1.5),
])
def test_unary(name, lo, hi, fn, bound):
    x = np.random.default_rng(1).uniform(lo, hi, N).astype(np.float32)
    assert ulp_err(O.math_fn(name, x), fn(x.astype(np.float64))) <= bound


def test_atan_and_atan2():
    rng = np.random.default_rng(2)
    x = np.abs(rng.standard_cauchy(N)).astype(np.float32)
    assert ulp_err(O.math_fn("atan", x), np.arctan(x.astype(np.float64))) <= 3.5
    a, b = rng.standard_normal(N).astype(np.float32), rng.standard_normal(N).astype(np.float32)
    assert ulp_err(O.math_fn("atan2", a, b), np.arctan2(a.astype(np.float64), b.astype(np.float64))) <= 4.0


def test_log_and_pow():
    rng = np.random.default_rng(3)
    x = np.exp(rng.uniform(-40, 40, N)).astype(np.float32)
    assert ulp_err(O.math_fn("log", x), np.log(x.astype(np.float64))) <= 1.5
    x = rng.uniform(0, 1, N).astype(np.float32)
    g = np.full(N, 1.0 / 2.2, np.float32)
    assert ulp_err(O.math_fn("pow", x, g), x.astype(np.float64) ** np.float64(g[0])) <= 8.0
    five = np.full(N, 5.0, np.float32)
    assert ulp_err(O.math_fn("pow", x, five), x.astype(np.float64) ** 5) <= 3.5


def test_special_values():
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    # rand() can return exactly 1.0, so atan(alpha * sqrt(1/0)) = atan(inf) must be pi/2 (brdf.hpp:66)
    assert O.math_fn("atan", np.float32([inf]))[0] == np.float32(np.pi / 2)
    assert np.isnan(O.math_fn("atan", np.float32([nan]))[0])
    assert O.math_fn("exp", np.float32([-200.0, 0.0]))[0] == 0 and O.math_fn("exp", np.float32([0.0]))[0] == 1
    assert O.math_fn("pow", np.float32([0.0]), np.float32([1 / 2.2]))[0] == 0
    assert O.math_fn("acos", np.float32([1.0]))[0] == 0
    assert O.math_fn("sin", np.float32([0.0]))[0] == 0 and O.math_fn("cos", np.float32([0.0]))[0] == 1
