"""acn_detmath.h (the arithmetic contract shared by oracle and GPU) against glibc's libm, on the CPU."""
import numpy as np

OPS = {"sin": 0, "cos": 1, "tan": 2, "acos": 3, "log": 4, "exp": 5, "pow": 6, "sqrt": 7, "div": 8, "u64_to_f64": 9,
       "frexp_mant": 10}


def cpu_eval(lib, op, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    yp = None
    if y is not None:
        y = np.ascontiguousarray(y, dtype=np.float64)
        yp = y.ctypes.data
    lib.detmath_eval(OPS[op], x.ctypes.data, yp, out.ctypes.data, x.size)
    return out


def ulp_err(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_sin_cos_tan(detmath_cpu):
    rng = np.random.default_rng(1)
    x = rng.uniform(0, 2 * np.pi, 400000)
    s, c = np.sin(x), np.cos(x)
    ok = np.abs(s) > 1e-3
    assert ulp_err(cpu_eval(detmath_cpu, "sin", x)[ok], s[ok]).max() <= 1.0
    ok = np.abs(c) > 1e-3
    assert ulp_err(cpu_eval(detmath_cpu, "cos", x)[ok], c[ok]).max() <= 1.0
    # near the zeros the kernels stay absolutely accurate
    assert np.abs(cpu_eval(detmath_cpu, "sin", x) - s).max() < 2.3e-16
    assert np.abs(cpu_eval(detmath_cpu, "cos", x) - c).max() < 2.3e-16
    xt = rng.uniform(0, np.pi, 400000)
    ok = np.abs(np.cos(xt)) > 1e-3
    assert ulp_err(cpu_eval(detmath_cpu, "tan", xt)[ok], np.tan(xt)[ok]).max() <= 2.0
    # odd / even symmetry
    assert np.array_equal(cpu_eval(detmath_cpu, "sin", -x), -cpu_eval(detmath_cpu, "sin", x))
    assert np.array_equal(cpu_eval(detmath_cpu, "cos", -x), cpu_eval(detmath_cpu, "cos", x))


def test_acos_log_exp(detmath_cpu):
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, 400000)
    assert ulp_err(cpu_eval(detmath_cpu, "acos", x), np.arccos(x)).max() <= 1.0
    assert cpu_eval(detmath_cpu, "acos", np.array([1.0]))[0] == 0.0
    assert cpu_eval(detmath_cpu, "acos", np.array([-1.0]))[0] == np.pi
    assert np.isnan(cpu_eval(detmath_cpu, "acos", np.array([1.0000000000000002]))[0])
    xl = np.exp(rng.uniform(-12, 12, 400000))
    assert ulp_err(cpu_eval(detmath_cpu, "log", xl), np.log(xl)).max() <= 1.0
    assert cpu_eval(detmath_cpu, "log", np.array([1.0]))[0] == 0.0
    assert cpu_eval(detmath_cpu, "log", np.array([0.0]))[0] == -np.inf
    xe = rng.uniform(-40, 40, 400000)
    assert ulp_err(cpu_eval(detmath_cpu, "exp", xe), np.exp(xe)).max() <= 1.0
    assert cpu_eval(detmath_cpu, "exp", np.array([800.0]))[0] == np.inf
    assert cpu_eval(detmath_cpu, "exp", np.array([-800.0]))[0] == 0.0


def test_pow_colour_semantics(detmath_cpu):
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 1, 200000)
    y = rng.uniform(0, 6, 200000)
    got = cpu_eval(detmath_cpu, "pow", x, y)
    ref = np.power(x, y)
    assert (np.abs(got - ref) <= 1e-13 * ref + 1e-300).all()
    # exact identities the renderer relies on: gamma == 1 must not move a colour, black stays black
    assert np.array_equal(cpu_eval(detmath_cpu, "pow", x, np.ones_like(x)), x)
    assert np.array_equal(cpu_eval(detmath_cpu, "pow", np.zeros(4), np.array([0.7, 0.9, 1.0, 2.0])), np.zeros(4))
    assert cpu_eval(detmath_cpu, "pow", np.array([1e40]), np.array([0.9]))[0] > 1.0


def test_frexp_and_conversion(detmath_cpu):
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.normal(size=100000) * 10.0 ** rng.integers(-300, 300, 100000), [0.0, -0.0, 5e-324, -5e-324]])
    m, _ = np.frexp(x)
    assert np.array_equal(cpu_eval(detmath_cpu, "frexp_mant", x), m)
    u = rng.integers(0, 2 ** 63, 100000, dtype=np.uint64) * 2 + rng.integers(0, 2, 100000, dtype=np.uint64)
    got = cpu_eval(detmath_cpu, "u64_to_f64", u.view(np.float64))
    assert np.array_equal(got, u.astype(np.float64))
