"""CPU tests: the C-ABI shared library loads and exports every symbol include/*.h declares; without a GPU the
product fails loudly (no CPU fallback)."""
import ctypes as ct
import os
import re
import pytest
import parity_util as P

INCLUDE = os.path.join(P.ROOT, 'include')


def declared_functions(header):
    txt = open(os.path.join(INCLUDE, header)).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    names = re.findall(r'\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;', txt)
    return sorted(set(n for n in names if n not in ('defined',)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(P.pkg.PRODUCT_LIB):
        pytest.skip('libupside_hip.so not built (run __graft_entry__.build())')
    return ct.CDLL(P.pkg.PRODUCT_LIB)


@pytest.mark.parametrize('header', ['upside_engine_c.h', 'upside_hip_kernels.h'])
def test_every_declared_symbol_is_exported(lib, header):
    names = declared_functions(header)
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, 'declared in %s but not exported: %s' % (header, missing)


def test_reference_abi_names_present(lib):
    # /root/reference/src/engine_c_library.h:12-32 + main.h
    for n in ['construct_deriv_engine', 'free_deriv_engine', 'evaluate_energy', 'evaluate_deriv', 'set_param',
              'get_param_deriv', 'get_param', 'get_output_dims', 'get_output', 'get_sens', 'get_value_by_name',
              'clamped_spline_solve', 'clamped_spline_value', 'get_clamped_value_and_deriv',
              'get_clamped_coeff_deriv', 'upside_main']:
        assert hasattr(lib, n), n


def test_engine_free_spline_helpers_match_oracle(lib):
    import numpy as np
    if not os.path.exists(P.ORACLE_LIB):
        pytest.skip('oracle not built')
    prod = P.pkg.UpsideLibrary(P.pkg.PRODUCT_LIB)
    orc = P.oracle_library()
    rs = np.random.RandomState(3)
    vals = rs.normal(size=14).astype('f4')
    c1, c2 = prod.clamped_spline_solve(vals), orc.clamped_spline_solve(vals)
    assert np.array_equal(c1, c2)
    x = np.linspace(-0.5, 15.5, 101).astype('f4')
    assert np.array_equal(prod.clamped_spline_value(c1, x), orc.clamped_spline_value(c1, x))
    assert np.array_equal(prod.clamped_value_and_deriv(c1, x), orc.clamped_value_and_deriv(c1, x))
    assert np.array_equal(prod.clamped_coeff_deriv(c1, x), orc.clamped_coeff_deriv(c1, x))


def test_no_cpu_fallback(lib):
    """without a GPU, construction must fail (NULL) with an explicit message instead of computing on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    lib.construct_deriv_engine.restype = ct.c_void_p
    lib.construct_deriv_engine.argtypes = [ct.c_int, ct.c_char_p, ct.c_bool]
    lib.upside_hip_last_error.restype = ct.c_char_p
    e = lib.construct_deriv_engine(60, P.fixture('trpcage20_7A').encode(), True)
    assert not e
    assert b'no HIP device' in lib.upside_hip_last_error()
    with pytest.raises(RuntimeError):
        P.pkg.Upside(P.fixture('trpcage20_7A'))


def test_plugin_library_registers_external_node_types(lib):
    """the plug-in side of the boundary (include/upside_hip_plugin.h; the reference's registry, deriv_engine.h:239-335):
    a shared library built against include/ only registers its node types through its static initialisers when loaded"""
    plug = os.path.join(P.ROOT, 'tests', 'plugin', 'libhost_pull.so')
    if not os.path.exists(plug):
        pytest.skip('tests/plugin/libhost_pull.so not built (run __graft_entry__.build())')
    lib.upside_hip_load_plugin.argtypes = [ct.c_char_p]
    lib.upside_hip_node_type_registered.argtypes = [ct.c_char_p]
    lib.upside_hip_last_error.restype = ct.c_char_p
    for builtin in (b'rotamer', b'environment_coverage', b'atom_pos_spring', b'pos'):
        assert lib.upside_hip_node_type_registered(builtin) == 1
    assert lib.upside_hip_load_plugin(b'/nonexistent/libnothing.so') == 1
    assert b'libnothing' in lib.upside_hip_last_error()
    assert lib.upside_hip_load_plugin(plug.encode()) == 0, lib.upside_hip_last_error()
    assert lib.upside_hip_node_type_registered(b'host_pull') == 1 and lib.upside_hip_node_type_registered(b'host_scale') == 1
    assert lib.upside_hip_node_type_registered(b'host_nothing') == 0
    assert lib.upside_hip_load_plugin(plug.encode()) == 0      # twice: no-op, no duplicate-prefix error


def test_quadspline_polynomial_table_equals_the_splines(lib):
    """the per-interval polynomial tables the pair passes stage in LDS (nodes.cpp: quadspline_poly_row) against a direct
    evaluation of the reference's splines (spline.h:136-174 de Boor cubic B-spline; angular: unclamped on [-1, 1],
    spline.h:228-242; radial: clamped ends, spline.h:275-310) at many points of every interval, values and slopes"""
    import numpy as np
    lib.upside_hip_quadspline_poly_width.argtypes = [ct.c_int, ct.c_int]
    lib.upside_hip_quadspline_poly_row.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p]
    rs = np.random.RandomState(7)

    def bspline(c, x):            # value and derivative of the uniform cubic B-spline with coefficients c at knot coordinate x
        b = int(np.floor(x)); y = x - b
        w = np.array([(1 - y) ** 3, 3 * y ** 3 - 6 * y ** 2 + 4, -3 * y ** 3 + 3 * y ** 2 + 3 * y + 1, y ** 3]) / 6.
        d = np.array([-3 * (1 - y) ** 2, 9 * y ** 2 - 12 * y, -9 * y ** 2 + 6 * y + 3, 3 * y ** 2]) / 6.
        win = c[b - 1:b + 3]
        return float(win @ w), float(win @ d)

    for ka, k in ((8, 12), (15, 12), (8, 7), (15, 16)):
        p = rs.normal(size=2 * ka + 2 * k).astype('f4')
        n = lib.upside_hip_quadspline_poly_width(ka, k)
        assert n == 8 * (ka - 3) + 8 * (k - 1)
        poly = np.zeros(n, 'f4')
        assert lib.upside_hip_quadspline_poly_row(p.ctypes.data, ka, k, poly.ctypes.data) == 0
        pd = p.astype('f8')

        def cubic(c4, y):
            return c4[0] + y * (c4[1] + y * (c4[2] + y * c4[3])), c4[1] + y * (2 * c4[2] + 3 * y * c4[3])
        # angular splines: knot coordinate x = (cos + 1) * (ka - 3) / 2 + 1 in [1, ka - 2]; interval i = floor(x) - 1
        for a in range(2):
            for x in np.linspace(1., ka - 2 - 1e-6, 97):
                i = min(int(np.floor(x)) - 1, ka - 4)
                v, d = cubic(poly[(a * (ka - 3) + i) * 4:][:4].astype('f8'), x - 1 - i)
                rv, rd = bspline(pd[a * ka:(a + 1) * ka], x)
                assert abs(v - rv) < 2e-6 * (1 + abs(rv)) and abs(d - rd) < 1e-5 * (1 + abs(rd)), (ka, k, a, x)
        # radial splines: interior as above; below 1 and from k - 2 on the clamped constants with zero slope
        rad = poly[8 * (ka - 3):].astype('f8')
        for w in range(2):
            c = pd[2 * ka + w * k:][:k]
            for x in np.concatenate((np.linspace(0., 0.999, 5), np.linspace(1., k - 2 - 1e-6, 131), np.linspace(k - 2, k + 3., 7))):
                i = min(int(np.floor(x)), k - 2)
                v, d = cubic(rad[i * 8 + w * 4:][:4], x - i)
                if x < 1.:
                    rv, rd = (c[0] + 4 * c[1] + c[2]) / 6., 0.
                elif x >= k - 2:
                    rv, rd = (c[k - 3] + 4 * c[k - 2] + c[k - 1]) / 6., 0.
                else:
                    rv, rd = bspline(c, x)
                assert abs(v - rv) < 2e-6 * (1 + abs(rv)) and abs(d - rd) < 1e-5 * (1 + abs(rd)), (ka, k, w, x)
