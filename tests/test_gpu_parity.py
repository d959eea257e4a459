"""GPU parity tests (run with -m gpu on an MI355X): the HIP product, driven through its C-ABI, against the CPU
oracle on the same seeded inputs, plus size-independent properties at the benchmark size.

Tolerance (BASELINE.json north_star): pair-list indices bit-exact; forces / energies / node values within 1e-5
relative fp32, where "relative" for an array is the reference's own relative RMS deviation
(/root/reference/src/deriv_engine.h:345-357) and for the total energy is relative to the sum of the magnitudes of
the per-node potentials (the total is a difference of large terms)."""
import ctypes as ct
import os
import sys
import numpy as np
import pytest
import parity_util as P

pytestmark = pytest.mark.gpu
RTOL = 1e-5
FIXTURES = ['trpcage20_7A', 'proteinG56_7A', 'syn150_10A', 'syn300_10A', 'syn300_7A']
IGRAPH_NODES = ['rotamer', 'hbond_coverage', 'hbond_coverage_hydrophobe', 'environment_coverage', 'protein_hbond']


@pytest.fixture(scope='module')
def hip():
    import torch
    assert torch.cuda.is_available(), 'these tests need a GPU'
    lib = P.pkg.default_library()     # raises when the HIP extension is missing: no fallback
    c = lib.calc
    c.upside_hip_construct.restype = ct.c_void_p
    c.upside_hip_construct.argtypes = [ct.c_int, ct.c_char_p, ct.c_int, ct.c_bool]
    c.upside_hip_get_pairlist.restype = ct.c_int
    c.upside_hip_get_pairlist.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p]
    c.upside_hip_rotamer_iterations.argtypes = [ct.c_void_p, ct.c_void_p]
    c.upside_hip_set_pos.argtypes = [ct.c_void_p, ct.c_void_p]
    c.upside_hip_get_pos.argtypes = [ct.c_void_p, ct.c_void_p]
    c.upside_hip_get_mom.argtypes = [ct.c_void_p, ct.c_void_p]
    c.upside_hip_set_mom.argtypes = [ct.c_void_p, ct.c_void_p]
    c.upside_hip_compute.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p]
    c.upside_hip_init_md.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_uint32, ct.c_float, ct.c_float, ct.c_int]
    c.upside_hip_run_md.argtypes = [ct.c_void_p, ct.c_int]
    c.upside_hip_replica_swap.argtypes = [ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_uint32, ct.c_uint64, ct.c_void_p]
    c.upside_hip_last_error.restype = ct.c_char_p
    return lib


def hip_pairlist(up, node, sys=0, cap=400000):
    i1 = np.zeros(cap, 'i4'); i2 = np.zeros(cap, 'i4')
    n = up.calc.upside_hip_get_pairlist(up.engine, node.encode(), sys, cap, i1.ctypes.data, i2.ctypes.data)
    assert 0 <= n <= cap
    return np.column_stack((i1[:n], i2[:n]))


def assert_close(ref, act, keys=None):
    bad = P.compare(ref, act, keys=keys, rtol=RTOL)
    assert not bad, 'outside 1e-5 relative: %r' % (bad,)


@pytest.mark.parametrize('name', FIXTURES)
def test_force_pass_matches_oracle(hip, name):
    """every node output / sensitivity, per-node potentials, total energy and forces; first call (pair list
    built) and a second, perturbed call through the cached pair list."""
    g = P.golden(name)
    up = P.pkg.Upside(P.fixture(name))
    orc = P.pkg.Upside(P.fixture(name), library=P.oracle_library())
    for tag in ('pos', 'pos2'):
        ref = P.evaluate_all(orc, g[tag])
        act = P.evaluate_all(up, g[tag])
        keys = [k for k in ref if k != 'energy']
        assert_close(ref, act, keys)
        scale = sum(abs(float(ref['pot/' + k])) for k in P.POTENTIAL_NODES)
        assert abs(float(ref['energy']) - float(act['energy'])) <= RTOL * scale
        # pair-list indices: bit-exact, canonical order
        for node in IGRAPH_NODES:
            po = P.oracle_pairlist(orc, node)
            ph = hip_pairlist(up, node)
            assert ph.shape == po.shape and np.array_equal(ph, po), (node, tag, ph.shape, po.shape)
        it = np.zeros(1, 'i4')
        up.calc.upside_hip_rotamer_iterations(up.engine, it.ctypes.data)
        assert int(it[0]) == orc.calc.oracle_rotamer_iterations(orc.engine)
        # marginals (what predict_chi1.py reads): node and pair marginals in graph order
        n = int(g['rotamer/n_node'])
        if n <= 60:
            em_h = up.get_value_by_name((n, n, 6, 6), 'rotamer', 'edge_marginal_in_graph_order')
            em_o = orc.get_value_by_name((n, n, 6, 6), 'rotamer', 'edge_marginal_in_graph_order')
            assert np.abs(em_h - em_o).max() < 2e-6
    up.close(); orc.close()


@pytest.mark.parametrize('name', FIXTURES)
def test_forces_match_reference_golden(hip, name):
    """the committed golden vectors of the unmodified reference, all five fixtures, both structures (the second one through
    the cached pair lists).  Forces and sensitivities within P.golden_tol = max(1e-5, 2 x the reference's own build-to-build
    spread on that fixture and structure): 1e-5 on the benchmark fixture syn300_10A and on trpcage20_7A's first structure,
    up to 9e-5 on the over-compact syn150_10A (tests/golden/reference_noise_floor.json); node outputs within 1e-5; the
    reference's own canonical pair list of the side-chain graph bit for bit."""
    g = P.golden(name)
    up = P.pkg.Upside(P.fixture(name))
    for tag, dkey in (('pos', 'deriv'), ('pos2', 'deriv2')):
        act = P.evaluate_all(up, g[tag])
        err = P.rel_rms(g[dkey], act['deriv'])
        assert err < P.golden_tol(name, tag, 'deriv'), (tag, err, P.golden_tol(name, tag, 'deriv'))
        if tag == 'pos':
            for k in g:
                if k.startswith('out/'):
                    assert P.rel_rms(g[k], act[k]) < 1e-5, k
                elif k.startswith('sens/') and k in act:
                    assert P.rel_rms(g[k], act[k]) < P.golden_tol(name, tag, 'sens'), (k, P.rel_rms(g[k], act[k]))
            assert np.array_equal(hip_pairlist(up, 'rotamer'), g['pairlist/edges'][:, :2])
            for node in IGRAPH_NODES[1:]:      # the four asymmetric graphs against the reference's own edge lists
                assert np.array_equal(hip_pairlist(up, node), g['pairlist/' + node]), node
    up.close()


@pytest.mark.parametrize('name', FIXTURES)
def test_param_derivs_match_reference_golden(hip, name):
    """get_param_deriv (engine_c_library.h:20) against the reference compiled with -DPARAM_DERIV (golden vectors from
    tools/make_fixtures.py).  Tolerance: 1e-5 relative RMS of the table (measured 2e-7 .. 7e-6) except the fixed
    placements, whose derivative is a sum of the sensitivities that flow back through the belief-propagation solve:
    5e-5 there (measured up to 2.1e-5; the reference's own converged marginals spread more than that between
    builds, profiles/r01_reference_noise_floor.txt)."""
    g = P.golden(name)
    up = P.pkg.Upside(P.fixture(name))
    up.deriv(g['pos'])
    keys = sorted(k for k in g if k.startswith('param_deriv/'))
    assert len(keys) >= 9
    for k in keys:
        node = k.split('/', 1)[1]
        act = up.get_param_deriv(g[k].shape, node)
        if not np.any(g[k]):
            assert not np.any(act), k                    # environment_coverage: zeros (environment.cpp:62-65)
            continue
        tol = 5e-5 if node.startswith('placement') else 1e-5
        assert P.rel_rms(g[k], act) < tol, (k, P.rel_rms(g[k], act))
        assert np.array_equal(g[k] == 0, act == 0) or node.startswith('placement'), k   # same support in the tables
    up.close()


def test_optional_restraint_nodes_match_oracle_and_reference(hip):
    """the optional restraint / external-field nodes (z_flat_bottom, tension, AFM, atom_pos_spring, contact,
    membrane_potential, linear_coupling_uniform / _with_inactivation, slice, placement_fixed_point_only): energies,
    forces, node outputs and sensitivities against the oracle (1e-5) and the reference's golden vectors"""
    name = 'proteinG56_restraints'
    g = P.golden(name)
    up = P.pkg.Upside(P.fixture(name))
    orc = P.pkg.Upside(P.fixture(name), library=P.oracle_library())
    act = P.evaluate_all(up, g['pos'], P.RESTRAINT_COORDS, P.RESTRAINT_POTENTIALS)
    ref = P.evaluate_all(orc, g['pos'], P.RESTRAINT_COORDS, P.RESTRAINT_POTENTIALS)
    keys = [k for k in ref if k != 'energy']
    assert_close(ref, act, keys=keys)
    scale = sum(abs(float(ref['pot/' + k])) for k in P.POTENTIAL_NODES + P.RESTRAINT_POTENTIALS)
    assert abs(float(ref['energy']) - float(act['energy'])) < RTOL * scale
    assert P.rel_rms(g['deriv'], act['deriv']) < P.golden_tol(name, 'pos', 'deriv')      # 2 x the reference's own spread, as above
    for k in g:
        if k.startswith('pot/') and k[4:] in P.RESTRAINT_POTENTIALS:
            assert abs(float(g[k]) - float(act[k])) < RTOL * max(1., abs(float(g[k]))), k
        if k.startswith('param_deriv/'):
            assert P.rel_rms(g[k], up.get_param_deriv(g[k].shape, k.split('/', 1)[1])) < RTOL, k
    # a second structure, through the cached pair lists, oracle only
    act2 = P.evaluate_all(up, g['pos'] + np.float32(0.05) * np.random.RandomState(3).normal(size=g['pos'].shape).astype('f4'),
                          P.RESTRAINT_COORDS, P.RESTRAINT_POTENTIALS)
    ref2 = P.evaluate_all(orc, g['pos'] + np.float32(0.05) * np.random.RandomState(3).normal(size=g['pos'].shape).astype('f4'),
                          P.RESTRAINT_COORDS, P.RESTRAINT_POTENTIALS)
    assert_close(ref2, act2, keys=keys)
    up.close(); orc.close()


@pytest.mark.parametrize('name', ['edge_gly5', 'edge_pro6', 'edge_awa3'])
def test_degenerate_sequences(hip, name):
    """empty interaction classes and tiny systems: all-glycine (belief propagation without a single edge), all-proline
    (one donor site), three residues; force pass against the oracle and the reference's golden vectors, pair lists
    bit-exact, then a short batched MD run that must stay finite"""
    g = P.golden(name)
    up = P.pkg.Upside(P.fixture(name))
    orc = P.pkg.Upside(P.fixture(name), library=P.oracle_library())
    ref = P.evaluate_all(orc, g['pos']); act = P.evaluate_all(up, g['pos'])
    assert_close(ref, act, keys=[k for k in ref if k != 'energy' and np.asarray(ref[k]).size])
    scale = sum(abs(float(ref['pot/' + k])) for k in P.POTENTIAL_NODES)
    assert abs(float(ref['energy']) - float(act['energy'])) <= RTOL * scale
    assert P.rel_rms(g['deriv'], act['deriv']) < P.golden_tol(name, 'pos', 'deriv')
    assert abs(float(g['energy']) - float(act['energy'])) < 1e-4 * max(1., scale)
    for node in IGRAPH_NODES:
        assert np.array_equal(hip_pairlist(up, node), P.oracle_pairlist(orc, node)), node
    up.close(); orc.close()
    c = hip.calc
    n_atom = g['pos'].shape[0]
    e = c.upside_hip_construct(n_atom, P.fixture(name).encode(), 5, True)
    x = np.tile(g['pos'][None], (5, 1, 1)).astype('f4')
    assert c.upside_hip_set_pos(e, x.ctypes.data) == 0
    temps = np.full(5, 0.8, 'f4')
    assert c.upside_hip_init_md(e, temps.ctypes.data, 3, 5.0, 0.009, 1) == 0
    assert c.upside_hip_run_md(e, 200) == 0
    assert c.upside_hip_get_pos(e, x.ctypes.data) == 0
    assert np.isfinite(x).all() and np.abs(x).max() < 100.
    en = np.zeros(5, 'f4')
    assert c.upside_hip_compute(e, en.ctypes.data, None) == 0 and np.isfinite(en).all()
    c.free_deriv_engine(ct.c_void_p(e))


@pytest.mark.parametrize('name', ['trpcage20_7A', 'proteinG56_7A', 'syn300_10A'])
def test_rotamer_named_values(hip, name):
    """get_value_by_name of the side-chain node (rotamer.cpp:675-773): node energies with the 1-state partners folded in,
    per-residue free energies and belief-weighted 1-body energies, against the reference's golden values (tolerances of
    tests/test_oracle_pinning.py) and the oracle; the read-out must leave the engine state intact."""
    g = P.golden(name)
    up = P.pkg.Upside(P.fixture(name))
    orc = P.pkg.Upside(P.fixture(name), library=P.oracle_library())
    e0 = up.energy(g['pos']); orc.energy(g['pos'])
    n = int(g['rotamer/n_node'])
    ne = up.get_value_by_name((n, 6), 'rotamer', 'node_energy')
    m = g['rotamer/node_energy'] < 1e4
    assert np.abs(ne - g['rotamer/node_energy'])[m].max() < 5e-4 and np.array_equal(ne[~m], g['rotamer/node_energy'][~m])
    assert np.abs(ne - orc.get_value_by_name((n, 6), 'rotamer', 'node_energy'))[m].max() < 2e-5
    fe = up.get_value_by_name((n,), 'rotamer', 'rotamer_free_energy')
    assert np.abs(fe - g['rotamer/rotamer_free_energy']).max() < 2e-4
    assert np.abs(fe - orc.get_value_by_name((n,), 'rotamer', 'rotamer_free_energy')).max() < 5e-5
    e1 = up.get_value_by_name((n, 3), 'rotamer', 'rotamer_1body_energy')
    assert np.abs(e1 - g['rotamer/rotamer_1body_energy']).max() < 2e-4
    if n <= 60:
        ee = up.get_value_by_name((n, n, 6, 6), 'rotamer', 'edge_energy')
        eo = orc.get_value_by_name((n, n, 6, 6), 'rotamer', 'edge_energy')
        assert np.array_equal(ee != 0, eo != 0) and np.abs(ee - eo).max() < 2e-5 * max(1., np.abs(eo).max())
    assert int(up.get_value_by_name((1,), 'rotamer', 'read n_bad_solve and reset')[0]) == 0
    assert abs(float(fe.sum()) - float(up.get_output('rotamer')[0, 0])) < 1e-3 * max(1., np.abs(fe).sum())   # the parts add up to the node's potential
    assert up.energy(g['pos']) == e0 and np.array_equal(up.deriv(g['pos']), up.deriv(g['pos']))              # state left clean
    up.close(); orc.close()


def test_truncated_solve_is_counted_and_matches_oracle(hip, tmp_path):
    """a belief-propagation solve cut off by max_iter: same forces as the oracle's equally truncated solve, and the
    node counts it (rotamer.cpp:784-785, `read n_bad_solve`)"""
    import shutil
    from upside_md_amd import h5lite
    name = 'proteinG56_7A'
    f = str(tmp_path / 'short.up')
    shutil.copyfile(P.fixture(name), f)
    with h5lite.open_file(f, 'r+') as h:
        h.group('input/potential/rotamer').set_attr('max_iter', 6)
    g = P.golden(name)
    up = P.pkg.Upside(f); orc = P.pkg.Upside(f, library=P.oracle_library())
    for k in range(3):
        assert P.rel_rms(orc.deriv(g['pos']), up.deriv(g['pos'])) < RTOL
    it = np.zeros(1, 'i4'); up.calc.upside_hip_rotamer_iterations(up.engine, it.ctypes.data)
    assert int(it[0]) == 6 == orc.calc.oracle_rotamer_iterations(orc.engine)
    assert int(up.get_value_by_name((1,), 'rotamer', 'read n_bad_solve')[0]) == 3
    assert int(up.get_value_by_name((1,), 'rotamer', 'read n_bad_solve and reset')[0]) == 3
    assert int(up.get_value_by_name((1,), 'rotamer', 'read n_bad_solve')[0]) == 0
    up.close(); orc.close()


def test_param_deriv_of_every_system(hip):
    """the batched extension returns each system's own derivative"""
    name = 'proteinG56_7A'
    g = P.golden(name)
    c = hip.calc
    c.upside_hip_get_param_deriv.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_int, ct.c_int, ct.c_void_p]
    e = c.upside_hip_construct(g['pos'].shape[0], P.fixture(name).encode(), 3, True)
    x = np.stack([g['pos'], g['pos2'], g['pos']]).astype('f4')
    assert c.upside_hip_set_pos(e, x.ctypes.data) == 0
    en = np.zeros(3, 'f4')
    assert c.upside_hip_compute(e, en.ctypes.data, None) == 0
    shp = g['param_deriv/rotamer'].shape
    out = np.zeros((3,) + shp, 'f4')
    for s in range(3):
        assert c.upside_hip_get_param_deriv(e, b'rotamer', s, int(np.prod(shp)), out[s].ctypes.data) == 0
    assert c.upside_hip_get_param_deriv(e, b'rotamer', 3, int(np.prod(shp)), out[0].ctypes.data) == 1   # no such system
    assert P.rel_rms(g['param_deriv/rotamer'], out[0]) < 1e-5
    assert P.rel_rms(out[0], out[2]) < 1e-6
    assert P.rel_rms(out[0], out[1]) > 1e-3          # a different structure
    orc = P.pkg.Upside(P.fixture(name), library=P.oracle_library())    # ... whose derivative the oracle confirms
    orc.deriv(g['pos2'])
    assert P.rel_rms(orc.get_param_deriv(shp, 'rotamer'), out[1]) < RTOL
    orc.close()
    c.free_deriv_engine(ct.c_void_p(e))


def test_empty_and_degenerate_requests(hip):
    """error behaviour of the C-ABI mirrors engine_c_library.cpp: wrong sizes and unknown names return 1"""
    up = P.pkg.Upside(P.fixture('trpcage20_7A'))
    up.energy(up.initial_pos)
    buf = np.zeros(5, 'f4')
    assert up.calc.get_output(5, buf.ctypes.data, up.engine, b'rama_coord') == 1        # wrong size
    assert up.calc.get_output(1, buf.ctypes.data, up.engine, b'no_such_node') == 1
    assert up.calc.get_param_deriv(1, buf.ctypes.data, up.engine, b'rotamer') == 1      # wrong size (engine_c_library.cpp:96-98)
    assert up.calc.get_param_deriv(0, buf.ctypes.data, up.engine, b'protein_hbond') == 0   # no derivative: empty vector
    assert up.calc.get_output(1, buf.ctypes.data, up.engine, b'rotamer') == 0           # potential node -> (1,1)
    with pytest.raises((RuntimeError, OSError)):
        P.pkg.Upside('/nonexistent/file.up')
    # MD steps before upside_hip_init_md: an error with a message, not a kernel launched on momenta that were never allocated
    up.calc.upside_hip_run_steps.argtypes = [ct.c_void_p, ct.c_int]
    assert up.calc.upside_hip_run_steps(up.engine, 3) == 1
    up.calc.upside_hip_last_error.restype = ct.c_char_p
    assert b'upside_hip_init_md' in up.calc.upside_hip_last_error()
    up.energy(up.initial_pos)                     # the engine is still usable
    up.close()


def test_batched_systems_are_independent_and_identical(hip):
    """S copies of one system in one engine give S times the single-system answer (replica independence)."""
    name = 'proteinG56_7A'
    c = hip.calc
    g = P.golden(name)
    n_atom = g['pos'].shape[0]
    S = 5
    eng = c.upside_hip_construct(n_atom, P.fixture(name).encode(), S, True)
    assert eng
    pos = np.stack([g['pos'], g['pos2'], g['pos'], g['pos2'], g['pos']]).astype('f4')
    assert c.upside_hip_set_pos(eng, pos.ctypes.data) == 0
    en = np.zeros(S, 'f4'); der = np.zeros((S, n_atom, 3), 'f4')
    assert c.upside_hip_compute(eng, en.ctypes.data, der.ctypes.data) == 0
    assert en[0] == en[2] == en[4] and en[1] == en[3]
    assert np.array_equal(der[0], der[2]) and np.array_equal(der[1], der[3])
    # a fresh single-system engine per structure reproduces the batched result bit for bit (same pair-list
    # build, same summation order); an engine that reaches pos2 through the cached list of pos only to rounding
    for tag, s in (('pos', 0), ('pos2', 1)):
        single = P.pkg.Upside(P.fixture(name))
        assert np.array_equal(single.deriv(g[tag]), der[s])
        single.close()
    single = P.pkg.Upside(P.fixture(name))
    single.deriv(g['pos'])
    assert P.rel_rms(der[1], single.deriv(g['pos2'])) < 1e-6
    c.free_deriv_engine(ct.c_void_p(eng)); single.close()


def test_thermostat_and_integrator_match_oracle(hip):
    """Threefry/Box-Muller momenta and a short leapfrog trajectory against the oracle's MD loop
    (main.cpp:515-523,616-667).  Momenta after initialisation agree to rounding of sinf/cosf/logf; positions are
    compared after few enough steps that chaotic growth stays below 1e-4."""
    name = 'trpcage20_7A'
    c = hip.calc
    orc = P.pkg.Upside(P.fixture(name), library=P.oracle_library())
    pos0 = orc.initial_pos.copy(); n_atom = pos0.shape[0]
    T, seed, dt = 0.8, 12345, 0.009
    eng = c.upside_hip_construct(n_atom, P.fixture(name).encode(), 2, True)
    both = np.stack([pos0, pos0]).astype('f4')
    assert c.upside_hip_set_pos(eng, both.ctypes.data) == 0
    temps = np.array([T, T], 'f4')
    assert c.upside_hip_init_md(eng, temps.ctypes.data, seed, 5.0, dt, 1) == 0
    mom_h = np.zeros((2, n_atom, 3), 'f4')
    assert c.upside_hip_get_mom(eng, mom_h.ctypes.data) == 0
    for s in range(2):      # system s uses seed + s (main.cpp:459)
        p = pos0.copy(); m = np.zeros_like(p)
        orc.calc.oracle_run_md(orc.engine, p.ctypes.data, m.ctypes.data, 0, dt, T, seed + s, 5.0, 1)
        assert np.abs(m - mom_h[s]).max() < 2e-6 * np.abs(m).max()
    n_round = 8
    assert c.upside_hip_run_md(eng, n_round) == 0
    pos_h = np.zeros((2, n_atom, 3), 'f4'); c.upside_hip_get_pos(eng, pos_h.ctypes.data)
    c.upside_hip_get_mom(eng, mom_h.ctypes.data)
    for s in range(2):
        p = pos0.copy(); m = np.zeros_like(p)
        orc.calc.oracle_run_md(orc.engine, p.ctypes.data, m.ctypes.data, n_round, dt, T, seed + s, 5.0, 1)
        assert P.rel_rms(p, pos_h[s]) < 1e-5
        assert P.rel_rms(m, mom_h[s]) < 1e-3
    assert not np.array_equal(pos_h[0], pos_h[1])
    c.free_deriv_engine(ct.c_void_p(eng))


def test_predescu_integrator_matches_oracle(hip):
    """IntegratorType Predescu (deriv_engine.h:230, the stage weights of deriv_engine.cpp:173-180) through upside_hip_set_integrator:
    a six-step (two-cycle) trajectory equals the oracle's loop with the same constants and differs from the Verlet trajectory."""
    name = 'trpcage20_7A'
    c = hip.calc
    c.upside_hip_set_integrator.argtypes = [ct.c_void_p, ct.c_int]
    orc = P.pkg.Upside(P.fixture(name), library=P.oracle_library())
    orc.calc.oracle_run_md_integrator.restype = ct.c_int
    orc.calc.oracle_run_md_integrator.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_float, ct.c_float, ct.c_uint32,
                                                  ct.c_float, ct.c_int, ct.c_int]
    pos0 = orc.initial_pos.copy(); n_atom = pos0.shape[0]
    T, seed, dt, n_round = 0.8, 4321, 0.009, 2
    out = {}
    for kind in (0, 1):
        eng = c.upside_hip_construct(n_atom, P.fixture(name).encode(), 1, True)
        x = pos0[None].astype('f4').copy(); temps = np.array([T], 'f4')
        assert c.upside_hip_set_pos(eng, x.ctypes.data) == 0
        assert c.upside_hip_init_md(eng, temps.ctypes.data, seed, 5.0, dt, 1) == 0
        assert c.upside_hip_set_integrator(ct.c_void_p(eng), 7) != 0            # refused with a message
        assert c.upside_hip_set_integrator(ct.c_void_p(eng), kind) == 0
        assert c.upside_hip_run_md(eng, n_round) == 0
        pos_h = np.zeros((1, n_atom, 3), 'f4'); mom_h = np.zeros((1, n_atom, 3), 'f4')
        c.upside_hip_get_pos(eng, pos_h.ctypes.data); c.upside_hip_get_mom(eng, mom_h.ctypes.data)
        c.free_deriv_engine(ct.c_void_p(eng))
        p = pos0.copy(); m = np.zeros_like(p)
        assert orc.calc.oracle_run_md_integrator(orc.engine, p.ctypes.data, m.ctypes.data, n_round, dt, T, seed, 5.0, 1, kind) == 0
        assert P.rel_rms(p, pos_h[0]) < 1e-6 and P.rel_rms(m, mom_h[0]) < 1e-4, (kind, P.rel_rms(p, pos_h[0]), P.rel_rms(m, mom_h[0]))
        out[kind] = pos_h[0].copy()
    assert P.rel_rms(out[0], out[1]) > 1e-5          # the two integrators do take different steps


def test_cached_pairlist_path_after_md(hip):
    """after MD steps (cached Verlet lists, some rebuilds) the device's in-range pair lists and forces still equal
    the oracle's evaluated at the device's current coordinates"""
    name = 'proteinG56_7A'
    c = hip.calc
    g = P.golden(name); n_atom = g['pos'].shape[0]
    eng = c.upside_hip_construct(n_atom, P.fixture(name).encode(), 1, True)
    pos = g['pos'][None].astype('f4').copy()
    c.upside_hip_set_pos(eng, pos.ctypes.data)
    temps = np.array([0.9], 'f4')
    c.upside_hip_init_md(eng, temps.ctypes.data, 7, 5.0, 0.009, 1)
    orc = P.pkg.Upside(P.fixture(name), library=P.oracle_library())
    for chunk in range(3):
        assert c.upside_hip_run_md(eng, 15) == 0
        x = np.zeros((1, n_atom, 3), 'f4'); c.upside_hip_get_pos(eng, x.ctypes.data)
        en = np.zeros(1, 'f4'); der = np.zeros((1, n_atom, 3), 'f4')
        assert c.upside_hip_compute(eng, en.ctypes.data, der.ctypes.data) == 0
        d_ref = orc.deriv(x[0])
        assert P.rel_rms(d_ref, der[0]) < RTOL
        up_view = type('V', (), {'calc': c, 'engine': ct.c_void_p(eng)})
        for node in IGRAPH_NODES:
            assert np.array_equal(hip_pairlist(up_view, node), P.oracle_pairlist(orc, node)), node
    c.free_deriv_engine(ct.c_void_p(eng))


def test_list_statistics_agree_with_the_pair_lists(hip):
    """upside_hip_igraph_stats (diagnostics, tools/list_stats.py): after MD steps the per-system sums of the in-range (hit) list
    lengths of every graph equal the sizes of its canonical pair list -- the rows the pair passes walk hold exactly the
    reference's edges (interaction_graph.h:201-257) --, cached lists are at least as long, capacities are not exceeded."""
    name = 'proteinG56_7A'
    c = hip.calc
    c.upside_hip_igraph_stats.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_void_p]
    g = P.golden(name); n_atom = g['pos'].shape[0]
    S = 3
    eng = c.upside_hip_construct(n_atom, P.fixture(name).encode(), S, True)
    pos = np.ascontiguousarray(np.stack([g['pos'], g['pos2'], g['pos']]).astype('f4'))
    c.upside_hip_set_pos(eng, pos.ctypes.data)
    temps = np.full(S, 0.85, 'f4')
    c.upside_hip_init_md(eng, temps.ctypes.data, 3, 5.0, 0.009, 1)
    assert c.upside_hip_run_md(eng, 12) == 0
    en = np.zeros(S, 'f4')
    assert c.upside_hip_compute(eng, en.ctypes.data, None) == 0
    up_view = type('V', (), {'calc': c, 'engine': ct.c_void_p(eng)})
    out = np.zeros(11)
    for node in IGRAPH_NODES:
        assert c.upside_hip_igraph_stats(eng, node.encode(), out.ctypes.data) == 0, node
        n1, n2, cap1, cap2, cut, cache_cut, c1, c2, h1, h2, sides = out
        assert cache_cut > cut > 0 and n1 > 0 and n2 > 0
        edges = sum(len(hip_pairlist(up_view, node, sys=k)) for k in range(S)) / S
        symmetric = node == 'rotamer'
        for side, (cached, hits, n_rows, cap) in enumerate(((c1, h1, n1, cap1), (c2, h2, n2, cap2)), 1):
            if not (int(sides) & side) or (symmetric and side == 2):
                continue
            assert abs(hits - edges) < 1e-6 * max(edges, 1.), (node, side, hits, edges)     # (each pair once per walked side)
            assert cached >= hits and cached <= n_rows * cap
    assert c.upside_hip_igraph_stats(eng, b'pos', out.ctypes.data) != 0      # not an interaction-graph node
    c.free_deriv_engine(ct.c_void_p(eng))


def test_benchmark_size_properties(hip):
    """BASELINE-size checks that need no oracle run: translation invariance (sum of forces ~ 0), rigid-rotation
    invariance of the energy, energy conservation trend of the leapfrog integrator without thermostat noise,
    and finite-difference agreement of the analytic force along a random direction."""
    name = 'syn300_10A'
    up = P.pkg.Upside(P.fixture(name))
    x = up.initial_pos.copy()
    e0 = up.energy(x); f = up.deriv(x)
    assert np.abs(f.sum(axis=0)).max() < 5e-3 * np.abs(f).max()
    # rigid rotation + translation
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], 'f4')
    x2 = (x.dot(R.T) + np.array([1.5, -2., 0.5], 'f4')).astype('f4')
    e1 = up.energy(x2); f2 = up.deriv(x2)
    scale = sum(abs(float(up.get_output(k)[0, 0])) for k in P.POTENTIAL_NODES)
    assert abs(e1 - e0) < 2e-5 * scale
    assert P.rel_rms(f.dot(R.T), f2) < 1e-4
    # directional finite difference (deriv_engine.cpp:291-342 in one direction)
    rs = np.random.RandomState(0)
    d = rs.normal(size=x.shape).astype('f4'); d /= np.sqrt((d ** 2).sum())
    eps = 2e-2
    fd = (float(up.energy((x + eps * d).astype('f4'))) - float(up.energy((x - eps * d).astype('f4')))) / (2 * eps)
    an = float((f * d).sum())
    assert abs(fd - an) < 3e-2 * max(1., abs(an))
    up.close()


def test_replica_exchange_swap(hip):
    """Metropolis swap of main.cpp:251-273 on the device: with equal temperatures every proposed swap is
    accepted (lboltz_diff = 0 is not < 0) and coordinates are exchanged; with a huge temperature gap favouring
    the current assignment the swap is rejected."""
    name = 'trpcage20_7A'
    c = hip.calc
    g = P.golden(name); n_atom = g['pos'].shape[0]
    eng = c.upside_hip_construct(n_atom, P.fixture(name).encode(), 4, True)
    pos = np.stack([g['pos'], g['pos2'], g['pos'], g['pos2']]).astype('f4')
    c.upside_hip_set_pos(eng, pos.ctypes.data)
    temps = np.array([0.8, 0.8, 0.8, 0.8], 'f4')
    c.upside_hip_init_md(eng, temps.ctypes.data, 5, 5.0, 0.009, 1)
    pairs = np.array([[0, 1], [2, 3]], 'i4'); acc = np.zeros(3, 'i4')
    assert c.upside_hip_replica_swap(eng, 2, pairs.ctypes.data, 99, 3, acc.ctypes.data) == 0
    assert list(acc[:2]) == [1, 1]
    out = np.zeros_like(pos); c.upside_hip_get_pos(eng, out.ctypes.data)
    assert np.array_equal(out[0], pos[1]) and np.array_equal(out[1], pos[0])
    # E(pos) < E(pos2) here; system 0 cold holding the low-energy structure, system 1 hot: swapping is uphill
    c.upside_hip_set_pos(eng, pos.ctypes.data)
    temps = np.array([0.01, 100.0, 0.8, 0.8], 'f4')
    c.upside_hip_init_md(eng, temps.ctypes.data, 5, 5.0, 0.009, 1)
    pairs = np.array([[0, 1]], 'i4')
    assert c.upside_hip_replica_swap(eng, 1, pairs.ctypes.data, 99, 4, acc.ctypes.data) == 0
    e = [float(P.golden(name)['energy']), float(P.golden(name)['energy2'])]
    expect_reject = (1 / 0.01 - 1 / 100.0) * (e[0] - e[1]) < -50
    if expect_reject:
        assert acc[0] == 0
        c.upside_hip_get_pos(eng, out.ctypes.data)
        assert np.array_equal(out[0], pos[0])
    c.free_deriv_engine(ct.c_void_p(eng))


def test_replica_swap_next_reuses_energies(hip):
    """the later swap sets of one attempt: `upside_hip_replica_swap_next` (no new force evaluation, energies of accepted
    pairs traded) gives the verdicts, generator position and coordinates of `upside_hip_replica_swap_from` (which
    re-evaluates, like main.cpp:251-259)"""
    name = 'trpcage20_7A'
    c = hip.calc
    for f in (c.upside_hip_replica_swap_from, c.upside_hip_replica_swap_next):
        f.argtypes = [ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_uint32, ct.c_uint64, ct.c_int, ct.c_void_p]
    g = P.golden(name); n_atom = g['pos'].shape[0]
    rs = np.random.RandomState(4)
    pos = np.stack([g['pos'] + np.float32(0.03 * k) * rs.normal(size=g['pos'].shape).astype('f4') for k in range(6)]).astype('f4')
    temps = np.array([0.7, 0.74, 0.78, 0.82, 0.86, 0.9], 'f4')
    sets = [np.array([[0, 1], [2, 3], [4, 5]], 'i4'), np.array([[1, 2], [3, 4]], 'i4')]
    results = []
    for later in (c.upside_hip_replica_swap_from, c.upside_hip_replica_swap_next):
        eng = c.upside_hip_construct(n_atom, P.fixture(name).encode(), 6, True)
        c.upside_hip_set_pos(eng, pos.ctypes.data)
        c.upside_hip_init_md(eng, temps.ctypes.data, 5, 5.0, 0.009, 1)
        verdicts = []
        for rnd in range(1, 9):
            draw = 0
            for k, pairs in enumerate(sets):
                acc = np.zeros(len(pairs) + 1, 'i4')
                fn = c.upside_hip_replica_swap_from if k == 0 else later
                assert fn(eng, len(pairs), pairs.ctypes.data, 11, rnd, draw, acc.ctypes.data) == 0
                draw = int(acc[-1]); verdicts.append(acc.copy())
        out = np.zeros_like(pos); c.upside_hip_get_pos(eng, out.ctypes.data)
        results.append((np.concatenate(verdicts), out))
        c.free_deriv_engine(ct.c_void_p(eng))
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
    v = results[0][0]
    assert 0 < v[v < 2].sum() < len(v)            # both accepted and rejected swaps occurred (entries > 1 are draw counters)


def test_rccl_exchange_loopback_matches_device_swap(hip):
    """The C++ RCCL exchange (upside_hip_comm_*: device-side energy sum, ncclAllGather, device Metropolis, stream-ordered
    coordinate moves) on a world of ONE rank must reproduce upside_hip_replica_swap_from / _next bit for bit: same verdicts
    over several attempts of two alternating swap sets, same final coordinates.  (Cross-GPU pairs need a multi-GPU node; the
    rank arithmetic they add is covered on CPU in tests/test_replicas_gloo.py.)"""
    name = 'trpcage20_7A'
    c = hip.calc
    for f in (c.upside_hip_replica_swap_from, c.upside_hip_replica_swap_next):
        f.argtypes = [ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_uint32, ct.c_uint64, ct.c_int, ct.c_void_p]
    c.upside_hip_comm_get_unique_id.argtypes = [ct.c_char_p]
    c.upside_hip_comm_init.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_char_p, ct.c_void_p]
    c.upside_hip_comm_replica_swap.argtypes = [ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_uint32, ct.c_uint64, ct.c_int, ct.c_void_p]
    c.upside_hip_comm_free.argtypes = [ct.c_void_p]
    c.upside_hip_run_steps.argtypes = [ct.c_void_p, ct.c_int]
    g = P.golden(name); n_atom = g['pos'].shape[0]
    rs = np.random.RandomState(4)
    pos = np.stack([g['pos'] + np.float32(0.03 * k) * rs.normal(size=g['pos'].shape).astype('f4') for k in range(6)]).astype('f4')
    temps = np.array([0.7, 0.74, 0.78, 0.82, 0.86, 0.9], 'f4')
    sets = [np.array([[0, 1], [2, 3], [4, 5]], 'i4'), np.array([[1, 2], [3, 4]], 'i4')]
    results = []
    for use_comm in (False, True):
        eng = c.upside_hip_construct(n_atom, P.fixture(name).encode(), 6, True)
        c.upside_hip_set_pos(eng, pos.ctypes.data)
        c.upside_hip_init_md(eng, temps.ctypes.data, 5, 5.0, 0.009, 1)
        if use_comm:
            uid = ct.create_string_buffer(128)
            assert c.upside_hip_comm_get_unique_id(uid) == 0, c.upside_hip_last_error()
            assert c.upside_hip_comm_init(eng, 0, 1, uid, temps.ctypes.data) == 0, c.upside_hip_last_error()
        verdicts = []
        for rnd in range(1, 9):
            draw = 0
            for k, pairs in enumerate(sets):
                if use_comm:
                    acc = np.zeros(len(pairs), 'i4')
                    assert c.upside_hip_comm_replica_swap(eng, len(pairs), pairs.ctypes.data, 11, rnd, int(k == 0), acc.ctypes.data) == 0, c.upside_hip_last_error()
                    verdicts.append(acc.copy())
                else:
                    acc = np.zeros(len(pairs) + 1, 'i4')
                    fn = c.upside_hip_replica_swap_from if k == 0 else c.upside_hip_replica_swap_next
                    assert fn(eng, len(pairs), pairs.ctypes.data, 11, rnd, draw, acc.ctypes.data) == 0
                    draw = int(acc[-1]); verdicts.append(acc[:-1].copy())
            c.upside_hip_run_steps(eng, 3)      # a round of MD between attempts, as in a run
        out = np.zeros_like(pos); c.upside_hip_get_pos(eng, out.ctypes.data)
        results.append((np.concatenate(verdicts), out))
        if use_comm:
            c.upside_hip_comm_free(eng)
        c.free_deriv_engine(ct.c_void_p(eng))
    assert np.array_equal(results[0][0], results[1][0]), (results[0][0], results[1][0])
    assert np.array_equal(results[0][1], results[1][1])
    v = results[0][0]
    assert 0 < v.sum() < len(v)            # both accepted and rejected swaps occurred


def test_ensemble_exchange_matches_device_swap(hip):
    """The cross-rank replica-exchange path (host verdicts from gathered energies + per-system coordinate moves,
    upside-md_amd/replicas.py) and the single-engine device kernel must agree: same accepted pairs, same final
    coordinates, over a temperature ladder where some swaps are rejected."""
    name = 'trpcage20_7A'
    g = P.golden(name)
    S = 6
    temps = P.pkg.replicas.geometric_ladder(0.3, 3.0, S)
    beta = (1.0 / temps).astype('f4')
    base = np.stack([g['pos'] if s % 2 == 0 else g['pos2'] for s in range(S)]).astype('f4')
    base += 0.02 * np.random.RandomState(1).randn(*base.shape).astype('f4')
    a = P.pkg.engine.Ensemble(P.fixture(name), S, library=hip)
    b = P.pkg.engine.Ensemble(P.fixture(name), S, library=hip)
    for e in (a, b):
        e.set_pos(base); e.init_md(temps, 5)
    c = hip.calc
    n_acc = n_rej = 0
    for round_num in range(8):
        for pairs in P.pkg.replicas.neighbour_swap_sets(S):
            a.set_pos(base); b.set_pos(base)
            acc, _ = P.pkg.replicas.exchange_swap_set(None, a, pairs, beta, 7, round_num, 0)
            pr = np.asarray(pairs, 'i4'); out = np.zeros(len(pairs) + 1, 'i4')
            assert c.upside_hip_replica_swap(b.engine, len(pairs), pr.ctypes.data, 7, round_num, out.ctypes.data) == 0
            assert list(out[:len(pairs)]) == [int(x) for x in acc]
            assert np.array_equal(a.get_pos(), b.get_pos())
            n_acc += int(acc.sum()); n_rej += int((~acc).sum())
    assert n_acc > 0 and n_rej > 0
    # per-system accessors
    x = a.get_system_pos(2); a.swap_systems(2, 3)
    assert np.array_equal(a.get_system_pos(3), x)
    a.set_system_pos(0, x); assert np.array_equal(a.get_pos()[0], x)
    a.close(); b.close()


@pytest.mark.gpu
def test_ideal_chain_alignment_frames(hip):
    """an ideal chain built at the origin (its first residue lies exactly in a coordinate plane of its reference frame: the all-zero
    Householder vector of affine_alignment's eigensolver, see tests/test_oracle_pinning.py): frames and forces of the REFERENCE"""
    name = 'trpcage20_7A'
    g = dict(np.load(os.path.join(P.GOLD, name + '.ideal_chain.npz')))
    up = P.pkg.Upside(P.fixture(name))
    d = up.deriv(g['pos']); e = up.energy(g['pos']); a = up.get_output('affine_alignment')
    up.close()
    assert np.abs(np.abs((g['affine_alignment'][:, 3:] * a[:, 3:]).sum(axis=1)) - 1.).max() < 1e-5
    assert abs(e - g['energy']) < 1e-4 * max(1., abs(g['energy'])) and P.rel_rms(g['deriv'], d) < 1e-4


ALT_PATHS = [
    {'UPSIDE_HIP_BP_CLUSTER': '1'},          # one-workgroup belief propagation instead of the cluster solve
    {'UPSIDE_HIP_BP_CLUSTER': '3'},          # cluster too small for the pair matrices: on-device fallback flag
    {'UPSIDE_HIP_BP_SPLIT': '3'},            # split cluster solve: 3 workgroups per system over global-memory matrices
    {'UPSIDE_HIP_BP_CLUSTER': '6', 'UPSIDE_HIP_BP_CLUSTER_TEST_ABORT': '1'},   # a cluster workgroup never arrives: barriers give up, the one-workgroup solve re-solves
    {'UPSIDE_HIP_ASYNC_PREPARE': '0'},       # list upkeep inline on the main stream
    {'UPSIDE_HIP_IG_UNSTAGED': '1'},         # coverage graphs through the kernels for systems too large for LDS
    {'UPSIDE_HIP_IG_WGS': '4096'},           # many thin workgroups per pair kernel
    {'UPSIDE_HIP_ROT_UNSTAGED': '1'},        # rotamer pair kernels with bead rows in global memory (large systems)
    {'UPSIDE_HIP_GRAPH': '1'},               # MD loop replayed from a captured hipGraph (the default up to 16 systems)
    {'UPSIDE_HIP_GRAPH': '0'},               # ... and launched step by step
    {'UPSIDE_HIP_ROTAMER_ATOMIC': '1'},      # pair matrices accumulated with atomics (libraries with several beads per state)
    {'UPSIDE_HIP_PLB_UNSTAGED': '1'},        # list build reading the other side from global memory (very large systems)
    {'UPSIDE_HIP_SKIN_SCALE': '1.0'},        # the reference's cached-list margin
    {'UPSIDE_HIP_IG_POLY': '0'},             # coverage pair passes on the spline-coefficient table (tables too large for the polynomial form)
    {'UPSIDE_HIP_ROT_SORT_BEADS': '0'},      # side-chain beads in the configuration's own order (no renumbering by pair-matrix node)
    {'UPSIDE_HIP_ROT_POLY': '0'},            # side-chain energy pass on the spline-coefficient table (tables too large for the polynomial form)
    {'UPSIDE_HIP_BP_CLUSTER': '1', 'UPSIDE_HIP_BP_ENERGY_TABLE': '1'},  # one-workgroup BP taking exp(-E) of the pair matrices itself
    {'UPSIDE_HIP_BP_CLUSTER': '1', 'UPSIDE_HIP_BP_COMPACT': '0'},       # one-workgroup BP of 1024 lanes streaming every pair matrix over the cached inbox layout
    {'UPSIDE_HIP_BP_CLUSTER': '1', 'UPSIDE_HIP_BP_LAYOUT': '1'},        # one-workgroup BP with the inbox layout and the fold as a launch of their own (the choice from 512 systems on)
    {'UPSIDE_HIP_BP_CLUSTER': '1', 'UPSIDE_HIP_BP_LAYOUT': '2'},        # ... whose scratch is too small for the system: every solve lays its inbox out itself after all
    {'UPSIDE_HIP_BP_CLUSTER': '1', 'UPSIDE_HIP_BP_LDS_MSG_KB': '0'},    # one-workgroup BP, every message in global memory
    {'UPSIDE_HIP_BP_CLUSTER': '1', 'UPSIDE_HIP_BP_LDS_MSG_KB': '8'},    # LDS boundary inside the rows to 3-state nodes
    {'UPSIDE_HIP_BP_CLUSTER': '1', 'UPSIDE_HIP_BP_LDS_MSG_KB': '60'},   # LDS boundary inside the rows to 6-state nodes
    {'UPSIDE_HIP_FUSE': '0'},                # every per-element op launched on its own instead of through the fused-op queue
    {'UPSIDE_HIP_BATCH': '0'},               # no merged launches: every upkeep kernel and pair pass as a launch of its own, upkeep on side streams
    {'UPSIDE_HIP_BATCH': '1'},               # merged launches whatever the batch size
    {'UPSIDE_HIP_FUSE_BARRIERS': '1'},       # a workgroup barrier in front of every fused op (no dependency analysis)
    {'UPSIDE_HIP_SCHEDULE': 'bfs'},          # the reference's level-by-level order of the sweep instead of the grouped one
    {'UPSIDE_HIP_FUSE_THREADS': '1024'},     # fused launches with 1024-lane workgroups (the instance that spills the alignment ops)
    {'UPSIDE_HIP_FUSE_THREADS': '128'},      # ... and with two wavefronts per system
    {'UPSIDE_HIP_NODE_PROB_IN_SOLVE': '0'},  # node probabilities by a kernel of their own (large batches) instead of in the solve's prologue
    {'UPSIDE_HIP_NODE_PROB_IN_SOLVE': '0', 'UPSIDE_HIP_BP_CLUSTER': '6'},   # ... in front of the cluster solve
    {'UPSIDE_HIP_BACKBONE_LIST': '0'},       # backbone sterics scanning all residue pairs every step (no cached residue-pair lists)
    {'UPSIDE_HIP_BACKBONE_SKIN': '0.5'},     # ... and with a short margin (rebuilds every other step)
    {'UPSIDE_HIP_SLOT_SPLIT': '1'},          # slot numbering by one workgroup per system (the large-batch choice)
    {'UPSIDE_HIP_SLOT_SPLIT': '3'},          # ... and by three
    {'UPSIDE_HIP_PAIR2': '0'},               # scalar (one partner per lane) forms of the side-chain gradient and coverage passes
]


@pytest.mark.parametrize('name', ['syn150_10A', 'syn300_10A'])
def test_alternate_code_paths_agree(tmp_path, name):
    """every selectable code path gives the golden forces and the same short MD trajectory as the default path -- on the 150-residue fixture
    and on the benchmark protein (whose belief-propagation inbox, pair tables and staged elements sit at the LDS limits the smaller one is
    far from)"""
    import subprocess
    g = P.golden(name)

    def run(env_extra, tag):
        out = str(tmp_path / (tag + '.npz'))
        env = dict(os.environ); env.update(env_extra)
        worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'alt_path_worker.py')
        subprocess.run([sys.executable, worker, name, out], check=True, env=env, timeout=600,
                       cwd=os.path.dirname(os.path.abspath(__file__)))
        return np.load(out)

    base = run({}, 'default')
    for tag in ('pos', 'pos2'):
        assert P.rel_rms(g['deriv' if tag == 'pos' else 'deriv2'], base['force_' + tag]) < P.golden_tol(name, tag, 'deriv')
    assert P.rel_rms(base['force_pos'], base['ens_force'][0]) < 1e-6
    for i, env_extra in enumerate(ALT_PATHS):
        alt = run(env_extra, 'alt%d' % i)
        for k in ('force_pos', 'force_pos2', 'ens_force'):
            assert P.rel_rms(base[k], alt[k]) < 2e-6, (env_extra, k)
        for k in ('energy_pos', 'energy_pos2', 'ens_energy'):
            assert np.allclose(base[k], alt[k], rtol=2e-6, atol=1e-3), (env_extra, k)
        assert P.rel_rms(base['md_pos'], alt['md_pos']) < 1e-5, env_extra


def test_large_batch_solver_on_every_fixture():
    """what a LARGE batch runs is chosen by batch size (one-workgroup belief propagation, list upkeep on one side stream per graph, every
    kernel a launch of its own, no graph replay); forced here for the small parity cases -- every fixture incl. the degenerate
    sequences (empty slot classes), named values, truncated solves -- by re-running those tests in a child pytest: once with the
    pinned-matrix solve over the dense inbox, once with the streaming solve over the cached inbox layout (the two compiled solves)"""
    import subprocess
    for extra in ({}, {'UPSIDE_HIP_BP_COMPACT': '0'}, {'UPSIDE_HIP_BP_LAYOUT': '1'}):
        env = dict(os.environ, UPSIDE_HIP_BP_CLUSTER='1', UPSIDE_HIP_BATCH='0', UPSIDE_HIP_GRAPH='0', **extra)
        out = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
                              '-k', 'force_pass or degenerate or named_values or truncated or golden'], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200).stdout.decode()
        assert ' passed' in out and 'failed' not in out and 'error' not in out.lower(), (extra, out[-3000:])


def test_side_chain_node_limit_is_refused_with_a_message():
    """the device solve serves one system per workgroup and keeps the residue-pair bookkeeping in LDS: more than 1024
    side-chain nodes are refused at construction with a message naming the limit (never a silent wrong answer).  The limit
    is lowered through the environment so that a shipped fixture exceeds it."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import parity_util as P, ctypes as ct\n"
            "lib = P.pkg.default_library(); c = lib.calc\n"
            "c.construct_deriv_engine.restype = ct.c_void_p; c.upside_hip_last_error.restype = ct.c_char_p\n"
            "e = c.construct_deriv_engine(900, P.fixture('syn300_10A').encode(), True)\n"
            "print('ENGINE', bool(e)); print('MSG', c.upside_hip_last_error().decode())\n") % (P.ROOT, os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UPSIDE_HIP_MAX_ROTAMER_NODES='299')
    out = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300).stdout.decode()
    assert 'ENGINE False' in out and '300 side-chain nodes' in out and 'at most 299' in out, out[-2000:]
    env = dict(os.environ, UPSIDE_HIP_MAX_ROTAMER_NODES='300')
    out = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300).stdout.decode()
    assert 'ENGINE True' in out, out[-2000:]


def test_capacity_overflow_fails_loudly():
    """a neighbour list that outgrows its row capacity (or the slot table) is reported by the evaluating call, not
    silently truncated: run with a deliberately tiny capacity in a child process"""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import parity_util as P\n"
            "up = P.pkg.Upside(P.fixture('proteinG56_7A'))\n"
            "try:\n    up.energy(up.initial_pos)\n    print('NO ERROR')\n"
            "except RuntimeError as e:\n    print('RAISED', up.lib.calc.upside_hip_last_error())\n") % (P.ROOT, os.path.join(P.ROOT, 'tests'))
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, UPSIDE_HIP_NBR_CAP='8'), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=300).stdout.decode()
    assert 'RAISED' in out and 'capacity overflow' in out, out


def test_upside_main_stops_cleanly_on_sigint(hip, tmp_path):
    """SIGINT during a run (main.cpp:24-92, 616, 669-674, 742-743): the run stops at the next chunk boundary, the frames
    logged so far are in the file, "Received early termination signal" goes to stderr, the caller's handlers are back
    and -- `in_process_upside` passes --re-raise-signal as py/upside_engine.py does -- Python sees KeyboardInterrupt."""
    import shutil
    import signal
    import subprocess
    cfg = str(tmp_path / 'sig.up')
    shutil.copyfile(P.fixture('proteinG56_7A'), cfg)
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import parity_util as P\n"
            "lib = P.pkg.default_library()\n"
            "try:\n"
            "    lib.in_process_upside(['--duration', '1e7', '--frame-interval', '0.27', '--temperature', '0.8', '--seed', '1', %r])\n"
            "    print('RETURNED', flush=True)\n"
            "except KeyboardInterrupt:\n"
            "    print('INTERRUPTED', flush=True)\n") % (P.ROOT, os.path.join(P.ROOT, 'tests'), cfg)
    child = subprocess.Popen([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    lines = []
    try:
        while True:                       # wait for a few progress lines: the run is inside its MD loop
            ln = child.stdout.readline().decode()
            assert ln, ''.join(lines)
            lines.append(ln)
            if sum('hbonds' in x for x in lines) >= 3:
                break
        child.send_signal(signal.SIGINT)
        rest = child.communicate(timeout=120)[0].decode()
    finally:
        if child.poll() is None:
            child.kill()
    out = ''.join(lines) + rest
    assert 'Received early termination signal' in out, out
    assert 'INTERRUPTED' in out and 'RETURNED' not in out, out
    assert child.returncode == 0
    got, _ = _read_output(cfg)
    n_frame = got['pos'].shape[0]
    assert n_frame >= 3 and got['time'].shape == (n_frame,) and got['potential'].shape[0] == n_frame
    assert np.isfinite(got['pos']).all()


def _read_output(path):
    from upside_md_amd import h5lite
    with h5lite.open_file(path) as f:
        out = f.group('output')
        return {k: out.read(k) for k in out.keys()}, out.get_attr('invocation') if out.has_attr('invocation') else None


def test_upside_main_potential_deriv_agreement_matches_reference(tmp_path):
    """--potential-deriv-agreement (the reference's finite-difference self-check, /root/reference/src/main.cpp:279-315, 368-372, 506-513):
    the executable prints every potential term of the initial structure and the relative RMS deviation of the analytic derivative from
    central differences of the total potential; same terms and (to the noise of an fp32 difference quotient) the same figure as the
    unmodified reference executable on the same file"""
    import re
    import shutil
    import subprocess
    ref_exe = os.path.join(P.ROOT, 'oracle', '_ref', 'upside_7A')
    exe = os.path.join(P.ROOT, 'upside-md_amd', 'csrc', 'upside_hip')
    if not os.path.exists(ref_exe) or not os.path.exists(exe):
        pytest.skip('executables not built')
    a = str(tmp_path / 'ref.up'); b = str(tmp_path / 'hip.up')
    shutil.copyfile(P.fixture('trpcage20_7A'), a); shutil.copyfile(P.fixture('trpcage20_7A'), b)
    args = ['--duration', '0.05', '--frame-interval', '0.05', '--temperature', '0.8', '--seed', '1', '--potential-deriv-agreement']
    out = {}
    for tag, cmd in (('ref', [ref_exe] + args + [a]), ('hip', [exe] + args + [b])):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=dict(os.environ, OMP_NUM_THREADS='1'))
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        txt = r.stdout.decode()
        err = float(re.search(r'overall potential relative error:\s+([0-9.eE+-]+)', txt).group(1))
        block = txt.split('Initial potential:')[1].split('overall potential')[0]
        terms = dict((m.group(1), float(m.group(2))) for m in re.finditer(r'^(\w+):\s+(-?[0-9.]+)\s*$', block, re.M))
        out[tag] = (err, terms)
    assert set(out['ref'][1]) == set(out['hip'][1]) and len(out['ref'][1]) >= 9, (out['ref'][1], out['hip'][1])
    for k, v in out['ref'][1].items():
        assert abs(out['hip'][1][k] - v) <= 2e-3 + 1e-4 * abs(v), (k, v, out['hip'][1][k])
    assert out['hip'][0] < 2e-3 and abs(out['hip'][0] - out['ref'][0]) < 5e-4, (out['hip'][0], out['ref'][0])


def test_upside_main_output_matches_reference(hip, tmp_path):
    """`upside_main` (the CLI entry of the C-ABI) against the unmodified reference executable on the same
    configuration: same /output datasets, shapes and types; identical frame 0 and time axis; kinetic energy of the
    first frames from the same thermostat stream; trajectories agree over the first rounds (fp32 MD diverges after
    that).  Also a two-system replica-exchange run: replica_index and temperature datasets."""
    import shutil
    import subprocess
    ref_exe = os.path.join(P.ROOT, 'oracle', '_ref', 'upside_7A')
    if not os.path.exists(ref_exe):
        pytest.skip('reference executable not built (oracle/_ref)')
    name = 'proteinG56_7A'
    a = str(tmp_path / 'ref.up'); b = str(tmp_path / 'hip.up')
    shutil.copyfile(P.fixture(name), a); shutil.copyfile(P.fixture(name), b)
    args = ['--duration', '1.08', '--frame-interval', '0.27', '--temperature', '0.8', '--seed', '7']   # 40 rounds, frame every 10
    subprocess.run([ref_exe] + args + [a], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                   env=dict(os.environ, OMP_NUM_THREADS='1'))
    hip.in_process_upside(args + [b], verbose=False)
    ref, inv_ref = _read_output(a)
    got, inv = _read_output(b)
    assert inv is not None and '--duration' in (inv.decode() if isinstance(inv, bytes) else str(inv))
    for k in ('pos', 'kinetic', 'potential', 'time', 'temperature'):
        assert k in got, k
        assert got[k].shape == ref[k].shape and got[k].dtype == ref[k].dtype, (k, got[k].shape, ref[k].shape, got[k].dtype, ref[k].dtype)
    n_atom = P.golden(name)['pos'].shape[0]
    assert got['pos'].shape == (4, 1, n_atom, 3) and got['time'].shape == (4,)
    assert np.array_equal(got['time'], ref['time'])
    assert np.allclose(got['temperature'], 0.8)
    assert np.abs(got['pos'][0] - ref['pos'][0]).max() < 2e-5                 # recentred initial structure
    assert abs(got['potential'][0, 0] - ref['potential'][0, 0]) < 1e-4 * max(1., abs(ref['potential'][0, 0]))
    assert abs(got['kinetic'][0, 0] - ref['kinetic'][0, 0]) < 1e-5 * ref['kinetic'][0, 0]   # same Threefry/Box-Muller draws
    # every dataset the reference writes at its default log level exists here with the same shape and type, and frame 0
    # (same structure in both programs) holds the same numbers: rama, hbond, nonbonded_spring_energy, rama_map_potential,
    # nonlinear_coupling, rotamer_free_energy, rotamer_1body_energy0-2, rotamer_bad_solves_cumulative
    assert set(ref) <= set(got), sorted(set(ref) - set(got))
    for k in sorted(ref):
        assert got[k].shape == ref[k].shape and got[k].dtype == ref[k].dtype, (k, got[k].shape, ref[k].shape, got[k].dtype, ref[k].dtype)
        if k in ('pos', 'kinetic', 'potential', 'time', 'temperature'):
            continue
        r0, g0 = np.asarray(ref[k][0], 'f8'), np.asarray(got[k][0], 'f8')
        assert np.abs(r0 - g0).max() < 3e-4 * max(1., np.abs(r0).max()), (k, np.abs(r0 - g0).max())
        assert np.abs(np.asarray(ref[k][1], 'f8') - np.asarray(got[k][1], 'f8')).max() < 2e-2 * max(1., np.abs(r0).max()), k   # 30 steps later
    # 30 and 60 MD steps later: the two fp32 trajectories are still the same trajectory
    assert P.rel_rms(ref['pos'][1], got['pos'][1]) < 1e-4
    assert P.rel_rms(ref['pos'][2], got['pos'][2]) < 1e-3
    assert abs(got['kinetic'][1, 0] - ref['kinetic'][1, 0]) < 2e-3 * ref['kinetic'][1, 0]

    # replica exchange between two temperatures (README.md:189-193 pattern): per-system files, replica_index logged
    files = []
    for tag in ('ref', 'hip'):
        fs = [str(tmp_path / ('%s_%d.up' % (tag, i))) for i in range(2)]
        for f in fs:
            shutil.copyfile(P.fixture(name), f)
        files.append(fs)
    rargs = ['--duration', '0.54', '--frame-interval', '0.135', '--temperature', '0.8,0.85', '--seed', '3',
             '--replica-interval', '0.135', '--swap-set', '0-1']
    subprocess.run([ref_exe] + rargs + files[0], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                   env=dict(os.environ, OMP_NUM_THREADS='2'))
    hip.in_process_upside(rargs + files[1], verbose=False)
    for i in range(2):
        r, _ = _read_output(files[0][i]); g_, _ = _read_output(files[1][i])
        assert g_['replica_index'].shape == r['replica_index'].shape and g_['replica_index'].dtype == r['replica_index'].dtype
        assert np.array_equal(g_['replica_index'], r['replica_index']), (g_['replica_index'].ravel(), r['replica_index'].ravel())
        assert np.allclose(g_['temperature'], r['temperature'])
        assert P.rel_rms(r['pos'][1], g_['pos'][1]) < 1e-3


def test_upside_main_with_restraint_nodes_matches_reference(hip, tmp_path):
    """MD through `upside_main` on the configuration that carries every optional restraint / external-field node,
    against the reference executable: same first frames.  The AFM tip moves with every force evaluation of the
    integrator (bonds.cpp:150-151), which the logged potential of the later frames only matches if the node counts
    the evaluations the way the reference does."""
    import shutil
    import subprocess
    ref_exe = os.path.join(P.ROOT, 'oracle', '_ref', 'upside_7A')
    if not os.path.exists(ref_exe):
        pytest.skip('reference executable not built (oracle/_ref)')
    name = 'proteinG56_restraints'
    a = str(tmp_path / 'ref.up'); b = str(tmp_path / 'hip.up')
    shutil.copyfile(P.fixture(name), a); shutil.copyfile(P.fixture(name), b)
    args = ['--duration', '1.08', '--frame-interval', '0.27', '--temperature', '0.8', '--seed', '11']   # 40 rounds, frame every 10
    # z-dependent potentials refuse full recentring in both programs (main.cpp:548-556)
    assert subprocess.run([ref_exe] + args + [a], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300).returncode != 0
    with pytest.raises(RuntimeError):
        hip.in_process_upside(args + [b], verbose=False)
    args += ['--disable-z-recentering']
    subprocess.run([ref_exe] + args + [a], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                   env=dict(os.environ, OMP_NUM_THREADS='1'))
    hip.in_process_upside(args + [b], verbose=False)
    ref, _ = _read_output(a)
    got, _ = _read_output(b)
    assert got['pos'].shape == ref['pos'].shape and np.array_equal(got['time'], ref['time'])
    assert np.abs(got['pos'][0] - ref['pos'][0]).max() < 2e-5
    assert P.rel_rms(ref['pos'][1], got['pos'][1]) < 1e-4
    assert P.rel_rms(ref['pos'][2], got['pos'][2]) < 1e-3
    for fr in (0, 1, 2):     # the AFM term alone moves by ~1 energy unit per frame on this fixture
        assert abs(got['potential'][fr, 0] - ref['potential'][fr, 0]) < 2e-4 * abs(ref['potential'][fr, 0]), fr


def test_upside_main_annealing_matches_reference(hip, tmp_path):
    """--anneal-factor / --anneal-duration (main.cpp:432-442, 658-660): the thermostat temperature follows the reference's
    schedule (logged per frame) and the trajectory stays the reference's"""
    import shutil
    import subprocess
    ref_exe = os.path.join(P.ROOT, 'oracle', '_ref', 'upside_7A')
    if not os.path.exists(ref_exe):
        pytest.skip('reference executable not built (oracle/_ref)')
    name = 'trpcage20_7A'
    a = str(tmp_path / 'ref.up'); b = str(tmp_path / 'hip.up')
    shutil.copyfile(P.fixture(name), a); shutil.copyfile(P.fixture(name), b)
    args = ['--duration', '1.62', '--frame-interval', '0.27', '--temperature', '0.9', '--seed', '5', '--anneal-factor', '0.25',
            '--anneal-duration', '1.08', '--thermostat-interval', '0.054']      # thermostat every 2 rounds, annealing over the last 40 of 60
    subprocess.run([ref_exe] + args + [a], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                   env=dict(os.environ, OMP_NUM_THREADS='1'))
    hip.in_process_upside(args + [b], verbose=False)
    ref, _ = _read_output(a); got, _ = _read_output(b)
    assert got['temperature'].shape == ref['temperature'].shape
    assert np.abs(got['temperature'] - ref['temperature']).max() < 1e-6, (got['temperature'].ravel(), ref['temperature'].ravel())
    assert ref['temperature'][-1, 0] < 0.5 * ref['temperature'][0, 0]          # the schedule really ran
    assert P.rel_rms(ref['pos'][1], got['pos'][1]) < 1e-4 and P.rel_rms(ref['pos'][3], got['pos'][3]) < 5e-3
    assert abs(got['kinetic'][2, 0] - ref['kinetic'][2, 0]) < 5e-3 * ref['kinetic'][2, 0]


def test_upside_main_set_param_matches_reference(hip, tmp_path):
    """--set-param FILE (main.cpp:384-395,498-499): one 1-D dataset per node name, handed to that node's set_param"""
    import shutil
    import subprocess
    from upside_md_amd import h5lite
    ref_exe = os.path.join(P.ROOT, 'oracle', '_ref', 'upside_7A')
    if not os.path.exists(ref_exe):
        pytest.skip('reference executable not built (oracle/_ref)')
    name = 'trpcage20_7A'
    a = str(tmp_path / 'ref.up'); b = str(tmp_path / 'hip.up'); pf = str(tmp_path / 'param.h5')
    shutil.copyfile(P.fixture(name), a); shutil.copyfile(P.fixture(name), b)
    with h5lite.open_file(pf, 'w') as h:
        h.write('hbond_energy', np.array([-3.5], 'f4'))
    base = ['--duration', '0.27', '--frame-interval', '0.27', '--temperature', '0.8', '--seed', '2']
    pots = {}
    for tag, extra in (('plain', []), ('set', ['--set-param', pf])):
        subprocess.run([ref_exe] + base + extra + [a], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                       env=dict(os.environ, OMP_NUM_THREADS='1'))
        hip.in_process_upside(base + extra + [b], verbose=False)
        pots[tag] = (float(_read_output(a)[0]['potential'][0, 0]), float(_read_output(b)[0]['potential'][0, 0]))
    assert abs(pots['plain'][0] - pots['set'][0]) > 0.1                      # the parameter matters
    for tag in pots:
        assert abs(pots[tag][0] - pots[tag][1]) < 1e-4 * max(1., abs(pots[tag][0])), (tag, pots[tag])


def test_upside_main_pivot_moves_match_reference(hip, tmp_path):
    """Monte-Carlo pivot moves (monte_carlo_sampler.cpp) through `upside_main --monte-carlo-interval`: the same
    proposals (random stream 2), the same Metropolis verdicts and therefore the same `pivot_stats` and the same
    trajectory as the unmodified reference executable on a configuration with /input/pivot_moves."""
    import shutil
    import subprocess
    ref_exe = os.path.join(P.ROOT, 'oracle', '_ref', 'upside_7A')
    if not os.path.exists(ref_exe):
        pytest.skip('reference executable not built (oracle/_ref)')
    name = 'trpcage20_7A'                      # small and hot enough that a good fraction of the pivots is accepted
    a = str(tmp_path / 'ref.up'); b = str(tmp_path / 'hip.up')
    shutil.copyfile(P.fixture(name), a)
    P.pkg.config.add_pivot_moves(a)
    shutil.copyfile(a, b)
    args = ['--duration', '0.54', '--frame-interval', '0.135', '--temperature', '2.5', '--seed', '11',
            '--monte-carlo-interval', '0.027']         # 20 rounds, a pivot attempt every round, a frame every 5
    subprocess.run([ref_exe] + args + [a], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                   env=dict(os.environ, OMP_NUM_THREADS='1'))
    hip.in_process_upside(args + [b], verbose=False)
    ref, _ = _read_output(a)
    got, _ = _read_output(b)
    assert got['pivot_stats'].shape == ref['pivot_stats'].shape and got['pivot_stats'].dtype == ref['pivot_stats'].dtype
    assert np.array_equal(got['pivot_stats'], ref['pivot_stats']), (got['pivot_stats'], ref['pivot_stats'])
    n_try, n_ok = int(ref['pivot_stats'][:, 1].sum()), int(ref['pivot_stats'][:, 0].sum())
    assert n_try == 15 and 0 < n_ok < n_try, ref['pivot_stats']          # both verdicts occurred
    for f in range(1, 4):
        assert P.rel_rms(ref['pos'][f], got['pos'][f]) < 2e-3, f


def test_bench_contract(tmp_path):
    """bench.py prints ONE JSON line with the fields the driver reads (metric, value, steps exactly as asked,
    roofline of the dominant kernel with live HIP-event timing, cpu_baseline object at N=1)."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(P.ROOT, 'bench.py'), '--gpus', '1', '--steps', '7', '--warmup', '4',
                          '--replicas', '3', '--workload', 'syn150_10A'], check=True, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, UPSIDE_BENCH_CPU_BUDGET_S='1')).stdout.decode()
    lines = [ln for ln in out.strip().split('\n') if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['steps'] == 7 and d['warmup'] == 4 and d['n_gpus'] == 1 and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 3 * 7 / (d['ms_per_step'] * 7e-3)) < 1e-6 * d['value']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s'
    # (a fraction is printed only when it is one: a model rate above the HBM peak -- bytes served on chip -- is withheld, never shown as > 1)
    assert r['frac'] is None or (0 < r['frac'] <= 1.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12), r
    assert 'frac_range' not in r and (r.get('frac_model') is None or r['frac_model'] <= 1.0)
    assert 'igraph' in r and r['igraph']['unit'] == 'GB/s' and r['igraph']['valu']['achieved'] > 0 and r['igraph']['valu']['unit'] == 'TFLOP/s'
    if r.get('kernel', '').startswith('bp:'):        # the solve's floor: active pair matrices in, marginals out (below the model's per-sweep re-reads)
        assert 0 < r['min_bytes_per_launch'] < r['algorithmic_bytes_per_launch']
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'host_cores_total', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('reference', 'port') and c['value'] > 0 and c['host_cores_total'] >= c['cores']
    pc = d['parity_check']          # the timed engine itself against the oracle, behind the timed region
    assert pc['ok'] is True and pc['n'] == 3 and pc['max_rel_rms'] <= 1e-5 and pc['deriv_vs_one_system_engine'] <= 1e-6, pc
    ig = r['igraph']          # north_star's criterion as stated (HBM fraction, target 0.5); the arithmetic view beside it
    assert ig['bound'] == 'hbm' and ig['target_frac'] == 0.5 and 'HBM' in ig['target_of'] and (ig['frac'] is None or 0 < ig['frac'] <= 1.0), ig
    assert ig['valu']['bound'] == 'valu' and 0.3 < ig['valu']['hbm_frac_at_fp32_peak'] < 1.0, ig


def test_bench_self_launch_two_ranks_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2 --workload remd64_proteinG56` with NO launcher in the environment: bench.py starts its two ranks
    itself (child torch.distributed.run), each builds its half of the 64-temperature ladder and they exchange through
    upside_hip_comm_* -- on this one-GPU box over tests/plugin/libshmccl.so (UPSIDE_HIP_COMM_LIB) with both ranks on device 0 and
    gloo for the barrier (RCCL cannot put two ranks on one device; a real node sets neither variable).  One JSON line, n_gpus 2,
    strong scaling, exchange attempts inside the timed region."""
    import json
    import subprocess
    shm = os.path.join(P.ROOT, 'tests', 'plugin', 'libshmccl.so')
    env = dict(os.environ, UPSIDE_HIP_COMM_LIB=shm, UPSIDE_HIP_TESTING='1', UPSIDE_BENCH_ONE_DEVICE='1', UPSIDE_BENCH_DIST_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(P.ROOT, 'bench.py'), '--gpus', '2', '--steps', '1200', '--warmup', '30',
                        '--workload', 'remd64_proteinG56', '--no-cpu-baseline'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().split('\n') if ln.strip().startswith('{')]
    assert len(lines) == 1, r.stdout.decode()
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 1200 and d['scaling'] == 'strong' and d['config']['replicas_per_gpu'] == 32
    assert d['config']['exchange_attempts_timed'] >= 2 and d['value'] > 0
    assert d['parity_check']['ok'] is True, d['parity_check']


def test_long_md_is_thermalised_and_stable(hip):
    """3000 MD steps of 8 replicas (graph replay, list rebuilds, side streams all exercised): the kinetic energy
    equilibrates to 1.5 kT per atom (the reference prints this ratio at the end of a run, main.cpp:686-697), the
    potential energy stays bounded, every replica follows its own thermostat stream, and the run is reproducible."""
    name = 'syn150_10A'
    T = 0.8

    def run():
        ens = P.pkg.engine.Ensemble(P.fixture(name), 8, library=hip)
        ens.set_pos(P.golden(name)['pos'])
        ens.init_md(T, 21)
        e0 = ens.energies()
        ratios = []
        for block in range(50):
            ens.run_steps(60)
            if block >= 25:
                m = ens.get_mom()
                ratios.append(0.5 * (m ** 2).sum(axis=(1, 2)) / m.shape[1] / (1.5 * T))
        e1 = ens.energies(); pos = ens.get_pos()
        ens.close()
        return e0, e1, np.array(ratios), pos

    e0, e1, ratios, pos = run()
    assert np.all(np.isfinite(e1)) and np.all(np.abs(e1 - e0) < 0.5 * abs(e0[0]) + 200.)
    assert abs(ratios.mean() - 1.0) < 0.05, ratios.mean()
    assert np.all(np.abs(ratios.mean(axis=0) - 1.0) < 0.12), ratios.mean(axis=0)
    assert len({round(float(x), 3) for x in e1}) == 8          # independent thermostat streams -> distinct trajectories
    e0b, e1b, ratios_b, pos_b = run()
    assert np.array_equal(pos, pos_b) and np.array_equal(e1, e1b)   # bit-reproducible run to run


def test_upside_main_mixed_potentials_match_reference(hip, tmp_path):
    """Configuration files with DIFFERENT /input/potential in one `upside_main` run (the reference builds one engine per file,
    main.cpp:450-571: Hamiltonian replica exchange, mixed runs): here one batched engine per distinct potential.  Two copies
    of the 56-residue protein with and two without the restraint / external-field nodes (same atoms, different Hamiltonians),
    four temperatures, two alternating swap sets whose pairs cross the two potentials, so every verdict needs the second
    energy pass of main.cpp:251-259.  Against the unmodified reference executable on the same four files: replica_index of
    every frame identical, the potentials of frame 0 equal (each file under ITS OWN force field), trajectories still the same
    trajectory after the first exchanges."""
    import shutil
    import subprocess
    ref_exe = os.path.join(P.ROOT, 'oracle', '_ref', 'upside_7A')
    if not os.path.exists(ref_exe):
        pytest.skip('reference executable not built (oracle/_ref)')
    names = ['proteinG56_7A', 'proteinG56_restraints', 'proteinG56_7A', 'proteinG56_restraints']
    rargs = ['--duration', '0.27', '--frame-interval', '0.054', '--temperature', '0.80,0.82,0.84,0.86', '--seed', '3',
             '--replica-interval', '0.055', '--swap-set', '0-1,2-3', '--swap-set', '1-2', '--disable-recentering']
    out = {}
    for tag in ('ref', 'hip'):
        fs = [str(tmp_path / ('%s_%d.up' % (tag, i))) for i in range(4)]
        for f, nm in zip(fs, names):
            shutil.copyfile(P.fixture(nm), f)
        if tag == 'ref':
            subprocess.run([ref_exe] + rargs + fs, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                           env=dict(os.environ, OMP_NUM_THREADS='4'))
        else:
            hip.in_process_upside(rargs + fs, verbose=False)
        out[tag] = [_read_output(f)[0] for f in fs]
    n_frame = out['ref'][0]['replica_index'].shape[0]
    assert n_frame >= 5
    for s_ in range(4):
        r, g_ = out['ref'][s_], out['hip'][s_]
        assert np.array_equal(g_['replica_index'], r['replica_index']), (s_, g_['replica_index'].ravel(), r['replica_index'].ravel())
        assert abs(g_['potential'][0, 0] - r['potential'][0, 0]) < 1e-4 * max(1., abs(r['potential'][0, 0])), s_
        assert abs(g_['kinetic'][0, 0] - r['kinetic'][0, 0]) < 1e-5 * r['kinetic'][0, 0]       # seeds follow the system index of the RUN
        assert P.rel_rms(r['pos'][1], g_['pos'][1]) < 1e-3
        assert set(r.keys()) <= set(g_.keys())                                               # every file logs its own nodes' values
        # the exchange bookkeeping of main.cpp:203-217: partners of this system's swap pairs, running (success, attempt) counts
        assert np.array_equal(g_['replica_swap_partner'], r['replica_swap_partner'])
        assert np.array_equal(g_['replica_cumulative_swaps'], r['replica_cumulative_swaps'])
    # the two force fields really differ on the same structure
    assert abs(out['hip'][0]['potential'][0, 0] - out['hip'][1]['potential'][0, 0]) > 1e-2
    # Hamiltonian replica exchange proper: the same protein under two strengths of the backbone hydrogen-bond energy (the
    # attribute of /input/potential/hbond_energy scaled by 0.9 in two of the four files) -- close enough for accepted swaps
    fs_by_tag = {}
    for tag in ('ref', 'hip'):
        fs = [str(tmp_path / ('h%s_%d.up' % (tag, i))) for i in range(4)]
        for i, f in enumerate(fs):
            shutil.copyfile(P.fixture('proteinG56_7A'), f)
            if i % 2:
                with P.pkg.h5lite.open_file(f, 'r+') as t:
                    g = t.group('input/potential/hbond_energy')
                    g.set_attr('protein_hbond_energy', np.float64(0.9 * float(np.asarray(g.get_attr('protein_hbond_energy')).ravel()[0])))
        fs_by_tag[tag] = fs
    hargs = [a for a in rargs if a != '--disable-recentering']
    subprocess.run([ref_exe] + hargs + fs_by_tag['ref'], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                   env=dict(os.environ, OMP_NUM_THREADS='4'))
    hip.in_process_upside(hargs + fs_by_tag['hip'], verbose=False)
    ref = [_read_output(f)[0] for f in fs_by_tag['ref']]; got = [_read_output(f)[0] for f in fs_by_tag['hip']]
    for s_ in range(4):
        assert np.array_equal(got[s_]['replica_index'], ref[s_]['replica_index']), s_
        assert np.array_equal(got[s_]['replica_cumulative_swaps'], ref[s_]['replica_cumulative_swaps']), s_
        assert abs(got[s_]['potential'][0, 0] - ref[s_]['potential'][0, 0]) < 1e-4 * max(1., abs(ref[s_]['potential'][0, 0]))
        assert P.rel_rms(ref[s_]['pos'][-1], got[s_]['pos'][-1]) < 1e-3
    assert abs(got[0]['potential'][0, 0] - got[1]['potential'][0, 0]) > 1e-3          # two Hamiltonians on one structure
    ri = np.stack([o['replica_index'].reshape(-1) for o in got])
    assert (ri[:, -1] != np.arange(4)).any(), 'no exchange between the two Hamiltonians was accepted'
    # identical potentials in different files still share one engine
    a = str(tmp_path / 'a.up'); c = str(tmp_path / 'c.up')
    shutil.copyfile(P.fixture('proteinG56_7A'), a); shutil.copyfile(P.fixture('proteinG56_7A'), c)
    hip.in_process_upside(['--duration', '0.27', '--frame-interval', '0.27', '--temperature', '0.8', '--seed', '3', a, c], verbose=False)
    # different atom counts cannot exchange coordinates: refused
    d = str(tmp_path / 'd.up'); shutil.copyfile(P.fixture('trpcage20_7A'), d)
    with pytest.raises(RuntimeError):
        hip.in_process_upside(['--duration', '0.27', '--frame-interval', '0.27', '--temperature', '0.8', '--seed', '3', a, d], verbose=False)


def test_replica_swap_next_rejects_stale_energies(hip):
    """a later swap set may only reuse the energies of its own attempt: after MD steps, new coordinates or another round the
    call must fail instead of testing Metropolis on stale numbers"""
    name = 'trpcage20_7A'
    c = hip.calc
    for f in (c.upside_hip_replica_swap_from, c.upside_hip_replica_swap_next):
        f.argtypes = [ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_uint32, ct.c_uint64, ct.c_int, ct.c_void_p]
    c.upside_hip_run_steps.argtypes = [ct.c_void_p, ct.c_int]
    g = P.golden(name); n_atom = g['pos'].shape[0]
    pos = np.stack([g['pos']] * 4).astype('f4'); temps = np.array([0.7, 0.8, 0.9, 1.0], 'f4')
    eng = c.upside_hip_construct(n_atom, P.fixture(name).encode(), 4, True)
    c.upside_hip_set_pos(eng, pos.ctypes.data); c.upside_hip_init_md(eng, temps.ctypes.data, 5, 5.0, 0.009, 1)
    p0 = np.array([[0, 1], [2, 3]], 'i4'); p1 = np.array([[1, 2]], 'i4')
    acc = np.zeros(3, 'i4'); acc1 = np.zeros(2, 'i4')
    assert c.upside_hip_replica_swap_from(eng, 2, p0.ctypes.data, 11, 1, 0, acc.ctypes.data) == 0
    assert c.upside_hip_replica_swap_next(eng, 1, p1.ctypes.data, 11, 1, int(acc[-1]), acc1.ctypes.data) == 0      # same attempt: fine
    assert c.upside_hip_replica_swap_next(eng, 1, p1.ctypes.data, 11, 2, 0, acc1.ctypes.data) != 0                  # another round
    assert c.upside_hip_replica_swap_from(eng, 2, p0.ctypes.data, 11, 2, 0, acc.ctypes.data) == 0
    assert c.upside_hip_run_steps(eng, 3) == 0
    assert c.upside_hip_replica_swap_next(eng, 1, p1.ctypes.data, 11, 2, int(acc[-1]), acc1.ctypes.data) != 0      # MD in between
    assert c.upside_hip_replica_swap_from(eng, 2, p0.ctypes.data, 11, 3, 0, acc.ctypes.data) == 0
    c.upside_hip_set_pos(eng, pos.ctypes.data)
    assert c.upside_hip_replica_swap_next(eng, 1, p1.ctypes.data, 11, 3, int(acc[-1]), acc1.ctypes.data) != 0      # new coordinates
    c.free_deriv_engine(ct.c_void_p(eng))


def test_protein_g_10k_steps_match_reference_statistics(hip, tmp_path):
    """BASELINE.json configs[1] at full length: Protein G (56 residues), constant-T Langevin MD, 10 k force evaluations
    (--duration 90 = 3334 rounds), through `upside_main` on the GPU and through the unmodified reference executable on the
    host, 8 independent copies each (different seeds, one config file per copy as in the reference).  fp32 MD is chaotic, so
    the two programs are compared as samplers of the same ensemble (what the reference itself prints at the end of a run,
    main.cpp:684-697): the mean potential over the second half of the run agrees within 4 combined standard errors (block
    averages over the copies), the kinetic energy equilibrates to 1.5 kT per atom in both, and every copy stays finite."""
    import shutil
    import subprocess
    ref_exe = os.path.join(P.ROOT, 'oracle', '_ref', 'upside_7A')
    if not os.path.exists(ref_exe):
        pytest.skip('reference executable not built (oracle/_ref)')
    name = 'proteinG56_7A'
    n_copy, T = 8, 0.8
    args = ['--duration', '90', '--frame-interval', '0.9', '--temperature', str(T), '--seed', '31']     # a frame every 33 rounds: 101 frames
    files = {}
    for who in ('ref', 'hip'):
        files[who] = [str(tmp_path / ('%s%d.up' % (who, i))) for i in range(n_copy)]
        for f in files[who]:
            shutil.copyfile(P.fixture(name), f)
    subprocess.run([ref_exe] + args + files['ref'], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900,
                   env=dict(os.environ, OMP_NUM_THREADS=str(n_copy)))
    hip.in_process_upside(args + files['hip'], verbose=False)
    n_atom = P.golden(name)['pos'].shape[0]
    stats = {}
    for who in ('ref', 'hip'):
        pot, kin = [], []
        for f in files[who]:
            out, _ = _read_output(f)
            assert out['potential'].shape[0] >= 100 and np.all(np.isfinite(out['pos'])), (who, out['potential'].shape)
            half = out['potential'].shape[0] // 2
            pot.append(np.asarray(out['potential'][half:, 0], 'f8')); kin.append(np.asarray(out['kinetic'][half:, 0], 'f8'))
        pot, kin = np.array(pot), np.array(kin)                 # (copy, frame)
        per_copy = pot.mean(axis=1)
        stats[who] = dict(mean=per_copy.mean(), se=per_copy.std(ddof=1) / np.sqrt(n_copy), kin=kin.mean() / (1.5 * T))
    # kinetic is logged per atom: <p^2>/2 per atom = 1.5 kT in equilibrium (main.cpp:686-697)
    for who in ('ref', 'hip'):
        assert abs(stats[who]['kin'] - 1.0) < 0.03, (who, stats[who])
    d = abs(stats['ref']['mean'] - stats['hip']['mean'])
    se = np.hypot(stats['ref']['se'], stats['hip']['se'])
    assert d < 4. * se + 1e-3 * abs(stats['ref']['mean']), (stats, d, se)


def test_upside_main_jump_moves_match_reference(hip, tmp_path):
    """rigid-body jump moves (monte_carlo_sampler.cpp:157-251), alone and together with pivots: `jump_stats` /
    `pivot_stats` and the trajectory equal the reference executable's.  The "chains" are two tail segments of the
    single-chain fixture moved by a small fraction of an Angstrom, so that some moves are accepted."""
    import shutil
    import subprocess
    ref_exe = os.path.join(P.ROOT, 'oracle', '_ref', 'upside_7A')
    if not os.path.exists(ref_exe):
        pytest.skip('reference executable not built (oracle/_ref)')
    name = 'trpcage20_7A'
    n_atom = P.golden(name)['pos'].shape[0]
    for with_pivot in (False, True):
        a = str(tmp_path / ('ref%d.up' % with_pivot)); b = str(tmp_path / ('hip%d.up' % with_pivot))
        shutil.copyfile(P.fixture(name), a)
        P.pkg.config.add_jump_moves(a, [[0, 9], [n_atom - 12, n_atom]], [0.15, 0.1], [0.05, 0.08])
        if with_pivot:
            P.pkg.config.add_pivot_moves(a)
        shutil.copyfile(a, b)
        args = ['--duration', '0.54', '--frame-interval', '0.135', '--temperature', '2.5', '--seed', '5',
                '--monte-carlo-interval', '0.027']
        subprocess.run([ref_exe] + args + [a], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                       env=dict(os.environ, OMP_NUM_THREADS='1'))
        hip.in_process_upside(args + [b], verbose=False)
        ref, _ = _read_output(a)
        got, _ = _read_output(b)
        keys = ['jump_stats'] + (['pivot_stats'] if with_pivot else [])
        for k in keys:
            assert got[k].shape == ref[k].shape and got[k].dtype == ref[k].dtype, k
            assert np.array_equal(got[k], ref[k]), (k, got[k], ref[k])
        n_try, n_ok = int(ref['jump_stats'][:, 1].sum()), int(ref['jump_stats'][:, 0].sum())
        assert n_try == 15 and 0 < n_ok < n_try, ref['jump_stats']
        for f in range(1, 4):
            assert P.rel_rms(ref['pos'][f], got['pos'][f]) < 2e-3, (with_pivot, f)


def test_external_plugin_node(hip, tmp_path):
    """node types defined OUTSIDE the library (tests/plugin/host_pull.cpp: compiled against include/upside_hip_plugin.h only,
    linked against libupside_hip.so, registered by its static initialisers -- the contract of deriv_engine.h:239-335):
    unknown before upside_hip_load_plugin, constructed from the configuration after it; the host-fallback potential node
    reproduces the built-in device node of the same maths (atom_pos_spring, bonds.cpp:9-50), the host-fallback coordinate
    node feeds a built-in node and back-propagates through it, and MD runs with both in a batch"""
    import shutil
    plug = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'plugin', 'libhost_pull.so')
    assert os.path.exists(plug), 'tests/plugin/libhost_pull.so is missing: run __graft_entry__.build()'
    name = 'proteinG56_7A'
    g = P.golden(name); pos = g['pos']; n_atom = pos.shape[0]
    rs = np.random.RandomState(11)
    atoms = rs.choice(n_atom, 25, replace=False).astype('i4')
    x0 = (pos[atoms] + rs.normal(size=(25, 3))).astype('f4'); k = rs.uniform(0.5, 3., 25).astype('f4')
    scale = np.float32(0.5)

    def config(path, nodes):
        shutil.copyfile(P.fixture(name), path)
        with P.pkg.h5lite.open_file(path, 'r+') as f:
            pot = f.group('input').group('potential')
            for node, arg, idname in nodes:
                grp = pot.create_group(node); grp.set_attr('arguments', [arg])
                if node.startswith('host_scale'):
                    grp.set_attr('scale', np.float32(scale))
                else:
                    grp.write(idname, atoms); grp.write('x0', x0); grp.write('spring_const', k)
        return path
    builtin = config(str(tmp_path / 'builtin.up'), [('atom_pos_spring', 'pos', 'id')])
    plugged = config(str(tmp_path / 'plugged.up'), [('host_pull', 'pos', 'atom')])
    chained = config(str(tmp_path / 'chained.up'), [('host_scale', 'pos', None), ('atom_pos_spring', 'host_scale', 'id')])

    c = hip.calc
    c.upside_hip_load_plugin.argtypes = [ct.c_char_p]
    c.upside_hip_node_type_registered.argtypes = [ct.c_char_p]
    if not c.upside_hip_node_type_registered(b'host_pull'):
        with pytest.raises(RuntimeError):                       # no such node type yet
            P.pkg.Upside(plugged)
        assert c.upside_hip_load_plugin(b'/nonexistent/libnothing.so') == 1 and b'libnothing' in c.upside_hip_last_error()
        assert c.upside_hip_load_plugin(plug.encode()) == 0, c.upside_hip_last_error()
    assert c.upside_hip_load_plugin(plug.encode()) == 0          # loading twice is a no-op

    base = P.pkg.Upside(P.fixture(name)); e_base = base.energy(pos); d_base = base.deriv(pos); base.close()
    ref = P.pkg.Upside(builtin); act = P.pkg.Upside(plugged)
    for x in (pos, g['pos2']):
        e_ref, e_act = ref.energy(x), act.energy(x)
        d_ref, d_act = ref.deriv(x), act.deriv(x)
        assert abs(float(ref.get_output('atom_pos_spring')[0, 0]) - float(act.get_output('host_pull')[0, 0])) < 1e-5 * max(1., abs(e_ref))
        assert abs(e_ref - e_act) < 1e-5 * max(1., abs(e_ref), abs(float(ref.get_output('atom_pos_spring')[0, 0])))
        assert P.rel_rms(d_ref, d_act) < 1e-6
    ref.close(); act.close()
    # and against the closed form, so the pair above is not a shared mistake
    dx = pos[atoms] - x0
    e_pull = float((0.5 * k[:, None] * dx * dx).sum())
    up = P.pkg.Upside(plugged)
    assert abs(up.energy(pos) - (e_base + e_pull)) < 1e-5 * max(1., abs(e_base) + e_pull)
    want = d_base.copy(); want[atoms] += k[:, None] * dx
    assert P.rel_rms(want, up.deriv(pos)) < 1e-5
    up.close()

    # host coordinate node -> built-in potential node: value, chain rule back to pos
    up = P.pkg.Upside(chained)
    e = up.energy(pos); d = up.deriv(pos)
    assert np.abs(up.get_output('host_scale') - scale * pos).max() < 1e-6
    dxs = scale * pos[atoms] - x0
    e_scaled = float((0.5 * k[:, None] * dxs * dxs).sum())
    assert abs(e - (e_base + e_scaled)) < 1e-5 * max(1., abs(e_base) + e_scaled)
    want = d_base.copy(); want[atoms] += scale * k[:, None] * dxs
    assert P.rel_rms(want, d) < 1e-5
    assert np.abs(up.get_sens('host_scale')[atoms] - k[:, None] * dxs).max() < 1e-4
    up.close()

    # a batch under MD: the host nodes synchronise every force pass (no hipGraph replay) and stay in step with the
    # device-only configuration of the same physics
    def run(path):
        eng = c.upside_hip_construct(n_atom, path.encode(), 3, True)
        assert eng, c.upside_hip_last_error()
        x = np.tile(pos[None], (3, 1, 1)).astype('f4'); temps = np.array([0.7, 0.8, 0.9], 'f4')
        assert c.upside_hip_set_pos(eng, x.ctypes.data) == 0
        assert c.upside_hip_init_md(eng, temps.ctypes.data, 7, 5.0, 0.009, 1) == 0
        assert c.upside_hip_run_md(eng, 30) == 0, c.upside_hip_last_error()
        assert c.upside_hip_get_pos(eng, x.ctypes.data) == 0
        c.free_deriv_engine(ct.c_void_p(eng))
        return x
    xa, xb = run(builtin), run(plugged)
    assert np.isfinite(xb).all() and np.abs(xa - xb).max() < 1e-3


def test_upside_main_exchange_through_rccl_equals_in_engine_exchange(hip, tmp_path):
    """`upside_main` one-process-per-GPU mode (RANK / WORLD_SIZE from the launcher; here a world of one, forced with
    UPSIDE_HIP_COMM=1): the replica exchange goes through upside_hip_comm_* (energies all-gathered over RCCL, verdicts on
    the device) instead of the in-engine swap calls.  Same ladder, seeds and swap sets must give the same verdicts and the
    same trajectories as the in-engine path, frame for frame (main.cpp:227-275, 616-672)."""
    import shutil
    name = 'trpcage20_7A'
    runs = {}
    rargs = ['--duration', '2.7', '--frame-interval', '0.27', '--temperature', '0.70,0.74,0.78,0.82', '--seed', '5',
             '--replica-interval', '0.135', '--swap-set', '0-1,2-3', '--swap-set', '1-2']
    for tag, env in (('engine', {}), ('rccl', {'UPSIDE_HIP_COMM': '1'})):
        fs = [str(tmp_path / ('%s_%d.up' % (tag, i))) for i in range(4)]
        for f in fs:
            shutil.copyfile(P.fixture(name), f)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            hip.in_process_upside(rargs + fs, verbose=False)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        runs[tag] = [_read_output(f)[0] for f in fs]
    # the same through the stand-alone executable, launched the way a rank is (RANK / WORLD_SIZE / LOCAL_RANK in the environment)
    exe = os.path.join(P.ROOT, 'upside-md_amd', 'csrc', 'upside_hip')
    assert os.path.exists(exe), 'upside_hip is missing: run __graft_entry__.build()'
    import subprocess
    fs = [str(tmp_path / ('exe_%d.up' % i)) for i in range(4)]
    for f in fs:
        shutil.copyfile(P.fixture(name), f)
    subprocess.run([exe] + rargs + fs, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                   env=dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', UPSIDE_HIP_COMM='1'))
    for a, b in zip(runs['rccl'], [_read_output(f)[0] for f in fs]):
        assert np.array_equal(a['replica_index'], b['replica_index']) and np.array_equal(a['pos'], b['pos'])
    swapped = False
    for a, b in zip(runs['engine'], runs['rccl']):
        assert a['replica_index'].shape == b['replica_index'].shape and a['replica_index'].shape[0] >= 10
        assert np.array_equal(a['replica_index'], b['replica_index'])
        assert np.array_equal(a['pos'], b['pos']) and np.array_equal(a['potential'], b['potential'])
        swapped = swapped or len(np.unique(a['replica_index'])) > 1
    assert swapped, 'no exchange was accepted: the comparison would be vacuous'


def _env_patch(env):
    class _Ctx:
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
        def __exit__(self, *a):
            for k, v in self.old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return _Ctx()


def test_remd64_proteinG56(hip, tmp_path):
    """BASELINE.json configs[3] at its full width on one GPU: 64 temperatures of the 56-residue protein (geometric ladder
    0.5..1.0), two alternating neighbour swap sets, an exchange attempt every second round for ten rounds -- (i) through
    `upside_main` with the in-engine device swap, (ii) through `upside_main` with upside_hip_comm_* (world of one) and (iii)
    through the unmodified reference executable on the same 64 files.  replica_index of every file and frame must be
    IDENTICAL in all three (same Metropolis verdicts from the same counter-based random stream, main.cpp:227-275); the two
    device paths must also agree on every coordinate bit.  (The attempts come early so that the fp32 trajectories of the
    two programs have not separated yet: a verdict only flips if a Boltzmann factor lands within ~1e-5 of its uniform.)"""
    import shutil
    import subprocess
    ref_exe = os.path.join(P.ROOT, 'oracle', '_ref', 'upside_7A')
    if not os.path.exists(ref_exe):
        pytest.skip('reference executable not built (oracle/_ref)')
    name, n = 'proteinG56_7A', 64
    ladder = P.pkg.replicas.geometric_ladder(0.5, 1.0, n)
    sets = P.pkg.replicas.neighbour_swap_sets(n)
    rargs = ['--duration', '0.27', '--frame-interval', '0.054', '--temperature', ','.join('%.6f' % t for t in ladder), '--seed', '11',
             '--replica-interval', '0.055']
    for st in sets:
        rargs += ['--swap-set', ','.join('%d-%d' % (a, b) for a, b in np.asarray(st).reshape(-1, 2))]
    out = {}
    for tag in ('ref', 'engine', 'comm'):
        fs = [str(tmp_path / ('%s_%02d.up' % (tag, i))) for i in range(n)]
        for f in fs:
            shutil.copyfile(P.fixture(name), f)
        if tag == 'ref':
            subprocess.run([ref_exe] + rargs + fs, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900,
                           env=dict(os.environ, OMP_NUM_THREADS=str(min(16, os.cpu_count() or 1))))
        else:
            with _env_patch({'UPSIDE_HIP_COMM': '1'} if tag == 'comm' else {}):
                hip.in_process_upside(rargs + fs, verbose=False)
        out[tag] = [_read_output(f)[0] for f in fs]
    n_frame = out['ref'][0]['replica_index'].shape[0]
    assert n_frame >= 5
    ri = {tag: np.stack([o['replica_index'].reshape(n_frame) for o in out[tag]]) for tag in out}     # [slot][frame]
    assert np.array_equal(ri['engine'], ri['ref']), np.argwhere(ri['engine'] != ri['ref'])[:10]
    assert np.array_equal(ri['comm'], ri['ref'])
    for a, b in zip(out['engine'], out['comm']):
        assert np.array_equal(a['pos'], b['pos']) and np.array_equal(a['potential'], b['potential'])
    # both swap sets accepted something, and every frame is a permutation of the replicas
    assert all(sorted(ri['ref'][:, f]) == list(range(n)) for f in range(n_frame))
    moved = (ri['ref'][:, -1] != np.arange(n)).sum()
    assert moved >= 8, moved
    assert len({abs(int(ri['ref'][s, -1]) - s) for s in range(n)}) >= 2            # some replica travelled through both sets
    for tag in ('engine', 'comm'):
        for s in (0, 31, 63):
            assert P.rel_rms(out['ref'][s]['pos'][-1], out[tag][s]['pos'][-1]) < 1e-3
            assert np.allclose(out[tag][s]['temperature'], ladder[s])


def test_ens512_syn150(hip):
    """BASELINE.json configs[4] on one GPU: 512 independent 150-residue proteins in one engine, every one started from its own
    structure.  Eight of them, drawn at random, must (i) give the forces of a fresh single-system engine on the same structure bit for bit,
    (ii) agree with the CPU oracle within 1e-5 (forces: relative RMS; energy: relative to the sum of |node potentials| as
    everywhere in this file); then 300 MD steps of all 512: everything finite, kinetic energy at 1.5 kT per atom."""
    name, S, T = 'syn150_10A', 512, 0.9       # (the fixture is a frame of the reference's T = 0.9 ensemble: its potential energy is thermal at 0.9)
    c = hip.calc
    g = P.golden(name)
    n_atom = g['pos'].shape[0]
    rng = np.random.RandomState(512)
    # independent starting structures: points on the segment pos..pos2 (two structures of the fixture) plus 0.02 A of noise
    w = rng.uniform(0., 0.15, size=S).astype('f4')
    pos = (g['pos'][None] * (1 - w[:, None, None]) + g['pos2'][None] * w[:, None, None] + rng.normal(0., 0.02, (S, n_atom, 3))).astype('f4')
    pos = np.ascontiguousarray(pos)
    eng = c.upside_hip_construct(n_atom, P.fixture(name).encode(), S, True)
    assert eng
    assert c.upside_hip_set_pos(eng, pos.ctypes.data) == 0
    en = np.zeros(S, 'f4'); der = np.zeros((S, n_atom, 3), 'f4')
    assert c.upside_hip_compute(eng, en.ctypes.data, der.ctypes.data) == 0, c.upside_hip_last_error()
    assert np.isfinite(en).all() and np.isfinite(der).all()
    orc = P.pkg.Upside(P.fixture(name), library=P.oracle_library())
    for s in rng.choice(S, 8, replace=False):
        single = P.pkg.Upside(P.fixture(name), library=hip)
        d1 = single.deriv(pos[s]); e1 = single.energy(pos[s])
        assert np.array_equal(d1, der[s]), (s, P.rel_rms(d1, der[s]))
        single.close()
        ref_e = orc.energy(pos[s]); ref_d = orc.deriv(pos[s])
        assert P.rel_rms(ref_d, der[s]) < RTOL, (s, P.rel_rms(ref_d, der[s]))
        scale = sum(abs(orc.get_output(nm)[0, 0]) for nm in P.POTENTIAL_NODES + ['rotamer'])
        # (the free energy is summed over the lanes of the solve's workgroup, 512 in a large batch and 1024 for one system: the
        #  two totals differ in the rounding of the last additions, the forces in no bit)
        assert abs(e1 - en[s]) < 1e-6 * scale, (s, e1, en[s])
        assert abs(ref_e - en[s]) < RTOL * scale, (s, ref_e, en[s], scale)
    orc.close()
    temps = np.full(S, T, 'f4')
    assert c.upside_hip_init_md(eng, temps.ctypes.data, 4242, 5.0, 0.009, 1) == 0
    assert c.upside_hip_run_md(eng, 100) == 0, c.upside_hip_last_error()          # 100 rounds = 300 force evaluations
    mom = np.zeros((S, n_atom, 3), 'f4'); p2 = np.zeros((S, n_atom, 3), 'f4')
    assert c.upside_hip_get_mom(eng, mom.ctypes.data) == 0 and c.upside_hip_get_pos(eng, p2.ctypes.data) == 0
    assert c.upside_hip_compute(eng, en.ctypes.data, None) == 0
    assert np.isfinite(mom).all() and np.isfinite(p2).all() and np.isfinite(en).all()
    ratio = 0.5 * (mom.astype('f8') ** 2).sum(axis=(1, 2)) / n_atom / (1.5 * T)
    assert abs(ratio.mean() - 1.0) < 0.03, ratio.mean()
    assert len({round(float(x), 2) for x in en}) > S // 2                          # distinct trajectories
    c.free_deriv_engine(ct.c_void_p(eng))


def test_benchmark_engine_2048_systems_matches_oracle(hip):
    """The configuration the headline is measured in -- >= 2048 systems of syn300_10A in ONE engine, which selects the large-batch
    code paths by itself (256-lane fused lists with the 8-waves cap, no merged launches, upkeep on side streams, the one-workgroup
    solve; bench.py's default is 4096 of them) -- against the oracle: every system starts from its own structure, 15 MD steps put the
    cached-list path and a few rebuilds behind them, then six systems drawn at random must agree with the CPU oracle at their
    current positions within 1e-5 (forces: relative RMS; energy: relative to the sum of |node potentials|) and with a fresh
    one-system engine (another solve variant: 1e-6), and every system of the batch must be finite."""
    name, S = 'syn300_10A', 2048
    c = hip.calc
    g = P.golden(name)
    n_atom = g['pos'].shape[0]
    rng = np.random.RandomState(2048)
    pos = np.ascontiguousarray((g['pos'][None] + rng.normal(0., 0.05, (S, n_atom, 3))).astype('f4'))
    eng = c.upside_hip_construct(n_atom, P.fixture(name).encode(), S, True)
    assert eng, c.upside_hip_last_error()
    try:
        assert c.upside_hip_set_pos(eng, pos.ctypes.data) == 0
        temps = np.full(S, 0.8, 'f4')
        assert c.upside_hip_init_md(eng, temps.ctypes.data, 77, 5.0, 0.009, 1) == 0
        assert c.upside_hip_run_md(eng, 5) == 0, c.upside_hip_last_error()
        assert c.upside_hip_get_pos(eng, pos.ctypes.data) == 0
        en = np.zeros(S, 'f4'); der = np.zeros((S, n_atom, 3), 'f4')
        assert c.upside_hip_compute(eng, en.ctypes.data, der.ctypes.data) == 0, c.upside_hip_last_error()
        assert np.isfinite(en).all() and np.isfinite(der).all() and np.isfinite(pos).all()
        orc = P.pkg.Upside(P.fixture(name), library=P.oracle_library())
        single = P.pkg.Upside(P.fixture(name), library=hip)
        for s in rng.choice(S, 6, replace=False):
            ref_e = orc.energy(pos[s]); ref_d = orc.deriv(pos[s])
            scale = sum(abs(orc.get_output(nm)[0, 0]) for nm in P.POTENTIAL_NODES)
            assert P.rel_rms(ref_d, der[s]) < RTOL, (s, P.rel_rms(ref_d, der[s]))
            assert abs(ref_e - en[s]) < RTOL * scale, (s, ref_e, en[s], scale)
            d1 = single.deriv(pos[s])
            assert P.rel_rms(d1, der[s]) < 1e-6, (s, P.rel_rms(d1, der[s]))
        single.close(); orc.close()
    finally:
        c.free_deriv_engine(ct.c_void_p(eng))


def test_two_ranks_exchange_across_the_rank_boundary(hip, tmp_path):
    """The cross-rank half of csrc/comm_rccl.cpp and the multi-rank branch of `upside_main` (main.cpp:227-275, 616-672 for a
    ladder spread over processes), executed for real: TWO processes on this one GPU, each with RANK / WORLD_SIZE in its
    environment before its first GPU call, 2 x 4 replicas, swap sets whose pair 3-4 straddles the rank boundary.  The
    collective library is tests/plugin/libshmccl.so (UPSIDE_HIP_COMM_LIB): the nine RCCL entry points over POSIX shared
    memory, because RCCL itself cannot put two ranks on one device.  Everything else is the product path: plan of local /
    straddling pairs, all-gather of the energies, device Metropolis on every rank, unconditional grouped send / receive of the
    straddling coordinates into the staging rows, k_replica_apply kind 2.  Frame for frame, coordinates, potentials and
    replica_index of the eight files must equal the single-engine run of the same eight replicas BIT FOR BIT."""
    import shutil
    import subprocess
    exe = os.path.join(P.ROOT, 'upside-md_amd', 'csrc', 'upside_hip')
    shm = os.path.join(P.ROOT, 'tests', 'plugin', 'libshmccl.so')
    assert os.path.exists(exe) and os.path.exists(shm), 'run __graft_entry__.build()'
    name, n = 'trpcage20_7A', 8
    temps = ','.join('%.3f' % (0.70 + 0.02 * i) for i in range(n))
    rargs = ['--duration', '2.7', '--frame-interval', '0.27', '--temperature', temps, '--seed', '9',
             '--replica-interval', '0.135', '--swap-set', '0-1,2-3,4-5,6-7', '--swap-set', '1-2,3-4,5-6']
    one = [str(tmp_path / ('one_%d.up' % i)) for i in range(n)]
    two = [str(tmp_path / ('two_%d.up' % i)) for i in range(n)]
    for f in one + two:
        shutil.copyfile(P.fixture(name), f)
    hip.in_process_upside(rargs + one, verbose=False)
    rendezvous = str(tmp_path / 'comm_id')
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK='0', UPSIDE_HIP_COMM_LIB=shm, UPSIDE_HIP_TESTING='1', UPSIDE_HIP_COMM_FILE=rendezvous)
        env.pop('UPSIDE_HIP_COMM', None)
        procs.append(subprocess.Popen([exe] + rargs + two, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env))
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=600)[0].decode())
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), '\n'.join(logs)
    assert not os.path.exists(rendezvous), 'rank 0 leaves no rendezvous file behind'
    a = [_read_output(f)[0] for f in one]; b = [_read_output(f)[0] for f in two]
    for s in range(n):
        assert a[s]['replica_index'].shape[0] >= 10
        assert np.array_equal(a[s]['replica_index'], b[s]['replica_index']), s
        assert np.array_equal(a[s]['pos'], b[s]['pos']), s
        assert np.array_equal(a[s]['potential'], b[s]['potential']) and np.array_equal(a[s]['kinetic'], b[s]['kinetic']), s
    ri = np.stack([x['replica_index'].reshape(-1) for x in b])                       # [slot][frame]
    crossed = set(ri[:4].ravel()) & set(range(4, 8))
    assert crossed, 'no replica crossed the rank boundary: the comparison would not exercise the transfer'
    # a rank whose files hold another potential than rank 0's is refused (the device Metropolis assumes one Hamiltonian):
    # same protein with and without restraint nodes -- same atom count, different /input/potential
    other = [str(tmp_path / ('mix_%d.up' % i)) for i in range(4)]
    for i, f in enumerate(other):
        shutil.copyfile(P.fixture('proteinG56_7A' if i < 2 else 'proteinG56_restraints'), f)
    margs = ['--duration', '0.27', '--frame-interval', '0.27', '--temperature', '0.8,0.82,0.84,0.86', '--seed', '3',
             '--replica-interval', '0.135', '--swap-set', '0-1,2-3', '--swap-set', '1-2', '--disable-recentering']
    procs = [subprocess.Popen([exe] + margs + other, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0', UPSIDE_HIP_COMM_LIB=shm, UPSIDE_HIP_TESTING='1',
                                       UPSIDE_HIP_COMM_FILE=rendezvous + '2')) for r in range(2)]
    outs = []
    try:
        for p in procs:                     # the check is a collective after the rendezvous: BOTH ranks end, neither waits
            outs.append(p.communicate(timeout=300)[0].decode())
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, o in zip(procs, outs):
        assert p.returncode not in (0, None) and p.returncode > 0 and 'ranks 0 and 1 hold different /input/potential' in o, o
    assert not os.path.exists(rendezvous + '2')
    # a record another launch left under the same name is not this launch's: rank 1 alone waits for its own and gives up
    # (here: quickly, through the nonce of a launch that has no rank 0)
    def fnv(text):
        h = 1469598103934665603
        for ch in text.encode():
            h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    import time as _time
    # (a) another launch's nonce; (b) THIS launch line's nonce on a record written long before this attempt started (what a crashed
    # earlier attempt from the same shell leaves behind: same nonce, same file name)
    for record in (b'\0' * 128 + (12345).to_bytes(8, 'little') + int(_time.time()).to_bytes(8, 'little'),
                   b'\0' * 128 + fnv('same-launch-line').to_bytes(8, 'little') + int(_time.time() - 1000).to_bytes(8, 'little')):
        with open(rendezvous + '3', 'wb') as f:
            f.write(record)
        p1 = subprocess.Popen([exe] + margs + other[:2] + other[:2], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(os.environ, RANK='1', WORLD_SIZE='2', LOCAL_RANK='0', UPSIDE_HIP_COMM_LIB=shm, UPSIDE_HIP_TESTING='1',
                                       UPSIDE_HIP_COMM_FILE=rendezvous + '3', UPSIDE_HIP_COMM_WAIT_S='3',
                                       UPSIDE_HIP_COMM_NONCE='same-launch-line'))
        try:
            o = p1.communicate(timeout=300)[0].decode()
        finally:
            if p1.poll() is None:
                p1.kill()
        os.remove(rendezvous + '3')
        assert p1.returncode != 0 and 'no communicator id of this launch' in o, o
