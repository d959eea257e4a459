"""CPU tests of the python3 configuration helpers (upside-md_amd/config.py): the PDB reader that stands in for
py/PDB_to_initial_structure.py (no ProDy / Python 2 here) and the restraint writers."""
import os
import numpy as np
import pytest
import parity_util as P

cfg = P.pkg.config


def _fixture_structure(name='trpcage20_7A'):
    with P.pkg.h5lite.open_file(P.fixture(name)) as h:
        return [x.decode() for x in h.read('input/sequence')], h.read('input/pos', 'f8')[:, :, 0]


def test_pdb_round_trip(tmp_path):
    seq, pos = _fixture_structure()
    f = str(tmp_path / 'a.pdb')
    cfg.write_pdb_backbone(f, seq, pos)
    fasta, x, first = cfg.read_pdb_backbone(f, recenter=False)
    assert list(fasta) == seq and first == []
    assert np.abs(x - pos).max() < 6e-4                      # three decimals in the file
    _, xc, _ = cfg.read_pdb_backbone(f)
    assert np.abs(xc.mean(axis=0)).max() < 1e-9              # recentred (PDB_to_initial_structure.py:158-159)


def test_pdb_chain_breaks_and_chains(tmp_path):
    seq, pos = _fixture_structure()
    # two chains: the second starts far away -> an EXPECTED break, recorded; chain selection by id
    fa, fb = str(tmp_path / 'a.pdb'), str(tmp_path / 'b.pdb')
    cfg.write_pdb_backbone(fa, seq[:10], pos[:30], chain='A')
    cfg.write_pdb_backbone(fb, seq[10:], pos[30:] + 30., chain='B')
    both = str(tmp_path / 'ab.pdb')
    open(both, 'w').write(open(fa).read().replace('END\n', 'TER\n') + open(fb).read())
    fasta, x, first = cfg.read_pdb_backbone(both, recenter=False)
    assert list(fasta) == seq and first == [10]
    fasta_b, xb, _ = cfg.read_pdb_backbone(both, chains=['B'], recenter=False)
    assert list(fasta_b) == seq[10:] and np.abs(xb - (pos[30:] + 30.)).max() < 6e-4
    with pytest.raises(ValueError):
        cfg.read_pdb_backbone(both, chains=['C'])
    # a residue missing inside a chain -> UNEXPECTED break: an error unless explicitly allowed (:166-169)
    gap = str(tmp_path / 'gap.pdb')
    lines = [ln for ln in open(fa) if not (ln.startswith('ATOM') and int(ln[22:26]) == 5)]
    open(gap, 'w').write(''.join(lines))
    with pytest.raises(ValueError):
        cfg.read_pdb_backbone(gap)
    fasta_g, _, _ = cfg.read_pdb_backbone(gap, allow_unexpected_chain_breaks=True)
    assert len(fasta_g) == 9


def test_pdb_residue_conventions(tmp_path):
    seq, pos = _fixture_structure()
    f = str(tmp_path / 'a.pdb')
    cfg.write_pdb_backbone(f, seq, pos)
    text = open(f).read().splitlines(True)
    out = []
    for ln in text:
        if ln.startswith('ATOM') and int(ln[22:26]) == 3:
            out.append('HETATM' + ln[6:17] + 'MSE' + ln[20:])                 # selenomethionine -> MET (:26-28)
            if ln[12:16].strip() == 'CA':                                      # a second alternate location is ignored
                out.append(ln[:16] + 'B' + ln[17:30] + '%8.3f' % 99. + ln[38:])
        elif ln.startswith('ATOM') and int(ln[22:26]) == 7 and ln[12:16].strip() == 'C':
            continue                                                           # incomplete backbone: residue dropped (:133)
        else:
            out.append(ln)
    out.insert(-1, 'HETATM 9000  O   HOH A 900       0.000   0.000   0.000  1.00  0.00           O\n')   # water is skipped
    open(f, 'w').write(''.join(out))
    fasta, x, _ = cfg.read_pdb_backbone(f, recenter=False, allow_unexpected_chain_breaks=True)
    assert len(fasta) == len(seq) - 1 and fasta[2] == 'MET'
    assert np.abs(x[:18] - pos[:18]).max() < 6e-4                              # the alternate CA did not replace the first


def test_cis_proline_is_named_cpr(tmp_path):
    # a proline whose preceding peptide bond is cis (|omega| < 90 degrees) becomes CPR (:88)
    pos = cfg.helix_chain(4)
    seq = ['ALA', 'ALA', 'PRO', 'ALA']
    f = str(tmp_path / 'p.pdb')
    cfg.write_pdb_backbone(f, seq, pos)
    assert list(cfg.read_pdb_backbone(f)[0]) == seq                            # trans in an ideal helix
    # make residue 2's peptide bond cis: rotate everything from its CA on by 180 degrees about the C(1)-N(2) axis
    a, b = pos[5], pos[6]
    u = (b - a) / np.linalg.norm(b - a)
    y = pos.copy()
    for i in range(7, len(y)):
        r = y[i] - b
        y[i] = b + 2. * u * r.dot(u) - r                                       # rotation by pi about u
    cfg.write_pdb_backbone(f, seq, y)
    assert list(cfg.read_pdb_backbone(f)[0]) == ['ALA', 'ALA', 'CPR', 'ALA']


def test_model_selection(tmp_path):
    seq, pos = _fixture_structure()
    f1, f2, f = str(tmp_path / '1.pdb'), str(tmp_path / '2.pdb'), str(tmp_path / 'nmr.pdb')
    cfg.write_pdb_backbone(f1, seq, pos); cfg.write_pdb_backbone(f2, seq, pos + 1.)
    body = lambda p: ''.join(ln for ln in open(p) if ln.startswith('ATOM'))
    open(f, 'w').write('MODEL        1\n' + body(f1) + 'ENDMDL\nMODEL        2\n' + body(f2) + 'ENDMDL\nEND\n')
    x1 = cfg.read_pdb_backbone(f, recenter=False)[1]
    x2 = cfg.read_pdb_backbone(f, model=2, recenter=False)[1]
    assert len(x1) == len(pos) and np.abs(x1 - pos).max() < 6e-4 and np.abs(x2 - pos - 1.).max() < 6e-4


def test_break_chains_on_a_fixture(tmp_path):
    """py/ugly_hack_break_chain.py restated (config.break_chains): bonded terms across a junction go, the rama_coord row at the
    junction loses the dihedral it cannot have, one jump move per chain appears -- and a configuration whose hydrogen-bond
    sites still draw on two chains is refused, as the reference script complains (:134-139)."""
    import shutil
    f = str(tmp_path / 'two_chains.up')
    shutil.copyfile(P.fixture('trpcage20_7A'), f)
    with pytest.raises(ValueError):
        cfg.break_chains(f, chain_first_residue=[10])          # written as ONE chain: residue 10's donor site uses C of residue 9
    # drop the junction residues' sites by hand (what write_config(chain_first_residue=...) does), then cut
    with P.pkg.h5lite.open_file(f, 'r+') as t:
        pot = t.group('input/potential')
        before = dict((nm, pot.group(nm).shape('id')[0]) for nm in ('dist_spring', 'angle_spring', 'dihedral_spring'))
        for side in ('donors', 'acceptors'):
            g = pot.group('infer_H_O').group(side)
            res = g.read('residue', 'i4')
            keep = ~np.isin(res, [9, 10])
            for nm in ('residue', 'bond_length', 'id'):
                arr = g.read(nm); g.delete(nm); g.write(nm, arr[keep])
    removed = cfg.break_chains(f, chain_first_residue=[10])
    assert removed['dist_spring'] == 1 and removed['angle_spring'] == 2 and removed['dihedral_spring'] == 1
    with P.pkg.h5lite.open_file(f) as t:
        inp = t.group('input'); pot = inp.group('potential')
        assert list(inp.group('chain_break').read('chain_first_residue', 'i4')) == [10]
        for nm in before:
            ids = pot.group(nm).read('id', 'i4')
            assert ids.shape[0] == before[nm] - removed[nm]
            assert all(len(set(r // 30)) == 1 for r in ids)                     # atoms 0..29 = chain 0, 30..59 = chain 1
            assert pot.group(nm).shape('equil_dist')[0] == ids.shape[0]
        rama = pot.group('rama_coord').read('id', 'i4')
        assert rama[9, 4] == -1 and rama[10, 0] == -1 and (rama[9, :4] >= 0).all() and (rama[10, 1:] >= 0).all()
        jm = inp.group('jump_moves')
        assert jm.read('atom_range', 'i4').tolist() == [[0, 30], [30, 60]]
        assert np.allclose(jm.read('sigma_trans', 'f4'), 5.) and np.allclose(jm.read('sigma_rot', 'f4'), np.pi / 6.)
    # (the C restatement of the engine loads the cut file and evaluates it; a dihedral id of -1 is the dummy angle of bonds.cpp:219-220)
    orc = P.pkg.Upside(f, library=P.oracle_library())
    e = orc.energy(orc.initial_pos.copy())
    assert np.isfinite(e)
    orc.close()


@pytest.mark.skipif(not os.path.isdir('/root/reference/parameters'), reason='needs the reference parameter directory (build container)')
def test_two_chain_configuration_from_scratch(tmp_path):
    """write_config(chain_first_residue=...) + break_chains(): the path of py/upside_config.py --chain-break-from-file followed
    by py/ugly_hack_break_chain.py --chain-break-from-file.  The junction residues have no hydrogen-bond sites (:1445-1449), the
    chain break is recorded, and the unmodified reference library and the C restatement agree on the resulting potential."""
    PARAM = '/root/reference/parameters'
    fasta = cfg.fasta_from_one_letter(cfg.PROTEIN_G[:24])
    pos = cfg.helix_chain(24)
    pos[36:] += np.array([0., 9., 0.])                                           # second chain (residues 12..23) moved beside the first
    f = str(tmp_path / 'dimer.up')
    cfg.write_config(f, fasta, pos, sidechain_lib=os.path.join(PARAM, 'ff_1', 'sidechain.h5'),
                     environment_lib=os.path.join(PARAM, 'ff_1', 'environment.h5'),
                     rama_ref=cfg.load_rama_reference(os.path.join(PARAM, 'common', 'rama_reference.pkl')),
                     hbond_energy=float(open(os.path.join(PARAM, 'ff_1', 'hbond')).read()), chain_first_residue=[12])
    with P.pkg.h5lite.open_file(f) as t:      # (no group handle outlives the block: the file is reopened for writing below)
        don = t.group('input/potential/infer_H_O').group('donors').read('residue', 'i4')
        acc = t.group('input/potential/infer_H_O').group('acceptors').read('residue', 'i4')
    assert not set(don) & {11, 12} and not set(acc) & {11, 12}
    removed = cfg.break_chains(f)                                               # chain_first_residue from the file
    assert removed['dist_spring'] == 1 and removed['angle_spring'] == 2 and removed['dihedral_spring'] == 1
    with P.pkg.h5lite.open_file(f) as t:
        rama = t.group('input/potential/rama_coord').read('id', 'i4')
    assert rama[11, 4] == -1 and rama[12, 0] == -1 and (rama[11, :4] >= 0).all() and (rama[12, 1:] >= 0).all()
    orc = P.pkg.Upside(f, library=P.oracle_library())
    x = orc.initial_pos.copy()
    e_o, d_o = orc.energy(x), orc.deriv(x)
    orc.close()
    assert np.isfinite(e_o) and np.isfinite(d_o).all()
    ref_lib = os.path.join(P.REF_DIR, 'libupside_7A.so')
    if os.path.exists(ref_lib):
        ref = P.pkg.Upside(f, library=P.pkg.UpsideLibrary(ref_lib))
        e_r, d_r = ref.energy(x), ref.deriv(x)
        ref.close()
        assert abs(e_r - e_o) < 1e-4 * max(1., abs(e_r)) and P.rel_rms(d_r, d_o) < 1e-4
        # no force crosses the junction through a bonded term: moving chain 2 rigidly far away leaves chain 1's bonded energy alone


# ---- Ramachandran library, fixed rotamers, loose hydrogen-bond criteria (py/upside_config.py:567-734, 884-959, 316-321) ----------
LIB_RESTYPE = ['ALA', 'GLY', 'PRO', 'VAL', 'LEU', 'ALL', 'CPR']


def _synthetic_rama_library(path, seed=3, n_grid=24):
    """a library file in the documented layout (config.py: groups coil and sheet, attributes restype / dir, dimer_pot, dimer_weight)
    filled with smooth random maps: every (centre, direction, neighbour) map is normalised like a -log probability"""
    rs = np.random.RandomState(seed)
    phi = np.linspace(-np.pi, np.pi, n_grid, endpoint=False)[:, None]; psi = np.linspace(-np.pi, np.pi, n_grid, endpoint=False)[None, :]
    lib = {}
    with P.pkg.h5lite.open_file(path, 'w') as h:
        for name, restype in (('coil', LIB_RESTYPE), ('sheet', LIB_RESTYPE[:-1])):      # the sheet library has no cis-proline
            n = len(restype)
            pot = np.zeros((n, 2, n, n_grid, n_grid)); wt = rs.uniform(0.5, 5., size=(n, 2, n))
            for idx in np.ndindex(n, 2, n):
                c = rs.normal(size=(2, 2, 2))
                m = sum(c[a, b, 0] * np.cos((a + 1) * phi + b * psi) + c[a, b, 1] * np.sin(a * phi + (b + 1) * psi) for a in range(2) for b in range(2))
                pot[idx] = m + np.log(np.exp(-m).sum())
            g = h.create_group(name)
            g.set_attr('restype', restype); g.set_attr('dir', ['left', 'right'])
            g.write('dimer_pot', pot.astype('f4')); g.write('dimer_weight', wt.astype('f4'))
            lib[name] = (restype, pot.astype('f4').astype('f8'), wt.astype('f4').astype('f8'))
    return lib


def _logsumexp_mix(w, pots):
    """-log sum_k w_k exp(-pot_k), weights normalised -- written independently of config.mixture_potential (straight sums in float128)"""
    w = np.asarray(w, dtype=np.longdouble); w = w / w.sum(axis=0)
    dens = sum(w[k].reshape(w[k].shape + (1,) * (np.asarray(pots[k]).ndim - w[k].ndim)) * np.exp(-np.asarray(pots[k], dtype=np.longdouble)) for k in range(len(pots)))
    return np.asarray(-np.log(dens), dtype='f8')


def test_rama_library_maps(tmp_path):
    f = str(tmp_path / 'rama_lib.h5')
    lib = _synthetic_rama_library(f)
    seq = ['ALA', 'CPR', 'VAL', 'GLY', 'PRO', 'LEU', 'ALA']
    restype, pot, wt = lib['coil']
    r = dict((x, i) for i, x in enumerate(restype))
    nb = lambda s: r['PRO' if s == 'CPR' else s]
    norm = lambda m: m + np.log(np.exp(-m).sum())
    with P.pkg.h5lite.open_file(f) as h:
        maps, w = cfg.rama_maps_and_weights(seq, h.group('coil'))
        maps_p, w_p = cfg.rama_maps_and_weights(seq, h.group('coil'), mode='product')
        maps_s, w_s = cfg.rama_maps_and_weights(seq, h.group('sheet'), allow_cpr=False)
    assert np.allclose(np.exp(-maps).sum(axis=(1, 2)), 1.) and np.allclose(np.exp(-maps_p).sum(axis=(1, 2)), 1.)
    # ends: the one neighbour they have; interior: mixture (or product) of the left- and right-neighbour maps; cis-proline is CPR as the
    # centre (coil) but PRO as a neighbour and in the sheet library
    assert np.allclose(maps[0], norm(pot[r['ALA'], 1, nb('CPR')])) and np.allclose(maps[-1], norm(pot[r['ALA'], 0, r['LEU']]))
    i = 2      # VAL between CPR and GLY
    want = _logsumexp_mix([wt[r['VAL'], 0, r['PRO']], wt[r['VAL'], 1, r['GLY']]], [pot[r['VAL'], 0, r['PRO']], pot[r['VAL'], 1, r['GLY']]])
    assert np.allclose(maps[i], norm(want), atol=1e-10)
    assert np.isclose(w[i], 0.5 * (wt[r['VAL'], 0, r['PRO']] + wt[r['VAL'], 1, r['GLY']]))
    want_p = pot[r['VAL'], 0, r['PRO']] + pot[r['VAL'], 1, r['GLY']] - pot[r['VAL'], 1, r['ALL']]
    assert np.allclose(maps_p[i], norm(want_p), atol=1e-10)
    i = 1      # the cis-proline itself
    want = _logsumexp_mix([wt[r['CPR'], 0, r['ALA']], wt[r['CPR'], 1, r['VAL']]], [pot[r['CPR'], 0, r['ALA']], pot[r['CPR'], 1, r['VAL']]])
    assert np.allclose(maps[i], norm(want), atol=1e-10)
    sp, swt = lib['sheet'][1], lib['sheet'][2]
    want = _logsumexp_mix([swt[r['PRO'], 0, r['ALA']], swt[r['PRO'], 1, r['VAL']]], [sp[r['PRO'], 0, r['ALA']], sp[r['PRO'], 1, r['VAL']]])
    assert np.allclose(maps_s[i], norm(want), atol=1e-10)
    # sheet mixing: the sheet weights scaled by exp(-energy); a large energy returns the coil maps
    mixed = cfg.read_weighted_maps(seq, f, sheet_mixing=0.7)
    assert np.allclose(mixed, _logsumexp_mix([w, w_s * np.exp(-0.7)], [maps, maps_s]), atol=1e-10)
    assert np.allclose(cfg.read_weighted_maps(seq, f, sheet_mixing=60.), maps, atol=1e-9)
    assert np.allclose(cfg.read_weighted_maps(seq, f), maps)
    # the node's potential: optional basin biases, then the mean energy of every map removed
    bias = [(2, 'helix', 1.5), (4, 'sheet', -0.5)]
    rp = cfg.library_rama_potential(seq, f, sheet_mixing=0.7, secstr_bias=bias)
    assert np.allclose((rp * np.exp(-rp)).sum(axis=(1, 2))[[0, 1, 3, 5, 6]], 0., atol=1e-9)      # (a biased map is shifted by ITS mean energy, taken before)
    helix, sheet = cfg.secstr_bias_maps(rp.shape[1], rp.shape[2])
    d = (rp - cfg.library_rama_potential(seq, f, sheet_mixing=0.7))
    assert np.allclose(d[2] - d[2].mean(), 1.5 * (helix - helix.mean()), atol=1e-9) and np.allclose(d[0], 0.) and np.abs(d[4]).max() > 0.1
    assert helix[6, 10] > 0.9 and helix[6, 22] < 0.1 and sheet[6, 22] > 0.9           # phi = -90: psi = -30 is helical, psi = 150 is sheet
    with pytest.raises(ValueError):
        cfg.library_rama_potential(seq, f, secstr_bias=[(1, 'coil', 1.)])
    t = str(tmp_path / 'bias.txt'); open(t, 'w').write('residue secstr energy\n2 helix 1.5\n4 sheet -0.5\n')
    assert cfg.read_secstr_bias(t) == bias


def test_fixed_rotamer_states():
    # library table: restype number, chi1, chi2, state -- three chi1 thirds x two chi2 values for type 1, one state per third for type 2
    deg = np.pi / 180.
    tab = [(1, c1 * deg, c2 * deg, 2 * k + j) for k, c1 in enumerate((60., 180., -60.)) for j, c2 in enumerate((90., -90.))]
    tab += [(2, c1 * deg, 0., k) for k, c1 in enumerate((60., 180., -60.))]
    order = ['GLY', 'LEU', 'VAL']
    fasta = ['GLY', 'LEU', 'VAL', 'LEU', 'LEU', 'PRO']
    rows = [('0', 'GLY', 'A', '1', 'nan', 'nan'), ('1', 'LEU', 'A', '2', '70', '-100'), ('2', 'VAL', 'A', '3', '-50', 'nan'),
            ('3', 'LEU', 'A', '4', 'nan', '10'), ('4', 'LEU', 'A', '5', '170', 'nan')]
    fix = cfg.fixed_rotamer_states(fasta, rows, order, tab)
    # glycine: state 0; valine: one state per chi1 third; rows with an unknown angle that is needed are skipped; leucine (chi1 third 0):
    # the reference takes the argmin of the SIGNED periodic chi2 difference (upside_config.py:930-933) -- for chi2 = -100 the candidates
    # 90 and -90 differ by -170 and +10 degrees and the first one wins; restated as is
    assert fix == {0: 0, 1: 0, 2: 2}
    assert list(cfg.chi1_state(np.array([0., 119.9, 120., -120., -0.1, 180.]) * deg)) == [0, 0, 1, 2, 2, 1]
    with pytest.raises(ValueError):
        cfg.fixed_rotamer_states(fasta, [('1', 'VAL', 'A', '2', '60', '60')], order, tab)


@pytest.mark.skipif(not os.path.isdir('/root/reference/parameters'), reason='needs the reference parameter directory (build container)')
def test_configuration_with_rama_library_fixed_rotamers_and_loose_criteria(tmp_path):
    """write_config with --rama-library / --rama-sheet-mixing-energy / --secstr-bias, --fix-rotamer and --loose-hbond-criteria:
    the unmodified reference library loads the file and agrees with the C restatement; the options leave their marks in it"""
    PARAM = '/root/reference/parameters'
    libf = str(tmp_path / 'rama_lib.h5')
    with P.pkg.h5lite.open_file(libf, 'w') as h:       # a library over the 20 residue types + ALL + CPR, 72 x 72 like the README's
        rs = np.random.RandomState(5)
        restype = sorted(cfg.three_letter_aa.values()) + ['ALL', 'CPR'] if hasattr(cfg, 'three_letter_aa') else None
        restype = restype or (list(cfg.aa_sorted) + ['ALL', 'CPR'])
        phi = np.linspace(-np.pi, np.pi, 72, endpoint=False)[:, None]; psi = np.linspace(-np.pi, np.pi, 72, endpoint=False)[None, :]
        for name, rt in (('coil', restype), ('sheet', restype[:-1])):
            n = len(rt)
            c = rs.normal(size=(n, 2, n, 4)) * 0.7
            pot = (c[..., 0, None, None] * np.cos(phi + psi) + c[..., 1, None, None] * np.sin(phi) + c[..., 2, None, None] * np.cos(2 * psi) + c[..., 3, None, None] * np.sin(phi - psi))
            pot += np.log(np.exp(-pot).sum(axis=(-2, -1), keepdims=True))
            g = h.create_group(name); g.set_attr('restype', rt); g.set_attr('dir', ['left', 'right'])
            g.write('dimer_pot', pot.astype('f4')); g.write('dimer_weight', rs.uniform(1., 4., size=(n, 2, n)).astype('f4'))
    fasta = list(cfg.fasta_from_one_letter(cfg.PROTEIN_G[:20])); fasta[7] = 'CPR' if fasta[7] == 'PRO' else fasta[7]
    pos = cfg.helix_chain(20)
    fixrows = [(str(i), fasta[i] if fasta[i] != 'CPR' else 'PRO', 'A', str(i + 1), '-65', '170') for i in (2, 5, 11)]
    common = dict(sidechain_lib=os.path.join(PARAM, 'ff_1', 'sidechain.h5'), environment_lib=os.path.join(PARAM, 'ff_1', 'environment.h5'),
                  rama_ref=cfg.load_rama_reference(os.path.join(PARAM, 'common', 'rama_reference.pkl')),
                  hbond_energy=float(open(os.path.join(PARAM, 'ff_1', 'hbond')).read()))
    f0, f1 = str(tmp_path / 'plain.up'), str(tmp_path / 'options.up')
    i0 = cfg.write_config(f0, fasta, pos, **common)
    i1 = cfg.write_config(f1, fasta, pos, rama_library=libf, rama_sheet_mixing_energy=1.0, secstr_bias=[(3, 'helix', -1.0)], fix_rotamer=fixrows,
                          loose_hbond_criteria=True, **common)
    assert i1['n_bead'] < i0['n_bead']                                   # fixed residues keep the beads of one state
    with P.pkg.h5lite.open_file(f1) as t:
        g = t.group('input/potential/rama_map_pot')
        rp = g.read('rama_pot', 'f8'); more = g.read('more_sheet_rama_pot', 'f8'); less = g.read('less_sheet_rama_pot', 'f8')
        assert rp.shape == (20, 72, 72) and abs(float(np.ravel(g.get_attr('sheet_eps'))[0]) - 1e-2) < 1e-12
        assert np.allclose(np.delete((rp * np.exp(-rp)).sum(axis=(1, 2)), 3), 0., atol=1e-5) and np.abs(more - less).max() > 1e-6
        fx = t.group('input/potential/placement_fixed_point_vector_only').read('fix_rotamer', 'i4')
        assert list(fx[:, 0]) == [2, 5, 11]
        ids = t.group('input/potential/rotamer').group('pair_interaction').read('id', 'i4')
        n_rot_of = dict()
        for v in ids: n_rot_of.setdefault(int(v) >> 8, int(v >> 4) & 15)
        assert t.group('input/potential/protein_hbond').read('interaction_param', 'f8')[0, 0, 0] == 0.5
        assert [x.decode() for x in t.read('input/sequence')][7] == fasta[7]
    orc = P.pkg.Upside(f1, library=P.oracle_library())
    x = orc.initial_pos.copy()
    e_o, d_o = orc.energy(x), orc.deriv(x)
    orc.close()
    assert np.isfinite(e_o) and np.isfinite(d_o).all()
    ref_lib = os.path.join(P.REF_DIR, 'libupside_7A.so')
    if os.path.exists(ref_lib):
        ref = P.pkg.Upside(f1, library=P.pkg.UpsideLibrary(ref_lib))
        e_r, d_r = ref.energy(x), ref.deriv(x)
        ref.close()
        assert abs(e_r - e_o) < 1e-4 * max(1., abs(e_r)) and P.rel_rms(d_r, d_o) < 1e-4
