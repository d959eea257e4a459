"""python3 configuration writer for the README force field.

Restates the subset of the reference's Python-2/PyTables generator that the README MD recipe uses
(/root/reference/README.md:143-153), writing the same `/input` schema through libhdf5 (h5lite):

    write_dist_spring       /root/reference/py/upside_config.py:480-498
    write_angle_spring      ...:500-512
    write_dihedral_spring   ...:514-525
    write_rotamer_placement ...:885-1006
    write_infer_H_O         ...:187-212
    write_count_hbond       ...:295-378
    write_environment       ...:215-292
    write_rama_map_pot      ...:692-734 (+ reference-state block 1480-1491)
    write_backbone_pair     ...:149-165
    write_rotamer           ...:1009-1035
    write_rama_coord        ...:855-863
    write_affine_alignment  ...:168-184
    write_cavity_radial     ...:37-43   (only used to compact synthetic chains)
    random_initial_config   ...:414-476

`parameters/common/rama.dat` is a missing blob in the reference checkout, so Ramachandran maps are
synthetic: -log(reference-state map) plus a smooth per-residue Fourier perturbation (seeded).
"""
import numpy as np
from . import h5lite

three_letter_aa = dict(
    A='ALA', C='CYS', D='ASP', E='GLU', F='PHE', G='GLY', H='HIS', I='ILE', K='LYS', L='LEU',
    M='MET', N='ASN', P='PRO', Q='GLN', R='ARG', S='SER', T='THR', V='VAL', W='TRP', Y='TYR')
aa_sorted = sorted(three_letter_aa.values())
deg = np.deg2rad(1.)
n_bit_rotamer = 4

TRP_CAGE = "NLYIQWLKDGGPSSGRPPPS"
PROTEIN_G = "MTYKLILNGKTLKGETTTEAVDAATAEKVFKQYANDNGVDGEWTYDDATKTFTVTE"


def fasta_from_one_letter(s):
    return np.array([three_letter_aa[c] for c in s])


def random_fasta(n_res, seed):
    """uniform random sequence over the alphabetical 3-letter codes."""
    idx = np.random.RandomState(seed).randint(0, 20, n_res)
    return np.array([aa_sorted[i] for i in idx])


# ---------------------------------------------------------------------------------------------
# structure generation (upside_config.py:414-476)
def _tab_matrices(phi, theta, bond_length):
    r = np.zeros(phi.shape + (4, 4))
    cp, sp, ct, st, l = np.cos(phi), np.sin(phi), np.cos(theta), np.sin(theta), bond_length
    r[..., 0, 0] = -ct;    r[..., 0, 1] = -st;     r[..., 0, 2] = 0;   r[..., 0, 3] = -l * ct
    r[..., 1, 0] = cp * st; r[..., 1, 1] = -cp * ct; r[..., 1, 2] = -sp; r[..., 1, 3] = l * cp * st
    r[..., 2, 0] = sp * st; r[..., 2, 1] = -sp * ct; r[..., 2, 2] = cp;  r[..., 2, 3] = l * sp * st
    r[..., 3, 3] = 1
    return r


def chain_from_rama(rama):
    """rama (n_res,2) phi,psi in radians -> (3*n_res,3) N,CA,C positions with ideal geometry."""
    n_res = rama.shape[0]
    r3 = np.zeros((n_res, 3))
    r3[:, 0:2] = rama
    r3[:, 2] = np.pi
    angles = np.zeros_like(r3)
    lengths = np.zeros_like(r3)
    angles[:, 0] = 120.0 * deg
    angles[:, 1] = 120.0 * deg
    angles[:, 2] = 109.5 * deg
    lengths[:, 0] = 1.453
    lengths[:, 1] = 1.526
    lengths[:, 2] = 1.300
    t = np.zeros(3 * n_res)
    t[3::3] = r3[:-1, 1]
    t[4::3] = r3[:-1, 2]
    t[5::3] = r3[1:, 0]
    tr = _tab_matrices(t, angles.ravel(), lengths.ravel())
    cur = np.eye(4)
    pos = np.zeros((3 * n_res, 3))
    for i, m in enumerate(tr):
        cur = cur.dot(m)
        pos[i] = cur[:3, 3]
    return pos - pos.mean(axis=0)


def random_chain(n_res, seed):
    rs = np.random.RandomState(seed)
    return chain_from_rama(rs.random_sample((n_res, 2)) * 2 * np.pi - np.pi)


def helix_chain(n_res):
    return chain_from_rama(np.tile(np.array([[-60. * deg, -45. * deg]]), (n_res, 1)))


# ---------------------------------------------------------------------------------------------
def synthetic_rama_maps(n_res, ref_map, seed, amplitude=1.0):
    """(n_res,72,72) potentials: -log(ref) + smooth seeded perturbation, then the reference's
    normalisation (upside_config.py:730)."""
    rs = np.random.RandomState(seed)
    nx = ref_map.shape[0]
    base = -np.log(ref_map)
    phi = np.linspace(-np.pi, np.pi, nx, endpoint=False)[:, None]
    psi = np.linspace(-np.pi, np.pi, nx, endpoint=False)[None, :]
    pots = np.zeros((n_res, nx, nx))
    for i in range(n_res):
        c = rs.normal(size=(3, 3, 2)) * amplitude / 3.
        p = base.copy()
        for a in range(3):
            for b in range(3):
                if a == 0 and b == 0:
                    continue
                p += c[a, b, 0] * np.cos(a * phi + b * psi) + c[a, b, 1] * np.sin(a * phi + b * psi)
        pots[i] = p
    pots += np.log(np.exp(-pots).sum(axis=(-2, -1), keepdims=True))   # exp(-pot) sums to 1 per map
    pots -= (pots * np.exp(-pots)).sum(axis=(-2, -1), keepdims=True)
    return pots


# ---------------------------------------------------------------------------------------------
# Ramachandran library (py/upside_config.py:567-640, 692-734): neighbour-dependent maps of the coil and sheet libraries, mixed.
# Library file (HDF5, the README's rama.dat): groups `coil` and `sheet`, each with
#   attributes restype (n_restype names, among them ALL and -- coil only -- CPR) and dir (left, right),
#   dimer_pot    [n_restype][2][n_restype][n_phi][n_psi]   -log probability of (phi, psi) of the central residue given its neighbour
#   dimer_weight [n_restype][2][n_restype]                 occurrence weights
def mixture_potential(weights, potentials):
    """-log of the weighted mixture of the densities exp(-potentials[k]) (weights normalised over k); upside_config.py:567-581.
    Computed as a log-sum-exp around the smallest term."""
    potentials = np.asarray(potentials, dtype='f8'); weights = np.asarray(weights, dtype='f8')
    assert len(weights) == len(potentials)
    weights = weights / weights.sum(axis=0)
    weights = weights.reshape(weights.shape + (1,) * (potentials.ndim - weights.ndim))
    shifted = potentials - np.log(weights)
    low = shifted.min(axis=0)
    return low - np.log(np.exp(low - shifted).sum(axis=0))


def rama_maps_and_weights(seq, group, mode='mixture', allow_cpr=True):
    """per-residue maps (n_res, n_phi, n_psi) and weights (n_res,) of one library group (an h5lite group): every residue's map
    given its left and right neighbours, combined by `mode` (upside_config.py:584-627); maps normalised so that exp(-map) sums to 1"""
    if mode not in ('mixture', 'product'):
        raise ValueError('combining rule must be mixture or product')
    if len(seq) < 3:
        raise ValueError('the Ramachandran library needs at least three residues')
    as_str = lambda x: x.decode() if isinstance(x, bytes) else str(x)
    restype = [as_str(x) for x in np.atleast_1d(group.get_attr('restype'))]
    dirs = [as_str(x) for x in np.atleast_1d(group.get_attr('dir'))]
    r_of = dict((x, i) for i, x in enumerate(restype)); d_of = dict((x, i) for i, x in enumerate(dirs))
    pot = group.read('dimer_pot', 'f8'); wt = group.read('dimer_weight', 'f8')
    # cis-proline is its own type only as the CENTRAL residue (and only where the library has it); as a neighbour it is PRO
    centre = lambda r: r_of[r if (r != 'CPR' or allow_cpr) else 'PRO']
    nbr = lambda r: r_of['PRO' if r == 'CPR' else r]
    V = lambda c, d, n: pot[centre(c), d_of[d], nbr(n)]
    W = lambda c, d, n: wt[centre(c), d_of[d], nbr(n)]
    n = len(seq)
    maps = np.zeros((n,) + pot.shape[-2:]); weights = np.zeros(n)
    maps[0] = V(seq[0], 'right', seq[1]); weights[0] = W(seq[0], 'right', seq[1])
    for i in range(1, n - 1):
        l, c, r = seq[i - 1], seq[i], seq[i + 1]
        if mode == 'product':
            maps[i] = V(c, 'left', l) + V(c, 'right', r) - V(c, 'right', 'ALL')
        else:
            maps[i] = mixture_potential([W(c, 'left', l), W(c, 'right', r)], [V(c, 'left', l), V(c, 'right', r)])
        weights[i] = 0.5 * (W(c, 'left', l) + W(c, 'right', r))
    maps[-1] = V(seq[-1], 'left', seq[-2]); weights[-1] = W(seq[-1], 'left', seq[-2])
    maps += np.log(np.exp(-maps).sum(axis=(-2, -1), keepdims=True))
    return maps, weights


def read_weighted_maps(seq, library_path, sheet_mixing=None, mode='mixture'):
    """the coil maps of `seq`, or -- with a sheet mixing energy -- their mixture with the sheet maps, the sheet weights scaled by
    exp(-sheet_mixing) (upside_config.py:630-639)"""
    with h5lite.open_file(library_path) as lib:
        coil, coil_w = rama_maps_and_weights(seq, lib.group('coil'), mode=mode)
        if sheet_mixing is None:
            return coil
        sheet, sheet_w = rama_maps_and_weights(seq, lib.group('sheet'), allow_cpr=False)
    return mixture_potential([coil_w, sheet_w * np.exp(-sheet_mixing)], [coil, sheet])


def secstr_bias_maps(n_phi, n_psi):
    """smooth indicator maps of the helical and the sheet basin on the library's grid (upside_config.py:706-713)"""
    phi = np.linspace(-np.pi, np.pi, n_phi, endpoint=False)[:, None]
    psi = np.linspace(-np.pi, np.pi, n_psi, endpoint=False)[None, :]
    below = lambda a, b: 1. / (1. + np.exp(-(b - a) / (10. * deg)))          # smooth (a < b)
    helix = below(phi, 0.) * below(-100. * deg, psi) * below(psi, 50. * deg)
    sheet = below(phi, 0.) * (below(psi, -100. * deg) + below(50. * deg, psi))
    return helix, sheet


def library_rama_potential(seq, library_path, sheet_mixing=None, secstr_bias=(), mode='mixture'):
    """rama_pot of the rama_map_pot node from a Ramachandran library (upside_config.py:692-734): the weighted maps, optional
    per-residue basin biases [(residue, 'helix' | 'sheet', energy), ...], then each map's mean energy removed"""
    maps = read_weighted_maps(seq, library_path, sheet_mixing, mode)
    if len(secstr_bias):
        helix, sheet = secstr_bias_maps(maps.shape[1], maps.shape[2])
        for residue, kind, energy in secstr_bias:
            if kind not in ('helix', 'sheet'):
                raise ValueError('secstr in a secstr-bias table must be helix or sheet')
            maps[int(residue)] += float(energy) * (helix if kind == 'helix' else sheet)
    maps -= (maps * np.exp(-maps)).sum(axis=(-2, -1), keepdims=True)
    return maps


def read_secstr_bias(path):
    """table with the header `residue secstr energy` (upside_config.py:715-725)"""
    rows = [ln.split() for ln in open(path) if ln.strip()]
    if rows[0] != 'residue secstr energy'.split():
        raise ValueError('first line of a secstr-bias table must be "residue secstr energy"')
    return [(int(r), k, float(e)) for r, k, e in rows[1:]]


# ---------------------------------------------------------------------------------------------
# fixed rotamers (py/upside_config.py --fix-rotamer, :884-935)
def chi1_state(angles):
    """0: [0, 120) degrees, 2: [-120, 0), 1: the rest (upside_config.py:885-889)"""
    angles = np.asarray(angles, dtype='f8')
    st = np.ones(angles.shape, dtype='i4')
    st[(0. <= angles) & (angles < 120. * deg)] = 0
    st[(-120. * deg <= angles) & (angles < 0.)] = 2
    return st


def fixed_rotamer_states(fasta, table, restype_order, restype_chi_state):
    """residue -> rotamer state for the rows of a --fix-rotamer table [(residue, restype, chain, resnum, chi1, chi2), ...] (angles in
    degrees, NaN = unknown): the library state of the residue type whose chi1 falls in the same third of the circle and whose chi2 is
    closest (periodically); residue types with one state (GLY, ALA) take state 0; rows with an unknown angle that is needed are skipped.
    restype_chi_state: the library's restype_and_chi_and_state table (restype number, chi1, chi2, state)."""
    tab = np.asarray(restype_chi_state, dtype='f8')
    lib_restype = tab[:, 0].astype('i4'); lib_chi1_state = chi1_state(tab[:, 1]); lib_chi2 = tab[:, 2]; lib_state = tab[:, 3].astype('i4')
    num = dict((aa, i) for i, aa in enumerate(restype_order))
    fix = {}
    for residue, restype, chain, resnum, chi1, chi2 in table:
        residue = int(residue)
        if fasta[residue] != (restype if restype != 'CPR' else 'PRO'):
            raise ValueError('fix-rotamer table does not match the sequence: residue %i is %s, the table says %s' % (residue, fasta[residue], restype))
        chi1 = float(chi1) * deg; chi2 = float(chi2) * deg
        if restype in ('GLY', 'ALA'):
            fix[residue] = 0
            continue
        if np.isnan(chi1):
            continue
        ok = (lib_restype == num[fasta[residue]]) & (lib_chi1_state == chi1_state(np.array([chi1]))[0])
        states, chi2s = lib_state[ok], lib_chi2[ok]
        if len(states) == 1:
            fix[residue] = int(states[0])
            continue
        if np.isnan(chi2):
            continue
        d = (chi2s - chi2) % (2. * np.pi)
        d[d > np.pi] -= 2. * np.pi
        fix[residue] = int(states[np.argmin(d)])       # (sic: the signed difference, as the reference selects it)
    return fix


def read_fix_rotamer(path):
    rows = [ln.split() for ln in open(path) if ln.strip()]
    if [x.lower() for x in rows[0]] != 'residue restype chain resnum chi1 chi2'.split():
        raise ValueError('first line of a fix-rotamer table must be "residue restype chain resnum chi1 chi2"')
    return rows[1:]


def load_rama_reference(path):
    import pickle
    with open(path, 'rb') as f:
        return np.asarray(pickle.load(f, encoding='latin1'), dtype='f8')


# ---------------------------------------------------------------------------------------------
def _args(g, names):
    g.set_attr('arguments', list(names))


def write_config(path, fasta, init_pos, sidechain_lib, environment_lib, rama_ref, hbond_energy,
                 rama_seed=0, cavity_radius=0., rotamer_damping=0.4, bond_stiffness=48.,
                 angle_stiffness=175., per_residue_rama=True, chain_first_residue=(), hbond_exclude_residues=(),
                 rama_library=None, rama_sheet_mixing_energy=None, secstr_bias=(), rama_combining_rule='mixture',
                 fix_rotamer=(), loose_hbond_criteria=False):
    """fasta: array of 3-letter codes; init_pos (3*n_res,3); sidechain_lib / environment_lib: paths to
    the parameter HDF5 libraries; rama_ref: (72,72) reference-state probabilities.
    chain_first_residue: first residue of every chain but the first (py/upside_config.py --chain-break-from-file,
    :1413-1451): recorded as /input/chain_break/chain_first_residue, and the two residues at every junction join
    hbond_exclude_residues (no donor / acceptor site inferred from atoms of two chains).  The bonded terms across the
    junctions are removed afterwards by break_chains(), as the reference does with py/ugly_hack_break_chain.py.
    rama_library (+ rama_sheet_mixing_energy, secstr_bias, rama_combining_rule): the Ramachandran maps from a library file
    (py/upside_config.py --rama-library ..., :692-734) instead of the synthetic maps; fasta may then name cis-prolines CPR.
    fix_rotamer: rows of a --fix-rotamer table (read_fix_rotamer): those residues keep ONE rotamer state (:905-959).
    loose_hbond_criteria: the permissive hydrogen-bond geometry of --loose-hbond-criteria (:317-321; not for simulation)."""
    fasta_cpr = np.asarray(fasta)                                             # cis-prolines named CPR (Ramachandran library, dihedral springs)
    fasta = np.array(['PRO' if s == 'CPR' else s for s in fasta_cpr])         # ... and PRO everywhere else (upside_config.py:1395-1405)
    n_res = len(fasta)
    n_atom = 3 * n_res
    assert init_pos.shape == (n_atom, 3)

    with h5lite.open_file(sidechain_lib) as lib:
        restype_order = [x.decode() for x in lib.read('restype_order')]
        bead_order = [x.decode() for x in lib.read('bead_order')]
        rotamer_center_fixed = lib.read('rotamer_center_fixed', 'f8')
        rotamer_prob = lib.read('rotamer_prob', 'f8')
        start_stop = lib.read('rotamer_start_stop_bead', 'i8')
        restype_chi_state = lib.read('restype_and_chi_and_state', 'f8') if len(fix_rotamer) else None
        pair_interaction = lib.read('pair_interaction', 'f8')
        coverage_interaction = lib.read('coverage_interaction', 'f8')
        hydrophobe_placement = lib.read('hydrophobe_placement', 'f8')
        hydrophobe_interaction = lib.read('hydrophobe_interaction', 'f8')
    with h5lite.open_file(environment_lib) as lib:
        env_energies = lib.read('energies', 'f8')
        env_offset = float(lib.get_attr('offset', 'energies'))
        env_inv_dx = float(lib.get_attr('inv_dx', 'energies'))
        env_restype = dict((x.decode(), i) for i, x in enumerate(lib.read('restype_order')))
        env_coverage_param = lib.read('coverage_param', 'f8')
    restype_num = dict((aa, i) for i, aa in enumerate(restype_order))
    bead_num = dict((k, i) for i, k in enumerate(bead_order))

    f = h5lite.open_file(path, 'w')
    inp = f.create_group('input')
    inp.write('sequence', fasta_cpr)                     # (cis-prolines keep their CPR name here, upside_config.py:1374)
    inp.write('pos', init_pos.reshape(n_atom, 3, 1).astype('f4'))
    pot = inp.create_group('potential')

    # --- bonded springs -------------------------------------------------------------------
    g = pot.create_group('dist_spring'); _args(g, ['pos'])
    ids = np.arange(n_atom - 1)
    ids = np.column_stack((ids, ids + 1))
    eq = np.zeros(ids.shape[0]); eq[0::3] = 1.453; eq[1::3] = 1.526; eq[2::3] = 1.300
    g.write('id', ids.astype('i4')); g.write('equil_dist', eq)
    g.write('spring_const', bond_stiffness * np.ones(ids.shape[0]))
    g.write('bonded_atoms', np.ones(ids.shape[0], dtype='i4'))

    g = pot.create_group('angle_spring'); _args(g, ['pos'])
    ids = np.arange(n_atom - 2)
    ids = np.column_stack((ids, ids + 2, ids + 1))
    eq = np.zeros(ids.shape[0])
    eq[0::3] = np.cos(109.5 * deg); eq[1::3] = np.cos(120.0 * deg); eq[2::3] = np.cos(120.0 * deg)
    g.write('id', ids.astype('i4')); g.write('equil_dist', eq)
    g.write('spring_const', angle_stiffness * np.ones(ids.shape[0]))

    g = pot.create_group('dihedral_spring'); _args(g, ['pos'])
    ids = np.arange(1, n_atom - 3, 3)
    ids = np.column_stack((ids, ids + 1, ids + 2, ids + 3))
    g.write('id', ids.astype('i4'))
    g.write('equil_dist', np.where(fasta_cpr[1:] == 'CPR', 0. * deg, 180. * deg))
    g.write('spring_const', 30.0 * np.ones(ids.shape[0]))

    # --- rotamer placement (fixed placement, dynamic 1-body) ---------------------------------
    placement_pos = rotamer_center_fixed
    placement_energy = -np.log(rotamer_prob.transpose((2, 0, 1)))[..., None]
    rama_residue, affine_residue, layer_index, beadtype_seq, id_seq = [], [], [], [], []
    count_by_n_rot = dict()
    fix = fixed_rotamer_states(fasta, fix_rotamer, restype_order, restype_chi_state) if len(fix_rotamer) else {}
    for rnum, aa in enumerate(fasta):
        start, stop, n_bead = [int(x) for x in start_stop[restype_num[aa]]]
        assert (stop - start) % n_bead == 0
        n_rot = (stop - start) // n_bead
        if rnum in fix:                                   # a fixed residue keeps the beads of one state: a 1-state node
            if not 0 <= fix[rnum] < n_rot:
                raise ValueError('invalid fix rotamer state')
            start, stop = start + n_bead * fix[rnum], start + n_bead * (fix[rnum] + 1)
            n_rot = 1
        base_id = (count_by_n_rot.get(n_rot, 0) << n_bit_rotamer) + n_rot
        count_by_n_rot[n_rot] = count_by_n_rot.get(n_rot, 0) + 1
        rama_residue.extend([rnum] * (stop - start))
        affine_residue.extend([rnum] * (stop - start))
        layer_index.extend(range(start, stop))
        beadtype_seq.extend(['%s_%i' % (aa, i) for i in range(n_bead)] * n_rot)
        id_seq.extend(np.arange(stop - start) // n_bead + (base_id << n_bit_rotamer))
    affine_residue = np.array(affine_residue, dtype='i4')
    n_sc = len(affine_residue)

    sc_node_name, pl_node_name = 'placement_fixed_point_vector_only', 'placement_scalar'
    g = pot.create_group(sc_node_name); _args(g, ['affine_alignment'])
    g.write('rama_residue', np.array(rama_residue, dtype='i4'))
    g.write('affine_residue', affine_residue)
    g.write('layer_index', np.array(layer_index, dtype='i4'))
    g.write('placement_data', placement_pos[..., :6])
    g.write('beadtype_seq', np.array(beadtype_seq))
    g.write('id_seq', np.array(id_seq, dtype='i4'))
    g.write('fix_rotamer', np.array(sorted(fix.items()), dtype='i4').reshape(-1, 2))

    g = pot.create_group(pl_node_name); _args(g, ['affine_alignment', 'rama_coord'])
    g.write('rama_residue', np.array(rama_residue, dtype='i4'))
    g.write('affine_residue', affine_residue)
    g.write('layer_index', np.array(layer_index, dtype='i4'))
    g.write('placement_data', placement_energy.astype('f4'))

    # --- hydrogen bonds -------------------------------------------------------------------
    chain_first_residue = [int(i) for i in chain_first_residue]
    if any(i <= 0 or i >= n_res for i in chain_first_residue) or sorted(set(chain_first_residue)) != chain_first_residue:
        raise ValueError('chain_first_residue must be increasing residue indices in (0, n_res)')
    excluded = set(int(i) for i in hbond_exclude_residues) | set(i + j for i in chain_first_residue for j in (-1, 0))   # :1445-1449
    if chain_first_residue:
        inp.create_group('chain_break').write('chain_first_residue', np.array(chain_first_residue, dtype='i4'))
    donor_res = np.array([i for i in range(n_res) if i > 0 and i not in excluded and fasta[i] != 'PRO'], dtype='i4')   # :190-191
    acc_res = np.array([i for i in range(n_res) if i < n_res - 1 and i not in excluded], dtype='i4')
    n_donor, n_acceptor = len(donor_res), len(acc_res)
    g = pot.create_group('infer_H_O'); _args(g, ['pos'])
    don = g.create_group('donors'); acc = g.create_group('acceptors')
    don.write('residue', donor_res); acc.write('residue', acc_res)
    don.write('bond_length', 0.88 * np.ones(n_donor)); acc.write('bond_length', 1.24 * np.ones(n_acceptor))
    don.write('id', (np.array((-1, 0, 1))[None, :] + 3 * donor_res[:, None]).astype('i4'))
    acc.write('id', (np.array((1, 2, 3))[None, :] + 3 * acc_res[:, None]).astype('i4'))

    g = pot.create_group('protein_hbond'); _args(g, ['infer_H_O'])
    g.write('index1', np.arange(n_donor, dtype='i4')); g.write('type1', np.zeros(n_donor, dtype='i4'))
    g.write('id1', donor_res)
    g.write('index2', np.arange(n_donor, n_donor + n_acceptor, dtype='i4'))
    g.write('type2', np.zeros(n_acceptor, dtype='i4')); g.write('id2', acc_res)
    # inner barrier, inner scale, outer barrier, outer scale, wall_dp, inv_dp_width (upside_config.py:316-321)
    g.write('interaction_param', np.array([[[0.5 if loose_hbond_criteria else 1.4, 1. / 0.10, 3.1 if loose_hbond_criteria else 2.5, 1. / 0.125,
                                             0.182 if loose_hbond_criteria else 0.682, 1. / 0.05, 0., 0.]]]))

    g = pot.create_group('hbond_coverage'); _args(g, ['protein_hbond', sc_node_name])
    g.write('interaction_param', coverage_interaction)
    g.write('index1', np.arange(n_donor + n_acceptor, dtype='i4'))
    g.write('type1', (1 * (np.arange(n_donor + n_acceptor) >= n_donor)).astype('i4'))
    g.write('id1', np.concatenate([donor_res, acc_res]))
    g.write('index2', np.arange(n_sc, dtype='i4'))
    g.write('type2', np.array([bead_num[s] for s in beadtype_seq], dtype='i4'))
    g.write('id2', affine_residue)

    g = pot.create_group('placement_fixed_point_vector_scalar'); _args(g, ['affine_alignment'])
    g.write('affine_residue', (np.arange(3 * n_res) // 3).astype('i4'))
    g.write('layer_index', (np.arange(3 * n_res) % 3).astype('i4'))
    g.write('placement_data', hydrophobe_placement)

    g = pot.create_group('hbond_coverage_hydrophobe')
    _args(g, ['placement_fixed_point_vector_scalar', sc_node_name])
    g.write('interaction_param', hydrophobe_interaction)
    g.write('index1', np.arange(3 * n_res, dtype='i4'))
    g.write('type1', (np.arange(3 * n_res) % 3).astype('i4'))
    g.write('id1', (np.arange(3 * n_res) // 3).astype('i4'))
    g.write('index2', np.arange(n_sc, dtype='i4'))
    g.write('type2', np.array([bead_num[s] for s in beadtype_seq], dtype='i4'))
    g.write('id2', affine_residue)

    g = pot.create_group('hbond_energy'); _args(g, ['protein_hbond'])
    g.set_attr('protein_hbond_energy', float(hbond_energy))

    # --- environment ----------------------------------------------------------------------
    g = pot.create_group('placement_fixed_point_vector_only_CB'); _args(g, ['affine_alignment'])
    ref_pos = np.zeros((4, 3))
    ref_pos[0] = (-1.19280531, -0.83127186, 0.)
    ref_pos[1] = (0., 0., 0.)
    ref_pos[2] = (1.25222632, -0.87268266, 0.)
    ref_pos[3] = (0., 0.94375626, 1.2068012)
    ref_pos -= ref_pos.mean(axis=0, keepdims=1)   # sic: the reference centres on all four atoms here
    pd = np.zeros((1, 6))
    pd[0, 0:3] = ref_pos[3]
    v = ref_pos[3] - ref_pos[2]
    pd[0, 3:6] = v / np.sqrt((v ** 2).sum())
    g.write('affine_residue', np.arange(n_res, dtype='i4'))
    g.write('layer_index', np.zeros(n_res, dtype='i4'))
    g.write('placement_data', pd)

    g = pot.create_group('weighted_pos'); _args(g, [sc_node_name, pl_node_name])
    g.write('index_pos', np.arange(n_sc, dtype='i4')); g.write('index_weight', np.arange(n_sc, dtype='i4'))

    g = pot.create_group('environment_coverage')
    _args(g, ['placement_fixed_point_vector_only_CB', 'weighted_pos'])
    g.write('index1', np.arange(n_res, dtype='i4'))
    g.write('type1', np.array([env_restype[s] for s in fasta], dtype='i4'))
    g.write('id1', np.arange(n_res, dtype='i4'))
    g.write('index2', np.arange(n_sc, dtype='i4'))
    g.write('type2', np.zeros(n_sc, dtype='i4'))
    g.write('id2', affine_residue)
    g.write('interaction_param', env_coverage_param)

    g = pot.create_group('nonlinear_coupling_environment'); _args(g, ['environment_coverage'])
    g.write('coeff', env_energies)
    g.set_attr('spline_offset', env_offset, 'coeff')
    g.set_attr('spline_inv_dx', env_inv_dx, 'coeff')
    g.write('coupling_types', np.array([env_restype[s] for s in fasta], dtype='i4'))

    # --- Ramachandran maps ------------------------------------------------------------------
    g = pot.create_group('rama_map_pot'); _args(g, ['rama_coord'])
    if rama_library:
        seq = [str(x) for x in fasta_cpr]
        rama_pot = library_rama_potential(seq, rama_library, rama_sheet_mixing_energy, secstr_bias, rama_combining_rule)
        map_id = np.arange(n_res, dtype='i4')
        if rama_sheet_mixing_energy is not None:      # the maps a little more / less sheet-like: finite differences in the mixing energy (:698-703)
            eps = 1e-2
            g.set_attr('sheet_eps', eps)
            g.write('more_sheet_rama_pot', read_weighted_maps(seq, rama_library, rama_sheet_mixing_energy + eps).astype('f4'))
            g.write('less_sheet_rama_pot', read_weighted_maps(seq, rama_library, rama_sheet_mixing_energy - eps).astype('f4'))
    elif per_residue_rama:
        rama_pot = synthetic_rama_maps(n_res, rama_ref, rama_seed)
        map_id = np.arange(n_res, dtype='i4')
    else:
        rama_pot = synthetic_rama_maps(20, rama_ref, rama_seed)
        map_id = np.array([aa_sorted.index(s if s != 'CPR' else 'PRO') for s in fasta], dtype='i4')
    g.write('residue_id', np.arange(n_res, dtype='i4'))
    g.write('rama_map_id', map_id)
    g.write('rama_pot', rama_pot.astype('f4'))

    ref_cor = np.log(rama_ref)
    ref_cor -= ref_cor.mean()
    g = pot.create_group('rama_map_pot_ref'); _args(g, ['rama_coord'])
    g.set_attr('log_pot', 0)
    g.write('residue_id', np.arange(n_res, dtype='i4'))
    g.write('rama_map_id', np.zeros(n_res, dtype='i4'))
    g.write('rama_pot', ref_cor[None])

    if cavity_radius:
        g = pot.create_group('cavity_radial'); _args(g, ['pos'])
        g.write('id', np.arange(n_atom, dtype='i4'))
        g.write('radius', np.ones(n_atom) * cavity_radius)
        g.write('spring_constant', np.ones(n_atom) * 5.)

    # --- backbone sterics -------------------------------------------------------------------
    g = pot.create_group('backbone_pairs'); _args(g, ['affine_alignment'])
    rp = np.zeros((n_res, 4, 3))
    rp[:, 0] = (-1.19280531, -0.83127186, 0.)
    rp[:, 1] = (0., 0., 0.)
    rp[:, 2] = (1.25222632, -0.87268266, 0.)
    rp[:, 3] = (0., 0.94375626, 1.2068012)
    rp[fasta == 'GLY', 3] = np.nan
    rp -= rp[:, :3].mean(axis=1)[:, None]
    g.write('id', np.arange(n_res, dtype='i4'))
    g.write('ref_pos', rp)
    g.write('n_atom', np.isfinite(rp.sum(axis=-1)).sum(axis=-1).astype('i4'))

    # --- rotamer ---------------------------------------------------------------------------
    g = pot.create_group('rotamer')
    _args(g, [sc_node_name, pl_node_name, 'hbond_coverage', 'hbond_coverage_hydrophobe'])
    g.set_attr('max_iter', 1000); g.set_attr('tol', 1e-3)
    g.set_attr('damping', float(rotamer_damping)); g.set_attr('iteration_chunk_size', 2)
    pg = g.create_group('pair_interaction')
    pg.write('interaction_param', pair_interaction.astype('f4'))
    pg.write('index', np.arange(n_sc, dtype='i4'))
    pg.write('type', np.array([bead_num[s] for s in beadtype_seq], dtype='i4'))
    pg.write('id', np.array(id_seq, dtype='i4'))

    # --- backbone-derived coordinates -------------------------------------------------------
    g = pot.create_group('rama_coord'); _args(g, ['pos'])
    N_id = 3 * np.arange(n_res)
    ids = np.column_stack((N_id - 1, N_id, N_id + 1, N_id + 2, N_id + 3))
    ids[ids >= n_atom] = -1
    g.write('id', ids.astype('i4'))

    g = pot.create_group('affine_alignment'); _args(g, ['pos'])
    rg = np.zeros((n_res, 3, 3))
    rg[:, 0] = (-1.19280531, -0.83127186, 0.)
    rg[:, 1] = (0., 0., 0.)
    rg[:, 2] = (1.25222632, -0.87268266, 0.)
    rg -= rg.mean(axis=1)[:, None]
    g.write('atoms', np.column_stack((N_id, N_id + 1, N_id + 2)).astype('i4'))
    g.write('ref_geom', rg)

    f.close()
    return dict(n_res=n_res, n_atom=n_atom, n_bead=n_sc, n_donor=n_donor, n_acceptor=n_acceptor)


def read_pos(path):
    with h5lite.open_file(path) as f:
        return f.read('input/pos', 'f4')[:, :, 0]


def read_last_frame(path):
    """last frame of /output/pos (frame,1,n_atom,3) written by the reference's H5Logger
    (/root/reference/src/main.cpp:526-531)."""
    with h5lite.open_file(path) as f:
        p = f.read('output/pos', 'f4')
    return p[-1, 0]


def add_pivot_moves(path):
    """write /input/pivot_moves into an existing configuration, exactly as py/upside_config.py:1660-1669 does from the
    rama_coord and rama_map_pot nodes: one pivot location per non-terminal residue (the 5 Ramachandran atoms), the
    rigid tail [nextN+1, n_atom) that a pivot rotates, and the residue's Ramachandran map as proposal potential."""
    with h5lite.open_file(path, 'r+') as f:
        inp = f.group('input')
        pot = inp.group('potential')
        pivot_atom = pot.group('rama_coord').read('id', 'i4')
        rama_pot = pot.group('rama_map_pot').read('rama_pot', 'f4')
        map_id = pot.group('rama_map_pot').read('rama_map_id', 'i4')
        n_atom = inp.shape('pos')[0]
        keep = ~(pivot_atom == -1).any(axis=1)
        if 'pivot_moves' in inp:
            inp.delete('pivot_moves')
        g = inp.create_group('pivot_moves')
        g.write('proposal_pot', rama_pot)
        g.write('pivot_atom', pivot_atom[keep])
        g.write('pivot_restype', map_id[keep])
        g.write('pivot_range', np.column_stack((pivot_atom[keep][:, 4] + 1, np.full(int(keep.sum()), n_atom, 'i4'))).astype('i4'))


def break_chains(path, chain_first_residue=None, rl_chains=None, jump_length_scale=5., jump_rotation_scale=30., remove_pivot=False):
    """py/ugly_hack_break_chain.py restated: cut a configuration into chains at `chain_first_residue` (default: the
    /input/chain_break/chain_first_residue that write_config recorded).
      * bonded terms whose atoms lie in more than one chain are removed: every angle_spring and dihedral_spring row, the
        dist_spring rows that are bonds (bonded_atoms != 0; non-bonded springs across chains are restraints and stay) (:129-131);
      * a rama_coord row across a junction loses the dihedral it cannot have: phi of a chain's first residue (id[0] = -1),
        psi of a chain's last (id[4] = -1) (:143-157);
      * /input/jump_moves gets one rigid-body move per chain, or per receptor / ligand group with rl_chains = (n_receptor
        chains, n_ligand chains) (:96-120), widths in Angstrom / degrees; remove_pivot drops /input/pivot_moves (:92-93);
      * infer_H_O sites drawing on two chains are an error (:134-139: they must have been excluded when the file was written).
    Returns the number of rows removed per node."""
    with h5lite.open_file(path, 'r+') as f:
        inp = f.group('input')
        pot = inp.group('potential')
        n_atom = inp.shape('pos')[0]
        if chain_first_residue is None:
            if 'chain_break' not in inp:
                raise ValueError('no /input/chain_break in %s and no chain_first_residue given' % path)
            chain_first_residue = inp.group('chain_break').read('chain_first_residue', 'i4')
        else:
            chain_first_residue = np.asarray(chain_first_residue, 'i4').reshape(-1)
            if 'chain_break' in inp:
                inp.delete('chain_break')
            if len(chain_first_residue):
                inp.create_group('chain_break').write('chain_first_residue', chain_first_residue)
        if rl_chains is not None:
            if 'chain_break' not in inp:
                raise ValueError('rl_chains needs chain breaks')
            g = inp.group('chain_break')
            if 'rl_chains' in g:
                g.delete('rl_chains')
            g.write('rl_chains', np.asarray(rl_chains, 'i4').reshape(2))
        elif 'chain_break' in inp and 'rl_chains' in inp.group('chain_break'):
            rl_chains = inp.group('chain_break').read('rl_chains', 'i4')
        starts = np.concatenate(([0], 3 * np.asarray(chain_first_residue, 'i8')))       # first atom of every chain
        chain_of = lambda ids: (np.asarray(ids)[..., None] >= starts).sum(axis=-1)      # (negative ids: chain 0, as the reference counts them)
        multichain = lambda ids: np.array([len(set(r)) > 1 for r in chain_of(ids)], dtype=bool)
        removed = {}
        # (checked before anything is modified: a refused file is left as it was)
        if 'infer_H_O' in pot:
            io = pot.group('infer_H_O')
            if multichain(io.group('donors').read('id', 'i4')).any() or multichain(io.group('acceptors').read('id', 'i4')).any():
                raise ValueError('an infer_H_O site draws on atoms of two chains: write the configuration with chain_first_residue '
                                 '(or hbond_exclude_residues) so that the junction residues are excluded')

        def cut(name, others, consider=None):
            if name not in pot:
                return
            g = pot.group(name)
            ids = g.read('id', 'i4')
            drop = multichain(ids)
            if consider is not None:
                drop &= consider(g)
            removed[name] = int(drop.sum())
            keep = ~drop
            for nm in ['id'] + others:
                arr = g.read(nm)
                g.delete(nm)
                g.write(nm, arr[keep])

        cut('angle_spring', ['equil_dist', 'spring_const'])
        cut('dihedral_spring', ['equil_dist', 'spring_const'])
        cut('dist_spring', ['equil_dist', 'spring_const', 'bonded_atoms'], lambda g: g.read('bonded_atoms', 'i4') != 0)
        if 'rama_coord' in pot:
            g = pot.group('rama_coord')
            ids = g.read('id', 'i4')
            n_edit = 0
            for loc in np.nonzero(multichain(ids))[0]:
                c = chain_of(ids[loc])
                if not (c[1] == c[2] == c[3] and (c[0] == c[1] or c[3] == c[4])):
                    raise ValueError('weird rama_coord row %d, unable to proceed' % loc)
                if c[0] == c[1]:
                    ids[loc, 4] = -1      # cut psi
                else:
                    ids[loc, 0] = -1      # cut phi
                n_edit += 1
            g.delete('id'); g.write('id', ids)
            removed['rama_coord (dihedrals cut)'] = n_edit
        if remove_pivot and 'pivot_moves' in inp:
            inp.delete('pivot_moves')
        ends = np.concatenate((starts, [n_atom]))
        if rl_chains is None:
            ranges = [[ends[i], ends[i + 1]] for i in range(len(starts))]
        else:
            ranges = [[ends[0], ends[int(rl_chains[0])]], [ends[int(rl_chains[0])], ends[-1]]]
    add_jump_moves(path, ranges, [jump_length_scale] * len(ranges), [jump_rotation_scale * np.pi / 180.] * len(ranges))
    return removed


def add_jump_moves(path, atom_ranges, sigma_trans, sigma_rot):
    """write /input/jump_moves (monte_carlo_sampler.cpp:173-201): rigid-body Monte-Carlo moves of the atom ranges
    [first, end) -- the chains of a multi-chain system -- with the given translation / rotation scales."""
    with h5lite.open_file(path, 'r+') as f:
        inp = f.group('input')
        if 'jump_moves' in inp:
            inp.delete('jump_moves')
        g = inp.create_group('jump_moves')
        g.write('atom_range', np.asarray(atom_ranges, 'i4').reshape(-1, 2))
        g.write('sigma_trans', np.asarray(sigma_trans, 'f4').reshape(-1))
        g.write('sigma_rot', np.asarray(sigma_rot, 'f4').reshape(-1))


def radial_spline_params(rs, n_type1, n_type2, symmetric, inv_dx=1.5, scale=0.3):
    """interaction_param (n_type1, n_type2, 17) obeying the four conditions of sidechain_radial.cpp:23-27: p[0] = 1/dx,
    p[1] == p[3] (flat at the origin), p[-3] == p[-1] and zero value at the cut-off"""
    c = scale * rs.normal(size=(n_type1, n_type2, 16))
    c[..., 2] = c[..., 0]
    c[..., 15] = c[..., 13]
    c[..., 14] = -0.5 * c[..., 13]              # (1/6) c13 + (2/3) c14 + (1/6) c15 = 0
    if symmetric:
        c = 0.5 * (c + c.transpose(1, 0, 2))
    return np.concatenate((np.full((n_type1, n_type2, 1), inv_dx), c), axis=2)


def add_restraints(path, z_flat_bottom=None, tension=None, afm=None, pos_spring=None, contacts=None, membrane=None,
                   linear_coupling=None, slice_spring=None, radial=None, hbond_sc_radial=None):
    """add the optional restraint / external-field nodes of py/upside_config.py to an existing configuration, with
    the dataset names and argument lists the reference's writers use:
      z_flat_bottom: rows (residue, z0, radius, spring_constant)        upside_config.py:46-79  (CA atom = 3*res+1)
      tension:       rows (residue, tx, ty, tz)                         upside_config.py:82-108
      afm:           (rows (residue, k, tip xyz, vel xyz), time_initial, time_step)   upside_config.py:111-146
      pos_spring:    rows (atom, x0 xyz, k)                             atom_pos_spring, src/bonds.cpp:9-50
      contacts:      rows (res1, res2, energy, distance, width)         upside_config.py:814-853 (+ write_CB, :795-812)
      membrane:      dict(cb_energy (n_restype,nz), uhb_energy (2,nz), z_min, z_max, cov_midpoint, cov_sharpness,
                          residue_type (n_res))                         the datasets of upside_config.py:1137-1149
      linear_coupling: dict(couplings (n_type), inactivation (bool))    linear_coupling_* on environment_coverage
                                                                        (the commented-out writer, upside_config.py:274-285)
      slice_spring:  atom ids: a `slice` of pos (src/bonds.cpp:589-621) feeding an `atom_pos_spring`
      radial:        interaction_param (n_type, n_type, 17), types = the residue types of nonlinear_coupling_environment:
                     the CB-CB pair potential of upside_config.py:866-883 (`radial`, src/sidechain_radial.cpp:81-104)
      hbond_sc_radial: interaction_param (2, n_type, 17): the same functor between the inferred H / O sites (type 0 / 1)
                     and the CB points (src/sidechain_radial.cpp:107-136; no writer in upside_config.py)
    """
    with h5lite.open_file(path, 'r+') as f:
        inp = f.group('input')
        pot = inp.group('potential')
        n_atom = inp.shape('pos')[0]
        n_res = n_atom // 3

        def ca(rows):
            res = np.asarray([int(r[0]) for r in rows])
            assert ((0 <= res) & (res < n_res)).all()
            return (res * 3 + 1).astype('i4')
        if z_flat_bottom is not None:
            r = np.asarray(z_flat_bottom, 'f8')
            g = pot.create_group('z_flat_bottom'); _args(g, ['pos'])
            g.write('atom', ca(r)); g.write('z0', r[:, 1]); g.write('radius', r[:, 2]); g.write('spring_constant', r[:, 3])
        if tension is not None:
            r = np.asarray(tension, 'f8')
            g = pot.create_group('tension'); _args(g, ['pos'])
            g.write('atom', ca(r)); g.write('tension_coeff', r[:, 1:4])
        if afm is not None:
            rows, time_initial, time_step = afm
            r = np.asarray(rows, 'f8')
            g = pot.create_group('AFM'); _args(g, ['pos'])
            g.write('atom', ca(r)); g.write('spring_const', r[:, 1]); g.write('starting_tip_pos', r[:, 2:5]); g.write('pulling_vel', r[:, 5:8])
            g.set_attr('time_initial', float(time_initial), 'pulling_vel'); g.set_attr('time_step', float(time_step), 'pulling_vel')
        if pos_spring is not None:
            r = np.asarray(pos_spring, 'f8')
            g = pot.create_group('atom_pos_spring'); _args(g, ['pos'])
            g.write('id', r[:, 0].astype('i4')); g.write('x0', r[:, 1:4]); g.write('spring_const', r[:, 4])
        if (contacts is not None or membrane is not None or radial is not None or hbond_sc_radial is not None) \
                and 'placement_fixed_point_only_CB' not in pot:
            g = pot.create_group('placement_fixed_point_only_CB'); _args(g, ['affine_alignment'])
            ref = np.array([(-1.19280531, -0.83127186, 0.), (0., 0., 0.), (1.25222632, -0.87268266, 0.), (0., 0.94375626, 1.2068012)])
            ref -= ref[:3].mean(axis=0, keepdims=True)
            g.write('affine_residue', np.arange(n_res, dtype='i4')); g.write('layer_index', np.zeros(n_res, 'i4'))
            g.write('placement_data', ref[3:4].copy())
        if contacts is not None:
            r = np.asarray(contacts, 'f8')
            assert (r[:, 4] > 0).all()
            g = pot.create_group('contact'); _args(g, ['placement_fixed_point_only_CB'])
            g.write('id', r[:, :2].astype('i4')); g.write('energy', r[:, 2]); g.write('distance', r[:, 3]); g.write('width', r[:, 4])
        if membrane is not None:
            seq = [x.decode() for x in inp.read('sequence')]
            g = pot.create_group('membrane_potential')
            _args(g, ['placement_fixed_point_only_CB', 'environment_coverage', 'protein_hbond'])
            donors = np.array([i for i in range(n_res) if i > 0 and seq[i] != 'PRO'], 'i4')       # upside_config.py:1058-1059
            acceptors = np.array([i for i in range(n_res) if i < n_res - 1], 'i4')
            g.write('cb_index', np.arange(n_res, dtype='i4')); g.write('env_index', np.arange(n_res, dtype='i4'))
            g.write('residue_type', np.asarray(membrane['residue_type'], 'i4'))
            g.write('cov_midpoint', np.asarray(membrane['cov_midpoint'], 'f8')); g.write('cov_sharpness', np.asarray(membrane['cov_sharpness'], 'f8'))
            g.write('cb_energy', np.asarray(membrane['cb_energy'], 'f8')); g.write('uhb_energy', np.asarray(membrane['uhb_energy'], 'f8'))
            g.write('donor_residue_ids', donors); g.write('acceptor_residue_ids', acceptors)
            for nm in ('cb_energy', 'uhb_energy'):
                g.set_attr('z_min', float(membrane['z_min']), nm); g.set_attr('z_max', float(membrane['z_max']), nm)
        if linear_coupling is not None:
            types = pot.group('nonlinear_coupling_environment').read('coupling_types', 'i4')
            name = 'linear_coupling_with_inactivation_env' if linear_coupling.get('inactivation') else 'linear_coupling_uniform_env'
            g = pot.create_group(name)
            if linear_coupling.get('inactivation'):
                # one residue per environment element; the burial of a residue is switched off by the hydrogen-bond
                # probability (component 6) of a protein_hbond row -- a synthetic pairing, only to exercise the node
                _args(g, ['environment_coverage', 'slice_hbond_for_coupling'])
                g.set_attr('inactivation_dim', 6)
                sg = pot.create_group('slice_hbond_for_coupling'); _args(sg, ['protein_hbond'])
                n_hb = pot.group('protein_hbond').shape('index1')[0] + pot.group('protein_hbond').shape('index2')[0]
                sg.write('id', (np.arange(len(types)) % n_hb).astype('i4'))
            else:
                _args(g, ['environment_coverage'])
            g.write('couplings', np.asarray(linear_coupling['couplings'], 'f8')); g.write('coupling_types', types)
        if radial is not None or hbond_sc_radial is not None:
            restype = pot.group('nonlinear_coupling_environment').read('coupling_types', 'i4')
            res = np.arange(n_res, dtype='i4')
        if radial is not None:
            g = pot.create_group('radial'); _args(g, ['placement_fixed_point_only_CB'])
            g.write('index', res); g.write('type', restype); g.write('id', res)
            g.write('interaction_param', np.asarray(radial, 'f8'))
        if hbond_sc_radial is not None:
            ph = pot.group('protein_hbond')
            id1, id2 = ph.read('id1', 'i4'), ph.read('id2', 'i4')
            g = pot.create_group('hbond_sc_radial'); _args(g, ['infer_H_O', 'placement_fixed_point_only_CB'])
            g.write('index1', np.arange(len(id1) + len(id2), dtype='i4'))
            g.write('type1', np.concatenate((np.zeros(len(id1), 'i4'), np.ones(len(id2), 'i4'))))
            g.write('id1', np.concatenate((id1, id2)))
            g.write('index2', res); g.write('type2', restype); g.write('id2', res)
            g.write('interaction_param', np.asarray(hbond_sc_radial, 'f8'))
        if slice_spring is not None:
            ids = np.asarray(slice_spring, 'i4')
            sg = pot.create_group('slice_pos_for_spring'); _args(sg, ['pos'])
            sg.write('id', ids)
            g = pot.create_group('atom_pos_spring_on_slice'); _args(g, ['slice_pos_for_spring'])
            g.write('id', np.arange(len(ids), dtype='i4')); g.write('x0', np.zeros((len(ids), 3))); g.write('spring_const', np.full(len(ids), 0.3))


# --- real structures: the main path of py/PDB_to_initial_structure.py without ProDy ---------------------------------
def _dihedral(x1, x2, x3, x4):
    b1, b2, b3 = x2 - x1, x3 - x2, x4 - x3
    b2b3 = np.cross(b2, b3)
    return np.arctan2(np.sqrt((b2 ** 2).sum()) * (b1 * b2b3).sum(), (np.cross(b1, b2) * b2b3).sum())


def read_pdb_backbone(path, chains=None, model=None, recenter=True, allow_unexpected_chain_breaks=False):
    """N, CA, C coordinates and the sequence of a PDB file, as py/PDB_to_initial_structure.py:104-186 prepares them:
    residues with a complete backbone only, MSE read as MET (:26-28), unknown residue types skipped (:58-60), a proline
    after a cis peptide bond named CPR (:88), chains concatenated in file order with an error for a chain break inside a
    chain (N more than 2 A from the previous C, :147-153), centre of mass moved to the origin unless recenter=False.
    Alternate locations other than the first are ignored; `model` selects one MODEL of an NMR file (default: the first).
    Returns (fasta: array of three-letter codes, pos: (3*n_res, 3) float64, chain_first_residue: list of indices)."""
    standard = set(x for x in ('ALA ARG ASN ASP CYS GLN GLU GLY HIS ILE LEU LYS MET PHE PRO SER THR TRP TYR VAL').split())
    wanted = set(chains) if chains else None
    residues = []          # [chain, resid, restype, {atom: xyz}]
    cur_model = None
    for ln in open(path):
        rec = ln[:6]
        if rec.startswith('MODEL'):
            cur_model = int(ln.split()[1])
            continue
        if rec.startswith('ENDMDL') and (model is None or cur_model == model) and residues:
            break
        if rec not in ('ATOM  ', 'HETATM'):
            continue
        if model is not None and cur_model is not None and cur_model != model:
            continue
        altloc, chain = ln[16], ln[21]
        if wanted is not None and chain not in wanted:
            continue
        name, resname, resid = ln[12:16].strip(), ln[17:20].strip(), ln[22:27]
        resname = {'MSE': 'MET'}.get(resname, resname)
        if resname not in standard:
            continue
        key = (chain, resid)
        if not residues or residues[-1][:2] != list(key):
            residues.append([chain, resid, resname, {}, altloc])
        if altloc not in (' ', residues[-1][4]) and residues[-1][4] != ' ':
            continue
        if altloc != ' ' and residues[-1][4] == ' ':
            residues[-1][4] = altloc
        residues[-1][3].setdefault(name, np.array([float(ln[30:38]), float(ln[38:46]), float(ln[46:54])]))
    if wanted is not None:
        missing = wanted.difference(r[0] for r in residues)
        if missing:
            raise ValueError('missing chain %s' % ','.join(sorted(missing)))
    residues = [r for r in residues if all(a in r[3] for a in ('N', 'CA', 'C'))]
    if not residues:
        raise ValueError('no residue with a complete backbone in %s' % path)
    fasta, coords, chain_first, unexpected = [], [], [], []
    for i, (chain, resid, resname, atoms, _) in enumerate(residues):
        new_chain = i > 0 and chain != residues[i - 1][0]
        if coords:
            dist = float(np.sqrt(((atoms['N'] - coords[-1]) ** 2).sum()))
            if dist > 2.:
                if new_chain:
                    chain_first.append(len(fasta))
                else:
                    unexpected.append((len(fasta), dist))
            if resname == 'PRO' and dist <= 2.:
                omega = _dihedral(coords[-2], coords[-1], atoms['N'], atoms['CA'])
                if abs(omega) < 0.5 * np.pi:
                    resname = 'CPR'
        fasta.append(resname)
        coords.extend([atoms['N'], atoms['CA'], atoms['C']])
    if unexpected and not allow_unexpected_chain_breaks:
        raise ValueError('unexpected chain break(s) at residue(s) %s (probably residues missing from the structure)'
                         % ', '.join('%i (%.1f A)' % u for u in unexpected))
    pos = np.array(coords, dtype='f8')
    if recenter:
        pos -= pos.mean(axis=0)
    return np.array(fasta), pos, chain_first


def write_pdb_backbone(path, fasta, pos, chain='A'):
    """the inverse, for round trips and for looking at structures: N, CA, C records only"""
    with open(path, 'w') as f:
        k = 1
        for i, aa in enumerate(fasta):
            for j, (nm, el) in enumerate((('N', 'N'), ('CA', 'C'), ('C', 'C'))):
                x = pos[3 * i + j]
                f.write('ATOM  %5i  %-3s %3s %1s%4i    %8.3f%8.3f%8.3f  1.00  0.00          %2s\n'
                        % (k, nm, 'PRO' if aa == 'CPR' else aa, chain, i + 1, x[0], x[1], x[2], el))
                k += 1
        f.write('END\n')
