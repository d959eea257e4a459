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
