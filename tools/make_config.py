#!/usr/bin/env python3
"""Build a `.up` configuration for a real structure (python3 stand-in for PDB_to_initial_structure.py + the README call
of upside_config.py; needs the reference's parameter directory, so it runs where /root/reference is available).

    python tools/make_config.py --pdb 1abc.pdb --chains A --out 1abc.up [--param-dir /root/reference/parameters]
                                [--cutoff 7|10] [--cavity-radius R] [--contacts table] [--z-flat-bottom table] ...

The Ramachandran maps come from --rama-library (the README's rama.dat: neighbour-dependent coil / sheet maps, with
--rama-sheet-mixing-energy, --rama-library-combining-rule, --secstr-bias as in py/upside_config.py) or, without one, are the
synthetic per-residue maps of config.synthetic_rama_maps around the shipped reference state (rama.dat is not part of the
parameter directory here).  --fix-rotamer, --hbond-exclude-residues and --loose-hbond-criteria as in py/upside_config.py."""
import argparse
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def table(path):
    rows = [ln.split() for ln in open(path) if ln.strip()]
    return np.array([[float(x) for x in r] for r in rows[1:]])     # first line = header, as in upside_config.py


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--pdb'); ap.add_argument('--chains', default='')
    ap.add_argument('--model', type=int, default=None)
    ap.add_argument('--allow-unexpected-chain-breaks', action='store_true')
    ap.add_argument('--out', required=True)
    ap.add_argument('--param-dir', default='/root/reference/parameters')
    ap.add_argument('--cutoff', choices=['7', '10'], default='7')
    ap.add_argument('--cavity-radius', type=float, default=0.)
    ap.add_argument('--contacts'); ap.add_argument('--z-flat-bottom'); ap.add_argument('--tension')
    ap.add_argument('--pivot-moves', action='store_true', help='also write /input/pivot_moves (Monte-Carlo)')
    ap.add_argument('--join-chains', action='store_true', help='keep the chains of the PDB file bonded end to end (default: cut them, '
                    'py/upside_config.py --chain-break-from-file + py/ugly_hack_break_chain.py)')
    ap.add_argument('--rl-chains', nargs=2, type=int, default=None, help='numbers of receptor and ligand chains: two collective jump moves')
    ap.add_argument('--rama-library', default='', help='Ramachandran library file (groups coil and sheet: dimer_pot, dimer_weight)')
    ap.add_argument('--rama-library-combining-rule', default='mixture', choices=['mixture', 'product'])
    ap.add_argument('--rama-sheet-mixing-energy', type=float, default=None, help='energy of the sheet library relative to the coil library')
    ap.add_argument('--secstr-bias', default='', help='table "residue secstr energy" (secstr: helix | sheet)')
    ap.add_argument('--fix-rotamer', default='', help='table "residue restype chain resnum chi1 chi2" (degrees): those residues keep one rotamer state')
    ap.add_argument('--hbond-exclude-residues', default='', help='comma-separated residues without backbone H-bond sites (ranges a-b allowed)')
    ap.add_argument('--loose-hbond-criteria', action='store_true', help='permissive H-bond geometry (static structures only)')
    a = ap.parse_args()
    pkg = load_package(); cfg = pkg.config
    if not a.pdb:
        ap.error('--pdb is required')
    fasta, pos, first = cfg.read_pdb_backbone(a.pdb, chains=[c for c in a.chains.split(',') if c] or None, model=a.model,
                                              allow_unexpected_chain_breaks=a.allow_unexpected_chain_breaks)
    if first and a.join_chains:
        print('note: chains are concatenated; first residues of later chains: %s (--join-chains: bonded terms across them are kept)' % first)
    P = a.param_dir
    sclib = 'ff_1/sidechain.h5' if a.cutoff == '7' else 'packing/sidechain_10A_cutoff.h5'
    info = cfg.write_config(a.out, fasta, pos, sidechain_lib=os.path.join(P, sclib),
                            environment_lib=os.path.join(P, 'ff_1', 'environment.h5'),
                            rama_ref=cfg.load_rama_reference(os.path.join(P, 'common', 'rama_reference.pkl')),
                            hbond_energy=float(open(os.path.join(P, 'ff_1', 'hbond')).read()), cavity_radius=a.cavity_radius,
                            chain_first_residue=() if a.join_chains else first,
                            rama_library=a.rama_library or None, rama_sheet_mixing_energy=a.rama_sheet_mixing_energy,
                            rama_combining_rule=a.rama_library_combining_rule,
                            secstr_bias=cfg.read_secstr_bias(a.secstr_bias) if a.secstr_bias else (),
                            fix_rotamer=cfg.read_fix_rotamer(a.fix_rotamer) if a.fix_rotamer else (),
                            hbond_exclude_residues=[r for seg in a.hbond_exclude_residues.split(',') if seg
                                                    for r in (range(int(seg.split('-')[0]), int(seg.split('-')[-1]) + 1))],
                            loose_hbond_criteria=a.loose_hbond_criteria)
    if first and not a.join_chains:      # upside_config.py --chain-break-from-file + ugly_hack_break_chain.py --chain-break-from-file
        removed = cfg.break_chains(a.out, rl_chains=a.rl_chains)
        print('chains start at residues %s: removed across the junctions %s; one jump move per %s' %
              ([0] + list(first), removed, 'receptor / ligand group' if a.rl_chains else 'chain'))
    extra = {}
    if a.contacts: extra['contacts'] = table(a.contacts)
    if a.z_flat_bottom: extra['z_flat_bottom'] = table(a.z_flat_bottom)
    if a.tension: extra['tension'] = table(a.tension)
    if extra:
        cfg.add_restraints(a.out, **extra)
    if a.pivot_moves:
        cfg.add_pivot_moves(a.out)
    print('%s: %i residues, %i side-chain beads' % (a.out, info['n_res'], info['n_bead']))


if __name__ == '__main__':
    main()
