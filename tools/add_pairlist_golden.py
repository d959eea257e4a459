#!/usr/bin/env python3
"""Add the reference's own edge lists of the four asymmetric interaction graphs (protein_hbond, hbond_coverage,
hbond_coverage_hydrophobe, environment_coverage) to tests/golden/<fixture>.golden.npz as `pairlist/<node>` (n_edge, 2) int32,
in the reference's edge order.  Build container only: runs oracle/_ref/pairlist_dump_<variant> (oracle/pairlist_dump.cpp, the
unmodified reference's nodes after a force pass on the fixture's structure).  Everything else in the files is left as it is."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
REF = os.path.join(ROOT, 'oracle', '_ref')
FIX = {'trpcage20_7A': '7A', 'proteinG56_7A': '7A', 'syn300_7A': '7A', 'syn150_10A': '10A', 'syn300_10A': '10A'}
for name, variant in sorted(FIX.items()):
    dump = '/tmp/_pairs_%s.txt' % name
    subprocess.check_call([os.path.join(REF, 'pairlist_dump_' + variant), os.path.join(GOLD, name + '.up'), dump])
    lines = open(dump).read().split('\n')
    g = dict(np.load(os.path.join(GOLD, name + '.golden.npz')))
    i = 0
    while i < len(lines):
        if lines[i].startswith('graph '):
            _, node, _, n = lines[i].split(); n = int(n)
            arr = np.array([ln.split() for ln in lines[i + 1:i + 1 + n]], dtype='i4').reshape(n, 2)
            g['pairlist/' + node] = arr
            print(name, node, n)
            i += n + 1
        else:
            i += 1
    # the rotamer list recorded earlier must be what the rebuilt dumper prints (the reference did not change)
    n0 = int(lines[0].split()[1])
    rot = np.array([ln.split()[:2] for ln in lines[1:1 + n0]], dtype='i4')
    assert np.array_equal(rot, g['pairlist/edges'][:, :2]), name
    os.remove(dump)
    np.savez_compressed(os.path.join(GOLD, name + '.golden.npz'), **g)
