#!/usr/bin/env python3
"""Quick GPU-side parity report: HIP product vs the C oracle on every fixture (writes gpurun_out/parity_report.txt)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity_util as P

names = sys.argv[1:] or ['trpcage20_7A', 'proteinG56_7A', 'syn150_10A', 'syn300_10A', 'syn300_7A']
out = open(os.path.join(ROOT, 'gpurun_out', 'parity_report.txt'), 'w')
def log(*a):
    s = ' '.join(str(x) for x in a)
    print(s); out.write(s + '\n'); out.flush()

import ctypes as ct
for name in names:
    log('=====', name)
    t0 = time.time()
    up = P.pkg.Upside(P.fixture(name))
    log('construct %.2fs' % (time.time() - t0))
    orc = P.pkg.Upside(P.fixture(name), library=P.oracle_library())
    g = P.golden(name)
    for tag, x in (('pos', g['pos']), ('pos2', g['pos2'])):
        ref = P.evaluate_all(orc, x)
        t0 = time.time(); act = P.evaluate_all(up, x); t1 = time.time() - t0
        rows = []
        for k in sorted(ref):
            r, a = np.asarray(ref[k]), np.asarray(act[k])
            if r.ndim == 0:
                rows.append((k, abs(float(r) - float(a)) / max(1., abs(float(r))), 0.))
            else:
                rows.append((k, P.rel_rms(r, a), P.max_rel_to_scale(r, a)))
        log('--', tag, 'eval %.3fs' % t1, ' energy hip %.6f oracle %.6f golden %.6f' % (act['energy'], ref['energy'], g['energy' if tag == 'pos' else 'energy2']))
        for k, e1, e2 in rows:
            flag = '' if e1 <= 1e-5 else ('  <-- ' + ('BAD' if e1 > 1e-4 else 'warn'))
            log('   %-46s %.2e %.2e%s' % (k, e1, e2, flag))
        # pair lists
        up.calc.upside_hip_get_pairlist.restype = ct.c_int
        up.calc.upside_hip_get_pairlist.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p]
        for node in ('rotamer', 'hbond_coverage', 'hbond_coverage_hydrophobe', 'environment_coverage', 'protein_hbond'):
            po = P.oracle_pairlist(orc, node)
            i1 = np.zeros(len(po) + 1000, 'i4'); i2 = np.zeros(len(po) + 1000, 'i4')
            n = up.calc.upside_hip_get_pairlist(up.engine, node.encode(), 0, len(i1), i1.ctypes.data, i2.ctypes.data)
            ph = np.column_stack((i1[:max(n, 0)], i2[:max(n, 0)]))
            same = n == len(po) and np.array_equal(ph, po)
            log('   pairlist %-28s oracle %6d hip %6d %s' % (node, len(po), n, 'EXACT' if same else 'MISMATCH'))
        it = np.zeros(1, 'i4')
        up.calc.upside_hip_rotamer_iterations.argtypes = [ct.c_void_p, ct.c_void_p]
        up.calc.upside_hip_rotamer_iterations(up.engine, it.ctypes.data)
        log('   BP sweeps hip %d oracle %d' % (it[0], orc.calc.oracle_rotamer_iterations(orc.engine)))
    up.close()
log('done')
