#!/usr/bin/env python3
"""Noise floor of the REFERENCE itself (build container only): the same reference sources compiled -O1 without fast-math
(throw-away builds under /tmp, tools/build_ref_O1.sh) compared with the golden vectors of the -O3 -ffast-math build, next to the
distance of our C restatement from both.  Writes profiles/r03_reference_noise_floor.txt (the table) and
tests/golden/reference_noise_floor.json: per fixture and structure the relative RMS deviation of the forces and the worst one
over the nodes' sensitivities.  The tests derive their tolerances against the golden vectors from that file (2x the floor)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity_util as P, numpy as np
orc = P.oracle_library()
O1 = {'7A': '/tmp/refO1_7A/libupside_O1.so', '10A': '/tmp/refO1_10A/libupside_O1.so'}
FIX = [('trpcage20_7A', '7A'), ('proteinG56_7A', '7A'), ('syn300_7A', '7A'), ('syn150_10A', '10A'), ('syn300_10A', '10A'),
       ('proteinG56_restraints', '7A'), ('edge_gly5', '7A'), ('edge_pro6', '7A'), ('edge_awa3', '7A')]
table, lines = {}, []
for name, variant in FIX:
    lib = P.pkg.UpsideLibrary(O1[variant])
    g = P.golden(name)
    extra = (P.RESTRAINT_COORDS, P.RESTRAINT_POTENTIALS) if name.endswith('restraints') else ((), ())
    up = P.pkg.Upside(P.fixture(name), library=lib)
    uo = P.pkg.Upside(P.fixture(name), library=orc)
    table[name] = {}
    for tag in ('pos', 'pos2'):
        if tag not in g or ('deriv2' if tag == 'pos2' else 'deriv') not in g:
            continue
        a = P.evaluate_all(up, g[tag], *extra); o = P.evaluate_all(uo, g[tag], *extra)
        gd = g['deriv' if tag == 'pos' else 'deriv2']
        entry = dict(deriv=float(P.rel_rms(gd, a['deriv'])), oracle_deriv=float(P.rel_rms(gd, o['deriv'])))
        line = '%-22s %-4s refO1-vs-golden: deriv %.2e' % (name, tag, entry['deriv'])
        if tag == 'pos':
            sens = [(float(P.rel_rms(g[k], a[k])), k) for k in g if k.startswith('sens/') and k in a and np.asarray(g[k]).size]
            osens = [(float(P.rel_rms(g[k], o[k])), k) for k in g if k.startswith('sens/') and k in o and np.asarray(g[k]).size]
            if sens:
                entry['sens'] = max(sens)[0]; entry['oracle_sens'] = max(osens)[0]
                line += ' worst sens %.2e (%s)' % max(sens)
                line += ' | oracle-vs-golden: deriv %.2e worst sens %.2e' % (entry['oracle_deriv'], entry['oracle_sens'])
        else:
            line += ' | oracle-vs-golden: deriv %.2e' % entry['oracle_deriv']
        line += ' | oracle-vs-refO1: deriv %.2e' % P.rel_rms(a['deriv'], o['deriv'])
        table[name][tag] = entry
        lines.append(line); print(line)
open(os.path.join(ROOT, 'profiles', 'r03_reference_noise_floor.txt'), 'w').write('\n'.join(lines) + '\n')
json.dump(table, open(os.path.join(ROOT, 'tests', 'golden', 'reference_noise_floor.json'), 'w'), indent=1, sort_keys=True)
