#!/usr/bin/env python3
"""Noise floor of the REFERENCE itself (build container only): the same reference sources compiled -O1 without
fast-math (a throw-away build under /tmp) compared with the golden vectors of the -O3 -ffast-math
build, next to the distance of our C restatement from both."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity_util as P, numpy as np
orc = P.oracle_library()
O1 = {'7A': '/tmp/refO1/libupside_O1.so', '10A': '/tmp/refO1_10A/libupside_O1.so'}
for name in ['trpcage20_7A', 'proteinG56_7A', 'syn300_7A', 'syn150_10A', 'syn300_10A']:
    lib = P.pkg.UpsideLibrary(O1[name.split('_')[1]])
    g = P.golden(name)
    up = P.pkg.Upside(P.fixture(name), library=lib)
    uo = P.pkg.Upside(P.fixture(name), library=orc)
    for tag in ('pos', 'pos2'):
        a = P.evaluate_all(up, g[tag]); o = P.evaluate_all(uo, g[tag])
        gd = g['deriv' if tag == 'pos' else 'deriv2']
        line = '%-14s %-4s refO1-vs-golden: deriv %.2e' % (name, tag, P.rel_rms(gd, a['deriv']))
        if tag == 'pos':
            worst = max((P.rel_rms(g[k], a[k]), k) for k in g if k.startswith('sens/'))
            line += ' worst sens %.2e (%s)' % worst
            worst = max((P.rel_rms(g[k], o[k]), k) for k in g if k.startswith('sens/'))
            line += ' | oracle-vs-golden: deriv %.2e worst sens %.2e' % (P.rel_rms(gd, o['deriv']), worst[0])
            worst = max((P.rel_rms(a[k], o[k]), k) for k in a if k.startswith('sens/'))
            line += ' | oracle-vs-refO1: deriv %.2e worst sens %.2e' % (P.rel_rms(a['deriv'], o['deriv']), worst[0])
        else:
            line += ' | oracle-vs-golden: deriv %.2e | oracle-vs-refO1 deriv %.2e' % (P.rel_rms(gd, o['deriv']), P.rel_rms(a['deriv'], o['deriv']))
        print(line)
