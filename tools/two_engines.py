#!/usr/bin/env python3
"""Experiment (GPU box): R systems held as G engines of R / G systems each, stepped concurrently (one host thread per engine; every
engine has its own streams), against one engine of R.  usage: two_engines.py workload R G [steps] [stagger ms]"""
import ctypes as ct, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from __graft_entry__ import load_package

def main():
    w, R, G = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
    stagger_ms = float(sys.argv[5]) if len(sys.argv) > 5 else 0.      # engine g starts g * stagger_ms late: its steps run out of phase with the others'
    pkg = load_package(); c = bench.bind(pkg.default_library())
    bench.check(c, c.upside_hip_set_device(0), 'set_device')
    fixture = os.path.join(ROOT, 'tests', 'golden', w + '.up')
    pos0 = pkg.config.read_pos(fixture); n_atom = pos0.shape[0]
    engs = []
    for g in range(G):
        r = R // G
        e = c.upside_hip_construct(n_atom, fixture.encode(), r, True)
        pos = np.tile(pos0[None], (r, 1, 1)).astype('f4')
        for k in range(r): pos[k] += np.random.RandomState(977 + g * r + k).normal(0., 0.05, pos0.shape).astype('f4')
        pos = np.ascontiguousarray(pos)
        bench.check(c, c.upside_hip_set_pos(e, pos.ctypes.data), 'set_pos')
        t = np.full(r, bench.TEMPERATURE, dtype='f4')
        bench.check(c, c.upside_hip_init_md(e, t.ctypes.data, 1000 + g * r, 5.0, bench.DT, 1), 'init_md')
        engs.append(e)
    def run(n):
        def go(e, g):
            if stagger_ms: time.sleep(g * stagger_ms * 1e-3)
            bench.check(c, c.upside_hip_run_steps(e, n), 'run_steps')
        th = [threading.Thread(target=go, args=(e, g)) for g, e in enumerate(engs)]
        for t in th: t.start()
        for t in th: t.join()
    run(60)
    t0 = time.perf_counter(); run(steps); dt = time.perf_counter() - t0
    print('stagger %.1f ms ' % stagger_ms + '%s R=%d as %d engine(s): %.0f system-steps/s, %.1f us per step of all systems' % (w, R, G, R * steps / dt, dt / steps * 1e6))
main()
