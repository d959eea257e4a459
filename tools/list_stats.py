"""Sizes of the cached and in-range pair lists per interaction graph.  usage: python tools/list_stats.py [replicas] [warm steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as ct
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import bench
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lib = pkg.default_library(); c = bench.bind(lib)
c.upside_hip_igraph_stats.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_void_p]
fx = os.path.join(bench.ROOT, 'tests', 'golden', 'syn300_10A.up')
pos0 = pkg.config.read_pos(fx)
eng = c.upside_hip_construct(pos0.shape[0], fx.encode(), R, True)
pos = np.ascontiguousarray(np.tile(pos0[None], (R, 1, 1)).astype('f4'))
c.upside_hip_set_pos(eng, pos.ctypes.data)
temps = np.full(R, bench.TEMPERATURE, dtype='f4')
c.upside_hip_init_md(eng, temps.ctypes.data, 1000, 5.0, bench.DT, 1)
c.upside_hip_run_steps(eng, warm)
out = np.zeros(11)
print('%-22s %5s %5s %5s %5s %6s %6s | cached/sys s1 s2 | hits/sys s1 s2 | mean row s1 s2 (cached) s1 s2 (hit) | sides' % ('node', 'n1', 'n2', 'cap1', 'cap2', 'cut', 'cache'))
for node in (b'rotamer', b'hbond_coverage', b'hbond_coverage_hydrophobe', b'environment_coverage', b'protein_hbond'):
    if c.upside_hip_igraph_stats(eng, node, out.ctypes.data): print(node.decode(), 'absent'); continue
    n1, n2 = out[0], out[1]
    print('%-22s %5d %5d %5d %5d %6.2f %6.2f | %8.0f %8.0f | %8.0f %8.0f | %5.1f %5.1f  %5.1f %5.1f | %d' % (
        node.decode(), n1, n2, out[2], out[3], out[4], out[5], out[6], out[7], out[8], out[9],
        out[6] / max(n1, 1), out[7] / max(n2, 1), out[8] / max(n1, 1), out[9] / max(n2, 1), int(out[10])))
