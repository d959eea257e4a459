"""How often do systems rebuild their pair lists during MD?  usage: python tools/rebuild_rate.py [replicas] [warm steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as ct
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import bench
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lib = pkg.default_library(); c = bench.bind(lib)
c.upside_hip_rebuild_flags.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_void_p]
fx = os.path.join(bench.ROOT, 'tests', 'golden', 'syn300_10A.up')
pos0 = pkg.config.read_pos(fx)
eng = c.upside_hip_construct(pos0.shape[0], fx.encode(), R, True)
pos = np.ascontiguousarray(np.tile(pos0[None], (R, 1, 1)).astype('f4'))
c.upside_hip_set_pos(eng, pos.ctypes.data)
temps = np.full(R, bench.TEMPERATURE, dtype='f4')
c.upside_hip_init_md(eng, temps.ctypes.data, 1000, 5.0, bench.DT, 1)
c.upside_hip_run_steps(eng, warm)
fl = np.zeros(R, dtype='i4')
for node in (b'rotamer', b'hbond_coverage', b'environment_coverage', b'protein_hbond'):
    tot = 0; n = 60
    for i in range(n):
        c.upside_hip_run_steps(eng, 1)
        assert c.upside_hip_rebuild_flags(eng, node, fl.ctypes.data) == 0
        tot += int(fl.sum())
    print('%-22s %.2f systems of %d rebuild per step  -> every %.1f steps per system' % (node.decode(), tot / n, R, R * n / max(tot, 1)))
p = np.zeros_like(pos); c.upside_hip_get_pos(eng, p.ctypes.data)
d = p - pos
print('rms displacement from start after %d steps: %.2f A; radius of gyration %.2f -> %.2f' % (
    warm + 240, np.sqrt((d ** 2).sum(-1).mean()), np.sqrt(((pos[0] - pos[0].mean(0)) ** 2).sum(-1).mean()),
    np.sqrt(((p[0] - p[0].mean(0)) ** 2).sum(-1).mean())))
# per-step displacement statistics of the backbone atoms
q0 = np.zeros_like(pos); q1 = np.zeros_like(pos)
c.upside_hip_get_pos(eng, q0.ctypes.data)
ref = q0.copy(); mx = []
for i in range(40):
    c.upside_hip_run_steps(eng, 1)
    c.upside_hip_get_pos(eng, q1.ctypes.data)
    step = np.sqrt(((q1 - q0) ** 2).sum(-1))
    cum = np.sqrt(((q1 - ref) ** 2).sum(-1))
    mx.append((step.max(), np.sqrt((step ** 2).mean()), cum.max(axis=1).mean(), cum.max()))
    q0[:] = q1
for i in (0, 1, 2, 5, 10, 20, 39):
    print('step %2d: max |dx| %.3f  rms |dx| %.4f   since ref: mean over systems of max atom %.3f, overall max %.3f' % ((i,) + mx[i]))
