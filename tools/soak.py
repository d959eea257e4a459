"""Long batched run as a stability check (gpurun -- 'python tools/soak.py [replicas] [steps]'): R replicas of the benchmark system,
N MD steps, then every replica's energy must be finite and the engine must report no capacity overflow."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
pkg = load_package()
fixture = os.path.join(ROOT, 'tests', 'golden', 'syn300_10A.up')
ens = pkg.engine.Ensemble(fixture, R)
ens.set_pos(pkg.config.read_pos(fixture))
ens.init_md(np.linspace(0.7, 1.0, R).astype('f4'), 17)
t0 = time.perf_counter()
done = 0
while done < N:
    n = min(600, N - done)
    ens.run_steps(n); done += n
    e = ens.energies()
    print('steps %6d  energy min %.1f mean %.1f max %.1f  finite %s' % (done, e.min(), e.mean(), e.max(), bool(np.isfinite(e).all())), flush=True)
    assert np.isfinite(e).all()
x = ens.get_pos()
assert np.isfinite(x).all() and np.abs(x).max() < 1e3
print('ok: %d replicas x %d steps in %.1f s' % (R, N, time.perf_counter() - t0))
