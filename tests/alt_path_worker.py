"""Worker of test_alternate_code_paths_agree: evaluates one fixture through the product library under whatever
UPSIDE_HIP_* environment the parent set (the knobs are read once per process) and saves the results."""
import sys
import numpy as np
import parity_util as P

name, out = sys.argv[1], sys.argv[2]
g = P.golden(name)
up = P.pkg.Upside(P.fixture(name))
res = {}
for tag in ('pos', 'pos2'):
    res['energy_' + tag] = up.energy(g[tag])
    res['force_' + tag] = up.deriv(g[tag])
up.close()
ens = P.pkg.engine.Ensemble(P.fixture(name), 3)
ens.set_pos(np.stack([g['pos'], g['pos2'], g['pos']]).astype('f4'))
e, d = ens.energies_and_derivs()
res['ens_energy'] = e; res['ens_force'] = d
ens.init_md(0.8, 3); ens.run_steps(12)
res['md_pos'] = ens.get_pos()
ens.close()
np.savez(out, **res)
