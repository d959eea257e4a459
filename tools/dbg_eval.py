import sys, os
sys.path.insert(0, 'tests')
import parity_util as P
name = sys.argv[1]
up = P.pkg.Upside(P.fixture(name))
g = P.golden(name)
print('energy', up.energy(g['pos']))
print('deriv ok', up.deriv(g['pos']).shape)
import ctypes as ct, numpy as np
c = up.calc
c.upside_hip_compute.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p]
c.upside_hip_last_error.restype = ct.c_char_p
en = np.zeros(1, 'f4'); der = np.zeros((g['pos'].shape[0], 3), 'f4')
r = c.upside_hip_compute(up.engine, en.ctypes.data, der.ctypes.data)
print('compute rc', r, c.upside_hip_last_error() if r else '')
