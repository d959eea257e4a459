import sys, os, ctypes as ct, numpy as np, faulthandler
faulthandler.enable()
if len(sys.argv) > 1 and sys.argv[1] == 'torch':
    import torch; print('torch loaded', torch.cuda.is_available())
sys.path.insert(0, 'tests')
import parity_util as P
lib = P.pkg.default_library(); c = lib.calc
c.upside_hip_construct.restype = ct.c_void_p
c.upside_hip_construct.argtypes = [ct.c_int, ct.c_char_p, ct.c_int, ct.c_bool]
c.upside_hip_comm_get_unique_id.argtypes = [ct.c_char_p]
c.upside_hip_comm_init.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_char_p, ct.c_void_p]
c.upside_hip_last_error.restype = ct.c_char_p
uid = ct.create_string_buffer(128)
print('uid rc', c.upside_hip_comm_get_unique_id(uid), c.upside_hip_last_error()); sys.stdout.flush()
name = 'trpcage20_7A'
g = P.golden(name)
eng = c.upside_hip_construct(g['pos'].shape[0], P.fixture(name).encode(), 6, True)
temps = np.array([0.7, 0.74, 0.78, 0.82, 0.86, 0.9], 'f4')
print('init rc', c.upside_hip_comm_init(eng, 0, 1, uid, temps.ctypes.data), c.upside_hip_last_error())
c.upside_hip_set_pos.argtypes = [ct.c_void_p, ct.c_void_p]
c.upside_hip_init_md.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_uint32, ct.c_float, ct.c_float, ct.c_int]
c.upside_hip_comm_replica_swap.argtypes = [ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_uint32, ct.c_uint64, ct.c_int, ct.c_void_p]
pos = np.stack([g['pos']] * 6).astype('f4')
print('set_pos', c.upside_hip_set_pos(eng, pos.ctypes.data)); print('init_md', c.upside_hip_init_md(eng, temps.ctypes.data, 5, 5.0, 0.009, 1)); sys.stdout.flush()
pairs = np.array([[0, 1], [2, 3], [4, 5]], 'i4'); acc = np.zeros(3, 'i4')
print('swap rc', c.upside_hip_comm_replica_swap(eng, 3, pairs.ctypes.data, 11, 1, 1, acc.ctypes.data), c.upside_hip_last_error(), acc)
