"""python3 ctypes mirror of the reference's `py/upside_engine.py` (`Upside` class, lines 159-242, and the
spline helpers, lines 93-156) over the engine_c_library C-ABI (/root/reference/src/engine_c_library.h:12-32).

`UpsideLibrary(path)` binds any shared object exporting that ABI: the HIP product
(`libupside_hip.so`), or -- in tests only -- the compiled reference under oracle/_ref.  The product
library is the default and there is NO fallback: if it cannot be loaded the import of
`default_library()` raises.
"""
import ctypes as ct
import os
import numpy as np
from . import h5lite

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_HERE, 'csrc', 'libupside_hip.so')


def _b(s):
    return s if isinstance(s, bytes) else str(s).encode()


class UpsideLibrary(object):
    def __init__(self, path):
        self.path = path
        # libhdf5 is a dependency of every engine build; load it first with RTLD_GLOBAL so a
        # library linked without an rpath still resolves it.
        try:
            h5lite.lib()
        except OSError:
            pass
        c = self.calc = ct.CDLL(path)
        c.construct_deriv_engine.restype = ct.c_void_p
        c.construct_deriv_engine.argtypes = [ct.c_int, ct.c_char_p, ct.c_bool]
        c.free_deriv_engine.restype = None
        c.free_deriv_engine.argtypes = [ct.c_void_p]
        for nm in ('evaluate_energy', 'evaluate_deriv'):
            getattr(c, nm).restype = ct.c_int
            getattr(c, nm).argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p]
        for nm in ('set_param', 'get_param_deriv', 'get_param', 'get_sens', 'get_output'):
            getattr(c, nm).restype = ct.c_int
            getattr(c, nm).argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_char_p]
        c.get_output_dims.restype = ct.c_int
        c.get_output_dims.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_char_p]
        c.get_value_by_name.restype = ct.c_int
        c.get_value_by_name.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_char_p, ct.c_char_p]
        c.get_clamped_value_and_deriv.restype = ct.c_int
        c.get_clamped_value_and_deriv.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p]
        c.clamped_spline_value.restype = ct.c_int
        c.clamped_spline_value.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p]
        c.clamped_spline_solve.restype = ct.c_int
        c.clamped_spline_solve.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p]
        c.get_clamped_coeff_deriv.restype = ct.c_int
        c.get_clamped_coeff_deriv.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_float]
        c.upside_main.restype = ct.c_int

    # -- free functions (upside_engine.py:93-156) ---------------------------------------------
    def clamped_spline_value(self, bspline_coeff, x):
        x = np.require(x, dtype='f4', requirements='C')
        bspline_coeff = np.require(bspline_coeff, dtype='f4', requirements='C')
        result = np.zeros(len(x), dtype='f4')
        if self.calc.clamped_spline_value(len(bspline_coeff), result.ctypes.data, bspline_coeff.ctypes.data,
                                          len(x), x.ctypes.data):
            raise RuntimeError("spline evaluation error")
        return result

    def clamped_spline_solve(self, values):
        values = np.require(values, dtype='f4', requirements='C')
        coeff = np.zeros(len(values) + 2, dtype='f4')
        if self.calc.clamped_spline_solve(len(coeff), coeff.ctypes.data, values.ctypes.data):
            raise RuntimeError("spline solve error")
        return coeff

    def clamped_value_and_deriv(self, bspline_coeff, x):
        x = np.require(x, dtype='f4', requirements='C')
        bspline_coeff = np.require(bspline_coeff, dtype='f4', requirements='C')
        result = np.zeros((len(x), 2), dtype='f4')
        if self.calc.get_clamped_value_and_deriv(len(bspline_coeff), result.ctypes.data,
                                                 bspline_coeff.ctypes.data, len(x), x.ctypes.data):
            raise RuntimeError("spline evaluation error")
        return result

    def clamped_coeff_deriv(self, bspline_coeff, x):
        x = np.asarray(x, dtype='f4')
        bspline_coeff = np.require(bspline_coeff, dtype='f4', requirements='C')
        result = np.zeros((len(x), len(bspline_coeff)), dtype='f4')
        for i, y in enumerate(x):
            if self.calc.get_clamped_coeff_deriv(len(bspline_coeff), result[i].ctypes.data,
                                                 bspline_coeff.ctypes.data, float(y)):
                raise RuntimeError("spline evaluation error")
        return result

    def in_process_upside(self, args, verbose=True):
        exec_args = [b'python_library', b'--re-raise-signal'] + [_b(a) for a in args]
        arr_t = ct.c_char_p * len(exec_args)
        arr = arr_t(*exec_args)
        self.calc.upside_main.argtypes = [ct.c_int, arr_t, ct.c_int]
        ret = self.calc.upside_main(len(exec_args), arr, int(verbose))
        if ret:
            raise RuntimeError('In process Upside returned %i' % ret)


_default = None


def default_library():
    """the HIP product library; raises if it has not been built (no CPU fallback)."""
    global _default
    if _default is None:
        if not os.path.exists(PRODUCT_LIB):
            raise RuntimeError('HIP extension %s is missing: run __graft_entry__.build()' % PRODUCT_LIB)
        _default = UpsideLibrary(PRODUCT_LIB)
    return _default


class Upside(object):
    """same method set as the reference's `Upside` (py/upside_engine.py:159-242)."""

    def __init__(self, config_file_path, quiet=True, library=None):
        self.lib = library if library is not None else default_library()
        self.calc = self.lib.calc
        self.config_file_path = str(config_file_path)
        with h5lite.open_file(self.config_file_path) as t:
            self.initial_pos = t.read('input/pos', 'f4')[:, :, 0]
            self.n_atom = self.initial_pos.shape[0]
            self.sequence = [x.decode() for x in t.read('input/sequence')] if 'input/sequence' in t else None
        self.engine = self.calc.construct_deriv_engine(self.n_atom, _b(self.config_file_path), bool(quiet))
        if not self.engine:
            raise RuntimeError('Unable to initialize upside engine for %s' % (config_file_path,))

    def __repr__(self):
        return 'Upside(%r, %r)' % (self.n_atom, self.config_file_path)

    def energy(self, pos):
        pos = np.require(pos, dtype='f4', requirements='C')
        assert pos.shape == (self.n_atom, 3)
        energy = np.zeros(1, dtype='f4')
        if self.calc.evaluate_energy(energy.ctypes.data, self.engine, pos.ctypes.data):
            raise RuntimeError('Unable to evaluate energy')
        return energy[0]

    def deriv(self, pos):
        pos = np.require(pos, dtype='f4', requirements='C')
        assert pos.shape == (self.n_atom, 3)
        deriv = np.zeros_like(pos)
        if self.calc.evaluate_deriv(deriv.ctypes.data, self.engine, pos.ctypes.data):
            raise RuntimeError('Unable to evaluate derivative')
        return deriv

    def set_param(self, param, node_name):
        param = np.require(np.asarray(param).ravel(), dtype='f4', requirements='C')
        if self.calc.set_param(int(param.shape[0]), param.ctypes.data, self.engine, _b(node_name)):
            raise RuntimeError('Unable to set param for node %s' % node_name)

    def get_param_deriv(self, param_shape, node_name):
        deriv = np.zeros(param_shape, dtype='f4')
        if self.calc.get_param_deriv(int(np.prod(param_shape)), deriv.ctypes.data, self.engine, _b(node_name)):
            raise RuntimeError('Unable to get param deriv')
        return deriv

    def get_param(self, param_shape, node_name):
        param = np.zeros(param_shape, dtype='f4')
        if self.calc.get_param(int(np.prod(param_shape)), param.ctypes.data, self.engine, _b(node_name)):
            raise RuntimeError('Unable to get param')
        return param

    def get_output_dims(self, node_name):
        n_elem = np.zeros(1, dtype=np.intc)
        elem_width = np.zeros(1, dtype=np.intc)
        if self.calc.get_output_dims(n_elem.ctypes.data, elem_width.ctypes.data, self.engine, _b(node_name)):
            raise RuntimeError('Unable to get output dims')
        return int(n_elem[0]), int(elem_width[0])

    def get_sens(self, node_name):
        shape = self.get_output_dims(node_name)
        out = np.zeros(shape, dtype='f4')
        if self.calc.get_sens(int(np.prod(shape)), out.ctypes.data, self.engine, _b(node_name)):
            raise RuntimeError('Unable to get sens')
        return out

    def get_output(self, node_name):
        shape = self.get_output_dims(node_name)
        out = np.zeros(shape, dtype='f4')
        if self.calc.get_output(int(np.prod(shape)), out.ctypes.data, self.engine, _b(node_name)):
            raise RuntimeError('Unable to get output')
        return out

    def get_value_by_name(self, value_shape, node_name, log_name):
        value = np.zeros(value_shape, dtype='f4')
        if self.calc.get_value_by_name(int(np.prod(value_shape)), value.ctypes.data, self.engine,
                                       _b(node_name), _b(log_name)):
            raise RuntimeError('Unable to get value by name')
        return value

    def close(self):
        if getattr(self, 'engine', None):
            self.calc.free_deriv_engine(self.engine)
            self.engine = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Ensemble(object):
    """S systems of one topology resident on one GPU, driven through the `upside_hip_*` extension of the C-ABI
    (include/upside_engine_c.h): what `upside_main` does for its `systems` vector (main.cpp:441-700), with the
    state kept on the device between calls."""

    def __init__(self, config_file_path, n_system, device=None, quiet=True, library=None):
        self.lib = library if library is not None else default_library()
        c = self.calc = self.lib.calc
        self._bind(c)
        self.config_file_path = str(config_file_path)
        self.n_system = int(n_system)
        with h5lite.open_file(self.config_file_path) as t:
            self.initial_pos = t.read('input/pos', 'f4')[:, :, 0]
        self.n_atom = self.initial_pos.shape[0]
        if device is not None:
            self._check(c.upside_hip_set_device(int(device)), 'set_device')
        self.engine = c.upside_hip_construct(self.n_atom, _b(self.config_file_path), self.n_system, bool(quiet))
        if not self.engine:
            raise RuntimeError('Unable to initialize upside engine: %s' % c.upside_hip_last_error().decode())

    @staticmethod
    def _bind(c):
        if getattr(c, '_ensemble_bound', False):
            return
        vp, i32, u32, u64, f32 = ct.c_void_p, ct.c_int, ct.c_uint32, ct.c_uint64, ct.c_float
        c.upside_hip_set_device.argtypes = [i32]
        c.upside_hip_construct.restype = vp
        c.upside_hip_construct.argtypes = [i32, ct.c_char_p, i32, ct.c_bool]
        for nm in ('set_pos', 'get_pos', 'set_mom', 'get_mom'):
            getattr(c, 'upside_hip_' + nm).argtypes = [vp, vp]
        c.upside_hip_compute.argtypes = [vp, vp, vp]
        c.upside_hip_init_md.argtypes = [vp, vp, u32, f32, f32, i32]
        c.upside_hip_run_md.argtypes = [vp, i32]
        c.upside_hip_run_steps.argtypes = [vp, i32]
        c.upside_hip_recenter.argtypes = [vp]
        c.upside_hip_replica_swap.argtypes = [vp, i32, vp, u32, u64, vp]
        c.upside_replica_decide.argtypes = [i32, vp, vp, vp, u32, u64, i32, vp]
        c.upside_hip_get_system_pos.argtypes = [vp, i32, vp]
        c.upside_hip_set_system_pos.argtypes = [vp, i32, vp]
        c.upside_hip_swap_systems.argtypes = [vp, i32, i32]
        c.upside_hip_swap_system_pairs.argtypes = [vp, i32, vp]
        c.upside_hip_comm_get_unique_id.argtypes = [ct.c_char_p]
        c.upside_hip_comm_init.argtypes = [vp, i32, i32, ct.c_char_p, vp]
        c.upside_hip_comm_replica_swap.argtypes = [vp, i32, vp, u32, u64, i32, vp]
        c.upside_hip_comm_free.argtypes = [vp]
        c.upside_hip_last_error.restype = ct.c_char_p
        c._ensemble_bound = True

    def _check(self, rc, what):
        if rc:
            raise RuntimeError('%s failed: %s' % (what, self.calc.upside_hip_last_error().decode()))

    # -- state ----------------------------------------------------------------------------------
    def set_pos(self, pos):
        pos = np.require(pos, dtype='f4', requirements='C')
        if pos.shape == (self.n_atom, 3):
            pos = np.ascontiguousarray(np.broadcast_to(pos, (self.n_system, self.n_atom, 3)))
        assert pos.shape == (self.n_system, self.n_atom, 3)
        self._check(self.calc.upside_hip_set_pos(self.engine, pos.ctypes.data), 'set_pos')

    def get_pos(self):
        out = np.zeros((self.n_system, self.n_atom, 3), 'f4')
        self._check(self.calc.upside_hip_get_pos(self.engine, out.ctypes.data), 'get_pos')
        return out

    def get_mom(self):
        out = np.zeros((self.n_system, self.n_atom, 3), 'f4')
        self._check(self.calc.upside_hip_get_mom(self.engine, out.ctypes.data), 'get_mom')
        return out

    def get_system_pos(self, system):
        out = np.zeros((self.n_atom, 3), 'f4')
        self._check(self.calc.upside_hip_get_system_pos(self.engine, int(system), out.ctypes.data), 'get_system_pos')
        return out

    def set_system_pos(self, system, pos):
        pos = np.require(pos, dtype='f4', requirements='C')
        assert pos.shape == (self.n_atom, 3)
        self._check(self.calc.upside_hip_set_system_pos(self.engine, int(system), pos.ctypes.data), 'set_system_pos')

    def swap_systems(self, s1, s2):
        self._check(self.calc.upside_hip_swap_systems(self.engine, int(s1), int(s2)), 'swap_systems')

    def swap_system_pairs(self, pairs):
        p = np.ascontiguousarray(np.asarray(pairs, 'i4').reshape(-1, 2))
        if len(p):
            self._check(self.calc.upside_hip_swap_system_pairs(self.engine, int(len(p)), p.ctypes.data), 'swap_system_pairs')

    # -- force pass / MD ------------------------------------------------------------------------
    def energies(self):
        e = np.zeros(self.n_system, 'f4')
        self._check(self.calc.upside_hip_compute(self.engine, e.ctypes.data, None), 'compute')
        return e

    def energies_and_derivs(self):
        e = np.zeros(self.n_system, 'f4'); d = np.zeros((self.n_system, self.n_atom, 3), 'f4')
        self._check(self.calc.upside_hip_compute(self.engine, e.ctypes.data, d.ctypes.data), 'compute')
        return e, d

    def init_md(self, temperature, base_seed, thermostat_timescale=5.0, dt=0.009, thermostat_interval=1):
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(temperature, 'f4'), (self.n_system,)))
        self.temperature = t.copy()
        self._check(self.calc.upside_hip_init_md(self.engine, t.ctypes.data, int(base_seed) & 0xFFFFFFFF,
                                                 float(thermostat_timescale), float(dt), int(thermostat_interval)), 'init_md')

    def run_steps(self, n_step):
        self._check(self.calc.upside_hip_run_steps(self.engine, int(n_step)), 'run_steps')

    def run_rounds(self, n_round):
        self._check(self.calc.upside_hip_run_md(self.engine, int(n_round)), 'run_md')

    # -- replica exchange across the engines of a job, inside the library (comm_rccl.cpp) -------------
    COMM_ID_BYTES = 128

    def comm_unique_id(self):
        """rank 0: the rendezvous token every rank passes to comm_init (hand it around with any host channel)"""
        uid = ct.create_string_buffer(self.COMM_ID_BYTES)
        self._check(self.calc.upside_hip_comm_get_unique_id(uid), 'comm_get_unique_id')
        return uid.raw

    def comm_init(self, rank, world, unique_id, temperature_global):
        """joins this engine (rank `rank` of `world`, equal system counts) to the exchange group; temperature_global:
        the whole ladder, rank r owns entries [r*n_system, (r+1)*n_system)"""
        t = np.ascontiguousarray(np.asarray(temperature_global, 'f4'))
        assert t.shape == (int(world) * self.n_system,)
        uid = ct.create_string_buffer(bytes(unique_id), self.COMM_ID_BYTES)
        self._check(self.calc.upside_hip_comm_init(self.engine, int(rank), int(world), uid, t.ctypes.data), 'comm_init')

    def comm_replica_swap(self, pairs_global, base_seed, round_num, first_set, want_accepted=False):
        """one swap set over GLOBAL replica indices: energies all-gathered over RCCL (first set of an attempt only),
        Metropolis verdicts on the device, coordinates of accepted pairs traded (ncclSend/ncclRecv when they straddle
        ranks).  Everything is enqueued behind the MD steps; want_accepted reads the verdicts back (synchronises)."""
        p = np.ascontiguousarray(np.asarray(pairs_global, 'i4').reshape(-1, 2))
        acc = np.zeros(len(p), 'i4') if want_accepted else None
        self._check(self.calc.upside_hip_comm_replica_swap(self.engine, int(len(p)), p.ctypes.data, int(base_seed) & 0xFFFFFFFF,
                                                           int(round_num), int(bool(first_set)),
                                                           acc.ctypes.data if want_accepted else None), 'comm_replica_swap')
        return acc.astype(bool) if want_accepted else None

    def close(self):
        if getattr(self, 'engine', None):
            self.calc.free_deriv_engine(ct.c_void_p(self.engine))
            self.engine = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def replica_decide(pairs, beta, energy, base_seed, round_num, draw0=0, library=None):
    """Metropolis verdicts of one swap set from known energies (host arithmetic of main.cpp:251-273; needs no GPU).
    Returns (accepted[n_pair] as bool array, next draw index)."""
    lib = library if library is not None else default_library()
    Ensemble._bind(lib.calc)
    pairs = np.require(pairs, dtype='i4', requirements='C').reshape(-1, 2)
    beta = np.require(beta, dtype='f4', requirements='C')
    energy = np.require(energy, dtype='f4', requirements='C')
    if pairs.size and (pairs.min() < 0 or pairs.max() >= len(energy) or len(beta) != len(energy)):
        raise ValueError('swap pairs outside the system list')
    acc = np.zeros(len(pairs) + 1, 'i4')
    if lib.calc.upside_replica_decide(len(pairs), pairs.ctypes.data, beta.ctypes.data, energy.ctypes.data,
                                      int(base_seed) & 0xFFFFFFFF, int(round_num), int(draw0), acc.ctypes.data):
        raise RuntimeError('replica_decide failed')
    return acc[:-1].astype(bool), int(acc[-1])
