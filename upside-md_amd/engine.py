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
