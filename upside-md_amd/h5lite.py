"""Minimal HDF5 reader/writer over libhdf5's C API (ctypes).

The image has libhdf5 (HDF5 1.10.6, /opt/conda/lib) but neither h5py nor PyTables, and the
reference's own tooling (py/upside_config.py, Python 2 + PyTables) cannot run here.  This module is
the small subset needed to read the reference's parameter libraries and to write/read `.up`
configuration files in the schema `initialize_engine_from_hdf5` consumes
(/root/reference/src/deriv_engine.cpp:195-270, /root/reference/src/h5_support.cpp:57-107).
"""
import ctypes as C
import os
import numpy as np

_LIB_CANDIDATES = [
    os.environ.get("UPSIDE_HDF5_LIB", ""),
    "/opt/conda/lib/libhdf5.so.103",
    "/opt/conda/lib/libhdf5.so",
    "libhdf5.so.103",
    "libhdf5.so",
    "libhdf5_serial.so",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    err = None
    for p in _LIB_CANDIDATES:
        if not p:
            continue
        try:
            _lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
            break
        except OSError as e:  # pragma: no cover
            err = e
    if _lib is None:
        raise OSError("libhdf5 not found (%s)" % err)
    L = _lib
    hid = C.c_int64
    L.H5open.restype = C.c_int
    L.H5open()
    for name, res, args in [
        ("H5Fopen", hid, [C.c_char_p, C.c_uint, hid]),
        ("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]),
        ("H5Fclose", C.c_int, [hid]),
        ("H5Gcreate2", hid, [hid, C.c_char_p, hid, hid, hid]),
        ("H5Gopen2", hid, [hid, C.c_char_p, hid]),
        ("H5Gclose", C.c_int, [hid]),
        ("H5Oopen", hid, [hid, C.c_char_p, hid]),
        ("H5Oclose", C.c_int, [hid]),
        ("H5Dopen2", hid, [hid, C.c_char_p, hid]),
        ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]),
        ("H5Dread", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
        ("H5Dwrite", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
        ("H5Dclose", C.c_int, [hid]),
        ("H5Dget_space", hid, [hid]),
        ("H5Dget_type", hid, [hid]),
        ("H5Sget_simple_extent_ndims", C.c_int, [hid]),
        ("H5Sget_simple_extent_dims", C.c_int, [hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        ("H5Screate_simple", hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        ("H5Screate", hid, [C.c_int]),
        ("H5Sclose", C.c_int, [hid]),
        ("H5Tcopy", hid, [hid]),
        ("H5Tset_size", C.c_int, [hid, C.c_size_t]),
        ("H5Tget_size", C.c_size_t, [hid]),
        ("H5Tget_class", C.c_int, [hid]),
        ("H5Tis_variable_str", C.c_int, [hid]),
        ("H5Tclose", C.c_int, [hid]),
        ("H5Acreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid]),
        ("H5Awrite", C.c_int, [hid, hid, C.c_void_p]),
        ("H5Aread", C.c_int, [hid, hid, C.c_void_p]),
        ("H5Aopen", hid, [hid, C.c_char_p, hid]),
        ("H5Aexists", C.c_int, [hid, C.c_char_p]),
        ("H5Adelete", C.c_int, [hid, C.c_char_p]),
        ("H5Aget_type", hid, [hid]),
        ("H5Aget_space", hid, [hid]),
        ("H5Aclose", C.c_int, [hid]),
        ("H5Lexists", C.c_int, [hid, C.c_char_p, hid]),
        ("H5Ldelete", C.c_int, [hid, C.c_char_p, hid]),
        ("H5Literate", C.c_int, [hid, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]),
        ("H5Eset_auto2", C.c_int, [hid, C.c_void_p, C.c_void_p]),
        ("H5Oget_info_by_name", C.c_int, [hid, C.c_char_p, C.c_void_p, hid]),
    ]:
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    L.H5Eset_auto2(0, None, None)  # silence the error stack printer; we raise ourselves
    return L


def _g(name):
    return C.c_int64.in_dll(lib(), name).value


H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5S_SCALAR = 0


def _native(dtype):
    dtype = np.dtype(dtype)
    table = {
        np.dtype("f4"): "H5T_NATIVE_FLOAT_g",
        np.dtype("f8"): "H5T_NATIVE_DOUBLE_g",
        np.dtype("i4"): "H5T_NATIVE_INT_g",
        np.dtype("i8"): "H5T_NATIVE_LONG_g",
        np.dtype("u4"): "H5T_NATIVE_UINT_g",
        np.dtype("u8"): "H5T_NATIVE_ULONG_g",
        np.dtype("i1"): "H5T_NATIVE_SCHAR_g",
        np.dtype("u1"): "H5T_NATIVE_UCHAR_g",
        np.dtype("i2"): "H5T_NATIVE_SHORT_g",
        np.dtype("u2"): "H5T_NATIVE_USHORT_g",
    }
    return _g(table[dtype])


def _chk(v, what):
    if v < 0:
        raise IOError("HDF5 failure in %s" % what)
    return v


class Node(object):
    """A group (or file root).  Mirrors the bits of the PyTables API the reference's config writer uses."""

    def __init__(self, hid, owner=None, is_file=False):
        self.hid = hid
        self._owner = owner
        self._is_file = is_file

    # --- navigation -------------------------------------------------------------------------
    def __contains__(self, name):
        L = lib()
        parts = [p for p in name.split("/") if p]
        cur = ""
        for p in parts:
            cur = (cur + "/" + p) if cur else p
            if L.H5Lexists(self.hid, cur.encode(), 0) <= 0:
                return False
        return True

    def group(self, name):
        return Node(_chk(lib().H5Gopen2(self.hid, name.encode(), 0), "H5Gopen2 " + name), self)

    def create_group(self, name):
        return Node(_chk(lib().H5Gcreate2(self.hid, name.encode(), 0, 0, 0), "H5Gcreate2 " + name), self)

    def require_group(self, name):
        return self.group(name) if name in self else self.create_group(name)

    def delete(self, name):
        _chk(lib().H5Ldelete(self.hid, name.encode(), 0), "H5Ldelete")

    def keys(self):
        names = []
        CB = C.CFUNCTYPE(C.c_int, C.c_int64, C.c_char_p, C.c_void_p, C.c_void_p)

        def cb(g, nm, info, data):
            names.append(nm.decode())
            return 0

        idx = C.c_uint64(0)
        fn = CB(cb)
        _chk(lib().H5Literate(self.hid, 0, 0, C.byref(idx), C.cast(fn, C.c_void_p), None), "H5Literate")
        return sorted(names)

    def is_group(self, name):
        L = lib()
        h = L.H5Gopen2(self.hid, name.encode(), 0)
        if h < 0:
            return False
        L.H5Gclose(h)
        return True

    # --- datasets ---------------------------------------------------------------------------
    def shape(self, name):
        L = lib()
        d = _chk(L.H5Dopen2(self.hid, name.encode(), 0), "H5Dopen2 " + name)
        s = L.H5Dget_space(d)
        nd = L.H5Sget_simple_extent_ndims(s)
        dims = (C.c_uint64 * max(nd, 1))()
        L.H5Sget_simple_extent_dims(s, dims, None)
        L.H5Sclose(s)
        L.H5Dclose(d)
        return tuple(int(dims[i]) for i in range(nd))

    def read(self, name, dtype=None):
        L = lib()
        d = _chk(L.H5Dopen2(self.hid, name.encode(), 0), "H5Dopen2 " + name)
        try:
            s = L.H5Dget_space(d)
            nd = L.H5Sget_simple_extent_ndims(s)
            dims = (C.c_uint64 * max(nd, 1))()
            L.H5Sget_simple_extent_dims(s, dims, None)
            L.H5Sclose(s)
            shape = tuple(int(dims[i]) for i in range(nd))
            t = L.H5Dget_type(d)
            cls = L.H5Tget_class(t)
            size = L.H5Tget_size(t)
            if cls == H5T_STRING:
                if L.H5Tis_variable_str(t) > 0:
                    L.H5Tclose(t)
                    raise IOError("variable-length strings unsupported")
                out = np.zeros(shape, dtype="S%d" % size)
                mt = L.H5Tcopy(_g("H5T_C_S1_g"))
                L.H5Tset_size(mt, size)
                _chk(L.H5Dread(d, mt, 0, 0, 0, out.ctypes.data_as(C.c_void_p)), "H5Dread " + name)
                L.H5Tclose(mt)
                L.H5Tclose(t)
                return out
            L.H5Tclose(t)
            if dtype is None:
                dtype = {H5T_INTEGER: {1: "i1", 2: "i2", 4: "i4", 8: "i8"}.get(size, "i8"),
                         H5T_FLOAT: {4: "f4", 8: "f8"}.get(size, "f8")}[cls]
            out = np.zeros(shape, dtype=dtype)
            _chk(L.H5Dread(d, _native(dtype), 0, 0, 0, out.ctypes.data_as(C.c_void_p)), "H5Dread " + name)
            return out
        finally:
            L.H5Dclose(d)

    def write(self, name, arr, dtype=None):
        """create a contiguous dataset holding arr (strings are stored fixed-length)."""
        L = lib()
        arr = np.asarray(arr)
        if arr.dtype.kind in "US":
            arr = np.ascontiguousarray(arr.astype("S"))
            ft = L.H5Tcopy(_g("H5T_C_S1_g"))
            L.H5Tset_size(ft, max(arr.dtype.itemsize, 1))
            close_t = True
        else:
            if dtype is not None:
                arr = arr.astype(dtype)
            elif arr.dtype == np.bool_:
                arr = arr.astype("i1")
            arr = np.ascontiguousarray(arr)
            ft = _native(arr.dtype)
            close_t = False
        dims = (C.c_uint64 * max(arr.ndim, 1))(*arr.shape)
        s = L.H5Screate_simple(arr.ndim, dims, None) if arr.ndim else L.H5Screate(H5S_SCALAR)
        d = _chk(L.H5Dcreate2(self.hid, name.encode(), ft, s, 0, 0, 0), "H5Dcreate2 " + name)
        if arr.size:
            _chk(L.H5Dwrite(d, ft, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)), "H5Dwrite " + name)
        L.H5Dclose(d)
        L.H5Sclose(s)
        if close_t:
            L.H5Tclose(ft)

    # --- attributes -------------------------------------------------------------------------
    def _open_obj(self, obj):
        if obj in (".", ""):
            return self.hid, False
        return _chk(lib().H5Oopen(self.hid, obj.encode(), 0), "H5Oopen " + obj), True

    def set_attr(self, name, value, obj="."):
        L = lib()
        oid, close = self._open_obj(obj)
        try:
            if L.H5Aexists(oid, name.encode()) > 0:        # overwrite = delete + create
                _chk(L.H5Adelete(oid, name.encode()), "H5Adelete " + name)
            if isinstance(value, (list, tuple, np.ndarray)) and len(value) and isinstance(
                    np.asarray(value).flat[0], (str, bytes, np.str_, np.bytes_)):
                arr = np.ascontiguousarray(np.asarray(value).astype("S"))
                t = L.H5Tcopy(_g("H5T_C_S1_g"))
                L.H5Tset_size(t, arr.dtype.itemsize)
                dims = (C.c_uint64 * 1)(arr.shape[0])
                s = L.H5Screate_simple(1, dims, None)
                a = _chk(L.H5Acreate2(oid, name.encode(), t, s, 0, 0), "H5Acreate2 " + name)
                L.H5Awrite(a, t, arr.ctypes.data_as(C.c_void_p))
                L.H5Aclose(a)
                L.H5Sclose(s)
                L.H5Tclose(t)
                return
            if isinstance(value, (str, bytes)):
                b = value.encode() if isinstance(value, str) else value
                t = L.H5Tcopy(_g("H5T_C_S1_g"))
                L.H5Tset_size(t, max(len(b), 1))
                s = L.H5Screate(H5S_SCALAR)
                a = _chk(L.H5Acreate2(oid, name.encode(), t, s, 0, 0), "H5Acreate2 " + name)
                buf = C.create_string_buffer(b, max(len(b), 1))
                L.H5Awrite(a, t, buf)
                L.H5Aclose(a)
                L.H5Sclose(s)
                L.H5Tclose(t)
                return
            arr = np.asarray(value)
            if arr.dtype.kind == "f":
                arr = arr.astype("f8")
            elif arr.dtype.kind in "iub":
                arr = arr.astype("i8")
            arr = np.ascontiguousarray(arr)
            t = _native(arr.dtype)
            if arr.ndim == 0:
                s = L.H5Screate(H5S_SCALAR)
            else:
                dims = (C.c_uint64 * arr.ndim)(*arr.shape)
                s = L.H5Screate_simple(arr.ndim, dims, None)
            a = _chk(L.H5Acreate2(oid, name.encode(), t, s, 0, 0), "H5Acreate2 " + name)
            L.H5Awrite(a, t, arr.ctypes.data_as(C.c_void_p))
            L.H5Aclose(a)
            L.H5Sclose(s)
        finally:
            if close:
                L.H5Oclose(oid)

    def has_attr(self, name, obj="."):
        L = lib()
        oid, close = self._open_obj(obj)
        try:
            return L.H5Aexists(oid, name.encode()) > 0
        finally:
            if close:
                L.H5Oclose(oid)

    def get_attr(self, name, obj="."):
        L = lib()
        oid, close = self._open_obj(obj)
        try:
            a = _chk(L.H5Aopen(oid, name.encode(), 0), "H5Aopen " + name)
            t = L.H5Aget_type(a)
            s = L.H5Aget_space(a)
            nd = L.H5Sget_simple_extent_ndims(s)
            dims = (C.c_uint64 * max(nd, 1))()
            L.H5Sget_simple_extent_dims(s, dims, None)
            shape = tuple(int(dims[i]) for i in range(nd))
            cls = L.H5Tget_class(t)
            size = L.H5Tget_size(t)
            if cls == H5T_STRING:
                out = np.zeros(shape, dtype="S%d" % size)
                mt = L.H5Tcopy(_g("H5T_C_S1_g"))
                L.H5Tset_size(mt, size)
                L.H5Aread(a, mt, out.ctypes.data_as(C.c_void_p))
                L.H5Tclose(mt)
                res = [x.decode() for x in out.ravel()] if nd else out[()].decode()
            else:
                dt = "f8" if cls == H5T_FLOAT else "i8"
                out = np.zeros(shape, dtype=dt)
                L.H5Aread(a, _native(dt), out.ctypes.data_as(C.c_void_p))
                res = out if nd else out[()]
            L.H5Sclose(s)
            L.H5Tclose(t)
            L.H5Aclose(a)
            return res
        finally:
            if close:
                L.H5Oclose(oid)

    # --- lifetime ---------------------------------------------------------------------------
    def close(self):
        if self.hid is not None and self.hid >= 0:
            if self._is_file:
                lib().H5Fclose(self.hid)
            else:
                lib().H5Gclose(self.hid)
            self.hid = -1

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def open_file(path, mode="r"):
    L = lib()
    if mode == "r":
        h = L.H5Fopen(path.encode(), H5F_ACC_RDONLY, 0)
    elif mode in ("a", "r+"):
        h = L.H5Fopen(path.encode(), H5F_ACC_RDWR, 0)
    elif mode == "w":
        h = L.H5Fcreate(path.encode(), H5F_ACC_TRUNC, 0, 0)
    else:
        raise ValueError(mode)
    if h < 0:
        raise IOError("cannot open %s (mode %s)" % (path, mode))
    return Node(h, is_file=True)
