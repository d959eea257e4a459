// Minimal HDF5 (C API) reading helpers for the configuration loader.  Replaces the parts of
// /root/reference/src/h5_support.{h,cpp} the engine needs at construction time (get_dset_size, traverse_dset,
// read_attribute incl. fixed-length string arrays, node_names_in_group).  Errors are thrown as std::string,
// exactly like the reference does (deriv_engine.cpp:100,114,208,...), and converted at the C-ABI boundary.
#pragma once
#include <hdf5.h>
#include <algorithm>
#include <string>
#include <vector>

namespace h5u {

struct Handle {   // RAII for hid_t
    hid_t id; herr_t (*closer)(hid_t);
    Handle(hid_t id_, herr_t (*c)(hid_t)) : id(id_), closer(c) {}
    Handle(const Handle&) = delete;
    Handle(Handle&& o) : id(o.id), closer(o.closer) { o.id = -1; }
    ~Handle() { if (id >= 0 && closer) closer(id); }
    operator hid_t() const { return id; }
};

inline Handle open_group(hid_t loc, const std::string& name) {
    hid_t g = H5Gopen2(loc, name.c_str(), H5P_DEFAULT);
    if (g < 0) throw std::string("unable to open group '") + name + "'";
    return Handle(g, H5Gclose);
}
inline bool exists(hid_t loc, const std::string& name) { return H5Lexists(loc, name.c_str(), H5P_DEFAULT) > 0; }

inline std::vector<hsize_t> dset_size(int ndims, hid_t loc, const std::string& name) {
    hid_t d = H5Dopen2(loc, name.c_str(), H5P_DEFAULT);
    if (d < 0) throw std::string("while getting size of '") + name + "', dataset not found";
    Handle dh(d, H5Dclose);
    Handle sp(H5Dget_space(d), H5Sclose);
    int nd = H5Sget_simple_extent_ndims(sp);
    if (nd != ndims)
        throw std::string("while getting size of '") + name + "', wrong number of dimensions (expected " +
            std::to_string(ndims) + ", but got " + std::to_string(nd) + ")";
    std::vector<hsize_t> dims(ndims > 0 ? ndims : 1, 0);
    H5Sget_simple_extent_dims(sp, dims.data(), NULL);
    dims.resize(ndims);
    return dims;
}

template <typename T> inline hid_t native();
template <> inline hid_t native<float>() { return H5T_NATIVE_FLOAT; }
template <> inline hid_t native<double>() { return H5T_NATIVE_DOUBLE; }
template <> inline hid_t native<int>() { return H5T_NATIVE_INT; }

template <typename T>
inline std::vector<T> read(hid_t loc, const std::string& name, int ndims, std::vector<hsize_t>* dims_out = nullptr) {
    auto dims = dset_size(ndims, loc, name);
    size_t n = 1;
    for (auto d : dims) n *= d;
    std::vector<T> buf(n);
    Handle d(H5Dopen2(loc, name.c_str(), H5P_DEFAULT), H5Dclose);
    if (n && H5Dread(d, native<T>(), H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.data()) < 0)
        throw std::string("unable to read dataset '") + name + "'";
    if (dims_out) *dims_out = dims;
    return buf;
}

inline void check_size(hid_t loc, const std::string& name, std::vector<size_t> expected) {
    auto dims = dset_size((int)expected.size(), loc, name);
    for (size_t i = 0; i < expected.size(); ++i)
        if (dims[i] != expected[i]) throw std::string("dimensions of '") + name + "' do not match the expected size";
}

template <typename T>
inline bool read_attr(T& out, hid_t loc, const std::string& path, const std::string& attr) {
    if (H5Aexists_by_name(loc, path.c_str(), attr.c_str(), H5P_DEFAULT) <= 0) return false;
    Handle a(H5Aopen_by_name(loc, path.c_str(), attr.c_str(), H5P_DEFAULT, H5P_DEFAULT), H5Aclose);
    if (H5Aread(a, native<T>(), &out) < 0) throw std::string("while reading attribute '") + attr + "' of '" + path + "'";
    return true;
}
template <typename T>
inline T attr(hid_t loc, const std::string& path, const std::string& name) {
    T v;
    if (!read_attr(v, loc, path, name)) throw std::string("missing attribute '") + name + "' of '" + path + "'";
    return v;
}
template <typename T>
inline T attr(hid_t loc, const std::string& path, const std::string& name, T dflt) {
    T v;
    return read_attr(v, loc, path, name) ? v : dflt;
}

// fixed-length string array attribute (h5_support.cpp:72-107); variable-length strings are rejected
inline std::vector<std::string> attr_strings(hid_t loc, const std::string& path, const std::string& name) {
    hid_t aid = H5Aopen_by_name(loc, path.c_str(), name.c_str(), H5P_DEFAULT, H5P_DEFAULT);
    if (aid < 0) throw std::string("while reading attribute '") + name + "' of '" + path + "', attribute not found";
    Handle a(aid, H5Aclose);
    Handle sp(H5Aget_space(a), H5Sclose);
    Handle ty(H5Aget_type(a), H5Tclose);
    if (H5Tis_variable_str(ty) > 0) throw std::string("variable-length strings not supported");
    size_t maxchars = H5Tget_size(ty);
    if (H5Sget_simple_extent_ndims(sp) != 1) {
        // an empty numpy array may be stored with zero extent; treat non-1d as "no arguments" only if empty
        if (H5Sget_simple_extent_npoints(sp) == 0) return {};
        throw std::string("wrong size for attribute");
    }
    hsize_t dims[1];
    H5Sget_simple_extent_dims(sp, dims, NULL);
    std::vector<char> tmp(dims[0] * maxchars + 1, '\0');
    if (dims[0] && H5Aread(a, ty, tmp.data()) < 0) throw std::string("unable to read attribute ") + name;
    std::vector<std::string> ret;
    for (hsize_t i = 0; i < dims[0]; ++i) {
        std::string s(tmp.data() + i * maxchars, maxchars);
        while (s.size() && s.back() == '\0') s.pop_back();
        ret.push_back(s);
    }
    return ret;
}

inline std::vector<std::string> node_names_in_group(hid_t loc) {   // h5_support.cpp:277-297 (name order)
    std::vector<std::string> names;
    auto cb = [](hid_t, const char* name, const H5L_info_t*, void* data) -> herr_t {
        static_cast<std::vector<std::string>*>(data)->push_back(name);
        return 0;
    };
    hsize_t idx = 0;
    H5Literate(loc, H5_INDEX_NAME, H5_ITER_INC, &idx, cb, &names);
    std::sort(names.begin(), names.end());
    return names;
}

// 64-bit FNV-1a digest of everything below a group: object paths, attribute names and raw bytes, dataset shapes and raw
// bytes (in the file's own types).  Two configuration files hold the same potential iff their /input/potential digests agree.
struct DigestCtx {
    unsigned long long h = 1469598103934665603ull; hid_t root = -1; std::string error;
    void bytes(const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } }
    void text(const std::string& s) { bytes(s.data(), s.size()); const unsigned char z = 0; bytes(&z, 1); }
};
inline herr_t digest_attribute(hid_t obj, const char* name, const H5A_info_t*, void* data) {
    DigestCtx& c = *static_cast<DigestCtx*>(data);
    c.text(std::string("@") + name);
    Handle a(H5Aopen(obj, name, H5P_DEFAULT), H5Aclose);
    Handle ty(H5Aget_type(a), H5Tclose);
    Handle sp(H5Aget_space(a), H5Sclose);
    const hssize_t n = H5Sget_simple_extent_npoints(sp);
    if (H5Tis_variable_str(ty) > 0) {      // (h5py-style metadata: hashed by content -- the pointers HDF5 hands back are not content)
        std::vector<char*> strs((size_t)(n > 0 ? n : 0), nullptr);
        Handle mt(H5Tcopy(H5T_C_S1), H5Tclose); H5Tset_size(mt, H5T_VARIABLE);
        if (!strs.empty() && H5Aread(a, mt, strs.data()) < 0) { c.error = std::string("unable to read attribute ") + name; return -1; }
        for (char* x : strs) c.text(x ? x : "");
        if (!strs.empty()) H5Dvlen_reclaim(mt, sp, H5P_DEFAULT, strs.data());
        return 0;
    }
    std::vector<unsigned char> buf((size_t)(n > 0 ? n : 0) * H5Tget_size(ty));
    if (!buf.empty() && H5Aread(a, ty, buf.data()) < 0) { c.error = std::string("unable to read attribute ") + name; return -1; }
    c.bytes(buf.data(), buf.size());
    return 0;
}
inline herr_t digest_object(hid_t, const char* name, const H5O_info_t* info, void* data) {
    DigestCtx& c = *static_cast<DigestCtx*>(data);
    c.text(name);
    Handle obj(H5Oopen(c.root, name, H5P_DEFAULT), H5Oclose);
    if (obj < 0) { c.error = std::string("unable to open ") + name; return -1; }
    hsize_t idx = 0;
    if (H5Aiterate2(obj, H5_INDEX_NAME, H5_ITER_INC, &idx, digest_attribute, data) < 0) return -1;
    if (info->type == H5O_TYPE_DATASET) {
        Handle sp(H5Dget_space(obj), H5Sclose);
        Handle ty(H5Dget_type(obj), H5Tclose);
        const int nd = H5Sget_simple_extent_ndims(sp);
        std::vector<hsize_t> dims(nd > 0 ? nd : 1, 0);
        if (nd > 0) H5Sget_simple_extent_dims(sp, dims.data(), NULL);
        c.bytes(dims.data(), sizeof(hsize_t) * (size_t)(nd > 0 ? nd : 0));
        const hssize_t n = H5Sget_simple_extent_npoints(sp);
        if (H5Tdetect_class(ty, H5T_VLEN) > 0 || H5Tis_variable_str(ty) > 0) {
            // variable-length data would be read as pointers: no node of this library reads such a dataset, its shape is its digest
            c.text("<variable-length>");
            return 0;
        }
        std::vector<unsigned char> buf((size_t)(n > 0 ? n : 0) * H5Tget_size(ty));
        if (!buf.empty() && H5Dread(obj, ty, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.data()) < 0) { c.error = std::string("unable to read ") + name; return -1; }
        c.bytes(buf.data(), buf.size());
    }
    return 0;
}
inline unsigned long long group_digest(hid_t group) {
    DigestCtx c; c.root = group;
    if (H5Ovisit(group, H5_INDEX_NAME, H5_ITER_INC, digest_object, &c) < 0) throw std::string("while hashing a group: ") + c.error;
    return c.h;
}

}  // namespace h5u
