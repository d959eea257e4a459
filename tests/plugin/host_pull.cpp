// Example of node types defined OUTSIDE libupside_hip.so, against the public contract include/upside_hip_plugin.h only.
// Built by tests/plugin/Makefile into libhost_pull.so; loaded by tests/test_gpu_parity.py::test_external_plugin_node
// through upside_hip_load_plugin.  Both nodes do their maths on the host (HostPotentialNode / HostCoordNode: explicit
// device<->host round trip per evaluation), written the way a node of the reference is written
// (/root/reference/src/bonds.cpp:9-50 is the harmonic pull this mirrors).
//
//   host_pull   (1 argument, width >= 3): E = sum_t k_t/2 |x[atom_t] - x0_t|^2      datasets atom (n), x0 (n,3), spring_const (n)
//   host_scale  (1 argument, width 3):    out_i = scale * x_i  (a derived coordinate)  attribute scale on the group
#include "upside_hip_plugin.h"
#include <hdf5.h>

namespace {

template <typename T> hid_t h5type();
template <> hid_t h5type<int>() { return H5T_NATIVE_INT; }
template <> hid_t h5type<float>() { return H5T_NATIVE_FLOAT; }

template <typename T>
std::vector<T> read_dataset(hid_t_compat grp, const char* name, size_t inner) {
    hid_t d = H5Dopen2((hid_t)grp, name, H5P_DEFAULT);
    if (d < 0) throw std::string("host_pull plug-in: missing dataset ") + name;
    hid_t sp = H5Dget_space(d);
    const hssize_t n = H5Sget_simple_extent_npoints(sp);
    std::vector<T> v((size_t)n);
    const herr_t st = n ? H5Dread(d, h5type<T>(), H5S_ALL, H5S_ALL, H5P_DEFAULT, v.data()) : 0;
    H5Sclose(sp); H5Dclose(d);
    if (st < 0 || (size_t)n % inner) throw std::string("host_pull plug-in: bad dataset ") + name;
    return v;
}

struct HostPull : public HostPotentialNode {
    std::vector<int> atom; std::vector<float> x0, k;
    HostPull(DeviceCtx* c, hid_t_compat grp, CoordNode& pos) : HostPotentialNode(c, {&pos}) {
        check_elem_width_lower_bound(pos, 3);
        atom = read_dataset<int>(grp, "atom", 1); x0 = read_dataset<float>(grp, "x0", 3); k = read_dataset<float>(grp, "spring_const", 1);
        if (x0.size() != 3 * atom.size() || k.size() != atom.size()) throw std::string("host_pull: dataset sizes disagree");
        for (int a : atom) if (a < 0 || a >= pos.n_elem) throw std::string("host_pull: atom index out of range");
    }
    float host_potential(int, const std::vector<const float*>& in, const std::vector<float*>& d_in) override {
        const int w = args[0]->elem_width;
        float e = 0.f;
        for (size_t t = 0; t < atom.size(); ++t)
            for (int d = 0; d < 3; ++d) {
                const float dx = in[0][(size_t)atom[t] * w + d] - x0[t * 3 + d];
                e += 0.5f * k[t] * dx * dx;
                d_in[0][(size_t)atom[t] * w + d] += k[t] * dx;
            }
        return e;
    }
};

struct HostScale : public HostCoordNode {
    float scale = 1.f;
    HostScale(DeviceCtx* c, hid_t_compat grp, CoordNode& pos) : HostCoordNode(c, pos.n_elem, 3, {&pos}) {
        check_elem_width(pos, 3);
        hid_t a = H5Aopen((hid_t)grp, "scale", H5P_DEFAULT);
        if (a < 0 || H5Aread(a, H5T_NATIVE_FLOAT, &scale) < 0) throw std::string("host_scale: missing attribute scale");
        H5Aclose(a);
    }
    void host_value(int, const std::vector<const float*>& in, float* out) override {
        for (int i = 0; i < n_elem * 3; ++i) out[i] = scale * in[0][i];
    }
    void host_deriv(int, const std::vector<const float*>&, const float* d_out, const std::vector<float*>& d_in) override {
        for (int i = 0; i < n_elem * 3; ++i) d_in[0][i] += scale * d_out[i];
    }
};

RegisterNodeType<HostPull, 1> host_pull_node("host_pull");
RegisterNodeType<HostScale, 1> host_scale_node("host_scale");

}  // namespace
