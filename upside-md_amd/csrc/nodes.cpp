// DerivComputation nodes of the README force field, hosted on the device.  Each class mirrors the reference
// node of the same registry prefix (cited per class); constructors read the same HDF5 datasets, methods only
// enqueue the kernels of include/upside_hip_kernels.h.
#include "engine.h"
#include "h5util.h"
#include "spline_fit.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <set>

using namespace std;
using namespace h5u;

namespace {

inline hid_t H(hid_t_compat g) { return (hid_t)g; }

// A node type of this file enqueues device work through the launchers of upside_hip_kernels.h only: the engine need not run the
// fused-op queue in front of its methods (include/upside_hip_plugin.h: library_launchers_only)
template <class T> struct Builtin : T {
    template <class... A> Builtin(A&&... a) : T(std::forward<A>(a)...) { this->library_launchers_only = true; }
};

vector<int> iota_targets(int n) { vector<int> v(n); for (int i = 0; i < n; ++i) v[i] = i; return v; }

int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
float env_float(const char* name, float dflt) { const char* e = getenv(name); return e ? (float)atof(e) : dflt; }

void require_injective(const vector<int>& loc, int n_target, const char* what) {
    vector<char> seen(n_target, 0);
    for (int x : loc) {
        if (x < 0 || x >= n_target) throw string(what) + ": index out of range";
        if (seen[x]) throw string(what) + ": repeated index (the device gather path needs distinct elements)";
        seen[x] = 1;
    }
}

// ---------------------------------------------------------------------------------------------------
// bonded springs: bonds.cpp:252-320 (dist_spring), 430-489 (angle_spring), 492-547 (dihedral_spring)
struct SpringNode : public PotentialNode {
    int kind, n_elem;
    CoordNode& pos;
    DevBuf<int> id; DevBuf<float> equil, k;
    int src = -1;
    SpringNode(DeviceCtx* c, hid_t_compat grp, CoordNode& pos_, int kind_) : PotentialNode(c), kind(kind_), pos(pos_) {
        vector<hsize_t> dims;
        auto ids = read<int>(H(grp), "id", 2, &dims);
        n_elem = (int)dims[0];
        if ((int)dims[1] != kind) throw string("wrong width for id");
        check_size(H(grp), "equil_dist", {(size_t)n_elem});
        check_size(H(grp), "spring_const", {(size_t)n_elem});
        if (kind == 2) { check_size(H(grp), "bonded_atoms", {(size_t)n_elem}); bonded_atoms = read<int>(H(grp), "bonded_atoms", 1); }
        for (int x : ids) if (x < 0 || x >= pos.n_elem) throw string("atom index out of range");
        id.upload(ids);
        equil.upload(read<float>(H(grp), "equil_dist", 1));
        k.upload(read<float>(H(grp), "spring_const", 1));
        src = pos.scatter.add_source(n_elem, kind, 3, ids);
        alloc_terms(n_elem);
        fused_forward = fused_backward = true;
    }
    void compute_value(ComputeMode mode) override {
        upk_check(upk_spring(&ctx->L, kind, pos.coord(), id.p, equil.p, k.p, n_elem, pos.scatter.source_ptr(src), pos.scatter.arena_size,
                             mode == PotentialAndDerivMode ? pot_terms.p : nullptr), "spring");
        if (mode == PotentialAndDerivMode) reduce_terms();
    }
    vector<int> bonded_atoms;
    void add_loggers(vector<LogValue>& out) override;   // bonds.cpp:281-295 (dist_spring only)
};
// (SpringNode::add_loggers is defined after sys_slice)
struct DistSpring : SpringNode { DistSpring(DeviceCtx* c, hid_t_compat g, CoordNode& p) : SpringNode(c, g, p, 2) {} };
struct AngleSpring : SpringNode { AngleSpring(DeviceCtx* c, hid_t_compat g, CoordNode& p) : SpringNode(c, g, p, 3) {} };
struct DihedralSpring : SpringNode { DihedralSpring(DeviceCtx* c, hid_t_compat g, CoordNode& p) : SpringNode(c, g, p, 4) {} };
RegisterNodeType<Builtin<DistSpring>, 1> dist_spring_node("dist_spring");
RegisterNodeType<Builtin<AngleSpring>, 1> angle_spring_node("angle_spring");
RegisterNodeType<Builtin<DihedralSpring>, 1> dihedral_spring_node("dihedral_spring");

// cavity_radial: bonds.cpp:323-374 (used to compact synthetic chains)
struct CavityRadial : public PotentialNode {
    int n_term; CoordNode& pos;
    DevBuf<int> id; DevBuf<float> radius, k; int src;
    CavityRadial(DeviceCtx* c, hid_t_compat grp, CoordNode& pos_) : PotentialNode(c), pos(pos_) {
        auto ids = read<int>(H(grp), "id", 1);
        n_term = (int)ids.size();
        check_size(H(grp), "radius", {(size_t)n_term}); check_size(H(grp), "spring_constant", {(size_t)n_term});
        id.upload(ids); radius.upload(read<float>(H(grp), "radius", 1)); k.upload(read<float>(H(grp), "spring_constant", 1));
        src = pos.scatter.add_source(n_term, 1, 3, ids);
        alloc_terms(n_term);
    }
    void compute_value(ComputeMode mode) override {
        upk_check(upk_cavity_radial(&ctx->L, pos.coord(), id.p, radius.p, k.p, n_term, pos.scatter.source_ptr(src), pos.scatter.arena_size,
                                    mode == PotentialAndDerivMode ? pot_terms.p : nullptr), "cavity_radial");
        if (mode == PotentialAndDerivMode) reduce_terms();
    }
};
RegisterNodeType<Builtin<CavityRadial>, 1> cavity_radial_node("cavity_radial");

// ---------------------------------------------------------------------------------------------------
// rama_coord: bonds.cpp:171-249
struct RamaCoord : public CoordNode {
    CoordNode& pos;
    DevBuf<int> atom, dummy; DevBuf<float> jac; int src;
    RamaCoord(DeviceCtx* c, hid_t_compat grp, CoordNode& pos_) : CoordNode(c, (int)dset_size(2, H(grp), "id")[0], 2), pos(pos_) {
        check_size(H(grp), "id", {(size_t)n_elem, 5});
        auto a = read<int>(H(grp), "id", 2);
        vector<int> dm(n_elem * 2);
        for (int i = 0; i < n_elem; ++i) {
            dm[i * 2] = a[i * 5] == -1; dm[i * 2 + 1] = a[i * 5 + 4] == -1;
            if (dm[i * 2]) a[i * 5] = 0;
            if (dm[i * 2 + 1]) a[i * 5 + 4] = 0;
        }
        for (int x : a) if (x < 0 || x >= pos.n_elem) throw string("atom index out of range");
        atom.upload(a); dummy.upload(dm);
        jac.alloc((size_t)c->n_system * n_elem * UPK_RAMA_JAC);
        src = pos.scatter.add_source(n_elem, 5, 3, a);
        fused_forward = fused_backward = true;
    }
    void compute_value(ComputeMode) override { upk_check(upk_rama_fwd(&ctx->L, pos.coord(), atom.p, dummy.p, n_elem, coord(), jac.p), "rama_fwd"); }
    void add_loggers(vector<LogValue>& out) override;   // bonds.cpp:199-202
    void propagate_deriv() override {
        upk_check(upk_rama_bwd(&ctx->L, coord(), jac.p, n_elem, pos.scatter.source_ptr(src), pos.scatter.arena_size), "rama_bwd"); }
};
RegisterNodeType<Builtin<RamaCoord>, 1> rama_coord_node("rama_coord");

// affine_alignment: eig.cpp:277-473
struct AffineAlignment : public CoordNode {
    CoordNode& pos;
    DevBuf<int> atoms; DevBuf<float> ref_geom, eig; int src;
    AffineAlignment(DeviceCtx* c, hid_t_compat grp, CoordNode& pos_) : CoordNode(c, (int)dset_size(2, H(grp), "atoms")[0], 7), pos(pos_) {
        check_size(H(grp), "atoms", {(size_t)n_elem, 3});
        check_size(H(grp), "ref_geom", {(size_t)n_elem, 3, 3});
        auto a = read<int>(H(grp), "atoms", 2);
        for (int x : a) if (x < 0 || x >= pos.n_elem) throw string("atom index out of range");
        atoms.upload(a); ref_geom.upload(read<float>(H(grp), "ref_geom", 3));
        eig.alloc((size_t)c->n_system * n_elem * 20);
        src = pos.scatter.add_source(n_elem, 3, 3, a);
        fused_forward = fused_backward = true;
    }
    void compute_value(ComputeMode) override { upk_check(upk_affine_fwd(&ctx->L, pos.coord(), atoms.p, ref_geom.p, n_elem, coord(), eig.p), "affine_fwd"); }
    void propagate_deriv() override {
        upk_check(upk_affine_bwd(&ctx->L, coord(), ref_geom.p, eig.p, n_elem, pos.scatter.source_ptr(src), pos.scatter.arena_size), "affine_bwd"); }
};
RegisterNodeType<Builtin<AffineAlignment>, 1> affine_alignment_node("affine_alignment");

// infer_H_O: hbond.cpp:14-121
struct Infer_H_O : public CoordNode {
    CoordNode& pos; int n_donor, n_acceptor, n_virtual;
    DevBuf<int> atom; DevBuf<float> bond_length, dfd; int src;
    Infer_H_O(DeviceCtx* c, hid_t_compat grp, CoordNode& pos_)
        : CoordNode(c, (int)(dset_size(2, H(grp), "donors/id")[0] + dset_size(2, H(grp), "acceptors/id")[0]), 6), pos(pos_) {
        n_donor = (int)dset_size(2, H(grp), "donors/id")[0]; n_acceptor = (int)dset_size(2, H(grp), "acceptors/id")[0];
        n_virtual = n_donor + n_acceptor;
        check_size(H(grp), "donors/id", {(size_t)n_donor, 3}); check_size(H(grp), "acceptors/id", {(size_t)n_acceptor, 3});
        check_size(H(grp), "donors/bond_length", {(size_t)n_donor}); check_size(H(grp), "acceptors/bond_length", {(size_t)n_acceptor});
        auto a = read<int>(H(grp), "donors/id", 2); auto a2 = read<int>(H(grp), "acceptors/id", 2);
        a.insert(a.end(), a2.begin(), a2.end());
        auto b = read<float>(H(grp), "donors/bond_length", 1); auto b2 = read<float>(H(grp), "acceptors/bond_length", 1);
        b.insert(b.end(), b2.begin(), b2.end());
        for (int x : a) if (x < 0 || x >= pos.n_elem) throw string("atom index out of range");
        atom.upload(a); bond_length.upload(b);
        dfd.alloc((size_t)c->n_system * n_virtual * 12);
        src = pos.scatter.add_source(n_virtual, 3, 3, a);
        fused_forward = fused_backward = true;
    }
    void compute_value(ComputeMode) override { upk_check(upk_infer_fwd(&ctx->L, pos.coord(), atom.p, bond_length.p, n_virtual, coord(), dfd.p), "infer_fwd"); }
    void propagate_deriv() override {
        upk_check(upk_infer_bwd(&ctx->L, coord(), bond_length.p, dfd.p, n_virtual, pos.scatter.source_ptr(src), pos.scatter.arena_size), "infer_bwd"); }
};
RegisterNodeType<Builtin<Infer_H_O>, 1> infer_node("infer_H_O");

// elements [sys*n, (sys+1)*n) of a per-system device array
template <typename T>
static vector<T> sys_slice(const DevBuf<T>& b, int sys, size_t n) {
    vector<T> v(n);
    if ((size_t)(sys + 1) * n > b.n) throw string("system slice out of range");
    if (n) hip_check(hipMemcpy(v.data(), b.p + (size_t)sys * n, n * sizeof(T), hipMemcpyDeviceToHost), "D2H");
    return v;
}

// rows [0, n_elem) x columns [0, width) of one system's output
static vector<float> output_rows(const CoordNode& n, int sys, int col0, int n_col) {
    auto raw = sys_slice(n.output, sys, (size_t)n.n_elem * n.stride);
    vector<float> v((size_t)n.n_elem * n_col);
    for (int i = 0; i < n.n_elem; ++i) for (int c = 0; c < n_col; ++c) v[(size_t)i * n_col + c] = raw[(size_t)i * n.stride + col0 + c];
    return v;
}
void RamaCoord::add_loggers(vector<LogValue>& out) {
    LogValue l; l.name = "rama"; l.dims = {(size_t)n_elem, 2};
    l.fill = [this](int sys, float* b) { auto v = output_rows(*this, sys, 0, 2); copy(v.begin(), v.end(), b); };
    out.push_back(l);
}
void SpringNode::add_loggers(vector<LogValue>& out) {
    if (kind != 2) return;
    LogValue l; l.name = "nonbonded_spring_energy"; l.dims = {1};   // the frame's energy evaluation filled pot_terms
    l.fill = [this](int sys, float* b) {
        auto t = sys_slice(pot_terms, sys, (size_t)n_elem);
        float pot = 0.f;
        for (int i = 0; i < n_elem; ++i) if (!bonded_atoms[i]) pot += t[i];
        b[0] = pot; };
    out.push_back(l);
}

// get_param_deriv plumbing: a zeroed device table of the get_param() layout, filled by `launch` on the engine stream
template <typename F>
static vector<float> param_deriv_table(DeviceCtx* ctx, size_t n, F launch) {
    DevBuf<float> table(n);
    launch(table.p);
    hip_check(hipStreamSynchronize(ctx->stream), "sync");
    return table.download();
}

// ---------------------------------------------------------------------------------------------------
// placement: placement.cpp:233-325.  signature: 0 scalar, 1 vector, 2 point
struct PlacementNode : public CoordNode {
    CoordNode& alignment; CoordNode* rama;
    upk_placement_t P;
    DevBuf<int> affine_residue, layer, rama_residue; DevBuf<float> fixed_data, spline_coeff, rama_deriv;
    vector<float> host_fixed; int n_layer = 0;
    int src_aff = -1, src_rama = -1;
    bool has_geometry;
    static int sig_dim(const vector<int>& sig) { int d = 0; for (int s : sig) d += s == 0 ? 1 : 3; return d; }
    PlacementNode(DeviceCtx* c, hid_t_compat grp, CoordNode& alignment_, CoordNode* rama_, vector<int> sig)
        : CoordNode(c, (int)dset_size(1, H(grp), "layer_index")[0], sig_dim(sig)), alignment(alignment_), rama(rama_) {
        memset(&P, 0, sizeof(P));
        check_elem_width(alignment, 7);
        P.n_elem = n_elem; P.n_pos_dim = elem_width; P.n_sig = (int)sig.size();
        for (size_t i = 0; i < sig.size(); ++i) P.sig[i] = sig[i];
        has_geometry = false; for (int s : sig) if (s != 0) has_geometry = true;
        check_size(H(grp), "affine_residue", {(size_t)n_elem});
        auto ar = read<int>(H(grp), "affine_residue", 1);
        for (int x : ar) if (x < 0 || x >= alignment.n_elem) throw string("affine_residue out of range");
        auto ly = read<int>(H(grp), "layer_index", 1);
        affine_residue.upload(ar);
        if (rama) {
            check_elem_width(*rama, 2);
            check_size(H(grp), "rama_residue", {(size_t)n_elem});
            auto rr = read<int>(H(grp), "rama_residue", 1);
            for (int x : rr) if (x < 0 || x >= rama->n_elem) throw string("rama_residue out of range");
            rama_residue.upload(rr);
            vector<hsize_t> dims;
            auto raw = read<double>(H(grp), "placement_data", 4, &dims);
            if ((int)dims[3] != elem_width) throw string("placement_data has the wrong width");
            n_layer = (int)dims[0]; P.nx = (int)dims[1]; P.ny = (int)dims[2];
            spline_coeff.upload(fit_layered_periodic_spline2d(raw, n_layer, P.nx, P.ny, elem_width));   // placement.cpp:50-55
            rama_deriv.alloc((size_t)c->n_system * n_elem * 2 * elem_width);
            P.is_rama = 1;
            src_rama = rama->scatter.add_source(n_elem, 1, 2, rr);
        } else {
            vector<hsize_t> dims;
            host_fixed = read<float>(H(grp), "placement_data", 2, &dims);
            if ((int)dims[1] != elem_width) throw string("placement_data has the wrong width");
            n_layer = (int)dims[0];
            fixed_data.upload(host_fixed);
        }
        for (int x : ly) if (x < 0 || x >= n_layer) throw string("layer_index out of range");
        layer.upload(ly);
        if (has_geometry) src_aff = alignment.scatter.add_source(n_elem, 1, 6, ar);
        P.affine_residue = affine_residue.p; P.layer = layer.p; P.rama_residue = rama_residue.p;
        P.fixed_data = fixed_data.p; P.spline_coeff = spline_coeff.p;
        fused_forward = fused_backward = true;
    }
    void compute_value(ComputeMode) override {
        upk_coord_t rc; memset(&rc, 0, sizeof(rc));
        if (rama) rc = rama->coord();
        upk_check(upk_placement_fwd(&ctx->L, &P, alignment.coord(), rc, coord(), rama_deriv.p), "placement_fwd");
    }
    void propagate_deriv() override {
        // a pure scalar placement has no force/torque on the frame; its affine contribution slot is not registered
        static DevBuf<float>* dummy = nullptr;
        float* aff_ptr; long aff_stride;
        if (has_geometry) { aff_ptr = alignment.scatter.source_ptr(src_aff); aff_stride = alignment.scatter.arena_size; }
        else {
            if (!scalar_sink.p) scalar_sink.alloc((size_t)ctx->n_system * n_elem * 6);
            aff_ptr = scalar_sink.p; aff_stride = (long)n_elem * 6;
        }
        (void)dummy;
        upk_check(upk_placement_bwd(&ctx->L, &P, alignment.coord(), coord(), rama_deriv.p, aff_ptr, aff_stride,
                                    rama ? rama->scatter.source_ptr(src_rama) : nullptr, rama ? rama->scatter.arena_size : 0), "placement_bwd");
    }
    DevBuf<float> scalar_sink;
    vector<float> get_param() const override { return host_fixed; }
    vector<float> get_param_deriv(int system) override {   // placement.cpp:95-96 (rama placements: none), :156-160
        if (rama) return vector<float>();
        return param_deriv_table(ctx, host_fixed.size(), [&](float* t) {
            upk_check(upk_placement_param_deriv(&ctx->L, &P, alignment.coord(), coord(), system, t), "placement_param_deriv"); });
    }
    void set_param(const vector<float>& p) override {
        if (rama) return;
        if (p.size() != host_fixed.size()) throw string("wrong param size");
        host_fixed = p;
        hip_check(hipMemcpy(fixed_data.p, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice), "H2D");
    }
};
struct PlScalar : PlacementNode { PlScalar(DeviceCtx* c, hid_t_compat g, CoordNode& a, CoordNode& r) : PlacementNode(c, g, a, &r, {0}) {} };
struct PlFixedScalar : PlacementNode { PlFixedScalar(DeviceCtx* c, hid_t_compat g, CoordNode& a) : PlacementNode(c, g, a, nullptr, {0}) {} };
struct PlPointOnly : PlacementNode { PlPointOnly(DeviceCtx* c, hid_t_compat g, CoordNode& a, CoordNode& r) : PlacementNode(c, g, a, &r, {2}) {} };
struct PlFixedPointOnly : PlacementNode { PlFixedPointOnly(DeviceCtx* c, hid_t_compat g, CoordNode& a) : PlacementNode(c, g, a, nullptr, {2}) {} };
struct PlPointVector : PlacementNode { PlPointVector(DeviceCtx* c, hid_t_compat g, CoordNode& a, CoordNode& r) : PlacementNode(c, g, a, &r, {2, 1}) {} };
struct PlFixedPointVector : PlacementNode { PlFixedPointVector(DeviceCtx* c, hid_t_compat g, CoordNode& a) : PlacementNode(c, g, a, nullptr, {2, 1}) {} };
struct PlFixedPointVectorScalar : PlacementNode { PlFixedPointVectorScalar(DeviceCtx* c, hid_t_compat g, CoordNode& a) : PlacementNode(c, g, a, nullptr, {2, 1, 0}) {} };
// same seven registrations as placement.cpp:319-325
RegisterNodeType<Builtin<PlScalar>, 2> pl1("placement_scalar");
RegisterNodeType<Builtin<PlFixedScalar>, 1> pl2("placement_fixed_scalar");
RegisterNodeType<Builtin<PlPointOnly>, 2> pl3("placement_point_only");
RegisterNodeType<Builtin<PlFixedPointOnly>, 1> pl4("placement_fixed_point_only");
RegisterNodeType<Builtin<PlPointVector>, 2> pl5("placement_point_vector_only");
RegisterNodeType<Builtin<PlFixedPointVector>, 1> pl6("placement_fixed_point_vector_only");
RegisterNodeType<Builtin<PlFixedPointVectorScalar>, 1> pl7("placement_fixed_point_vector_scalar");

// ---------------------------------------------------------------------------------------------------
// rama_map_pot: rama_map_pot.cpp:15-93
struct RamaMapPot : public PotentialNode {
    int n_residue; CoordNode& rama; int nx;
    DevBuf<int> residue, map_id; DevBuf<float> coeff;
    RamaMapPot(DeviceCtx* c, hid_t_compat grp, CoordNode& rama_) : PotentialNode(c), rama(rama_) {
        check_elem_width(rama, 2);
        auto res = read<int>(H(grp), "residue_id", 1);
        n_residue = (int)res.size();
        check_size(H(grp), "rama_map_id", {(size_t)n_residue});
        auto mid = read<int>(H(grp), "rama_map_id", 1);
        vector<hsize_t> dims;
        auto raw = read<double>(H(grp), "rama_pot", 3, &dims);
        if (dims[1] != dims[2]) throw string("must have same x and y grid spacing for Rama maps");
        nx = (int)dims[1];
        require_injective(res, rama.n_elem, "rama_map_pot residue_id");
        for (int x : mid) if (x < 0 || x >= (int)dims[0]) throw string("rama_map_id out of range");
        residue.upload(res); map_id.upload(mid);
        coeff.upload(fit_layered_periodic_spline2d(raw, (int)dims[0], nx, nx, 1));
        alloc_terms(n_residue);
        log_pot = attr<int>(H(grp), ".", "log_pot", 1) != 0;   // rama_map_pot.cpp:34
        fused_forward = fused_backward = true;
    }
    bool log_pot = true;
    void add_loggers(vector<LogValue>& out) override {   // rama_map_pot.cpp:50-54
        if (!log_pot) return;
        LogValue l; l.name = "rama_map_potential"; l.dims = {(size_t)n_residue};
        l.fill = [this](int sys, float* b) { auto t = sys_slice(pot_terms, sys, (size_t)n_residue); copy(t.begin(), t.end(), b); };
        out.push_back(l);
    }
    void compute_value(ComputeMode mode) override {
        upk_check(upk_rama_map_pot(&ctx->L, rama.coord(), residue.p, map_id.p, n_residue, coeff.p, nx,
                                   mode == PotentialAndDerivMode ? pot_terms.p : nullptr), "rama_map_pot");
        if (mode == PotentialAndDerivMode) reduce_terms();
    }
};
RegisterNodeType<Builtin<RamaMapPot>, 1> rama_map_pot_node("rama_map_pot");

// ---------------------------------------------------------------------------------------------------
// backbone_pairs: backbone_steric.cpp:38-147
struct BackbonePairs : public PotentialNode {
    int n_residue; CoordNode& alignment;
    DevBuf<int> id, n_atom; DevBuf<float> ref_pos; float dist_cutoff; int src;
    DevBuf<int> bb_list, bb_cnt; DevBuf<float> bb_ref0, bb_ref1; upk_backbone_list_t cache;    // cached residue-pair lists (n_residue > 128)
    BackbonePairs(DeviceCtx* c, hid_t_compat grp, CoordNode& alignment_) : PotentialNode(c), alignment(alignment_) {
        check_elem_width(alignment, 7);
        auto ids = read<int>(H(grp), "id", 1);
        n_residue = (int)ids.size();
        check_size(H(grp), "n_atom", {(size_t)n_residue}); check_size(H(grp), "ref_pos", {(size_t)n_residue, 4, 3});
        auto na = read<int>(H(grp), "n_atom", 1); auto rp = read<float>(H(grp), "ref_pos", 3);
        for (int x : ids) if (x < 0 || x >= alignment.n_elem) throw string("residue id out of range");
        float max_dev = 0.f;
        for (int nr = 0; nr < n_residue; ++nr) for (int a = 0; a < na[nr]; ++a) {
            const float* p = &rp[(nr * 4 + a) * 3];
            max_dev = max(max_dev, sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]));
        }
        for (auto& x : rp) if (!(x == x)) x = 0.f;   // NaN padding of absent CB atoms is never read (n_atom) but keep it finite
        dist_cutoff = 2 * max_dev + sqrtf(3.f * 3.f + 0.1f * 3.f);
        id.upload(ids); n_atom.upload(na); ref_pos.upload(rp);
        src = alignment.scatter.add_source(n_residue, 1, 6, ids);
        alloc_terms(n_residue);
        fused_forward = fused_backward = true;
        memset(&cache, 0, sizeof(cache));
        if (n_residue > 128 && env_int("UPSIDE_HIP_BACKBONE_LIST", 1)) {
            // residue centres move by ~0.15 A per step: a 3 A skin is rebuilt every ~10 steps and keeps ~45 partners per row at 300 residues
            const size_t S = (size_t)ctx->n_system;
            cache.cap = min(n_residue, env_int("UPSIDE_HIP_BACKBONE_LIST_CAP", 128)); cache.skin = env_float("UPSIDE_HIP_BACKBONE_SKIN", 3.f);
            bb_list.alloc(S * n_residue * cache.cap); bb_cnt.alloc(S * n_residue);
            vector<float> far(S * n_residue * 4, 1e10f);
            bb_ref0.upload(far); bb_ref1.upload(far);
            cache.list = bb_list.p; cache.cnt = bb_cnt.p; cache.ref0 = bb_ref0.p; cache.ref1 = bb_ref1.p; cache.error_flag = ctx->error_flag.p;
        }
    }
    void compute_value(ComputeMode mode) override {
        // the reference centres of the cached list are double buffered by the parity of the force pass (the engine's counter: a graph
        // capture that is rolled back rolls this back with it; a host toggle of its own would be left flipped)
        cache.parity = (int)(ctx->n_pass & 1);
        upk_check(upk_backbone_pairs(&ctx->L, alignment.coord(), id.p, id.p, n_atom.p, ref_pos.p, n_residue, dist_cutoff,
                                     alignment.scatter.source_ptr(src), alignment.scatter.arena_size,
                                     mode == PotentialAndDerivMode ? pot_terms.p : nullptr, cache.list ? &cache : nullptr), "backbone_pairs");
        if (mode == PotentialAndDerivMode) reduce_terms();
    }
};
RegisterNodeType<Builtin<BackbonePairs>, 1> backbone_pairs_node("backbone_pairs");

// ---------------------------------------------------------------------------------------------------
// interaction graph host side: interaction_graph.h:261-398
// One row of the per-interval polynomial table of a quadspline pair potential (igraph_device.h, quadspline_pair<.., POLY>)
// from its spline coefficients p = [angular 1: ka][angular 2: ka][radial wide: k][radial narrow: k] (bead_interaction.h:30-84).
// The cubic B-spline over the window c0..c3 is  a + b y + c y^2 + d y^3  with the coefficients below; computed in double.
// Radial intervals 0 and k-2 are the clamped constants of spline.h:275-310.
static int quadspline_poly_width(int ka, int k) { return 8 * (ka - 3) + 8 * (k - 1); }
// Row STRIDE of a polynomial table in LDS: the width, plus one 16-byte piece of padding whenever the width is an even number of pieces.
// A lane reads the SAME piece position of its pair's row (the angular interval, the radial interval); the 10 A tables are 128 floats
// wide = two whole 256-byte bank rows, so every type pair's rows started in the same bank slot and the 64 gathers of an instruction fell
// into the four or five slots of the intervals in use.  With an odd number of pieces per row, type pair t starts in slot t mod 16.
static int quadspline_poly_stride(int ka, int k) { const int w = quadspline_poly_width(ka, k); return ((w / 4) % 2 == 0) ? w + 4 : w; }
static void quadspline_poly_row(const float* p, int ka, int k, float* out) {
    auto cubic = [](const float* c, float* o) {
        const double c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
        o[0] = (float)((c0 + 4. * c1 + c2) / 6.); o[1] = (float)((c2 - c0) / 2.);
        o[2] = (float)((c0 - 2. * c1 + c2) / 2.); o[3] = (float)((-c0 + 3. * c1 - 3. * c2 + c3) / 6.);
    };
    auto constant = [](const float* c, float* o) {
        o[0] = (float)((1. / 6.) * c[0] + (2. / 3.) * c[1] + (1. / 6.) * c[2]); o[1] = o[2] = o[3] = 0.f;
    };
    for (int a = 0; a < 2; ++a)
        for (int i = 0; i + 3 < ka; ++i) cubic(p + a * ka + i, out + (a * (ka - 3) + i) * 4);
    float* r = out + 8 * (ka - 3);
    for (int w = 0; w < 2; ++w) {
        const float* c = p + 2 * ka + w * k;
        constant(c, r + w * 4);
        for (int i = 1; i <= k - 3; ++i) cubic(c + i - 1, r + i * 8 + w * 4);
        constant(c + k - 3, r + (k - 2) * 8 + w * 4);
    }
}

// (exported for tests/test_abi.py: the table is host arithmetic and can be checked without a GPU)
extern "C" int upside_hip_quadspline_poly_row(const float* spline_coeff, int n_knot_angular, int n_knot, float* poly_out) {
    if (n_knot_angular < 4 || n_knot < 4) return 1;
    quadspline_poly_row(spline_coeff, n_knot_angular, n_knot, poly_out);
    return 0;
}
extern "C" int upside_hip_quadspline_poly_width(int n_knot_angular, int n_knot) { return quadspline_poly_width(n_knot_angular, n_knot); }

struct IGraphHost {
    DeviceCtx* ctx;
    upk_igraph_t G;
    CoordNode *node1, *node2;
    vector<int> loc1, loc2, type1, type2, id1, id2;
    vector<float> param;
    DevBuf<int> d_loc1, d_loc2, d_type1, d_type2, d_id1, d_id2, nbr1, cnt1, nbr2, cnt2, rebuild_flag, flagged;
    DevBuf<float> d_param, d_param_poly, cache_pos1, cache_pos2, cur_pos1, cur_pos2;
    DevBuf<int> hit1, hit2, hcnt1, hcnt2, hlo1;   // this step's in-range pairs per row (upk_pairlist_refine)
    DevBuf<unsigned short> ord1, ord2, ord1u;   // rows sorted by hit count (upk_pairlist_order)
    DevBuf<unsigned long long> gacc;   // upk_igraph_backward's cross-workgroup accumulators (small batches only)
    // Sides whose cached lists the MD path reads (bit 1 / bit 2).  A graph whose pair passes all gather over the rows of one side
    // (coverage graphs) rebuilds that side only; the other side's lists are brought up to date on demand, from the same
    // reference positions, by the few off-path readers (canonical pair list, parameter derivatives): ensure_all_sides().
    int md_sides = 3; bool other_side_stale = false;
    DevBuf<int> all_systems;           // flag list naming every system (ensure_all_sides)
    void use_sides(int sides) {        // called by the owning node once the graph is complete
        const int row_side = sides == 1 ? 1 : 2;
        md_sides = (!G.symmetric && (sides == 1 || sides == 2) && upk_igraph_passes_staged(&ctx->L, &G, row_side)) ? sides : 3;
        // the hit lists of a side that no pass walks are never written: give the memory back (1.2-1.8 MB per system and graph)
        if (!(md_sides & 1)) { hit1.alloc(0); hcnt1.alloc(0); ord1.alloc(0); G.hit1 = nullptr; G.hcnt1 = nullptr; G.ord1 = nullptr; }
        if (!(md_sides & 2)) { hit2.alloc(0); hcnt2.alloc(0); ord2.alloc(0); G.hit2 = nullptr; G.hcnt2 = nullptr; G.ord2 = nullptr; }
    }
    void ensure_all_sides() {
        if (!other_side_stale) return;
        const int S = ctx->n_system;
        if (!all_systems.n) { vector<int> v(S + 1); v[0] = S; for (int i = 0; i < S; ++i) v[1 + i] = i; all_systems.upload(v); }
        upk_igraph_t Gall = G;
        Gall.flagged = all_systems.p; Gall.parity = 0; Gall.flag_stride = S + 1;
        upk_check(upk_pairlist_build_sides(&ctx->L, &Gall, 3 & ~md_sides), "pairlist_build (other side)");
        other_side_stale = false;
    }

    float type_cutoff(const float* p) const {
        switch (G.itype) {
            case UPK_IT_ROTAMER: case UPK_IT_HBOND_COVERAGE: return (float)((G.n_knot - 2 - 1e-6) / G.inv_dx);
            case UPK_IT_RADIAL: case UPK_IT_HBOND_SC_RADIAL: return (float)((16 - 2 - 1e-6) / p[0]);   // sidechain_radial.cpp:31-34
            case UPK_IT_ENVIRONMENT: return p[0] + 1.f / p[1];
            default: return sqrtf(3.5f * 3.5f);
        }
    }
    void update_cutoffs() {   // interaction_graph.h:383-398
        float cutoff = 0.f;
        for (int t1 = 0; t1 < G.n_type1; ++t1) for (int t2 = 0; t2 < G.n_type2; ++t2)
            cutoff = max(cutoff, type_cutoff(&param[(size_t)(t1 * G.n_type2 + t2) * G.n_param]));
        if (G.itype == UPK_IT_ROTAMER)   // is_compatible, bead_interaction.h:209-218
            for (int t1 = 0; t1 < G.n_type1; ++t1) for (int t2 = 0; t2 < G.n_type2; ++t2) {
                const float* p1 = &param[(size_t)(t1 * G.n_type2 + t2) * G.n_param];
                const float* p2 = &param[(size_t)(t2 * G.n_type2 + t1) * G.n_param];
                for (int k = 0; k < G.n_knot_angular; ++k)
                    if (p1[k] != p2[k + G.n_knot_angular] || p1[k + G.n_knot_angular] != p2[k]) throw string("bad angular match");
                for (int k = 0; k < 2 * G.n_knot; ++k)
                    if (p1[2 * G.n_knot_angular + k] != p2[2 * G.n_knot_angular + k]) throw string("incompatible parameters");
            }
        G.cutoff = cutoff;
        G.cache_cutoff = cutoff + skin_scale() * (1.0f + 0.2f * cutoff);
    }

    // Width of the cached-list margin relative to the reference's 1 + 0.2*cutoff (interaction_graph.h:395): any margin
    // gives the same in-range pairs, it only trades rebuild frequency against list length.  Here the residue-pair slots of
    // the side-chain solve are cut from the cached list too, so a long list also makes every belief-propagation sweep
    // longer.  Measured at 1024 x 300 residues, system-steps/s by scale: 1.8: 73 k, 1.4: 79 k, 1.0: 85.6 k, 0.8: 89.2 k,
    // 0.6: 91.2 k, 0.45: 90.6 k (256 systems: 78.0 / 82.4 / 82.2 k at 1.0 / 0.6 / 0.45); with the straight-line list
    // build 0.6: 92.3 k, 0.5: 93.8 k, 0.4: 93.4 k.
    // (round 4: a batch of a few systems is a chain of launches, not work -- a longer list is cheap there and every rebuild lengthens the
    //  chain: 1.5 x the reference's margin up to 16 systems; 56 residues, 1 / 8 systems: 4.97 k / 34.4 k against 4.88 k / 32.8 k system-steps/s
    //  at 0.5, one 300-residue system 2.09 k against 1.99 k; 64 systems: 57 k against 62 k)
    // (round 5: with 16-bit list words and the cheaper refine a longer list costs less: 4096 systems 215.4 k at 0.75 against 212.7 k at 0.5, four
    //  alternating runs each on one box; 1024 and 256 systems no difference, 64 systems 0.5 still ahead by 0.6 %)
    // (the side-chain graph keeps 0.5 there: its cached residue pairs are the solve's slots, and the slot matrices of pairs that are cached
    //  but out of range lie between the active ones -- solve 4.98 -> 5.26 ms, pair energies 1.23 -> 1.35 ms at 0.75; with it at 0.5 and the
    //  others at 0.75: 217.1 against 214.6 k on one box, 220.6 against 219.8 k on another)
    float skin_scale() const {
        static const float v = env_float("UPSIDE_HIP_SKIN_SCALE", 0.f);
        if (v > 0.f) return v;
        if (ctx->n_system <= 16) return 1.5f;
        return (ctx->n_system <= 256 || G.itype == UPK_IT_ROTAMER) ? 0.5f : 0.75f;
    }
    IGraphHost(DeviceCtx* c, hid_t grp, int itype, CoordNode* n1, CoordNode* n2) : ctx(c), node1(n1), node2(n2 ? n2 : n1) {
        memset(&G, 0, sizeof(G));
        G.itype = itype; G.symmetric = itype == UPK_IT_ROTAMER || itype == UPK_IT_RADIAL;
        if (!(G.symmetric ^ bool(n2))) throw string("second node must be null iff symmetric interaction");
        switch (itype) {
            case UPK_IT_ROTAMER: G.dim1 = 6; G.dim2 = 6; break;
            case UPK_IT_HBOND_COVERAGE: G.dim1 = 7; G.dim2 = 6; break;
            case UPK_IT_ENVIRONMENT: G.dim1 = 6; G.dim2 = 4; break;
            case UPK_IT_RADIAL: case UPK_IT_HBOND_SC_RADIAL: G.dim1 = 3; G.dim2 = 3; break;
            default: G.dim1 = 6; G.dim2 = 6; break;
        }
        check_elem_width_lower_bound(*node1, G.dim1);
        check_elem_width_lower_bound(*node2, G.dim2);
        vector<hsize_t> dims;
        param = read<float>(grp, "interaction_param", 3, &dims);
        G.n_type1 = (int)dims[0]; G.n_type2 = (int)dims[1]; G.n_param = (int)dims[2];
        // the reference fixes the knot counts at compile time (bead_interaction.h:12-27); here they follow from
        // the shape of interaction_param: n_param = 2*n_knot_angular + 2*n_knot
        if (itype == UPK_IT_ROTAMER) {
            if (G.n_param == 34) { G.n_knot_angular = 8; G.n_knot = 9; G.inv_dx = 1.f; }
            else if (G.n_param == 40) { G.n_knot_angular = 8; G.n_knot = 12; G.inv_dx = 1.f; }
            else if (G.n_param == 62) { G.n_knot_angular = 15; G.n_knot = 16; G.inv_dx = 2.f; }
            else throw string("unsupported interaction_param width ") + to_string(G.n_param) + " for the rotamer pair interaction";
        } else if (itype == UPK_IT_HBOND_COVERAGE) {
            if (G.n_param == 30) { G.n_knot_angular = 8; G.n_knot = 7; G.inv_dx = 1.f; }
            else if (G.n_param == 40) { G.n_knot_angular = 8; G.n_knot = 12; G.inv_dx = 1.f; }
            else if (G.n_param == 54) { G.n_knot_angular = 15; G.n_knot = 12; G.inv_dx = 2.f; }
            else throw string("unsupported interaction_param width ") + to_string(G.n_param) + " for hbond_coverage";
        } else if (itype == UPK_IT_RADIAL || itype == UPK_IT_HBOND_SC_RADIAL) {
            if (G.n_param != 17) throw string("radial pair potential expects 1 + 16 parameters per type pair");
            if (itype == UPK_IT_RADIAL)      // is_compatible, sidechain_radial.cpp:36-39
                for (int t1 = 0; t1 < G.n_type1; ++t1) for (int t2 = 0; t2 < G.n_type2; ++t2) for (int k = 0; k < 17; ++k)
                    if (param[(size_t)(t1 * G.n_type2 + t2) * 17 + k] != param[(size_t)(t2 * G.n_type2 + t1) * 17 + k]) throw string("incompatible parameters");
        } else if (itype == UPK_IT_ENVIRONMENT) { if (G.n_param != 4) throw string("environment_coverage expects 4 parameters");
        } else if (G.n_param != 8) throw string("protein_hbond expects 8 parameters");
        G.inv_dtheta = (G.n_knot_angular - 3) / 2.f;
        update_cutoffs();

        const string s1 = G.symmetric ? "" : "1";
        loc1 = read<int>(grp, "index" + s1, 1); G.n1 = (int)loc1.size();
        check_size(grp, "type" + s1, {(size_t)G.n1}); check_size(grp, "id" + s1, {(size_t)G.n1});
        type1 = read<int>(grp, "type" + s1, 1); id1 = read<int>(grp, "id" + s1, 1);
        if (!G.symmetric) {
            loc2 = read<int>(grp, "index2", 1); G.n2 = (int)loc2.size();
            check_size(grp, "type2", {(size_t)G.n2}); check_size(grp, "id2", {(size_t)G.n2});
            type2 = read<int>(grp, "type2", 1); id2 = read<int>(grp, "id2", 1);
        } else { loc2 = loc1; type2 = type1; id2 = id1; G.n2 = G.n1; }
        for (int t : type1) if (t < 0 || t >= G.n_type1) throw string("type1 out of range");
        for (int t : type2) if (t < 0 || t >= G.n_type2) throw string("type2 out of range");
        require_injective(loc1, node1->n_elem, "interaction graph index1");
        require_injective(loc2, node2->n_elem, "interaction graph index2");

        const int cap_max = env_int("UPSIDE_HIP_NBR_CAP", 512);
        G.cap1 = min(round_up(G.n2, 4), cap_max);
        G.cap2 = min(round_up(G.n1, 4), cap_max);
        const int S = c->n_system;
        d_loc1.upload(loc1); d_type1.upload(type1); d_id1.upload(id1);
        d_param.upload(param);
        // list words: 16-bit element indices (two per allocated int), except the rotamer graph's bead | slot << 13 (igraph_device.h)
        G.word16 = itype != UPK_IT_ROTAMER;
        if (G.word16 && max(G.n1, G.n2) > 65534) throw string("pair lists hold 16-bit element indices (one value is the sentinel): at most 65534 elements per side");
        auto list_ints = [&](size_t words) { return G.word16 ? (words + 1) / 2 : words; };
        nbr1.alloc(list_ints((size_t)S * G.n1 * G.cap1)); cnt1.alloc((size_t)S * G.n1);
        cache_pos1.upload(vector<float>((size_t)S * G.n1 * 4, 1e10f));   // forces the first rebuild (interaction_graph.h:194-198)
        if (!G.symmetric) {
            d_loc2.upload(loc2); d_type2.upload(type2); d_id2.upload(id2);
            nbr2.alloc(list_ints((size_t)S * G.n2 * G.cap2)); cnt2.alloc((size_t)S * G.n2);
            cache_pos2.upload(vector<float>((size_t)S * G.n2 * 4, 1e10f));
        }
        rebuild_flag.alloc(S);
        G.nbr_j_bits = itype == UPK_IT_ROTAMER ? UPK_ROT_J_BITS : 0;
        if (G.nbr_j_bits && G.n1 >= (1 << G.nbr_j_bits)) throw string("rotamer pair lists pack the bead index into ") + to_string(G.nbr_j_bits) + " bits (one value is the sentinel): at most 8191 beads";
        cur_pos1.alloc((size_t)S * G.n1 * 4);
        if (!G.symmetric) cur_pos2.alloc((size_t)S * G.n2 * 4);
        if (itype != UPK_IT_RADIAL && itype != UPK_IT_HBOND_SC_RADIAL) {   // (the radial potentials walk the cached lists themselves)
            // (+4 words: the packed pair passes fetch a lane's two list words together, the second may lie one past the last row's end)
            hit1.alloc(list_ints((size_t)S * G.n1 * G.cap1 + 4)); hcnt1.alloc((size_t)S * G.n1); ord1.alloc((size_t)S * G.n1);
            if (G.symmetric) { /* (each pair once: the lists hold the partners above the row only) */ }
            else { hit2.alloc(list_ints((size_t)S * G.n2 * G.cap2 + 4)); hcnt2.alloc((size_t)S * G.n2); ord2.alloc((size_t)S * G.n2); }
        }
        if (!G.symmetric && S < 256) { gacc.alloc((size_t)S * max(G.n1, G.n2) * 8); G.gacc = gacc.p; }   // (from 256 systems on a system has one workgroup)
        G.hit1 = hit1.p; G.hit2 = hit2.p; G.hcnt1 = hcnt1.p; G.hcnt2 = hcnt2.p; G.hlo1 = hlo1.p;
        G.ord1 = ord1.p; G.ord2 = ord2.p; G.ord1u = ord1u.p;
        G.cur_pos1 = cur_pos1.p; G.cur_pos2 = G.symmetric ? cur_pos1.p : cur_pos2.p;
        G.loc1 = d_loc1.p; G.type1 = d_type1.p; G.id1 = d_id1.p;
        G.loc2 = G.symmetric ? d_loc1.p : d_loc2.p; G.type2 = G.symmetric ? d_type1.p : d_type2.p; G.id2 = G.symmetric ? d_id1.p : d_id2.p;
        G.param = d_param.p;
        pack_param_poly();
        G.nbr1 = nbr1.p; G.cnt1 = cnt1.p; G.nbr2 = nbr2.p; G.cnt2 = cnt2.p;
        G.cache_pos1 = cache_pos1.p; G.cache_pos2 = G.symmetric ? cache_pos1.p : cache_pos2.p;
        G.rebuild_flag = rebuild_flag.p; G.error_flag = c->error_flag.p;
        flagged.alloc(2 * (size_t)(S + 1)); G.flagged = flagged.p; G.flag_stride = S + 1; G.parity = 0;
        G.node1 = node1->coord(); G.node2 = node2->coord();
        if (!G.symmetric && node1 == node2) {    // the two sides of one node may name the same element (pair kernels: atomic row updates then)
            set<int> side1(loc1.begin(), loc1.end());
            for (int x : loc2) if (side1.count(x)) G.sens_overlap = 1;
        }
    }
    // the polynomial image of the table for the LDS-staged pair passes (hbond_coverage); follows every set_param
    void pack_param_poly() {
        if (G.itype != UPK_IT_HBOND_COVERAGE) return;
        G.n_poly = quadspline_poly_stride(G.n_knot_angular, G.n_knot);
        vector<float> poly((size_t)G.n_type1 * G.n_type2 * G.n_poly, 0.f);
        for (int t = 0; t < G.n_type1 * G.n_type2; ++t)
            quadspline_poly_row(&param[(size_t)t * G.n_param], G.n_knot_angular, G.n_knot, &poly[(size_t)t * G.n_poly]);
        if (d_param_poly.n != poly.size()) { d_param_poly.upload(poly); G.param_poly = d_param_poly.p; }
        else hip_check(hipMemcpy(d_param_poly.p, poly.data(), poly.size() * sizeof(float), hipMemcpyHostToDevice), "H2D");
    }
    // Renumber the elements of a symmetric graph (new element k = old element perm[k]); everything derived from the element
    // arrays afterwards sees the new order only.  The canonical pair list and the edge counts keep reporting the ORIGINAL
    // numbering (orig_of).  Used by the side-chain node, which sorts its beads by (residue-pair-matrix node, rotamer state):
    // with i < j implying node(i) <= node(j) the row bead always owns the FIRST index of the pair matrix, so the 8 lanes of a
    // row group (consecutive partners = the rotamer states of one partner residue) store one contiguous 24-byte record of
    // the matrix instead of six scattered words whenever the partner residue has the lower node id.
    vector<int> orig_of, new_of, type_by_orig;     // new -> original element, original -> new, type by original element
    void permute_elements(const vector<int>& perm) {
        if (!G.symmetric) throw string("element renumbering is for symmetric graphs");
        if ((int)perm.size() != G.n1) throw string("bad permutation");
        type_by_orig = type1; orig_of = perm;
        new_of.assign(perm.size(), 0); for (size_t k = 0; k < perm.size(); ++k) new_of[perm[k]] = (int)k;
        auto apply = [&](vector<int>& v) { vector<int> t(v.size()); for (size_t k = 0; k < v.size(); ++k) t[k] = v[perm[k]]; v.swap(t); };
        apply(loc1); apply(type1); apply(id1);
        loc2 = loc1; type2 = type1; id2 = id1;
        hip_check(hipMemcpy(d_loc1.p, loc1.data(), loc1.size() * sizeof(int), hipMemcpyHostToDevice), "H2D");
        hip_check(hipMemcpy(d_type1.p, type1.data(), type1.size() * sizeof(int), hipMemcpyHostToDevice), "H2D");
        hip_check(hipMemcpy(d_id1.p, id1.data(), id1.size() * sizeof(int), hipMemcpyHostToDevice), "H2D");
    }
    void begin_step() { G.parity ^= 1; }
    // K1 + K2 + K2b/c for the rows of `sides` (bit 1: side 1, bit 2: side 2); the other side's hit lists follow in refine()
    void update_lists(int sides = 3) {
        begin_step();
        upk_check(upk_pairlist_check(&ctx->L, &G), "pairlist_check");
        upk_check(upk_pairlist_build_sides(&ctx->L, &G, md_sides), "pairlist_build");
        other_side_stale = md_sides != 3;
        refine(G, sides);
    }
    // this step's in-range pairs + the row order of the pair passes (G_: the graph, possibly with a substituted source node)
    // (The row order only balances the work of the pair passes -- their results do not depend on it -- and hit counts drift
    //  slowly: it is recomputed every 4th step instead of every step; the first steps always, so that the order
    //  never predates the first list.)
    long n_refine = 0;
    void refine(const upk_igraph_t& G_, int sides = 3) {
        if (!G_.hit1 && !G_.hit2) return;
        static const int order_every = 4;
        const bool reorder = n_refine < 2 || n_refine % order_every == 0;
        ++n_refine;
        for (int side = 1; side <= (G_.symmetric ? 1 : 2); ++side) {
            if (!(sides & side)) continue;
            upk_check(upk_pairlist_refine(&ctx->L, &G_, side), "pairlist_refine");
            if (reorder) upk_check(upk_pairlist_order(&ctx->L, &G_, side), "pairlist_order");
        }
    }
    void set_param(const vector<float>& p) {
        if (p.size() != param.size()) throw string("Bad param size, got ") + to_string(p.size()) + " params, but expected " + to_string(param.size());
        param = p; update_cutoffs();
        hip_check(hipMemcpy(d_param.p, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice), "H2D");
        pack_param_poly();
        // force a rebuild with the new cutoffs: refill the existing reference positions of BOTH sides in place (the buffers
        // keep their addresses: other structs hold copies of them)
        for (DevBuf<float>* b : {&cache_pos1, &cache_pos2})
            if (b->n) hip_check(hipMemsetD32Async((hipDeviceptr_t)b->p, 0x501502f9 /* 1e10f */, b->n, ctx->stream), "fill");
    }
    // Algorithmic bytes: one force evaluation of one graph moves
    //   8*(n1*d1 + n2*d2) + 16*E + 4*n_type1*n_type2*n_param   bytes
    // (coordinates read + gradients written, two indices + value + sensitivity per in-range pair, the table
    // once).  The forward launch accounts for the coordinate read, the index pair and the table; the backward
    // launches share the gradient write and the per-pair value/sensitivity.  E = in-range pairs of system 0.
    double edge_count = -1.;
    double coord_bytes() const { return 4. * (G.n1 * (double)G.dim1 + (G.symmetric ? 0. : G.n2 * (double)G.dim2)); }
    double edges() { if (edge_count < 0.) edge_count = (double)pairlist(0).size(); return edge_count; }
    double bytes_fwd() { return ctx->n_system * (coord_bytes() + 8. * edges() + 4. * G.n_type1 * G.n_type2 * G.n_param); }
    double bytes_bwd(int n_launch) { return ctx->n_system * (coord_bytes() + 8. * edges()) / n_launch; }
    double algorithmic_bytes() { return bytes_fwd() + bytes_bwd(1); }
    struct Prof {   // brackets ONE kernel launch with HIP events when profiling is on
        DeviceCtx* c; std::string nm; double bytes; double pairs = 0.;   // pairs: functor evaluations of the launch (in-range pairs of system 0 x systems; every pass visits a pair once)
        Prof(IGraphHost& ig, const std::string& owner, const char* kind, int bwd_launches) : c(ig.ctx) {
            if (!c->profile) return;
            nm = std::string(kind) + ":" + owner; bytes = bwd_launches ? ig.bytes_bwd(bwd_launches) : ig.bytes_fwd();
            pairs = ig.edges() * c->n_system;
            c->begin(nm);
        }
        ~Prof() { if (c->profile) c->end(nm, bytes, pairs); }
    };
    // canonical in-range pair list of one system (parity/diagnostics)
    vector<pair<int, int>> pairlist(int sys) {
        ensure_all_sides();
        DevBuf<unsigned char> flags((size_t)ctx->n_system * G.n1 * G.cap1);
        upk_check(upk_igraph_inrange(&ctx->L, &G, flags.p), "igraph_inrange");
        hip_check(hipStreamSynchronize(ctx->stream), "sync");
        auto f = flags.download(); auto nb = nbr1.download(); auto ct = cnt1.download();
        vector<pair<int, int>> out;
        for (int i = 0; i < G.n1; ++i)
            for (int k = 0; k < ct[(size_t)sys * G.n1 + i]; ++k) {
                size_t idx = ((size_t)sys * G.n1 + i) * G.cap1 + k;
                const int w = G.word16 ? (int)((const unsigned short*)nb.data())[idx] : nb[idx];
                int j = G.nbr_j_bits ? (w & ((1 << G.nbr_j_bits) - 1)) : w;
                if (!f[idx]) continue;
                if (G.symmetric && j <= i) continue;
                if (orig_of.empty()) out.emplace_back(i, j);
                else out.emplace_back(min(orig_of[i], orig_of[j]), max(orig_of[i], orig_of[j]));     // the caller's numbering
            }
        sort(out.begin(), out.end(), [](const pair<int, int>& a, const pair<int, int>& b) {   // (i1>>2, i2, i1&3)
            if ((a.first >> 2) != (b.first >> 2)) return (a.first >> 2) < (b.first >> 2);
            if (a.second != b.second) return a.second < b.second;
            return (a.first & 3) < (b.first & 3); });
        return out;
    }
    vector<float> count_edges_by_type(int sys) {   // interaction_graph.h:427-441
        vector<float> r((size_t)G.n_type1 * G.n_type2, 0.f);
        const vector<int>& t1 = type_by_orig.empty() ? type1 : type_by_orig;     // (pairlist() reports original element numbers)
        const vector<int>& t2 = type_by_orig.empty() ? type2 : type_by_orig;
        for (auto& e : pairlist(sys)) r[(size_t)t1[e.first] * G.n_type2 + t2[e.second]] += 1.f;
        return r;
    }
};

// protein_hbond: hbond.cpp:290-368
struct ProteinHBond : public CoordNode {
    CoordNode& infer; IGraphHost ig; int n_donor, n_acceptor; DevBuf<float> sens_scaled;
    ProteinHBond(DeviceCtx* c, hid_t_compat grp, CoordNode& infer_)
        : CoordNode(c, (int)(dset_size(1, H(grp), "index1")[0] + dset_size(1, H(grp), "index2")[0]), 7), infer(infer_),
          ig(c, H(grp), UPK_IT_PROTEIN_HBOND, &infer_, &infer_), n_donor(ig.G.n1), n_acceptor(ig.G.n2) {
        // this node mirrors infer_H_O element for element (hbond.cpp:320-323 copies row nv to row nv)
        if (n_donor + n_acceptor != infer.n_elem) throw string("protein_hbond expects one row per infer_H_O site");
        sens_scaled.alloc((size_t)c->n_system * n_elem);
    }
    bool has_prepare() const override { return true; }
    void prepare() override { ig.update_lists(); }
    void compute_value(ComputeMode) override {
        { IGraphHost::Prof pr(ig, name, "igraph_fwd", 0);   // donor rows, then acceptor rows, in one launch
          upk_check(upk_igraph_rows(&ctx->L, &ig.G, 3, 0, output.p, sys_stride(), stride, 6, 0, n_donor, nullptr, 0, nullptr, nullptr, 0, 0), "protein_hbond values"); }
        upk_check(upk_protein_hbond_finish(&ctx->L, infer.coord(), coord()), "protein_hbond_finish");
    }
    void propagate_deriv() override {
        upk_check(upk_protein_hbond_bwd_pre(&ctx->L, coord(), sens_scaled.p), "protein_hbond_bwd_pre");
        { IGraphHost::Prof pr(ig, name, "igraph_bwd", 1);
          upk_check(upk_igraph_rows(&ctx->L, &ig.G, 3, 2, nullptr, 0, 0, 0, 0, 0, nullptr, 3, sens_scaled.p, sens_scaled.p + n_donor, n_elem, 1), "protein_hbond backward"); }
        upk_check(upk_protein_hbond_passthrough(&ctx->L, coord(), infer.coord(), ig.G.loc1, n_donor, ig.G.loc2, n_acceptor), "protein_hbond_passthrough");
    }
    vector<float> get_param() const override { return ig.param; }
    void set_param(const vector<float>& p) override { ig.set_param(p); }
    void add_loggers(vector<LogValue>& out) override {   // hbond.cpp:306-310
        LogValue l; l.name = "hbond"; l.dims = {(size_t)n_elem};
        l.fill = [this](int sys, float* b) { auto v = output_rows(*this, sys, 6, 1); copy(v.begin(), v.end(), b); };
        out.push_back(l);
    }
    // no get_param_deriv: the reference's ProteinHBond does not override it either (hbond.cpp:290-368) -> empty
};
RegisterNodeType<Builtin<ProteinHBond>, 1> hbond_node("protein_hbond");

// hbond_coverage: hbond.cpp:371-414
struct HBondCoverage : public CoordNode {
    IGraphHost ig;
    HBondCoverage(DeviceCtx* c, hid_t_compat grp, CoordNode& infer_, CoordNode& sidechains_)
        : CoordNode(c, (int)dset_size(1, H(grp), "index2")[0], 1), ig(c, H(grp), UPK_IT_HBOND_COVERAGE, &infer_, &sidechains_) {
        // protein_hbond copies the inferred H/O sites through unchanged (hbond.cpp:320-335) and only adds the bond
        // probability: the list upkeep needs positions only, so it can read them from infer_H_O and start before
        // protein_hbond's own pair kernels have run
        if (auto* ph = dynamic_cast<ProteinHBond*>(&infer_)) {
            if (ph->infer.n_elem == infer_.n_elem && ph->infer.stride == infer_.stride) {
                site_positions = &ph->infer;
                prepare_deps.push_back(site_positions); prepare_deps.push_back(&sidechains_);
            }
        }
        ig.use_sides(2);        // both pair passes gather over the bead rows
    }
    const CoordNode* site_positions = nullptr;
    bool has_prepare() const override { return true; }
    void prepare() override {   // lists + this step's hit lists of the bead rows (both passes gather over them)
        if (!site_positions) { ig.update_lists(2); return; }
        ig.begin_step();
        upk_igraph_t Gp = ig.G;
        Gp.node1 = site_positions->coord();
        upk_check(upk_pairlist_check(&ctx->L, &Gp), "pairlist_check");
        upk_check(upk_pairlist_build_sides(&ctx->L, &Gp, ig.md_sides), "pairlist_build");
        ig.other_side_stale = ig.md_sides != 3;
        ig.refine(Gp, 2);
    }
    void compute_value(ComputeMode) override {   // rows = beads
        IGraphHost::Prof pr(ig, name, "igraph_fwd", 0);
        upk_check(upk_igraph_rows(&ctx->L, &ig.G, 2, 0, output.p, sys_stride(), stride, 0, 0, 0, nullptr, 0, nullptr, nullptr, 0, 0), "hbond_coverage values");
    }
    void propagate_deriv() override {   // pair sensitivity = the bead's (hbond.cpp:395-397); beads and sites in one visit per pair
        IGraphHost::Prof pr(ig, name, "igraph_bwd", 1);
        upk_check(upk_igraph_backward(&ctx->L, &ig.G, 2, 2, nullptr, sens.p, sys_stride(), stride), "hbond_coverage backward");
    }
    vector<float> get_param() const override { return ig.param; }
    void set_param(const vector<float>& p) override { ig.set_param(p); }
    vector<float> get_param_deriv(int system) override {   // hbond.cpp:401-402 (this class); pair sensitivity of :395-397
        ig.ensure_all_sides();
        return param_deriv_table(ctx, ig.param.size(), [&](float* t) {
            upk_check(upk_igraph_param_deriv(&ctx->L, &ig.G, system, 2, nullptr, sens.p, sys_stride(), stride, t), "hbond_coverage param_deriv"); });
    }
    vector<float> get_value_by_name(const char* log_name) override {
        if (!strcmp(log_name, "count_edges_by_type")) return ig.count_edges_by_type(0);
        throw string("Value ") + log_name + string(" not implemented");
    }
};
RegisterNodeType<Builtin<HBondCoverage>, 2> coverage_node("hbond_coverage");

// environment_coverage: environment.cpp:71-109
struct EnvironmentCoverage : public CoordNode {
    IGraphHost ig;
    EnvironmentCoverage(DeviceCtx* c, hid_t_compat grp, CoordNode& cb_pos_, CoordNode& weighted_sidechains_)
        : CoordNode(c, (int)dset_size(1, H(grp), "index1")[0], 1), ig(c, H(grp), UPK_IT_ENVIRONMENT, &cb_pos_, &weighted_sidechains_) {
        ig.use_sides(1);        // both pair passes gather over the CB rows
    }
    bool has_prepare() const override { return true; }
    void prepare() override { ig.update_lists(1); }   // rows = CB frames, for both passes
    void compute_value(ComputeMode) override {
        IGraphHost::Prof pr(ig, name, "igraph_fwd", 0);
        upk_check(upk_igraph_rows(&ctx->L, &ig.G, 1, 0, output.p, sys_stride(), stride, 0, 0, 0, nullptr, 0, nullptr, nullptr, 0, 0), "environment_coverage values");
    }
    void propagate_deriv() override {   // pair sensitivity = the CB frame's (environment.cpp:93-101)
        IGraphHost::Prof pr(ig, name, "igraph_bwd", 1);
        upk_check(upk_igraph_backward(&ctx->L, &ig.G, 1, 1, sens.p, nullptr, sys_stride(), stride), "environment_coverage backward");
    }
    vector<float> get_param() const override { return ig.param; }
    void set_param(const vector<float>& p) override { ig.set_param(p); }
    vector<float> get_param_deriv(int system) override {   // environment.cpp:104-105; the functor's derivative is all zeros (:62-65)
        ig.ensure_all_sides();
        return param_deriv_table(ctx, ig.param.size(), [&](float* t) {
            upk_check(upk_igraph_param_deriv(&ctx->L, &ig.G, system, 1, sens.p, nullptr, sys_stride(), stride, t), "environment_coverage param_deriv"); });
    }
};
RegisterNodeType<Builtin<EnvironmentCoverage>, 2> environment_coverage_node("environment_coverage");

// hbond_energy: hbond.cpp:417-456
struct HBondEnergy : public HBondCounter {
    CoordNode& protein_hbond; float E_protein;
    HBondEnergy(DeviceCtx* c, hid_t_compat grp, CoordNode& ph) : HBondCounter(c), protein_hbond(ph), E_protein(attr<float>(H(grp), ".", "protein_hbond_energy")) {
        check_elem_width(ph, 7);
        alloc_terms(ph.n_elem);
        fused_forward = fused_backward = true;
    }
    void compute_value(ComputeMode mode) override {
        upk_check(upk_hbond_energy(&ctx->L, protein_hbond.coord(), E_protein, mode == PotentialAndDerivMode ? pot_terms.p : nullptr), "hbond_energy");
        if (mode == PotentialAndDerivMode) reduce_terms();
    }
    vector<float> get_param() const override { return vector<float>(1, E_protein); }
    vector<float> get_param_deriv(int system) override {   // hbond.cpp:447-448: n_hbond of the last evaluation
        return param_deriv_table(ctx, 1, [&](float* t) { upk_check(upk_column_sum(&ctx->L, protein_hbond.coord(), 6, system, t), "hbond_energy param_deriv"); });
    }
    void set_param(const vector<float>& p) override {
        if (p.size() != 1u) throw string("expected 1 param to hbond_energy but got " + to_string(p.size()));
        E_protein = p[0];
    }
};
RegisterNodeType<Builtin<HBondEnergy>, 1> hbond_energy_node("hbond_energy");

// weighted_pos: environment.cpp:112-156
struct WeightedPos : public CoordNode {
    CoordNode &pos, &energy; DevBuf<int> index_pos, index_weight;
    WeightedPos(DeviceCtx* c, hid_t_compat grp, CoordNode& pos_, CoordNode& energy_)
        : CoordNode(c, (int)dset_size(1, H(grp), "index_pos")[0], 4), pos(pos_), energy(energy_) {
        check_elem_width_lower_bound(pos, 3);
        auto ip = read<int>(H(grp), "index_pos", 1); auto iw = read<int>(H(grp), "index_weight", 1);
        if ((int)iw.size() != n_elem) throw string("index_weight has the wrong size");
        require_injective(ip, pos.n_elem, "weighted_pos index_pos");
        require_injective(iw, energy.n_elem, "weighted_pos index_weight");
        index_pos.upload(ip); index_weight.upload(iw);
        fused_forward = fused_backward = true;
    }
    void compute_value(ComputeMode) override { upk_check(upk_weighted_pos_fwd(&ctx->L, pos.coord(), energy.coord(), index_pos.p, index_weight.p, coord()), "weighted_pos_fwd"); }
    void propagate_deriv() override { upk_check(upk_weighted_pos_bwd(&ctx->L, pos.coord(), energy.coord(), index_pos.p, index_weight.p, coord()), "weighted_pos_bwd"); }
};
RegisterNodeType<Builtin<WeightedPos>, 2> weighted_pos_node("weighted_pos");

// nonlinear_coupling: environment.cpp:324-397
struct NonlinearCoupling : public PotentialNode {
    CoordNode& input; int n_restype, n_coeff; float spline_offset, spline_inv_dx;
    vector<float> coeff; DevBuf<float> d_coeff; DevBuf<int> types;
    NonlinearCoupling(DeviceCtx* c, hid_t_compat grp, CoordNode& input_) : PotentialNode(c), input(input_) {
        check_elem_width(input, 1);
        vector<hsize_t> dims;
        coeff = read<float>(H(grp), "coeff", 2, &dims);
        n_restype = (int)dims[0]; n_coeff = (int)dims[1];
        spline_offset = attr<float>(H(grp), "coeff", "spline_offset"); spline_inv_dx = attr<float>(H(grp), "coeff", "spline_inv_dx");
        check_size(H(grp), "coupling_types", {(size_t)input.n_elem});
        auto t = read<int>(H(grp), "coupling_types", 1);
        for (int i : t) if (i < 0 || i >= n_restype) throw string("invalid coupling type");
        d_coeff.upload(coeff); types.upload(t);
        alloc_terms(input.n_elem);
        fused_forward = fused_backward = true;
    }
    void compute_value(ComputeMode mode) override {
        upk_check(upk_nonlinear_coupling(&ctx->L, input.coord(), types.p, d_coeff.p, n_coeff, spline_offset, spline_inv_dx,
                                         mode == PotentialAndDerivMode ? pot_terms.p : nullptr), "nonlinear_coupling");
        if (mode == PotentialAndDerivMode) reduce_terms();
    }
    void add_loggers(vector<LogValue>& out) override {   // environment.cpp:348-356
        LogValue l; l.name = "nonlinear_coupling"; l.dims = {(size_t)input.n_elem};
        l.fill = [this](int sys, float* b) { auto t = sys_slice(pot_terms, sys, (size_t)input.n_elem); copy(t.begin(), t.end(), b); };
        out.push_back(l);
    }
    vector<float> get_param() const override { return coeff; }
    vector<float> get_param_deriv(int system) override {   // environment.cpp:375-389
        return param_deriv_table(ctx, coeff.size(), [&](float* t) {
            upk_check(upk_nonlinear_coupling_param_deriv(&ctx->L, input.coord(), types.p, n_coeff, spline_offset, spline_inv_dx, system, t), "nonlinear_coupling param_deriv"); });
    }
    void set_param(const vector<float>& p) override {
        if (p.size() != coeff.size()) throw string("attempting to change size of coeff vector on set_param");
        coeff = p; hip_check(hipMemcpy(d_coeff.p, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice), "H2D");
    }
};
RegisterNodeType<Builtin<NonlinearCoupling>, 1> nonlinear_coupling_node("nonlinear_coupling");

// ---------------------------------------------------------------------------------------------------
// Optional restraint / external-field nodes (not emitted for the README force field)

// atom_pos_spring (bonds.cpp:9-50), tension (:53-90), AFM (:93-168), z_flat_bottom (:377-427): one atom per term
struct PointPotential : public PotentialNode {
    int kind, n_term; CoordNode& pos; DevBuf<int> id; DevBuf<float> par; int src;
    float time_initial = 0.f, time_step = 0.f; int round_num = 0;     // AFM only
    PointPotential(DeviceCtx* c, hid_t_compat grp, CoordNode& pos_, int kind_) : PotentialNode(c), kind(kind_), pos(pos_) {
        check_elem_width_lower_bound(pos, 3);
        const char* id_name = kind == 0 ? "id" : "atom";
        auto ids = read<int>(H(grp), id_name, 1);
        n_term = (int)ids.size();
        for (int x : ids) if (x < 0 || x >= pos.n_elem) throw string("atom index out of range");
        vector<float> p((size_t)n_term * 8, 0.f);
        auto col = [&](const char* name, int off) {
            check_size(H(grp), name, {(size_t)n_term});
            auto v = read<float>(H(grp), name, 1);
            for (int i = 0; i < n_term; ++i) p[(size_t)i * 8 + off] = v[i]; };
        auto vec3 = [&](const char* name, int off) {
            check_size(H(grp), name, {(size_t)n_term, 3});
            auto v = read<float>(H(grp), name, 2);
            for (int i = 0; i < n_term; ++i) for (int d = 0; d < 3; ++d) p[(size_t)i * 8 + off + d] = v[(size_t)i * 3 + d]; };
        if (kind == 0) { vec3("x0", 0); col("spring_const", 3); }
        else if (kind == 1) vec3("tension_coeff", 0);
        else if (kind == 2) {
            col("spring_const", 0); vec3("starting_tip_pos", 1); vec3("pulling_vel", 4);
            time_initial = attr<float>(H(grp), "pulling_vel", "time_initial"); time_step = attr<float>(H(grp), "pulling_vel", "time_step");
        } else { col("z0", 0); col("radius", 1); col("spring_constant", 2); }
        id.upload(ids); par.upload(p);
        src = pos.scatter.add_source(n_term, 1, 3, ids);
        alloc_terms(n_term);
    }
    bool capturable() const override { return kind != 2; }   // the AFM tip position travels as a kernel argument
    void add_loggers(vector<LogValue>& out) override {   // bonds.cpp:130-145 (AFM only, basic level)
        if (kind != 2) return;
        LogValue tip; tip.name = "tip_pos"; tip.dims = {(size_t)n_term, 3}; tip.level = 0;
        tip.fill = [this](int, float* b) {
            auto p = par.download(); const float t = time_initial + time_step * round_num;
            for (int i = 0; i < n_term; ++i) for (int d = 0; d < 3; ++d) b[i * 3 + d] = p[(size_t)i * 8 + 1 + d] + p[(size_t)i * 8 + 4 + d] * t; };
        out.push_back(tip);
        LogValue te; te.name = "time_estimate"; te.dims = {1}; te.level = 0;
        te.fill = [this](int, float* b) { b[0] = time_initial + time_step * round_num; };
        out.push_back(te);
    }
    void compute_value(ComputeMode mode) override {
        float time = 0.f;
        if (kind == 2) {                                   // bonds.cpp:150-151: the tip advances on every DerivMode evaluation
            if (mode == DerivMode) round_num += 1;
            time = time_initial + time_step * round_num;
        }
        upk_check(upk_point_potential(&ctx->L, kind, pos.coord(), id.p, par.p, n_term, time, pos.scatter.source_ptr(src), pos.scatter.arena_size,
                                      mode == PotentialAndDerivMode ? pot_terms.p : nullptr), "point_potential");
        if (mode == PotentialAndDerivMode) reduce_terms();
    }
};
struct PosSpring : PointPotential { PosSpring(DeviceCtx* c, hid_t_compat g, CoordNode& p) : PointPotential(c, g, p, 0) {} };
struct TensionPotential : PointPotential { TensionPotential(DeviceCtx* c, hid_t_compat g, CoordNode& p) : PointPotential(c, g, p, 1) {} };
struct AFMPotential : PointPotential { AFMPotential(DeviceCtx* c, hid_t_compat g, CoordNode& p) : PointPotential(c, g, p, 2) {} };
struct ZFlatBottom : PointPotential { ZFlatBottom(DeviceCtx* c, hid_t_compat g, CoordNode& p) : PointPotential(c, g, p, 3) {} };
RegisterNodeType<Builtin<PosSpring>, 1> pos_spring_node("atom_pos_spring");
RegisterNodeType<Builtin<TensionPotential>, 1> tension_node("tension");
RegisterNodeType<Builtin<AFMPotential>, 1> AFM_node("AFM");
RegisterNodeType<Builtin<ZFlatBottom>, 1> z_flat_bottom_node("z_flat_bottom");

// contact: sidechain_radial.cpp:139-205
struct ContactEnergy : public PotentialNode {
    int n_contact; CoordNode& bead_pos; DevBuf<int> id; DevBuf<float> par; int src; vector<int> host_id;
    ContactEnergy(DeviceCtx* c, hid_t_compat grp, CoordNode& bead_pos_) : PotentialNode(c), bead_pos(bead_pos_) {
        check_elem_width_lower_bound(bead_pos, 3);
        vector<hsize_t> dims;
        auto ids = read<int>(H(grp), "id", 2, &dims);
        n_contact = (int)dims[0];
        if ((int)dims[1] != 2) throw string("wrong width for id");
        for (int x : ids) if (x < 0 || x >= bead_pos.n_elem) throw string("contact index out of range");
        check_size(H(grp), "energy", {(size_t)n_contact}); check_size(H(grp), "distance", {(size_t)n_contact}); check_size(H(grp), "width", {(size_t)n_contact});
        auto en = read<float>(H(grp), "energy", 1), dist = read<float>(H(grp), "distance", 1), width = read<float>(H(grp), "width", 1);
        vector<float> p((size_t)n_contact * 4);
        for (int i = 0; i < n_contact; ++i) {
            const float scale = 1.f / width[i];
            p[(size_t)i * 4] = en[i]; p[(size_t)i * 4 + 1] = dist[i]; p[(size_t)i * 4 + 2] = scale; p[(size_t)i * 4 + 3] = dist[i] + 1.f / scale;   // :171
        }
        id.upload(ids); par.upload(p); host_id = ids;
        src = bead_pos.scatter.add_source(n_contact, 2, 3, ids);
        alloc_terms(n_contact);
    }
    void add_loggers(vector<LogValue>& out) override {   // sidechain_radial.cpp:171-183: every contact's energy, half to each of its two beads
        LogValue l; l.name = "contact_energy"; l.dims = {(size_t)bead_pos.n_elem};
        l.fill = [this](int sys, float* b) {
            fill_n(b, bead_pos.n_elem, 0.f);
            auto t = sys_slice(pot_terms, sys, (size_t)n_contact);        // of the frame's energy evaluation (zero beyond the cutoff, as the sigmoid is)
            for (int nc = 0; nc < n_contact; ++nc) { b[host_id[nc * 2]] += 0.5f * t[nc]; b[host_id[nc * 2 + 1]] += 0.5f * t[nc]; }
        };
        out.push_back(l);
    }
    void compute_value(ComputeMode mode) override {
        upk_check(upk_contact(&ctx->L, bead_pos.coord(), id.p, par.p, n_contact, bead_pos.scatter.source_ptr(src), bead_pos.scatter.arena_size,
                              mode == PotentialAndDerivMode ? pot_terms.p : nullptr), "contact");
        if (mode == PotentialAndDerivMode) reduce_terms();
    }
};
RegisterNodeType<Builtin<ContactEnergy>, 1> contact_node("contact");

// radial (symmetric, sidechain_radial.cpp:81-104) and hbond_sc_radial (two nodes, :107-136): the sum over in-range pairs of
// a clamped spline of their distance; every pair has sensitivity 1.  Old-style potentials without shipped parameters:
// they run on the generic per-row kernels (`k_igraph_rowsum` / `k_igraph_grad`), not on the LDS-staged ones.
struct RadialPairs : public PotentialNode {
    IGraphHost ig;
    RadialPairs(DeviceCtx* c, hid_t_compat grp, CoordNode& a, CoordNode* b)
        : PotentialNode(c), ig(c, H(grp), b ? UPK_IT_HBOND_SC_RADIAL : UPK_IT_RADIAL, &a, b) { alloc_terms(ig.G.n1); }
    bool has_prepare() const override { return true; }
    void prepare() override { ig.update_lists(); }
    void compute_value(ComputeMode mode) override {
        upk_check(upk_igraph_grad(&ctx->L, &ig.G, 1, 0, nullptr, nullptr, 0, 0), "radial grad side 1");
        if (!ig.G.symmetric) upk_check(upk_igraph_grad(&ctx->L, &ig.G, 2, 0, nullptr, nullptr, 0, 0), "radial grad side 2");
        if (mode == PotentialAndDerivMode) {
            upk_check(upk_igraph_rowsum(&ctx->L, &ig.G, 1, pot_terms.p, ig.G.n1, 1, 0, 0, nullptr), "radial rowsum");
            reduce_terms();
            if (ig.G.symmetric) upk_check(upk_scale(&ctx->L, potential_dev.p, ctx->n_system, 0.5f), "radial halve");   // a row sum sees each pair from both ends
        }
    }
    vector<float> get_param() const override { return ig.G.symmetric ? vector<float>() : ig.param; }     // sidechain_radial.cpp:131-135: the two-node form only
    void set_param(const vector<float>& p) override { if (!ig.G.symmetric) ig.set_param(p); }
    vector<float> get_param_deriv(int system) override {
        if (ig.G.symmetric) return vector<float>();
        return param_deriv_table(ctx, ig.param.size(), [&](float* t) {
            upk_check(upk_igraph_param_deriv(&ctx->L, &ig.G, system, 0, nullptr, nullptr, 0, 0, t), "hbond_sc_radial param_deriv"); });
    }
};
struct SidechainRadialPairs : RadialPairs { SidechainRadialPairs(DeviceCtx* c, hid_t_compat g, CoordNode& a) : RadialPairs(c, g, a, nullptr) {} };
struct HBondSidechainRadialPairs : RadialPairs { HBondSidechainRadialPairs(DeviceCtx* c, hid_t_compat g, CoordNode& a, CoordNode& b) : RadialPairs(c, g, a, &b) {} };
RegisterNodeType<Builtin<SidechainRadialPairs>, 1> radial_node("radial");
RegisterNodeType<Builtin<HBondSidechainRadialPairs>, 2> hbond_sc_radial_node("hbond_sc_radial");

// constant: bonds.cpp:550-587
struct ConstantCoord : public CoordNode {
    vector<float> value; DevBuf<float> d_value;
    static int dim(hid_t_compat grp, int k) { return (int)dset_size(2, H(grp), "value")[k]; }
    ConstantCoord(DeviceCtx* c, hid_t_compat grp) : CoordNode(c, dim(grp, 0), dim(grp, 1)) {
        value = read<float>(H(grp), "value", 2); d_value.upload(value);
    }
    void compute_value(ComputeMode) override { upk_check(upk_broadcast_rows(&ctx->L, d_value.p, coord()), "constant"); }
    void propagate_deriv() override {}
    vector<float> get_param() const override { return value; }
    void set_param(const vector<float>& p) override {
        if (p.size() != value.size()) throw string("invalid size to set_param");
        value = p; hip_check(hipMemcpy(d_value.p, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice), "H2D");
    }
};
RegisterNodeType<Builtin<ConstantCoord>, 0> constant_coord_node("constant");

// slice: bonds.cpp:589-621
struct Slice : public CoordNode {
    CoordNode& pos; DevBuf<int> id; int src;
    Slice(DeviceCtx* c, hid_t_compat grp, CoordNode& pos_) : CoordNode(c, (int)dset_size(1, H(grp), "id")[0], pos_.elem_width), pos(pos_) {
        auto ids = read<int>(H(grp), "id", 1);
        for (int x : ids) if (x < 0 || x >= pos.n_elem) throw string("slice index out of range");
        id.upload(ids);
        src = pos.scatter.add_source(n_elem, 1, elem_width, ids);
    }
    void compute_value(ComputeMode) override { upk_check(upk_slice_fwd(&ctx->L, pos.coord(), id.p, coord()), "slice_fwd"); }
    void propagate_deriv() override { upk_check(upk_slice_bwd(&ctx->L, coord(), pos.scatter.source_ptr(src), pos.scatter.arena_size), "slice_bwd"); }
};
RegisterNodeType<Builtin<Slice>, 1> slice_node("slice");

// uniform_transform: environment.cpp:158-235
struct UniformTransform : public CoordNode {
    CoordNode& input; int n_coeff; float spline_offset, spline_inv_dx;
    vector<float> coeff; DevBuf<float> d_coeff, jac;
    UniformTransform(DeviceCtx* c, hid_t_compat grp, CoordNode& input_) : CoordNode(c, input_.n_elem, 1), input(input_) {
        check_elem_width(input, 1);
        coeff = read<float>(H(grp), "bspline_coeff", 1); n_coeff = (int)coeff.size();
        spline_offset = attr<float>(H(grp), "bspline_coeff", "spline_offset"); spline_inv_dx = attr<float>(H(grp), "bspline_coeff", "spline_inv_dx");
        d_coeff.upload(coeff); jac.alloc((size_t)c->n_system * n_elem);
    }
    void compute_value(ComputeMode) override {
        upk_check(upk_uniform_transform_fwd(&ctx->L, input.coord(), d_coeff.p, n_coeff, spline_offset, spline_inv_dx, coord(), jac.p), "uniform_transform_fwd"); }
    void propagate_deriv() override { upk_check(upk_uniform_transform_bwd(&ctx->L, input.coord(), coord(), jac.p), "uniform_transform_bwd"); }
    vector<float> get_param() const override {                     // environment.cpp:197-203
        vector<float> r(2 + n_coeff); r[0] = spline_offset; r[1] = spline_inv_dx; copy(coeff.begin(), coeff.end(), r.begin() + 2); return r; }
    vector<float> get_param_deriv(int system) override {           // environment.cpp:205-221
        return param_deriv_table(ctx, 2 + n_coeff, [&](float* t) {
            upk_check(upk_uniform_transform_param_deriv(&ctx->L, input.coord(), d_coeff.p, n_coeff, spline_offset, spline_inv_dx, system, t), "uniform_transform param_deriv"); });
    }
    void set_param(const vector<float>& p) override {              // environment.cpp:223-233
        if (p.size() < size_t(2 + 4)) throw string("too small of size for spline");
        n_coeff = (int)p.size() - 2; spline_offset = p[0]; spline_inv_dx = p[1];
        coeff.assign(p.begin() + 2, p.end()); d_coeff.upload(coeff);
    }
};
RegisterNodeType<Builtin<UniformTransform>, 1> uniform_transform_node("uniform_transform");

// linear_coupling_uniform / linear_coupling_with_inactivation: environment.cpp:237-321
struct LinearCoupling : public PotentialNode {
    CoordNode& input; CoordNode* inactivation; int inactivation_dim = 0;
    vector<float> couplings; DevBuf<float> d_couplings; DevBuf<int> types;
    LinearCoupling(DeviceCtx* c, hid_t_compat grp, CoordNode& input_, CoordNode* inact_) : PotentialNode(c), input(input_), inactivation(inact_) {
        check_elem_width(input, 1);
        if (inactivation) {
            inactivation_dim = attr<int>(H(grp), ".", "inactivation_dim");
            if (input.n_elem != inactivation->n_elem) throw string("Inactivation size must match input size");
            check_elem_width_lower_bound(*inactivation, inactivation_dim + 1);
        }
        couplings = read<float>(H(grp), "couplings", 1);
        check_size(H(grp), "coupling_types", {(size_t)input.n_elem});
        auto t = read<int>(H(grp), "coupling_types", 1);
        for (int i : t) if (i < 0 || i >= (int)couplings.size()) throw string("invalid coupling type");
        d_couplings.upload(couplings); types.upload(t);
        alloc_terms(input.n_elem);
    }
    upk_coord_t inact_coord() const { upk_coord_t z; memset(&z, 0, sizeof(z)); return inactivation ? inactivation->coord() : z; }
    void compute_value(ComputeMode mode) override {
        upk_check(upk_linear_coupling(&ctx->L, input.coord(), types.p, d_couplings.p, inact_coord(), inactivation != nullptr, inactivation_dim,
                                      mode == PotentialAndDerivMode ? pot_terms.p : nullptr), "linear_coupling");
        if (mode == PotentialAndDerivMode) reduce_terms();
    }
    void add_loggers(vector<LogValue>& out) override {   // environment.cpp:271-279: coupling * input, without the inactivation factor
        LogValue l; l.name = inactivation ? "linear_coupling_with_inactivation" : "linear_coupling_uniform"; l.dims = {(size_t)input.n_elem};
        l.fill = [this](int sys, float* b) {
            auto v = output_rows(input, sys, 0, 1); auto t = types.download();
            for (int i = 0; i < input.n_elem; ++i) b[i] = couplings[t[i]] * v[i]; };
        out.push_back(l);
    }
    vector<float> get_param() const override { return couplings; }
    vector<float> get_param_deriv(int system) override {           // environment.cpp:301-312
        return param_deriv_table(ctx, couplings.size(), [&](float* t) {
            upk_check(upk_linear_coupling_param_deriv(&ctx->L, input.coord(), types.p, inact_coord(), inactivation != nullptr, inactivation_dim, system, t), "linear_coupling param_deriv"); });
    }
    void set_param(const vector<float>& p) override {
        if (p.size() != couplings.size()) throw string("attempting to change size of couplings vector on set_param");
        couplings = p; hip_check(hipMemcpy(d_couplings.p, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice), "H2D");
    }
};
struct LinearCouplingUniform : LinearCoupling { LinearCouplingUniform(DeviceCtx* c, hid_t_compat g, CoordNode& in) : LinearCoupling(c, g, in, nullptr) {} };
struct LinearCouplingInactivation : LinearCoupling {
    LinearCouplingInactivation(DeviceCtx* c, hid_t_compat g, CoordNode& in, CoordNode& inact) : LinearCoupling(c, g, in, &inact) {} };
RegisterNodeType<Builtin<LinearCouplingUniform>, 1> linear_coupling_node1("linear_coupling_uniform");
RegisterNodeType<Builtin<LinearCouplingInactivation>, 2> linear_coupling_node2("linear_coupling_with_inactivation");

// membrane_potential: membrane_potential.cpp:13-155
struct MembranePotential : public PotentialNode {
    CoordNode &res_pos, &environment_coverage, &protein_hbond;
    upk_membrane_t M;
    DevBuf<int> cb_index, env_index, restype; DevBuf<float> cov_midpoint, cov_sharpness, cb_coeff, cb_table, uhb_coeff, uhb_table;
    MembranePotential(DeviceCtx* c, hid_t_compat grp, CoordNode& res_pos_, CoordNode& env_, CoordNode& hb_)
        : PotentialNode(c), res_pos(res_pos_), environment_coverage(env_), protein_hbond(hb_) {
        memset(&M, 0, sizeof(M));
        check_elem_width_lower_bound(res_pos, 3); check_elem_width_lower_bound(environment_coverage, 1); check_elem_width(protein_hbond, 7);
        auto ci = read<int>(H(grp), "cb_index", 1);
        M.n_res = (int)ci.size();
        check_size(H(grp), "env_index", {(size_t)M.n_res}); check_size(H(grp), "residue_type", {(size_t)M.n_res});
        auto ei = read<int>(H(grp), "env_index", 1), rt = read<int>(H(grp), "residue_type", 1);
        vector<hsize_t> dims;
        auto cb_e = read<double>(H(grp), "cb_energy", 2, &dims);
        const int n_restype = (int)dims[0]; M.cb_nx = (int)dims[1];
        auto uhb_e = read<double>(H(grp), "uhb_energy", 2, &dims);
        if ((int)dims[0] != 2) throw string("uhb_energy must have 2 rows (unpaired donor, unpaired acceptor)");
        M.uhb_nx = (int)dims[1];
        check_size(H(grp), "cov_midpoint", {(size_t)n_restype}); check_size(H(grp), "cov_sharpness", {(size_t)n_restype});
        const int n_donor = (int)dset_size(1, H(grp), "donor_residue_ids")[0], n_acceptor = (int)dset_size(1, H(grp), "acceptor_residue_ids")[0];
        if (n_donor + n_acceptor != protein_hbond.n_elem) throw string("membrane_potential: donor/acceptor counts do not match protein_hbond");
        M.n_donor = n_donor;
        for (int x : rt) if (x < 0 || x >= n_restype) throw string("residue_type out of range");
        require_injective(ci, res_pos.n_elem, "membrane_potential cb_index");
        require_injective(ei, environment_coverage.n_elem, "membrane_potential env_index");
        M.cb_z_shift = -attr<float>(H(grp), "cb_energy", "z_min");
        M.cb_z_scale = (M.cb_nx - 1) / (attr<float>(H(grp), "cb_energy", "z_max") + M.cb_z_shift);
        M.uhb_z_shift = -attr<float>(H(grp), "uhb_energy", "z_min");
        M.uhb_z_scale = (M.uhb_nx - 1) / (attr<float>(H(grp), "uhb_energy", "z_max") + M.uhb_z_shift);
        cb_index.upload(ci); env_index.upload(ei); restype.upload(rt);
        cov_midpoint.upload(read<float>(H(grp), "cov_midpoint", 1)); cov_sharpness.upload(read<float>(H(grp), "cov_sharpness", 1));
        cb_coeff.upload(fit_layered_clamped_spline1d(cb_e, n_restype, M.cb_nx)); uhb_coeff.upload(fit_layered_clamped_spline1d(uhb_e, 2, M.uhb_nx));
        cb_table.upload(vector<float>(cb_e.begin(), cb_e.end())); uhb_table.upload(vector<float>(uhb_e.begin(), uhb_e.end()));
        M.cb_index = cb_index.p; M.env_index = env_index.p; M.restype = restype.p;
        M.cov_midpoint = cov_midpoint.p; M.cov_sharpness = cov_sharpness.p;
        M.cb_coeff = cb_coeff.p; M.cb_table = cb_table.p; M.uhb_coeff = uhb_coeff.p; M.uhb_table = uhb_table.p;
        alloc_terms(M.n_res + protein_hbond.n_elem);
    }
    void compute_value(ComputeMode mode) override {
        upk_check(upk_membrane(&ctx->L, &M, res_pos.coord(), environment_coverage.coord(), protein_hbond.coord(),
                               mode == PotentialAndDerivMode ? pot_terms.p : nullptr), "membrane_potential");
        if (mode == PotentialAndDerivMode) reduce_terms();
    }
};
RegisterNodeType<Builtin<MembranePotential>, 3> membrane_potential_node("membrane_potential");

// ---------------------------------------------------------------------------------------------------
// rotamer: rotamer.cpp:581-1082
struct RotamerSidechain : public PotentialNode {
    vector<CoordNode*> prob_nodes;
    IGraphHost ig;
    upk_rotamer_t R;
    int n_node, n1, n3, n6;
    vector<int> node_nrot, bead_node, bead_rot;
    DevBuf<long long> bp_trace;
    DevBuf<unsigned char> mark;
    DevBuf<int> bp_bar, bp_fallback; DevBuf<float> bp_nbx, bp_dev, bp_en_part, bead_pack; DevBuf<unsigned long long> grad_acc; DevBuf<float> param_tri, param_tri_poly; DevBuf<int> d_bead_orig;
    bool bp_C_chosen = false;
    DevBuf<int> d_node_nrot, d_bead_node, d_bead_rot, d_nb_start, d_nb_list, n_slot, slot_a, slot_b, slot_of, slot_active, adj_cnt, adj_slot, iters, bp_start, slot_off, class_start, slot_active_last, d_bead_meta, bp_rec, row_start, slot_row, bp_layout;
    DevBuf<float> node_prob, node_off, nb_cur, P, msg_cur, marg, energy;
    DevBuf<const float*> d_prob_out; DevBuf<float*> d_prob_sens; DevBuf<int> d_prob_stride; DevBuf<long> d_prob_sys_stride;
    DevBuf<int> n_bad;

    RotamerSidechain(DeviceCtx* c, hid_t_compat grp, const ArgList& args)
        : PotentialNode(c), prob_nodes(args.begin() + 1, args.end()),
          ig(c, h5u::open_group(H(grp), "pair_interaction"), UPK_IT_ROTAMER, args[0], nullptr) {
        memset(&R, 0, sizeof(R));
        for (size_t i = 0; i < prob_nodes.size(); ++i)
            if (ig.node1->n_elem != prob_nodes[i]->n_elem)
                throw string("rotamer positions have " + to_string(ig.node1->n_elem) + " elements but the " + to_string(i) +
                             "-th (0-indexed) probability node has only " + to_string(prob_nodes[i]->n_elem) + " elements.");
        R.damping = attr<float>(H(grp), ".", "damping"); R.max_iter = attr<int>(H(grp), ".", "max_iter");
        R.tol = attr<float>(H(grp), ".", "tol"); R.chunk = attr<int>(H(grp), ".", "iteration_chunk_size");
        if (R.chunk < 1) throw string("iteration_chunk_size must be positive");
        // calculate_n_elem (rotamer.cpp:559-578) + the id bit-field (rotamer.cpp:812-816)
        const unsigned selector = 15u;
        int n_elem_rot[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int id_ : ig.id1) {
            unsigned id = (unsigned)id_; unsigned rot = id & selector; id >>= 4; unsigned n_rot = id & selector; id >>= 4;
            if (rot >= n_rot) throw string("invalid rotamer number");
            if (n_rot != 1 && n_rot != 3 && n_rot != 6) throw string("invalid rotamer count ") + to_string(n_rot);
            n_elem_rot[n_rot] = max(n_elem_rot[n_rot], (int)id + 1);
        }
        n1 = n_elem_rot[1]; n3 = n_elem_rot[3]; n6 = n_elem_rot[6]; n_node = n1 + n3 + n6;
        // one lane per node in the slot numbering and the LDS bit matrix of residue pairs (kernels_rotamer.hip: 1024 x 1024 bits = 128 KB);
        // UPSIDE_HIP_MAX_ROTAMER_NODES lowers the limit so that the refusal can be tested with the shipped fixtures
        const int max_node = min(1024, env_int("UPSIDE_HIP_MAX_ROTAMER_NODES", 1024));
        if (n_node > max_node)
            throw string("rotamer: ") + to_string(n_node) + " side-chain nodes, but the device belief-propagation solve handles at most " + to_string(max_node) +
                  " (one system per workgroup: slot numbering and pair-matrix bookkeeping live in the 160 KB LDS of a CU)";
        const int start[7] = {0, 0, 0, n1, 0, 0, n1 + n3};
        if (env_int("UPSIDE_HIP_ROT_SORT_BEADS", 1)) {      // beads in (node, rotamer state) order: see IGraphHost::permute_elements
            vector<int> perm(ig.G.n1);
            for (int i = 0; i < ig.G.n1; ++i) perm[i] = i;
            auto key = [&](int i) {
                unsigned id = (unsigned)ig.id1[i]; const unsigned rot = id & selector; id >>= 4; const unsigned n_rot = id & selector; id >>= 4;
                return (long)(start[n_rot] + (int)id) * 16 + (long)rot; };
            stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return key(a) < key(b); });
            ig.permute_elements(perm);
            d_bead_orig.upload(ig.orig_of);
        }
        node_nrot.assign(n_node, 0);
        for (int g = 0; g < n_node; ++g) node_nrot[g] = g < n1 ? 1 : (g < n1 + n3 ? 3 : 6);
        int n_bead = ig.G.n1;
        bead_node.resize(n_bead); bead_rot.resize(n_bead);
        vector<vector<int>> beads_of(n_node * 6);
        for (int i = 0; i < n_bead; ++i) {
            unsigned id = (unsigned)ig.id1[i]; unsigned rot = id & selector; id >>= 4; unsigned n_rot = id & selector; id >>= 4;
            bead_node[i] = start[n_rot] + (int)id; bead_rot[i] = (int)rot;
            beads_of[bead_node[i] * 6 + rot].push_back(i);
        }
        vector<int> nb_start(n_node * 6 + 1, 0), nb_list;
        for (int k = 0; k < n_node * 6; ++k) { nb_start[k] = (int)nb_list.size(); nb_list.insert(nb_list.end(), beads_of[k].begin(), beads_of[k].end()); }
        nb_start[n_node * 6] = (int)nb_list.size();
        one_bead_per_state = true;           // then every pair-matrix entry has a single contributing bead pair
        for (auto& b : beads_of) if (b.size() > 1u) one_bead_per_state = false;
        if (env_int("UPSIDE_HIP_ROTAMER_ATOMIC", 0)) one_bead_per_state = false;   // tests: the general accumulation path
        const int S = c->n_system;
        d_node_nrot.upload(node_nrot); d_bead_node.upload(bead_node); d_bead_rot.upload(bead_rot); d_nb_start.upload(nb_start); d_nb_list.upload(nb_list);
        vector<int> bead_meta(n_bead);
        for (int i = 0; i < n_bead; ++i) {
            if (ig.type1[i] > 255) throw string("more than 256 bead types");
            bead_meta[i] = ig.type1[i] | (bead_rot[i] << 8) | (node_nrot[bead_node[i]] << 12);
        }
        d_bead_meta.upload(bead_meta);
        const float slot_factor = env_float("UPSIDE_HIP_SLOT_FACTOR", 96.f);
        long full = (long)n_node * (n_node - 1) / 2;
        R.slot_cap = (int)max(1L, min(full, (long)(n_node * slot_factor)));
        R.adj_cap = min(max(n_node, 1), env_int("UPSIDE_HIP_ADJ_CAP", 256));
        n_slot.alloc(S); slot_a.alloc((size_t)S * R.slot_cap); slot_b.alloc((size_t)S * R.slot_cap); slot_active.alloc((size_t)S * R.slot_cap);
        slot_of.alloc((size_t)S * n_node * n_node); adj_cnt.alloc((size_t)S * n_node); adj_slot.alloc((size_t)S * n_node * R.adj_cap);
        bp_rec.alloc((size_t)S * R.slot_cap * 4);
        // the dense inbox layout as a launch of its own in front of the one-workgroup solve (kernels_rotamer.hip: k_rotamer_bp_layout) once the
        // solves of a launch run in several rounds over the CUs (a solve owns a whole CU): UPSIDE_HIP_BP_LAYOUT=0/1 forces it off / on (tests)
        {
            const int want = env_int("UPSIDE_HIP_BP_LAYOUT", -1);
            if (want >= 1 || (want < 0 && S >= 512)) { bp_layout.alloc((size_t)S * (UPK_BP_LAYOUT_PER_NODE * n_node + UPK_BP_LAYOUT_EXTRA)); R.bp_layout = bp_layout.p; }
        }
        iters.alloc(S); n_bad.alloc(S); energy.alloc(S); bp_start.alloc((size_t)S * (n_node + 1)); slot_off.alloc((size_t)S * R.slot_cap * 2);
        class_start.alloc((size_t)S * 6); slot_active_last.alloc((size_t)S * R.slot_cap);
        row_start.alloc((size_t)S * (n_node + 2)); slot_row.alloc((size_t)S * R.slot_cap * 2);
        ig.G.mark_ld = ((n_node + 63) / 64) * 64;
        ig.G.mark_stride = n_node * ig.G.mark_ld;
        mark.alloc((size_t)S * ig.G.mark_stride);
        ig.G.mark_table = mark.p; ig.G.mark_node = d_bead_node.p; ig.G.mark_n = n_node;
        ig.G.mark_start3 = n1; ig.G.mark_start6 = n1 + n3;
        node_prob.alloc((size_t)S * n_node * 6); node_off.alloc((size_t)S * n_node); nb_cur.alloc((size_t)S * n_node * 6);
        P.alloc((size_t)S * R.slot_cap * 36); marg.alloc((size_t)S * R.slot_cap * 36);
        msg_cur.alloc((size_t)S * R.slot_cap * 16);
    }
    void finalize() override {
        vector<const float*> po; vector<float*> ps; vector<int> st; vector<long> ss;
        for (auto* p : prob_nodes) { po.push_back(p->output.p); ps.push_back(p->sens.p); st.push_back(p->stride); ss.push_back(p->sys_stride()); }
        if (po.empty()) { po.push_back(nullptr); ps.push_back(nullptr); st.push_back(0); ss.push_back(0); }
        d_prob_out.upload(po); d_prob_sens.upload(ps); d_prob_stride.upload(st); d_prob_sys_stride.upload(ss);
        fill_struct();
    }
    bool one_bead_per_state = false;
    void pack_param_tri() {   // upper triangle of the pair table, the form the bead-pair kernels stage in LDS
        const int nt = ig.G.n_type1, np = ig.G.n_param;
        vector<float> tri; tri.reserve((size_t)nt * (nt + 1) / 2 * np);
        for (int lo = 0; lo < nt; ++lo) for (int hi = lo; hi < nt; ++hi)
            tri.insert(tri.end(), ig.param.begin() + (size_t)(lo * nt + hi) * np, ig.param.begin() + (size_t)(lo * nt + hi + 1) * np);
        param_tri.upload(tri); R.param_tri = param_tri.p;
        // ... and as per-interval polynomials for the energy pass (quadspline_poly_row)
        const int npoly = quadspline_poly_stride(ig.G.n_knot_angular, ig.G.n_knot);
        vector<float> poly((size_t)nt * (nt + 1) / 2 * npoly, 0.f);
        for (size_t r = 0; r < (size_t)nt * (nt + 1) / 2; ++r) quadspline_poly_row(&tri[r * np], ig.G.n_knot_angular, ig.G.n_knot, &poly[r * npoly]);
        param_tri_poly.upload(poly); R.param_tri_poly = param_tri_poly.p; R.n_poly = npoly;
    }
    void fill_struct() {
        R.G = ig.G;
        pack_param_tri();
        R.one_bead_per_state = one_bead_per_state ? 1 : 0;
        R.n_node = n_node; R.n_node1 = n1; R.n_node3 = n3;
        R.node_nrot = d_node_nrot.p; R.bead_node = d_bead_node.p; R.bead_rot = d_bead_rot.p; R.bead_orig = d_bead_orig.p;
        R.node_bead_start = d_nb_start.p; R.node_bead_list = d_nb_list.p;
        R.n_prob = (int)prob_nodes.size(); R.prob_out = d_prob_out.p; R.prob_sens = d_prob_sens.p; R.prob_stride = d_prob_stride.p;
        R.prob_sys_stride = d_prob_sys_stride.p;
        R.node_prob = node_prob.p; R.node_off = node_off.p; R.nb_cur = nb_cur.p;
        R.n_slot = n_slot.p; R.slot_a = slot_a.p; R.slot_b = slot_b.p; R.slot_of = slot_of.p; R.slot_active = slot_active.p; R.mark = mark.p;
        R.adj_cnt = adj_cnt.p; R.adj_slot = adj_slot.p; R.bp_start = bp_start.p; R.slot_off = slot_off.p;
        R.class_start = class_start.p; R.slot_active_last = slot_active_last.p; R.bead_meta = d_bead_meta.p;
        R.row_start = row_start.p; R.slot_row = slot_row.p;
        R.P = P.p; R.msg_cur = msg_cur.p; R.marg = marg.p;
        R.iters = iters.p; R.n_bad = n_bad.p; R.bp_rec = bp_rec.p; R.energy = energy.p;
        { const size_t nS = ctx->n_system; bp_bar.alloc(nS); bp_fallback.alloc(nS); bp_nbx.alloc(nS * 2 * n_node * 8); bp_dev.alloc(nS * 32); bp_en_part.alloc(nS * 16); }
        R.bp_bar = bp_bar.p; R.bp_fallback = bp_fallback.p; R.bp_nbx = bp_nbx.p; R.bp_dev = bp_dev.p; R.bp_en_part = bp_en_part.p;
        if (R.slot_cap >= UPK_ROT_SLOT_NONE) throw string("rotamer pair lists pack the residue-pair slot into 19 bits: lower UPSIDE_HIP_SLOT_FACTOR");
        R.bp_C = 1;
        R.bead_pack = nullptr;   // packed global bead rows: only when table + beads exceed the LDS budget of the pair kernels
        grad_acc.alloc((size_t)ctx->n_system * ig.G.n1 * 6); R.grad_acc = grad_acc.p;
        if (((size_t)(ig.G.n_type1 * (ig.G.n_type1 + 1) / 2) * ig.G.n_param + (size_t)ig.G.n1 * 22 + 8) * sizeof(float) > 158 * 1024 ||   // triangle table + beads + accumulators + row order
            env_int("UPSIDE_HIP_ROT_UNSTAGED", 0)) {
            bead_pack.alloc((size_t)ctx->n_system * ig.G.n1 * 8); R.bead_pack = bead_pack.p;
        }
        R.bp_trace = nullptr;
        prepare_deps.push_back(ig.node1);   // the list upkeep reads the bead positions only, not the 1-body energies
        if (getenv("UPSIDE_HIP_BP_TRACE")) { bp_trace.alloc((size_t)ctx->n_system * 32); R.bp_trace = bp_trace.p; }
    }
    bool has_prepare() const override { return true; }
    void prepare() override {   // pair list + residue-pair slots of the systems that moved (depends on the bead positions only)
        ig.begin_step();
        R.G = ig.G;
        upk_check(upk_pairlist_check(&ctx->L, &ig.G), "pairlist_check");
        upk_check(upk_pairlist_build(&ctx->L, &ig.G), "pairlist_build");
        upk_check(upk_rotamer_build_slots(&ctx->L, &R), "rotamer_build_slots");
        upk_check(upk_rotamer_nbr_slots(&ctx->L, &R), "rotamer_nbr_slots");
        ig.refine(ig.G);
    }
    // workgroups per system of the belief-propagation solve: enough that the exp(-E) matrices of the multi-state
    // residue pairs fit their LDS (15% slack for the fluctuation of the pair count; systems that outgrow it fall
    // back to the one-workgroup kernel on the device).  Decided once, from the first pair list.
    void choose_bp_cluster() {
        bp_C_chosen = true;
        ctx->flush();
        const int want = env_int("UPSIDE_HIP_BP_CLUSTER", -1);   // 1 disables, >1 forces
        R.bp_resident = 1;
        if (want == 1) { R.bp_C = 1; set_matrix_form(); return; }
        hip_check(hipStreamSynchronize(ctx->stream), "sync");
        auto cs = class_start.download();
        long need = 0, widest = 0;
        for (int s = 0; s < ctx->n_system; ++s) {
            const int* c = &cs[(size_t)s * 6];
            need = max(need, 9L * (c[1] - c[0]) + 18L * (c[2] - c[1]) + 36L * (c[3] - c[2]));
            widest = max(widest, (long)max(c[1] - c[0], max(c[2] - c[1], c[3] - c[2])));
        }
        const int cap = upk_rotamer_bp_cluster_capacity(&R), lanes = upk_rotamer_bp_cluster_threads();
        int C = cap > 0 ? (int)((need * 115 / 100 + cap - 1) / cap) : 1;
        C = max(C, (int)((widest * 115 / 100 + lanes - 1) / lanes));   // one slot per class per lane
        if (want > 1) C = want;
        if (C > 16 || n_node - R.n_node1 < C) C = 1;             // too large for a co-resident cluster: one-workgroup solve
        R.bp_resident = 1;
        // The cluster solve trades HBM/L2 traffic for two device-scope barriers per sweep.  Re-measured after the
        // one-workgroup kernel got its batched loads and the short list margin (BP ms, cluster vs one workgroup):
        //   300 residues / 10 A (C = 8):  1 system 0.19 vs 0.31, 8: 0.23 vs 0.35, 32: 0.35 vs 0.43, 48 (two launches or the
        //                                 split form): 0.63 vs 0.47, 64: 0.56 vs 0.48
        //   300 residues / 7 A, 150 residues / 10 A (C = 3..4): 0.22 vs 0.19 at 1 system, 0.31 vs 0.24 at 32
        //   56 and 20 residues (C = 1): the one-workgroup solve by construction
        // so the resident cluster is used when the pair matrices need many workgroups (C >= 6) and the batch fits one
        // launch (CUs / C systems) or a quarter more; everything else takes the one-workgroup solve.  The split form
        // (cluster over global-memory matrices) no longer wins anywhere; UPSIDE_HIP_BP_SPLIT keeps it testable.
        const int n_cu = upk_device_cu_count();
        int per_launch = C > 1 ? n_cu / C : 0;
        if (per_launch >= 8) per_launch &= ~7;
        // (round 2, after the one-workgroup solve got its look-ahead loads: 32 systems 0.37 vs 0.38 ms, 40: 0.41 vs 0.39, 48 (two launches):
        //  0.61 vs 0.39 -- the cluster up to 4/5 of one launch)
        const int resident_limit = C >= 6 ? per_launch * 4 / 5 : 0;
        if (want <= 1 && ctx->n_system > resident_limit) C = 1;
        if (env_int("UPSIDE_HIP_BP_SPLIT", 0) > 1) { C = env_int("UPSIDE_HIP_BP_SPLIT", 0); R.bp_resident = 0; }   // experiments / tests
        R.bp_C = C < 1 ? 1 : C;
        R.bp_test_abort = env_int("UPSIDE_HIP_BP_CLUSTER_TEST_ABORT", 0);
        set_matrix_form();
    }
    // One-workgroup solve with single-writer matrices: the pair-energy kernel writes exp(-E) itself and the matrices rest at 1
    // (= no interaction) between evaluations, which removes the solve's exp pass over every active matrix.  The cluster
    // solves and the accumulating (several beads per state) path keep energies resting at 0.
    void set_matrix_form() {
        // (a small batch is a chain of launches: the 1-body pass rides in the prologue of the one-workgroup solve)
        R.node_prob_in_solve = ((ctx->n_system <= 16 || ctx->L.batch) && env_int("UPSIDE_HIP_NODE_PROB_IN_SOLVE", 1)) ? 1 : 0;   // (a launch less where launches are what a step waits for; the cluster solve does the same for its own nodes)
        R.p_prob = (R.bp_C <= 1 && one_bead_per_state && !env_int("UPSIDE_HIP_BP_ENERGY_TABLE", 0)) ? 1 : 0;
        rest_matrices();
    }
    void rest_matrices() {
        if (R.p_prob) hip_check(hipMemsetD32Async((hipDeviceptr_t)P.p, 0x3f800000 /* 1.0f */, P.n, ctx->stream), "fill");
        else hip_check(hipMemsetAsync(P.p, 0, P.n * sizeof(float), ctx->stream), "memset");
    }
    void compute_value(ComputeMode mode) override {   // rotamer.cpp:779-789
        if (!bp_C_chosen) choose_bp_cluster();
        upk_check(upk_rotamer_node_prob(&ctx->L, &R), "rotamer_node_prob");
        { IGraphHost::Prof pr(ig, name, "igraph_fwd", 0);
          upk_check(upk_rotamer_pair_energy(&ctx->L, &R), "rotamer_pair_energy"); }
        if (ctx->profile) ctx->begin("bp:" + name);
        upk_check(upk_rotamer_bp(&ctx->L, &R, mode == PotentialAndDerivMode), "rotamer_bp");
        if (ctx->profile) ctx->end("bp:" + name, 0.);
        { IGraphHost::Prof pr(ig, name, "igraph_bwd", 1);
          upk_check(upk_rotamer_grad(&ctx->L, &R), "rotamer_grad"); }
        if (mode == PotentialAndDerivMode)
            hip_check(hipMemcpyAsync(potential_dev.p, energy.p, ctx->n_system * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream), "D2D");
    }
    vector<float> get_param() const override { return ig.param; }
    void set_param(const vector<float>& p) override { ig.set_param(p); R.G = ig.G; pack_param_tri(); }
    vector<float> get_param_deriv(int system) override {   // rotamer.cpp:1064-1066
        return param_deriv_table(ctx, ig.param.size(), [&](float* t) { upk_check(upk_rotamer_param_deriv(&ctx->L, &R, system, t), "rotamer param_deriv"); });
    }

    template <typename T> static vector<T> head(const DevBuf<T>& b, size_t n) { return sys_slice(b, 0, n); }
    // Pair energies of one system for the current structure (diagnostics / logging): the solve clears its accumulators,
    // so the pair-energy kernel produces them once more; read back, then cleared again for every system.  Inside
    // begin_log_frame / end_log_frame the kernel runs once for the read-outs of all systems.
    bool log_frame_open = false;
    void fill_pair_energies() {
        upk_rotamer_t Re = R;                      // energies wanted here, whatever form the solve keeps
        if (R.p_prob) { Re.p_prob = 0; hip_check(hipMemsetAsync(P.p, 0, P.n * sizeof(float), ctx->stream), "memset"); }
        upk_check(upk_rotamer_pair_energy(&ctx->L, &Re), "rotamer_pair_energy");
        hip_check(hipStreamSynchronize(ctx->stream), "sync");
    }
    void clear_pair_energies() {
        rest_matrices();
        hip_check(hipMemsetAsync(slot_active.p, 0, slot_active.n * sizeof(int), ctx->stream), "memset");
        hip_check(hipStreamSynchronize(ctx->stream), "sync");
    }
    void begin_log_frame() override { fill_pair_energies(); log_frame_open = true; }
    void end_log_frame() override { if (log_frame_open) { clear_pair_energies(); log_frame_open = false; } }
    void pair_energies(int sys, vector<float>& E, vector<int>& act) {
        if (!log_frame_open) fill_pair_energies();
        E = sys_slice(P, sys, (size_t)R.slot_cap * 36); act = sys_slice(slot_active, sys, (size_t)R.slot_cap);
        if (!log_frame_open) clear_pair_energies();
    }
    void pair_energies_of_system0(vector<float>& E, vector<int>& act) { pair_energies(0, E, act); }
    vector<float> one_body_energies(int sys) {   // rotamer.cpp:904-926: belief-weighted 1-body energy per node and parent
        const int np = (int)prob_nodes.size();
        auto nb = sys_slice(nb_cur, sys, (size_t)n_node * 6);
        vector<float> per_node((size_t)n_node * np, 0.f);
        for (int ip = 0; ip < np; ++ip) {
            auto out = sys_slice(prob_nodes[ip]->output, sys, (size_t)prob_nodes[ip]->n_elem * prob_nodes[ip]->stride);
            for (int i = 0; i < ig.G.n1; ++i)
                per_node[(size_t)bead_node[i] * np + ip] += nb[bead_node[i] * 6 + bead_rot[i]] * out[(size_t)ig.loc1[i] * prob_nodes[ip]->stride];
        }
        return per_node;
    }
    // node energies with the 1-state partners folded in (rotamer.cpp:697-711) or per-node free energies (:868-902),
    // assembled on the host from device results
    vector<float> folded_or_free_energies(int sys, bool want_node_energy) {
        const size_t cap = R.slot_cap;
        vector<float> E; vector<int> act;
        pair_energies(sys, E, act);
        auto sa = sys_slice(slot_a, sys, cap), sb = sys_slice(slot_b, sys, cap);
        auto cs = sys_slice(class_start, sys, 6); auto prob = sys_slice(node_prob, sys, (size_t)n_node * 6); auto off = sys_slice(node_off, sys, (size_t)n_node);
        auto nb = sys_slice(nb_cur, sys, (size_t)n_node * 6); auto mg = sys_slice(marg, sys, cap * 36);
        // classes in slot order: 3x3, 3x6, 6x6, 1x1, 1xN (kernels_rotamer.hip); a < b in node order, so a 1xN slot has a = the 1-state node
        for (int sl = cs[4]; sl < cs[5]; ++sl) if (act[sl])                                           // move_edge_prob_to_node2, rotamer.cpp:378-385
            for (int r = 0; r < node_nrot[sb[sl]]; ++r) prob[sb[sl] * 6 + r] *= expf(-E[((size_t)0 * cap + sl) * 6 + r]);
        if (want_node_energy) {
            vector<float> ne((size_t)n_node * 6);
            for (int g = 0; g < n_node; ++g) for (int r = 0; r < 6; ++r) ne[g * 6 + r] = r < node_nrot[g] ? -logf(prob[g * 6 + r]) : 1e5f;
            return ne;
        }
        vector<float> fe(n_node, 0.f);
        for (int g = 0; g < n_node; ++g) {                                                            // node_free_energy, rotamer.cpp:292-302
            float e = off[g];
            for (int r = 0; r < node_nrot[g]; ++r) { const float b = nb[g * 6 + r]; e += b * logf((1e-10f + b) / (1e-10f + prob[g * 6 + r])); }
            fe[g] += e;
        }
        for (int sl = cs[3]; sl < cs[4]; ++sl) if (act[sl]) { const float en = E[(size_t)sl * 6]; fe[sa[sl]] += 0.5f * en; fe[sb[sl]] += 0.5f * en; }   // -log(prob) of a 1x1 edge
        for (int sl = cs[0]; sl < cs[3]; ++sl) if (act[sl]) {                                          // edge_free_energy, rotamer.cpp:431-451
            const int a = sa[sl], b = sb[sl];
            float en = 0.f;
            for (int i = 0; i < node_nrot[a]; ++i) for (int j = 0; j < node_nrot[b]; ++j) {
                const float p = mg[((size_t)i * cap + sl) * 6 + j], pr = expf(-E[((size_t)i * cap + sl) * 6 + j]);   // slot matrices: [i][slot][j]
                en += p * logf((1e-10f + p) / (1e-10f + pr * nb[a * 6 + i] * nb[b * 6 + j]));
            }
            fe[a] += 0.5f * en; fe[b] += 0.5f * en;
        }
        return fe;
    }
    void add_loggers(vector<LogValue>& out) override {   // rotamer.cpp:657-672
        LogValue bad; bad.name = "rotamer_bad_solves_cumulative"; bad.dims = {1}; bad.as_long = true;
        bad.fill = [this](int sys, float* b) { b[0] = (float)sys_slice(n_bad, sys, 1)[0]; };
        out.push_back(bad);
        LogValue fe; fe.name = "rotamer_free_energy"; fe.dims = {(size_t)n_node};
        fe.fill = [this](int sys, float* b) { auto v = arrange_by_residue(folded_or_free_energies(sys, false), 1); copy(v.begin(), v.end(), b); };
        out.push_back(fe);
        const int np = (int)prob_nodes.size();
        for (int ip = 0; ip < np; ++ip) {
            LogValue e1; e1.name = "rotamer_1body_energy" + to_string(ip); e1.dims = {(size_t)n_node};
            e1.fill = [this, ip, np](int sys, float* b) {
                auto v = arrange_by_residue(one_body_energies(sys), np);
                for (int g = 0; g < n_node; ++g) b[g] = v[(size_t)g * np + ip]; };
            out.push_back(e1);
        }
    }
    // per-node values -> the reference's residue order: nodes in the order their first bead appears (rotamer.cpp:928-953)
    vector<float> arrange_by_residue(const vector<float>& per_node, int width) const {
        vector<float> out; out.reserve(per_node.size());
        vector<char> seen(n_node, 0);
        for (int o = 0; o < ig.G.n1; ++o) {          // residues in the order of the configuration's bead list
            const int i = ig.new_of.empty() ? o : ig.new_of[o];
            const int g = bead_node[i];
            if (bead_rot[i] != 0 || seen[g]) continue;
            seen[g] = 1;
            for (int k = 0; k < width; ++k) out.push_back(per_node[(size_t)g * width + k]);
        }
        if ((int)out.size() != n_node * width) throw string("wrong number of residues");
        return out;
    }
    vector<float> get_value_by_name(const char* log_name) override {   // rotamer.cpp:675-773 (system 0)
        hip_check(hipStreamSynchronize(ctx->stream), "sync");
        if (!strcmp(log_name, "n_node")) return vector<float>(1, (float)n_node);
        if (!strcmp(log_name, "count_edges_by_type")) return ig.count_edges_by_type(0);
        if (!strcmp(log_name, "rotamer_1body_energy")) return arrange_by_residue(one_body_energies(0), (int)prob_nodes.size());
        if (!strcmp(log_name, "node_energy")) return folded_or_free_energies(0, true);
        if (!strcmp(log_name, "rotamer_free_energy")) return arrange_by_residue(folded_or_free_energies(0, false), 1);
        if (!strcmp(log_name, "edge_marginal_in_graph_order") || !strcmp(log_name, "edge_energy")) {   // rotamer.cpp:712-763
            const bool do_marginal = !strcmp(log_name, "edge_marginal_in_graph_order");
            const size_t cap = R.slot_cap;
            auto nb = head(nb_cur, (size_t)n_node * 6); auto sa = head(slot_a, cap), sb = head(slot_b, cap);
            int ns = head(n_slot, 1)[0];
            vector<float> mg; vector<int> act;
            if (do_marginal) { mg = head(marg, cap * 36); act = head(slot_active_last, cap); }
            else pair_energies_of_system0(mg, act);            // -log(prob) of an edge entry is its pair energy
            vector<float> ev((size_t)n_node * n_node * 36, 0.f);
            if (do_marginal)
                for (int i1 = 0; i1 < n_node; ++i1) for (int i2 = 0; i2 < n_node; ++i2) for (int r1 = 0; r1 < 6; ++r1) for (int r2 = 0; r2 < 6; ++r2)
                    ev[(((size_t)i1 * n_node + i2) * 6 + r1) * 6 + r2] = (i1 == i2) ? nb[i1 * 6 + r1] * (r1 == r2) : nb[i1 * 6 + r1] * nb[i2 * 6 + r2];
            for (int sl = 0; sl < ns; ++sl) {
                if (!act[sl]) continue;
                int a = sa[sl], b = sb[sl];
                if (node_nrot[a] == 1 && node_nrot[b] != 1) continue;   // 1-3 / 1-6 edges are not listed (rotamer.cpp:745)
                for (int r1 = 0; r1 < node_nrot[a]; ++r1) for (int r2 = 0; r2 < node_nrot[b]; ++r2) {
                    float v = (do_marginal && node_nrot[b] == 1) ? 1.f : mg[((size_t)r1 * R.slot_cap + sl) * 6 + r2];
                    ev[(((size_t)a * n_node + b) * 6 + r1) * 6 + r2] = v;
                    ev[(((size_t)b * n_node + a) * 6 + r2) * 6 + r1] = v;
                }
            }
            return ev;
        }
        if (!strcmp(log_name, "bp_trace")) {   // diagnostics: phase clocks (10 ns units) of system 0's last solve
            if (!R.bp_trace) throw string("set UPSIDE_HIP_BP_TRACE=1 before constructing the engine");
            auto t = bp_trace.download(); return vector<float>(t.begin(), t.begin() + 32);
        }
        if (!strcmp(log_name, "read n_bad_solve") || !strcmp(log_name, "read n_bad_solve and reset")) {   // rotamer.cpp:764-770
            vector<float> r(1, float(head(n_bad, 1)[0]));
            if (strstr(log_name, "reset")) n_bad.fill_bytes(0);
            return r;
        }
        throw string("Value ") + log_name + string(" not implemented");
    }
};
struct RegisterRotamer {
    RegisterRotamer(string name_prefix) {
        add_node_creation_function(name_prefix, [name_prefix](DeviceCtx* c, hid_t_compat grp, const ArgList& args) -> DerivComputation* {
            if (args.size() < 1u) throw string("node " + name_prefix + " needs at least 1 arg");
            auto* node = new RotamerSidechain(c, grp, args); node->library_launchers_only = true; return node; });
    }
};
RegisterRotamer rotamer_node("rotamer");

}  // namespace

// accessors used by the C-ABI layer (engine_c_api.cpp)
int engine_pairlist(DerivEngine& e, const string& node_name, int sys, vector<pair<int, int>>& out) {
    auto* c = e.get(node_name).computation.get();
    IGraphHost* ig = nullptr;
    if (auto* r = dynamic_cast<RotamerSidechain*>(c)) ig = &r->ig;
    else if (auto* h = dynamic_cast<HBondCoverage*>(c)) ig = &h->ig;
    else if (auto* en = dynamic_cast<EnvironmentCoverage*>(c)) ig = &en->ig;
    else if (auto* p = dynamic_cast<ProteinHBond*>(c)) ig = &p->ig;
    if (!ig) return -1;
    out = ig->pairlist(sys);
    return (int)out.size();
}
// Algorithmic bytes of the LAST belief-propagation launch, all systems:
//   per sweep  4 * sum_t e_t * (P_t + 2 b_t) + 8 * sum_nodes w   with (P,b) = (12,8), (24,12), (48,16) floats for the
//   3x3 / 3x6 / 6x6 residue pairs that are active this step (the reference's padded rows, rotamer.cpp:331-334) and
//   w = 4 / 8 floats of belief per 3- / 6-state node; times the sweeps the solve took (+1 initial pass).
// what a belief-propagation launch must move at least once: the matrices of the active residue pairs in, their marginals out
// (same records), node probabilities in and beliefs out -- of the last solve, summed over the systems
double engine_bp_min_bytes(DerivEngine& e) {
    for (auto& n : e.nodes)
        if (auto* r = dynamic_cast<RotamerSidechain*>(n.computation.get())) {
            e.sync();
            auto cs = r->class_start.download(); auto act = r->slot_active_last.download();
            const int S = e.ctx.n_system, cap = r->R.slot_cap;
            const double Pt[3] = {12, 24, 48};      // floats of a 3x3 / 3x6 / 6x6 matrix as stored ([row][slot][6]-records, engine_bp_bytes)
            double total = 0.;
            for (int s = 0; s < S; ++s) {
                for (int c = 0; c < 3; ++c) {
                    long e_t = 0;
                    for (int sl = cs[(size_t)s * 6 + c]; sl < cs[(size_t)s * 6 + c + 1]; ++sl) e_t += act[(size_t)s * cap + sl] != 0;
                    total += 2. * 4. * e_t * Pt[c];
                }
                total += 2. * 4. * 6. * r->R.n_node;
            }
            return total;
        }
    return 0.;
}
double engine_bp_bytes(DerivEngine& e) {
    for (auto& n : e.nodes)
        if (auto* r = dynamic_cast<RotamerSidechain*>(n.computation.get())) {
            e.sync();
            auto cs = r->class_start.download(); auto act = r->slot_active_last.download(); auto it = r->iters.download();
            const int S = e.ctx.n_system, cap = r->R.slot_cap;
            const double Pt[3] = {12, 24, 48}, bt[3] = {8, 12, 16};
            const int n3 = r->R.n_node3, n6 = r->R.n_node - r->R.n_node1 - r->R.n_node3;
            double total = 0.;
            for (int s = 0; s < S; ++s) {
                double per_sweep = 8. * (4. * n3 + 8. * n6);
                for (int c = 0; c < 3; ++c) {
                    long e_t = 0;
                    for (int sl = cs[(size_t)s * 6 + c]; sl < cs[(size_t)s * 6 + c + 1]; ++sl) e_t += act[(size_t)s * cap + sl] != 0;
                    per_sweep += 4. * e_t * (Pt[c] + 2. * bt[c]);
                }
                total += per_sweep * (it[s] + 1);
            }
            if (getenv("UPSIDE_HIP_BP_STATS")) {   // diagnostics: slot occupancy of the last launch
                long n_slot[6] = {0}, n_act[6] = {0}, sweeps = 0;
                for (int s = 0; s < S; ++s) {
                    sweeps += it[s];
                    for (int c = 0; c < 5; ++c)
                        for (int sl = cs[(size_t)s * 6 + c]; sl < cs[(size_t)s * 6 + c + 1]; ++sl) { n_slot[c]++; n_act[c] += act[(size_t)s * cap + sl] != 0; }
                }
                fprintf(stderr, "bp stats: %d systems, %.1f sweeps/solve; slots (active/all) per system:", S, sweeps / (double)S);
                for (int c = 0; c < 5; ++c) fprintf(stderr, " c%d %.0f/%.0f", c, n_act[c] / (double)S, n_slot[c] / (double)S);
                fprintf(stderr, "; nodes 1/3/6: %d/%d/%d\n", r->R.n_node1, n3, n6);
            }
            return total;
        }
    return 0.;
}
int engine_rebuild_flags(DerivEngine& e, const string& node_name, vector<int>& flags) {
    auto* c = e.get(node_name).computation.get();
    IGraphHost* ig = nullptr;
    if (auto* r = dynamic_cast<RotamerSidechain*>(c)) ig = &r->ig;
    else if (auto* h = dynamic_cast<HBondCoverage*>(c)) ig = &h->ig;
    else if (auto* en = dynamic_cast<EnvironmentCoverage*>(c)) ig = &en->ig;
    else if (auto* p = dynamic_cast<ProteinHBond*>(c)) ig = &p->ig;
    if (!ig) return -1;
    e.sync();
    flags = ig->rebuild_flag.download();
    return 0;
}
int engine_igraph_stats(DerivEngine& e, const string& node_name, double* out) {
    // diagnostics (tools/list_stats.py): sizes and per-system mean list lengths of a node's interaction graph
    auto* c = e.get(node_name).computation.get();
    IGraphHost* ig = nullptr;
    if (auto* r = dynamic_cast<RotamerSidechain*>(c)) ig = &r->ig;
    else if (auto* h = dynamic_cast<HBondCoverage*>(c)) ig = &h->ig;
    else if (auto* en = dynamic_cast<EnvironmentCoverage*>(c)) ig = &en->ig;
    else if (auto* p = dynamic_cast<ProteinHBond*>(c)) ig = &p->ig;
    if (!ig) return -1;
    e.sync();
    const double S = e.ctx.n_system;
    auto total = [&](DevBuf<int>& b) { double t = 0.; if (b.n) for (int v : b.download()) t += v; return t / S; };
    out[0] = ig->G.n1; out[1] = ig->G.n2; out[2] = ig->G.cap1; out[3] = ig->G.cap2; out[4] = ig->G.cutoff; out[5] = ig->G.cache_cutoff;
    out[6] = total(ig->cnt1); out[7] = total(ig->cnt2); out[8] = total(ig->hcnt1); out[9] = total(ig->hcnt2); out[10] = ig->md_sides;
    return 0;
}
int engine_rotamer_iterations(DerivEngine& e, vector<int>& iters) {
    for (auto& n : e.nodes)
        if (auto* r = dynamic_cast<RotamerSidechain*>(n.computation.get())) { e.sync(); iters = r->iters.download(); return 0; }
    return -1;
}
double engine_igraph_bytes(DerivEngine& e) {
    // algorithmic bytes of all interaction graphs for one force pass of ONE system
    double b = 0.;
    for (auto& n : e.nodes) {
        auto* c = n.computation.get();
        IGraphHost* ig = nullptr;
        if (auto* r = dynamic_cast<RotamerSidechain*>(c)) ig = &r->ig;
        else if (auto* h = dynamic_cast<HBondCoverage*>(c)) ig = &h->ig;
        else if (auto* en = dynamic_cast<EnvironmentCoverage*>(c)) ig = &en->ig;
        else if (auto* p = dynamic_cast<ProteinHBond*>(c)) ig = &p->ig;
        if (ig) b += ig->algorithmic_bytes() / e.ctx.n_system;
    }
    return b;
}
