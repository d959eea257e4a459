// Public plug-in contract of libupside_hip.so: everything a node type defined OUTSIDE the library needs in order to be
// registered, constructed from the configuration file and scheduled by the engine.
//
// It is the device-resident counterpart of /root/reference/src/deriv_engine.h:48-118 (DerivComputation, CoordNode,
// PotentialNode, HBondCounter) and :239-335 (NodeCreationFunction, add_node_creation_function, RegisterNodeType<T,N>,
// check_elem_width, check_arguments_length): same class names, same virtuals, same registry-by-name-prefix, so a node
// written for the reference maps one to one.  What differs is WHERE the data lives: output / sens are device buffers that
// hold n_system independent systems ([system][element][stride] floats, stride = width rounded up to 4), and a node's
// methods ENQUEUE work on ctx->stream instead of computing in place.
//
// Two ways to write a node against this header:
//   * device node: compute_value()/propagate_deriv() launch your own HIP kernels (or the launchers of
//     upside_hip_kernels.h) on ctx->stream; derivative contributions to a parent go through the parent's ScatterPlan
//     (add_source at construction, write [term][slot][width] floats at source_ptr(id) + system*arena_size in compute_value);
//   * host node: derive from HostPotentialNode / HostCoordNode below and implement the maths on host arrays exactly like a
//     reference node does; the base class does the explicit device->host->device round trip (a stream synchronisation per
//     force evaluation: correct, slow, and it disables hipGraph replay of the MD step for that engine).
//
// A plug-in is a shared library linked against libupside_hip.so whose static initialisers register its node types
// (`static RegisterNodeType<MyNode,1> reg("my_node");`); it is loaded with upside_hip_load_plugin(path)
// (upside_engine_c.h) -- or by listing it in the UPSIDE_HIP_PLUGINS environment variable (':'-separated), which
// upside_main and upside_hip_create_engine* read -- before the configuration that names the node is opened.
// tests/plugin/host_pull.cpp is a complete example; tests/test_gpu_parity.py::test_external_plugin_node drives it.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "upside_hip_kernels.h"

typedef long long hid_t_compat;

inline int round_up(int i, int a) { return ((i + a - 1) / a) * a; }
inline int ru(int i) { return i == 1 ? i : round_up(i, 4); }   // vector_math.h:23-25

void hip_check(hipError_t e, const char* what);
void upk_check(int code, const char* what);

// ---- device memory ----------------------------------------------------------------------------------
template <typename T>
struct DevBuf {
    T* p = nullptr; size_t n = 0;
    DevBuf() {}
    explicit DevBuf(size_t n_) { alloc(n_); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void alloc(size_t n_) {
        if (p) { (void)hipFree(p); p = nullptr; }
        n = n_;
        // hipMemset runs on the NULL stream and may still be in flight when the call returns; engine streams are
        // non-blocking (they do not order against the NULL stream), so drain it before anyone can touch the buffer
        if (n) { hip_check(hipMalloc((void**)&p, n * sizeof(T)), "hipMalloc"); hip_check(hipMemset(p, 0, n * sizeof(T)), "hipMemset"); hip_check(hipStreamSynchronize(nullptr), "sync"); }
    }
    void upload(const std::vector<T>& v) { alloc(v.size()); if (n) hip_check(hipMemcpy(p, v.data(), n * sizeof(T), hipMemcpyHostToDevice), "H2D"); }
    std::vector<T> download() const {
        std::vector<T> v(n);
        if (n) hip_check(hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost), "D2H");
        return v;
    }
    void fill_bytes(int byte) { if (n) { hip_check(hipMemset(p, byte, n * sizeof(T)), "hipMemset"); hip_check(hipStreamSynchronize(nullptr), "sync"); } }
};

enum ComputeMode { DerivMode = 0, PotentialAndDerivMode = 1 };   // deriv_engine.h:42-45

struct DerivEngine;

// per-engine launch context shared by all nodes
struct DeviceCtx {
    int n_system = 1;
    hipStream_t stream = nullptr;
    upk_launch_t L{};
    DevBuf<int> error_flag;          // [1]: pair-list / slot capacity overflow
    // profiling (bench.py): HIP-event timing of kernel families on `stream`
    bool profile = false;
    struct Family { double ms = 0; long launches = 0; double bytes = 0; double pairs = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; };   // one family = one kernel of one node
    std::map<std::string, Family> families;
    void begin(const std::string& fam);
    void end(const std::string& fam, double algorithmic_bytes, double pair_evaluations = 0.);
    void flush_profile();
    // run the queued per-element ops (upside_hip_kernels.h: upk_launch_t::fuse).  A node that enqueues its OWN work on `stream`
    // (kernels, copies, events) calls this first; the launchers of upside_hip_kernels.h do it themselves.
    void flush();
    unsigned long long n_pass = 0;   // number of the force pass being enqueued (DerivEngine::compute), for nodes that double-buffer by step parity
};

// deferred derivative contributions gathered into a CoordNode's sens (see upk_gather_contrib)
struct ScatterPlan {
    struct Source { int n_term, n_slot, width; std::vector<int> targets; long offset; };
    std::vector<Source> sources;
    long arena_size = 0;
    int width = 0;
    DevBuf<float> arena; DevBuf<int> csr_start, csr_entry;
    bool finalized = false;
    // targets[term*n_slot+slot] = element of the owning node (or -1); returns the source id
    int add_source(int n_term, int n_slot, int width, const std::vector<int>& targets);
    float* source_ptr(int id) const { return arena.p + sources[id].offset; }
    void finalize(int n_target, int n_system);
};

// one named per-frame quantity of a node in the /output group (the reference's default_logger->add_logger calls)
struct LogValue {
    std::string name; std::vector<size_t> dims; bool as_long = false; int level = 1;   // 0 basic, 1 detailed, 2 extensive
    std::function<void(int system, float* buffer)> fill;                              // as_long: values are written as int64
};

struct DerivComputation {   // deriv_engine.h:48-80
    const bool potential_term;
    DeviceCtx* ctx = nullptr;
    std::string name;   // graph name, set by DerivEngine::add_node (used for profiling labels)
    explicit DerivComputation(bool potential_term_) : potential_term(potential_term_) {}
    virtual ~DerivComputation() {}
    virtual void compute_value(ComputeMode mode) = 0;
    virtual void propagate_deriv() = 0;
    virtual std::vector<float> get_param() const { return std::vector<float>(); }
    virtual void set_param(const std::vector<float>&) {}
    // derivative of the total potential w.r.t. get_param(), for system `system` of the batch, from the state the last
    // evaluate_deriv left on the device (deriv_engine.h:71-74; always compiled here)
    virtual std::vector<float> get_param_deriv(int system) { (void)system; return std::vector<float>(); }
    virtual std::vector<float> get_value_by_name(const char*) { throw std::string("No values implemented"); }
    virtual void finalize() {}   // called once after the whole graph exists (scatter plans, device pointer tables)
    // /output loggers of this node (state_logger.h add_logger); begin/end bracket the frame's read-outs of all systems
    virtual void add_loggers(std::vector<LogValue>&) {}
    virtual void begin_log_frame() {}
    virtual void end_log_frame() {}
    virtual bool capturable() const { return true; }   // false: kernel arguments change from step to step (no hipGraph replay)
    // true: compute_value / propagate_deriv / prepare enqueue work through the launchers of upside_hip_kernels.h ONLY (which keep
    // the fused-op queue in order themselves); false (the default, any plug-in node): the engine runs the queue before every call
    bool library_launchers_only = false;
    // compute_value (fused_forward) / gather + propagate_deriv (fused_backward) enqueue fused per-element ops ONLY: the engine orders
    // the sweep so that such steps run back to back and share launches (DerivEngine::finalize)
    bool fused_forward = false, fused_backward = false;
    // Work that depends on the parents' outputs only and is not on every step's critical path (pair-list upkeep).
    // The engine enqueues it on a side stream as soon as the last parent is computed, so a straggling rebuild of a
    // few systems overlaps with the nodes scheduled in between; compute_value() runs after it (event-ordered).
    virtual bool has_prepare() const { return false; }
    virtual void prepare() {}
    std::vector<const DerivComputation*> prepare_deps;   // parents prepare() reads (empty = all of them)
};

struct CoordNode : public DerivComputation {   // deriv_engine.h:83-96
    int n_elem, elem_width, stride;
    DevBuf<float> output, sens;   // [S][n_elem][stride]
    ScatterPlan scatter;
    CoordNode(DeviceCtx* c, int n_elem_, int elem_width_);
    upk_coord_t coord() const { upk_coord_t r; r.out = output.p; r.sens = sens.p; r.n_elem = n_elem; r.width = elem_width; r.stride = stride; return r; }
    long sys_stride() const { return (long)n_elem * stride; }
    void gather_contributions();
    void finalize() override { if (!scatter.sources.empty()) scatter.finalize(n_elem, ctx->n_system); }
};

struct PotentialNode : public DerivComputation {   // deriv_engine.h:100-110
    DevBuf<float> potential_dev;      // [S]
    DevBuf<float> pot_terms;          // [S][n_term] scratch for the deterministic reduction
    int n_pot_term = 0;
    std::vector<float> potential;     // host copy, valid after DerivEngine::fetch_potentials
    explicit PotentialNode(DeviceCtx* c) : DerivComputation(true) { ctx = c; potential_dev.alloc(c->n_system); potential.assign(c->n_system, 0.f); }
    void alloc_terms(int n) { n_pot_term = n; pot_terms.alloc((size_t)ctx->n_system * n); }
    void reduce_terms() { upk_check(upk_reduce_sum(&ctx->L, pot_terms.p, n_pot_term, potential_dev.p, 0), "reduce_sum"); }
    void propagate_deriv() override {}
};

struct HBondCounter : public PotentialNode {   // deriv_engine.h:114-118
    using PotentialNode::PotentialNode;
};

// ---- host-fallback nodes ------------------------------------------------------------------------------
// The maths of a host node runs on the CPU on dense per-system arrays ([n_elem][elem_width], no padding), written the way
// a reference node writes it.  Every evaluation synchronises ctx->stream, copies the arguments' outputs to the host,
// calls the virtual below once per system and copies the results back (explicit host<->device sync; see the header note).

// A potential term: host_potential returns the energy of one system and ADDS d(energy)/d(argument a) into d_in[a]
// (zeroed by the caller) -- the role of a reference PotentialNode::compute_value that updates its arguments' sens.
struct HostPotentialNode : public PotentialNode {
    std::vector<CoordNode*> args;
    HostPotentialNode(DeviceCtx* c, const std::vector<CoordNode*>& args_);
    virtual float host_potential(int system, const std::vector<const float*>& in, const std::vector<float*>& d_in) = 0;
    void compute_value(ComputeMode mode) override final;
    bool capturable() const override { return false; }
private:
    std::vector<int> src_;                        // one scatter source per argument (identity targets)
    std::vector<std::vector<float>> in_, din_, stage_;
};

// A derived coordinate: host_value fills out[n_elem][elem_width] of one system from the arguments; host_deriv receives
// d(potential)/d(out) and ADDS d(potential)/d(argument a) into d_in[a] (zeroed by the caller) -- compute_value and
// propagate_deriv of a reference CoordNode.
struct HostCoordNode : public CoordNode {
    std::vector<CoordNode*> args;
    HostCoordNode(DeviceCtx* c, int n_elem, int elem_width, const std::vector<CoordNode*>& args_);
    virtual void host_value(int system, const std::vector<const float*>& in, float* out) = 0;
    virtual void host_deriv(int system, const std::vector<const float*>& in, const float* d_out, const std::vector<float*>& d_in) = 0;
    void compute_value(ComputeMode mode) override final;
    void propagate_deriv() override final;
    bool capturable() const override { return false; }
private:
    std::vector<int> src_;
    std::vector<std::vector<float>> in_, din_, stage_;
    std::vector<float> out_, dout_;
};

// ---- registry (deriv_engine.h:239-335) -----------------------------------------------------------------
typedef std::vector<CoordNode*> ArgList;
typedef std::function<DerivComputation*(DeviceCtx*, hid_t_compat, const ArgList&)> NodeCreationFunction;
typedef std::map<std::string, NodeCreationFunction> NodeCreationMap;
NodeCreationMap& node_creation_map();
bool is_prefix(const std::string& s1, const std::string& s2);
void add_node_creation_function(std::string name_prefix, NodeCreationFunction fcn);
void check_elem_width(const CoordNode& node, int expected);
void check_elem_width_lower_bound(const CoordNode& node, int lower_bound);
void check_arguments_length(const ArgList& arguments, int n_expected);

template <typename NodeClass, int n_args>
struct RegisterNodeType { RegisterNodeType(std::string name_prefix); };
template <typename NodeClass>
struct RegisterNodeType<NodeClass, -1> {
    RegisterNodeType(std::string name_prefix) {
        add_node_creation_function(name_prefix, [](DeviceCtx* c, hid_t_compat grp, const ArgList& args) {
            if (!args.size()) throw std::string("Expected at least 1 arg");
            return new NodeClass(c, grp, args); });
    }
};
template <typename NodeClass>
struct RegisterNodeType<NodeClass, 0> {
    RegisterNodeType(std::string name_prefix) {
        add_node_creation_function(name_prefix, [](DeviceCtx* c, hid_t_compat grp, const ArgList& args) {
            check_arguments_length(args, 0); return new NodeClass(c, grp); });
    }
};
template <typename NodeClass>
struct RegisterNodeType<NodeClass, 3> {
    RegisterNodeType(std::string name_prefix) {
        add_node_creation_function(name_prefix, [](DeviceCtx* c, hid_t_compat grp, const ArgList& args) {
            check_arguments_length(args, 3); return new NodeClass(c, grp, *args[0], *args[1], *args[2]); });
    }
};
template <typename NodeClass>
struct RegisterNodeType<NodeClass, 1> {
    RegisterNodeType(std::string name_prefix) {
        add_node_creation_function(name_prefix, [](DeviceCtx* c, hid_t_compat grp, const ArgList& args) {
            check_arguments_length(args, 1); return new NodeClass(c, grp, *args[0]); });
    }
};
template <typename NodeClass>
struct RegisterNodeType<NodeClass, 2> {
    RegisterNodeType(std::string name_prefix) {
        add_node_creation_function(name_prefix, [](DeviceCtx* c, hid_t_compat grp, const ArgList& args) {
            check_arguments_length(args, 2); return new NodeClass(c, grp, *args[0], *args[1]); });
    }
};
