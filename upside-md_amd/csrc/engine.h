// Host-side engine: the reference's differentiable-graph contract re-hosted for device-resident data.
//
// The class names and the plug-in contract follow /root/reference/src/deriv_engine.h:48-335 so that a node
// written for the reference maps one-to-one: DerivComputation (compute_value / propagate_deriv / get_param /
// set_param / get_value_by_name), CoordNode (n_elem, elem_width, output, sens), PotentialNode (potential),
// DerivEngine (nodes, pos, potential, compute, integration_cycle) and the name-prefix registry
// (node_creation_map / add_node_creation_function / RegisterNodeType<T,N>).
// What differs is WHERE data lives: output/sens are device buffers holding n_system independent systems, and a
// node's methods enqueue HIP kernels (through the C launchers of include/upside_hip_kernels.h) on the engine's
// stream instead of computing on the host.  There is no host fallback for any node.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../include/upside_hip_kernels.h"

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
    // Work that depends on the parents' outputs only and is not on every step's critical path (pair-list upkeep).
    // The engine enqueues it on a side stream as soon as the last parent is computed, so a straggling rebuild of a
    // few systems overlaps with the nodes scheduled in between; compute_value() runs after it (event-ordered).
    virtual bool has_prepare() const { return false; }
    virtual void prepare() {}
    // second part of the upkeep, needed by propagate_deriv() only (the hit lists of the side the backward pass gathers
    // over): enqueued after every node's prepare(), so it runs beside the forward passes instead of in front of them
    virtual bool has_prepare_backward() const { return false; }
    virtual void prepare_backward() {}
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

struct Pos : public CoordNode {   // deriv_engine.h:122-141
    int n_atom;
    Pos(DeviceCtx* c, int n_atom_) : CoordNode(c, n_atom_, 3), n_atom(n_atom_) {}
    void compute_value(ComputeMode) override {}
    void propagate_deriv() override {}
};

struct DerivEngine {   // deriv_engine.h:145-237
    struct Node {
        std::string name;
        std::unique_ptr<DerivComputation> computation;
        std::vector<size_t> parents, children;
        int germ_exec_level = -1, deriv_exec_level = -1;
    };
    DeviceCtx ctx;
    std::vector<Node> nodes;
    Pos* pos = nullptr;
    std::vector<float> potential;   // [S]

    // MD state (System of main.cpp:93-111, one entry per system)
    DevBuf<float> mom;              // [S][n_atom][4]
    DevBuf<uint32_t> seed; DevBuf<float> mom_scale, noise_scale;
    std::vector<float> temperature; std::vector<uint32_t> seeds;
    float thermostat_timescale = 5.f, dt = 0.009f; int thermostat_interval = 1;
    uint64_t n_invocations = 0, round_num = 0;      // host mirrors; the thermostat reads the device copy below
    DevBuf<unsigned long long> n_invocations_dev;   // [1]
    void set_invocations(uint64_t n);

    // execution order of one force pass, fixed at finalize() (the BFS of deriv_engine.cpp:124-169 unrolled)
    struct Step { int node; bool backward; bool prepare = false; };
    std::vector<Step> schedule;
    struct Side { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr, join_bwd = nullptr; bool owns_stream = true; };
    std::map<int, Side> side;                  // node index -> side stream of its prepare() (empty when disabled)
    int last_prepare_step = -1;                // index in `schedule` of the last prepare step
    DevBuf<float*> zero_ptrs; DevBuf<long> zero_sizes; int n_zero = 0;   // every CoordNode's sens, cleared by one launch per force pass

    DerivEngine(int n_atom, int n_system);
    ~DerivEngine();
    void add_node(const std::string& name, std::unique_ptr<DerivComputation> fcn, std::vector<std::string> argument_names);
    Node& get(const std::string& name);
    int get_idx(const std::string& name, bool must_exist = true);
    template <typename T> T& get_computation(const std::string& name) {
        auto c = get(name).computation.get();
        if (!c) throw std::string("impossible pointer value");
        return dynamic_cast<T&>(*c);
    }
    void finalize();
    void compute(ComputeMode mode);            // enqueue; no synchronisation
    void fetch_potentials();                   // D2H of every PotentialNode::potential + engine total
    void integration_cycle(float dt, float max_force = 0.f);   // deriv_engine.cpp:172-192 (Verlet weights)
    void integration_stage(int stage, float dt, float max_force);   // one force evaluation + leapfrog sub-step (deriv_engine.cpp:172-192)
    int stage_num = 0;                                              // sub-step the next upside_hip_run_steps call starts with
    void md_step();                                                 // thermostat (at round starts) + one integration stage, enqueued
    void run_steps(int n_step);                                     // n_step MD steps; replays a captured hipGraph where it can

    // A launch-bound batch (few systems) spends more time between kernels than in them: 6 MD steps (two rounds: the
    // period of the leapfrog stage AND of the list-parity pattern) are captured once into a hipGraph, side streams
    // included, and replayed.  Invalidated by anything that changes a kernel argument.
    hipGraph_t md_graph = nullptr; hipGraphExec_t md_graph_exec = nullptr;
    bool md_graph_ready = false; int md_graph_parity = 0; uint64_t steps_done = 0, n_compute = 0;
    std::vector<float> swap_energy;   // energies seen by the last replica-swap set, accepted pairs already traded (upside_hip_replica_swap_next)
    bool graph_failed = false;   // capture was refused once: stay on plain launches
    void invalidate_graph();
    bool capture_md_graph();
    // Monte-Carlo pivot sampler (monte_carlo_sampler.cpp): loaded from /input/pivot_moves, all systems step together
    struct Pivot {
        bool loaded = false; upk_pivot_t P{};
        DevBuf<int> atoms, range, restype, stats; DevBuf<float> pot, cdf, pos_copy, delta_lprob, e_old, e_new, temperature;
    } pivot;
    struct Jump {   // rigid-body moves of chain segments (monte_carlo_sampler.cpp:157-251)
        bool loaded = false; upk_jump_t J{};
        DevBuf<int> range, stats; DevBuf<float> sigma_trans, sigma_rot;
    } jump;
    void load_pivot_moves(hid_t_compat input_group);   // throws if the group is malformed
    void load_jump_moves(hid_t_compat input_group);
    void mc_step(uint64_t round);                      // every loaded sampler in the reference's order (pivot, jump): two
                                                       // energy evaluations + proposal + Metropolis each, every system
    void* comm = nullptr; void (*comm_free)(void*) = nullptr;   // replica exchange across GPUs (comm_rccl.cpp), owned by the engine
    void check_device_errors();                // throws if a capacity overflow was flagged
    void sync();
};

DerivEngine* initialize_engine_from_hdf5(int n_atom, int n_system, hid_t_compat potential_group, bool quiet = false);

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
