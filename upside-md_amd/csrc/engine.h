// Host-side engine: the reference's differentiable-graph contract re-hosted for device-resident data.
//
// The class names and the plug-in contract follow /root/reference/src/deriv_engine.h:48-335 so that a node
// written for the reference maps one-to-one: DerivComputation (compute_value / propagate_deriv / get_param /
// set_param / get_value_by_name), CoordNode (n_elem, elem_width, output, sens), PotentialNode (potential),
// DerivEngine (nodes, pos, potential, compute, integration_cycle) and the name-prefix registry
// (node_creation_map / add_node_creation_function / RegisterNodeType<T,N>).
// What differs is WHERE data lives: output/sens are device buffers holding n_system independent systems, and a
// node's methods enqueue HIP kernels (through the C launchers of include/upside_hip_kernels.h) on the engine's
// stream instead of computing on the host.  None of the built-in nodes has a host fallback; node types defined outside
// the library plug in through include/upside_hip_plugin.h (device nodes, or HostPotentialNode / HostCoordNode whose maths
// runs on the host behind an explicit synchronisation).
#pragma once
#include "../../include/upside_hip_plugin.h"   // the public part of the contract: DerivComputation, CoordNode, PotentialNode, registry

struct Pos : public CoordNode {   // deriv_engine.h:122-141
    int n_atom;
    Pos(DeviceCtx* c, int n_atom_) : CoordNode(c, n_atom_, 3), n_atom(n_atom_) { library_launchers_only = true; fused_forward = fused_backward = true; }
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
    int integrator_type = 0;        // IntegratorType of deriv_engine.h:230: 0 = Verlet, 1 = Predescu
    uint64_t n_invocations = 0, round_num = 0;      // host mirrors; the thermostat reads the device copy below
    DevBuf<unsigned long long> n_invocations_dev;   // [S] (equal entries: each system's workgroup advances its own)
    void set_invocations(uint64_t n);

    // execution order of one force pass, fixed at finalize() (the BFS of deriv_engine.cpp:124-169 unrolled)
    struct Step { int node; bool backward; bool prepare = false; int batch = -1; bool skip_prepare = false; };   // batch: steps of one merged-launch group (consecutive, mutually independent)
    std::vector<Step> schedule;
    struct Side { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool owns_stream = true; };
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
    void print_schedule();
    int n_batch_group = 0;
    void compute(ComputeMode mode, bool keep_pending = false);   // enqueue; no synchronisation.  keep_pending: leave queued fused ops for the caller to extend (MD loop)
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
    DevBuf<float> swap_row;           // staging row of upside_hip_swap_between (exchange between engines of different potentials)
    std::vector<float> swap_energy;   // energies seen by the last replica-swap set, accepted pairs already traded (upside_hip_replica_swap_next)
    uint64_t swap_energy_round = ~0ull, swap_energy_compute = 0;   // the attempt they belong to: (round, force passes done when they were captured)
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

