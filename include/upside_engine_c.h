/* Drop-in boundary of the MI355X-native Upside engine (libupside_hip.so).
 *
 * Part 1 repeats, symbol for symbol, the reference's C-ABI
 *     /root/reference/src/engine_c_library.h:12-32   (+ upside_main, /root/reference/src/main.h:1-3)
 * so that `py/upside_engine.py:19-64` (ctypes) binds to this library unchanged.  Semantics kept:
 *   - construct_deriv_engine opens `potential_file` read-only and builds the DerivComputation graph from
 *     group /input/potential (engine_c_library.cpp:9-20); returns NULL on failure, never throws.
 *   - all other calls return 0 on success, 1 on failure after printing "ERROR: ..." to stderr
 *     (engine_c_library.cpp:37-45); get_param_deriv behaves like a -DPARAM_DERIV build of the reference (:93-100):
 *     the derivative of the potential w.r.t. get_param() of the named node, from the state the last evaluate_deriv
 *     left behind; a node without parameter derivatives has an empty vector (n_param must then be 0).
 *   - pos / deriv are dense row-major (n_atom,3) fp32 HOST buffers owned by the caller (:30-33,57-60);
 *     node outputs are dense (n_elem, elem_width) (:146-149); potential nodes report (1,1).
 *   - evaluate_* always run PotentialAndDerivMode and leave node output/sens inspectable (:34,55).
 * The force pass itself runs as HIP kernels on the current device; if no GPU / kernel image is usable the
 * calls FAIL (return NULL / 1) -- there is no CPU fallback in this library.
 *
 * Part 2 is the batched / device-resident extension the reference has no single call for: its MD loop
 * lives in upside_main (/root/reference/src/main.cpp:616-667).  These entry points keep S independent
 * systems (replicas / ensemble members) of one topology resident in HBM and step them together.
 */
#ifndef UPSIDE_ENGINE_C_H
#define UPSIDE_ENGINE_C_H
#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct DerivEngine;
typedef struct DerivEngine DerivEngine;

/* ---- Part 1: engine_c_library.h:12-32 ------------------------------------------------------ */
DerivEngine* construct_deriv_engine(int n_atom, const char* potential_file, bool quiet); /* engine_c_library.h:12  */
void free_deriv_engine(DerivEngine* engine);                                             /* :13 */
int evaluate_energy(float* energy, DerivEngine* engine, const float* pos);               /* :15 */
int evaluate_deriv(float* deriv, DerivEngine* engine, const float* pos);                 /* :16 */
int set_param(int n_param, const float* param, DerivEngine* engine, const char* node_name);        /* :18 */
int get_param_deriv(int n_param, float* deriv, DerivEngine* engine, const char* node_name);        /* :20 */
int get_param(int n_param, float* param, DerivEngine* engine, const char* node_name);              /* :21 */
int get_output_dims(int* n_elem, int* elem_width, DerivEngine* engine, const char* node_name);     /* :22 */
int get_output(int n_output, float* output, DerivEngine* engine, const char* node_name);           /* :23 */
int get_sens(int n_output, float* output, DerivEngine* engine, const char* node_name);             /* :24 */
int get_value_by_name(int n_output, float* output, DerivEngine* engine,
                      const char* node_name, const char* log_name);                                /* :26-27 */
int clamped_spline_solve(int N, float* bspline_coeff, const float* values);                        /* :29 */
int clamped_spline_value(int N, float* result, const float* bspline_coeff, int nx, float* x);      /* :30 */
int get_clamped_value_and_deriv(int N, float* result, const float* bspline_coeff, int nx, float* x); /* :31 */
int get_clamped_coeff_deriv(int N, float* result, const float* bspline_coeff, float x);            /* :32 */
int upside_main(int argc, const char* const* argv, int verbose);                                   /* main.h:1-3 */

/* ---- Part 2: batched device-resident extension ---------------------------------------------- */

/* select the HIP device used by engines constructed afterwards on this thread (one process per GPU) */
int upside_hip_set_device(int device);

/* Same as construct_deriv_engine but with n_system independent copies (replicas / ensemble members) of
 * the topology, all resident on the current HIP device.  construct_deriv_engine == n_system 1. */
DerivEngine* upside_hip_construct(int n_atom, const char* potential_file, int n_system, bool quiet);
int upside_hip_n_system(DerivEngine* engine);

/* host (n_system, n_atom, 3) <-> device positions / momenta */
int upside_hip_set_pos(DerivEngine* engine, const float* pos);
int upside_hip_get_pos(DerivEngine* engine, float* pos);
int upside_hip_set_mom(DerivEngine* engine, const float* mom);
int upside_hip_get_mom(DerivEngine* engine, float* mom);

/* force pass on all systems from the device-resident positions: energy (n_system) may be NULL
 * (DerivMode, deriv_engine.h:42-45); deriv (n_system,n_atom,3) may be NULL. */
int upside_hip_compute(DerivEngine* engine, float* energy, float* deriv);

/* Thermostat set-up of upside_main (main.cpp:515-523): per-system temperature and seed
 * (seed_s = base_seed + s, main.cpp:459), momenta fully resampled, n_invocations reset. */
int upside_hip_init_md(DerivEngine* engine, const float* temperature, uint32_t base_seed,
                       float thermostat_timescale, float dt, int thermostat_interval_rounds);
/* ... with one seed per system (a run whose systems are spread over several engines keeps seed = base + GLOBAL index) */
int upside_hip_init_md_seeds(DerivEngine* engine, const float* temperature, const uint32_t* seeds,
                             float thermostat_timescale, float dt, int thermostat_interval_rounds);

/* n_round integration cycles (3 leapfrog stages each, deriv_engine.cpp:172-192) with the
 * Ornstein-Uhlenbeck thermostat (thermostat.cpp:9-18) every thermostat_interval rounds; everything
 * stays on the device, the call returns after the stream has drained. */
int upside_hip_run_md(DerivEngine* engine, int n_round);
/* the weights of the three stages of a cycle (IntegratorType, /root/reference/src/deriv_engine.h:230, deriv_engine.cpp:173-180):
 * 0 = Verlet (the default; what main.cpp:663 uses), 1 = Predescu et al. 2012.  Between cycles only. */
int upside_hip_set_integrator(DerivEngine* engine, int type);
/* change the thermostat temperature of every system between run_md calls (simulated annealing, main.cpp:436-442,658-660) */
int upside_hip_set_temperature(DerivEngine* engine, const float* temperature);
/* the same loop counted in MD steps (one force evaluation + one leapfrog stage each, the unit of the reference's
 * "steps/s", main.cpp:677-682); a cycle left unfinished is resumed by the next call. */
int upside_hip_run_steps(DerivEngine* engine, int n_step);

/* Monte-Carlo moves (monte_carlo_sampler.cpp): load /input/pivot_moves and /input/jump_moves of a configuration
 * (returns the number of samplers found, 0 if neither group is present, -1 on error); one MC step of EVERY system at
 * `round` = the round number of main.cpp:628-630: each loaded sampler in the reference's order (pivot: proposal
 * from the Ramachandran proposal map, random stream 2; jump: rigid-body move of a chain segment, stream 3) spends two
 * energy evaluations and a Metropolis test at the system's temperature.  stats (n_system,2) = {n_success,
 * n_attempt} of sampler 0 (pivot) or 1 (jump). */
int upside_hip_load_mc(DerivEngine* engine, const char* config_file);
int upside_hip_mc_step(DerivEngine* engine, uint64_t round);
int upside_hip_mc_stats(DerivEngine* engine, int sampler, int* stats, int reset);
int upside_hip_mc_loaded(DerivEngine* engine, int sampler);

/* recenter (deriv_engine.cpp:37-48) all systems */
int upside_hip_recenter(DerivEngine* engine);
/* the same with the z component left alone when xy_only != 0 (--disable-z-recentering, main.cpp:358-360,416) */
int upside_hip_recenter_axes(DerivEngine* engine, int xy_only);

/* Replica-exchange swap attempt among the systems of THIS engine (main.cpp:227-275): pairs (n_pair,2) are
 * one swap set; energies are evaluated, Metropolis tested with the REPLICA_EXCHANGE random stream
 * (random.h:26, keyed by round) and accepted pairs exchange coordinates.  accepted (n_pair) out. */
int upside_hip_replica_swap(DerivEngine* engine, int n_pair, const int* pairs, uint32_t base_seed,
                            uint64_t round, int* accepted);
/* the same for the second and later swap sets of one attempt (main.cpp:249: ONE generator per attempt_swaps call):
 * draw0 = accepted[n_pair] returned for the previous set; accepted has n_pair+1 entries here. */
int upside_hip_replica_swap_from(DerivEngine* engine, int n_pair, const int* pairs, uint32_t base_seed,
                                 uint64_t round, int draw0, int* accepted);
/* a later swap set of the SAME attempt without a new force evaluation: in a temperature-only exchange the accepted pairs
 * of the earlier sets merely traded their energies (the reference re-evaluates, main.cpp:251-259, and gets the same numbers) */
int upside_hip_replica_swap_next(DerivEngine* engine, int n_pair, const int* pairs, uint32_t base_seed,
                                 uint64_t round, int draw0, int* accepted);

/* Replica exchange ACROSS engines / GPUs.  Each rank all-gathers one energy per system
 * (upside_hip_compute), every rank then calls upside_replica_decide on the identical global arrays: host arithmetic
 * only (no device needed), the same Metropolis test and random stream as main.cpp:251-273 with
 * lboltz_diff = (beta1-beta2)(E1-E2); pairs (n_pair,2) index the GLOBAL system list; accepted has n_pair+1 entries,
 * the last one is the generator position to pass as draw0 for the next swap set of the same round.
 * Accepted pairs exchange coordinates (momenta stay with the temperature slot, main.cpp:244-247): two systems of one
 * engine with upside_hip_swap_systems, otherwise get/set_system_pos around a point-to-point transfer. */
int upside_replica_decide(int n_pair, const int* pairs, const float* beta, const float* energy, uint32_t base_seed,
                          uint64_t round, int draw0, int* accepted);
/* Mixed Hamiltonians (one engine per distinct potential, main.cpp:450-571): the reference's own procedure, main.cpp:251-273 --
 * log-Boltzmann factors of every system before and after trading the coordinates of a set's pairs,
 * lboltz_diff[p] = (new[s1]+new[s2]) - (old[s1]+old[s2]), this Metropolis test on them (same generator, a uniform drawn
 * only for a rejectable pair), rejected pairs traded back.  upside_hip_swap_between moves coordinates device to device. */
int upside_replica_decide_lboltz(int n_pair, const float* lboltz_diff, uint32_t base_seed, uint64_t round, int draw0, int* accepted);
int upside_hip_swap_between(DerivEngine* engine1, int system1, DerivEngine* engine2, int system2);
int upside_hip_get_system_pos(DerivEngine* engine, int system, float* pos);        /* host (n_atom,3) */
int upside_hip_set_system_pos(DerivEngine* engine, int system, const float* pos);
int upside_hip_swap_systems(DerivEngine* engine, int system1, int system2);
/* the same for n_pair disjoint pairs (pairs: host array (n_pair,2)) in one launch: the accepted on-GPU pairs of a swap set */
int upside_hip_swap_system_pairs(DerivEngine* engine, int n_pair, const int* pairs);

/* Replica exchange across GPUs over RCCL / xGMI, one process per GPU, no host staging (csrc/comm_rccl.cpp; reference
 * semantics src/main.cpp:227-275 with the loop of :616-672).  Global system g = rank * n_system + local index, so a rank
 * holds a contiguous block of the temperature ladder and only the pairs straddling a block boundary cross GPUs.
 *   upside_hip_comm_get_unique_id: rank 0 makes the 128-byte ncclUniqueId; the launcher hands it to every rank (file, env,
 *     torch.distributed, MPI ...).  Ranks must be started BEFORE any of them touches the GPU.
 *   upside_hip_comm_init: ncclCommInitRank on the engine's device; temperature_global = the whole ladder (world * n_system).
 *   upside_hip_comm_replica_swap: ONE swap set (pairs of GLOBAL system ids).  first_set != 0: the first set of an attempt
 *     (force pass, device-side energy sum, ncclAllGather of one fp32 per replica); later sets of the attempt reuse the
 *     gathered energies, accepted pairs having traded theirs.  Verdicts are computed on the device, identically on every rank
 *     (Metropolis test and random stream of main.cpp:262-271); coordinates of straddling pairs move by grouped
 *     ncclSend/ncclRecv on the engine's stream; momenta and temperatures stay with the slot.  accepted (n_pair) may be NULL:
 *     then the call enqueues everything and returns without synchronising.
 * With world = 1 the result is bit for bit that of upside_hip_replica_swap_from / _next. */
#define UPSIDE_HIP_COMM_ID_BYTES 128
int upside_hip_comm_get_unique_id(char* id_out /* [128] */);
int upside_hip_comm_init(DerivEngine* engine, int rank, int world, const char* id /* [128] */, const float* temperature_global);
int upside_hip_comm_replica_swap(DerivEngine* engine, int n_pair, const int* pairs_global, uint32_t base_seed, uint64_t round,
                                 int first_set, int* accepted);
/* collective: *first_differing_rank = the lowest rank whose value differs from rank 0's, or -1 when all agree (the command
 * line uses it on the digest of /input/potential, so that a mixed launch ends on every rank together) */
int upside_hip_comm_agree(DerivEngine* engine, unsigned long long value, int* first_differing_rank);
int upside_hip_comm_free(DerivEngine* engine);

/* diagnostics: flags[s] = 1 where system s rebuilt the cached pair list of `node_name` in the last force pass */
int upside_hip_rebuild_flags(DerivEngine* engine, const char* node_name, int* flags);
/* diagnostics: out11 = n1, n2, cap1, cap2, cutoff, cache cutoff, then per-system means of the summed cached list lengths of
   side 1 / side 2 and of the in-range (hit) list lengths of side 1 / side 2, and the sides the MD path walks (bit mask) */
int upside_hip_igraph_stats(DerivEngine* engine, const char* node_name, double* out11);
/* Parity/diagnostic access: the in-range pair list of an interaction-graph node of system `sys` after the
 * last force pass, canonical order of interaction_graph.h:122-157.  Returns n_edge or -1. */
int upside_hip_get_pairlist(DerivEngine* engine, const char* node_name, int sys, int max_edge, int* i1, int* i2);

/* number of BP sweeps of the last rotamer solve per system (rotamer.cpp:1038-1051) */
int upside_hip_rotamer_iterations(DerivEngine* engine, int* iters);

/* last error text of this thread ("" if none) */
/* get_param_deriv (engine_c_library.h:20) for any system of the batch */
int upside_hip_get_param_deriv(DerivEngine* engine, const char* node_name, int system, int n_param, float* deriv);
const char* upside_hip_last_error(void);

/* Per-kernel timing hooks used by bench.py.  With profiling enabled every interaction-graph / BP kernel
 * launch is bracketed by HIP events on the engine's stream.  upside_hip_profile_dump writes one text line per
 * kernel: "<kind>:<node> <total ms> <launches> <total algorithmic bytes> <total pair evaluations>" (bytes as defined in
 * DESIGN.md; pair evaluations = in-range pairs of system 0 x systems, per launch of a pair pass). */
int upside_hip_profile_reset(DerivEngine* engine, int enable);
int upside_hip_profile_dump(DerivEngine* engine, char* buf, int buflen);
/* VALU issue ceilings measured on this device with a known-instruction-count kernel of the pair kernels' launch shape
 * (wave-level fp32 FMA instructions per second: rates[0] dependent scalar chain, rates[1] four independent chains per lane):
 * the denominator of bench.py's roofline.igraph */
int upside_hip_calibrate_valu(double* rates /* [2] */);
/* algorithmic bytes of all interaction graphs for one force evaluation of one system */
double upside_hip_igraph_bytes_per_system(DerivEngine* engine);
/* bytes a belief-propagation launch must move at least once (active pair matrices in, marginals out, node rows), all systems, last solve */
double upside_hip_bp_min_bytes(DerivEngine* engine);

/* The per-interval polynomial image of one quadspline parameter row ([angular 1: ka][angular 2: ka][radial wide: k][radial narrow: k],
 * /root/reference/src/bead_interaction.h:30-84) that the LDS-staged pair passes read (layout: upside_hip_kernels.h, param_poly);
 * host arithmetic, exported so that it can be checked without a GPU.  poly_out: upside_hip_quadspline_poly_width floats. */
int upside_hip_quadspline_poly_width(int n_knot_angular, int n_knot);
int upside_hip_quadspline_poly_row(const float* spline_coeff, int n_knot_angular, int n_knot, float* poly_out);

/* Node types defined outside this library (include/upside_hip_plugin.h): load a shared library whose static initialisers
 * register them (the equivalent of linking another node's .cpp into the reference's libupside.so,
 * /root/reference/src/deriv_engine.h:297-335).  Must be called before the configuration that names the node is opened.
 * The libraries listed in the environment variable UPSIDE_HIP_PLUGINS (':'-separated) are loaded by the first engine
 * construction.  Returns 0 on success, 1 on failure (upside_hip_last_error has the dlopen / registry message). */
int upside_hip_load_plugin(const char* shared_library_path);
/* 1 if a node type is registered under exactly this name prefix (built in or from a plug-in), else 0 */
int upside_hip_node_type_registered(const char* name_prefix);

#ifdef __cplusplus
}
#endif
#endif
