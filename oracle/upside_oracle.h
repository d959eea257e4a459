/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, scalar fp32, sequential) of the reference's MD inner loop: the
 * DerivEngine force pass over the README force field plus the leapfrog/Ornstein-Uhlenbeck integrator.
 * It is the CHECKER the HIP product is compared with; nothing in the product path may include, link or
 * call it (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do).
 *
 * Parity is PINNED: tests/test_oracle_vs_golden.py checks this restatement against golden vectors
 * recorded from the unmodified reference compiled from /root/reference/src (oracle/Makefile,
 * tools/make_fixtures.py) and against the Random123 known-answer vectors of SURVEY.md Appendix D.
 *
 * The exported names deliberately repeat the reference's C-ABI
 * (/root/reference/src/engine_c_library.h:12-32) so the same ctypes wrapper drives the reference,
 * this oracle and the product.
 */
#ifndef UPSIDE_ORACLE_H
#define UPSIDE_ORACLE_H
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleEngine DerivEngine;

DerivEngine* construct_deriv_engine(int n_atom, const char* potential_file, bool quiet);
void free_deriv_engine(DerivEngine* engine);
int evaluate_energy(float* energy, DerivEngine* engine, const float* pos);
int evaluate_deriv(float* deriv, DerivEngine* engine, const float* pos);
int set_param(int n_param, const float* param, DerivEngine* engine, const char* node_name);
int get_param_deriv(int n_param, float* deriv, DerivEngine* engine, const char* node_name);
int get_param(int n_param, float* param, DerivEngine* engine, const char* node_name);
int get_output_dims(int* n_elem, int* elem_width, DerivEngine* engine, const char* node_name);
int get_output(int n_output, float* output, DerivEngine* engine, const char* node_name);
int get_sens(int n_output, float* output, DerivEngine* engine, const char* node_name);
int get_value_by_name(int n_output, float* output, DerivEngine* engine, const char* node_name,
                      const char* log_name);
int clamped_spline_solve(int N, float* bspline_coeff, const float* values);
int clamped_spline_value(int N, float* result, const float* bspline_coeff, int nx, float* x);
int get_clamped_value_and_deriv(int N, float* result, const float* bspline_coeff, int nx, float* x);
int get_clamped_coeff_deriv(int N, float* result, const float* bspline_coeff, float x);
int upside_main(int argc, const char* const* argv, int verbose);

/* ---- oracle-only helpers (no reference C-ABI counterpart) ---------------------------------- */

/* Threefry4x32-20 block (Random123/threefry.h:110-117,172,182-183). */
void oracle_threefry4x32(uint32_t out[4], const uint32_t ctr[4], const uint32_t key[4]);
/* RandomGenerator(seed, stream, atom, timestep).normal() -- /root/reference/src/random.h:19-67 */
void oracle_random_normal4(float out[4], uint32_t seed, uint32_t stream, uint32_t atom, uint64_t timestep);
/* ... .uniform_open_closed() on a fresh generator with counter word 3 = draw */
void oracle_random_uniform4(float out[4], uint32_t seed, uint32_t stream, uint32_t atom, uint64_t timestep,
                            uint32_t draw);

/* Pair list of an interaction-graph node after the last evaluate_*: canonical order
 * (interaction_graph.h:122-157).  node_name: "rotamer", "hbond_coverage", "hbond_coverage_hydrophobe",
 * "environment_coverage", "protein_hbond".  Returns n_edge (or -1); fills up to max_edge entries. */
int oracle_get_pairlist(DerivEngine* engine, const char* node_name, int max_edge, int* i1, int* i2);

/* number of BP sweeps of the last rotamer solve (rotamer.cpp:1038-1051 `iter`) */
int oracle_rotamer_iterations(DerivEngine* engine);

/* MD loop of upside_main for one system (main.cpp:515-523, 616-667): thermostat init
 * (T=1, dt=1e8 -> set_temperature -> apply), then n_round integration cycles of 3 leapfrog stages with the
 * thermostat applied every thermostat_interval rounds.  pos (n_atom,3) in/out; mom (n_atom,3) out. */
int oracle_run_md(DerivEngine* engine, float* pos, float* mom, int n_round, float dt, float temperature,
                  uint32_t seed, float thermostat_timescale, int thermostat_interval_rounds);
/* ... with the integrator chosen: 0 = Verlet, 1 = Predescu (deriv_engine.h:230, deriv_engine.cpp:173-180) */
int oracle_run_md_integrator(DerivEngine* engine, float* pos, float* mom, int n_round, float dt, float temperature,
                             uint32_t seed, float thermostat_timescale, int thermostat_interval_rounds, int integrator_type);

#ifdef __cplusplus
}
#endif
#endif
