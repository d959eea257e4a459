// C-ABI of libupside_hip.so: the reference's engine_c_library entry points
// (/root/reference/src/engine_c_library.cpp) over the device engine, plus the batched extension declared in
// include/upside_engine_c.h.  No exception crosses the boundary: failures print "ERROR: ..." to stderr and
// return NULL / 1 exactly like engine_c_library.cpp:15-20,37-45.
#include "../../include/upside_engine_c.h"
#include "engine.h"
#include "h5util.h"
#include "spline_fit.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace std;

int engine_pairlist(DerivEngine& e, const string& node_name, int sys, vector<pair<int, int>>& out);
int engine_rotamer_iterations(DerivEngine& e, vector<int>& iters);
int engine_rebuild_flags(DerivEngine& e, const string& node_name, vector<int>& flags);
int engine_igraph_stats(DerivEngine& e, const string& node_name, double* out);
double engine_bp_bytes(DerivEngine& e);
double engine_bp_min_bytes(DerivEngine& e);
double engine_igraph_bytes(DerivEngine& e);
int upside_main_impl(int argc, const char* const* argv, int verbose);

static thread_local string g_last_error;
static int fail(const string& s) { g_last_error = s; fprintf(stderr, "ERROR: %s\n", s.c_str()); return 1; }
#define API_TRY try {
#define API_CATCH(ret) } catch (const string& s) { fail(s); return ret; } catch (const char* s) { fail(s); return ret; } \
    catch (const std::exception& e) { fail(e.what()); return ret; } catch (...) { fail("unknown error"); return ret; }

extern "C" const char* upside_hip_last_error(void) { return g_last_error.c_str(); }
extern "C" int upside_hip_calibrate_valu(double* rates) { API_TRY upk_check(upk_calibrate_valu(rates), "calibrate_valu"); return 0; API_CATCH(1) }
void load_plugin_library(const string& path);   // engine.cpp
extern "C" int upside_hip_load_plugin(const char* path) { API_TRY load_plugin_library(path); return 0; API_CATCH(1) }
extern "C" int upside_hip_node_type_registered(const char* prefix) { API_TRY return node_creation_map().count(prefix) ? 1 : 0; API_CATCH(0) }
extern "C" void upside_hip_set_last_error(const char* msg) { fail(msg); }   // (for the other translation units of the C-ABI)

// ---- construction ----------------------------------------------------------------------------------
extern "C" int upside_hip_set_device(int device) {
    API_TRY hip_check(hipSetDevice(device), "hipSetDevice"); return 0; API_CATCH(1) }
extern "C" DerivEngine* upside_hip_construct(int n_atom, const char* potential_file, int n_system, bool quiet) {
    API_TRY
    if (n_system < 1) throw string("n_system must be positive");
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t f = H5Fopen(potential_file, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f < 0) throw string("unable to open ") + potential_file;
    h5u::Handle config(f, H5Fclose);
    auto potential_group = h5u::open_group(config, "/input/potential");
    return initialize_engine_from_hdf5(n_atom, n_system, (hid_t_compat)(hid_t)potential_group, quiet);
    API_CATCH(nullptr)
}
extern "C" DerivEngine* construct_deriv_engine(int n_atom, const char* potential_file, bool quiet) {
    return upside_hip_construct(n_atom, potential_file, 1, quiet);
}
extern "C" void free_deriv_engine(DerivEngine* engine) { delete engine; }
extern "C" int upside_hip_n_system(DerivEngine* engine) { return engine ? engine->ctx.n_system : 0; }

// ---- positions / momenta ----------------------------------------------------------------------------
static void upload_pos(DerivEngine* e, const float* pos, int n_sys_in) {
    const int S = e->ctx.n_system, na = e->pos->n_atom, st = e->pos->stride;
    vector<float> buf((size_t)S * na * st, 0.f);
    for (int s = 0; s < S; ++s) {
        const float* p = pos + (size_t)(n_sys_in == 1 ? 0 : s) * na * 3;
        for (int a = 0; a < na; ++a) for (int d = 0; d < 3; ++d) buf[((size_t)s * na + a) * st + d] = p[a * 3 + d];
    }
    hip_check(hipMemcpyAsync(e->pos->output.p, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice, e->ctx.stream), "H2D pos");
    e->sync();
}
static void download_rows(DerivEngine* e, const float* dev, int n_elem, int stride, int width, float* out, int n_sys_out) {
    const int S = e->ctx.n_system;
    vector<float> buf((size_t)S * n_elem * stride);
    e->sync();
    hip_check(hipMemcpy(buf.data(), dev, buf.size() * sizeof(float), hipMemcpyDeviceToHost), "D2H");
    for (int s = 0; s < n_sys_out; ++s) for (int i = 0; i < n_elem; ++i) for (int d = 0; d < width; ++d)
        out[((size_t)s * n_elem + i) * width + d] = buf[((size_t)s * n_elem + i) * stride + d];
}

extern "C" int upside_hip_set_pos(DerivEngine* e, const float* pos) { API_TRY upload_pos(e, pos, e->ctx.n_system); e->swap_energy.clear(); return 0; API_CATCH(1) }
extern "C" int upside_hip_get_pos(DerivEngine* e, float* pos) {
    API_TRY download_rows(e, e->pos->output.p, e->pos->n_atom, e->pos->stride, 3, pos, e->ctx.n_system); return 0; API_CATCH(1) }
extern "C" int upside_hip_set_mom(DerivEngine* e, const float* mom) {
    API_TRY
    const int S = e->ctx.n_system, na = e->pos->n_atom;
    vector<float> buf((size_t)S * na * 4, 0.f);
    for (size_t i = 0; i < (size_t)S * na; ++i) for (int d = 0; d < 3; ++d) buf[i * 4 + d] = mom[i * 3 + d];
    hip_check(hipMemcpy(e->mom.p, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice), "H2D mom");
    return 0;
    API_CATCH(1)
}
extern "C" int upside_hip_get_mom(DerivEngine* e, float* mom) {
    API_TRY download_rows(e, e->mom.p, e->pos->n_atom, 4, 3, mom, e->ctx.n_system); return 0; API_CATCH(1) }

// ---- evaluation (engine_c_library.cpp:29-64) ----------------------------------------------------------
extern "C" int upside_hip_compute(DerivEngine* e, float* energy, float* deriv) {
    API_TRY
    e->compute(energy ? PotentialAndDerivMode : DerivMode);
    e->check_device_errors();
    if (energy) { e->fetch_potentials(); for (int s = 0; s < e->ctx.n_system; ++s) energy[s] = e->potential[s]; }
    if (deriv) download_rows(e, e->pos->sens.p, e->pos->n_atom, e->pos->stride, 3, deriv, e->ctx.n_system);
    return 0;
    API_CATCH(1)
}
extern "C" int evaluate_energy(float* energy, DerivEngine* e, const float* pos) {
    API_TRY
    upload_pos(e, pos, 1);
    e->compute(PotentialAndDerivMode);
    e->check_device_errors();
    e->fetch_potentials();
    *energy = e->potential[0];
    return 0;
    API_CATCH(1)
}
extern "C" int evaluate_deriv(float* deriv, DerivEngine* e, const float* pos) {
    API_TRY
    upload_pos(e, pos, 1);
    e->compute(PotentialAndDerivMode);
    e->check_device_errors();
    e->fetch_potentials();
    download_rows(e, e->pos->sens.p, e->pos->n_atom, e->pos->stride, 3, deriv, 1);
    return 0;
    API_CATCH(1)
}

// ---- parameters and node inspection (engine_c_library.cpp:67-193) ---------------------------------------
extern "C" int set_param(int n_param, const float* param, DerivEngine* e, const char* node_name) {
    API_TRY
    e->invalidate_graph();   // cut-offs travel as kernel arguments
    e->get(string(node_name)).computation->set_param(vector<float>(param, param + n_param)); return 0; API_CATCH(1) }
extern "C" int get_param(int n_param, float* param, DerivEngine* e, const char* node_name) {
    API_TRY
    auto v = e->get(string(node_name)).computation->get_param();
    if (v.size() != size_t(n_param)) throw string("Wrong number of parameters, expected ") + to_string(v.size()) + " but got " + to_string(n_param);
    copy(begin(v), end(v), param);
    return 0;
    API_CATCH(1)
}
// engine_c_library.cpp:93-109 as built with -DPARAM_DERIV: system 0 of the batch
static int param_deriv_of(int n_param, float* deriv, DerivEngine* e, const char* node_name, int system) {
    if (system < 0 || system >= e->ctx.n_system) throw string("system index out of range");
    auto v = e->get(string(node_name)).computation->get_param_deriv(system);
    if (v.size() != size_t(n_param)) throw string("Wrong number of parameters, expected ") + to_string(v.size()) + " but got " + to_string(n_param);
    copy(begin(v), end(v), deriv);
    return 0;
}
extern "C" int get_param_deriv(int n_param, float* deriv, DerivEngine* e, const char* node_name) {
    API_TRY
    return param_deriv_of(n_param, deriv, e, node_name, 0); API_CATCH(1) }
extern "C" int upside_hip_get_param_deriv(DerivEngine* e, const char* node_name, int system, int n_param, float* deriv) {
    API_TRY
    return param_deriv_of(n_param, deriv, e, node_name, system); API_CATCH(1) }

extern "C" int get_output_dims(int* n_elem, int* elem_width, DerivEngine* e, const char* node_name) {
    API_TRY
    auto& dc = *e->get(string(node_name)).computation;
    if (dc.potential_term) { *n_elem = 1; *elem_width = 1; }
    else { auto& c = dynamic_cast<CoordNode&>(dc); *n_elem = c.n_elem; *elem_width = c.elem_width; }
    return 0;
    API_CATCH(1)
}
static int get_array(int n_output, float* out, DerivEngine* e, const char* node_name, bool want_sens) {
    API_TRY
    auto& dc = *e->get(string(node_name)).computation;
    if (dc.potential_term) {
        if (n_output != 1) throw string("wrong size for potential node");
        auto& p = dynamic_cast<PotentialNode&>(dc);
        e->sync();
        *out = p.potential_dev.download()[0];
    } else {
        auto& c = dynamic_cast<CoordNode&>(dc);
        if (n_output != c.n_elem * c.elem_width) throw string("wrong size for CoordNode");
        download_rows(e, want_sens ? c.sens.p : c.output.p, c.n_elem, c.stride, c.elem_width, out, 1);
    }
    return 0;
    API_CATCH(1)
}
extern "C" int get_output(int n_output, float* output, DerivEngine* e, const char* node_name) { return get_array(n_output, output, e, node_name, false); }
extern "C" int get_sens(int n_output, float* output, DerivEngine* e, const char* node_name) { return get_array(n_output, output, e, node_name, true); }
extern "C" int get_value_by_name(int n_output, float* output, DerivEngine* e, const char* node_name, const char* log_name) {
    API_TRY
    auto value = e->get(string(node_name)).computation->get_value_by_name(log_name);
    if (n_output != int(value.size()))
        throw string("expected size (") + to_string(n_output) + " elements) inconsistent with actual size (" + to_string(value.size()) + ")";
    copy(begin(value), end(value), output);
    return 0;
    API_CATCH(1)
}

// ---- engine-free spline helpers (engine_c_library.cpp:196-276); host arithmetic as in the reference -------
namespace {
void de_boor(float& val, float& der, const float* c, float x) {   // spline.h:136-174 on the window starting at c[int(x)-1]
    int x_bin = (int)x; float e = x - x_bin; const float* p = c + (x_bin - 1);
    float yu1 = e + 2.f, yu2 = e + 1.f, yu3 = e, f13 = 1.f / 3.f;
    float a11 = f13 * yu1, a12 = f13 * yu2, a13 = f13 * yu3;
    float c11 = (1.f - a11) * p[0] + a11 * p[1], d11 = p[1] - p[0];
    float c12 = (1.f - a12) * p[1] + a12 * p[2], d12 = p[2] - p[1];
    float c13 = (1.f - a13) * p[2] + a13 * p[3], d13 = p[3] - p[2];
    float a22 = 0.5f * yu2, a23 = 0.5f * yu3;
    float c22 = (1.f - a22) * c11 + a22 * c12, d22 = (1.f - a22) * d11 + a22 * d12;
    float c23 = (1.f - a23) * c12 + a23 * c13, d23 = (1.f - a23) * d12 + a23 * d13;
    val = (1.f - yu3) * c22 + yu3 * c23; der = (1.f - yu3) * d22 + yu3 * d23;
}
void clamped_de_boor(float& val, float& der, const float* c, float x, int n, bool strict) {
    bool lo = strict ? x < 1.f : x <= 1.f, hi = (float)(n - 2) <= x;
    if (lo) { val = (1.f / 6.f) * c[0] + (2.f / 3.f) * c[1] + (1.f / 6.f) * c[2]; der = 0.f; return; }
    if (hi) { val = (1.f / 6.f) * c[n - 3] + (2.f / 3.f) * c[n - 2] + (1.f / 6.f) * c[n - 1]; der = 0.f; return; }
    de_boor(val, der, c, x);
}
}  // namespace
extern "C" int clamped_spline_solve(int N, float* bspline_coeff, const float* values) {
    if (N < 3) return 1;
    const vector<double> v(values, values + (N - 2));
    const vector<double> c = tablefit::clamped_control_values_with_ghosts(v.data(), N - 2);
    for (int i = 0; i < N; ++i) bspline_coeff[i] = (float)c[i];
    return 0;
}
extern "C" int clamped_spline_value(int N, float* result, const float* bspline_coeff, int nx, float* x) {
    for (int i = 0; i < nx; ++i) { float d; clamped_de_boor(result[i], d, bspline_coeff, x[i], N, true); }
    return 0;
}
extern "C" int get_clamped_value_and_deriv(int N, float* result, const float* bspline_coeff, int nx, float* x) {
    for (int i = 0; i < nx; ++i) clamped_de_boor(result[i * 2], result[i * 2 + 1], bspline_coeff, x[i], N, false);
    return 0;
}
extern "C" int get_clamped_coeff_deriv(int N, float* result, const float*, float x) {
    for (int i = 0; i < N; ++i) result[i] = 0.f;
    int start; float data[4];
    if (x <= 1.f) { start = 0; data[0] = 1.f / 6.f; data[1] = 2.f / 3.f; data[2] = 1.f / 6.f; data[3] = 0.f; }
    else if (x >= N - 2) { start = N - 4; data[0] = 0.f; data[1] = 1.f / 6.f; data[2] = 2.f / 3.f; data[3] = 1.f / 6.f; }
    else {
        int x_bin = (int)x; start = x_bin - 1; float y = x - x_bin + 1.f;
        for (int i = 0; i < 4; ++i) { float dc[4] = {0.f, 0.f, 0.f, 0.f}; dc[i] = 1.f; float d; de_boor(data[i], d, dc, y); }
    }
    for (int i = 0; i < 4; ++i) result[start + i] = data[i];
    return 0;
}

// ---- MD on the device (main.cpp:515-523, 616-667; thermostat.h:9-12) -------------------------------------
static void thermostat_params(DerivEngine* e, float delta_t) {
    const int S = e->ctx.n_system;
    vector<float> ms(S), ns(S);
    for (int s = 0; s < S; ++s) {
        ms[s] = (float)exp(-delta_t / e->thermostat_timescale);
        ns[s] = sqrtf(e->temperature[s] * (1 - ms[s] * ms[s]));
    }
    e->mom_scale.upload(ms); e->noise_scale.upload(ns);
}
// thermostat temperature of every system from now on (System::set_temperature, main.cpp:107-110; used by simulated annealing)
extern "C" int upside_hip_set_temperature(DerivEngine* e, const float* temperature) {
    API_TRY
    const int S = e->ctx.n_system;
    if ((int)e->noise_scale.n != S) throw string("upside_hip_set_temperature needs upside_hip_init_md first");
    const float delta_t = e->thermostat_interval * 3 * e->dt;
    vector<float> ns(S);
    for (int s = 0; s < S; ++s) {
        e->temperature[s] = temperature[s];
        const float ms = (float)exp(-delta_t / e->thermostat_timescale);
        ns[s] = sqrtf(temperature[s] * (1 - ms * ms));
    }
    e->sync();
    hip_check(hipMemcpy(e->noise_scale.p, ns.data(), S * sizeof(float), hipMemcpyHostToDevice), "H2D");
    return 0;
    API_CATCH(1)
}
extern "C" int upside_hip_init_md(DerivEngine* e, const float* temperature, uint32_t base_seed, float thermostat_timescale, float dt,
                                  int thermostat_interval_rounds) {
    std::vector<uint32_t> seeds(e ? e->ctx.n_system : 0);
    for (size_t s = 0; s < seeds.size(); ++s) seeds[s] = base_seed + (uint32_t)s;   // main.cpp:459
    return upside_hip_init_md_seeds(e, temperature, seeds.data(), thermostat_timescale, dt, thermostat_interval_rounds);
}
extern "C" int upside_hip_init_md_seeds(DerivEngine* e, const float* temperature, const uint32_t* seeds, float thermostat_timescale, float dt,
                                        int thermostat_interval_rounds) {
    API_TRY
    const int S = e->ctx.n_system;
    if (thermostat_interval_rounds < 1) throw string("thermostat interval must be at least one round");
    e->thermostat_timescale = thermostat_timescale; e->dt = dt; e->thermostat_interval = thermostat_interval_rounds;
    for (int s = 0; s < S; ++s) { e->temperature[s] = temperature[s]; e->seeds[s] = seeds[s]; }
    e->seed.upload(e->seeds);
    e->mom.fill_bytes(0);
    e->invalidate_graph();
    e->set_invocations(0); e->round_num = 0; e->stage_num = 0;
    thermostat_params(e, 1e8f);                     // mom_scale = 0: momenta fully resampled (main.cpp:515-522)
    upk_check(upk_thermostat(&e->ctx.L, e->mom.p, e->pos->n_atom, e->seed.p, e->n_invocations_dev.p, e->mom_scale.p, e->noise_scale.p), "thermostat");
    e->n_invocations++;
    e->sync();
    thermostat_params(e, thermostat_interval_rounds * 3 * dt);   // main.cpp:523
    return 0;
    API_CATCH(1)
}
extern "C" int upside_hip_set_integrator(DerivEngine* e, int type) {
    API_TRY
    if (type != 0 && type != 1) throw string("integrator type must be 0 (Verlet) or 1 (Predescu)");
    if (e->stage_num != 0) throw string("an integration cycle is in progress");
    e->sync();
    if (type != e->integrator_type) e->invalidate_graph();      // (the stage weights are launch arguments of the recorded steps)
    e->integrator_type = type;
    return 0;
    API_CATCH(1)
}
static void require_md(DerivEngine* e, const char* who) {
    if ((int)e->noise_scale.n != e->ctx.n_system) throw string(who) + " needs upside_hip_init_md first (the thermostat's seeds and scales are set there)";
}
extern "C" int upside_hip_run_md(DerivEngine* e, int n_round) {
    API_TRY
    require_md(e, "upside_hip_run_md");
    if (e->stage_num != 0) throw string("an integration cycle is in progress (upside_hip_run_steps left it unfinished)");
    e->run_steps(3 * n_round);                             // main.cpp:657-663
    e->check_device_errors();
    return 0;
    API_CATCH(1)
}
extern "C" int upside_hip_run_steps(DerivEngine* e, int n_step) {
    API_TRY
    require_md(e, "upside_hip_run_steps");
    e->run_steps(n_step);
    e->check_device_errors();
    return 0;
    API_CATCH(1)
}
// ---- Monte-Carlo pivot moves (monte_carlo_sampler.cpp; main.cpp:628-630) ---------------------------------
extern "C" int upside_hip_load_mc(DerivEngine* e, const char* config_file) {
    API_TRY
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t f = H5Fopen(config_file, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f < 0) throw string("unable to open ") + config_file;
    h5u::Handle config(f, H5Fclose);
    auto input = h5u::open_group(config, "/input");
    int n = 0;
    e->invalidate_graph();
    if (h5u::exists(input, "pivot_moves")) { e->load_pivot_moves((hid_t_compat)(hid_t)input); n += e->pivot.loaded; }
    if (h5u::exists(input, "jump_moves")) { e->load_jump_moves((hid_t_compat)(hid_t)input); n += e->jump.loaded; }
    return n;                                   // number of samplers loaded
    API_CATCH(-1)
}
extern "C" int upside_hip_mc_step(DerivEngine* e, uint64_t round) {
    API_TRY e->mc_step(round); e->check_device_errors(); return 0; API_CATCH(1)
}
extern "C" int upside_hip_mc_stats(DerivEngine* e, int sampler, int* stats, int reset) {
    API_TRY
    if (sampler != 0 && sampler != 1) throw string("sampler must be 0 (pivot) or 1 (jump)");
    if (sampler == 0 ? !e->pivot.loaded : !e->jump.loaded) throw string("sampler not loaded");
    auto& buf = sampler == 0 ? e->pivot.stats : e->jump.stats;
    e->sync();
    auto st = buf.download();
    for (size_t i = 0; i < st.size(); ++i) stats[i] = st[i];
    if (reset) buf.fill_bytes(0);
    return 0;
    API_CATCH(1)
}
extern "C" int upside_hip_mc_loaded(DerivEngine* e, int sampler) { return e ? (sampler == 0 ? e->pivot.loaded : (sampler == 1 ? e->jump.loaded : 0)) : 0; }
extern "C" int upside_hip_recenter(DerivEngine* e) {
    API_TRY upk_check(upk_recenter(&e->ctx.L, e->pos->coord(), 0), "recenter"); e->sync(); return 0; API_CATCH(1) }
extern "C" int upside_hip_recenter_axes(DerivEngine* e, int xy_only) {
    API_TRY upk_check(upk_recenter(&e->ctx.L, e->pos->coord(), xy_only ? 1 : 0), "recenter"); e->sync(); return 0; API_CATCH(1) }

static int replica_swap_impl(DerivEngine* e, int n_pair, const int* pairs, uint32_t base_seed, uint64_t round, int draw0, int* accepted, bool reuse_energy) {
    const int S = e->ctx.n_system;
    for (int i = 0; i < 2 * n_pair; ++i) if (pairs[i] < 0 || pairs[i] >= S) throw string("invalid system");
    if (reuse_energy) {
        // the energies must be those of THIS attempt: captured in the same round with no force pass (MD, Monte Carlo, another
        // evaluation) and no new coordinates since
        if ((int)e->swap_energy.size() != S || e->swap_energy_round != round || e->swap_energy_compute != e->n_compute)
            throw string("upside_hip_replica_swap_next needs a preceding upside_hip_replica_swap(_from) of the same attempt (same round, nothing evaluated or moved in between)");
    } else {
        e->compute(PotentialAndDerivMode);
        e->fetch_potentials();
        e->swap_energy = e->potential;
        e->swap_energy_round = round; e->swap_energy_compute = e->n_compute;
    }
    if ((int)e->temperature.size() != S) throw string("replica exchange needs the systems' temperatures: call upside_hip_init_md first");
    vector<float> beta(S);
    for (int s = 0; s < S; ++s) beta[s] = 1.f / e->temperature[s];
    DevBuf<float> d_en, d_beta; d_en.upload(e->swap_energy); d_beta.upload(beta);
    DevBuf<int> d_pairs, d_acc; d_pairs.upload(vector<int>(pairs, pairs + 2 * n_pair)); d_acc.alloc(n_pair + 1);
    upk_check(upk_replica_swap(&e->ctx.L, e->pos->coord(), d_en.p, d_beta.p, n_pair, d_pairs.p, base_seed, round, draw0, d_acc.p), "replica_swap");
    e->sync();
    auto acc = d_acc.download();
    for (int i = 0; i <= n_pair; ++i) accepted[i] = acc[i];   // accepted[n_pair] = generator position for the next set
    for (int i = 0; i < n_pair; ++i) if (acc[i]) swap(e->swap_energy[pairs[2 * i]], e->swap_energy[pairs[2 * i + 1]]);   // the coordinates traded places
    return 0;
}
extern "C" int upside_hip_replica_swap_from(DerivEngine* e, int n_pair, const int* pairs, uint32_t base_seed, uint64_t round, int draw0, int* accepted) {
    API_TRY
    return replica_swap_impl(e, n_pair, pairs, base_seed, round, draw0, accepted, false);
    API_CATCH(1)
}
extern "C" int upside_hip_replica_swap_next(DerivEngine* e, int n_pair, const int* pairs, uint32_t base_seed, uint64_t round, int draw0, int* accepted) {
    API_TRY
    return replica_swap_impl(e, n_pair, pairs, base_seed, round, draw0, accepted, true);
    API_CATCH(1)
}
extern "C" int upside_hip_replica_swap(DerivEngine* e, int n_pair, const int* pairs, uint32_t base_seed, uint64_t round, int* accepted) {
    std::vector<int> acc((size_t)n_pair + 1);
    const int rc = upside_hip_replica_swap_from(e, n_pair, pairs, base_seed, round, 0, acc.data());
    if (!rc) for (int i = 0; i < n_pair; ++i) accepted[i] = acc[i];
    return rc;
}

// ---- replica exchange across engines / GPUs (main.cpp:227-275) --------------------------------
// Host arithmetic only: every rank holds the same all-gathered energies and reaches the same verdicts.
namespace {
inline uint32_t h_rotl32(uint32_t x, unsigned n) { return (x << (n & 31)) | (x >> ((32 - n) & 31)); }
void h_threefry4x32_20(uint32_t X[4], const uint32_t key[4]) {   // Random123/threefry.h:110-117,172,296-430
    static const unsigned R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
    uint32_t ks[5]; ks[4] = 0x1BD11BDAu;
    for (int i = 0; i < 4; ++i) { ks[i] = key[i]; ks[4] ^= key[i]; }
    for (int i = 0; i < 4; ++i) X[i] += ks[i];
    for (int r = 0; r < 20; ++r) {
        if (r % 2 == 0) { X[0] += X[1]; X[1] = h_rotl32(X[1], R[r % 8][0]); X[1] ^= X[0]; X[2] += X[3]; X[3] = h_rotl32(X[3], R[r % 8][1]); X[3] ^= X[2]; }
        else            { X[0] += X[3]; X[3] = h_rotl32(X[3], R[r % 8][0]); X[3] ^= X[0]; X[2] += X[1]; X[1] = h_rotl32(X[1], R[r % 8][1]); X[1] ^= X[2]; }
        if (r % 4 == 3) { const int k = r / 4 + 1; for (int i = 0; i < 4; ++i) X[i] += ks[(k + i) % 5]; X[3] += k; }
    }
}
float h_u01(uint32_t in) {   // uniform.hpp:145-179, the product and the sum rounded separately
#pragma clang fp contract(off)
    const float factor = 1.f / 4294967296.f;
    volatile float t = (float)in * factor;
    return t + 0.5f * factor;
}
}  // namespace
// the Metropolis rule of main.cpp:262-272 on given log-Boltzmann differences (any mixture of Hamiltonians): a uniform of the
// round's generator is drawn only for a rejectable pair; accepted[n_pair] = generator position after this set
extern "C" int upside_replica_decide_lboltz(int n_pair, const float* lboltz_diff, uint32_t base_seed, uint64_t round, int draw0, int* accepted) {
    API_TRY
    int draw = draw0;
    for (int p = 0; p < n_pair; ++p) {
        int ok = 1;
        if (lboltz_diff[p] < 0.f) {
            const uint32_t key[4] = {base_seed, 1u /* REPLICA_EXCHANGE_RANDOM_STREAM */, 0u, 0u};
            uint32_t X[4] = {(uint32_t)(round & 0xffffffffu), (uint32_t)(round >> 32), 0u, (uint32_t)draw};
            h_threefry4x32_20(X, key);
            ++draw;
            if (expf(lboltz_diff[p]) < h_u01(X[0])) ok = 0;
        }
        accepted[p] = ok;
    }
    accepted[n_pair] = draw;
    return 0;
    API_CATCH(1)
}
// trade the coordinates of system s1 of engine e1 and system s2 of engine e2 (same atom count; device to device)
extern "C" int upside_hip_swap_between(DerivEngine* e1, int s1, DerivEngine* e2, int s2) {
    API_TRY
    if (!e1 || !e2 || s1 < 0 || s2 < 0 || s1 >= e1->ctx.n_system || s2 >= e2->ctx.n_system) throw string("invalid system");
    if (e1->pos->n_elem != e2->pos->n_elem) throw string("the two systems differ in their number of atoms");
    if (e1 == e2 && s1 == s2) return 0;
    const size_t row = (size_t)e1->pos->n_elem * e1->pos->stride;
    // Both engines have drained their streams (their pending work reads or writes the rows); the three copies go through e1's
    // stream into a staging row the engine keeps (an allocation and three blocking copies per pair before: a rejected pair of a
    // Hamiltonian exchange attempt comes through here twice), and e2 is made to wait for them by the final synchronisation.
    e1->sync(); e2->sync();
    if (e1->swap_row.n < row) e1->swap_row.alloc(row);
    float* a = e1->pos->output.p + (size_t)s1 * row; float* b = e2->pos->output.p + (size_t)s2 * row;
    hipStream_t st = e1->ctx.stream;
    hip_check(hipMemcpyAsync(e1->swap_row.p, a, row * sizeof(float), hipMemcpyDeviceToDevice, st), "D2D");
    hip_check(hipMemcpyAsync(a, b, row * sizeof(float), hipMemcpyDeviceToDevice, st), "D2D");
    hip_check(hipMemcpyAsync(b, e1->swap_row.p, row * sizeof(float), hipMemcpyDeviceToDevice, st), "D2D");
    hip_check(hipStreamSynchronize(st), "hipStreamSynchronize");
    e1->swap_energy.clear(); e2->swap_energy.clear();
    return 0;
    API_CATCH(1)
}
extern "C" int upside_replica_decide(int n_pair, const int* pairs, const float* beta, const float* energy, uint32_t base_seed,
                                     uint64_t round, int draw0, int* accepted) {
    API_TRY
    int draw = draw0;
    for (int p = 0; p < n_pair; ++p) {
        const int s1 = pairs[p * 2], s2 = pairs[p * 2 + 1];
        if (s1 < 0 || s2 < 0) throw string("invalid system");
        // temperature exchange of one Hamiltonian: (new_lboltz[s1]+new_lboltz[s2]) - (old_lboltz[s1]+old_lboltz[s2])
        const float lb = (-beta[s1] * energy[s2] + -beta[s2] * energy[s1]) - (-beta[s1] * energy[s1] + -beta[s2] * energy[s2]);
        int ok = 1;
        if (lb < 0.f) {   // a uniform is drawn only for a rejectable pair (main.cpp:268)
            const uint32_t key[4] = {base_seed, 1u /* REPLICA_EXCHANGE_RANDOM_STREAM */, 0u, 0u};
            uint32_t X[4] = {(uint32_t)(round & 0xffffffffu), (uint32_t)(round >> 32), 0u, (uint32_t)draw};
            h_threefry4x32_20(X, key);
            ++draw;
            if (expf(lb) < h_u01(X[0])) ok = 0;
        }
        accepted[p] = ok;
    }
    accepted[n_pair] = draw;   // generator position for the next swap set of this round
    return 0;
    API_CATCH(1)
}
extern "C" int upside_hip_get_system_pos(DerivEngine* e, int sys, float* pos) {
    API_TRY
    if (sys < 0 || sys >= e->ctx.n_system) throw string("invalid system");
    const int na = e->pos->n_atom, st = e->pos->stride;
    vector<float> buf((size_t)na * st);
    e->sync();
    hip_check(hipMemcpy(buf.data(), e->pos->output.p + (size_t)sys * na * st, buf.size() * sizeof(float), hipMemcpyDeviceToHost), "D2H");
    for (int a = 0; a < na; ++a) for (int d = 0; d < 3; ++d) pos[a * 3 + d] = buf[(size_t)a * st + d];
    return 0;
    API_CATCH(1)
}
extern "C" int upside_hip_set_system_pos(DerivEngine* e, int sys, const float* pos) {
    API_TRY
    if (sys < 0 || sys >= e->ctx.n_system) throw string("invalid system");
    const int na = e->pos->n_atom, st = e->pos->stride;
    vector<float> buf((size_t)na * st, 0.f);
    for (int a = 0; a < na; ++a) for (int d = 0; d < 3; ++d) buf[(size_t)a * st + d] = pos[a * 3 + d];
    e->sync();
    hip_check(hipMemcpy(e->pos->output.p + (size_t)sys * na * st, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice), "H2D");
    return 0;
    API_CATCH(1)
}
extern "C" int upside_hip_swap_system_pairs(DerivEngine* e, int n_pair, const int* pairs) {
    API_TRY
    const int S = e->ctx.n_system;
    if (n_pair <= 0) return 0;
    vector<char> used(S, 0);
    for (int i = 0; i < 2 * n_pair; ++i) {
        if (pairs[i] < 0 || pairs[i] >= S) throw string("invalid system");
        if (used[pairs[i]]) throw string("Overlapping indices in swap set.");
        used[pairs[i]] = 1;
    }
    DevBuf<int> d; d.upload(vector<int>(pairs, pairs + 2 * n_pair));
    upk_check(upk_swap_system_pairs(&e->ctx.L, e->pos->coord(), n_pair, d.p), "swap_system_pairs");
    e->sync();
    return 0;
    API_CATCH(1)
}
extern "C" int upside_hip_swap_systems(DerivEngine* e, int s1, int s2) {
    API_TRY
    const int S = e->ctx.n_system;
    if (s1 < 0 || s1 >= S || s2 < 0 || s2 >= S) throw string("invalid system");
    if (s1 == s2) return 0;
    const size_t n = (size_t)e->pos->n_atom * e->pos->stride;
    DevBuf<float> tmp; tmp.alloc(n);
    float* a = e->pos->output.p + (size_t)s1 * n; float* b = e->pos->output.p + (size_t)s2 * n;
    e->sync();
    hip_check(hipMemcpy(tmp.p, a, n * sizeof(float), hipMemcpyDeviceToDevice), "D2D");
    hip_check(hipMemcpy(a, b, n * sizeof(float), hipMemcpyDeviceToDevice), "D2D");
    hip_check(hipMemcpy(b, tmp.p, n * sizeof(float), hipMemcpyDeviceToDevice), "D2D");
    hip_check(hipDeviceSynchronize(), "sync");
    return 0;
    API_CATCH(1)
}
extern "C" int upside_hip_rebuild_flags(DerivEngine* e, const char* node_name, int* flags) {
    API_TRY
    vector<int> f;
    if (engine_rebuild_flags(*e, node_name, f)) throw string("node has no interaction graph");
    for (size_t i = 0; i < f.size(); ++i) flags[i] = f[i];
    return 0;
    API_CATCH(1)
}
extern "C" int upside_hip_igraph_stats(DerivEngine* e, const char* node_name, double* out11) {
    API_TRY
    if (engine_igraph_stats(*e, node_name, out11)) throw string("node has no interaction graph");
    return 0;
    API_CATCH(1)
}
extern "C" int upside_hip_get_pairlist(DerivEngine* e, const char* node_name, int sys, int max_edge, int* i1, int* i2) {
    API_TRY
    vector<pair<int, int>> pl;
    int n = engine_pairlist(*e, node_name, sys, pl);
    if (n < 0) throw string("node has no interaction graph");
    for (int k = 0; k < n && k < max_edge; ++k) { i1[k] = pl[k].first; i2[k] = pl[k].second; }
    return n;
    API_CATCH(-1)
}
extern "C" int upside_hip_rotamer_iterations(DerivEngine* e, int* iters) {
    API_TRY
    vector<int> it;
    if (engine_rotamer_iterations(*e, it)) throw string("no rotamer node");
    copy(it.begin(), it.end(), iters);
    return 0;
    API_CATCH(1)
}

extern "C" int upside_hip_profile_reset(DerivEngine* e, int enable) {
    API_TRY e->sync(); e->ctx.flush_profile(); e->ctx.families.clear(); e->ctx.profile = enable != 0; return 0; API_CATCH(1) }
extern "C" int upside_hip_profile_dump(DerivEngine* e, char* buf, int buflen) {
    API_TRY
    e->sync(); e->ctx.flush_profile();
    string out;
    char line[512];
    const double bp_bytes = engine_bp_bytes(*e);   // of the last solve; the step-to-step variation is a few per cent
    for (auto& kv : e->ctx.families) {
        if (kv.first.compare(0, 3, "bp:") == 0 && kv.second.bytes == 0.) kv.second.bytes = bp_bytes * kv.second.launches;
        snprintf(line, sizeof(line), "%s %.6f %ld %.1f %.1f\n", kv.first.c_str(), kv.second.ms, kv.second.launches, kv.second.bytes, kv.second.pairs);
        out += line;
    }
    if ((int)out.size() + 1 > buflen) throw string("profile buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
    API_CATCH(1)
}
extern "C" double upside_hip_bp_min_bytes(DerivEngine* e) {
    API_TRY return engine_bp_min_bytes(*e); API_CATCH(-1.)
}
extern "C" double upside_hip_igraph_bytes_per_system(DerivEngine* e) {
    API_TRY return engine_igraph_bytes(*e); API_CATCH(-1.)
}

extern "C" int upside_main(int argc, const char* const* argv, int verbose) {
    API_TRY return upside_main_impl(argc, argv, verbose); API_CATCH(1)
}
