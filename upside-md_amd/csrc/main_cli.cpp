// upside_main for the device engine: the MD loop of /root/reference/src/main.cpp:317-756 for systems that share
// one topology (replicas / ensemble members).  Supported flags are the ones that drive the hot path:
//   --duration --time-step --frame-interval --temperature a,b,c --seed --thermostat-timescale
//   --thermostat-interval --replica-interval --swap-set i-j,k-l (repeatable) --disable-recentering
// Trajectory output follows the reference's /output layout (state_logger.h:17-141, h5_support.cpp:181-274,
// main.cpp:473-541,594-596): extensible chunked datasets pos (frame,1,n_atom,3) f32, kinetic/potential/temperature
// (frame,1) f64, time (frame) f64, replica_index (frame,1) i32 under replica exchange, attribute `invocation`;
// a frame is taken at the START of every frame_interval-th round (frame 0 = the initial structure), after
// recentering, exactly as main.cpp:633-636.  The per-frame stdout line and the final "us/systems/step" line follow
// main.cpp:648-654,677-682.
#include "../../include/upside_engine_c.h"
#include "engine.h"
#include "h5util.h"
#include <chrono>
#include <csignal>
#include <ctime>
#include <fcntl.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

using namespace std;

// ---- /output logger ------------------------------------------------------------------------------------------
namespace {
struct EArray {   // h5_support.cpp:199-274: extensible along dimension 0, chunked (100 frames), shuffle + fletcher32 (+ deflate 1)
    hid_t dset = -1; hid_t type = -1; vector<hsize_t> row; size_t row_size = 1; vector<char> buffer; size_t elem = 4;
    void create(hid_t group, const char* name, hid_t type_, size_t elem_, vector<hsize_t> row_dims) {
        type = type_; elem = elem_; row = row_dims;
        vector<hsize_t> dims{0}, maxd{H5S_UNLIMITED}, chunk{100};
        for (auto d : row) { dims.push_back(d); maxd.push_back(d); chunk.push_back(d ? d : 1); row_size *= d; }
        hid_t space = H5Screate_simple((int)dims.size(), dims.data(), maxd.data());
        hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
        H5Pset_chunk(dcpl, (int)chunk.size(), chunk.data());
        H5Pset_shuffle(dcpl);
        H5Pset_fletcher32(dcpl);
        if (H5Zfilter_avail(H5Z_FILTER_DEFLATE) > 0) H5Pset_deflate(dcpl, 1);
        dset = H5Dcreate2(group, name, type, space, H5P_DEFAULT, dcpl, H5P_DEFAULT);
        H5Pclose(dcpl); H5Sclose(space);
        if (dset < 0) throw string("unable to create /output/") + name;
    }
    void push(const void* data) { const char* c = (const char*)data; buffer.insert(buffer.end(), c, c + row_size * elem); }
    void flush() {
        if (!row_size) return;      // (a dataset with an empty row shape never receives records)
        const size_t n_rec = buffer.size() / (row_size * elem);
        if (!n_rec) return;
        hid_t space = H5Dget_space(dset);
        vector<hsize_t> dims(1 + row.size());
        H5Sget_simple_extent_dims(space, dims.data(), nullptr); H5Sclose(space);
        vector<hsize_t> start(dims.size(), 0), count = dims;
        start[0] = dims[0]; count[0] = n_rec; dims[0] += n_rec;
        if (H5Dset_extent(dset, dims.data()) < 0) throw string("H5Dset_extent failed");
        hid_t fspace = H5Dget_space(dset);
        hid_t mspace = H5Screate_simple((int)count.size(), count.data(), nullptr);
        H5Sselect_hyperslab(fspace, H5S_SELECT_SET, start.data(), nullptr, count.data(), nullptr);
        const herr_t rc = H5Dwrite(dset, type, mspace, fspace, H5P_DEFAULT, buffer.data());
        H5Sclose(mspace); H5Sclose(fspace);
        if (rc < 0) throw string("H5Dwrite failed");
        buffer.clear();
    }
    void close() { if (dset >= 0) { flush(); H5Dclose(dset); dset = -1; } }
};
struct OutputLogger {   // one per system / configuration file (H5Logger, state_logger.h:70-141)
    hid_t file = -1, group = -1; int n_buffered = 0;
    EArray pos, kinetic, potential, time, temperature, replica_index, replica_cumulative_swaps, pivot_stats, jump_stats; bool log_replica = false, log_pivot = false, log_jump = false;
    // swap_partners: the other system of every swap pair this system takes part in, in swap-set order (main.cpp:203-217:
    // `replica_swap_partner` written once, `replica_cumulative_swaps` (n_success, n_attempt) per pair and frame)
    void open(const string& path, int n_atom, const string& invocation, bool with_replica_index, bool with_pivot, bool with_jump,
              const vector<int>& swap_partners = vector<int>()) {
        file = H5Fopen(path.c_str(), H5F_ACC_RDWR, H5P_DEFAULT);
        if (file < 0) throw string("Unable to open configuration file at ") + path;
        if (H5Lexists(file, "output", H5P_DEFAULT) > 0) H5Ldelete(file, "/output", H5P_DEFAULT);   // main.cpp:473-477
        group = H5Gcreate2(file, "output", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        if (group < 0) throw string("unable to create /output in ") + path;
        {   // write_string_attribute(config, "output", "invocation", ...), main.cpp:490
            hid_t st = H5Tcopy(H5T_C_S1); H5Tset_size(st, invocation.size() + 1);
            hid_t sp = H5Screate(H5S_SCALAR);
            hid_t at = H5Acreate2(group, "invocation", st, sp, H5P_DEFAULT, H5P_DEFAULT);
            if (at >= 0) { H5Awrite(at, st, invocation.c_str()); H5Aclose(at); }
            H5Sclose(sp); H5Tclose(st);
        }
        pos.create(group, "pos", H5T_NATIVE_FLOAT, 4, {1, (hsize_t)n_atom, 3});
        kinetic.create(group, "kinetic", H5T_NATIVE_DOUBLE, 8, {1});
        potential.create(group, "potential", H5T_NATIVE_DOUBLE, 8, {1});
        time.create(group, "time", H5T_NATIVE_DOUBLE, 8, {});
        temperature.create(group, "temperature", H5T_NATIVE_DOUBLE, 8, {1});
        log_replica = with_replica_index;
        if (log_replica) {
            replica_index.create(group, "replica_index", H5T_NATIVE_INT, 4, {1});
            EArray partner; partner.create(group, "replica_swap_partner", H5T_NATIVE_INT, 4, {});
            for (int sp : swap_partners) partner.push(&sp);
            partner.close();
            replica_cumulative_swaps.create(group, "replica_cumulative_swaps", H5T_NATIVE_INT, 4, {(hsize_t)swap_partners.size(), 2});
        }
        log_pivot = with_pivot;
        if (log_pivot) pivot_stats.create(group, "pivot_stats", H5T_NATIVE_INT, 4, {2});   // monte_carlo_sampler.h:33-37
        log_jump = with_jump;
        if (log_jump) jump_stats.create(group, "jump_stats", H5T_NATIVE_INT, 4, {2});
    }
    // the nodes' own per-frame quantities (default_logger->add_logger in the reference's node constructors)
    vector<EArray> extra; vector<LogValue> extra_spec;
    void add_node_loggers(const vector<LogValue>& specs) {
        for (auto& l : specs) {
            if (H5Lexists(group, l.name.c_str(), H5P_DEFAULT) > 0) continue;      // a second node of the same kind: first one wins
            vector<hsize_t> row(l.dims.begin(), l.dims.end());
            extra.emplace_back();
            if (l.as_long) extra.back().create(group, l.name.c_str(), H5T_NATIVE_LONG, sizeof(long), row);
            else extra.back().create(group, l.name.c_str(), H5T_NATIVE_FLOAT, 4, row);
            extra_spec.push_back(l);
        }
    }
    void sample_node_loggers(int system) {
        vector<float> buf; vector<long> lbuf;
        for (size_t i = 0; i < extra.size(); ++i) {
            size_t n = 1; for (auto d : extra_spec[i].dims) n *= d;
            buf.assign(n, 0.f);
            extra_spec[i].fill(system, buf.data());
            if (extra_spec[i].as_long) { lbuf.assign(buf.begin(), buf.end()); extra[i].push(lbuf.data()); }
            else extra[i].push(buf.data());
        }
    }
    void sample(const float* x, double kin, double pot, double t, double temp, int rep, const int* mc, const int* mcj, const int* cum_swaps) {
        pos.push(x); kinetic.push(&kin); potential.push(&pot); time.push(&t); temperature.push(&temp);
        if (log_replica) { replica_index.push(&rep); if (replica_cumulative_swaps.row_size) replica_cumulative_swaps.push(cum_swaps); }
        if (log_pivot) pivot_stats.push(mc);
        if (log_jump) jump_stats.push(mcj);
        if (!(++n_buffered % 100)) flush();                       // state_logger.h:91-92
    }
    void flush() {
        pos.flush(); kinetic.flush(); potential.flush(); time.flush(); temperature.flush(); if (log_replica) { replica_index.flush(); replica_cumulative_swaps.flush(); }
        if (log_pivot) pivot_stats.flush();
        if (log_jump) jump_stats.flush();
        for (auto& x : extra) x.flush();
        if (file >= 0) H5Fflush(file, H5F_SCOPE_LOCAL);
    }
    void close() {
        if (file < 0) return;
        pos.close(); kinetic.close(); potential.close(); time.close(); temperature.close(); replica_index.close(); replica_cumulative_swaps.close(); pivot_stats.close(); jump_stats.close();
        for (auto& x : extra) x.close();
        H5Gclose(group); H5Fclose(file); file = group = -1;
    }
    ~OutputLogger() { try { close(); } catch (...) {} }
};
}  // namespace

// Orderly termination (main.cpp:24-92): SIGINT / SIGTERM stop the run at the next chunk boundary, the buffered frames
// are written, the caller's handlers come back (RAII, the library may be running inside Python), and with
// --re-raise-signal the signal is raised again for the caller.
namespace {
volatile sig_atomic_t g_received_signal = -1;
void note_signal(int signum) { g_received_signal = signum; }
struct SignalGuard {
    int signum; void (*old_handler)(int);
    explicit SignalGuard(int s) : signum(s), old_handler(signal(s, note_signal)) {
        if (old_handler == SIG_ERR) fprintf(stderr, "Warning: problem installing signal handler. Does not affect correctness of simulation.\n");
    }
    ~SignalGuard() { if (old_handler != SIG_ERR) signal(signum, old_handler); }
};
}  // namespace

static vector<string> split_string(const string& src, const string& sep) {
    vector<string> ret;
    size_t pos = 0;
    while (pos <= src.size()) {
        size_t nxt = src.find(sep, pos);
        if (nxt == string::npos) nxt = src.size();
        ret.emplace_back(src.substr(pos, nxt - pos));
        pos = nxt + sep.size();
        if (nxt == src.size()) break;
    }
    return ret;
}

int upside_main_impl(int argc, const char* const* argv, int verbose) {
    const time_t process_start = time(nullptr);      // (of this run: the rendezvous below ignores records older than it)
    double duration = -1., frame_interval = -1., time_step = 0.009, thermostat_timescale = 5., thermostat_interval = -1., replica_interval = 0., mc_interval = 0.;
    string temperature_str = "1.0";
    unsigned long seed = 42;
    bool recenter = true, write_output = true, xy_recenter_only = false;
    int log_level = 1;
    double anneal_factor = 1., anneal_duration = -1.;
    string set_param_file;
    bool re_raise_signal = false, deriv_agreement = false;
    vector<string> swap_sets, files;
    for (int i = 1; i < argc; ++i) {
        string a = argv[i];
        auto need = [&](const char* nm) -> string { if (i + 1 >= argc) throw string("missing value for ") + nm; return argv[++i]; };
        if (a == "--duration") duration = stod(need("--duration"));
        else if (a == "--frame-interval") frame_interval = stod(need("--frame-interval"));
        else if (a == "--time-step") time_step = stod(need("--time-step"));
        else if (a == "--temperature") temperature_str = need("--temperature");
        else if (a == "--seed") seed = stoul(need("--seed"));
        else if (a == "--thermostat-timescale") thermostat_timescale = stod(need("--thermostat-timescale"));
        else if (a == "--thermostat-interval") thermostat_interval = stod(need("--thermostat-interval"));
        else if (a == "--replica-interval") replica_interval = stod(need("--replica-interval"));
        else if (a == "--swap-set") swap_sets.push_back(need("--swap-set"));
        else if (a == "--disable-recentering") recenter = false;
        else if (a == "--no-output") write_output = false;       // extension: leave the configuration files untouched
        else if (a == "--disable-z-recentering") xy_recenter_only = true;   // main.cpp:358-360, 416
        else if (a == "--re-raise-signal") re_raise_signal = true;
        else if (a == "--monte-carlo-interval") mc_interval = stod(need("--monte-carlo-interval"));
        else if (a == "--log-level") {   // main.cpp:474-479: "" = detailed
            const string v = need("--log-level");
            if (v == "basic") log_level = 0; else if (v == "detailed" || v == "") log_level = 1; else if (v == "extensive") log_level = 2;
            else throw string("Illegal value for --log-level");
        }
        else if (a == "--anneal-factor") anneal_factor = stod(need("--anneal-factor"));
        else if (a == "--anneal-duration") anneal_duration = stod(need("--anneal-duration"));
        else if (a == "--set-param") set_param_file = need("--set-param");
        else if (a == "--potential-deriv-agreement") deriv_agreement = true;      // main.cpp:368-372
        else if (a.size() && a[0] == '-') throw string("unsupported flag ") + a;
        else files.push_back(a);
    }
    if (duration < 0.) throw string("--duration is required");
    if (frame_interval < 0.) throw string("--frame-interval is required");
    if (files.empty()) throw string("no configuration given");
    // One process per GPU (started by any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK, e.g. torch.distributed.run, or
    // UPSIDE_HIP_RANK / _WORLD / _LOCAL_RANK): every rank receives the SAME command line; rank r simulates a contiguous
    // block of the configuration files on device LOCAL_RANK and replica exchange runs over RCCL inside the library
    // (upside_hip_comm_*, comm_rccl.cpp).  UPSIDE_HIP_COMM=1 takes the same path with a single process (tests).
    auto env_int_of = [](const char* a, const char* b, int dflt) {
        const char* v = getenv(a); if (!v) v = getenv(b);
        return v ? atoi(v) : dflt; };
    const int world = max(1, env_int_of("UPSIDE_HIP_WORLD", "WORLD_SIZE", 1));
    const int rank = env_int_of("UPSIDE_HIP_RANK", "RANK", 0), local_rank = env_int_of("UPSIDE_HIP_LOCAL_RANK", "LOCAL_RANK", 0);
    const bool use_comm = world > 1 || (getenv("UPSIDE_HIP_COMM") && atoi(getenv("UPSIDE_HIP_COMM")));
    if (rank < 0 || rank >= world) throw string("invalid rank");
    const int n_total = (int)files.size();
    if (n_total % world) throw to_string(n_total) + " systems do not divide over " + to_string(world) + " processes";
    const int n_system = n_total / world, sys_lo = rank * n_system;
    const vector<string> all_files = files;
    files.assign(all_files.begin() + sys_lo, all_files.begin() + sys_lo + n_system);
    if (world > 1 && upside_hip_set_device(local_rank)) throw string(upside_hip_last_error());
    const float dt = (float)time_step;
    // intervals in rounds of 3 steps (main.cpp:399-411,445-447)
    const uint64_t n_round = (uint64_t)round(duration / (3 * dt));
    const int frame_rounds = (int)max(1., round(frame_interval / (3 * dt)));                                   // main.cpp:400
    const int thermo_rounds = thermostat_interval <= 0. ? 1 : (int)max(1., round(thermostat_interval / (3 * dt)));   // main.cpp:399
    const int replica_rounds = replica_interval > 0. ? max(1, (int)(replica_interval / (3 * dt))) : 0;
    const int mc_rounds = mc_interval > 0. ? max(1, (int)(mc_interval / (3 * dt))) : 0;   // main.cpp:411
    const uint32_t base_seed = (uint32_t)(seed % 4294967291ul);   // main.cpp:403-404

    vector<float> temps_global;
    for (auto& t : split_string(temperature_str, ",")) temps_global.push_back((float)stod(t));
    if (temps_global.size() != 1u && (int)temps_global.size() != n_total) throw string("Received ") + to_string(temps_global.size()) + " temperatures but have " + to_string(n_total) + " systems";
    if (temps_global.size() == 1u) temps_global.assign(n_total, temps_global[0]);
    vector<float> temps(temps_global.begin() + sys_lo, temps_global.begin() + sys_lo + n_system);
    if (use_comm && anneal_factor != 1.) throw string("--anneal-factor is not available together with replica exchange across processes");

    if (anneal_duration == -1.) anneal_duration = duration;      // main.cpp:434
    const vector<float> initial_temps = temps;
    // One engine per DISTINCT potential (the reference builds one engine per file, main.cpp:450-571, which is what Hamiltonian
    // replica exchange and mixed runs use): the files are grouped by a digest of /input/potential -- node names, arguments,
    // attributes and dataset bytes alike -- and every group becomes one batched engine.  Only /input/pos (and the output)
    // differ inside a group.  The common case is one group.
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    int n_atom = 0;
    unsigned long long potential_digest = 0;
    vector<float> all_pos;
    vector<unsigned long long> digest_of_group; vector<vector<int>> members; vector<int> group_of(n_system), local_of(n_system);
    for (int ns = 0; ns < n_system; ++ns) {
        hid_t f = H5Fopen(files[ns].c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
        if (f < 0) throw string("unable to open ") + files[ns];
        h5u::Handle fh(f, H5Fclose);
        vector<hsize_t> dims;
        auto p = h5u::read<float>(f, "/input/pos", 3, &dims);
        if (dims[1] != 3 || dims[2] != 1) throw string("invalid dimensions for initial position");
        // (hashed only when there is something to compare with: a single-file run loads whatever the node loader accepts)
        const unsigned long long dg = (n_total > 1) ? h5u::group_digest(h5u::open_group(f, "/input/potential")) : 0ull;
        if (ns == 0) { n_atom = (int)dims[0]; potential_digest = dg; }
        else if ((int)dims[0] != n_atom)      // (replica exchange trades coordinates between any two systems of the run)
            throw string("the systems of one run must have the same number of atoms: ") + files[ns] + " differs from " + files[0];
        size_t g = 0;
        while (g < digest_of_group.size() && digest_of_group[g] != dg) ++g;
        if (g == digest_of_group.size()) { digest_of_group.push_back(dg); members.emplace_back(); }
        group_of[ns] = (int)g; local_of[ns] = (int)members[g].size(); members[g].push_back(ns);
        all_pos.insert(all_pos.end(), p.begin(), p.end());
    }
    const int n_group = (int)members.size();
    if (n_group > 1 && use_comm)
        throw string("systems must share one potential when the run is spread over several processes: /input/potential of ") + files[members[1][0]] +
              " differs from that of " + files[0];
    if (n_group > 1 && mc_interval > 0.) throw string("Monte-Carlo moves are not available in a run that mixes potentials");
    vector<DerivEngine*> engines(n_group, nullptr);
    struct Guard { vector<DerivEngine*>& v; ~Guard() { for (auto* x : v) delete x; } } guard{engines};
    for (int g = 0; g < n_group; ++g) {
        engines[g] = upside_hip_construct(n_atom, files[members[g][0]].c_str(), (int)members[g].size(), !verbose);
        if (!engines[g]) throw string("unable to construct the engine: ") + upside_hip_last_error();
    }
    DerivEngine* e = engines[0];      // the only engine of an ordinary run
    // per-system arrays (n floats each) <-> the engines' own orders
    auto to_group = [&](int g, const float* all, size_t n) { vector<float> v; v.reserve(members[g].size() * n); for (int ns : members[g]) v.insert(v.end(), all + (size_t)ns * n, all + (size_t)(ns + 1) * n); return v; };
    auto from_group = [&](int g, const vector<float>& v, float* all, size_t n) { for (size_t l = 0; l < members[g].size(); ++l) memcpy(all + (size_t)members[g][l] * n, v.data() + l * n, n * sizeof(float)); };
    auto fleet_compute = [&](float* energy_out) {      // force pass of every system, potentials by system
        if (n_group == 1) { if (upside_hip_compute(e, energy_out, nullptr)) throw string(upside_hip_last_error()); return; }
        for (auto* x : engines) x->compute(PotentialAndDerivMode);          // enqueued on the engines' own streams, then collected
        for (int g = 0; g < n_group; ++g) { engines[g]->check_device_errors(); engines[g]->fetch_potentials(); engines[g]->swap_energy.clear(); from_group(g, engines[g]->potential, energy_out, 1); }
    };
    if (!set_param_file.empty()) {   // main.cpp:384-395, 498-499: one 1-D float dataset per node name
        hid_t pf = H5Fopen(set_param_file.c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
        if (pf < 0) throw string("unable to open ") + set_param_file;
        h5u::Handle pfh(pf, H5Fclose);
        for (const string& node_name : h5u::node_names_in_group(pf)) {
            auto values = h5u::read<float>(pf, node_name, 1);
            bool applied = false;      // (a mixed run: every engine that has a node of this name)
            for (auto* x : engines) {
                if (n_group > 1 && x->get_idx(node_name, false) < 0) continue;
                if (set_param((int)values.size(), values.data(), x, node_name.c_str())) throw string("--set-param: ") + upside_hip_last_error();
                applied = true;
            }
            if (!applied) throw string("--set-param: no node named ") + node_name;
        }
    }
    // main.cpp:548-564: recentring would fight a potential that is not translation invariant
    for (auto* x : engines) for (auto& n : x->nodes) {
        auto pre = [&](const char* p) { return n.name == string(p).substr(0, n.name.size()); };   // is_prefix(n.name, p), deriv_engine.cpp:72-74
        if (recenter && !xy_recenter_only && (pre("membrane_potential") || pre("z_flat_bottom") || pre("tension") || pre("AFM")))
            throw string("You have z-centering and a z-dependent potential turned on.  This is not what you want.  "
                         "Consider --disable-z-recentering or --disable-recentering.");
        if (recenter && pre("cavity_radial"))
            throw string("You have re-centering and a radial potential turned on.  This is not what you want.  Consider --disable-recentering.");
    }
    // (thermostat streams are keyed by the GLOBAL system index, main.cpp:459)
    if (n_group == 1) {
        if (upside_hip_set_pos(e, all_pos.data())) throw string(upside_hip_last_error());
        if (upside_hip_init_md(e, temps.data(), base_seed + (uint32_t)sys_lo, (float)thermostat_timescale, dt, thermo_rounds)) throw string(upside_hip_last_error());
    } else for (int g = 0; g < n_group; ++g) {
        auto gp = to_group(g, all_pos.data(), (size_t)n_atom * 3); auto gt = to_group(g, temps.data(), 1);
        vector<uint32_t> seeds; for (int ns : members[g]) seeds.push_back(base_seed + (uint32_t)(sys_lo + ns));
        if (upside_hip_set_pos(engines[g], gp.data())) throw string(upside_hip_last_error());
        if (upside_hip_init_md_seeds(engines[g], gt.data(), seeds.data(), (float)thermostat_timescale, dt, thermo_rounds)) throw string(upside_hip_last_error());
    }

    // swap sets (main.cpp:146-219)
    vector<vector<int>> sets;
    for (auto& s : swap_sets) {
        vector<int> prs; set<int> used;
        for (auto& ps : split_string(s, ",")) {
            auto p = split_string(ps, "-");
            if (p.size() != 2u) throw string("invalid swap pair");
            int a = stoi(p[0]), b = stoi(p[1]);
            if (a >= n_total || b >= n_total || a < 0 || b < 0) throw string("invalid system");
            if (used.count(a) || used.count(b) || a == b) throw string("Overlapping indices in swap set.");
            used.insert(a); used.insert(b); prs.push_back(a); prs.push_back(b);
        }
        sets.push_back(prs);
    }

    if (!use_comm)      // (inside one process the swap sets address its own systems)
        for (auto& st : sets) for (int x : st) if (x >= n_system) throw string("invalid system");
    if (use_comm && !sets.empty()) {
        // Rendezvous: rank 0 creates the communicator id and leaves it in a file every rank of the job can see, stamped with a
        // nonce of THIS launch -- what every rank of one launch shares and another launch does not: the launcher's address and
        // port, its process id, torchrun's run id and restart count (or UPSIDE_HIP_COMM_NONCE).  A rank that finds a record
        // with another nonce (an earlier crashed attempt under the same file name) keeps waiting for this launch's.  Rank 0
        // publishes with an exclusive create + rename and removes the file once every rank has joined.
        string launch;
        bool job_wide = false;      // the nonce is made of variables every rank of the job shares, on whatever node it runs
        if (const char* x = getenv("UPSIDE_HIP_COMM_NONCE")) { launch = x; job_wide = true; }
        else {
            for (const char* v : {"MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT"}) {
                if (getenv(v)) job_wide = true;
                launch += string(getenv(v) ? getenv(v) : "") + "|";
            }
            launch += to_string(world) + "|";
            // the launcher's pid is the same for the ranks of ONE node only: it stands in for the job-wide variables where a launcher
            // exports none (plain RANK / WORLD_SIZE from a shell), never beside them (a multi-node job shares the file, not the agent)
            if (!job_wide) launch += to_string((long)getppid());
        }
        unsigned long long nonce = 1469598103934665603ull;               // FNV-1a
        for (unsigned char ch : launch) { nonce ^= ch; nonce *= 1099511628211ull; }
        string path;
        if (const char* f = getenv("UPSIDE_HIP_COMM_FILE")) path = f;
        else {
            path = string("/tmp/upside_hip_comm_") + (getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0") + "_" + to_string((long)getppid());
            for (const char* v : {"TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT"}) if (const char* x = getenv(v)) path += string("_") + x;
        }
        // Two attempts started one after the other from the same shell share nonce and file name: the record also carries the wall-clock
        // second it was written, and a rank takes only records written after (its own start - UPSIDE_HIP_COMM_SKEW_S, default 20 s:
        // how far apart the ranks of one launch may start) -- what a crashed earlier attempt left behind is older than that.
        struct Record { char id[UPSIDE_HIP_COMM_ID_BYTES]; unsigned long long nonce; long long stamp; } rec;
        memset(&rec, 0, sizeof(rec));
        // (default = the wait window below: a rank that starts late -- a staggered multi-node launch -- or whose node clock lags still accepts
        //  rank 0's record for as long as rank 0 waits for it)
        const int wait_s = getenv("UPSIDE_HIP_COMM_WAIT_S") ? max(1, atoi(getenv("UPSIDE_HIP_COMM_WAIT_S"))) : 120;
        const long long skew_s = getenv("UPSIDE_HIP_COMM_SKEW_S") ? max(0, atoi(getenv("UPSIDE_HIP_COMM_SKEW_S"))) : wait_s;
        if (rank == 0) {
            remove(path.c_str());                                   // a stale record of an earlier attempt under this name
            if (upside_hip_comm_get_unique_id(rec.id)) throw string(upside_hip_last_error());
            rec.nonce = nonce; rec.stamp = (long long)time(nullptr);
            const string tmp = path + ".tmp." + to_string((long)getpid());
            const int fd = open(tmp.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0600);
            if (fd < 0 || write(fd, &rec, sizeof(rec)) != (ssize_t)sizeof(rec)) { if (fd >= 0) close(fd); throw string("cannot write ") + tmp; }
            close(fd);
            if (rename(tmp.c_str(), path.c_str())) { remove(tmp.c_str()); throw string("cannot publish ") + path; }
        } else {
            bool got = false;
            for (int tries = 0; tries < wait_s * 10 && !got; ++tries) {
                FILE* f = fopen(path.c_str(), "rb");
                if (f) { got = fread(&rec, 1, sizeof(rec), f) == sizeof(rec); fclose(f); }
                if (got && rec.nonce != nonce) got = false;           // another launch's record
                if (got && rec.stamp < (long long)process_start - skew_s) got = false;      // an earlier attempt of the same launch line
                if (!got) this_thread::sleep_for(chrono::milliseconds(100));
            }
            if (!got) throw string("no communicator id of this launch at ") + path + " (is rank 0 running?  a multi-node job needs UPSIDE_HIP_COMM_FILE on a "
                                   "shared file system and either a launcher that exports MASTER_ADDR / MASTER_PORT or UPSIDE_HIP_COMM_NONCE; a record older than this "
                                   "rank's start by more than UPSIDE_HIP_COMM_SKEW_S = " + to_string(skew_s) + " s -- late start or clock skew between nodes -- is ignored; "
                                   "waited UPSIDE_HIP_COMM_WAIT_S = " + to_string(wait_s) + " s)";
        }
        if (upside_hip_comm_init(e, rank, world, rec.id, temps_global.data())) throw string(upside_hip_last_error());
        if (rank == 0) remove(path.c_str());                         // (ncclCommInitRank returns when every rank has joined)
        // every rank keeps its own files under ONE engine and the Metropolis kernel assumes one Hamiltonian for the whole ladder:
        // the ranks' potentials must agree as the files of one rank must.  Checked with the communicator, so that every rank
        // learns of a mismatch and none is left waiting for a peer that has gone.
        int differs = -1;
        if (upside_hip_comm_agree(e, potential_digest, &differs)) throw string(upside_hip_last_error());
        if (differs >= 0) {
            upside_hip_comm_free(e);
            // (which of the two has the wrong files the digests cannot tell: with more than two ranks the odd one out is usually it)
            throw string("the configuration files of ranks 0 and ") + to_string(differs) + " hold different /input/potential groups";
        }
    }

    // --potential-deriv-agreement (developer check, main.cpp:279-315 with deriv_engine.cpp:291-342): central differences of the total
    // potential, eps = 1e-3 per coordinate, against the derivative the backward sweep produced; relative RMS deviation per system
    // (relative_rms_deviation, deriv_engine.h:345-357: the finite differences are the reference).  All systems of an engine are displaced
    // in the same coordinate at once: 2 x 3 n_atom force passes per engine whatever the number of systems.
    if (deriv_agreement) {
        const float eps = 1e-3f;
        const size_t n3 = (size_t)n_atom * 3;
        for (int g = 0; g < n_group; ++g) {
            DerivEngine* eg = engines[g];
            const int S = (int)members[g].size();
            vector<float> pos0 = to_group(g, all_pos.data(), n3), work, en((size_t)S), deriv((size_t)S * n3), ep((size_t)S), em((size_t)S);
            if (upside_hip_compute(eg, en.data(), deriv.data())) throw string(upside_hip_last_error());
            if (verbose) {
                printf("Initial potential:\n");
                eg->fetch_potentials();
                for (int l = 0; l < S; ++l) {
                    if (S > 1) printf("%s\n", files[members[g][l]].c_str());
                    for (auto& nd : eg->nodes)
                        if (nd.computation->potential_term) printf("%s: % 4.3f\n", nd.name.c_str(), static_cast<PotentialNode*>(nd.computation.get())->potential[l]);
                    printf("\n\n");
                }
            }
            vector<double> diff2((size_t)S, 0.), ref2((size_t)S, 0.);
            for (size_t ni = 0; ni < n3; ++ni) {
                for (int sign = -1; sign <= 1; sign += 2) {
                    work = pos0;
                    for (int l = 0; l < S; ++l) work[(size_t)l * n3 + ni] = pos0[(size_t)l * n3 + ni] + sign * eps;
                    if (upside_hip_set_pos(eg, work.data()) || upside_hip_compute(eg, sign < 0 ? em.data() : ep.data(), nullptr)) throw string(upside_hip_last_error());
                }
                for (int l = 0; l < S; ++l) {
                    const float fd = (ep[l] - em[l]) * (0.5f / eps);
                    const double d = (double)fd - (double)deriv[(size_t)l * n3 + ni];
                    diff2[l] += d * d; ref2[l] += (double)fd * (double)fd;
                }
            }
            if (upside_hip_set_pos(eg, pos0.data())) throw string(upside_hip_last_error());      // (restore the input: the caller is not surprised)
            for (int l = 0; l < S; ++l) {
                if (verbose) printf("overall potential relative error: ");
                printf(" %.5f", sqrt(diff2[l] / ref2[l]));
                if (verbose) printf("\n");
            }
        }
    }

    vector<float> energy(n_system);
    fleet_compute(energy.data());
    if (verbose) { printf("Initial potential energy:"); for (int ns = 0; ns < n_system; ++ns) printf(" %.2f", energy[ns]); printf("\n"); }

    // one logger per configuration file (the engine has closed its read-only handles by now)
    string invocation;
    for (int i = 0; i < argc; ++i) { if (i) invocation += " "; invocation += argv[i]; }
    vector<OutputLogger> loggers(n_system);
    bool have_mc = false;
    if (mc_rounds) {
        const int n_sampler = upside_hip_load_mc(e, files[0].c_str());
        if (n_sampler < 0) throw string(upside_hip_last_error());
        have_mc = n_sampler > 0;
    }
    const bool have_pivot = have_mc && upside_hip_mc_loaded(e, 0), have_jump = have_mc && upside_hip_mc_loaded(e, 1);
    vector<int> mc_stats((size_t)n_system * 2, 0), mcj_stats((size_t)n_system * 2, 0);
    // the swap pairs every system takes part in (set order, then pair order: main.cpp:176-192) and their running counts
    struct PairRef { int set, pair; };
    vector<vector<PairRef>> participating(n_total);
    vector<vector<int>> pair_success(sets.size()), pair_attempt(sets.size());
    for (size_t k = 0; k < sets.size(); ++k) {
        pair_success[k].assign(sets[k].size() / 2, 0); pair_attempt[k].assign(sets[k].size() / 2, 0);
        for (size_t i = 0; i < sets[k].size() / 2; ++i) { participating[sets[k][2 * i]].push_back(PairRef{(int)k, (int)i}); participating[sets[k][2 * i + 1]].push_back(PairRef{(int)k, (int)i}); }
    }
    if (write_output) for (int ns = 0; ns < n_system; ++ns) {
        vector<int> partners;
        for (auto& pr : participating[sys_lo + ns]) { const int a = sets[pr.set][2 * pr.pair], b = sets[pr.set][2 * pr.pair + 1]; partners.push_back(a != sys_lo + ns ? a : b); }
        loggers[ns].open(files[ns], n_atom, invocation, !sets.empty(), have_pivot, have_jump, partners);
    }
    vector<vector<LogValue>> node_loggers(n_group);      // per engine, in node order, filtered by --log-level (state_logger.h:17-27)
    bool any_node_logger = false;
    if (write_output) {
        for (int g = 0; g < n_group; ++g)
            for (auto& n : engines[g]->nodes) {
                vector<LogValue> v; n.computation->add_loggers(v);
                for (auto& l : v) if (l.level <= log_level) { node_loggers[g].push_back(l); any_node_logger = true; }
            }
        for (int ns = 0; ns < n_system; ++ns) loggers[ns].add_node_loggers(node_loggers[group_of[ns]]);
    }
    vector<int> replica_index(n_total);     // by GLOBAL slot; every rank keeps the whole table (the verdicts are identical everywhere)
    for (int ns = 0; ns < n_total; ++ns) replica_index[ns] = ns;
    vector<float> frame_pos((size_t)n_system * n_atom * 3), frame_mom((size_t)n_system * n_atom * 3);

    auto tstart = chrono::high_resolution_clock::now();
    vector<long> n_attempt(sets.size(), 0), n_success(sets.size(), 0);
    g_received_signal = -1;
    int stop_signal = -1;
    {
    SignalGuard on_int(SIGINT), on_term(SIGTERM);
    uint64_t sync_start = 0;      // round at which the reference's outer loop (main.cpp:616) last started an iteration
    for (uint64_t rnd = 0; rnd < n_round && g_received_signal == -1;) {
        // pivots before the frame of the same round, never at t = 0 (main.cpp:626-630)
        if (have_mc && rnd && !(rnd % mc_rounds)) if (upside_hip_mc_step(e, rnd)) throw string(upside_hip_last_error());
        if (!(rnd % frame_rounds)) {   // main.cpp:633-654: recenter, energy, log, print -- before the round is integrated
            if (recenter) for (auto* x : engines) upside_hip_recenter_axes(x, xy_recenter_only);
            fleet_compute(energy.data());
            if (n_group == 1) { if (upside_hip_get_pos(e, frame_pos.data()) || upside_hip_get_mom(e, frame_mom.data())) throw string(upside_hip_last_error()); }
            else for (int g = 0; g < n_group; ++g) {
                vector<float> gp(members[g].size() * (size_t)n_atom * 3), gm(gp.size());
                if (upside_hip_get_pos(engines[g], gp.data()) || upside_hip_get_mom(engines[g], gm.data())) throw string(upside_hip_last_error());
                from_group(g, gp, frame_pos.data(), (size_t)n_atom * 3); from_group(g, gm, frame_mom.data(), (size_t)n_atom * 3);
            }
            if (have_pivot && upside_hip_mc_stats(e, 0, mc_stats.data(), 1)) throw string(upside_hip_last_error());   // reset per frame
            if (have_jump && upside_hip_mc_stats(e, 1, mcj_stats.data(), 1)) throw string(upside_hip_last_error());
            if (write_output && any_node_logger) {
                for (auto* x : engines) for (auto& n : x->nodes) n.computation->begin_log_frame();
                for (int ns = 0; ns < n_system; ++ns) loggers[ns].sample_node_loggers(local_of[ns]);
                for (auto* x : engines) for (auto& n : x->nodes) n.computation->end_log_frame();
            }
            for (int ns = 0; ns < n_system; ++ns) {
                const float* x = &frame_pos[(size_t)ns * n_atom * 3]; const float* m = &frame_mom[(size_t)ns * n_atom * 3];
                double sum_kin = 0.;
                for (int i = 0; i < n_atom * 3; ++i) sum_kin += (double)(m[i] * m[i]);
                vector<int> cum;
                for (auto& pr : participating[sys_lo + ns]) { cum.push_back(pair_success[pr.set][pr.pair]); cum.push_back(pair_attempt[pr.set][pr.pair]); }
                if (write_output) loggers[ns].sample(x, (0.5 / n_atom) * sum_kin, (double)energy[ns], (double)(3 * dt * (float)rnd) /* fp32 product as main.cpp:540 */, (double)temps[ns], replica_index[sys_lo + ns], &mc_stats[(size_t)ns * 2], &mcj_stats[(size_t)ns * 2], cum.data());
                double com[3] = {0, 0, 0}, rg = 0.;
                for (int i = 0; i < n_atom; ++i) for (int d = 0; d < 3; ++d) com[d] += x[i * 3 + d];
                for (int d = 0; d < 3; ++d) com[d] /= n_atom;
                for (int i = 0; i < n_atom; ++i) for (int d = 0; d < 3; ++d) rg += (x[i * 3 + d] - com[d]) * (x[i * 3 + d] - com[d]);
                if (verbose) {   // the line of main.cpp:649-654, hydrogen-bond count included (get_n_hbond, main.cpp:28-35)
                    double n_hbond = 0.;
                    for (auto& n : engines[group_of[ns]]->nodes) if (dynamic_cast<HBondCounter*>(n.computation.get())) {
                        auto v = n.computation->get_param_deriv(local_of[ns]);      // d(potential)/d(E_protein) = the count
                        if (!v.empty()) n_hbond += v[0];
                    }
                    printf("%*.0f / %*.0f elapsed %2i system %.2f temp %5.1f hbonds, Rg %5.1f A, potential % 8.2f\n", 8, rnd * 3 * double(dt), 8,
                           duration, sys_lo + ns, temps[ns], n_hbond, sqrt(rg / n_atom), energy[ns]);
                }
            }
            fflush(stdout);
        }
        uint64_t next = min<uint64_t>(n_round, (rnd / frame_rounds + 1) * (uint64_t)frame_rounds);
        if (anneal_factor != 1.) {   // main.cpp:436-442, 658-660: the temperature is reset before every thermostat application
            if (!(rnd % thermo_rounds)) {
                const double time = 3 * dt * (float)(rnd + 1);
                const double fraction = max(0., (time - (duration - anneal_duration)) / anneal_duration);
                for (int ns = 0; ns < n_system; ++ns) {
                    const double T0 = initial_temps[ns], T1 = initial_temps[ns] * anneal_factor;
                    const double r = sqrt(T0) * (1. - fraction) + sqrt(T1) * fraction;
                    temps[ns] = (float)(r * r);
                }
                for (int g = 0; g < n_group; ++g) {
                    auto gt = n_group == 1 ? temps : to_group(g, temps.data(), 1);
                    if (upside_hip_set_temperature(engines[g], gt.data())) throw string(upside_hip_last_error());
                }
            }
            next = min<uint64_t>(next, (rnd / thermo_rounds + 1) * (uint64_t)thermo_rounds);
        }
        // the reference leaves its inner loop for an exchange attempt only at a round AFTER the one the loop started with
        // (main.cpp:665: nr > last_start), so with an interval of one round it attempts every second round
        if (replica_rounds) next = min<uint64_t>(next, ((sync_start + 2 + replica_rounds - 1) / replica_rounds) * (uint64_t)replica_rounds);
        if (have_mc) next = min<uint64_t>(next, (rnd / mc_rounds + 1) * (uint64_t)mc_rounds);
        if (n_group == 1) { if (upside_hip_run_md(e, (int)(next - rnd))) throw string(upside_hip_last_error()); }
        else {      // the engines run side by side on their own streams
            for (auto* x : engines) x->run_steps(3 * (int)(next - rnd));
            for (auto* x : engines) x->check_device_errors();
        }
        rnd = next;
        const bool at_sync = replica_rounds && (rnd == n_round || rnd == ((sync_start + 2 + replica_rounds - 1) / replica_rounds) * (uint64_t)replica_rounds);
        if (at_sync) sync_start = rnd;
        if (at_sync && !(rnd % replica_rounds)) {   // main.cpp:667-668; one generator per attempt (main.cpp:249)
            int draw = 0;
            for (size_t k = 0; k < sets.size(); ++k) {
                vector<int> acc(sets[k].size() / 2 + 1);
                // one force evaluation per attempt: the later sets see the energies the accepted pairs traded
                if (n_group > 1) {     // several Hamiltonians: the reference's own procedure (main.cpp:251-273), two energy passes per set
                    const int np = (int)sets[k].size() / 2;
                    auto log_boltzmann = [&]() { vector<float> en(n_system), lb(n_system); fleet_compute(en.data()); for (int i = 0; i < n_system; ++i) lb[i] = -(1.f / temps[i]) * en[i]; return lb; };
                    auto coord_swap = [&](int s1, int s2) {
                        if (upside_hip_swap_between(engines[group_of[s1]], local_of[s1], engines[group_of[s2]], local_of[s2])) throw string(upside_hip_last_error()); };
                    const auto old_lb = log_boltzmann();
                    for (int i = 0; i < np; ++i) coord_swap(sets[k][2 * i], sets[k][2 * i + 1]);
                    const auto new_lb = log_boltzmann();
                    vector<float> diff(np);
                    for (int i = 0; i < np; ++i) { const int s1 = sets[k][2 * i], s2 = sets[k][2 * i + 1]; diff[i] = (new_lb[s1] + new_lb[s2]) - (old_lb[s1] + old_lb[s2]); }
                    if (upside_replica_decide_lboltz(np, diff.data(), base_seed, rnd, draw, acc.data())) throw string(upside_hip_last_error());
                    draw = acc.back();
                    for (int i = 0; i < np; ++i) if (!acc[i]) coord_swap(sets[k][2 * i], sets[k][2 * i + 1]);      // a rejected swap is reversed
                } else if (use_comm) {     // global indices; energies all-gathered, verdicts on the device, straddling pairs over RCCL
                    if (upside_hip_comm_replica_swap(e, (int)sets[k].size() / 2, sets[k].data(), base_seed, rnd, k == 0, acc.data()))
                        throw string(upside_hip_last_error());
                } else {
                    if ((k == 0 ? upside_hip_replica_swap_from : upside_hip_replica_swap_next)(e, (int)sets[k].size() / 2, sets[k].data(), base_seed, rnd, draw, acc.data()))
                        throw string(upside_hip_last_error());
                    draw = acc.back();
                }
                for (size_t i = 0; i < sets[k].size() / 2; ++i) {
                    n_attempt[k]++; n_success[k] += acc[i]; pair_attempt[k][i]++; pair_success[k][i] += acc[i];
                    if (acc[i]) swap(replica_index[sets[k][2 * i]], replica_index[sets[k][2 * i + 1]]);
                }
            }
        }
    }
    for (auto* x : engines) x->sync();
    for (auto& lg : loggers) lg.close();          // buffered frames reach the files also after an early stop (and before the communicator goes)
    if (use_comm) upside_hip_comm_free(e);
    stop_signal = g_received_signal;
    }   // the caller's signal handlers are back
    if (stop_signal != -1) fprintf(stderr, "Received early termination signal\n");
    double elapsed = chrono::duration<double>(chrono::high_resolution_clock::now() - tstart).count();
    printf("\n\nfinished in %.1f seconds (%.2f us/systems/step, %.1e simulation_time_unit/hour)\n", elapsed,
           elapsed * 1e6 / n_system / max<uint64_t>(n_round, 1) / 3, n_round * 3 * time_step / elapsed * 3600.);
    for (size_t k = 0; k < sets.size(); ++k) printf("swap set %zu: %ld / %ld accepted\n", k, n_success[k], n_attempt[k]);
    if (re_raise_signal && stop_signal != -1) raise(stop_signal);     // main.cpp:742-743
    return 0;
}
