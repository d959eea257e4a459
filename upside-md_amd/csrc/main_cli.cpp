// upside_main for the device engine: the MD loop of /root/reference/src/main.cpp:317-756 for systems that share
// one topology (replicas / ensemble members).  Supported flags are the ones that drive the hot path:
//   --duration --time-step --frame-interval --temperature a,b,c --seed --thermostat-timescale
//   --thermostat-interval --replica-interval --swap-set i-j,k-l (repeatable) --disable-recentering
// Trajectory output to /output (H5Logger, state_logger.h) is not written by this build;
// the per-frame stdout line and the final "us/systems/step" line follow main.cpp:648-654,677-682.
#include "../../include/upside_engine_c.h"
#include "engine.h"
#include "h5util.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

using namespace std;

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
    double duration = -1., frame_interval = -1., time_step = 0.009, thermostat_timescale = 5., thermostat_interval = -1., replica_interval = 0.;
    string temperature_str = "1.0";
    unsigned long seed = 42;
    bool recenter = true;
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
        else if (a == "--re-raise-signal" || a == "--disable-z-recentering") {}
        else if (a == "--log-level" || a == "--monte-carlo-interval" || a == "--anneal-factor" || a == "--anneal-duration" || a == "--set-param") need(a.c_str());
        else if (a.size() && a[0] == '-') throw string("unsupported flag ") + a;
        else files.push_back(a);
    }
    if (duration < 0.) throw string("--duration is required");
    if (frame_interval < 0.) throw string("--frame-interval is required");
    if (files.empty()) throw string("no configuration given");
    const int n_system = (int)files.size();
    const float dt = (float)time_step;
    // intervals in rounds of 3 steps (main.cpp:399-411,445-447)
    const uint64_t n_round = (uint64_t)round(duration / (3 * dt));
    const int frame_rounds = max(1, (int)(frame_interval / (3 * dt)));
    const int thermo_rounds = thermostat_interval <= 0. ? 1 : max(1, (int)(thermostat_interval / (3 * dt)));
    const int replica_rounds = replica_interval > 0. ? max(1, (int)(replica_interval / (3 * dt))) : 0;
    const uint32_t base_seed = (uint32_t)(seed % 4294967291ul);   // main.cpp:403-404

    vector<float> temps;
    for (auto& t : split_string(temperature_str, ",")) temps.push_back((float)stod(t));
    if (temps.size() != 1u && (int)temps.size() != n_system) throw string("Received ") + to_string(temps.size()) + " temperatures but have " + to_string(n_system) + " systems";
    if (temps.size() == 1u) temps.assign(n_system, temps[0]);

    // all systems must share the topology of the first file; only /input/pos differs
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    int n_atom = 0;
    vector<float> all_pos;
    for (int ns = 0; ns < n_system; ++ns) {
        hid_t f = H5Fopen(files[ns].c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
        if (f < 0) throw string("unable to open ") + files[ns];
        h5u::Handle fh(f, H5Fclose);
        vector<hsize_t> dims;
        auto p = h5u::read<float>(f, "/input/pos", 3, &dims);
        if (dims[1] != 3 || dims[2] != 1) throw string("invalid dimensions for initial position");
        if (ns == 0) n_atom = (int)dims[0];
        else if ((int)dims[0] != n_atom) throw string("all systems of one run must share a topology");
        all_pos.insert(all_pos.end(), p.begin(), p.end());
    }
    DerivEngine* e = upside_hip_construct(n_atom, files[0].c_str(), n_system, !verbose);
    if (!e) throw string("unable to construct the engine: ") + upside_hip_last_error();
    struct Guard { DerivEngine* e; ~Guard() { delete e; } } guard{e};
    if (upside_hip_set_pos(e, all_pos.data())) throw string(upside_hip_last_error());
    if (upside_hip_init_md(e, temps.data(), base_seed, (float)thermostat_timescale, dt, thermo_rounds)) throw string(upside_hip_last_error());

    // swap sets (main.cpp:146-219)
    vector<vector<int>> sets;
    for (auto& s : swap_sets) {
        vector<int> prs; set<int> used;
        for (auto& ps : split_string(s, ",")) {
            auto p = split_string(ps, "-");
            if (p.size() != 2u) throw string("invalid swap pair");
            int a = stoi(p[0]), b = stoi(p[1]);
            if (a >= n_system || b >= n_system || a < 0 || b < 0) throw string("invalid system");
            if (used.count(a) || used.count(b) || a == b) throw string("Overlapping indices in swap set.");
            used.insert(a); used.insert(b); prs.push_back(a); prs.push_back(b);
        }
        sets.push_back(prs);
    }

    vector<float> energy(n_system);
    if (upside_hip_compute(e, energy.data(), nullptr)) throw string(upside_hip_last_error());
    if (verbose) for (int ns = 0; ns < n_system; ++ns) printf("%i: Initial potential energy: %.2f\n", ns, energy[ns]);

    auto tstart = chrono::high_resolution_clock::now();
    vector<long> n_attempt(sets.size(), 0), n_success(sets.size(), 0);
    for (uint64_t rnd = 0; rnd < n_round;) {
        uint64_t next = min<uint64_t>(n_round, (rnd / frame_rounds + 1) * (uint64_t)frame_rounds);
        if (replica_rounds) next = min<uint64_t>(next, (rnd / replica_rounds + 1) * (uint64_t)replica_rounds);
        if (upside_hip_run_md(e, (int)(next - rnd))) throw string(upside_hip_last_error());
        rnd = next;
        if (replica_rounds && !(rnd % replica_rounds))
            for (size_t k = 0; k < sets.size(); ++k) {
                vector<int> acc(sets[k].size() / 2 + 1);
                if (upside_hip_replica_swap(e, (int)sets[k].size() / 2, sets[k].data(), base_seed, rnd, acc.data())) throw string(upside_hip_last_error());
                for (size_t i = 0; i < sets[k].size() / 2; ++i) { n_attempt[k]++; n_success[k] += acc[i]; }
            }
        if (!(rnd % frame_rounds)) {
            if (recenter) upside_hip_recenter(e);
            if (upside_hip_compute(e, energy.data(), nullptr)) throw string(upside_hip_last_error());
            if (verbose) for (int ns = 0; ns < n_system; ++ns)
                printf("%*.0f / %*.0f | %5.1f%% | potential % 8.2f\n", 8, rnd * 3 * double(dt), 8, n_round * 3 * double(dt),
                       100. * rnd / double(n_round), energy[ns]);
        }
    }
    e->sync();
    double elapsed = chrono::duration<double>(chrono::high_resolution_clock::now() - tstart).count();
    printf("\n\nfinished in %.1f seconds (%.2f us/systems/step, %.1e simulation_time_unit/hour)\n", elapsed,
           elapsed * 1e6 / n_system / max<uint64_t>(n_round, 1) / 3, n_round * 3 * time_step / elapsed * 3600.);
    for (size_t k = 0; k < sets.size(); ++k) printf("swap set %zu: %ld / %ld accepted\n", k, n_success[k], n_attempt[k]);
    return 0;
}
