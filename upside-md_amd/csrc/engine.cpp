// Host-side engine core: graph construction from HDF5 (deriv_engine.cpp:195-270), the forward/backward
// schedule (deriv_engine.cpp:124-169) and the integration cycle (deriv_engine.cpp:172-192).  Everything here
// only enqueues work on the engine's HIP stream; data stays in HBM.
#include "engine.h"
#include "h5util.h"
#include <dlfcn.h>
#include <algorithm>
#include <cstdlib>
#include <set>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstring>

using namespace std;

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw string("HIP error in ") + what + ": " + hipGetErrorString(e);
}
void upk_check(int code, const char* what) {
    if (code == 0) return;
    if (code >= 9000) {
        const char* why = "";
        switch (code) {      // (the codes of the launchers in kernels_*.hip)
            case 9003: why = ": more than 1024 rotamer nodes"; break;
            case 9004: case 9005: case 9006: why = ": the system does not fit the 160 KB of LDS this kernel stages it in"; break;
            case 9007: case 9008: why = ": side / mode / interaction type outside what the launcher implements"; break;
            case 9009: why = ": more than 65535 rows (row ids are 16 bits wide)"; break;
            case 9010: why = ": the graph's pair lists were allocated with another word width than its kernels read (upk_igraph_t::word16: 16-bit element "
                             "indices for every graph but the rotamer's, <= 65534 elements per side)"; break;
            case 9011: why = ": the hit-list refine exists for the side-chain graph (symmetric, 32-bit words) and for two-sided graphs with 16-bit words; "
                             "a symmetric radial graph walks its cached lists and has no hit lists"; break;
            default: break;
        }
        throw string("kernel launcher ") + what + " rejected its arguments (code " + to_string(code) + ")" + why;
    }
    throw string("HIP launch failure in ") + what + ": " + hipGetErrorString((hipError_t)code);
}

// ---- DeviceCtx profiling --------------------------------------------------------------------------
void DeviceCtx::flush() { upk_check(upk_fuse_flush(&L), "fuse_flush"); }
void DeviceCtx::begin(const std::string& fam) {
    if (!profile) return;
    flush();      // the bracket times the launches made inside it
    hipEvent_t a, b;
    hip_check(hipEventCreate(&a), "hipEventCreate"); hip_check(hipEventCreate(&b), "hipEventCreate");
    hip_check(hipEventRecord(a, stream), "hipEventRecord");
    families[fam].pending.emplace_back(a, b);
}
void DeviceCtx::end(const std::string& fam, double algorithmic_bytes, double pair_evaluations) {
    if (!profile) return;
    flush();
    auto& f = families[fam];
    hip_check(hipEventRecord(f.pending.back().second, stream), "hipEventRecord");
    f.launches += 1; f.bytes += algorithmic_bytes; f.pairs += pair_evaluations;
}
void DeviceCtx::flush_profile() {
    for (auto& kv : families) {
        for (auto& ev : kv.second.pending) {
            hip_check(hipEventSynchronize(ev.second), "hipEventSynchronize");
            float ms = 0.f;
            hip_check(hipEventElapsedTime(&ms, ev.first, ev.second), "hipEventElapsedTime");
            kv.second.ms += ms;
            (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second);
        }
        kv.second.pending.clear();
    }
}

// ---- ScatterPlan -----------------------------------------------------------------------------------
int ScatterPlan::add_source(int n_term, int n_slot, int w, const vector<int>& targets) {
    if (finalized) throw string("scatter plan already finalized");
    if ((int)targets.size() != n_term * n_slot) throw string("scatter source: wrong target table size");
    if (width && width != w) throw string("scatter sources of one node must share a width");
    width = w;
    Source s; s.n_term = n_term; s.n_slot = n_slot; s.width = w; s.targets = targets; s.offset = arena_size;
    arena_size += (long)n_term * n_slot * w;
    sources.push_back(move(s));
    return (int)sources.size() - 1;
}
void ScatterPlan::finalize(int n_target, int n_system) {
    vector<int> count(n_target + 1, 0);
    for (auto& s : sources) for (int t : s.targets) if (t >= 0) { if (t >= n_target) throw string("scatter target out of range"); count[t + 1]++; }
    for (int t = 0; t < n_target; ++t) count[t + 1] += count[t];
    vector<int> fill(count.begin(), count.end() - 1), entry(count[n_target]);
    for (auto& s : sources)
        for (int i = 0; i < (int)s.targets.size(); ++i) {
            int t = s.targets[i];
            if (t >= 0) entry[fill[t]++] = (int)(s.offset + (long)i * s.width);
        }
    csr_start.upload(count); csr_entry.upload(entry);
    arena.alloc((size_t)n_system * (size_t)max(arena_size, 1L));
    // source offsets become per-system-relative pointers: source_ptr(id) + s*arena_size
    finalized = true;
}

CoordNode::CoordNode(DeviceCtx* c, int n_elem_, int elem_width_)
    : DerivComputation(false), n_elem(n_elem_), elem_width(elem_width_), stride(ru(elem_width_)) {
    ctx = c;
    output.alloc((size_t)c->n_system * n_elem * stride);
    sens.alloc((size_t)c->n_system * n_elem * stride);
}
void CoordNode::gather_contributions() {
    if (scatter.sources.empty()) return;
    upk_check(upk_gather_contrib(&ctx->L, scatter.arena.p, scatter.arena_size, scatter.csr_start.p, scatter.csr_entry.p, coord(),
                                 scatter.width, 0), "gather_contrib");
}

// ---- host-fallback nodes (include/upside_hip_plugin.h) ------------------------------------------------------
namespace {
vector<int> identity_targets(int n) { vector<int> v(n); for (int i = 0; i < n; ++i) v[i] = i; return v; }

// all systems of a CoordNode's output (or sens), padding dropped: dst[system][elem][width]
void fetch_dense(DeviceCtx* ctx, const CoordNode& n, const float* dev, vector<float>& stage, vector<float>& dst) {
    const size_t total = (size_t)ctx->n_system * n.n_elem * n.stride;
    stage.resize(total); dst.resize((size_t)ctx->n_system * n.n_elem * n.elem_width);
    if (!total) return;
    hip_check(hipMemcpyAsync(stage.data(), dev, total * sizeof(float), hipMemcpyDeviceToHost, ctx->stream), "D2H");
    hip_check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    for (size_t e = 0; e < (size_t)ctx->n_system * n.n_elem; ++e)
        for (int d = 0; d < n.elem_width; ++d) dst[e * n.elem_width + d] = stage[e * n.stride + d];
}

// din[a][system][elem][width] -> the scatter source `src[a]` of argument a (gathered into its sens in the backward sweep)
void push_arg_derivs(DeviceCtx* ctx, const vector<CoordNode*>& args, const vector<int>& src, const vector<vector<float>>& din) {
    for (size_t a = 0; a < args.size(); ++a) {
        CoordNode& n = *args[a];
        const size_t per_sys = (size_t)n.n_elem * n.elem_width;
        for (int s = 0; s < ctx->n_system && per_sys; ++s)
            hip_check(hipMemcpyAsync(n.scatter.source_ptr(src[a]) + (size_t)s * n.scatter.arena_size, din[a].data() + s * per_sys,
                                     per_sys * sizeof(float), hipMemcpyHostToDevice, ctx->stream), "H2D");
    }
    hip_check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");   // the host vectors may be reused right away
}
}  // namespace

HostPotentialNode::HostPotentialNode(DeviceCtx* c, const vector<CoordNode*>& args_) : PotentialNode(c), args(args_) {
    for (CoordNode* a : args) src_.push_back(a->scatter.add_source(a->n_elem, 1, a->elem_width, identity_targets(a->n_elem)));
    in_.resize(args.size()); din_.resize(args.size()); stage_.resize(args.size());
}
void HostPotentialNode::compute_value(ComputeMode) {
    for (size_t a = 0; a < args.size(); ++a) {
        fetch_dense(ctx, *args[a], args[a]->output.p, stage_[a], in_[a]);
        din_[a].assign(in_[a].size(), 0.f);
    }
    vector<const float*> in(args.size()); vector<float*> din(args.size());
    for (int s = 0; s < ctx->n_system; ++s) {
        for (size_t a = 0; a < args.size(); ++a) {
            const size_t off = (size_t)s * args[a]->n_elem * args[a]->elem_width;
            in[a] = in_[a].data() + off; din[a] = din_[a].data() + off;
        }
        potential[s] = host_potential(s, in, din);
    }
    hip_check(hipMemcpyAsync(potential_dev.p, potential.data(), ctx->n_system * sizeof(float), hipMemcpyHostToDevice, ctx->stream), "H2D");
    push_arg_derivs(ctx, args, src_, din_);
}

HostCoordNode::HostCoordNode(DeviceCtx* c, int n_elem_, int elem_width_, const vector<CoordNode*>& args_)
    : CoordNode(c, n_elem_, elem_width_), args(args_) {
    for (CoordNode* a : args) src_.push_back(a->scatter.add_source(a->n_elem, 1, a->elem_width, identity_targets(a->n_elem)));
    in_.resize(args.size()); din_.resize(args.size()); stage_.resize(args.size());
}
void HostCoordNode::compute_value(ComputeMode) {
    for (size_t a = 0; a < args.size(); ++a) fetch_dense(ctx, *args[a], args[a]->output.p, stage_[a], in_[a]);
    out_.assign((size_t)ctx->n_system * n_elem * stride, 0.f);
    vector<const float*> in(args.size()); vector<float> dense((size_t)n_elem * elem_width);
    for (int s = 0; s < ctx->n_system; ++s) {
        for (size_t a = 0; a < args.size(); ++a) in[a] = in_[a].data() + (size_t)s * args[a]->n_elem * args[a]->elem_width;
        host_value(s, in, dense.data());
        for (int e = 0; e < n_elem; ++e) for (int d = 0; d < elem_width; ++d) out_[((size_t)s * n_elem + e) * stride + d] = dense[(size_t)e * elem_width + d];
    }
    if (!out_.empty()) hip_check(hipMemcpyAsync(output.p, out_.data(), out_.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream), "H2D");
    // the arguments' derivative slots must not keep the previous evaluation's values if the backward sweep is skipped
    hip_check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}
void HostCoordNode::propagate_deriv() {
    vector<float> stage;
    fetch_dense(ctx, *this, sens.p, stage, dout_);
    vector<const float*> in(args.size()); vector<float*> din(args.size());
    for (size_t a = 0; a < args.size(); ++a) din_[a].assign(in_[a].size(), 0.f);
    for (int s = 0; s < ctx->n_system; ++s) {
        for (size_t a = 0; a < args.size(); ++a) {
            const size_t off = (size_t)s * args[a]->n_elem * args[a]->elem_width;
            in[a] = in_[a].data() + off; din[a] = din_[a].data() + off;
        }
        host_deriv(s, in, dout_.data() + (size_t)s * n_elem * elem_width, din);
    }
    push_arg_derivs(ctx, args, src_, din_);
}

// ---- registry (deriv_engine.cpp:50-92, 272-281) ------------------------------------------------------
NodeCreationMap& node_creation_map() {
    static NodeCreationMap m;
    if (!m.size())
        m[string("pos")] = NodeCreationFunction([](DeviceCtx*, hid_t_compat, const ArgList&) -> DerivComputation* {
            throw string("Cannot create pos-type node"); });
    return m;
}
bool is_prefix(const string& s1, const string& s2) { return s1 == s2.substr(0, s1.size()); }
void add_node_creation_function(string name_prefix, NodeCreationFunction fcn) {
    auto& m = node_creation_map();
    for (const auto& kv : m) {
        if (is_prefix(kv.first, name_prefix)) throw string("Internal error.  Type name ") + kv.first + " is a prefix of " + name_prefix + ".";
        if (is_prefix(name_prefix, kv.first)) throw string("Internal error.  Type name ") + name_prefix + " is a prefix of " + kv.first + ".";
    }
    m[name_prefix] = fcn;
}
void check_elem_width_lower_bound(const CoordNode& node, int lb) {
    if (node.elem_width < lb) throw string("expected argument with width at least ") + to_string(lb) + " but received argument with width " + to_string(node.elem_width);
}
void check_elem_width(const CoordNode& node, int w) {
    if (node.elem_width != w) throw string("expected argument with width ") + to_string(w) + " but received argument with width " + to_string(node.elem_width);
}
void check_arguments_length(const ArgList& a, int n) {
    if ((int)a.size() != n) throw string("expected ") + to_string(n) + " arguments but got " + to_string(a.size());
}

// ---- DerivEngine -----------------------------------------------------------------------------------
static void warn_removed_switches();
DerivEngine::DerivEngine(int n_atom, int n_system) {
    warn_removed_switches();
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev < 1) throw string("no HIP device available (this library has no CPU fallback)");
    ctx.n_system = n_system;
    hip_check(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking), "hipStreamCreate");   // (stream priorities were tried: a low-priority upkeep stream delays the joins, -10 to -27 %)
    ctx.L.n_system = n_system; ctx.L.stream = (void*)ctx.stream;
    ctx.L.fuse = upk_fuse_create(n_system);      // queue of fused per-element ops (kernels_basic.hip)
    if (!ctx.L.fuse) throw string("cannot allocate the fused-op table");
    // merged launches (kernels_batch.h) for batches that are chains of dependent launches rather than work: measured on one MI355X,
    // system-steps/s with / without -- 56 residues: 1 system 3.79 k / 3.18 k, 8: 25.9 k / 23.1 k; 300 residues: 1 system 1.83 k / 1.69 k,
    // 64: 56 k / 62 k, 4096: 149 k / 184 k (there the upkeep kernels want their own launch shapes and the side streams).
    // UPSIDE_HIP_BATCH=1 / 0 forces them on / off.  Where they stop winning depends on the protein as much as on the count (a launch of a
    // 20-residue system is latency at 64 systems still): with / without merged launches + graph replay, k system-steps/s --
    //   20 residues: 64 systems 290 / 207, 128: 532 / 407, 256: 729 / 715, 512: 940 / 1111;  56 residues: 32: 127 / 97, 64: 227 / 184,
    //   128: 407 / 357, 256: 571 / 612;  150 residues: 32: 56.5 / 51.1, 48: 79.1 / 74.0, 64: 81.5 / 85.8;  300 residues: 24: 31.7 / 31.2,
    //   32: 35.6 / 35.0 (54.1 / 52.5 at the 7 A list), 64: 56.3 / 62.1
    // -- up to 16 systems always, else while atoms x systems <= 24 000 and systems <= 160.
    { const char* e = getenv("UPSIDE_HIP_BATCH");
      const bool on = e ? atoi(e) != 0 : (n_system <= 16 || ((long)n_system * n_atom <= 24000 && n_system <= 160));
      ctx.L.batch = on ? upk_batch_create() : nullptr; }
    ctx.error_flag.alloc(1);
    potential.assign(n_system, 0.f);
    Node n; n.name = "pos"; n.computation.reset(new Pos(&ctx, n_atom));
    nodes.push_back(move(n));
    pos = dynamic_cast<Pos*>(nodes[0].computation.get());
    mom.alloc((size_t)n_system * n_atom * 4);
    seeds.assign(n_system, 0u); temperature.assign(n_system, 1.f);
}
DerivEngine::~DerivEngine() {
    if (ctx.stream) { (void)hipStreamSynchronize(ctx.stream); }
    if (comm && comm_free) { comm_free(comm); comm = nullptr; }
    invalidate_graph();
    for (auto& kv : side) {
        (void)hipStreamSynchronize(kv.second.stream);
        (void)hipEventDestroy(kv.second.fork); (void)hipEventDestroy(kv.second.join);
    }
    for (auto& kv : side) if (kv.second.owns_stream) (void)hipStreamDestroy(kv.second.stream);
    nodes.clear();
    upk_fuse_destroy(ctx.L.fuse); ctx.L.fuse = nullptr;
    upk_batch_destroy(ctx.L.batch); ctx.L.batch = nullptr;
    if (ctx.stream) (void)hipStreamDestroy(ctx.stream);
}
void DerivEngine::add_node(const string& name, unique_ptr<DerivComputation> fcn, vector<string> argument_names) {
    if (any_of(nodes.begin(), nodes.end(), [&](const Node& n) { return n.name == name; })) throw string("name conflict in DerivEngine");
    Node node; node.name = name; node.computation = move(fcn); node.computation->name = name;
    nodes.push_back(move(node));
    for (auto& nm : argument_names) {
        int parent_idx = get_idx(nm);
        nodes.back().parents.push_back(parent_idx);
        nodes[parent_idx].children.push_back(nodes.size() - 1);
    }
}
DerivEngine::Node& DerivEngine::get(const string& name) {
    auto loc = find_if(begin(nodes), end(nodes), [&](const Node& n) { return n.name == name; });
    if (loc == nodes.end()) throw string("name not found");
    return *loc;
}
int DerivEngine::get_idx(const string& name, bool must_exist) {
    auto loc = find_if(begin(nodes), end(nodes), [&](const Node& n) { return n.name == name; });
    if (must_exist && loc == nodes.end()) throw string("name not found");
    return loc != nodes.end() ? int(loc - begin(nodes)) : -1;
}

void DerivEngine::finalize() {
    for (auto& n : nodes) { n.computation->finalize(); if (!n.computation->capturable()) graph_failed = true; }
    // Unroll the level-synchronous sweep of deriv_engine.cpp:124-169 once; the order of events is static.
    schedule.clear();
    for (auto& n : nodes) n.germ_exec_level = n.deriv_exec_level = -1;
    // Order of the sweep.  Any order that runs a node after its parents (forward) / after its children (backward) computes the
    // same graph; the reference walks it level by level.  Here steps that only enqueue fused per-element ops are drawn together:
    // forward, every such step whose parents are done, then ONE other step, and so on; then the same backward.  The per-element
    // work of a force pass then reaches the device in a handful of launches instead of one per level.  UPSIDE_HIP_SCHEDULE=bfs
    // keeps the reference's level order (tests).
    const char* sched_env = getenv("UPSIDE_HIP_SCHEDULE");
    const bool grouped = !(sched_env && !strcmp(sched_env, "bfs"));
    if (grouped) {
        const size_t N = nodes.size();
        std::vector<char> done(N, 0);
        n_batch_group = 0;
        auto sweep = [&](bool backward) {
            std::fill(done.begin(), done.end(), 0);
            size_t n_pre = 0;
            if (backward)      // a potential term has no backward step (it pushed its derivatives in the forward sweep): nobody waits for it
                for (size_t i = 0; i < N; ++i) if (nodes[i].computation->potential_term) { done[i] = 1; ++n_pre; nodes[i].deriv_exec_level = 0; }
            auto ready = [&](size_t i) {
                const auto& deps = backward ? nodes[i].children : nodes[i].parents;
                return !done[i] && all_of(begin(deps), end(deps), [&](size_t d) { return done[d] != 0; });
            };
            auto fused = [&](size_t i) { auto* c = nodes[i].computation.get(); return backward ? c->fused_backward : c->fused_forward; };
            int lvl = 0;
            for (size_t n_done = n_pre; n_done < N;) {
                // (round by round: the fused steps that are ready NOW do not depend on each other -- their ops run without barriers
                //  between them, on different wavefronts of the system's workgroup -- then the ones they unlock, ...)
                for (;;) {
                    std::vector<size_t> round;
                    for (size_t i = 0; i < N; ++i) if (fused(i) && ready(i)) round.push_back(i);
                    if (round.empty()) break;
                    for (size_t i : round) {
                        schedule.push_back(Step{(int)i, backward}); done[i] = 1; ++n_done;
                        (backward ? nodes[i].deriv_exec_level : nodes[i].germ_exec_level) = lvl;
                    }
                    ++lvl;
                }
                ++lvl;
                // then every other step that is ready: they do not depend on each other, so their kernels may share launches
                // (kernels_batch.h); without merged launches, one at a time
                // A backward step (and the forward step of a potential term) ADDS into the sensitivities of its parents with plain
                // read-modify-writes: two steps share launches only if they have no parent in common
                std::vector<size_t> group;
                for (size_t i = 0; i < N; ++i) if (ready(i)) {
                    bool clash = false;
                    if (backward || nodes[i].computation->potential_term)
                        for (size_t j : group) for (size_t pa : nodes[i].parents)
                            if (std::find(begin(nodes[j].parents), end(nodes[j].parents), pa) != end(nodes[j].parents)) clash = true;
                    if (clash) continue;
                    group.push_back(i);      // (the same order with and without merged launches: the bits of a system's forces do not depend on the batch size)
                }
                const int gid = group.size() > 1 ? n_batch_group++ : -1;
                for (size_t i : group) {
                    Step st{(int)i, backward}; st.batch = gid;
                    schedule.push_back(st); done[i] = 1; ++n_done;
                    (backward ? nodes[i].deriv_exec_level : nodes[i].germ_exec_level) = lvl;
                }
                ++lvl;
            }
        };
        sweep(false); sweep(true);
    } else
    for (int lvl = 0, not_finished = 1;; ++lvl, not_finished = 0) {
        for (size_t i = 0; i < nodes.size(); ++i) {
            auto& n = nodes[i];
            if (n.germ_exec_level == -1) {
                not_finished = 1;
                bool all_parents = all_of(begin(n.parents), end(n.parents), [&](size_t ip) {
                    int l = nodes[ip].germ_exec_level; return l != -1 && l != lvl; });
                if (all_parents) { schedule.push_back(Step{(int)i, false}); n.germ_exec_level = lvl; }
            }
            if (n.deriv_exec_level == -1 && n.germ_exec_level != -1) {
                not_finished = 1;
                bool all_children = all_of(begin(n.children), end(n.children), [&](size_t ip) {
                    int l = nodes[ip].deriv_exec_level; return l != -1 && l != lvl; });
                if (all_children) { schedule.push_back(Step{(int)i, true}); n.deriv_exec_level = lvl; }
            }
        }
        if (!not_finished) break;
    }
    // one launch clears every sensitivity buffer at the start of a force pass
    {
        vector<float*> ptrs; vector<long> sizes;
        for (auto& n : nodes) {
            if (n.computation->potential_term) continue;
            auto* cn = static_cast<CoordNode*>(n.computation.get());
            if (!cn->sens.n) continue;
            ptrs.push_back(cn->sens.p); sizes.push_back((long)cn->sens.n);
        }
        n_zero = (int)ptrs.size();
        zero_ptrs.upload(ptrs); zero_sizes.upload(sizes);
    }
    // hoist prepare() of the nodes that have one to just after the forward step of the last parent it reads, on a
    // side stream
    if (ctx.L.batch && grouped) {
        // Merged launches: the list upkeep of ALL graphs as one group, placed behind the last forward step any of it reads (the
        // graphs' lists do not depend on each other: check / rebuild / refine of the five graphs run side by side in a handful of
        // launches, kernels_batch.h).  On the main stream: the per-element steps it could overlap with are one fused launch now.
        std::vector<size_t> prep;
        for (size_t i = 0; i < nodes.size(); ++i) if (nodes[i].computation->has_prepare()) prep.push_back(i);
        if (!prep.empty()) {
            std::vector<char> is_dep(nodes.size(), 0);
            for (size_t i : prep) {
                auto& named = nodes[i].computation->prepare_deps;
                if (named.empty()) for (size_t ip : nodes[i].parents) is_dep[ip] = 1;
                else for (size_t j = 0; j < nodes.size(); ++j)
                    if (std::find(begin(named), end(named), nodes[j].computation.get()) != end(named)) is_dep[j] = 1;
            }
            size_t last = 0;
            for (size_t k = 0; k < schedule.size(); ++k) if (!schedule[k].backward && is_dep[schedule[k].node]) last = k;
            // (a fused step behind `last` that nobody waits for stays behind the upkeep; the forward steps of the graphs come later anyway)
            std::vector<Step> out(schedule.begin(), schedule.begin() + last + 1);
            const int gid = prep.size() > 1 ? n_batch_group++ : -1;
            for (size_t i : prep) { Step ps{(int)i, false}; ps.prepare = true; ps.batch = gid; out.push_back(ps); }
            for (size_t k = last + 1; k < schedule.size(); ++k) {
                Step st = schedule[k];
                if (!st.backward && nodes[st.node].computation->has_prepare()) st.skip_prepare = true;
                out.push_back(st);
            }
            schedule.swap(out);
        }
        print_schedule();
        return;
    }
    const char* env = getenv("UPSIDE_HIP_ASYNC_PREPARE");
    if (env && atoi(env) == 0) { print_schedule(); return; }
    // nodes a prepare() reads: the ones it names (any node of the graph, e.g. a grandparent whose output a parent copies
    // through), else all its parents
    std::vector<std::vector<size_t>> deps_of(nodes.size());
    for (size_t i = 0; i < nodes.size(); ++i) {
        if (!nodes[i].computation->has_prepare()) continue;
        auto& named = nodes[i].computation->prepare_deps;
        if (named.empty()) deps_of[i] = nodes[i].parents;
        else for (size_t j = 0; j < nodes.size(); ++j)
            if (std::find(begin(named), end(named), nodes[j].computation.get()) != end(named)) deps_of[i].push_back(j);
        std::sort(begin(deps_of[i]), end(deps_of[i]));
        deps_of[i].erase(std::unique(begin(deps_of[i]), end(deps_of[i])), end(deps_of[i]));
    }
    {   // Run the forward steps that the upkeep waits for (and their ancestors) FIRST, everything else after them in
        // the original order: the list rebuilds then start as early as the graph allows and overlap with the
        // forward steps nobody is waiting for (springs, Ramachandran maps, backbone sterics).  The moved set is closed
        // under "parent of", so every step still follows the steps it depends on.
        std::vector<char> unlock(nodes.size(), 0);
        std::function<void(size_t)> mark = [&](size_t i) { if (unlock[i]) return; unlock[i] = 1; for (size_t ip : nodes[i].parents) mark(ip); };
        for (size_t i = 0; i < nodes.size(); ++i) for (size_t d : deps_of[i]) mark(d);
        std::vector<Step> first, rest;
        for (auto& st : schedule) ((!st.backward && unlock[st.node]) ? first : rest).push_back(st);
        first.insert(first.end(), rest.begin(), rest.end());
        schedule.swap(first);
    }
    // One upkeep stream per node: the rebuilds of different graphs run side by side.  (A single shared upkeep stream was an option until
    // round 6 -- UPSIDE_HIP_UPKEEP_STREAMS=1 -- and lost at every batch size since the per-element nodes became a dozen fused launches:
    // 128 systems 86.6 vs 98.3 k, 256: 124.4 vs 138.8 k, 1024: 168.9 vs 177.0 k, 4096: 182.8 vs 188.7 k system-steps/s.)
    std::vector<Step> hoisted;
    std::vector<int> n_dep_left(nodes.size(), -1);
    for (size_t i = 0; i < nodes.size(); ++i) if (nodes[i].computation->has_prepare()) n_dep_left[i] = (int)deps_of[i].size();
    for (auto& st : schedule) {
        hoisted.push_back(st);
        if (st.backward) continue;
        std::vector<size_t> ready;
        for (size_t c = 0; c < nodes.size(); ++c) {
            if (n_dep_left[c] <= 0 || !std::binary_search(begin(deps_of[c]), end(deps_of[c]), (size_t)st.node)) continue;
            if (--n_dep_left[c] == 0) { ready.push_back(c); n_dep_left[c] = -1; }
        }
        // several nodes may become ready at once: upkeep in the order of their forward steps
        std::sort(begin(ready), end(ready), [&](size_t a, size_t b) {
            auto pos = [&](size_t n) { for (size_t k = 0; k < schedule.size(); ++k) if (!schedule[k].backward && schedule[k].node == (int)n) return k; return schedule.size(); };
            return pos(a) < pos(b); });
        for (size_t c : ready) {
            Step ps{(int)c, false}; ps.prepare = true; hoisted.push_back(ps);
            Side sd;
            hip_check(hipStreamCreateWithFlags(&sd.stream, hipStreamNonBlocking), "hipStreamCreate");
            hip_check(hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming), "hipEventCreate");
            hip_check(hipEventCreateWithFlags(&sd.join, hipEventDisableTiming), "hipEventCreate");
            side[(int)c] = sd;
        }
    }
    schedule.swap(hoisted);
    last_prepare_step = -1;
    for (size_t k = 0; k < schedule.size(); ++k) if (schedule[k].prepare) last_prepare_step = (int)k;
    print_schedule();
}
void DerivEngine::print_schedule() {
    if (!getenv("UPSIDE_HIP_PRINT_SCHEDULE")) return;
    for (auto& st : schedule) {
        auto& n = nodes[st.node];
        fprintf(stderr, "%-8s L%-2d batch %-2d %s%-44s parents:", st.prepare ? "prepare" : (st.backward ? "backward" : "forward"),
                st.backward ? n.deriv_exec_level : n.germ_exec_level, st.batch,
                (st.backward ? n.computation->fused_backward : n.computation->fused_forward) && !st.prepare ? "[fused] " : "        ", n.name.c_str());
        for (size_t ip : n.parents) fprintf(stderr, " %s", nodes[ip].name.c_str());
        fprintf(stderr, "\n");
    }
}

void DerivEngine::compute(ComputeMode mode, bool keep_pending) {
    ++n_compute;
    ctx.n_pass = n_compute;      // (nodes that double-buffer by step parity read it: always in step with what a graph capture rolled back)
    // zero sensitivity for later derivative writing (deriv_engine.cpp:147-151), all nodes at once: nothing writes a
    // node's sens before that node's own forward step
    upk_check(upk_zero_many(&ctx.L, zero_ptrs.p, zero_sizes.p, n_zero), "zero_many");
    auto on_stream = [&](hipStream_t st, const std::function<void()>& f) {     // (the fused-op queue is empty on entry; what f queues runs on st)
        hipStream_t main_stream = ctx.stream;
        ctx.stream = st; ctx.L.stream = (void*)st;
        try { f(); ctx.flush(); } catch (...) { ctx.stream = main_stream; ctx.L.stream = (void*)main_stream; throw; }
        ctx.stream = main_stream; ctx.L.stream = (void*)main_stream;
    };
    int open_batch = -1, chain = 0;
    for (size_t k = 0; k < schedule.size(); ++k) {
        const Step& st = schedule[k];
        auto* c = nodes[st.node].computation.get();
        // merged launches: the steps of a group append to their own chains; the group's launches go out at its end
        const int want_batch = (ctx.profile || !ctx.L.batch) ? -1 : st.batch;      // (profiling brackets single launches with events: no merged launches then)
        if (want_batch != open_batch) {
            if (open_batch >= 0) upk_check(upk_batch_end(&ctx.L), "batch_end");
            open_batch = want_batch; chain = 0;
            if (open_batch >= 0) upk_check(upk_batch_begin(&ctx.L), "batch_begin");
        }
        if (open_batch >= 0) {
            if (!c->library_launchers_only) { upk_check(upk_batch_run(&ctx.L), "batch_run"); }
            upk_batch_chain(&ctx.L, chain++);
        }
        if (!c->library_launchers_only) ctx.flush();       // a node that may enqueue work of its own on the stream
        if (st.prepare && side.find(st.node) == side.end()) { c->prepare(); continue; }     // (merged launches: upkeep inline, see finalize)
        if (st.prepare) {   // fork: side stream waits for everything enqueued so far, runs the upkeep, records `join`
            Side& sd = side[st.node];
            ctx.flush();
            hip_check(hipEventRecord(sd.fork, ctx.stream), "hipEventRecord");
            hip_check(hipStreamWaitEvent(sd.stream, sd.fork, 0), "hipStreamWaitEvent");
            on_stream(sd.stream, [&] { c->prepare(); });
            hip_check(hipEventRecord(sd.join, sd.stream), "hipEventRecord");
            continue;
        }
        if (!st.backward) {
            if (c->has_prepare() && !st.skip_prepare) {
                auto it = side.find(st.node);
                if (it != side.end()) { ctx.flush(); hip_check(hipStreamWaitEvent(ctx.stream, it->second.join, 0), "hipStreamWaitEvent"); }
                else c->prepare();
            }
            c->compute_value(mode);
        } else if (!c->potential_term) {
            auto* cn = static_cast<CoordNode*>(c);
            cn->gather_contributions();
            if (!c->library_launchers_only) ctx.flush();       // (the gather is a fused op; the node may read its sens through the stream)
            c->propagate_deriv();
        }
    }
    if (open_batch >= 0) upk_check(upk_batch_end(&ctx.L), "batch_end");
    if (!keep_pending) ctx.flush();       // callers outside the MD loop find the queue empty
}

void DerivEngine::fetch_potentials() {
    sync();
    fill(potential.begin(), potential.end(), 0.f);
    for (auto& n : nodes) {
        if (!n.computation->potential_term) continue;
        auto* p = static_cast<PotentialNode*>(n.computation.get());
        p->potential = p->potential_dev.download();
        for (int s = 0; s < ctx.n_system; ++s) potential[s] += p->potential[s];   // node order, deriv_engine.cpp:143-146
    }
}

void DerivEngine::integration_stage(int stage, float dt_, float max_force) {
    // deriv_engine.cpp:176-177 (integrator from Predescu et al., 2012: double constants narrowed to float there as here)
    const float a = integrator_type == 1 ? (float)0.108991425403425322 : (float)(1. / 6.);
    const float b = integrator_type == 1 ? (float)0.290485609075128726 : (float)(1. / 3.);
    const float mom_update[] = {1.5f - 3.f * a, 1.5f - 3.f * a, 6.f * a};
    const float pos_update[] = {3.f * b, 3.0f - 6.f * b, 3.f * b};
    compute(DerivMode, true);       // the tail of the backward sweep, the leapfrog update and the head of the next force pass share a launch
    upk_check(upk_integration_stage(&ctx.L, mom.p, pos->coord(), dt_ * mom_update[stage], dt_ * pos_update[stage], max_force), "integration_stage");
}
void DerivEngine::integration_cycle(float dt_, float max_force) {
    for (int stage = 0; stage < 3; ++stage) integration_stage(stage, dt_, max_force);
}

void DerivEngine::set_invocations(uint64_t n) {
    n_invocations = n;
    if (!n_invocations_dev.n) n_invocations_dev.alloc(ctx.n_system);
    std::vector<unsigned long long> v(ctx.n_system, (unsigned long long)n);     // one copy per system (kernels_basic.hip: c_thermostat)
    sync();
    hip_check(hipMemcpy(n_invocations_dev.p, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice), "H2D");
}
void DerivEngine::md_step() {
    if (stage_num == 0 && !(round_num % thermostat_interval)) {   // main.cpp:657-662
        upk_check(upk_thermostat(&ctx.L, mom.p, pos->n_atom, seed.p, n_invocations_dev.p, mom_scale.p, noise_scale.p), "thermostat");
        n_invocations++;
    }
    integration_stage(stage_num, dt, 0.f);                        // main.cpp:663
    if (++stage_num == 3) { stage_num = 0; ++round_num; }
    ++steps_done;
}
void DerivEngine::invalidate_graph() {
    md_graph_ready = false;
    if (md_graph_exec) { (void)hipGraphExecDestroy(md_graph_exec); md_graph_exec = nullptr; }
    if (md_graph) { (void)hipGraphDestroy(md_graph); md_graph = nullptr; }
}
bool DerivEngine::capture_md_graph() {
    invalidate_graph();
    const int sn = stage_num; const uint64_t rn = round_num, ni = n_invocations, sd = steps_done, nc = n_compute;
    static const bool debug = getenv("UPSIDE_HIP_GRAPH_DEBUG") != nullptr;
    ctx.flush();
    hipError_t err = hipStreamBeginCapture(ctx.stream, hipStreamCaptureModeGlobal);
    if (err != hipSuccess) { if (debug) fprintf(stderr, "graph: begin capture failed: %s\n", hipGetErrorString(err)); (void)hipGetLastError(); return false; }
    bool ok = true;
    try { for (int i = 0; i < 6; ++i) md_step(); ctx.flush(); }
    catch (const string& e) { ok = false; if (debug) fprintf(stderr, "graph: capture threw: %s\n", e.c_str()); }
    catch (...) { ok = false; }
    hipGraph_t g = nullptr;
    err = hipStreamEndCapture(ctx.stream, &g);
    if (err != hipSuccess || !g) { if (debug) fprintf(stderr, "graph: end capture failed: %s\n", hipGetErrorString(err)); (void)hipGetLastError(); ok = false; }
    if (debug && ok) { size_t nn = 0; (void)hipGraphGetNodes(g, nullptr, &nn); fprintf(stderr, "graph: captured 6 MD steps, %zu nodes\n", nn); }
    stage_num = sn; round_num = rn; n_invocations = ni; steps_done = sd; n_compute = nc;   // nothing ran: capture only records
    if (!ok) { if (g) (void)hipGraphDestroy(g); return false; }
    md_graph = g;
    if (hipGraphInstantiate(&md_graph_exec, md_graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); invalidate_graph(); return false; }
    md_graph_parity = (int)(n_compute & 1);   // the pair-list double buffers alternate with every force pass
    md_graph_ready = true;
    return true;
}
void DerivEngine::run_steps(int n_step) {
    // UPSIDE_HIP_GRAPH: six MD steps replayed from a captured hipGraph.  Default: wherever launches are merged, where a step is a chain of ~17
    // dependent launches and a graph node boundary is cheaper than an eager one (one / eight 56-residue systems: 193 / 223 against
    // 201 / 233 us per step; one 300-residue system 465 against 474; 64 x 150 residues 716 against 689: off there)
    static const int graph_env = [] { const char* e = getenv("UPSIDE_HIP_GRAPH"); return e ? atoi(e) : -1; }();
    static const bool sync_diag = getenv("UPSIDE_HIP_FUSE_TRACE") != nullptr;      // (the trace reads its clocks back after every flush: a capturing stream cannot be synchronised)
    const int use_graph = sync_diag ? 0 : (graph_env >= 0 ? graph_env : (ctx.L.batch ? 1 : 0));        // (with the merged launches: one stream; a multi-stream capture replays slower than it launches)
    int left = n_step;
    while (left > 0) {
        const bool aligned = stage_num == 0 && thermostat_interval == 1 && !ctx.profile && steps_done >= 6;
        if (use_graph && !graph_failed && aligned && left >= 6 && (!md_graph_ready || md_graph_parity == (int)(n_compute & 1))) {
            if (!md_graph_ready && !capture_md_graph()) { graph_failed = true; md_step(); --left; continue; }
            ctx.flush();
            hip_check(hipGraphLaunch(md_graph_exec, ctx.stream), "hipGraphLaunch");
            round_num += 2; n_invocations += 2; steps_done += 6; n_compute += 6; left -= 6;
            continue;
        }
        md_step(); --left;
    }
    ctx.flush();
}
void DerivEngine::load_pivot_moves(hid_t_compat input_group_) {
    const hid_t input = (hid_t)input_group_;
    auto grp = h5u::open_group(input, "pivot_moves");
    vector<hsize_t> d;
    auto pot = h5u::read<float>(grp, "proposal_pot", 3, &d);
    const int n_layer = (int)d[0], n_bin = (int)d[1];
    if (d[2] != d[1]) throw string("proposal_pot must be (n_layer, n_bin, n_bin)");
    auto atoms = h5u::read<int>(grp, "pivot_atom", 2, &d);
    const int n_loc = (int)d[0];
    if (d[1] != 5) throw string("pivot_atom must be (n_pivot_loc, 5)");
    h5u::check_size(grp, "pivot_range", {(size_t)n_loc, 2}); h5u::check_size(grp, "pivot_restype", {(size_t)n_loc});
    auto range = h5u::read<int>(grp, "pivot_range", 2); auto restype = h5u::read<int>(grp, "pivot_restype", 1);
    for (int l = 0; l < n_loc; ++l) {   // monte_carlo_sampler.cpp:49-58
        if (restype[l] < 0 || restype[l] >= n_layer) throw string("invalid pivot restype");
        for (int a = 0; a < 5; ++a) {
            const int at = atoms[l * 5 + a];
            if (at < 0 || at >= pos->n_atom) throw string("pivot_atom out of range");
            if (range[l * 2] <= at && at < range[l * 2 + 1]) throw string("pivot_range cannot contain any atoms in pivot_atom");
        }
        if (range[l * 2] < 0 || range[l * 2 + 1] > pos->n_atom || range[l * 2] > range[l * 2 + 1]) throw string("invalid pivot_range");
    }
    // normalise -log p and build the cdf in double (monte_carlo_sampler.cpp:62-78)
    vector<float> cdf(pot.size());
    const size_t nb2 = (size_t)n_bin * n_bin;
    for (int nl = 0; nl < n_layer; ++nl) {
        double sum_prob = 0.;
        for (size_t i = 0; i < nb2; ++i) { sum_prob += exp(-pot[nl * nb2 + i]); cdf[nl * nb2 + i] = (float)sum_prob; }
        const double inv = 1. / sum_prob, ls = log(sum_prob);
        for (size_t i = 0; i < nb2; ++i) { cdf[nl * nb2 + i] = (float)(cdf[nl * nb2 + i] * inv); pot[nl * nb2 + i] = (float)(pot[nl * nb2 + i] + ls); }
        cdf[(nl + 1) * nb2 - 1] = 1.f;
    }
    const int S = ctx.n_system;
    pivot.atoms.upload(atoms); pivot.range.upload(range); pivot.restype.upload(restype); pivot.pot.upload(pot); pivot.cdf.upload(cdf);
    pivot.stats.alloc((size_t)S * 2); pivot.pos_copy.alloc((size_t)S * pos->n_atom * pos->stride);
    pivot.delta_lprob.alloc(S); pivot.e_old.alloc(S); pivot.e_new.alloc(S);
    pivot.P.n_loc = n_loc; pivot.P.n_bin = n_bin; pivot.P.n_layer = n_layer;
    pivot.P.atoms = pivot.atoms.p; pivot.P.range = pivot.range.p; pivot.P.restype = pivot.restype.p; pivot.P.pot = pivot.pot.p; pivot.P.cdf = pivot.cdf.p;
    pivot.loaded = n_loc > 0;
}
void DerivEngine::load_jump_moves(hid_t_compat input_group_) {
    const hid_t input = (hid_t)input_group_;
    auto grp = h5u::open_group(input, "jump_moves");
    vector<hsize_t> d;
    auto range = h5u::read<int>(grp, "atom_range", 2, &d);
    const int n_chain = (int)d[0];
    if (d[1] != 2) throw string("atom_range must be (n_chain, 2)");
    h5u::check_size(grp, "sigma_trans", {(size_t)n_chain}); h5u::check_size(grp, "sigma_rot", {(size_t)n_chain});
    auto st = h5u::read<float>(grp, "sigma_trans", 1); auto sr = h5u::read<float>(grp, "sigma_rot", 1);
    for (int i = 0; i < n_chain; ++i)
        if (range[i * 2] < 0 || range[i * 2 + 1] > pos->n_atom || range[i * 2] >= range[i * 2 + 1]) throw string("invalid jump atom_range");
    const int S = ctx.n_system;
    jump.range.upload(range); jump.sigma_trans.upload(st); jump.sigma_rot.upload(sr);
    jump.stats.alloc((size_t)S * 2);
    if (!pivot.pos_copy.n) { pivot.pos_copy.alloc((size_t)S * pos->n_atom * pos->stride); pivot.delta_lprob.alloc(S); pivot.e_old.alloc(S); pivot.e_new.alloc(S); }
    jump.J.n_chain = n_chain; jump.J.atom_range = jump.range.p; jump.J.sigma_trans = jump.sigma_trans.p; jump.J.sigma_rot = jump.sigma_rot.p;
    jump.loaded = n_chain > 0;
}
void DerivEngine::mc_step(uint64_t round) {   // MultipleMonteCarloSampler::execute, monte_carlo_sampler.cpp:255-288
    if (!pivot.loaded && !jump.loaded) throw string("no Monte-Carlo moves loaded");
    const size_t S = (size_t)ctx.n_system;
    if (pivot.temperature.n != S) pivot.temperature.alloc(S);          // (allocated once: no free / malloc / device-wide sync per move)
    auto refresh = [&](DevBuf<float>& b, const vector<float>& v) {
        if (b.n != S) b.alloc(S);
        hip_check(hipMemcpyAsync(b.p, v.data(), S * sizeof(float), hipMemcpyHostToDevice, ctx.stream), "H2D");
    };
    refresh(pivot.temperature, temperature);
    swap_energy.clear();                                                // coordinates may move: no swap set can reuse older energies
    for (int sampler = 0; sampler < 2; ++sampler) {
        if (sampler == 0 ? !pivot.loaded : !jump.loaded) continue;
        compute(PotentialAndDerivMode); fetch_potentials();
        refresh(pivot.e_old, potential);
        if (sampler == 0) upk_check(upk_pivot_propose(&ctx.L, pos->coord(), pivot.pos_copy.p, &pivot.P, seed.p, round, pivot.delta_lprob.p), "pivot_propose");
        else upk_check(upk_jump_propose(&ctx.L, pos->coord(), pivot.pos_copy.p, &jump.J, seed.p, round, pivot.delta_lprob.p), "jump_propose");
        compute(PotentialAndDerivMode); fetch_potentials();
        refresh(pivot.e_new, potential);
        // the acceptance uniform is the generator's next draw: the pivot proposal used one, the jump proposal two
        upk_check(upk_mc_accept(&ctx.L, pos->coord(), pivot.pos_copy.p, pivot.e_old.p, pivot.e_new.p, pivot.delta_lprob.p, pivot.temperature.p,
                                seed.p, round, sampler == 0 ? 2 : 3, sampler == 0 ? 1 : 2, (sampler == 0 ? pivot.stats : jump.stats).p), "mc_accept");
        sync();
    }
}
// switches that earlier rounds read and this build does not: said once, so that an A/B script does not compare a variant with itself
static void warn_removed_switches() {
    static bool done = false;
    if (done) return;
    done = true;
    for (const char* v : {"UPSIDE_HIP_BP_RESIDENT", "UPSIDE_HIP_BP_THREADS", "UPSIDE_HIP_BP_LDS_CAP_KB", "UPSIDE_HIP_FUSE_W8", "UPSIDE_HIP_BATCH_KINDS",
                          "UPSIDE_HIP_HB_THREADS", "UPSIDE_HIP_PLR_ROWS_SMALL", "UPSIDE_HIP_SLOT_WGS", "UPSIDE_HIP_ORDER_EVERY", "UPSIDE_HIP_CLEAR_SLOTS",
                          "UPSIDE_HIP_BP_CLUSTER_MIN_C", "UPSIDE_HIP_MAX_SYSTEMS", "UPSIDE_HIP_PAIR2_ENERGY", "UPSIDE_HIP_UPKEEP_STREAMS"})
        if (getenv(v)) fprintf(stderr, "upside_hip: %s is set but no longer read by this build (removed experiment switch, see INTEGRATION.md section 6)\n", v);
}
void DerivEngine::sync() { ctx.flush(); hip_check(hipStreamSynchronize(ctx.stream), "hipStreamSynchronize"); }

void DerivEngine::check_device_errors() {
    sync();
    auto f = ctx.error_flag.download();
    if (f[0]) {
        ctx.error_flag.fill_bytes(0);
        if (f[0] == 7)   // kernels_rotamer.hip: the LAST cluster_barrier (behind the epilogue) gave up waiting; earlier ones hand the system to the one-workgroup solve
            throw string("belief propagation: a workgroup of a solve cluster never arrived (the cluster solve needs all its workgroups "
                         "co-resident; something else occupied the device): the forces of this step are not valid.  Set UPSIDE_HIP_BP_CLUSTER=1 "
                         "to use the one-workgroup solve");
        if (f[0] == 8)   // kernels_rotamer.hip: RotGradOp2::flush
            throw string("side-chain gradient: a bead's gradient is not finite (NaN or overflow in the pair pass): the forces of this step are not valid");
        if (f[0] == 9)   // kernels_rotamer.hip: k_rotamer_bp, layout of the message inbox
            throw string("belief propagation: 3-float message rows were laid out for a solve that does not hold its inbox in LDS (internal error): "
                         "the forces of this step are not valid");
        if (f[0] == 4)   // kernels_basic.hip: c_backbone_pairs
            throw string("backbone sterics: a residue has more neighbours within cutoff + skin than its cached row holds: raise "
                         "UPSIDE_HIP_BACKBONE_LIST_CAP (default 128) or set UPSIDE_HIP_BACKBONE_LIST=0");
        throw string("device capacity overflow (code ") + to_string(f[0]) +
            "): raise UPSIDE_HIP_NBR_CAP / UPSIDE_HIP_SLOT_FACTOR (1 = neighbour list, 2 = residue-pair slots, 3 = node adjacency)";
    }
}

// ---- graph construction (deriv_engine.cpp:195-270) ---------------------------------------------------
// external node types (include/upside_hip_plugin.h): their static initialisers call add_node_creation_function
void load_plugin_library(const string& path) {
    static set<string> loaded;
    if (loaded.count(path)) return;
    // RTLD_LOCAL: the plug-in finds this library through its own DT_NEEDED entry (it must be linked against libupside_hip.so);
    // a global load would also promote this library's C++ symbols -- add_node_creation_function, node_creation_map: the
    // reference's own names -- to the global scope, where a later-loaded libupside.so would bind to them
    if (!dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL)) throw string("cannot load plug-in ") + path + ": " + dlerror();
    loaded.insert(path);
}
static void load_plugins_from_env() {
    static bool done = false;
    if (done) return;
    done = true;
    const char* e = getenv("UPSIDE_HIP_PLUGINS");
    if (!e) return;
    string all(e);
    for (size_t b = 0; b <= all.size();) {
        size_t c = all.find(':', b); if (c == string::npos) c = all.size();
        if (c > b) load_plugin_library(all.substr(b, c - b));
        b = c + 1;
    }
}

DerivEngine* initialize_engine_from_hdf5(int n_atom, int n_system, hid_t_compat potential_group_, bool quiet) {
    (void)quiet;
    load_plugins_from_env();
    hid_t potential_group = (hid_t)potential_group_;
    unique_ptr<DerivEngine> engine(new DerivEngine(n_atom, n_system));
    auto& m = node_creation_map();

    map<string, pair<bool, vector<string>>> dep_graph;
    dep_graph["pos"] = make_pair(true, vector<string>());
    for (const auto& name : h5u::node_names_in_group(potential_group))
        dep_graph[name] = make_pair(true, h5u::attr_strings(potential_group, name, "arguments"));
    for (auto& kv : dep_graph)
        for (auto& dep_name : kv.second.second)
            if (dep_graph.find(dep_name) == end(dep_graph))
                throw string("Node ") + kv.first + " takes " + dep_name + " as an argument, but no node of that name can be found.";

    vector<string> topo_order;
    auto in_topo = [&](const string& name) { return find(begin(topo_order), end(topo_order), name) != end(topo_order); };
    int graph_size = dep_graph.size();
    for (int round_num = 0; round_num < graph_size; ++round_num)
        for (auto it = begin(dep_graph); it != end(dep_graph); ++it) {
            if (!it->second.first) continue;
            if (all_of(begin(it->second.second), end(it->second.second), in_topo)) { topo_order.push_back(it->first); it->second.first = false; }
        }
    for (auto& kv : dep_graph) if (kv.second.first) throw string("Unsatisfiable dependency ") + kv.first + " in potential computation";

    for (auto& nm : topo_order) {
        if (nm == "pos") continue;
        string node_type_name = "";
        for (auto& kv : m) if (is_prefix(kv.first, nm)) node_type_name = kv.first;
        if (node_type_name == "") throw string("No node type found for name '") + nm + "'";
        NodeCreationFunction& node_func = m[node_type_name];
        auto argument_names = dep_graph[nm].second;
        ArgList arguments;
        for (const auto& arg_name : argument_names) {
            arguments.push_back(dynamic_cast<CoordNode*>(engine->get(arg_name).computation.get()));
            if (!arguments.back()) throw arg_name + " is not an intermediate value, but it is an argument of " + nm;
        }
        try {
            auto grp = h5u::open_group(potential_group, nm);
            auto computation = unique_ptr<DerivComputation>(node_func(&engine->ctx, (hid_t_compat)(hid_t)grp, arguments));
            engine->add_node(nm, move(computation), argument_names);
        } catch (const string& e) {
            throw "while adding '" + nm + "', " + e;
        }
    }
    engine->finalize();
    return engine.release();
}
