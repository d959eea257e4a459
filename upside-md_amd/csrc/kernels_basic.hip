// gfx950 kernels for the per-element nodes of the Upside force pass and for the integrator.
// One lane owns one element (atom / residue / virtual site / placed bead); grid.y is the system index.
// Scatter-adds of the reference are replaced by per-term contribution buffers that the parent node
// gathers deterministically (upk_gather_contrib) -- no float atomics anywhere in the force pass.
#include "device_math.h"
#include "../../include/upside_hip_kernels.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <initializer_list>
#include <type_traits>
#include <unordered_map>
#include <vector>

using namespace up;

#define UPK_BLOCK 256
#define ST(L) ((hipStream_t)(L)->stream)
// (an empty node -- e.g. a sequence without hydrogen-bond donors -- still gets one workgroup: a zero grid is a launch error)
static inline dim3 grid1(int n, int S) { return dim3((unsigned)(n > 0 ? (n + UPK_BLOCK - 1) / UPK_BLOCK : 1), (unsigned)S, 1); }
static inline int launch_status() { return (int)hipGetLastError(); }

#define C_OUT(c, s)  ((c).out  + (size_t)(s) * (c).n_elem * (c).stride)
#define C_SENS(c, s) ((c).sens + (size_t)(s) * (c).n_elem * (c).stride)

// ------------------------------------------------------------------------------------------------
// Fused per-element passes.  A force pass holds ~45 per-element steps (coordinate nodes, bonded terms, derivative gathers, the
// integrator) of a few hundred to a few thousand elements per system each.  As kernels of their own they cost a dependent launch
// apiece: 4.9 us of host time per eager launch on this platform, 1.7 us per node of a replayed hipGraph -- against 0.25-0.7 us for
// the same dependent step as a PHASE of a resident workgroup (tools/ubench/launch_chain.hip).  The launchers below therefore do not
// launch: they append an op record (kind + the argument block the kernel used to take) to the engine's queue
// (upk_launch_t::fuse), and the queue is run by ONE launch of k_fused_list -- one workgroup per system walks the ops in order, a
// workgroup barrier between consecutive ops; what an op writes for a later one travels through global memory inside one CU
// (L2 hits; workgroup-scope visibility needs no cache maintenance on gfx950).  Everything that is not fusable (pair passes,
// list upkeep, belief propagation, memory copies, events) flushes the queue first, so program order is the order of effects.
// Op records live in a device table that only grows: an op is registered the first time its (kind, arguments) are seen (an
// upload through the launch stream, drained) and referred to by a 16-bit id afterwards; a launch takes the ids of its ops as a kernel argument, so the
// steady state has no copies and can be captured into a hipGraph.
// UPSIDE_HIP_FUSE=0: every op is launched on its own (one workgroup per system), the order of effects is the same.
#define FUSE_PAYLOAD 232
#define FUSE_MAX_PENDING 224
struct FusedOp { int kind, n, lds_bytes, flags; unsigned char payload[FUSE_PAYLOAD]; };   // flags bit 0: no barrier needed in front of this op
struct FusedIds { int n; unsigned short id[FUSE_MAX_PENDING], rot[FUSE_MAX_PENDING]; };    // rot: first lane of the op's element loop
// What an op reads and writes, so that the queue knows where a workgroup barrier is needed: ops that touch disjoint data run without one
// (every wavefront walks the op list at its own pace, and the latencies of independent ops overlap).  A region is the per-system
// slice [lo + s * stride, + len) of a device array (lo = NULL: an absent optional buffer); FUSE_ALL: everything; FUSE_LDS: the workgroup's LDS scratch.
struct FuseRegion { const char* lo; size_t len, stride; bool write; };
#define FUSE_LDS ((const char*)1)
#define FUSE_ALL ((const char*)2)
static inline FuseRegion r_slice(const void* p, size_t len_bytes, size_t stride_bytes, bool w) { FuseRegion r = {(const char*)p, len_bytes, stride_bytes, w}; return r; }
static inline FuseRegion r_buf(const void* p, size_t per_system_bytes, bool w) { return r_slice(p, per_system_bytes, per_system_bytes, w); }
static inline FuseRegion r_out(const upk_coord_t& c, bool w) { return r_buf(c.out, (size_t)c.n_elem * c.stride * sizeof(float), w); }
static inline FuseRegion r_sens(const upk_coord_t& c, bool w) { return r_buf(c.sens, (size_t)c.n_elem * c.stride * sizeof(float), w); }
static inline FuseRegion r_all() { return r_slice(FUSE_ALL, 0, 0, true); }
static inline FuseRegion r_lds() { return r_slice(FUSE_LDS, 0, 0, true); }
static int fuse_submit_raw(const upk_launch_t* L, int kind, const void* args, size_t bytes, int n, int lds_bytes, const FuseRegion* regs, int n_regs);
// An op is looked up by the BYTES of its argument block, padding included: launchers fill a zeroed block field by field (FARGS),
// nested structs through cz / pz, which rebuild them member by member (a struct copy may carry the source's padding along).
template <typename A>
static inline int fuse_submit(const upk_launch_t* L, int kind, const A& args, int n, std::initializer_list<FuseRegion> regs, int lds_bytes = 0) {
    static_assert(sizeof(A) <= FUSE_PAYLOAD, "fused-op argument block too large");
    static_assert(std::is_trivially_copyable<A>::value, "fused-op arguments are copied bytewise");
    return fuse_submit_raw(L, kind, &args, sizeof(A), n, lds_bytes, regs.begin(), (int)regs.size());
}
#define FARGS(T, a) T a; memset((void*)&a, 0, sizeof(a))
static inline upk_coord_t cz(const upk_coord_t& c) {
    upk_coord_t r; memset((void*)&r, 0, sizeof(r));
    r.out = c.out; r.sens = c.sens; r.n_elem = c.n_elem; r.width = c.width; r.stride = c.stride;
    return r;
}
static inline upk_placement_t pz(const upk_placement_t& p) {
    upk_placement_t r; memset((void*)&r, 0, sizeof(r));
    r.n_elem = p.n_elem; r.n_pos_dim = p.n_pos_dim; r.n_sig = p.n_sig; r.sig[0] = p.sig[0]; r.sig[1] = p.sig[1]; r.sig[2] = p.sig[2];
    r.affine_residue = p.affine_residue; r.layer = p.layer; r.rama_residue = p.rama_residue; r.is_rama = p.is_rama;
    r.fixed_data = p.fixed_data; r.spline_coeff = p.spline_coeff; r.nx = p.nx; r.ny = p.ny;
    return r;
}
enum { FOP_ZERO_MANY = 1, FOP_REDUCE_SUM, FOP_GATHER_CONTRIB, FOP_INTEGRATION_STAGE, FOP_THERMOSTAT, FOP_AFFINE_FWD, FOP_AFFINE_BWD,
       FOP_RAMA_FWD, FOP_RAMA_BWD, FOP_INFER_FWD, FOP_INFER_BWD, FOP_SPRING, FOP_PLACEMENT_FWD, FOP_PLACEMENT_BWD, FOP_RAMA_MAP_POT,
       FOP_WEIGHTED_POS_FWD, FOP_WEIGHTED_POS_BWD, FOP_NONLINEAR_COUPLING, FOP_HBOND_ENERGY, FOP_PROTEIN_HBOND_FINISH,
       FOP_PROTEIN_HBOND_BWD_PRE, FOP_PROTEIN_HBOND_PASSTHROUGH, FOP_BACKBONE_PAIRS, FOP_N };

// ------------------------------------------------------------------------------------------------
// generic
// out[s] (+)= sum of in[s][0..n): ALWAYS summed as 256 strided partial sums (lane t of a virtual 256-lane block takes i = t, t + 256, ...)
// combined wave by wave in order, whatever the size of the workgroup that runs it: the value does not depend on the launch shape
struct ReduceSumArgs { const float* in; int n; float* out; int accumulate; };
__device__ __forceinline__ void c_reduce_sum(const ReduceSumArgs& A, int s, float* part /* LDS, 4 floats */) {
    const float* p = A.in + (size_t)s * A.n;
    for (int v = threadIdx.x; v < UPK_BLOCK; v += blockDim.x) {      // virtual lane v (whole wavefronts make the same trips)
        float acc = 0.f;
        for (int i = v; i < A.n; i += UPK_BLOCK) acc += p[i];
        acc = wave_sum(acc);
        if ((v & 63) == 0) part[v >> 6] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < UPK_BLOCK / UP_WAVE; ++w) t += part[w];
        A.out[s] = A.accumulate ? A.out[s] + t : t;
    }
}
// every buffer holds n_system equal slices: a workgroup clears the slices of its system
struct ZeroManyArgs { float* const* ptrs; const long* sizes; int n_buf, n_system; };
__device__ __forceinline__ void c_zero_many(const ZeroManyArgs& A, int s) {
    // one buffer per WAVEFRONT at a time: a buffer costs a chain of dependent loads (size, pointer) before its first store, and a system's
    // slice of a sensitivity array is a few KB -- walked buffer by buffer by the whole workgroup, a dozen buffers took 6 us of a
    // 56-residue pass
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), n_wave = (int)(blockDim.x >> 6), lane = threadIdx.x & 63;
    for (int b = wave; b < A.n_buf; b += n_wave) {
        const long total = A.sizes[b];
        const long per = total < (1L << 31) ? (long)((unsigned)total / (unsigned)A.n_system) : total / A.n_system;
        float* p = A.ptrs[b] + (size_t)s * per;
        if ((per & 3) == 0 && (((size_t)p) & 15) == 0) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            for (long i = lane; i < (per >> 2); i += 64) ((float4*)p)[i] = z;
        } else for (long i = lane; i < per; i += 64) p[i] = 0.f;
    }
}
extern "C" int upk_zero_many(const upk_launch_t* L, float* const* ptrs, const long* sizes, int n_buf) {
    if (n_buf <= 0) return 0;
    FARGS(ZeroManyArgs, a); a.ptrs = ptrs; a.sizes = sizes; a.n_buf = n_buf; a.n_system = L->n_system;
    return fuse_submit(L, FOP_ZERO_MANY, a, 0, {r_all()});
}
extern "C" int upk_reduce_sum(const upk_launch_t* L, const float* in, int n, float* out, int accumulate) {
    FARGS(ReduceSumArgs, a); a.in = in; a.n = n; a.out = out; a.accumulate = accumulate;
    return fuse_submit(L, FOP_REDUCE_SUM, a, 0, {r_buf(in, (size_t)n * 4, false), r_buf(out, 4, true), r_lds()});
}

__global__ void k_scale(float* __restrict__ x, int n, float f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= f;
}
extern "C" int upk_scale(const upk_launch_t* L, float* x, int n, float factor) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_scale, grid1(n, 1), dim3(UPK_BLOCK), 0, ST(L), x, n, factor);
    return launch_status();
}

// Eight lanes per target element, lane c of the group owning component c: a contribution row (3 to 8 consecutive floats) is one
// 32-byte access of the group instead of `width` instructions that each touch 64 different rows.  Every component is still
// summed by one lane in entry order (the same sum as one lane per element).
struct GatherContribArgs { const float* arena; long arena_stride; const int* csr_start; const int* csr_entry; upk_coord_t target; int width, comp_offset; };
__device__ __forceinline__ void b_gather_contrib(const int g, const int s, const float* __restrict__ arena, long arena_stride, const int* __restrict__ csr_start,
                                                 const int* __restrict__ csr_entry, upk_coord_t target, int width, int comp_offset) {
    const int t = g >> 3, c = g & 7;
    if (t >= target.n_elem || c >= width) return;
    const float* a = arena + (size_t)s * arena_stride + c;
    float acc = 0.f;
    const int e0 = csr_start[t], e1 = csr_start[t + 1];
    // GC_UNROLL contributions per trip: their offsets, then their values, are fetched as independent loads before the adds (a trip is
    // two dependent round trips; an atom of `pos` gathers 15-20 contributions)
#ifndef GC_UNROLL
#define GC_UNROLL 8
#endif
    for (int e = e0; e < e1; e += GC_UNROLL) {
        int off[GC_UNROLL]; float v[GC_UNROLL];
#pragma unroll
        for (int u = 0; u < GC_UNROLL; ++u) off[u] = csr_entry[e + u < e1 ? e + u : e];
#pragma unroll
        for (int u = 0; u < GC_UNROLL; ++u) v[u] = a[off[u]];
#pragma unroll
        for (int u = 0; u < GC_UNROLL; ++u) if (e + u < e1) acc += v[u];
    }
    C_SENS(target, s)[(size_t)t * target.stride + comp_offset + c] += acc;
}
extern "C" int upk_gather_contrib(const upk_launch_t* L, const float* arena, long arena_stride, const int* csr_start,
                                  const int* csr_entry, upk_coord_t target, int width, int comp_offset) {
    if (width > 8) return 9010;
    FARGS(GatherContribArgs, a); a.arena = arena; a.arena_stride = arena_stride; a.csr_start = csr_start; a.csr_entry = csr_entry; a.target = cz(target); a.width = width; a.comp_offset = comp_offset;
    return fuse_submit(L, FOP_GATHER_CONTRIB, a, target.n_elem * 8, {r_buf(arena, (size_t)arena_stride * 4, false), r_sens(target, true)});
}

// ------------------------------------------------------------------------------------------------
// integrator (deriv_engine.cpp:11-35), thermostat (thermostat.cpp:9-18), recenter (deriv_engine.cpp:37-48)
struct IntegrationStageArgs { float* mom; upk_coord_t pos; float vel_factor, pos_factor, max_force; };
__device__ __forceinline__ void b_integration_stage(const int na, const int s, float* __restrict__ mom, upk_coord_t pos, float vel_factor, float pos_factor, float max_force) {
    if (na >= pos.n_elem) return;
    const float* d_ = C_SENS(pos, s) + (size_t)na * pos.stride;
    float* x = C_OUT(pos, s) + (size_t)na * pos.stride;
    float* m = mom + ((size_t)s * pos.n_elem + na) * 4;
    f3 d = ld3(d_);
    if (max_force != 0.f) {
        const float f_mag = sqrtf(mag2(d)) + 1e-6f;
        const float scale = atanf(f_mag * ((0.5f * UP_PI_F) / max_force)) * (max_force / f_mag * (2.f / UP_PI_F));
        d = scale * d;
    }
    const f3 p = ld3(m) - vel_factor * d;
    m[0] = p.x; m[1] = p.y; m[2] = p.z;
    x[0] += pos_factor * p.x; x[1] += pos_factor * p.y; x[2] += pos_factor * p.z;
}
extern "C" int upk_integration_stage(const upk_launch_t* L, float* mom, upk_coord_t pos, float vel_factor, float pos_factor,
                                     float max_force) {
    FARGS(IntegrationStageArgs, a); a.mom = mom; a.pos = cz(pos); a.vel_factor = vel_factor; a.pos_factor = pos_factor; a.max_force = max_force;
    return fuse_submit(L, FOP_INTEGRATION_STAGE, a, pos.n_elem, {r_buf(mom, (size_t)pos.n_elem * 16, true), r_sens(pos, false), r_out(pos, true)});
}

struct ThermostatArgs { float* mom; int n_atom; const uint32_t* seed; unsigned long long* n_inv; const float* mom_scale; const float* noise_scale; };
__device__ __forceinline__ void b_thermostat(const int na, const int s, float* __restrict__ mom, int n_atom, const uint32_t* __restrict__ seed,
                                             const unsigned long long* __restrict__ n_inv, const float* __restrict__ mom_scale,
                                             const float* __restrict__ noise_scale) {
    if (na >= n_atom) return;
    const uint64_t t = n_inv[s];   // the invocation counter lives on the device (a captured graph can be replayed), one copy per system: the
                                   // workgroup that runs a system's ops reads it here and advances it behind a barrier (c_thermostat)
    const uint32_t key[4] = {seed[s], 0u /* THERMOSTAT_RANDOM_STREAM, random.h:25 */, 0u, 0u};
    uint32_t X[4] = {(uint32_t)(t & 0xffffffffu), (uint32_t)(t >> 32), (uint32_t)na, 0u};
    threefry4x32_20(X, key);
    float n0, n1, n2, n3;
    boxmuller(n0, n1, X[0], X[1]);
    boxmuller(n2, n3, X[2], X[3]);   // 4th normal discarded (random.h:62-66)
    float* m = mom + ((size_t)s * n_atom + na) * 4;
    const float ms = mom_scale[s], ns = noise_scale[s];
    m[0] = ms * m[0] + ns * n0; m[1] = ms * m[1] + ns * n1; m[2] = ms * m[2] + ns * n2;
}
__device__ __forceinline__ void c_thermostat(const ThermostatArgs& A, int s) {
    for (int i = threadIdx.x; i < A.n_atom; i += blockDim.x) b_thermostat(i, s, A.mom, A.n_atom, A.seed, A.n_inv, A.mom_scale, A.noise_scale);
    __syncthreads();                       // every lane has read the counter
    if (threadIdx.x == 0) A.n_inv[s] += 1ull;
}
extern "C" int upk_thermostat(const upk_launch_t* L, float* mom, int n_atom, const uint32_t* seed, unsigned long long* n_invocations,
                              const float* mom_scale, const float* noise_scale) {
    FARGS(ThermostatArgs, a); a.mom = mom; a.n_atom = n_atom; a.seed = seed; a.n_inv = n_invocations; a.mom_scale = mom_scale; a.noise_scale = noise_scale;
    return fuse_submit(L, FOP_THERMOSTAT, a, 0, {r_buf(mom, (size_t)n_atom * 16, true), r_buf(n_invocations, 8, true)});
}

__global__ void k_recenter(upk_coord_t pos, int xy_only) {
    __shared__ float part[3][UPK_BLOCK / UP_WAVE];
    __shared__ float center[3];
    const int s = blockIdx.y;
    float* x = C_OUT(pos, s);
    float a[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < pos.n_elem; i += blockDim.x) for (int c = 0; c < 3; ++c) a[c] += x[(size_t)i * pos.stride + c];
    for (int c = 0; c < 3; ++c) { a[c] = wave_sum(a[c]); if ((threadIdx.x & 63) == 0) part[c][threadIdx.x >> 6] = a[c]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.f;
        for (int w = 0; w < UPK_BLOCK / UP_WAVE; ++w) t += part[threadIdx.x][w];
        center[threadIdx.x] = (xy_only && threadIdx.x == 2) ? 0.f : t / (float)pos.n_elem;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < pos.n_elem; i += blockDim.x) for (int c = 0; c < 3; ++c) x[(size_t)i * pos.stride + c] -= center[c];
}
extern "C" int upk_recenter(const upk_launch_t* L, upk_coord_t pos, int xy_only) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_recenter, dim3(1, L->n_system), dim3(UPK_BLOCK), 0, ST(L), pos, xy_only);
    return launch_status();
}

__global__ void k_kinetic(const float* __restrict__ mom, int n_atom, float* __restrict__ kin) {
    __shared__ float part[UPK_BLOCK / UP_WAVE];
    const int s = blockIdx.y;
    float a = 0.f;
    for (int i = threadIdx.x; i < n_atom; i += blockDim.x) { const float* m = mom + ((size_t)s * n_atom + i) * 4; a += m[0] * m[0] + m[1] * m[1] + m[2] * m[2]; }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < UPK_BLOCK / UP_WAVE; ++w) t += part[w]; kin[s] = 0.5f * t / (float)n_atom; }
}
extern "C" int upk_kinetic(const upk_launch_t* L, const float* mom, int n_atom, float* kin) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_kinetic, dim3(1, L->n_system), dim3(UPK_BLOCK), 0, ST(L), mom, n_atom, kin);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// affine_alignment (eig.cpp).  The reference solves the 4x4 symmetric eigenproblem for 4 residues at a time
// and decides the QR control flow with any()/none() over those 4 SIMD lanes (eig.cpp:255-267).  Here one
// lane owns one residue; groups of 4 consecutive lanes take the same decisions through a wavefront ballot, so
// every residue receives exactly the sweeps it receives in the reference.
__device__ __forceinline__ int r_idx(int i, int j) {
    const int ii = i < j ? i : j, jj = i < j ? j : i;
    return (ii == 0 ? 0 : (ii == 1 ? 3 : (ii == 2 ? 5 : 6))) + jj;
}
__device__ __forceinline__ bool group_any(bool p) {
    const unsigned long long b = __ballot(p);
    const int lane = threadIdx.x & 63;
    return ((b >> (lane & ~3)) & 0xFull) != 0ull;
}

__device__ void house(int n, float* x, float& beta) {   // eig.cpp:56-73
    float sigma2 = 1e-20f;
    for (int i = 1; i < n; ++i) sigma2 += x[i] * x[i];
    // a vector that is zero to within 1e-9 needs no reflection (and the formulas below are not one there: see oracle/upside_oracle.c: house)
    if (x[0] * x[0] + sigma2 < 1e-18f) { for (int i = 1; i < n; ++i) x[i] = 0.f; beta = 0.f; return; }
    const float mu = sqrtf(x[0] * x[0] + sigma2);
    const float s = (0.f < x[0]) ? -sigma2 * rcp(x[0] + mu) : x[0] - mu;
    beta = 2.f * s * s * rcp(sigma2 + s * s);
    x[0] = mu;
    for (int i = 1; i < n; ++i) x[i] *= rcp(s);
}

// (N rows starting at row P of the tridiagonal matrix: both compile-time, so that d, u and rot are indexed by constants and stay
//  in registers -- with run-time N and P the compiler moved the three arrays of every lane to LDS, 64 KB per workgroup)
template <int N, int P>
__device__ __forceinline__ void qr_step(float (&d)[4], float (&u)[3], float (&rot)[16]) {   // eig.cpp:183-228
    const float dval = 0.5f * (d[P + N - 2] - d[P + N - 1]) + 1e-20f;
    const float un = u[P + N - 2];
    const float mu = d[P + N - 1] - un * un * rcp(dval + copysignf(sqrtf(dval * dval + un * un), dval));
    float x = d[P] - mu, z = u[P];
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {
        const float inv_r = rsqrt_(x * x + z * z);
        const bool trivial = (z == 0.f);
        const float c = trivial ? 1.f : x * inv_r;
        const float s = trivial ? 0.f : -z * inv_r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t1 = rot[(P + k) * 4 + j], t2 = rot[(P + k + 1) * 4 + j];
            rot[(P + k) * 4 + j] = c * t1 - s * t2;
            rot[(P + k + 1) * 4 + j] = s * t1 + c * t2;
        }
        if (k > 0) u[P + k - 1] = c * x - s * z;
        const float T00 = d[P + k], T11 = d[P + k + 1], T01 = u[P + k];
        d[P + k] = T00 * c * c - T01 * 2.f * c * s + T11 * s * s;
        d[P + k + 1] = T00 * s * s + T01 * 2.f * c * s + T11 * c * c;
        u[P + k] = (T00 - T11) * c * s + T01 * (c * c - s * s);
        x = u[P + k];
        if (k < N - 2) { z = -u[P + k + 1] * s; u[P + k + 1] *= c; }
    }
}

// (Round 5, measured and removed: residues dealt to the first 4-8 lanes of EVERY wavefront of a small system's workgroup, so that a wavefront's
//  QR iteration rotates one or two deflation blocks instead of the up-to-six its 16 groups of four may need: 5.50 -> 5.36 k steps/s for one
//  56-residue system -- the op is straight-line tridiagonalisation code more than it is the iteration, eight wavefronts running it cost the
//  SIMDs eight times the issue slots, and the rama / spring / H-O ops that used to run beside the one busy wavefront queue behind it.)
struct AffineFwdArgs { upk_coord_t pos; const int* atoms; const float* ref_geom; int n_res; upk_coord_t out; float* eig; };
// (called by ALL lanes of a wavefront with consecutive lane_res: the interpreter rounds the loop bound up to whole wavefronts)
__device__ __forceinline__ void b_affine_fwd(const int lane_res, const int s, upk_coord_t pos, const int* __restrict__ atoms, const float* __restrict__ ref_geom, int n_res,
                                             upk_coord_t out, float* __restrict__ eig) {
    const int n_pad = (n_res + 3) & ~3;
    // every lane of a wave must reach the ballots; lanes beyond n_pad replay residue 0 and are discarded
    const bool in_pad = lane_res < n_pad;
    int nr = lane_res;
    if (lane_res >= n_res) nr = in_pad ? (lane_res & ~3) : 0;   // padding duplicates lane 0 of its group (eig.cpp:307-314)
    const float* x = C_OUT(pos, s);
    f3 a1 = ld3(x + (size_t)atoms[nr * 3 + 0] * pos.stride), a2 = ld3(x + (size_t)atoms[nr * 3 + 1] * pos.stride),
       a3 = ld3(x + (size_t)atoms[nr * 3 + 2] * pos.stride);
    const f3 center = (1.f / 3.f) * (a1 + a2 + a3);
    a1 = a1 - center; a2 = a2 - center; a3 = a3 - center;
    const float* g = ref_geom + nr * 9;
    const float A1[3] = {a1.x, a1.y, a1.z}, A2[3] = {a2.x, a2.y, a2.z}, A3[3] = {a3.x, a3.y, a3.z};
    float R[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = A1[j] * g[i] + A2[j] * g[3 + i] + A3[j] * g[6 + i];
    float A[10] = {R[0][0] + R[1][1] + R[2][2], R[1][2] - R[2][1], R[2][0] - R[0][2], R[0][1] - R[1][0],
                   R[0][0] - R[1][1] - R[2][2], R[0][1] + R[1][0], R[0][2] + R[2][0],
                   -R[0][0] + R[1][1] - R[2][2], R[1][2] + R[2][1], -R[0][0] - R[1][1] + R[2][2]};
    // --- symmetric_tridiagonalize_4x4 (eig.cpp:105-134)
    float beta[2];
    for (int k = 0; k < 2; ++k) {
        const int m = 4 - (k + 1);
        house(m, A + r_idx(k, k + 1), beta[k]);
        float p[3], w[3];
#define VV(j) ((j) == 0 ? 1.f : A[r_idx(k, k + 1 + (j))])
        for (int i = 0; i < m; ++i) {
            p[i] = 0.f;
            for (int j = 0; j < m; ++j) p[i] += A[r_idx(i + k + 1, j + k + 1)] * VV(j);
            p[i] *= beta[k];
        }
        float pdv = 0.f;
        for (int i = 0; i < m; ++i) pdv += p[i] * VV(i);
        for (int i = 0; i < m; ++i) w[i] = p[i] - (0.5f * beta[k] * pdv) * VV(i);
        for (int i = 0; i < m; ++i) for (int j = i; j < m; ++j) A[r_idx(i + k + 1, j + k + 1)] -= VV(i) * w[j] + VV(j) * w[i];
#undef VV
    }
    // --- unpack_tridiagonalize_4x4 (eig.cpp:137-178)
    float d[4], u[3], rot[16];
    for (int i = 0; i < 4; ++i) d[i] = A[r_idx(i, i)];
    for (int i = 0; i < 3; ++i) u[i] = A[r_idx(i, i + 1)];
    for (int i = 0; i < 16; ++i) rot[i] = 0.f;
    for (int i = 0; i < 4; ++i) rot[i * 4 + i] = 1.f;
#define V0(j) ((j) == 0 ? 1.f : A[r_idx(0, 1 + (j))])
#define V1(j) ((j) == 0 ? 1.f : A[r_idx(1, 2 + (j))])
    for (int i = 1; i < 4; ++i) for (int j = i; j < 4; ++j) rot[i * 4 + j] -= beta[0] * (V0(i - 1) * V0(j - 1));
    for (int i = 2; i < 4; ++i) for (int j = i; j < 4; ++j) rot[i * 4 + j] -= beta[1] * (V1(i - 2) * V1(j - 2));
    for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) rot[j * 4 + i] = rot[i * 4 + j];
    const float coeff = beta[0] * beta[1] * (V0(1) * V1(0) + V0(2) * V1(1));
    for (int i = 2; i < 4; ++i) for (int j = 1; j < 4; ++j) rot[i * 4 + j] += coeff * V1(i - 2) * V0(j - 1);
#undef V0
#undef V1
    // --- symm_QR_4x4 main loop (eig.cpp:248-272) with 4-lane group decisions
    bool done = false;
    for (int k = 0; k < 100; ++k) {
        if (!done)
            for (int i = 0; i < 3; ++i) if (fabsf(u[i]) <= 1e-5f * (fabsf(d[i]) + fabsf(d[i + 1]))) u[i] = 0.f;
        const bool any2 = group_any(!done && u[2] != 0.f), any1 = group_any(!done && u[1] != 0.f), any0 = group_any(!done && u[0] != 0.f);
        if (!done) {
            int q;
            if (any2) q = 0; else if (any1) q = 1; else if (any0) q = 2; else { q = 3; done = true; }
            if (!done) {
                const bool anyu[3] = {any0, any1, any2};
                int p;
                for (p = 4 - q - 1; p > 0; --p) if (!anyu[p - 1]) break;
                switch (p * 8 + (4 - q - p)) {     // (p, rows): the six blocks a 4x4 tridiagonal matrix can deflate to
                    case 0 * 8 + 4: qr_step<4, 0>(d, u, rot); break;
                    case 0 * 8 + 3: qr_step<3, 0>(d, u, rot); break;
                    case 0 * 8 + 2: qr_step<2, 0>(d, u, rot); break;
                    case 1 * 8 + 3: qr_step<3, 1>(d, u, rot); break;
                    case 1 * 8 + 2: qr_step<2, 1>(d, u, rot); break;
                    case 2 * 8 + 2: qr_step<2, 2>(d, u, rot); break;
                    default: break;
                }
            }
        }
        if (__ballot(!done) == 0ull) break;
    }
    // --- largest eigenvalue to row 0 (eig.cpp:359-374)
    for (int i = 1; i < 4; ++i) if (d[0] < d[i]) {
        const float t = d[0]; d[0] = d[i]; d[i] = t;
        for (int c = 0; c < 4; ++c) { const float tt = rot[c]; rot[c] = rot[i * 4 + c]; rot[i * 4 + c] = tt; }
    }
    if (lane_res < n_res) {
        float* o = C_OUT(out, s) + (size_t)lane_res * out.stride;
        o[0] = center.x; o[1] = center.y; o[2] = center.z;
        o[3] = rot[0]; o[4] = rot[1]; o[5] = rot[2]; o[6] = rot[3];
        float* e = eig + ((size_t)s * n_res + lane_res) * 20;
        for (int i = 0; i < 4; ++i) e[i] = d[i];
        for (int i = 0; i < 16; ++i) e[4 + i] = rot[i];
    }
}
extern "C" int upk_affine_fwd(const upk_launch_t* L, upk_coord_t pos, const int* atoms, const float* ref_geom, int n_res,
                              upk_coord_t out, float* eig) {
    FARGS(AffineFwdArgs, a); a.pos = cz(pos); a.atoms = atoms; a.ref_geom = ref_geom; a.n_res = n_res; a.out = cz(out); a.eig = eig;
    return fuse_submit(L, FOP_AFFINE_FWD, a, (n_res + 63) & ~63, {r_out(pos, false), r_out(out, true), r_buf(eig, (size_t)n_res * 80, true)});
}

struct AffineBwdArgs { upk_coord_t aff; const float* ref_geom; const float* eig; int n_res; float* contrib; long contrib_stride; };
__device__ __forceinline__ void b_affine_bwd(const int nr, const int s, upk_coord_t aff, const float* __restrict__ ref_geom, const float* __restrict__ eig, int n_res,
                                             float* __restrict__ contrib, long contrib_stride) {   // eig.cpp:388-470
    if (nr >= n_res) return;
    const float* e = eig + ((size_t)s * n_res + nr) * 20;
    const float* evals = e; const float* ev = e + 4;
#define EV(k, i) ev[(k) * 4 + (i)]
    float inv_evals[4];
    for (int j = 1; j < 4; ++j) inv_evals[j] = rcp(evals[0] - evals[j]);
    const float* sn = C_SENS(aff, s) + (size_t)nr * aff.stride;
    const float sens3[3] = {sn[0], sn[1], sn[2]}, tq[3] = {sn[3], sn[4], sn[5]};
    const float quat_sens[4] = {
        2.f * (-tq[0] * EV(0, 1) - tq[1] * EV(0, 2) - tq[2] * EV(0, 3)),
        2.f * (tq[0] * EV(0, 0) + tq[1] * EV(0, 3) - tq[2] * EV(0, 2)),
        2.f * (tq[1] * EV(0, 0) + tq[2] * EV(0, 1) - tq[0] * EV(0, 3)),
        2.f * (tq[2] * EV(0, 0) + tq[0] * EV(0, 2) - tq[1] * EV(0, 1))};
    float qsdb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
#pragma unroll
        for (int i = 0; i < 4; ++i) qsdb[dd] += quat_sens[i] * EV(dd, i);
    }
    // symmetrised products of eigenvector k with eigenvector 0, m = r_idx(i, j), i <= j: they depend on neither the atom
    // nor the component, so they are formed once (the loops below are fully unrolled: every index is a constant and the
    // tables live in registers -- left to the compiler's partial unrolling they were indexed dynamically and spilled to
    // 280 bytes of scratch per lane)
    float t[3][10];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = i; j < 4; ++j)
                t[k - 1][r_idx(i, j)] = (i == j) ? EV(k, i) * EV(0, j) : EV(k, i) * EV(0, j) + EV(k, j) * EV(0, i);
        }
    }
    float* out = contrib + (size_t)s * contrib_stride + (size_t)nr * 9;
#pragma unroll
    for (int na = 0; na < 3; ++na) {
        const float* g = ref_geom + nr * 9 + na * 3;
        const float g0 = g[0], g1 = g[1], g2 = g[2];
        const float f[3][10] = {
            {g0, 0.f, g2, -g1, g0, g1, g2, -g0, 0.f, -g0},
            {g1, -g2, 0.f, g0, -g1, g0, 0.f, g1, g2, -g1},
            {g2, g1, -g0, 0.f, -g2, 0.f, g0, -g2, g1, g2}};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float deriv = (1.f / 3.f) * sens3[c];
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                float acc = 0.f;
#pragma unroll
                for (int m = 0; m < 10; ++m) acc += f[c][m] * t[k - 1][m];     // (m ascends in the order of the i <= j double loop)
                deriv += (inv_evals[k] * acc) * qsdb[k];
            }
            out[na * 3 + c] = deriv;
        }
    }
#undef EV
}
extern "C" int upk_affine_bwd(const upk_launch_t* L, upk_coord_t aff, const float* ref_geom, const float* eig, int n_res,
                              float* contrib, long contrib_stride) {
    FARGS(AffineBwdArgs, a); a.aff = cz(aff); a.ref_geom = ref_geom; a.eig = eig; a.n_res = n_res; a.contrib = contrib; a.contrib_stride = contrib_stride;
    return fuse_submit(L, FOP_AFFINE_BWD, a, n_res, {r_sens(aff, false), r_buf(eig, (size_t)n_res * 80, false), r_slice(contrib, (size_t)n_res * 36, (size_t)contrib_stride * 4, true)});
}

// ------------------------------------------------------------------------------------------------
// rama_coord (bonds.cpp:205-247)
struct RamaFwdArgs { upk_coord_t pos; const int* atom; const int* dummy; int n_res; upk_coord_t out; float* jac; };
__device__ __forceinline__ void b_rama_fwd(const int nt, const int s, upk_coord_t pos, const int* __restrict__ atom, const int* __restrict__ dummy, int n_res,
                                           upk_coord_t out, float* __restrict__ jac) {
    if (nt >= n_res) return;
    const float* x = C_OUT(pos, s);
    f3 p[5];
    for (int a = 0; a < 5; ++a) p[a] = ld3(x + (size_t)atom[nt * 5 + a] * pos.stride);
    float* o = C_OUT(out, s) + (size_t)nt * out.stride;
    // the Jacobian of a residue: 2 angles x 5 atoms x 3 = 30 floats in a 128-byte row (UPK_RAMA_JAC floats), stored as eight 16-byte words
    // instead of thirty dwords at a 120-byte stride
    float jv[UPK_RAMA_JAC];
#pragma unroll
    for (int k = 0; k < UPK_RAMA_JAC; ++k) jv[k] = 0.f;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        f3 d[5];
#pragma unroll
        for (int a = 0; a < 5; ++a) d[a] = mk3(0.f, 0.f, 0.f);
        if (dummy[nt * 2 + pp]) o[pp] = -1.3963f;
        else o[pp] = dihedral_germ(p[0 + pp], p[1 + pp], p[2 + pp], p[3 + pp], d[0 + pp], d[1 + pp], d[2 + pp], d[3 + pp]);
#pragma unroll
        for (int a = 0; a < 5; ++a) { jv[(pp * 5 + a) * 3 + 0] = d[a].x; jv[(pp * 5 + a) * 3 + 1] = d[a].y; jv[(pp * 5 + a) * 3 + 2] = d[a].z; }
    }
    float4* j = (float4*)(jac + ((size_t)s * n_res + nt) * UPK_RAMA_JAC);
#pragma unroll
    for (int k = 0; k < UPK_RAMA_JAC / 4; ++k) j[k] = make_float4(jv[4 * k], jv[4 * k + 1], jv[4 * k + 2], jv[4 * k + 3]);
}
extern "C" int upk_rama_fwd(const upk_launch_t* L, upk_coord_t pos, const int* atom, const int* dummy, int n_res, upk_coord_t out,
                            float* jac) {
    FARGS(RamaFwdArgs, a); a.pos = cz(pos); a.atom = atom; a.dummy = dummy; a.n_res = n_res; a.out = cz(out); a.jac = jac;
    return fuse_submit(L, FOP_RAMA_FWD, a, n_res, {r_out(pos, false), r_out(out, true), r_buf(jac, (size_t)n_res * UPK_RAMA_JAC * 4, true)});
}
struct RamaBwdArgs { upk_coord_t rama; const float* jac; int n_res; float* contrib; long contrib_stride; };
__device__ __forceinline__ void b_rama_bwd(const int idx /* (residue, atom slot) */, const int s, upk_coord_t rama, const float* __restrict__ jac, int n_res,
                                           float* __restrict__ contrib, long contrib_stride) {
    if (idx >= n_res * 5) return;
    const int nt = idx / 5, a = idx % 5;
    const float* sn = C_SENS(rama, s) + (size_t)nt * rama.stride;
    const float* j = jac + ((size_t)s * n_res + nt) * UPK_RAMA_JAC;
    float* o = contrib + (size_t)s * contrib_stride + (size_t)idx * 3;
    for (int c = 0; c < 3; ++c) o[c] = sn[0] * j[(0 * 5 + a) * 3 + c] + sn[1] * j[(1 * 5 + a) * 3 + c];
}
extern "C" int upk_rama_bwd(const upk_launch_t* L, upk_coord_t rama, const float* jac, int n_res, float* contrib, long contrib_stride) {
    FARGS(RamaBwdArgs, a); a.rama = cz(rama); a.jac = jac; a.n_res = n_res; a.contrib = contrib; a.contrib_stride = contrib_stride;
    return fuse_submit(L, FOP_RAMA_BWD, a, n_res * 5, {r_sens(rama, false), r_buf(jac, (size_t)n_res * UPK_RAMA_JAC * 4, false), r_slice(contrib, (size_t)n_res * 60, (size_t)contrib_stride * 4, true)});
}

// ------------------------------------------------------------------------------------------------
// infer_H_O (hbond.cpp:59-119)
struct InferFwdArgs { upk_coord_t pos; const int* atom; const float* bond_length; int n_virtual; upk_coord_t out; float* dfd; };
__device__ __forceinline__ void b_infer_fwd(const int nv, const int s, upk_coord_t pos, const int* __restrict__ atom, const float* __restrict__ bond_length, int n_virtual,
                                            upk_coord_t out, float* __restrict__ dfd) {
    if (nv >= n_virtual) return;
    const float* x = C_OUT(pos, s);
    const f3 prev_c = ld3(x + (size_t)atom[nv * 3] * pos.stride), curr_c = ld3(x + (size_t)atom[nv * 3 + 1] * pos.stride),
             next_c = ld3(x + (size_t)atom[nv * 3 + 2] * pos.stride);
    f3 prev = prev_c - curr_c; const float prev_im = rsqrt_(mag2(prev)); prev = prev_im * prev;
    f3 next = next_c - curr_c; const float next_im = rsqrt_(mag2(next)); next = next_im * next;
    f3 disp = prev + next; const float disp_im = rsqrt_(mag2(disp)); disp = disp_im * disp;
    const f3 dir = -disp;
    const f3 hp = bond_length[nv] * dir + curr_c;
    float* sd = dfd + ((size_t)s * n_virtual + nv) * 12;
    sd[0] = prev.x; sd[1] = prev.y; sd[2] = prev.z; sd[3] = prev_im;
    sd[4] = next.x; sd[5] = next.y; sd[6] = next.z; sd[7] = next_im;
    sd[8] = disp.x; sd[9] = disp.y; sd[10] = disp.z; sd[11] = disp_im;
    float* o = C_OUT(out, s) + (size_t)nv * out.stride;
    o[0] = hp.x; o[1] = hp.y; o[2] = hp.z; o[3] = dir.x; o[4] = dir.y; o[5] = dir.z;
}
extern "C" int upk_infer_fwd(const upk_launch_t* L, upk_coord_t pos, const int* atom, const float* bond_length, int n_virtual,
                             upk_coord_t out, float* dfd) {
    FARGS(InferFwdArgs, a); a.pos = cz(pos); a.atom = atom; a.bond_length = bond_length; a.n_virtual = n_virtual; a.out = cz(out); a.dfd = dfd;
    return fuse_submit(L, FOP_INFER_FWD, a, n_virtual, {r_out(pos, false), r_out(out, true), r_buf(dfd, (size_t)n_virtual * 48, true)});
}
struct InferBwdArgs { upk_coord_t infer; const float* bond_length; const float* dfd; int n_virtual; float* contrib; long contrib_stride; };
__device__ __forceinline__ void b_infer_bwd(const int nv, const int s, upk_coord_t infer, const float* __restrict__ bond_length, const float* __restrict__ dfd, int n_virtual,
                                            float* __restrict__ contrib, long contrib_stride) {
    if (nv >= n_virtual) return;
    const float* sn = C_SENS(infer, s) + (size_t)nv * infer.stride;
    const f3 sens_pos = ld3(sn), sens_dir = ld3(sn + 3);
    const f3 snu = sens_dir + bond_length[nv] * sens_pos;
    const float* sd = dfd + ((size_t)s * n_virtual + nv) * 12;
    const f3 prev = ld3(sd), next = ld3(sd + 4), disp = ld3(sd + 8);
    const float prev_im = sd[3], next_im = sd[7], disp_im = sd[11];
    const f3 sn_disp = disp_im * (dot(disp, snu) * disp - snu);
    const f3 sn_prev = (-prev_im) * (dot(prev, sn_disp) * prev - sn_disp);
    const f3 sn_next = (-next_im) * (dot(next, sn_disp) * next - sn_disp);
    const f3 mid = sens_pos - sn_prev - sn_next;
    float* o = contrib + (size_t)s * contrib_stride + (size_t)nv * 9;
    o[0] = sn_prev.x; o[1] = sn_prev.y; o[2] = sn_prev.z;
    o[3] = mid.x; o[4] = mid.y; o[5] = mid.z;
    o[6] = sn_next.x; o[7] = sn_next.y; o[8] = sn_next.z;
}
extern "C" int upk_infer_bwd(const upk_launch_t* L, upk_coord_t infer, const float* bond_length, const float* dfd, int n_virtual,
                             float* contrib, long contrib_stride) {
    FARGS(InferBwdArgs, a); a.infer = cz(infer); a.bond_length = bond_length; a.dfd = dfd; a.n_virtual = n_virtual; a.contrib = contrib; a.contrib_stride = contrib_stride;
    return fuse_submit(L, FOP_INFER_BWD, a, n_virtual, {r_sens(infer, false), r_buf(dfd, (size_t)n_virtual * 48, false), r_slice(contrib, (size_t)n_virtual * 36, (size_t)contrib_stride * 4, true)});
}

// ------------------------------------------------------------------------------------------------
// bonded springs (bonds.cpp:297-318, 457-487, 519-545)
struct SpringArgs { int kind; upk_coord_t pos; const int* id; const float* equil; const float* kk; int n; float* contrib; long contrib_stride; float* pot_terms; };
__device__ __forceinline__ void b_spring(const int nt, const int s, int kind, upk_coord_t pos, const int* __restrict__ id, const float* __restrict__ equil,
                                         const float* __restrict__ kk, int n, float* __restrict__ contrib, long contrib_stride,
                                         float* __restrict__ pot_terms) {
    if (nt >= n) return;
    const float* x = C_OUT(pos, s);
    float* o = contrib + (size_t)s * contrib_stride + (size_t)nt * kind * 3;
    float pot;
    if (kind == 2) {
        const f3 x1 = ld3(x + (size_t)id[nt * 2] * pos.stride), x2 = ld3(x + (size_t)id[nt * 2 + 1] * pos.stride);
        const f3 disp = x1 - x2;
        const float m2 = mag2(disp);
        const f3 deriv = (kk[nt] * (1.f - equil[nt] * rsqrt_(m2))) * disp;
        pot = 0.5f * kk[nt] * sqr(sqrtf(m2) - equil[nt]);
        o[0] = deriv.x; o[1] = deriv.y; o[2] = deriv.z; o[3] = -deriv.x; o[4] = -deriv.y; o[5] = -deriv.z;
    } else if (kind == 3) {
        const f3 a1 = ld3(x + (size_t)id[nt * 3] * pos.stride), a2 = ld3(x + (size_t)id[nt * 3 + 1] * pos.stride),
                 a3 = ld3(x + (size_t)id[nt * 3 + 2] * pos.stride);
        const f3 x1 = a1 - a3; const float inv_d1 = rsqrt_(mag2(x1)); const f3 x1h = inv_d1 * x1;
        const f3 x2 = a2 - a3; const float inv_d2 = rsqrt_(mag2(x2)); const f3 x2h = inv_d2 * x2;
        const float dp = dot(x1h, x2h);
        const float pref = kk[nt] * (dp - equil[nt]);
        const f3 d1 = (pref * inv_d1) * (x2h - dp * x1h);
        const f3 d2 = (pref * inv_d2) * (x1h - dp * x2h);
        const f3 d3 = -(d1 + d2);
        pot = 0.5f * kk[nt] * sqr(dp - equil[nt]);
        o[0] = d1.x; o[1] = d1.y; o[2] = d1.z; o[3] = d2.x; o[4] = d2.y; o[5] = d2.z; o[6] = d3.x; o[7] = d3.y; o[8] = d3.z;
    } else {
        f3 p[4], d[4];
        for (int a = 0; a < 4; ++a) p[a] = ld3(x + (size_t)id[nt * 4 + a] * pos.stride);
        const float dihedral = dihedral_germ(p[0], p[1], p[2], p[3], d[0], d[1], d[2], d[3]);
        float disp = dihedral - equil[nt];
        disp = (disp > UP_PI_F) ? disp - 2.f * UP_PI_F : disp;
        disp = (disp < -UP_PI_F) ? disp + 2.f * UP_PI_F : disp;
        const float sc = kk[nt] * disp;
        pot = 0.5f * kk[nt] * sqr(disp);
        for (int a = 0; a < 4; ++a) { o[a * 3] = sc * d[a].x; o[a * 3 + 1] = sc * d[a].y; o[a * 3 + 2] = sc * d[a].z; }
    }
    if (pot_terms) pot_terms[(size_t)s * n + nt] = pot;
}
extern "C" int upk_spring(const upk_launch_t* L, int kind, upk_coord_t pos, const int* id, const float* equil, const float* k, int n,
                          float* contrib, long contrib_stride, float* pot_terms) {
    FARGS(SpringArgs, a); a.kind = kind; a.pos = cz(pos); a.id = id; a.equil = equil; a.kk = k; a.n = n; a.contrib = contrib; a.contrib_stride = contrib_stride; a.pot_terms = pot_terms;
    return fuse_submit(L, FOP_SPRING, a, n, {r_out(pos, false), r_slice(contrib, (size_t)n * kind * 12, (size_t)contrib_stride * 4, true), r_buf(pot_terms, pot_terms ? (size_t)n * 4 : 0, true)});
}

__global__ void k_cavity(upk_coord_t pos, const int* __restrict__ id, const float* __restrict__ radius, const float* __restrict__ kk,
                         int n, float* __restrict__ contrib, long contrib_stride, float* __restrict__ pot_terms) {   // bonds.cpp:350-372
    const int nt = blockIdx.x * blockDim.x + threadIdx.x;
    if (nt >= n) return;
    const int s = blockIdx.y;
    const f3 x = ld3(C_OUT(pos, s) + (size_t)id[nt] * pos.stride);
    const float r2 = mag2(x);
    float pot = 0.f; f3 d = mk3(0.f, 0.f, 0.f);
    if (r2 > sqr(radius[nt])) {
        const float inv_r = rsqrt_(r2), r = r2 * inv_r, excess = r - radius[nt];
        pot = 0.5f * kk[nt] * sqr(excess);
        d = (kk[nt] * excess * inv_r) * x;
    }
    float* o = contrib + (size_t)s * contrib_stride + (size_t)nt * 3;
    o[0] = d.x; o[1] = d.y; o[2] = d.z;
    if (pot_terms) pot_terms[(size_t)s * n + nt] = pot;
}
extern "C" int upk_cavity_radial(const upk_launch_t* L, upk_coord_t pos, const int* id, const float* radius, const float* k, int n,
                                 float* contrib, long contrib_stride, float* pot_terms) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_cavity, grid1(n, L->n_system), dim3(UPK_BLOCK), 0, ST(L), pos, id, radius, k, n, contrib, contrib_stride, pot_terms);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// placement (placement.cpp:264-307; RamaPlacement 60-92; FixedPlacement 139-141)
struct PlacementFwdArgs { upk_placement_t P; upk_coord_t aff, rama, out; float* rama_deriv; };
__device__ __forceinline__ void b_placement_fwd(const int ne, const int s, const upk_placement_t& P, upk_coord_t aff, upk_coord_t rama, upk_coord_t out, float* __restrict__ rama_deriv) {
    if (ne >= P.n_elem) return;
    const int ar = P.affine_residue[ne];
    const float* a = C_OUT(aff, s) + (size_t)ar * aff.stride;
    const f3 t = ld3(a);
    float U[9]; quat_to_rot(U, a[3], a[4], a[5], a[6]);
    float val[8];
    if (P.is_rama) {
        const float scale_x = P.nx * (0.5f / UP_PI_F - 1e-7f), scale_y = P.ny * (0.5f / UP_PI_F - 1e-7f);
        const float* r = C_OUT(rama, s) + (size_t)P.rama_residue[ne] * rama.stride;
        const float xx = (r[0] + UP_PI_F) * scale_x, yy = (r[1] + UP_PI_F) * scale_y;
        const int x_bin = (int)xx, y_bin = (int)yy;
        const float fx = xx - x_bin, fy = yy - y_bin;
        const float* c = P.spline_coeff + ((size_t)P.layer[ne] * P.nx * P.ny + (size_t)x_bin * P.ny + y_bin) * 16 * P.n_pos_dim;
        float* rd = rama_deriv + ((size_t)s * P.n_elem + ne) * 2 * P.n_pos_dim;
        for (int id = 0; id < P.n_pos_dim; ++id) bicubic_vd(val[id], rd[id], rd[P.n_pos_dim + id], c + id * 16, fx, fy);
    } else {
        for (int c = 0; c < P.n_pos_dim; ++c) val[c] = P.fixed_data[P.layer[ne] * P.n_pos_dim + c];
    }
    float* o = C_OUT(out, s) + (size_t)ne * out.stride;
    int off = 0;
    for (int k = 0; k < P.n_sig; ++k) {
        if (P.sig[k] == 0) { o[off] = val[off]; off += 1; }
        else {
            const f3 v = mk3(val[off], val[off + 1], val[off + 2]);
            const f3 r = P.sig[k] == 1 ? apply_rotation(U, v) : apply_affine(U, t, v);
            o[off] = r.x; o[off + 1] = r.y; o[off + 2] = r.z; off += 3;
        }
    }
}
extern "C" int upk_placement_fwd(const upk_launch_t* L, const upk_placement_t* P, upk_coord_t aff, upk_coord_t rama, upk_coord_t out,
                                 float* rama_deriv) {
    FARGS(PlacementFwdArgs, a); a.P = pz(*P); a.aff = cz(aff); a.rama = cz(rama); a.out = cz(out); a.rama_deriv = rama_deriv;
    return fuse_submit(L, FOP_PLACEMENT_FWD, a, P->n_elem, {r_out(aff, false), r_out(rama, false), r_out(out, true), r_buf(rama_deriv, P->is_rama ? (size_t)P->n_elem * 2 * P->n_pos_dim * 4 : 0, true)});
}

struct PlacementBwdArgs { upk_placement_t P; upk_coord_t aff, out; const float* rama_deriv; float* aff_contrib; long aff_stride; float* rama_contrib; long rama_stride; };
__device__ __forceinline__ void b_placement_bwd(const int ne, const int s, const upk_placement_t& P, upk_coord_t aff, upk_coord_t out, const float* __restrict__ rama_deriv,
                                                float* __restrict__ aff_contrib, long aff_stride, float* __restrict__ rama_contrib, long rama_stride) {
    if (ne >= P.n_elem) return;
    const int ar = P.affine_residue[ne];
    const float* a = C_OUT(aff, s) + (size_t)ar * aff.stride;
    const f3 t = ld3(a);
    float U[9]; quat_to_rot(U, a[3], a[4], a[5], a[6]);
    const float* sn = C_SENS(out, s) + (size_t)ne * out.stride;
    const float* xo = C_OUT(out, s) + (size_t)ne * out.stride;
    float ref_sens[8];
    f3 com = mk3(0.f, 0.f, 0.f), torque = mk3(0.f, 0.f, 0.f);
    int off = 0;
    for (int k = 0; k < P.n_sig; ++k) {
        if (P.sig[k] == 0) { ref_sens[off] = sn[off]; off += 1; }
        else {
            const f3 sv = ld3(sn + off), xv = ld3(xo + off);
            const f3 rs = apply_inverse_rotation(U, sv);
            ref_sens[off] = rs.x; ref_sens[off + 1] = rs.y; ref_sens[off + 2] = rs.z;
            if (P.sig[k] == 2) { com = com + sv; torque = torque + cross(xv - t, sv); }
            else torque = torque + cross(xv, sv);
            off += 3;
        }
    }
    if (P.is_rama && rama_contrib) {
        const float scale_x = P.nx * (0.5f / UP_PI_F - 1e-7f), scale_y = P.ny * (0.5f / UP_PI_F - 1e-7f);
        const float* rd = rama_deriv + ((size_t)s * P.n_elem + ne) * 2 * P.n_pos_dim;
        float ax = 0.f, bx = 0.f;
        for (int c = 0; c < P.n_pos_dim; ++c) { ax += ref_sens[c] * rd[c]; bx += ref_sens[c] * rd[P.n_pos_dim + c]; }
        float* ro = rama_contrib + (size_t)s * rama_stride + (size_t)ne * 2;
        ro[0] = scale_x * ax; ro[1] = scale_y * bx;
    }
    float* ao = aff_contrib + (size_t)s * aff_stride + (size_t)ne * 6;
    ao[0] = com.x; ao[1] = com.y; ao[2] = com.z; ao[3] = torque.x; ao[4] = torque.y; ao[5] = torque.z;
}
extern "C" int upk_placement_bwd(const upk_launch_t* L, const upk_placement_t* P, upk_coord_t aff, upk_coord_t out,
                                 const float* rama_deriv, float* aff_contrib, long aff_stride, float* rama_contrib, long rama_stride) {
    FARGS(PlacementBwdArgs, a); a.P = pz(*P); a.aff = cz(aff); a.out = cz(out); a.rama_deriv = rama_deriv; a.aff_contrib = aff_contrib; a.aff_stride = aff_stride; a.rama_contrib = rama_contrib; a.rama_stride = rama_stride;
    return fuse_submit(L, FOP_PLACEMENT_BWD, a, P->n_elem, {r_out(aff, false), r_sens(out, false), r_out(out, false), r_buf(rama_deriv, P->is_rama ? (size_t)P->n_elem * 2 * P->n_pos_dim * 4 : 0, false),
                        r_slice(aff_contrib, (size_t)P->n_elem * 24, (size_t)aff_stride * 4, true), r_slice(rama_contrib, rama_contrib ? (size_t)P->n_elem * 8 : 0, (size_t)rama_stride * 4, true)});
}

// ------------------------------------------------------------------------------------------------
// rama_map_pot (rama_map_pot.cpp:57-82)
struct RamaMapPotArgs { upk_coord_t rama; const int* residue; const int* map_id; int n; const float* coeff; int nx; float* pot_terms; };
__device__ __forceinline__ void b_rama_map_pot(const int nr, const int s, upk_coord_t rama, const int* __restrict__ residue, const int* __restrict__ map_id, int n,
                                               const float* __restrict__ coeff, int nx, float* __restrict__ pot_terms) {
    if (nr >= n) return;
    const float scale = nx * (0.5f / UP_PI_F - 1e-7f);
    const int r = residue[nr];
    const float* rc = C_OUT(rama, s) + (size_t)r * rama.stride;
    const float xx = (rc[0] + UP_PI_F) * scale, yy = (rc[1] + UP_PI_F) * scale;
    const int x_bin = (int)xx, y_bin = (int)yy;
    float value, dx, dy;
    bicubic_vd(value, dx, dy, coeff + ((size_t)map_id[nr] * nx * nx + (size_t)x_bin * nx + y_bin) * 16, xx - x_bin, yy - y_bin);
    float* rs = C_SENS(rama, s) + (size_t)r * rama.stride;
    rs[0] += dx * scale; rs[1] += dy * scale;
    if (pot_terms) pot_terms[(size_t)s * n + nr] = value;
}
extern "C" int upk_rama_map_pot(const upk_launch_t* L, upk_coord_t rama, const int* residue, const int* map_id, int n,
                                const float* coeff, int nx, float* pot_terms) {
    FARGS(RamaMapPotArgs, a); a.rama = cz(rama); a.residue = residue; a.map_id = map_id; a.n = n; a.coeff = coeff; a.nx = nx; a.pot_terms = pot_terms;
    return fuse_submit(L, FOP_RAMA_MAP_POT, a, n, {r_out(rama, false), r_sens(rama, true), r_buf(pot_terms, pot_terms ? (size_t)n * 4 : 0, true)});
}

// weighted_pos (environment.cpp:132-154)
struct WeightedPosArgs { upk_coord_t pos, energy; const int* index_pos; const int* index_weight; upk_coord_t self; };
__device__ __forceinline__ void b_weighted_pos_fwd(const int i, const int s, upk_coord_t pos, upk_coord_t energy, const int* __restrict__ index_pos,
                                                   const int* __restrict__ index_weight, upk_coord_t out) {
    if (i >= out.n_elem) return;
    const float* p = C_OUT(pos, s) + (size_t)index_pos[i] * pos.stride;
    float* o = C_OUT(out, s) + (size_t)i * out.stride;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    o[3] = expf(-C_OUT(energy, s)[(size_t)index_weight[i] * energy.stride]);
}
extern "C" int upk_weighted_pos_fwd(const upk_launch_t* L, upk_coord_t pos, upk_coord_t energy, const int* index_pos,
                                    const int* index_weight, upk_coord_t out) {
    FARGS(WeightedPosArgs, a); a.pos = cz(pos); a.energy = cz(energy); a.index_pos = index_pos; a.index_weight = index_weight; a.self = cz(out);
    return fuse_submit(L, FOP_WEIGHTED_POS_FWD, a, out.n_elem, {r_out(pos, false), r_out(energy, false), r_out(out, true)});
}
__device__ __forceinline__ void b_weighted_pos_bwd(const int i, const int s, upk_coord_t pos, upk_coord_t energy, const int* __restrict__ index_pos,
                                                   const int* __restrict__ index_weight, upk_coord_t self) {
    if (i >= self.n_elem) return;
    // (rows of a coordinate node are padded to multiples of 4 floats: one 16-byte access per row instead of three or four dwords)
    const float4 sn = *(const float4*)(C_SENS(self, s) + (size_t)i * self.stride);
    const float o3 = (C_OUT(self, s) + (size_t)i * self.stride)[3];
    float* ps = C_SENS(pos, s) + (size_t)index_pos[i] * pos.stride;      // index_pos / index_weight are injective (checked on the host)
    if ((pos.stride & 3) == 0) { float4 v = *(float4*)ps; v.x += sn.x; v.y += sn.y; v.z += sn.z; *(float4*)ps = v; }
    else { ps[0] += sn.x; ps[1] += sn.y; ps[2] += sn.z; }
    C_SENS(energy, s)[(size_t)index_weight[i] * energy.stride] -= o3 * sn.w;
}
extern "C" int upk_weighted_pos_bwd(const upk_launch_t* L, upk_coord_t pos, upk_coord_t energy, const int* index_pos,
                                    const int* index_weight, upk_coord_t self) {
    FARGS(WeightedPosArgs, a); a.pos = cz(pos); a.energy = cz(energy); a.index_pos = index_pos; a.index_weight = index_weight; a.self = cz(self);
    return fuse_submit(L, FOP_WEIGHTED_POS_BWD, a, self.n_elem, {r_sens(self, false), r_out(self, false), r_sens(pos, true), r_sens(energy, true)});
}

// nonlinear_coupling (environment.cpp:358-369)
struct NonlinearCouplingArgs { upk_coord_t input; const int* types; const float* coeff; int n_coeff; float offset, inv_dx; float* pot_terms; };
__device__ __forceinline__ void b_nonlinear_coupling(const int i, const int s, upk_coord_t input, const int* __restrict__ types, const float* __restrict__ coeff, int n_coeff,
                                                     float offset, float inv_dx, float* __restrict__ pot_terms) {
    if (i >= input.n_elem) return;
    const float coord = (C_OUT(input, s)[(size_t)i * input.stride] - offset) * inv_dx;
    float v, dv;
    clamped_deBoor_vd_scalar(v, dv, coeff + types[i] * n_coeff, coord, n_coeff);
    C_SENS(input, s)[(size_t)i * input.stride] += dv * inv_dx;
    if (pot_terms) pot_terms[(size_t)s * input.n_elem + i] = v;
}
extern "C" int upk_nonlinear_coupling(const upk_launch_t* L, upk_coord_t input, const int* types, const float* coeff, int n_coeff,
                                      float offset, float inv_dx, float* pot_terms) {
    FARGS(NonlinearCouplingArgs, a); a.input = cz(input); a.types = types; a.coeff = coeff; a.n_coeff = n_coeff; a.offset = offset; a.inv_dx = inv_dx; a.pot_terms = pot_terms;
    return fuse_submit(L, FOP_NONLINEAR_COUPLING, a, input.n_elem, {r_out(input, false), r_sens(input, true), r_buf(pot_terms, pot_terms ? (size_t)input.n_elem * 4 : 0, true)});
}

// hbond_energy (hbond.cpp:430-444)
struct HBondEnergyArgs { upk_coord_t ph; float Ep; float* pot_terms; };
__device__ __forceinline__ void b_hbond_energy(const int nv, const int s, upk_coord_t ph, float Ep, float* __restrict__ pot_terms) {
    if (nv >= ph.n_elem) return;
    C_SENS(ph, s)[(size_t)nv * ph.stride + 6] += Ep;
    if (pot_terms) pot_terms[(size_t)s * ph.n_elem + nv] = C_OUT(ph, s)[(size_t)nv * ph.stride + 6] * Ep;
}
extern "C" int upk_hbond_energy(const upk_launch_t* L, upk_coord_t protein_hbond, float E_protein, float* pot_terms) {
    FARGS(HBondEnergyArgs, a); a.ph = cz(protein_hbond); a.Ep = E_protein; a.pot_terms = pot_terms;
    return fuse_submit(L, FOP_HBOND_ENERGY, a, protein_hbond.n_elem, {r_out(protein_hbond, false), r_sens(protein_hbond, true), r_buf(pot_terms, pot_terms ? (size_t)protein_hbond.n_elem * 4 : 0, true)});
}

// protein_hbond helpers (hbond.cpp:320-335, 343-365)
struct ProteinHBondFinishArgs { upk_coord_t infer, out; };
__device__ __forceinline__ void b_protein_hbond_finish(const int nv, const int s, upk_coord_t infer, upk_coord_t out) {
    if (nv >= out.n_elem) return;
    const float4* h = (const float4*)(C_OUT(infer, s) + (size_t)nv * infer.stride);      // (6 of 8 floats)
    float4* o = (float4*)(C_OUT(out, s) + (size_t)nv * out.stride);                          // (7 of 8 floats: [6] = the bond count so far)
    const float4 h0 = h[0], h1 = h[1], o1 = o[1];
    o[0] = h0;
    o[1] = make_float4(h1.x, h1.y, 1.f - expf(-o1.z), o1.w);
}
extern "C" int upk_protein_hbond_finish(const upk_launch_t* L, upk_coord_t infer, upk_coord_t out) {
    FARGS(ProteinHBondFinishArgs, a); a.infer = cz(infer); a.out = cz(out);
    return fuse_submit(L, FOP_PROTEIN_HBOND_FINISH, a, out.n_elem, {r_out(infer, false), r_out(out, true)});
}
struct ProteinHBondBwdPreArgs { upk_coord_t self; float* sens_scaled; };
__device__ __forceinline__ void b_protein_hbond_bwd_pre(const int nv, const int s, upk_coord_t self, float* __restrict__ sens_scaled) {
    if (nv >= self.n_elem) return;
    sens_scaled[(size_t)s * self.n_elem + nv] = C_SENS(self, s)[(size_t)nv * self.stride + 6] * (1.f - C_OUT(self, s)[(size_t)nv * self.stride + 6]);
}
extern "C" int upk_protein_hbond_bwd_pre(const upk_launch_t* L, upk_coord_t self, float* sens_scaled) {
    FARGS(ProteinHBondBwdPreArgs, a); a.self = cz(self); a.sens_scaled = sens_scaled;
    return fuse_submit(L, FOP_PROTEIN_HBOND_BWD_PRE, a, self.n_elem, {r_sens(self, false), r_out(self, false), r_buf(sens_scaled, (size_t)self.n_elem * 4, true)});
}
struct ProteinHBondPassthroughArgs { upk_coord_t self, infer; const int* loc1; int n1; const int* loc2; int n2; };
__device__ __forceinline__ void b_protein_hbond_passthrough(const int nv, const int s, upk_coord_t self, upk_coord_t infer, const int* __restrict__ loc1, int n1,
                                                            const int* __restrict__ loc2, int n2) {
    if (nv >= n1 + n2) return;
    const int tgt = nv < n1 ? loc1[nv] : loc2[nv - n1];
    const float4* sn = (const float4*)(C_SENS(self, s) + (size_t)nv * self.stride);     // (rows of 8 floats: 7 and 6 used)
    float4* t = (float4*)(C_SENS(infer, s) + (size_t)tgt * infer.stride);
    const float4 s0 = sn[0], s1 = sn[1];
    float4 t0 = t[0], t1 = t[1];
    t0.x += s0.x; t0.y += s0.y; t0.z += s0.z; t0.w += s0.w; t1.x += s1.x; t1.y += s1.y;
    t[0] = t0; t[1] = t1;
}
extern "C" int upk_protein_hbond_passthrough(const upk_launch_t* L, upk_coord_t self, upk_coord_t infer, const int* loc1, int n1,
                                             const int* loc2, int n2) {
    FARGS(ProteinHBondPassthroughArgs, a); a.self = cz(self); a.infer = cz(infer); a.loc1 = loc1; a.n1 = n1; a.loc2 = loc2; a.n2 = n2;
    return fuse_submit(L, FOP_PROTEIN_HBOND_PASSTHROUGH, a, n1 + n2, {r_sens(self, false), r_sens(infer, true)});
}

// ------------------------------------------------------------------------------------------------
// backbone_pairs (backbone_steric.cpp:81-145).  Residue counts are a few hundred, so instead of a cached pair
// list every residue scans all others each step (LDS-staged centres) and accumulates its OWN force and
// torque: each pair is visited from both ends, nothing is scattered.
__device__ __forceinline__ void nonbonded_kernel(float& v, float& dv_over_r, float r_mag2) {   // backbone_steric.cpp:18-30
    const float wall = 3.0f, width = 0.10f, sharpness = 1.f / (wall * width);
    float cs, dcs; compact_sigmoid(cs, dcs, r_mag2 - wall * wall, sharpness);
    v = 4.f * cs; dv_over_r = 2.f * (4.f * dcs);
}
#define BBP_QUEUE 72      // per-wave queue of close residues (64 new + < 4 left over)
struct BackbonePairsArgs { upk_coord_t aff; const int* residue; const int* id; const int* n_atom; const float* ref_pos; int n_res; float dist_cutoff;
                           float* aff_contrib; long aff_stride; float* pot_terms;
                           // cached residue-pair lists (standalone launches; list == nullptr: every residue scans all others each step)
                           int* list; int* cnt; const float4* refc_in; float4* refc_out; int cap; float skin; int* error_flag; };
// collective op of the system's workgroup (any number of wavefronts); lds: (n_res * 17 + waves * BBP_QUEUE) floats
// rows [row0, row1) of the system (the whole system as a fused op; a range of BBP_ROWS per workgroup in a launch of its own)
#define BBP_ROWS 64
__device__ __forceinline__ void c_backbone_pairs(const BackbonePairsArgs& A, const int s, float* lds, const int row0 = 0, int row1 = 1 << 30) {
    const upk_coord_t aff = A.aff; const int* __restrict__ residue = A.residue; const int* __restrict__ id = A.id; const int* __restrict__ n_atom = A.n_atom;
    const float* __restrict__ ref_pos = A.ref_pos; const int n_res = A.n_res; const float dist_cutoff = A.dist_cutoff;
    float* __restrict__ aff_contrib = A.aff_contrib; const long aff_stride = A.aff_stride; float* __restrict__ pot_terms = A.pot_terms;
    // One wavefront per residue: 64 lanes test 64 partner residues at a time (centre distance, sequence separation),
    // close partners are compacted into a small LDS queue and evaluated four at a time, one lane per atom pair
    // (4 partners x 4 x 4 atoms), so the steric kernel runs with dense lanes; wave reduction, no scatter.
    // lds per residue: 4 atoms x 3 + centre 3 + (n_atom,id) as int bits
    float* atoms = lds;                       // [n_res][12]
    float* ctr = lds + (size_t)n_res * 12;    // [n_res][3]
    int* meta = (int*)(ctr + (size_t)n_res * 3);   // [n_res][2]
    int* queues = meta + (size_t)n_res * 2;   // [n_wave][BBP_QUEUE]
    for (int nr = threadIdx.x; nr < n_res; nr += blockDim.x) {
        const float* a = C_OUT(aff, s) + (size_t)residue[nr] * aff.stride;
        float U[9]; quat_to_rot(U, a[3], a[4], a[5], a[6]);
        const f3 t = ld3(a);
        ctr[nr * 3] = t.x; ctr[nr * 3 + 1] = t.y; ctr[nr * 3 + 2] = t.z;
        for (int na = 0; na < 4; ++na) {
            const f3 r = apply_affine(U, t, ld3(ref_pos + (nr * 4 + na) * 3));
            atoms[nr * 12 + na * 3] = r.x; atoms[nr * 12 + na * 3 + 1] = r.y; atoms[nr * 12 + na * 3 + 2] = r.z;
        }
        meta[nr * 2] = n_atom[nr]; meta[nr * 2 + 1] = id[nr];
    }
    __syncthreads();
    const float cutoff2_atom = 3.f * 3.f + 0.1f * 3.f;
    const float cut2 = dist_cutoff * dist_cutoff;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_wave = blockDim.x >> 6;
    int* q = queues + wave * BBP_QUEUE;
    const int qi = lane >> 4, i1 = (lane >> 2) & 3, i2 = lane & 3;
    if (row1 > n_res) row1 = n_res;
    // Cached lists (the reference keeps none: backbone_steric.cpp:81-145 scans all residue pairs; here the scan was 0.48 ms of a
    // 4096-system step at the scalar issue ceiling): a row's partners within cutoff + skin of the REFERENCE centres, rebuilt when the two
    // largest centre displacements add up to the skin (the rule of k_pairlist_check).  Every workgroup of a system takes the same decision
    // from the same reference centres (refc_in, written by nobody in this launch) and hands the next step its rows' references
    // (refc_out: double buffered by the caller).  A list is ascending, so the in-range partners reach the queue in the order of the
    // all-pairs scan: the sums are the same bits with and without it.
    const bool use_list = A.list != nullptr;
    bool rebuild = false;
    if (use_list) {
        __shared__ float bb_top[2][16];
        __shared__ int bb_moved;
        const float4* rin = A.refc_in + (size_t)s * n_res;
        float d1 = 0.f, d2 = 0.f;
        for (int nr = threadIdx.x; nr < n_res; nr += blockDim.x) {
            const float4 r = rin[nr];
            const float dx = ctr[nr * 3] - r.x, dy = ctr[nr * 3 + 1] - r.y, dz = ctr[nr * 3 + 2] - r.z;
            const float dd = dx * dx + dy * dy + dz * dz;
            d2 = fmaxf(d2, fminf(d1, dd)); d1 = fmaxf(d1, dd);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o1 = __shfl_xor(d1, off, 64), o2 = __shfl_xor(d2, off, 64);
            const float n1 = fmaxf(d1, o1), n2 = fmaxf(fminf(d1, o1), fmaxf(d2, o2));
            d1 = n1; d2 = n2;
        }
        if (lane == 0) { bb_top[0][wave] = d1; bb_top[1][wave] = d2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f, b = 0.f;
            for (int w = 0; w < n_wave; ++w) {
                const float o1 = bb_top[0][w], o2 = bb_top[1][w];
                const float n1 = fmaxf(a, o1), n2 = fmaxf(fminf(a, o1), fmaxf(b, o2));
                a = n1; b = n2;
            }
            bb_moved = (sqrtf(a) + sqrtf(b) > 0.999f * A.skin) ? 1 : 0;      // (first step: the references sit at 1e10)
        }
        __syncthreads();
        rebuild = bb_moved != 0;
        float4* rout = A.refc_out + (size_t)s * n_res;
        for (int nr = row0 + threadIdx.x; nr < row1; nr += blockDim.x)
            rout[nr] = rebuild ? make_float4(ctr[nr * 3], ctr[nr * 3 + 1], ctr[nr * 3 + 2], 0.f) : rin[nr];
    }
    const float cutn = dist_cutoff + (use_list ? A.skin : 0.f), cutn2 = cutn * cutn;
    for (int nr1 = row0 + wave; nr1 < row1; nr1 += n_wave) {
        const f3 t1 = ld3(ctr + nr1 * 3);
        const int na1 = meta[nr1 * 2], id1 = meta[nr1 * 2 + 1];
        const f3 x1 = ld3(atoms + nr1 * 12 + i1 * 3);
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
        int nq = 0;
        auto eval4 = [&](int base, int n_valid) {   // partners q[base .. base+3], the first n_valid of them real
            if (qi < n_valid) {
                const int nr2 = q[base + qi];
                if (i1 < na1 && i2 < meta[nr2 * 2]) {
                    const f3 rr = x1 - ld3(atoms + nr2 * 12 + i2 * 3);
                    const float r2 = mag2(rr);
                    if (!(r2 > cutoff2_atom)) {
                        float v, dor; nonbonded_kernel(v, dor, r2);
                        const f3 g = dor * rr;
                        const f3 tq = cross(x1 - t1, g);
                        acc[0] += g.x; acc[1] += g.y; acc[2] += g.z; acc[3] += tq.x; acc[4] += tq.y; acc[5] += tq.z;
                        if (nr1 < nr2) acc[6] += v;                                  // each pair's energy once
                    }
                }
            }
        };
        int* row_list = use_list ? A.list + ((size_t)s * n_res + nr1) * A.cap : nullptr;
        const int n_cand = (use_list && !rebuild) ? A.cnt[(size_t)s * n_res + nr1] : n_res;      // candidates: the cached partners, or everybody
        int n_near = 0;
        for (int c0 = 0; c0 < n_cand; c0 += 64) {
            int nr2 = c0 + lane;
            bool hit = false, near = false;
            if (nr2 < n_cand) {
                if (use_list && !rebuild) nr2 = row_list[nr2];
                const int id2 = meta[nr2 * 2 + 1];
                const f3 t2 = ld3(ctr + nr2 * 3);
                const float dd = dist2_exact(t1.x, t1.y, t1.z, t2.x, t2.y, t2.z);
                const bool sep = (1 < id1 - id2) || (1 < id2 - id1);                 // backbone_steric.cpp:32-35
                hit = sep && (dd < cut2);
                near = sep && (dd < cutn2);
            }
            if (rebuild) {           // (wave-uniform) this row's cached list: everybody within cutoff + skin, ascending
                const unsigned long long mn = __ballot(near);
                const int pos = n_near + __popcll(mn & ((1ull << lane) - 1ull));
                if (near && pos < A.cap) row_list[pos] = nr2;
                n_near += __popcll(mn);
            }
            const unsigned long long m = __ballot(hit);
            if (hit) q[nq + __popcll(m & ((1ull << lane) - 1ull))] = nr2;
            nq += __popcll(m);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int done = 0;
            for (; done + 4 <= nq; done += 4) eval4(done, 4);
            const int left = nq - done;
            int keep = 0;
            if (lane < left) keep = q[done + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < left) q[lane] = keep;
            nq = left;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (nq > 0) eval4(0, nq);
        if (rebuild && lane == 0) {
            A.cnt[(size_t)s * n_res + nr1] = n_near < A.cap ? n_near : A.cap;
            if (n_near > A.cap) *A.error_flag = 4;      // (its own code: engine.cpp names UPSIDE_HIP_BACKBONE_LIST_CAP)
        }
        const float t = wave_sum8(acc, lane);       // lane 8*c holds component c
        const int c = lane >> 3;
        if ((lane & 7) == 0) {
            if (c < 6) aff_contrib[(size_t)s * aff_stride + (size_t)nr1 * 6 + c] = t;
            else if (c == 6 && pot_terms) pot_terms[(size_t)s * n_res + nr1] = t;
        }
    }
}
__global__ void __launch_bounds__(1024) k_backbone_pairs(BackbonePairsArgs A) {
    extern __shared__ __attribute__((aligned(16))) float bbp_lds[];
    c_backbone_pairs(A, blockIdx.y, bbp_lds, blockIdx.x * BBP_ROWS, (blockIdx.x + 1) * BBP_ROWS);
}
extern "C" int upk_backbone_pairs(const upk_launch_t* L, upk_coord_t aff, const int* residue, const int* id, const int* n_atom,
                                  const float* ref_pos, int n_res, float dist_cutoff, float* aff_contrib, long aff_stride,
                                  float* pot_terms, const upk_backbone_list_t* cache) {
    const size_t lds = ((size_t)n_res * (12 + 3 + 2) + 16 * BBP_QUEUE) * sizeof(float);
    if (lds > 150 * 1024) return 9001;   // > ~2200 residues: needs the tiled variant
    FARGS(BackbonePairsArgs, a); a.aff = cz(aff); a.residue = residue; a.id = id; a.n_atom = n_atom; a.ref_pos = ref_pos; a.n_res = n_res; a.dist_cutoff = dist_cutoff;
    a.aff_contrib = aff_contrib; a.aff_stride = aff_stride; a.pot_terms = pot_terms;
    // a system of more than a hundred residues gets several workgroups (each stages all centres, serves BBP_ROWS rows): as an op of the
    // fused list its N^2 / 2 centre tests would sit on the 16 wavefronts of one workgroup
    if (n_res > 2 * BBP_ROWS) {
        UPK_FLUSH(L);
        if (cache && cache->list) {      // cached residue-pair lists (see the kernel); parity: which reference buffer this step reads
            a.list = cache->list; a.cnt = cache->cnt; a.cap = cache->cap; a.skin = cache->skin; a.error_flag = cache->error_flag;
            a.refc_in = (const float4*)(cache->parity ? cache->ref1 : cache->ref0); a.refc_out = (float4*)(cache->parity ? cache->ref0 : cache->ref1);
        }
        hipLaunchKernelGGL(k_backbone_pairs, dim3((n_res + BBP_ROWS - 1) / BBP_ROWS, L->n_system), dim3(1024), lds, ST(L), a);
        return launch_status();
    }
    return fuse_submit(L, FOP_BACKBONE_PAIRS, a, 0, {r_out(aff, false), r_slice(aff_contrib, (size_t)n_res * 24, (size_t)aff_stride * 4, true), r_buf(pot_terms, pot_terms ? (size_t)n_res * 4 : 0, true), r_lds()}, (int)lds);
}

// ------------------------------------------------------------------------------------------------
// Monte-Carlo pivot move (monte_carlo_sampler.cpp:80-155): one workgroup per system
__device__ __forceinline__ void axis_angle_to_rot(float* U, float angle, f3 axis) {   // affine.h:49-64
    const float x = axis.x, y = axis.y, z = axis.z;
    const float c = cosf(angle), sn = sinf(angle), C = 1.f - c;
    U[0] = x * x * C + c;      U[1] = x * y * C - z * sn; U[2] = x * z * C + y * sn;
    U[3] = y * x * C + z * sn; U[4] = y * y * C + c;      U[5] = y * z * C - x * sn;
    U[6] = z * x * C - y * sn; U[7] = z * y * C + x * sn; U[8] = z * z * C + c;
}
__device__ __forceinline__ f3 rot_apply(const float* U, f3 r) {
    return mk3(U[0] * r.x + U[1] * r.y + U[2] * r.z, U[3] * r.x + U[4] * r.y + U[5] * r.z, U[6] * r.x + U[7] * r.y + U[8] * r.z);
}
__global__ void k_pivot_propose(upk_coord_t pos, float* __restrict__ pos_copy, upk_pivot_t Pv, const uint32_t* __restrict__ seed,
                                uint64_t round, float* __restrict__ delta_lprob) {
    __shared__ float sh_U[18], sh_o[6];
    __shared__ int sh_i[4];
    const int s = blockIdx.y;
    float* x = C_OUT(pos, s);
    float* cp = pos_copy + (size_t)s * pos.n_elem * pos.stride;
    for (int i = threadIdx.x; i < pos.n_elem * pos.stride; i += blockDim.x) cp[i] = x[i];
    if (threadIdx.x == 0) {
        const float PI = 3.14159265358979323846f;
        const uint32_t key[4] = {seed[s], 2u /* PIVOT_MOVE_RANDOM_STREAM, random.h:15 */, 0u, 0u};
        uint32_t X[4] = {(uint32_t)(round & 0xffffffffu), (uint32_t)(round >> 32), 0u, 0u};
        threefry4x32_20(X, key);
        const float u0 = u01f(X[0]), u1 = u01f(X[1]), u2 = u01f(X[2]), u3 = u01f(X[3]);
        int loc = (int)(Pv.n_loc * u2);
        if (loc == Pv.n_loc) loc--;
        const int* at = Pv.atoms + loc * 5;
        const int rt = Pv.restype[loc], nb2 = Pv.n_bin * Pv.n_bin;
        const float* cdf = Pv.cdf + (size_t)rt * nb2;
        int lo = 0, hi = nb2;                     // std::lower_bound: first entry with cdf >= value
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] < u3) lo = mid + 1; else hi = mid; }
        const int bin = lo < nb2 ? lo : nb2 - 1;
        const float new_lprob = Pv.pot[(size_t)rt * nb2 + bin];
        const int phi_bin = bin / Pv.n_bin, psi_bin = bin % Pv.n_bin;
        const float w = 2.f * PI / Pv.n_bin;
        const float new_phi = w * (phi_bin + u0 - 0.5f) - PI, new_psi = w * (psi_bin + u1 - 0.5f) - PI;
        const f3 prevC = ld3(x + (size_t)at[0] * pos.stride), N = ld3(x + (size_t)at[1] * pos.stride), CA = ld3(x + (size_t)at[2] * pos.stride),
                 Cc = ld3(x + (size_t)at[3] * pos.stride), nextN = ld3(x + (size_t)at[4] * pos.stride);
        f3 d1, d2, d3, d4;
        const float old_phi = dihedral_germ(prevC, N, CA, Cc, d1, d2, d3, d4), old_psi = dihedral_germ(N, CA, Cc, nextN, d1, d2, d3, d4);
        int ob1 = (int)((old_phi + PI) * (0.5f / PI) * Pv.n_bin + 0.5f), ob2 = (int)((old_psi + PI) * (0.5f / PI) * Pv.n_bin + 0.5f);
        ob1 = ob1 >= Pv.n_bin ? 0 : ob1; ob2 = ob2 >= Pv.n_bin ? 0 : ob2;
        const float old_lprob = Pv.pot[((size_t)rt * Pv.n_bin + ob1) * Pv.n_bin + ob2];
        const f3 a1 = CA - N, a2 = Cc - CA;
        axis_angle_to_rot(sh_U, new_phi - old_phi, (1.f / sqrtf(mag2(a1))) * a1);
        axis_angle_to_rot(sh_U + 9, new_psi - old_psi, (1.f / sqrtf(mag2(a2))) * a2);
        sh_o[0] = CA.x; sh_o[1] = CA.y; sh_o[2] = CA.z; sh_o[3] = Cc.x; sh_o[4] = Cc.y; sh_o[5] = Cc.z;
        sh_i[0] = at[3]; sh_i[1] = at[4]; sh_i[2] = Pv.range[loc * 2]; sh_i[3] = Pv.range[loc * 2 + 1];
        delta_lprob[s] = new_lprob - old_lprob;
    }
    __syncthreads();
    const f3 phi_o = mk3(sh_o[0], sh_o[1], sh_o[2]), psi_o = mk3(sh_o[3], sh_o[4], sh_o[5]);
    const int n_tail = sh_i[3] - sh_i[2];
    for (int t = threadIdx.x; t < n_tail + 2; t += blockDim.x) {
        const int na = t < 2 ? sh_i[t] : sh_i[2] + (t - 2);
        float* y = x + (size_t)na * pos.stride;
        const f3 after_psi = psi_o + rot_apply(sh_U + 9, ld3(y) - psi_o);
        const f3 after_phi = phi_o + rot_apply(sh_U, after_psi - phi_o);
        y[0] = after_phi.x; y[1] = after_phi.y; y[2] = after_phi.z;
    }
}
extern "C" int upk_pivot_propose(const upk_launch_t* L, upk_coord_t pos, float* pos_copy, const upk_pivot_t* P, const uint32_t* seed,
                                 uint64_t round, float* delta_lprob) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_pivot_propose, dim3(1, L->n_system), dim3(UPK_BLOCK), 0, ST(L), pos, pos_copy, *P, seed, round, delta_lprob);
    return launch_status();
}
__global__ void k_jump_propose(upk_coord_t pos, float* __restrict__ pos_copy, upk_jump_t J, const uint32_t* __restrict__ seed, uint64_t round,
                               float* __restrict__ delta_lprob) {
    __shared__ float sh_U[9], sh_v[3], sh_com[3], part[3][UPK_BLOCK / UP_WAVE];
    __shared__ int sh_i[3];
    const int s = blockIdx.y;
    float* x = C_OUT(pos, s);
    float* cp = pos_copy + (size_t)s * pos.n_elem * pos.stride;
    for (int i = threadIdx.x; i < pos.n_elem * pos.stride; i += blockDim.x) cp[i] = x[i];
    if (threadIdx.x == 0) {
        const uint32_t key[4] = {seed[s], 3u /* JUMP_MOVE_RANDOM_STREAM, random.h:16 */, 0u, 0u};
        uint32_t X[4] = {(uint32_t)(round & 0xffffffffu), (uint32_t)(round >> 32), 0u, 0u};
        threefry4x32_20(X, key);
        const int type = (int)(2 * u01f(X[0]));
        int chain = (int)(J.n_chain * u01f(X[3]));
        if (chain == J.n_chain) chain--;
        uint32_t Y[4] = {(uint32_t)(round & 0xffffffffu), (uint32_t)(round >> 32), 0u, 1u};   // second draw: the normals
        threefry4x32_20(Y, key);
        float n0, n1, n2, n3;
        boxmuller(n0, n1, Y[0], Y[1]); boxmuller(n2, n3, Y[2], Y[3]);
        sh_i[0] = type; sh_i[1] = J.atom_range[chain * 2]; sh_i[2] = J.atom_range[chain * 2 + 1];
        if (type == 0) {
            const float f = J.sigma_trans[chain] / sqrtf(3.f);
            sh_v[0] = f * n0; sh_v[1] = f * n1; sh_v[2] = f * n2;
        } else {
            f3 axis = mk3(n1, n2, n3);
            const float inv = 1.f / (sqrtf(mag2(axis)) + 1e-16f);
            axis_angle_to_rot(sh_U, J.sigma_rot[chain] * n0, inv * axis);
        }
        delta_lprob[s] = 0.f;
    }
    __syncthreads();
    const int first = sh_i[1], next = sh_i[2];
    if (sh_i[0] == 0) {
        for (int na = first + threadIdx.x; na < next; na += blockDim.x) {
            float* y = x + (size_t)na * pos.stride;
            y[0] = sh_v[0] + y[0]; y[1] = sh_v[1] + y[1]; y[2] = sh_v[2] + y[2];
        }
        return;
    }
    float a[3] = {0.f, 0.f, 0.f};   // centre of mass of the segment
    for (int na = first + threadIdx.x; na < next; na += blockDim.x) for (int c = 0; c < 3; ++c) a[c] += x[(size_t)na * pos.stride + c];
    for (int c = 0; c < 3; ++c) { a[c] = wave_sum(a[c]); if ((threadIdx.x & 63) == 0) part[c][threadIdx.x >> 6] = a[c]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.f;
        for (int w = 0; w < UPK_BLOCK / UP_WAVE; ++w) t += part[threadIdx.x][w];
        sh_com[threadIdx.x] = t * (1.f / (next - first));
    }
    __syncthreads();
    const f3 com = mk3(sh_com[0], sh_com[1], sh_com[2]);
    for (int na = first + threadIdx.x; na < next; na += blockDim.x) {
        float* y = x + (size_t)na * pos.stride;
        const f3 r = com + rot_apply(sh_U, ld3(y) - com);
        y[0] = r.x; y[1] = r.y; y[2] = r.z;
    }
}
extern "C" int upk_jump_propose(const upk_launch_t* L, upk_coord_t pos, float* pos_copy, const upk_jump_t* J, const uint32_t* seed,
                                uint64_t round, float* delta_lprob) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_jump_propose, dim3(1, L->n_system), dim3(UPK_BLOCK), 0, ST(L), pos, pos_copy, *J, seed, round, delta_lprob);
    return launch_status();
}
__global__ void k_mc_accept(upk_coord_t pos, const float* __restrict__ pos_copy, const float* __restrict__ e_old, const float* __restrict__ e_new,
                            const float* __restrict__ delta_lprob, const float* __restrict__ temperature, const uint32_t* __restrict__ seed,
                            uint64_t round, int stream, int accept_draw, int* __restrict__ stats) {
    __shared__ int accept;
    const int s = blockIdx.y;
    if (threadIdx.x == 0) {
        const float lb = delta_lprob[s] - (1.f / temperature[s]) * (e_new[s] - e_old[s]);   // monte_carlo_sampler.cpp:272
        int ok = 1;
        if (!(lb >= 0.f)) {
            const uint32_t key[4] = {seed[s], (uint32_t)stream, 0u, 0u};
            uint32_t X[4] = {(uint32_t)(round & 0xffffffffu), (uint32_t)(round >> 32), 0u, (uint32_t)accept_draw};   // next draw of the move's generator
            threefry4x32_20(X, key);
            ok = expf(lb) >= u01f(X[0]);
        }
        accept = ok;
        stats[s * 2] += ok; stats[s * 2 + 1] += 1;
    }
    __syncthreads();
    if (accept) return;
    float* x = C_OUT(pos, s);
    const float* cp = pos_copy + (size_t)s * pos.n_elem * pos.stride;
    for (int i = threadIdx.x; i < pos.n_elem * pos.stride; i += blockDim.x) x[i] = cp[i];
}
extern "C" int upk_mc_accept(const upk_launch_t* L, upk_coord_t pos, const float* pos_copy, const float* e_old, const float* e_new,
                             const float* delta_lprob, const float* temperature, const uint32_t* seed, uint64_t round, int stream,
                             int accept_draw, int* stats) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_mc_accept, dim3(1, L->n_system), dim3(UPK_BLOCK), 0, ST(L), pos, pos_copy, e_old, e_new, delta_lprob, temperature,
                       seed, round, stream, accept_draw, stats);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// replica exchange Metropolis on the device (main.cpp:251-273); one lane per swap pair, pairs are disjoint.
__global__ void k_replica_swap(upk_coord_t pos, const float* __restrict__ energy, const float* __restrict__ beta, int n_pair,
                               const int* __restrict__ pairs, uint32_t seed, uint64_t round, int draw0, int* __restrict__ accepted) {
    // uniforms are drawn sequentially from one generator, one 4-vector per REJECTABLE pair (main.cpp:268:
    // `expf(lboltz_diff) < random.uniform_open_closed().x()` is only evaluated when lboltz_diff < 0)
    __shared__ int draw_index[1024];
    __shared__ int acc[1024];
    if (threadIdx.x == 0) {
        int draw = draw0;
        for (int p = 0; p < n_pair; ++p) {
            const int s1 = pairs[p * 2], s2 = pairs[p * 2 + 1];
            // temperature exchange of one Hamiltonian: new_lboltz - old_lboltz = (beta1-beta2)(E1-E2)
            const float lb = (-beta[s1] * energy[s2] + -beta[s2] * energy[s1]) - (-beta[s1] * energy[s1] + -beta[s2] * energy[s2]);
            int ok = 1;
            if (lb < 0.f) {
                const uint32_t key[4] = {seed, 1u /* REPLICA_EXCHANGE_RANDOM_STREAM */, 0u, 0u};
                uint32_t X[4] = {(uint32_t)(round & 0xffffffffu), (uint32_t)(round >> 32), 0u, (uint32_t)draw};
                threefry4x32_20(X, key);
                ++draw;
                if (expf(lb) < u01f(X[0])) ok = 0;
            }
            acc[p] = ok; accepted[p] = ok; draw_index[p] = draw;
        }
        accepted[n_pair] = draw;   // generator position for the next swap set of this round
    }
    __syncthreads();
    const int n = pos.n_elem * pos.stride;
    for (int p = 0; p < n_pair; ++p) {
        if (!acc[p]) continue;
        float* a = C_OUT(pos, pairs[p * 2]); float* b = C_OUT(pos, pairs[p * 2 + 1]);
        for (int i = threadIdx.x; i < n; i += blockDim.x) { const float t = a[i]; a[i] = b[i]; b[i] = t; }
    }
}
extern "C" int upk_replica_swap(const upk_launch_t* L, upk_coord_t pos, const float* energy, const float* beta, int n_pair,
                                const int* pairs, uint32_t seed, uint64_t round, int draw0, int* accepted) {
    UPK_FLUSH(L);
    if (n_pair > 1024) return 9002;
    hipLaunchKernelGGL(k_replica_swap, dim3(1), dim3(UPK_BLOCK), 0, ST(L), pos, energy, beta, n_pair, pairs, seed, round, draw0, accepted);
    return launch_status();
}

// ---- replica exchange across GPUs (comm_rccl.cpp): everything below runs on the engine's stream between RCCL calls --------
// total potential of every system on the device: out[s] = sum over the potential nodes, in node order (the order and fp32
// arithmetic of DerivEngine::fetch_potentials / deriv_engine.cpp:143-146, so the two give the same bits)
__global__ void k_sum_potentials(const float* const* __restrict__ node_pot, int n_node, int S, float* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    float t = 0.f;
    for (int k = 0; k < n_node; ++k) t += node_pot[k][s];
    out[s] = t;
}
extern "C" int upk_sum_potentials(const upk_launch_t* L, const float* const* node_pot, int n_node, float* out) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_sum_potentials, grid1(L->n_system, 1), dim3(UPK_BLOCK), 0, ST(L), node_pot, n_node, L->n_system, out);
    return launch_status();
}
// Metropolis verdicts of one swap set over the GLOBAL ladder (main.cpp:251-273), identical on every rank: same gathered
// energies, same temperatures, same counter RNG.  draw_io: generator position within this attempt (in/out, device);
// accepted pairs trade their entries of energy_all (temperature exchange of one Hamiltonian), so the later sets of the
// attempt need no new evaluation.
__global__ void k_replica_decide(float* __restrict__ energy_all, const float* __restrict__ beta_all, int n_pair, const int* __restrict__ pairs,
                                 uint32_t seed, uint64_t round, int* __restrict__ draw_io, int* __restrict__ accepted) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int draw = *draw_io;
    for (int p = 0; p < n_pair; ++p) {
        const int s1 = pairs[p * 2], s2 = pairs[p * 2 + 1];
        const float lb = (-beta_all[s1] * energy_all[s2] + -beta_all[s2] * energy_all[s1]) - (-beta_all[s1] * energy_all[s1] + -beta_all[s2] * energy_all[s2]);
        int ok = 1;
        if (lb < 0.f) {
            const uint32_t key[4] = {seed, 1u /* REPLICA_EXCHANGE_RANDOM_STREAM */, 0u, 0u};
            uint32_t X[4] = {(uint32_t)(round & 0xffffffffu), (uint32_t)(round >> 32), 0u, (uint32_t)draw};
            threefry4x32_20(X, key);
            ++draw;
            if (expf(lb) < u01f(X[0])) ok = 0;
        }
        accepted[p] = ok;
        if (ok) { const float t = energy_all[s1]; energy_all[s1] = energy_all[s2]; energy_all[s2] = t; }
    }
    *draw_io = draw;
}
extern "C" int upk_replica_decide(const upk_launch_t* L, float* energy_all, const float* beta_all, int n_pair, const int* pairs,
                                  uint32_t seed, uint64_t round, int* draw_io, int* accepted) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_replica_decide, dim3(1), dim3(64), 0, ST(L), energy_all, beta_all, n_pair, pairs, seed, round, draw_io, accepted);
    return launch_status();
}
// apply the verdicts to this rank's coordinates.  plan[p] = {kind, a, b}: kind 1: both systems local (a, b = local ids:
// swap in place); kind 2: system a is local, its partner lives on another rank and its coordinates arrived in staging row b
__global__ void k_replica_apply(upk_coord_t pos, int n_pair, const int* __restrict__ plan, const int* __restrict__ accepted, const float* __restrict__ staging) {
    const int p = blockIdx.y;
    if (p >= n_pair || !accepted[p]) return;
    const int kind = plan[p * 3], a = plan[p * 3 + 1], b = plan[p * 3 + 2];
    const int n = pos.n_elem * pos.stride;
    float* xa = C_OUT(pos, a);
    if (kind == 1) { float* xb = C_OUT(pos, b); for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const float t = xa[i]; xa[i] = xb[i]; xb[i] = t; } }
    else if (kind == 2) { const float* in = staging + (size_t)b * n; for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) xa[i] = in[i]; }
}
extern "C" int upk_replica_apply(const upk_launch_t* L, upk_coord_t pos, int n_pair, const int* plan, const int* accepted, const float* staging) {
    UPK_FLUSH(L);
    if (n_pair <= 0) return 0;
    const int n = pos.n_elem * pos.stride;
    hipLaunchKernelGGL(k_replica_apply, dim3((unsigned)((n + UPK_BLOCK - 1) / UPK_BLOCK), (unsigned)n_pair), dim3(UPK_BLOCK), 0, ST(L), pos, n_pair, plan, accepted, staging);
    return launch_status();
}

// exchange the coordinates of disjoint pairs of systems in one launch (the accepted on-GPU pairs of a swap set)
__global__ void k_swap_system_pairs(upk_coord_t pos, const int* __restrict__ pairs) {
    const int p = blockIdx.y;
    float* a = C_OUT(pos, pairs[p * 2]); float* b = C_OUT(pos, pairs[p * 2 + 1]);
    const int n = pos.n_elem * pos.stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const float t = a[i]; a[i] = b[i]; b[i] = t; }
}
extern "C" int upk_swap_system_pairs(const upk_launch_t* L, upk_coord_t pos, int n_pair, const int* pairs) {
    UPK_FLUSH(L);
    if (n_pair <= 0) return 0;
    const int n = pos.n_elem * pos.stride;
    hipLaunchKernelGGL(k_swap_system_pairs, dim3((unsigned)((n + UPK_BLOCK - 1) / UPK_BLOCK), (unsigned)n_pair), dim3(UPK_BLOCK), 0, ST(L), pos, pairs);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// Parameter derivatives of the per-element nodes, for ONE system, into a zeroed table (the reference's
// get_param_deriv under PARAM_DERIV).  Off the MD path: global atomics.

// placement_fixed_*: sum over the elements of a layer of the sensitivity rotated into the reference frame
// (placement.cpp:144-148 called from :296, read back by :156-160)
__global__ void k_placement_param_deriv(upk_placement_t P, upk_coord_t aff, upk_coord_t out, int s, float* __restrict__ table) {
    const int ne = blockIdx.x * blockDim.x + threadIdx.x;
    if (ne >= P.n_elem) return;
    const float* a = C_OUT(aff, s) + (size_t)P.affine_residue[ne] * aff.stride;
    float U[9]; quat_to_rot(U, a[3], a[4], a[5], a[6]);
    const float* sn = C_SENS(out, s) + (size_t)ne * out.stride;
    float* t = table + (size_t)P.layer[ne] * P.n_pos_dim;
    int off = 0;
    for (int k = 0; k < P.n_sig; ++k) {
        if (P.sig[k] == 0) { atomicAdd(t + off, sn[off]); off += 1; }
        else {
            const f3 rs = apply_inverse_rotation(U, ld3(sn + off));
            atomicAdd(t + off, rs.x); atomicAdd(t + off + 1, rs.y); atomicAdd(t + off + 2, rs.z);
            off += 3;
        }
    }
}
extern "C" int upk_placement_param_deriv(const upk_launch_t* L, const upk_placement_t* P, upk_coord_t aff, upk_coord_t out, int system,
                                         float* table) {
    UPK_FLUSH(L);
    if (system < 0 || system >= L->n_system) return 9101;
    hipLaunchKernelGGL(k_placement_param_deriv, grid1(P->n_elem, 1), dim3(UPK_BLOCK), 0, ST(L), *P, aff, out, system, table);
    return launch_status();
}

// nonlinear_coupling: basis weights of the 4 coefficients under each element's coordinate (environment.cpp:375-389)
__global__ void k_nonlinear_coupling_param_deriv(upk_coord_t input, const int* __restrict__ types, int n_coeff, float offset, float inv_dx,
                                                 int s, float* __restrict__ table) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= input.n_elem) return;
    const float x = (C_OUT(input, s)[(size_t)i * input.stride] - offset) * inv_dx;
    int bin; float w[4];
    if (x <= 1.f) { bin = 0; w[0] = 1.f / 6.f; w[1] = 2.f / 3.f; w[2] = 1.f / 6.f; w[3] = 0.f; }                       // spline.h:375-392
    else if (x >= (float)(n_coeff - 2)) { bin = n_coeff - 4; w[0] = 0.f; w[1] = 1.f / 6.f; w[2] = 2.f / 3.f; w[3] = 1.f / 6.f; }
    else {                                                                                                              // spline.h:318-336
        const int x_bin = (int)x; bin = x_bin - 1;
        const float excess = x - (float)x_bin;
        float der;
        uniform_deBoor(w[0], der, 1.f, 0.f, 0.f, 0.f, excess);
        uniform_deBoor(w[1], der, 0.f, 1.f, 0.f, 0.f, excess);
        uniform_deBoor(w[2], der, 0.f, 0.f, 1.f, 0.f, excess);
        uniform_deBoor(w[3], der, 0.f, 0.f, 0.f, 1.f, excess);
    }
    for (int k = 0; k < 4; ++k) atomicAdd(table + (size_t)types[i] * n_coeff + bin + k, w[k]);
}
extern "C" int upk_nonlinear_coupling_param_deriv(const upk_launch_t* L, upk_coord_t input, const int* types, int n_coeff, float offset,
                                                  float inv_dx, int system, float* table) {
    UPK_FLUSH(L);
    if (system < 0 || system >= L->n_system) return 9101;
    hipLaunchKernelGGL(k_nonlinear_coupling_param_deriv, grid1(input.n_elem, 1), dim3(UPK_BLOCK), 0, ST(L), input, types, n_coeff, offset,
                       inv_dx, system, table);
    return launch_status();
}

// hbond_energy: d(potential)/d(E_protein) = the number of protein hydrogen bonds (hbond.cpp:436-448)
__global__ void k_column_sum(upk_coord_t c, int comp, int s, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n_elem) return;
    atomicAdd(out, C_OUT(c, s)[(size_t)i * c.stride + comp]);
}
extern "C" int upk_column_sum(const upk_launch_t* L, upk_coord_t c, int comp, int system, float* out) {
    UPK_FLUSH(L);
    if (system < 0 || system >= L->n_system) return 9101;
    hipLaunchKernelGGL(k_column_sum, grid1(c.n_elem, 1), dim3(UPK_BLOCK), 0, ST(L), c, comp, system, out);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// Optional restraint / external-field nodes of the reference (not part of the README force field; per element).

// single-atom potentials on `pos`: par is [n][8]
//   kind 0 atom_pos_spring (bonds.cpp:36-48)  par = x0[3], k
//   kind 1 tension         (bonds.cpp:73-88)  par = tension_coeff[3]
//   kind 2 AFM             (bonds.cpp:147-166) par = k, starting_tip_pos[3], pulling_vel[3];  time = time_estimate
//   kind 3 z_flat_bottom   (bonds.cpp:406-425) par = z0, radius, k
__global__ void k_point_potential(int kind, upk_coord_t pos, const int* __restrict__ id, const float* __restrict__ par, int n, float time,
                                  float* __restrict__ contrib, long contrib_stride, float* __restrict__ pot_terms) {
    const int nt = blockIdx.x * blockDim.x + threadIdx.x;
    if (nt >= n) return;
    const int s = blockIdx.y;
    const f3 x = ld3(C_OUT(pos, s) + (size_t)id[nt] * pos.stride);
    const float* p = par + (size_t)nt * 8;
    float pot = 0.f; f3 d = mk3(0.f, 0.f, 0.f);
    if (kind == 0) {
        const f3 disp = x - mk3(p[0], p[1], p[2]);
        pot = 0.5f * p[3] * mag2(disp);
        d = p[3] * disp;
    } else if (kind == 1) {
        const f3 c = mk3(p[0], p[1], p[2]);
        pot = -dot(x, c);
        d = mk3(-c.x, -c.y, -c.z);
    } else if (kind == 2) {
        const f3 tip = mk3(p[1], p[2], p[3]) + time * mk3(p[4], p[5], p[6]);
        const f3 diff = x - tip;
        pot = 0.5f * p[0] * mag2(diff);
        d = p[0] * diff;
    } else {
        const float dz = x.z - p[0];
        const float excess = dz > p[1] ? dz - p[1] : (dz < -p[1] ? dz + p[1] : 0.f);
        pot = 0.5f * p[2] * sqr(excess);
        d = mk3(0.f, 0.f, p[2] * excess);
    }
    float* o = contrib + (size_t)s * contrib_stride + (size_t)nt * 3;
    o[0] = d.x; o[1] = d.y; o[2] = d.z;
    if (pot_terms) pot_terms[(size_t)s * n + nt] = pot;
}
extern "C" int upk_point_potential(const upk_launch_t* L, int kind, upk_coord_t pos, const int* id, const float* par, int n, float time,
                                   float* contrib, long contrib_stride, float* pot_terms) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_point_potential, grid1(n, L->n_system), dim3(UPK_BLOCK), 0, ST(L), kind, pos, id, par, n, time, contrib,
                       contrib_stride, pot_terms);
    return launch_status();
}

// contact (sidechain_radial.cpp:187-204): par is [n][4] = energy, dist, scale (1/width), cutoff
__global__ void k_contact(upk_coord_t bead, const int* __restrict__ id, const float* __restrict__ par, int n, float* __restrict__ contrib,
                          long contrib_stride, float* __restrict__ pot_terms) {
    const int nc = blockIdx.x * blockDim.x + threadIdx.x;
    if (nc >= n) return;
    const int s = blockIdx.y;
    const float* x = C_OUT(bead, s);
    const f3 disp = ld3(x + (size_t)id[nc * 2] * bead.stride) - ld3(x + (size_t)id[nc * 2 + 1] * bead.stride);
    const float* p = par + (size_t)nc * 4;
    const float dist = sqrtf(mag2(disp));
    float pot = 0.f; f3 d = mk3(0.f, 0.f, 0.f);
    if (!(dist >= p[3])) {
        float v, dv;
        compact_sigmoid(v, dv, dist - p[1], p[2]);
        pot = p[0] * v;
        d = (p[0] * dv * rcp(dist)) * disp;
    }
    float* o = contrib + (size_t)s * contrib_stride + (size_t)nc * 6;
    o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = -d.x; o[4] = -d.y; o[5] = -d.z;
    if (pot_terms) pot_terms[(size_t)s * n + nc] = pot;
}
extern "C" int upk_contact(const upk_launch_t* L, upk_coord_t bead, const int* id, const float* par, int n, float* contrib,
                           long contrib_stride, float* pot_terms) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_contact, grid1(n, L->n_system), dim3(UPK_BLOCK), 0, ST(L), bead, id, par, n, contrib, contrib_stride, pot_terms);
    return launch_status();
}

// constant (bonds.cpp:550-587): the same values in every system;  slice (bonds.cpp:589-621)
__global__ void k_broadcast_rows(const float* __restrict__ value, upk_coord_t out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out.n_elem * out.width) return;
    const int ne = i / out.width, d = i - ne * out.width;
    C_OUT(out, blockIdx.y)[(size_t)ne * out.stride + d] = value[i];
}
extern "C" int upk_broadcast_rows(const upk_launch_t* L, const float* value, upk_coord_t out) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_broadcast_rows, grid1(out.n_elem * out.width, L->n_system), dim3(UPK_BLOCK), 0, ST(L), value, out);
    return launch_status();
}
__global__ void k_slice_fwd(upk_coord_t in, const int* __restrict__ id, upk_coord_t out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out.n_elem * out.width) return;
    const int s = blockIdx.y, na = i / out.width, d = i - na * out.width;
    C_OUT(out, s)[(size_t)na * out.stride + d] = C_OUT(in, s)[(size_t)id[na] * in.stride + d];
}
extern "C" int upk_slice_fwd(const upk_launch_t* L, upk_coord_t in, const int* id, upk_coord_t out) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_slice_fwd, grid1(out.n_elem * out.width, L->n_system), dim3(UPK_BLOCK), 0, ST(L), in, id, out);
    return launch_status();
}
// the slice's sensitivity rows become one contribution each of the sliced node
__global__ void k_slice_bwd(upk_coord_t self, float* __restrict__ contrib, long contrib_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= self.n_elem * self.width) return;
    const int s = blockIdx.y, na = i / self.width, d = i - na * self.width;
    contrib[(size_t)s * contrib_stride + i] = C_SENS(self, s)[(size_t)na * self.stride + d];
}
extern "C" int upk_slice_bwd(const upk_launch_t* L, upk_coord_t self, float* contrib, long contrib_stride) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_slice_bwd, grid1(self.n_elem * self.width, L->n_system), dim3(UPK_BLOCK), 0, ST(L), self, contrib, contrib_stride);
    return launch_status();
}

// uniform_transform (environment.cpp:158-235): clamped spline of a 1-wide input, element for element
__global__ void k_uniform_transform_fwd(upk_coord_t in, const float* __restrict__ coeff, int n_coeff, float offset, float inv_dx,
                                        upk_coord_t out, float* __restrict__ jac) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= in.n_elem) return;
    const int s = blockIdx.y;
    float v, dv;
    clamped_deBoor_vd_scalar(v, dv, coeff, (C_OUT(in, s)[(size_t)i * in.stride] - offset) * inv_dx, n_coeff);
    C_OUT(out, s)[(size_t)i * out.stride] = v;
    jac[(size_t)s * in.n_elem + i] = dv * inv_dx;
}
extern "C" int upk_uniform_transform_fwd(const upk_launch_t* L, upk_coord_t in, const float* coeff, int n_coeff, float offset, float inv_dx,
                                         upk_coord_t out, float* jac) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_uniform_transform_fwd, grid1(in.n_elem, L->n_system), dim3(UPK_BLOCK), 0, ST(L), in, coeff, n_coeff, offset, inv_dx, out, jac);
    return launch_status();
}
__global__ void k_uniform_transform_bwd(upk_coord_t in, upk_coord_t self, const float* __restrict__ jac) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= in.n_elem) return;
    const int s = blockIdx.y;
    C_SENS(in, s)[(size_t)i * in.stride] += jac[(size_t)s * in.n_elem + i] * C_SENS(self, s)[(size_t)i * self.stride];
}
extern "C" int upk_uniform_transform_bwd(const upk_launch_t* L, upk_coord_t in, upk_coord_t self, const float* jac) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_uniform_transform_bwd, grid1(in.n_elem, L->n_system), dim3(UPK_BLOCK), 0, ST(L), in, self, jac);
    return launch_status();
}
// d(potential)/d(offset, inv_dx, coefficients) of uniform_transform (environment.cpp:205-221), one system
__global__ void k_uniform_transform_param_deriv(upk_coord_t in, const float* __restrict__ coeff, int n_coeff, float offset, float inv_dx,
                                                int s, float* __restrict__ table) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= in.n_elem) return;
    const float x0 = C_OUT(in, s)[(size_t)i * in.stride] - offset, x = x0 * inv_dx;
    float v, dv;
    clamped_deBoor_vd_scalar(v, dv, coeff, x, n_coeff);
    int bin; float w[4];
    if (x <= 1.f) { bin = 0; w[0] = 1.f / 6.f; w[1] = 2.f / 3.f; w[2] = 1.f / 6.f; w[3] = 0.f; }
    else if (x >= (float)(n_coeff - 2)) { bin = n_coeff - 4; w[0] = 0.f; w[1] = 1.f / 6.f; w[2] = 2.f / 3.f; w[3] = 1.f / 6.f; }
    else {
        const int x_bin = (int)x; bin = x_bin - 1;
        const float excess = x - (float)x_bin; float der;
        uniform_deBoor(w[0], der, 1.f, 0.f, 0.f, 0.f, excess); uniform_deBoor(w[1], der, 0.f, 1.f, 0.f, 0.f, excess);
        uniform_deBoor(w[2], der, 0.f, 0.f, 1.f, 0.f, excess); uniform_deBoor(w[3], der, 0.f, 0.f, 0.f, 1.f, excess);
    }
    atomicAdd(table, dv); atomicAdd(table + 1, dv * x0);
    for (int k = 0; k < 4; ++k) atomicAdd(table + 2 + bin + k, w[k]);
}
extern "C" int upk_uniform_transform_param_deriv(const upk_launch_t* L, upk_coord_t in, const float* coeff, int n_coeff, float offset,
                                                 float inv_dx, int system, float* table) {
    UPK_FLUSH(L);
    if (system < 0 || system >= L->n_system) return 9101;
    hipLaunchKernelGGL(k_uniform_transform_param_deriv, grid1(in.n_elem, 1), dim3(UPK_BLOCK), 0, ST(L), in, coeff, n_coeff, offset, inv_dx, system, table);
    return launch_status();
}

// linear_coupling_uniform / _with_inactivation (environment.cpp:286-300)
__global__ void k_linear_coupling(upk_coord_t in, const int* __restrict__ types, const float* __restrict__ couplings, upk_coord_t inact,
                                  int has_inact, int inact_dim, float* __restrict__ pot_terms) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= in.n_elem) return;
    const int s = blockIdx.y;
    const float c = couplings[types[i]];
    const float act = has_inact ? sqr(1.f - C_OUT(inact, s)[(size_t)i * inact.stride + inact_dim]) : 1.f;
    const float val = C_OUT(in, s)[(size_t)i * in.stride];
    C_SENS(in, s)[(size_t)i * in.stride] += c * act;
    if (has_inact) C_SENS(inact, s)[(size_t)i * inact.stride + inact_dim] -= c * val;
    if (pot_terms) pot_terms[(size_t)s * in.n_elem + i] = c * val * act;
}
extern "C" int upk_linear_coupling(const upk_launch_t* L, upk_coord_t in, const int* types, const float* couplings, upk_coord_t inact,
                                   int has_inact, int inact_dim, float* pot_terms) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_linear_coupling, grid1(in.n_elem, L->n_system), dim3(UPK_BLOCK), 0, ST(L), in, types, couplings, inact, has_inact,
                       inact_dim, pot_terms);
    return launch_status();
}
// environment.cpp:301-312 (note: act = 1 - inactivation there, not its square)
__global__ void k_linear_coupling_param_deriv(upk_coord_t in, const int* __restrict__ types, upk_coord_t inact, int has_inact, int inact_dim,
                                              int s, float* __restrict__ table) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= in.n_elem) return;
    const float act = has_inact ? 1.f - C_OUT(inact, s)[(size_t)i * inact.stride + inact_dim] : 1.f;
    atomicAdd(table + types[i], C_OUT(in, s)[(size_t)i * in.stride] * act);
}
extern "C" int upk_linear_coupling_param_deriv(const upk_launch_t* L, upk_coord_t in, const int* types, upk_coord_t inact, int has_inact,
                                               int inact_dim, int system, float* table) {
    UPK_FLUSH(L);
    if (system < 0 || system >= L->n_system) return 9101;
    hipLaunchKernelGGL(k_linear_coupling_param_deriv, grid1(in.n_elem, 1), dim3(UPK_BLOCK), 0, ST(L), in, types, inact, has_inact, inact_dim, system, table);
    return launch_status();
}

// membrane_potential (membrane_potential.cpp:104-151).  Clamped cubic tables in monomial form (spline.h:496-515).
__device__ __forceinline__ void clamped_spline1d(float& value, float& deriv, const float* __restrict__ coeff, const float* __restrict__ table,
                                                 int layer, int nx, float x) {
    if (x >= (float)(nx - 1)) { deriv = 0.f; value = table[(size_t)layer * nx + nx - 1]; }
    else if (x <= 0.f) { deriv = 0.f; value = table[(size_t)layer * nx]; }
    else {
        const int x_bin = (int)x; const float fx = x - (float)x_bin, fx2 = fx * fx, fx3 = fx * fx2;
        const float* c = coeff + ((size_t)layer * (nx - 1) + x_bin) * 4;
        deriv = c[1] + 2.f * fx * c[2] + 3.f * fx2 * c[3];
        value = c[0] + fx * c[1] + fx2 * c[2] + fx3 * c[3];
    }
}
__global__ void k_membrane(upk_membrane_t M, upk_coord_t cb, upk_coord_t env, upk_coord_t hb, float* __restrict__ pot_terms) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    const int n_term = M.n_res + hb.n_elem;
    if (i >= n_term) return;
    float pot;
    if (i < M.n_res) {          // residue burial term: cb_index and env_index are injective (checked on the host)
        const int ci = M.cb_index[i], ei = M.env_index[i], rt = M.restype[i];
        float v, dv, sg, dsg;
        clamped_spline1d(v, dv, M.cb_coeff, M.cb_table, rt, M.cb_nx, (C_OUT(cb, s)[(size_t)ci * cb.stride + 2] + M.cb_z_shift) * M.cb_z_scale);
        compact_sigmoid(sg, dsg, C_OUT(env, s)[(size_t)ei * env.stride] - M.cov_midpoint[rt], M.cov_sharpness[rt]);
        pot = v * sg;
        C_SENS(cb, s)[(size_t)ci * cb.stride + 2] += dv * M.cb_z_scale * sg;
        C_SENS(env, s)[(size_t)ei * env.stride] += v * dsg;
    } else {                    // unpaired backbone hydrogen-bond sites
        const int nv = i - M.n_res;
        const float* h = C_OUT(hb, s) + (size_t)nv * hb.stride;
        float v, dv;
        clamped_spline1d(v, dv, M.uhb_coeff, M.uhb_table, nv >= M.n_donor ? 1 : 0, M.uhb_nx, (h[2] + M.uhb_z_shift) * M.uhb_z_scale);
        const float uhb = 1.f - h[6];
        pot = v * sqr(uhb);
        float* hs = C_SENS(hb, s) + (size_t)nv * hb.stride;
        hs[2] += dv * M.uhb_z_scale * sqr(uhb);
        hs[6] += -2.f * v * uhb;
    }
    if (pot_terms) pot_terms[(size_t)s * n_term + i] = pot;
}
extern "C" int upk_membrane(const upk_launch_t* L, const upk_membrane_t* M, upk_coord_t cb, upk_coord_t env, upk_coord_t hb, float* pot_terms) {
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_membrane, grid1(M->n_res + hb.n_elem, L->n_system), dim3(UPK_BLOCK), 0, ST(L), *M, cb, env, hb, pot_terms);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// The fused-op interpreter (see the top of this file) and its host-side queue.
template <typename A> __device__ __forceinline__ const A& fop_args(const FusedOp& op) { return *(const A*)op.payload; }
// element loops start at lane `rot` of the workgroup (a multiple of 64): consecutive ops without a barrier between them start on
// different wavefronts, so that short independent ops (a few wavefronts each) run side by side instead of queueing on wavefront 0
#define FOP_LOOP(i, n) for (int i = tid_r; i < (n); i += blockDim.x)
// HEAVY: the instance that also holds the affine-alignment backward op, whose unrolled tables want ~200 registers per lane (every
// other op fits 64); launches without that op take the light instance and its occupancy
template <bool HEAVY>
__device__ __forceinline__ void run_fused_op(const FusedOp& op, const int s, float* lds, const int rot) {
    const int n = op.n;
    int tid_r = (int)threadIdx.x - rot; if (tid_r < 0) tid_r += blockDim.x;
    switch (op.kind) {       // (wave-uniform: op is the same record for the whole workgroup)
        case FOP_ZERO_MANY: c_zero_many(fop_args<ZeroManyArgs>(op), s); break;
        case FOP_REDUCE_SUM: c_reduce_sum(fop_args<ReduceSumArgs>(op), s, lds); break;
        case FOP_THERMOSTAT: c_thermostat(fop_args<ThermostatArgs>(op), s); break;
        case FOP_BACKBONE_PAIRS: c_backbone_pairs(fop_args<BackbonePairsArgs>(op), s, lds); break;
        case FOP_GATHER_CONTRIB: { const auto& a = fop_args<GatherContribArgs>(op);
            FOP_LOOP(i, n) b_gather_contrib(i, s, a.arena, a.arena_stride, a.csr_start, a.csr_entry, a.target, a.width, a.comp_offset); } break;
        case FOP_INTEGRATION_STAGE: { const auto& a = fop_args<IntegrationStageArgs>(op);
            FOP_LOOP(i, n) b_integration_stage(i, s, a.mom, a.pos, a.vel_factor, a.pos_factor, a.max_force); } break;
        case FOP_AFFINE_FWD: { const auto& a = fop_args<AffineFwdArgs>(op);      // n is a multiple of 64: whole wavefronts reach the ballots
            FOP_LOOP(i, n) b_affine_fwd(i, s, a.pos, a.atoms, a.ref_geom, a.n_res, a.out, a.eig); } break;
        case FOP_AFFINE_BWD: if constexpr (HEAVY) { const auto& a = fop_args<AffineBwdArgs>(op);
            FOP_LOOP(i, n) b_affine_bwd(i, s, a.aff, a.ref_geom, a.eig, a.n_res, a.contrib, a.contrib_stride); } break;
        case FOP_RAMA_FWD: { const auto& a = fop_args<RamaFwdArgs>(op);
            FOP_LOOP(i, n) b_rama_fwd(i, s, a.pos, a.atom, a.dummy, a.n_res, a.out, a.jac); } break;
        case FOP_RAMA_BWD: { const auto& a = fop_args<RamaBwdArgs>(op);
            FOP_LOOP(i, n) b_rama_bwd(i, s, a.rama, a.jac, a.n_res, a.contrib, a.contrib_stride); } break;
        case FOP_INFER_FWD: { const auto& a = fop_args<InferFwdArgs>(op);
            FOP_LOOP(i, n) b_infer_fwd(i, s, a.pos, a.atom, a.bond_length, a.n_virtual, a.out, a.dfd); } break;
        case FOP_INFER_BWD: { const auto& a = fop_args<InferBwdArgs>(op);
            FOP_LOOP(i, n) b_infer_bwd(i, s, a.infer, a.bond_length, a.dfd, a.n_virtual, a.contrib, a.contrib_stride); } break;
        case FOP_SPRING: { const auto& a = fop_args<SpringArgs>(op);
            FOP_LOOP(i, n) b_spring(i, s, a.kind, a.pos, a.id, a.equil, a.kk, a.n, a.contrib, a.contrib_stride, a.pot_terms); } break;
        case FOP_PLACEMENT_FWD: { const auto& a = fop_args<PlacementFwdArgs>(op);
            FOP_LOOP(i, n) b_placement_fwd(i, s, a.P, a.aff, a.rama, a.out, a.rama_deriv); } break;
        case FOP_PLACEMENT_BWD: { const auto& a = fop_args<PlacementBwdArgs>(op);
            FOP_LOOP(i, n) b_placement_bwd(i, s, a.P, a.aff, a.out, a.rama_deriv, a.aff_contrib, a.aff_stride, a.rama_contrib, a.rama_stride); } break;
        case FOP_RAMA_MAP_POT: { const auto& a = fop_args<RamaMapPotArgs>(op);
            FOP_LOOP(i, n) b_rama_map_pot(i, s, a.rama, a.residue, a.map_id, a.n, a.coeff, a.nx, a.pot_terms); } break;
        case FOP_WEIGHTED_POS_FWD: { const auto& a = fop_args<WeightedPosArgs>(op);
            FOP_LOOP(i, n) b_weighted_pos_fwd(i, s, a.pos, a.energy, a.index_pos, a.index_weight, a.self); } break;
        case FOP_WEIGHTED_POS_BWD: { const auto& a = fop_args<WeightedPosArgs>(op);
            FOP_LOOP(i, n) b_weighted_pos_bwd(i, s, a.pos, a.energy, a.index_pos, a.index_weight, a.self); } break;
        case FOP_NONLINEAR_COUPLING: { const auto& a = fop_args<NonlinearCouplingArgs>(op);
            FOP_LOOP(i, n) b_nonlinear_coupling(i, s, a.input, a.types, a.coeff, a.n_coeff, a.offset, a.inv_dx, a.pot_terms); } break;
        case FOP_HBOND_ENERGY: { const auto& a = fop_args<HBondEnergyArgs>(op);
            FOP_LOOP(i, n) b_hbond_energy(i, s, a.ph, a.Ep, a.pot_terms); } break;
        case FOP_PROTEIN_HBOND_FINISH: { const auto& a = fop_args<ProteinHBondFinishArgs>(op);
            FOP_LOOP(i, n) b_protein_hbond_finish(i, s, a.infer, a.out); } break;
        case FOP_PROTEIN_HBOND_BWD_PRE: { const auto& a = fop_args<ProteinHBondBwdPreArgs>(op);
            FOP_LOOP(i, n) b_protein_hbond_bwd_pre(i, s, a.self, a.sens_scaled); } break;
        case FOP_PROTEIN_HBOND_PASSTHROUGH: { const auto& a = fop_args<ProteinHBondPassthroughArgs>(op);
            FOP_LOOP(i, n) b_protein_hbond_passthrough(i, s, a.self, a.infer, a.loc1, a.n1, a.loc2, a.n2); } break;
        default: break;
    }
}
// one workgroup per system walks the ops of the launch in order
// (T = the largest workgroup the instance may be launched with)
// (W: wavefronts per SIMD the register allocation aims at; 0 = whatever T allows)
template <bool HEAVY, int T, int W = 0>
__global__ void __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(W ? W : 1, W ? W : 8))) k_fused_list(const FusedOp* __restrict__ table, FusedIds ids, long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.x;
    // (Round 5, measured and removed: (i) requesting the records of all ops of the list through the scalar cache at the start -- one protein G
    //  5.52 -> 5.46 k steps/s: an op's own argument loads are not what its phase waits for; (ii) the sensitivity-clearing op ordered against the
    //  buffers it names only, with the leapfrog stage clearing pos.sens itself, one phase fewer per step -- no measurable change, and a recorded
    //  MD graph then depends on what ran before it: a frame's energy evaluation between two replays left forces in pos.sens that the
    //  replayed first pass no longer cleared (the 10 k-step protein-G run came out 5 % too hot).)
    for (int k = 0; k < ids.n; ++k) {
        const FusedOp& op = table[ids.id[k]];
        if (k && !(op.flags & 1)) __syncthreads();     // what the ops before wrote (global memory, this CU) is visible; LDS scratch is free again
        if (trace && s == 0 && threadIdx.x == 0) trace[k] = wall_clock64();      // (diagnostics, UPSIDE_HIP_FUSE_TRACE: when wavefront 0 of system 0 reaches op k)
        run_fused_op<HEAVY>(op, s, lds, ids.rot[k]);
    }
    if (trace && s == 0 && threadIdx.x == 0) trace[ids.n] = wall_clock64();
}
template <bool HEAVY, int T>
__global__ void __launch_bounds__(T) k_fused_one(FusedOp op) {       // an op on its own (no queue, or UPSIDE_HIP_FUSE=0)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    run_fused_op<HEAVY>(op, blockIdx.x, lds, 0);
}

namespace {
struct FuseQueue {
    int n_system = 1, threads = 1024;
    bool enabled = true;
    FusedOp* table_dev = nullptr; int cap = 4096;
    std::vector<FusedOp> table;                                   // host mirror of the registered ops
    std::unordered_map<unsigned long long, std::vector<int>> index;   // hash of (kind, n, lds, flags, payload) -> ids
    FusedIds pending; int pending_lds = 0; bool pending_heavy = false;
    std::vector<FuseRegion> live;                                 // regions touched since the last barrier of the pending list
    bool elide = true;                                            // UPSIDE_HIP_FUSE_BARRIERS=1: a barrier in front of every op
    int next_rot = 0;                                             // lanes taken by the ops since the last barrier (see FOP_LOOP)
    long n_launch = 0, n_ops_run = 0;
    long long* trace_dev = nullptr; double trace_us[FOP_N] = {0}; long trace_n[FOP_N] = {0};      // UPSIDE_HIP_FUSE_TRACE
};
unsigned long long fuse_hash(const FusedOp& op) {
    const unsigned char* b = (const unsigned char*)&op;
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < sizeof(FusedOp); i += 8) { unsigned long long w; memcpy(&w, b + i, 8); h = (h ^ w) * 1099511628211ull; h ^= h >> 29; }
    return h;
}
// workgroup size of a fused launch: enough lanes for the per-element loops of a system (a few hundred to a few thousand items) in a
// small batch, several systems resident per CU in a large one
int fuse_threads(int n_system) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("UPSIDE_HIP_FUSE_THREADS"); forced = e ? atoi(e) : 0; if (forced % 64 || forced > 1024) forced = 0; }
    if (forced) return forced;
    // (256-lane workgroups fill a CU eight at a time: worth it once there are systems for all of them -- 512 lanes against 256, k
    //  system-steps/s: 300 residues x 256 149 / 142, x 1024 178 / 175, x 2048 equal, x 4096 186 / 187; 56 residues x 256 664 / 613,
    //  x 512 893 / 846, x 2048 1069 / 1099)
    return n_system >= 2048 ? 256 : (n_system >= 256 ? 512 : 1024);
}
template <bool LIST, typename... Args>
void fuse_launch(bool heavy, int threads, int n_system, size_t lds, hipStream_t st, Args... args) {
    const dim3 g(n_system);
    if (heavy && threads > 512) threads = 512;           // (the heavy instances are built for at most 512 lanes)
    const dim3 b(threads);
    if constexpr (LIST) {
        if (heavy) { if (threads <= 256) hipLaunchKernelGGL((k_fused_list<true, 256>), g, b, lds, st, args...); else hipLaunchKernelGGL((k_fused_list<true, 512>), g, b, lds, st, args...); }
        else {
            if (threads <= 256) hipLaunchKernelGGL((k_fused_list<false, 256, 8>), g, b, lds, st, args...);
            else hipLaunchKernelGGL((k_fused_list<false, 1024>), g, b, lds, st, args...);
        }
    } else {
        if (heavy) { if (threads <= 256) hipLaunchKernelGGL((k_fused_one<true, 256>), g, b, lds, st, args...); else hipLaunchKernelGGL((k_fused_one<true, 512>), g, b, lds, st, args...); }
        else { if (threads <= 256) hipLaunchKernelGGL((k_fused_one<false, 256>), g, b, lds, st, args...); else hipLaunchKernelGGL((k_fused_one<false, 1024>), g, b, lds, st, args...); }
    }
}
}  // namespace
static_assert(sizeof(FusedOp) == 16 + FUSE_PAYLOAD && sizeof(FusedOp) % 8 == 0, "FusedOp layout");

extern "C" void* upk_fuse_create(int n_system) {
    FuseQueue* q = new FuseQueue;
    q->n_system = n_system; q->threads = fuse_threads(n_system);
    const char* e = getenv("UPSIDE_HIP_FUSE");
    q->enabled = !(e && !atoi(e));
    q->pending.n = 0;
    { const char* b = getenv("UPSIDE_HIP_FUSE_BARRIERS"); q->elide = !(b && atoi(b)); }
    if (getenv("UPSIDE_HIP_FUSE_TRACE")) (void)hipMalloc((void**)&q->trace_dev, sizeof(long long) * (FUSE_MAX_PENDING + 1));
    if (hipMalloc((void**)&q->table_dev, (size_t)q->cap * sizeof(FusedOp)) != hipSuccess) { delete q; return nullptr; }
    q->table.reserve(256);
    return q;
}
extern "C" int upk_fuse_table_size(const upk_launch_t* L) { return L->fuse ? (int)((FuseQueue*)L->fuse)->table.size() : 0; }
extern "C" void upk_fuse_destroy(void* fuse) {
    FuseQueue* q = (FuseQueue*)fuse;
    if (!q) return;
    if (q->trace_dev) {
        static const char* names[FOP_N] = {"", "zero_many", "reduce_sum", "gather_contrib", "integration_stage", "thermostat", "affine_fwd", "affine_bwd", "rama_fwd", "rama_bwd", "infer_fwd",
            "infer_bwd", "spring", "placement_fwd", "placement_bwd", "rama_map_pot", "weighted_pos_fwd", "weighted_pos_bwd", "nonlinear_coupling", "hbond_energy", "protein_hbond_finish",
            "protein_hbond_bwd_pre", "protein_hbond_passthrough", "backbone_pairs"};
        double tot = 0.; long steps = q->trace_n[FOP_ZERO_MANY] ? q->trace_n[FOP_ZERO_MANY] : 1;
        for (int k = 1; k < FOP_N; ++k) if (q->trace_n[k]) { fprintf(stderr, "fused op %-28s %6.2f us per call  x %5.2f per pass = %7.2f us\n", names[k], q->trace_us[k] / q->trace_n[k], (double)q->trace_n[k] / steps, q->trace_us[k] / steps); tot += q->trace_us[k] / steps; }
        fprintf(stderr, "fused ops per force pass: %.1f us\n", tot);
        (void)hipFree(q->trace_dev);
    }
    if (getenv("UPSIDE_HIP_FUSE_STATS")) fprintf(stderr, "fused ops: %ld launches, %ld ops, %zu distinct ops registered\n", q->n_launch, q->n_ops_run, q->table.size());
    if (q->table_dev) (void)hipFree(q->table_dev);
    delete q;
}
extern "C" int upk_fuse_pending(const upk_launch_t* L) { return L->fuse ? ((FuseQueue*)L->fuse)->pending.n : 0; }
extern "C" long upk_fuse_launch_count(const upk_launch_t* L) { return L->fuse ? ((FuseQueue*)L->fuse)->n_launch : 0; }
extern "C" int upk_fuse_flush(const upk_launch_t* L) {
    FuseQueue* q = (FuseQueue*)L->fuse;
    if (!q || !q->pending.n) return 0;
    if (L->batch) { const int r_ = upk_batch_run(L); if (r_) return r_; }      // items of an open merged launch precede the ops queued behind them
    size_t lds = (size_t)q->pending_lds; if (lds < 64) lds = 64;      // (c_reduce_sum's partial sums)
    static const bool debug = getenv("UPSIDE_HIP_FUSE_DEBUG") != nullptr;
    if (debug) { fprintf(stderr, "fused launch (%s):", q->pending_heavy ? "heavy" : "light"); for (int k = 0; k < q->pending.n; ++k) fprintf(stderr, " %d%s", q->table[q->pending.id[k]].kind, (q->table[q->pending.id[k]].flags & 1) ? "" : "|"); fprintf(stderr, "\n"); }
    { const int T = q->pending_heavy && q->threads > 512 ? 512 : q->threads; for (int k = 0; k < q->pending.n; ++k) q->pending.rot[k] %= T; }   // (the heavy instance runs 512 lanes)
    fuse_launch<true>(q->pending_heavy, q->threads, q->n_system, lds, ST(L), (const FusedOp*)q->table_dev, q->pending, q->trace_dev);
    if (q->trace_dev) {       // per-op times of wavefront 0 of system 0 (with barriers elided an op's time includes waiting for nothing: reach-to-reach)
        long long t[FUSE_MAX_PENDING + 1];
        if (hipStreamSynchronize(ST(L)) == hipSuccess && hipMemcpy(t, q->trace_dev, sizeof(long long) * (q->pending.n + 1), hipMemcpyDeviceToHost) == hipSuccess)
        {
            for (int k = 0; k < q->pending.n; ++k) { const int kind = q->table[q->pending.id[k]].kind; q->trace_us[kind] += (t[k + 1] - t[k]) * 0.01; q->trace_n[kind] += 1; }
            static const int level = atoi(getenv("UPSIDE_HIP_FUSE_TRACE"));
            if (level >= 2 && q->n_launch >= 1000 && q->n_launch < 1012) {      // (a dozen launches of a warmed-up run: "kind@us since the list's start", | = barrier in front)
                fprintf(stderr, "fused list %ld (%s):", q->n_launch, q->pending_heavy ? "heavy" : "light");
                for (int k = 0; k < q->pending.n; ++k) fprintf(stderr, " %s%d@%.1f", (q->table[q->pending.id[k]].flags & 1) ? "" : "| ", q->table[q->pending.id[k]].kind, (t[k] - t[0]) * 0.01);
                fprintf(stderr, " end@%.1f\n", (t[q->pending.n] - t[0]) * 0.01);
            }
        }
    }
    q->n_launch += 1; q->n_ops_run += q->pending.n;
    q->pending.n = 0; q->pending_lds = 0; q->pending_heavy = false; q->live.clear(); q->next_rot = 0;
    return launch_status();
}
static bool fuse_overlap(const FuseRegion& a, const FuseRegion& b, int n_system) {
    if (!a.write && !b.write) return false;
    if (!a.lo || !b.lo) return false;                                  // an optional buffer that is not there
    if (a.lo == FUSE_ALL || b.lo == FUSE_ALL) return true;
    if (a.lo == FUSE_LDS || b.lo == FUSE_LDS) return a.lo == b.lo;
    if (!a.len || !b.len) return false;
    const size_t d = a.lo < b.lo ? (size_t)(b.lo - a.lo) : (size_t)(a.lo - b.lo);
    if (a.stride == b.stride && d < a.stride) return a.lo < b.lo ? d < a.len : d < b.len;       // the same per-system layout: compare system 0's slices
    const char* ae = a.lo + (size_t)(n_system - 1) * a.stride + a.len; const char* be = b.lo + (size_t)(n_system - 1) * b.stride + b.len;   // else the hulls
    return a.lo < be && b.lo < ae;
}
static int fuse_submit_raw(const upk_launch_t* L, int kind, const void* args, size_t bytes, int n, int lds_bytes, const FuseRegion* regs, int n_regs) {
    FusedOp op; memset(&op, 0, sizeof(op));
    op.kind = kind; op.n = n; op.lds_bytes = lds_bytes; op.flags = 0;
    memcpy(op.payload, args, bytes);
    FuseQueue* q = (FuseQueue*)L->fuse;
    if (L->batch) upk_batch_fused_submitted(L);      // (inside an open merged launch, kernels_batch.h: the batch's items so far run before this op does -- upk_fuse_flush)
    if (q && q->enabled) {
        if (q->pending.n == FUSE_MAX_PENDING) UPK_FLUSH(L);
        // barrier in front of this op?  only if it touches something an op since the last barrier touched, one of them writing
        bool conflict = !q->elide;
        for (int i = 0; i < n_regs && !conflict; ++i)
            for (const FuseRegion& r : q->live) if (fuse_overlap(regs[i], r, q->n_system)) { conflict = true; break; }
        if (conflict) { q->live.clear(); q->next_rot = 0; } else op.flags |= 1;
        q->live.insert(q->live.end(), regs, regs + n_regs);
    }
    auto alone = [&]() {
        size_t lds = (size_t)lds_bytes; if (lds < 64) lds = 64;
        fuse_launch<false>(kind == FOP_AFFINE_BWD, q ? q->threads : fuse_threads(L->n_system), L->n_system, lds, ST(L), op);
        return launch_status();
    };
    // (an op that launches now, not through the queue: what an open merged launch holds runs first)
    if (!q || !q->enabled) { if (L->batch) { const int r_ = upk_batch_run(L); if (r_) return r_; } if (q) UPK_FLUSH(L); return alone(); }
    // look the op up; register it on first sight (blocking upload of one record: in-flight launches read older records only)
    const unsigned long long h = fuse_hash(op);
    int id = -1;
    auto& bucket = q->index[h];
    for (int cand : bucket) if (!memcmp(&q->table[cand], &op, sizeof(op))) { id = cand; break; }
    if (id < 0) {
        if ((int)q->table.size() >= q->cap || (int)q->table.size() >= 65535) { UPK_FLUSH(L); return alone(); }   // (arguments that change every step)
        id = (int)q->table.size();
        // (a blocking copy from pageable memory may return while the DMA is still in flight, and the engine's streams do not order against
        //  the NULL stream: the record goes through the launch's own stream, which is then drained -- once per distinct op)
        if (hipMemcpyAsync(q->table_dev + id, &op, sizeof(op), hipMemcpyHostToDevice, ST(L)) != hipSuccess || hipStreamSynchronize(ST(L)) != hipSuccess) {
            (void)hipGetLastError(); UPK_FLUSH(L); return alone();
        }
        q->table.push_back(op); bucket.push_back(id);
    }
    if (kind == FOP_AFFINE_BWD) {
        // a large batch, or a system whose other ops want all 1024 lanes, runs the register-hungry op in a launch of its own: the ops
        // around it keep the light instance (its occupancy, its workgroup size)
        if (q->n_system >= 256 || n > 64) { UPK_FLUSH(L); q->pending.rot[q->pending.n] = 0; q->pending.id[q->pending.n++] = (unsigned short)id; q->pending_heavy = true; return upk_fuse_flush(L); }
        q->pending_heavy = true;
    }
    {   // where this op's element loop starts: behind the lanes of the ops it may run beside (whole wavefronts; collective ops take them all)
        const int T = q->pending_heavy && q->threads > 512 ? 512 : q->threads;
        const int lanes = n > 0 ? (n + 63) & ~63 : T;
        q->pending.rot[q->pending.n] = (unsigned short)(q->next_rot % T);
        q->next_rot = (q->next_rot + (lanes < T ? lanes : T)) % T;
    }
    q->pending.id[q->pending.n++] = (unsigned short)id;
    if (lds_bytes > q->pending_lds) q->pending_lds = lds_bytes;
    return 0;
}
