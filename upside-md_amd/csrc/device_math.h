// Device-side math for the Upside force pass on gfx950 (wave64).  Scalar fp32 per lane; the reference's
// 4-wide SSE structure (/root/reference/src/Float4.h) is NOT reproduced -- one lane owns one element or one
// pair, and reductions use wavefront shuffles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UP_WAVE 64
#define UP_PI_F 3.141592653589793f

namespace up {

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float mag2(f3 a) { return dot(a, a); }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float sqr(float x) { return x * x; }
__device__ __forceinline__ float rcp(float x) { return 1.f / x; }
__device__ __forceinline__ float rsqrt_(float x) { return 1.f / sqrtf(x); }
__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// squared distance with one rounding per operation and a fixed association, so that the pair-list
// membership test `dist2 < cutoff2` (interaction_graph.h:230-232) is decided on exactly the same bits as
// the CPU oracle's (no FMA contraction).
__device__ __forceinline__ float dist2_exact(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ---- wavefront reductions (64 lanes) -----------------------------------------------------------
// Cross-lane steps as VALU operations of gfx950 (nothing on the LDS pipe, unlike the ds_bpermute behind __shfl_*):
// DPP quad_perm / row_half_mirror / row_mirror inside a 16-lane row, v_permlane16_swap / v_permlane32_swap across
// rows and wave halves.  dpp_xor<1|2> are true lane^1 / lane^2 exchanges; the mirrors pair a lane with one in the
// other half of its 8 / 16 lanes, which is all a butterfly reduction needs.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
#define UP_DPP_XOR1 0xB1         /* quad_perm [1,0,3,2] */
#define UP_DPP_XOR2 0x4E         /* quad_perm [2,3,0,1] */
#define UP_DPP_HALF_MIRROR 0x141 /* L <-> 7-L   */
#define UP_DPP_ROW_MIRROR 0x140  /* L <-> 15-L  */
template <typename Op>
__device__ __forceinline__ float wave_reduce(float v, Op op) {       // every lane ends with the reduction of all 64
    v = op(v, dpp_mov<UP_DPP_XOR1>(v));
    v = op(v, dpp_mov<UP_DPP_XOR2>(v));
    v = op(v, dpp_mov<UP_DPP_HALF_MIRROR>(v));
    v = op(v, dpp_mov<UP_DPP_ROW_MIRROR>(v));
    { const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      v = op(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      v = op(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce(v, [](float a, float b) { return a + b; }); }
// sum 8 per-lane values over the wavefront in 10 cross-lane steps instead of 48: three halving exchanges leave one
// partial value per lane, three butterflies finish it.  Returns the total of v[(lane >> 3) & 7] (the same in the
// 8 lanes of a group), i.e. lane 8*c holds the sum of component c.
// All ten steps are VALU cross-lane operations of gfx950 (v_permlane32_swap / v_permlane16_swap for the half- and
// row-exchanges, DPP row_mirror / row_half_mirror / quad_perm inside a row): no ds_bpermute, nothing on the LDS pipe.
__device__ __forceinline__ float wave_sum8(const float v[8], int lane) {
    float a[4], b[2], c;
    const bool h8 = lane & 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // lanes 0-31 end up with components 0-3, lanes 32-63 with 4-7
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 4]), false, false);
        a[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {      // even 16-lane rows keep the lower pair of components, odd rows the upper pair
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 2]), false, false);
        b[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const float keep = h8 ? b[1] : b[0], send = h8 ? b[0] : b[1];
        c = keep + dpp_mov<0x140>(send);        // row_mirror: lane L hears from 15-L, which sits in the other half of the row
    }
    c += dpp_mov<0x141>(c);                     // row_half_mirror: L <-> 7-L
    c += dpp_mov<0x4E>(c);                      // quad_perm [2,3,0,1]
    c += dpp_mov<0xB1>(c);                      // quad_perm [1,0,3,2]
    return c;
}
__device__ __forceinline__ float wave_max(float v) { return wave_reduce(v, [](float a, float b) { return fmaxf(a, b); }); }

// ---- B-splines: /root/reference/src/spline.h:136-174 (uniform de Boor), 228-242, 275-310 ----------
__device__ __forceinline__ void uniform_deBoor(float& val, float& der, float c00, float c01, float c02,
                                               float c03, float excess) {
    const float yu1 = excess + 2.f, yu2 = excess + 1.f, yu3 = excess;
    const float frac13 = 1.f / 3.f;
    const float a11 = frac13 * yu1, a12 = frac13 * yu2, a13 = frac13 * yu3;
    const float c11 = (1.f - a11) * c00 + a11 * c01, d11 = c01 - c00;
    const float c12 = (1.f - a12) * c01 + a12 * c02, d12 = c02 - c01;
    const float c13 = (1.f - a13) * c02 + a13 * c03, d13 = c03 - c02;
    const float a22 = 0.5f * yu2, a23 = 0.5f * yu3;
    const float c22 = (1.f - a22) * c11 + a22 * c12, d22 = (1.f - a22) * d11 + a22 * d12;
    const float c23 = (1.f - a23) * c12 + a23 * c13, d23 = (1.f - a23) * d12 + a23 * d13;
    val = (1.f - yu3) * c22 + yu3 * c23;
    der = (1.f - yu3) * d22 + yu3 * d23;
}

template <typename P>   // P: pointer-like to float (global or LDS)
__device__ __forceinline__ void deBoor_vd(float& val, float& der, P c, float x) {
    const int x_bin = (int)x;
    const float y = x - (float)x_bin;
    uniform_deBoor(val, der, c[x_bin - 1], c[x_bin], c[x_bin + 1], c[x_bin + 2], y);
}

template <typename P>
__device__ __forceinline__ void clamped_deBoor_vd(float& val, float& der, P c, float x, int n_knot) {
    const bool too_small = x < 1.f;
    const bool too_big = (float)(n_knot - 2) <= x;
    const float xc = (too_small || too_big) ? 1.f : x;
    deBoor_vd(val, der, c, xc);
    if (too_small || too_big) {
        der = 0.f;
        if (too_small) val = (1.f / 6.f) * c[0] + (2.f / 3.f) * c[1] + (1.f / 6.f) * c[2];
        if (too_big) val = (1.f / 6.f) * c[n_knot - 3] + (2.f / 3.f) * c[n_knot - 2] + (1.f / 6.f) * c[n_knot - 1];
    }
}

// scalar clamp convention of spline.h:268-272 (x<=1, x>=n_knot-2), used by nonlinear_coupling
template <typename P>
__device__ __forceinline__ void clamped_deBoor_vd_scalar(float& val, float& der, P c, float x, int n_knot) {
    if (x <= 1.f) { val = (1.f / 6.f) * c[0] + (2.f / 3.f) * c[1] + (1.f / 6.f) * c[2]; der = 0.f; return; }
    if (x >= (float)(n_knot - 2)) {
        val = (1.f / 6.f) * c[n_knot - 3] + (2.f / 3.f) * c[n_knot - 2] + (1.f / 6.f) * c[n_knot - 1]; der = 0.f; return; }
    deBoor_vd(val, der, c, x);
}

// bicubic patch: spline.h:60-80
__device__ __forceinline__ void bicubic_vd(float& value, float& dx, float& dy, const float* __restrict__ cp,
                                           float fx, float fy) {
    // (a patch is 16 floats at a 64-byte boundary of its table: four 16-byte loads instead of sixteen dwords)
    float c[16];
    if ((((size_t)cp) & 15) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float4 q = ((const float4*)cp)[k]; c[4 * k] = q.x; c[4 * k + 1] = q.y; c[4 * k + 2] = q.z; c[4 * k + 3] = q.w; }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = cp[k];
    }
    const float fx2 = fx * fx, fx3 = fx * fx2, fy2 = fy * fy;
    const float vx0 = c[0] + fy * (c[1] + fy * (c[2] + fy * c[3]));
    const float vx1 = c[4] + fy * (c[5] + fy * (c[6] + fy * c[7]));
    const float vx2 = c[8] + fy * (c[9] + fy * (c[10] + fy * c[11]));
    const float vx3 = c[12] + fy * (c[13] + fy * (c[14] + fy * c[15]));
    const float vy1 = c[1] + fx * (c[5] + fx * (c[9] + fx * c[13]));
    const float vy2 = c[2] + fx * (c[6] + fx * (c[10] + fx * c[14]));
    const float vy3 = c[3] + fx * (c[7] + fx * (c[11] + fx * c[15]));
    dx = vx1 + 2.f * fx * vx2 + 3.f * fx2 * vx3;
    dy = vy1 + 2.f * fy * vy2 + 3.f * fy2 * vy3;
    value = vx0 + fx * vx1 + fx2 * vx2 + fx3 * vx3;
}

// vector_math.h:626-631 and 639-658
__device__ __forceinline__ void sigmoid(float& w, float& dw, float x) {
    const float z = expf(-x); w = rcp(1.f + z); dw = z * w * w; }
__device__ __forceinline__ void compact_sigmoid(float& v, float& dv, float x, float sharpness) {
    const float y = x * sharpness;
    v = 0.25f * (y + 2.f) * (y - 1.f) * (y - 1.f);
    dv = (sharpness * 0.75f) * (sqr(y) - 1.f);
    if (y < -1.f) { v = 1.f; dv = 0.f; }
    else if (1.f < y) { v = 0.f; dv = 0.f; }
}

// affine.h:98-108, 8-40
__device__ __forceinline__ void quat_to_rot(float* U, float a, float b, float c, float d) {
    U[0] = a * a + b * b - c * c - d * d; U[1] = 2.f * b * c - 2.f * a * d; U[2] = 2.f * b * d + 2.f * a * c;
    U[3] = 2.f * b * c + 2.f * a * d; U[4] = a * a - b * b + c * c - d * d; U[5] = 2.f * c * d - 2.f * a * b;
    U[6] = 2.f * b * d - 2.f * a * c; U[7] = 2.f * c * d + 2.f * a * b; U[8] = a * a - b * b - c * c + d * d;
}
__device__ __forceinline__ f3 apply_rotation(const float* U, f3 r) {
    return mk3(U[0] * r.x + U[1] * r.y + U[2] * r.z, U[3] * r.x + U[4] * r.y + U[5] * r.z, U[6] * r.x + U[7] * r.y + U[8] * r.z); }
__device__ __forceinline__ f3 apply_inverse_rotation(const float* U, f3 r) {
    return mk3(U[0] * r.x + U[3] * r.y + U[6] * r.z, U[1] * r.x + U[4] * r.y + U[7] * r.z, U[2] * r.x + U[5] * r.y + U[8] * r.z); }
__device__ __forceinline__ f3 apply_affine(const float* U, f3 t, f3 r) {
    return mk3(U[0] * r.x + U[1] * r.y + U[2] * r.z + t.x, U[3] * r.x + U[4] * r.y + U[5] * r.z + t.y,
               U[6] * r.x + U[7] * r.y + U[8] * r.z + t.z); }

// vector_math.h:703-735 (Blondel & Karplus)
__device__ __forceinline__ float dihedral_germ(f3 r1, f3 r2, f3 r3, f3 r4, f3& d1, f3& d2, f3& d3, f3& d4) {
    const f3 F = r1 - r2, G = r2 - r3, H = r4 - r3;
    const f3 A = cross(F, G), B = cross(H, G), C = cross(B, A);
    const float inv_Amag2 = rcp(mag2(A)), inv_Bmag2 = rcp(mag2(B));
    const float Gmag2 = mag2(G), inv_Gmag = rsqrt_(Gmag2), Gmag = Gmag2 * inv_Gmag;
    d1 = (-Gmag * inv_Amag2) * A;
    d4 = (Gmag * inv_Bmag2) * B;
    const f3 f_mid = (dot(F, G) * inv_Amag2 * inv_Gmag) * A - (dot(H, G) * inv_Bmag2 * inv_Gmag) * B;
    d2 = -d1 + f_mid;
    d3 = -d4 - f_mid;
    return atan2f(dot(C, G), dot(A, B) * Gmag);
}

// ---- Threefry4x32-20 (Random123/threefry.h:110-117,172,296-430) and the RandomGenerator of random.h:19-67
__device__ __forceinline__ uint32_t rotl32(uint32_t x, unsigned n) { return (x << (n & 31)) | (x >> ((32 - n) & 31)); }
__device__ __forceinline__ void threefry4x32_20(uint32_t X[4], const uint32_t key[4]) {
    const unsigned R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
    uint32_t ks[5];
    ks[4] = 0x1BD11BDAu;
#pragma unroll
    for (int i = 0; i < 4; ++i) { ks[i] = key[i]; ks[4] ^= key[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) X[i] += ks[i];
#pragma unroll
    for (int r = 0; r < 20; ++r) {
        if (r % 2 == 0) { X[0] += X[1]; X[1] = rotl32(X[1], R[r % 8][0]); X[1] ^= X[0]; X[2] += X[3]; X[3] = rotl32(X[3], R[r % 8][1]); X[3] ^= X[2]; }
        else            { X[0] += X[3]; X[3] = rotl32(X[3], R[r % 8][0]); X[3] ^= X[0]; X[2] += X[1]; X[1] = rotl32(X[1], R[r % 8][1]); X[1] ^= X[2]; }
        if (r % 4 == 3) { const int k = r / 4 + 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) X[i] += ks[(k + i) % 5];
            X[3] += k; }
    }
}
// uniform.hpp:145-179.  __fmul_rn/__fadd_rn keep the mul and add separately rounded as on the CPU.
__device__ __forceinline__ float u01f(uint32_t in) { const float factor = 1.f / 4294967296.f; return __fadd_rn(__fmul_rn((float)in, factor), 0.5f * factor); }
__device__ __forceinline__ float uneg11f(uint32_t in) { const float factor = 1.f / 2147483648.f; return __fadd_rn(__fmul_rn((float)(int32_t)in, factor), 0.5f * factor); }
__device__ __forceinline__ void boxmuller(float& a, float& b, uint32_t u0, uint32_t u1) {   // boxmuller.hpp
    const float ang = __fmul_rn(3.1415926535897932f, uneg11f(u0));
    const float r = sqrtf(-2.f * logf(u01f(u1)));
    a = sinf(ang) * r; b = cosf(ang) * r;
}

}  // namespace up
