/* TEST INFRASTRUCTURE ONLY -- see upside_oracle.h.
 *
 * Plain-C restatement of the reference's hot path.  Every function cites the reference file:line it
 * follows (paths relative to /root/reference/).  Arithmetic is scalar IEEE fp32 (compiled with
 * -ffp-contract=off, no fast-math): where the reference uses SSE approximations with one Newton step
 * (src/Float4.h:203-212) this file uses exact 1/x and 1/sqrtf(x); where the reference's 4-wide SIMD
 * grouping changes *results* (not just rounding) the grouping is reproduced and noted.
 */
#include "upside_oracle.h"
#include <hdf5.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define M_PI_F 3.141592653589793f
#define MAX_ARGS 8
#define MAX_NODES 64
#define N_BIT_ROTAMER 4

static int round_up(int i, int a) { return ((i + a - 1) / a) * a; }
static int ru(int i) { return i == 1 ? i : round_up(i, 4); }        /* src/vector_math.h:23-25 */
static float sqr(float x) { return x * x; }
#if defined(ORACLE_SSE_APPROX) && defined(__SSE__)
/* Diagnostic build only (make oracle_sse): reproduce the reference's rsqrtps/rcpps + one Newton step
 * (src/Float4.h:203-212) to show that the residual oracle-vs-reference difference (~1e-5 relative) is the
 * reference's own approximation error, not a restatement error.  Host-CPU specific; never used for parity. */
#include <xmmintrin.h>
static float rcpf(float x) { float a = _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(x))); return a * (2.f - x * a); }
static float rsqrtf_(float x) { float a = _mm_cvtss_f32(_mm_rsqrt_ss(_mm_set_ss(x))); return a * (1.5f - (0.5f * x) * (a * a)); }
#else
static float rcpf(float x) { return 1.f / x; }
static float rsqrtf_(float x) { return 1.f / sqrtf(x); }
#endif
static float minf(float a, float b) { return a < b ? a : b; }
static float maxf(float a, float b) { return a > b ? a : b; }

static void* xcalloc(size_t n, size_t sz) {
    void* p = calloc(n ? n : 1, sz);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}

/* ============================================================================================
 * HDF5 helpers (C API); the schema is the one read by src/h5_support.h:228-250, src/h5_support.cpp:57-107
 * ========================================================================================== */
static int h5_dims(hid_t g, const char* name, int ndim_expected, hsize_t* dims) {
    hid_t d = H5Dopen2(g, name, H5P_DEFAULT);
    if (d < 0) return -1;
    hid_t s = H5Dget_space(d);
    int nd = H5Sget_simple_extent_ndims(s);
    if (nd != ndim_expected) { H5Sclose(s); H5Dclose(d); return -1; }
    H5Sget_simple_extent_dims(s, dims, NULL);
    H5Sclose(s); H5Dclose(d);
    return 0;
}
static size_t h5_count(int nd, const hsize_t* dims) { size_t n = 1; for (int i = 0; i < nd; ++i) n *= dims[i]; return n; }

static void* h5_read(hid_t g, const char* name, int nd, hsize_t* dims, hid_t memtype, size_t elsize) {
    if (h5_dims(g, name, nd, dims)) { fprintf(stderr, "oracle: cannot read dataset %s (ndim %i)\n", name, nd); return NULL; }
    void* buf = xcalloc(h5_count(nd, dims), elsize);
    hid_t d = H5Dopen2(g, name, H5P_DEFAULT);
    herr_t e = H5Dread(d, memtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf);
    H5Dclose(d);
    if (e < 0) { free(buf); return NULL; }
    return buf;
}
static float*  h5_read_f(hid_t g, const char* n, int nd, hsize_t* dims) { return (float*) h5_read(g, n, nd, dims, H5T_NATIVE_FLOAT,  sizeof(float)); }
static double* h5_read_d(hid_t g, const char* n, int nd, hsize_t* dims) { return (double*)h5_read(g, n, nd, dims, H5T_NATIVE_DOUBLE, sizeof(double)); }
static int*    h5_read_i(hid_t g, const char* n, int nd, hsize_t* dims) { return (int*)   h5_read(g, n, nd, dims, H5T_NATIVE_INT,    sizeof(int)); }

static int h5_attr(hid_t g, const char* path, const char* attr, hid_t memtype, void* out) {
    if (H5Aexists_by_name(g, path, attr, H5P_DEFAULT) <= 0) return 0;
    hid_t a = H5Aopen_by_name(g, path, attr, H5P_DEFAULT, H5P_DEFAULT);
    H5Aread(a, memtype, out);
    H5Aclose(a);
    return 1;
}
static float h5_attr_f(hid_t g, const char* path, const char* attr) {
    float v = 0.f; if (!h5_attr(g, path, attr, H5T_NATIVE_FLOAT, &v)) fprintf(stderr, "oracle: missing attr %s\n", attr); return v; }
static int h5_attr_i(hid_t g, const char* path, const char* attr, int dflt) {
    int v = dflt; h5_attr(g, path, attr, H5T_NATIVE_INT, &v); return v; }

/* fixed-length string-array attribute (src/h5_support.cpp:72-107) */
static int h5_attr_strings(hid_t g, const char* path, const char* attr, char out[][64], int max_n) {
    hid_t a = H5Aopen_by_name(g, path, attr, H5P_DEFAULT, H5P_DEFAULT);
    if (a < 0) return -1;
    hid_t s = H5Aget_space(a), t = H5Aget_type(a);
    size_t maxchars = H5Tget_size(t);
    hsize_t dims[1] = {0};
    int nd = H5Sget_simple_extent_ndims(s);
    if (nd == 1) H5Sget_simple_extent_dims(s, dims, NULL);
    char* tmp = (char*)xcalloc(dims[0] * maxchars + 1, 1);
    if (dims[0]) H5Aread(a, t, tmp);
    int n = 0;
    for (hsize_t i = 0; i < dims[0] && n < max_n; ++i, ++n) {
        size_t len = maxchars < 63 ? maxchars : 63;
        memcpy(out[n], tmp + i * maxchars, len);
        out[n][len] = 0;
    }
    free(tmp); H5Tclose(t); H5Sclose(s); H5Aclose(a);
    return n;
}

/* ============================================================================================
 * engine structures  (src/deriv_engine.h:48-237)
 * ========================================================================================== */
typedef struct { float* x; int row_width; int n_elem; } VecArray;     /* src/vector_math.h:38-92 */
#define VA(a, comp, elem) ((a).x[(comp) + (elem) * (a).row_width])

static VecArray va_alloc(int width, int n_elem) {
    VecArray a; a.row_width = ru(width); a.n_elem = n_elem;
    a.x = (float*)xcalloc((size_t)a.row_width * (n_elem + 4), sizeof(float));
    return a;
}
static void va_fill(VecArray a, float v) { for (int i = 0; i < a.row_width * a.n_elem; ++i) a.x[i] = v; }

struct Node;
typedef void (*node_fn)(struct OracleEngine*, struct Node*, int mode);
typedef void (*node_bwd)(struct OracleEngine*, struct Node*);

typedef struct Node {
    char name[64];
    int potential_term;
    int n_elem, elem_width;
    VecArray output, sens;
    float potential;
    int n_parent; int parents[MAX_ARGS];
    int n_child;  int children[MAX_NODES];
    int germ_level, deriv_level;
    node_fn compute_value;
    node_bwd propagate_deriv;
    void* data;
} Node;

struct OracleEngine {
    int n_node;
    Node nodes[MAX_NODES];
    int n_atom;
    float potential;
    int rotamer_iterations;
};
typedef struct OracleEngine Engine;

enum { DerivMode = 0, PotentialAndDerivMode = 1 };

static Node* parent(Engine* e, Node* n, int i) { return &e->nodes[n->parents[i]]; }

/* ============================================================================================
 * splines  (src/spline.h, src/spline.cpp)
 * ========================================================================================== */

/* src/spline.h:136-174 uniform_deBoor_algorithm; value -> r[0], derivative -> r[1] */
static void uniform_deBoor(float r[2], float c00, float c01, float c02, float c03, float excess) {
    float yu1 = excess + 2.f, yu2 = excess + 1.f, yu3 = excess;
    float frac13 = 1.f / 3.f;
    float alpha11 = frac13 * yu1, alpha12 = frac13 * yu2, alpha13 = frac13 * yu3;
    float c11 = (1.f - alpha11) * c00 + alpha11 * c01; float d11 = c01 - c00;
    float c12 = (1.f - alpha12) * c01 + alpha12 * c02; float d12 = c02 - c01;
    float c13 = (1.f - alpha13) * c02 + alpha13 * c03; float d13 = c03 - c02;
    float alpha22 = 0.5f * yu2, alpha23 = 0.5f * yu3;
    float c22 = (1.f - alpha22) * c11 + alpha22 * c12; float d22 = (1.f - alpha22) * d11 + alpha22 * d12;
    float c23 = (1.f - alpha23) * c12 + alpha23 * c13; float d23 = (1.f - alpha23) * d12 + alpha23 * d13;
    float alpha33 = yu3;
    r[0] = (1.f - alpha33) * c22 + alpha33 * c23;
    r[1] = (1.f - alpha33) * d22 + alpha33 * d23;
}

/* src/spline.h:228-242 deBoor_value_and_deriv: x_bin=trunc(x), window starts at coeff[x_bin-1] */
static void deBoor_vd(float r[2], const float* c, float x) {
    int x_bin = (int)x;
    float y = x - (float)x_bin;
    const float* p = c + (x_bin - 1);
    uniform_deBoor(r, p[0], p[1], p[2], p[3], y);
}

/* src/spline.h:275-310 clamped_deBoor_value_and_deriv (Float4 form: too_small = x<1, too_big = n_knot-2<=x) */
static void clamped_deBoor_vd(float r[2], const float* c, float x, int n_knot) {
    int too_small = x < 1.f;
    int too_big = (float)(n_knot - 2) <= x;
    float xc = (too_small || too_big) ? 1.f : x;
    deBoor_vd(r, c, xc);
    if (too_small || too_big) {
        r[1] = 0.f;
        if (too_small) r[0] = (1.f / 6.f) * c[0] + (2.f / 3.f) * c[1] + (1.f / 6.f) * c[2];
        if (too_big)   r[0] = (1.f / 6.f) * c[n_knot - 3] + (2.f / 3.f) * c[n_knot - 2] + (1.f / 6.f) * c[n_knot - 1];
    }
}

/* src/spline.h:268-272 scalar clamped variant (x<=1, x>=n_knot-2) used by NonlinearCoupling */
static void clamped_deBoor_vd_scalar(float r[2], const float* c, float x, int n_knot) {
    if (x <= 1.f) { r[0] = (1.f / 6.f) * c[0] + (2.f / 3.f) * c[1] + (1.f / 6.f) * c[2]; r[1] = 0.f; return; }
    if (x >= (float)(n_knot - 2)) {
        r[0] = (1.f / 6.f) * c[n_knot - 3] + (2.f / 3.f) * c[n_knot - 2] + (1.f / 6.f) * c[n_knot - 1]; r[1] = 0.f; return; }
    /* src/spline.h:97-128 scalar de Boor == the uniform algorithm */
    deBoor_vd(r, c, x);
}

/* src/spline.h:318-336, 375-392 */
static void deBoor_coeff_deriv(int* starting_bin, float result[4], float x) {
    int x_bin = (int)x;
    *starting_bin = x_bin - 1;
    float y = x - x_bin + 1.f;
    for (int i = 0; i < 4; ++i) {
        float dc[4] = {0.f, 0.f, 0.f, 0.f}; dc[i] = 1.f;
        float r[2]; deBoor_vd(r, dc, y); result[i] = r[0];
    }
}
static void clamped_deBoor_coeff_deriv(int* starting_bin, float result[4], float x, int n_knot) {
    if (x <= 1.f) { *starting_bin = 0; result[0] = 1.f / 6.f; result[1] = 2.f / 3.f; result[2] = 1.f / 6.f; result[3] = 0.f; }
    else if (x >= n_knot - 2) { *starting_bin = n_knot - 4; result[0] = 0.f; result[1] = 1.f / 6.f; result[2] = 2.f / 3.f; result[3] = 1.f / 6.f; }
    else deBoor_coeff_deriv(starting_bin, result, x);
}

/* src/spline.cpp:7-30 Thomas algorithm */
static void solve_tridiagonal_system(int n, double* d, double* a, double* b, double* c) {
    for (int k = 1; k < n; ++k) { double m = a[k - 1] / b[k - 1]; b[k] -= m * c[k - 1]; d[k] -= m * d[k - 1]; }
    d[n - 1] = d[n - 1] / b[n - 1];
    for (int k = n - 2; k >= 0; --k) d[k] = (d[k] - c[k] * d[k + 1]) / b[k];
}

/* src/spline.cpp:33-76 Sherman-Morrison periodic solve */
static void solve_periodic_tridiagonal_system(int n, double* solution, double* d, double* a, double* b, double* c, double* tmp) {
    double b1 = b[0], cn = c[n - 1], ratio = a[0] / b1;
    b[0] += b1; b[n - 1] += ratio * cn;
    memcpy(tmp, a, n * sizeof(double)); memcpy(tmp + n, b, n * sizeof(double)); memcpy(tmp + 2 * n, c, n * sizeof(double));
    solution[0] = -b1; for (int i = 1; i < n - 1; ++i) solution[i] = 0.; solution[n - 1] = cn;
    double* q = solution;
    solve_tridiagonal_system(n, q, a + 1, b, c);
    memcpy(a, tmp, n * sizeof(double)); memcpy(b, tmp + n, n * sizeof(double)); memcpy(c, tmp + 2 * n, n * sizeof(double));
    double* y = d;
    solve_tridiagonal_system(n, d, a + 1, b, c);
    double q_prefactor = (y[0] - y[n - 1] * ratio) / (1. + q[0] - q[n - 1] * ratio);
    for (int i = 0; i < n; ++i) solution[i] = y[i] - q_prefactor * q[i];
}

static const double bspline_coeffs[4][4] = {   /* src/spline.cpp:114-118 */
    {0. / 6., 0. / 6., 0. / 6., 1. / 6.}, {1. / 6., 3. / 6., 3. / 6., -3. / 6.},
    {4. / 6., 0. / 6., -6. / 6., 3. / 6.}, {1. / 6., -3. / 6., 3. / 6., -1. / 6.}};

/* src/spline.cpp:121-156 */
static void solve_periodic_1d_spline(int n, double* coefficients, const double* data, double* ts) {
    double *a = ts, *b = ts + n, *c = ts + 2 * n, *d = ts + 3 * n, *solution = ts + 4 * n, *later = ts + 5 * n;
    for (int i = 0; i < n; ++i) { a[i] = 1. / 6.; b[i] = 2. / 3.; c[i] = 1. / 6.; d[i] = data[i]; }
    solve_periodic_tridiagonal_system(n, solution, d, a, b, c, later);
    for (int i = 0; i < 4 * n; ++i) coefficients[i] = 0.;
    for (int i = 0; i < n; ++i) {
        double v = solution[i];
        for (int inc = 0; inc < 4; ++inc) {
            int idx = i + inc - 2; if (idx < 0) idx += n; if (idx >= n) idx -= n;
            for (int k = 0; k < 4; ++k) coefficients[idx * 4 + k] += v * bspline_coeffs[inc][k];
        }
    }
}

/* src/spline.cpp:262-292 */
static void solve_periodic_2d_spline(int nx, int ny, double* coefficients, const double* data, double* ts) {
    int sum_dim = nx + ny;
    double* splines_1d = ts;
    double* scratch = splines_1d + nx * ny * 4;
    double* values_temp = scratch + sum_dim * 8;
    double* coeffs_temp = values_temp + sum_dim * 4;
    for (int ix = 0; ix < nx; ++ix) solve_periodic_1d_spline(ny, splines_1d + ix * ny * 4, data + ix * ny, scratch);
    for (int iy = 0; iy < ny; ++iy)
        for (int py = 0; py < 4; ++py) {
            for (int ix = 0; ix < nx; ++ix) values_temp[ix] = splines_1d[ix * ny * 4 + iy * 4 + py];
            solve_periodic_1d_spline(nx, coeffs_temp, values_temp, scratch);
            for (int ix = 0; ix < nx; ++ix)
                for (int px = 0; px < 4; ++px) coefficients[ix * ny * 16 + iy * 16 + px * 4 + py] = coeffs_temp[ix * 4 + px];
        }
}

/* src/spline.h:396-451 LayeredPeriodicSpline2D<NDIM> (fit in double, store float) */
typedef struct { int n_layer, nx, ny, ndim; float* coeff; } PeriodicSpline2D;

static void spline2d_fit(PeriodicSpline2D* s, const double* data) {
    int nx = s->nx, ny = s->ny, nd = s->ndim;
    s->coeff = (float*)xcalloc((size_t)s->n_layer * nx * ny * nd * 16, sizeof(float));
    double* coeff_tmp = (double*)xcalloc((size_t)nx * ny * 16, sizeof(double));
    double* data_tmp = (double*)xcalloc((size_t)nx * ny, sizeof(double));
    double* ts = (double*)xcalloc((size_t)(nx + 8) * (ny + 8) * 4 + 64 * (nx + ny), sizeof(double));
    for (int il = 0; il < s->n_layer; ++il)
        for (int id = 0; id < nd; ++id) {
            for (int ix = 0; ix < nx; ++ix) for (int iy = 0; iy < ny; ++iy)
                data_tmp[ix * ny + iy] = data[((size_t)(il * nx + ix) * ny + iy) * nd + id];
            solve_periodic_2d_spline(nx, ny, coeff_tmp, data_tmp, ts);
            for (int ix = 0; ix < nx; ++ix) for (int iy = 0; iy < ny; ++iy) for (int ic = 0; ic < 16; ++ic)
                s->coeff[(((size_t)(il * nx + ix) * ny + iy) * nd + id) * 16 + ic] = (float)coeff_tmp[(ix * ny + iy) * 16 + ic];
        }
    free(coeff_tmp); free(data_tmp); free(ts);
}

/* src/spline.h:60-80 bicubic patch value + derivatives */
static void spline_value_and_deriv2(float* value, float* dx, float* dy, const float* c, float fx, float fy) {
    float fx2 = fx * fx, fx3 = fx * fx2, fy2 = fy * fy;
    float vx0 = c[0] + fy * (c[1] + fy * (c[2] + fy * c[3]));
    float vx1 = c[4] + fy * (c[5] + fy * (c[6] + fy * c[7]));
    float vx2 = c[8] + fy * (c[9] + fy * (c[10] + fy * c[11]));
    float vx3 = c[12] + fy * (c[13] + fy * (c[14] + fy * c[15]));
    float vy1 = c[1] + fx * (c[5] + fx * (c[9] + fx * c[13]));
    float vy2 = c[2] + fx * (c[6] + fx * (c[10] + fx * c[14]));
    float vy3 = c[3] + fx * (c[7] + fx * (c[11] + fx * c[15]));
    *dx = vx1 + 2.f * fx * vx2 + 3.f * fx2 * vx3;
    *dy = vy1 + 2.f * fy * vy2 + 3.f * fy2 * vy3;
    *value = vx0 + fx * vx1 + fx2 * vx2 + fx3 * vx3;
}
/* src/spline.h:434-450 */
static void spline2d_eval(const PeriodicSpline2D* s, float* value, float* dx, float* dy, int layer, float x, float y) {
    int x_bin = (int)x, y_bin = (int)y;
    float fx = x - x_bin, fy = y - y_bin;
    const float* c = s->coeff + ((size_t)layer * s->nx * s->ny + (size_t)x_bin * s->ny + y_bin) * 16 * s->ndim;
    for (int id = 0; id < s->ndim; ++id) spline_value_and_deriv2(value + id, dx + id, dy + id, c + id * 16, fx, fy);
}

/* src/spline.cpp:158-189 */
static void solve_clamped_1d_spline_for_bsplines(int n_coeff, double* coefficients, const double* data, double* ts) {
    int n = n_coeff - 2;
    double *a = ts, *b = ts + n_coeff, *c = ts + 2 * n_coeff;
    for (int i = 0; i < n; ++i) { a[i] = 1. / 6.; b[i] = 2. / 3.; c[i] = 1. / 6.; coefficients[i + 1] = data[i]; }
    a[n - 1] *= 2.; c[0] *= 2.;
    solve_tridiagonal_system(n_coeff - 2, coefficients + 1, a + 1, b, c);
    coefficients[0] = coefficients[2];
    coefficients[n_coeff - 1] = coefficients[n_coeff - 3];
}

/* ============================================================================================
 * small vector helpers
 * ========================================================================================== */
typedef struct { float v[3]; } f3;
static f3 f3_make(float x, float y, float z) { f3 r = {{x, y, z}}; return r; }
static f3 f3_load(VecArray a, int i) { return f3_make(VA(a, 0, i), VA(a, 1, i), VA(a, 2, i)); }
static f3 f3_sub(f3 a, f3 b) { return f3_make(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]); }
static f3 f3_add(f3 a, f3 b) { return f3_make(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]); }
static f3 f3_scale(float s, f3 a) { return f3_make(s * a.v[0], s * a.v[1], s * a.v[2]); }
static float f3_dot(f3 a, f3 b) { return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2]; }
static float f3_mag2(f3 a) { return f3_dot(a, a); }
static f3 f3_cross(f3 a, f3 b) {
    return f3_make(a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]); }
static void f3_update(VecArray a, int i, f3 d) { VA(a, 0, i) += d.v[0]; VA(a, 1, i) += d.v[1]; VA(a, 2, i) += d.v[2]; }

/* src/affine.h:98-108 */
static void quat_to_rot(float* U, const float* q) {
    float a = q[0], b = q[1], c = q[2], d = q[3];
    U[0] = a * a + b * b - c * c - d * d; U[1] = 2.f * b * c - 2.f * a * d; U[2] = 2.f * b * d + 2.f * a * c;
    U[3] = 2.f * b * c + 2.f * a * d; U[4] = a * a - b * b + c * c - d * d; U[5] = 2.f * c * d - 2.f * a * b;
    U[6] = 2.f * b * d - 2.f * a * c; U[7] = 2.f * c * d + 2.f * a * b; U[8] = a * a - b * b - c * c + d * d;
}
static f3 apply_rotation(const float* U, f3 r) {   /* src/affine.h:8-15 */
    return f3_make(U[0] * r.v[0] + U[1] * r.v[1] + U[2] * r.v[2], U[3] * r.v[0] + U[4] * r.v[1] + U[5] * r.v[2],
                   U[6] * r.v[0] + U[7] * r.v[1] + U[8] * r.v[2]); }
static f3 apply_inverse_rotation(const float* U, f3 r) {   /* src/affine.h:21-28 */
    return f3_make(U[0] * r.v[0] + U[3] * r.v[1] + U[6] * r.v[2], U[1] * r.v[0] + U[4] * r.v[1] + U[7] * r.v[2],
                   U[2] * r.v[0] + U[5] * r.v[1] + U[8] * r.v[2]); }
static f3 apply_affine(const float* U, f3 t, f3 r) {   /* src/affine.h:34-40 */
    return f3_make(U[0] * r.v[0] + U[1] * r.v[1] + U[2] * r.v[2] + t.v[0], U[3] * r.v[0] + U[4] * r.v[1] + U[5] * r.v[2] + t.v[1],
                   U[6] * r.v[0] + U[7] * r.v[1] + U[8] * r.v[2] + t.v[2]); }

/* src/vector_math.h:626-631 : value 1/(1+exp(-x)), derivative exp(-x)/(1+exp(-x))^2 */
static void sigmoid(float r[2], float x) { float z = expf(-x); float w = rcpf(1.f + z); r[0] = w; r[1] = z * w * w; }

/* src/vector_math.h:639-658 compact_sigmoid (non-NONCOMPACT branch) */
static void compact_sigmoid(float r[2], float x, float sharpness) {
    float y = x * sharpness;
    r[0] = 0.25f * (y + 2.f) * (y - 1.f) * (y - 1.f);
    r[1] = (sharpness * 0.75f) * (sqr(y) - 1.f);
    if (y < -1.f) { r[0] = 1.f; r[1] = 0.f; }
    else if (1.f < y) { r[0] = 0.f; r[1] = 0.f; }
}

/* src/vector_math.h:703-735 dihedral_germ (Blondel & Karplus) */
static float dihedral_germ(f3 r1, f3 r2, f3 r3, f3 r4, f3* d1, f3* d2, f3* d3, f3* d4) {
    f3 F = f3_sub(r1, r2), G = f3_sub(r2, r3), H = f3_sub(r4, r3);
    f3 A = f3_cross(F, G), B = f3_cross(H, G), C = f3_cross(B, A);
    float inv_Amag2 = rcpf(f3_mag2(A)), inv_Bmag2 = rcpf(f3_mag2(B));
    float Gmag2 = f3_mag2(G), inv_Gmag = rsqrtf_(Gmag2), Gmag = Gmag2 * inv_Gmag;
    *d1 = f3_scale(-Gmag * inv_Amag2, A);
    *d4 = f3_scale(Gmag * inv_Bmag2, B);
    f3 f_mid = f3_sub(f3_scale(f3_dot(F, G) * inv_Amag2 * inv_Gmag, A), f3_scale(f3_dot(H, G) * inv_Bmag2 * inv_Gmag, B));
    *d2 = f3_add(f3_scale(-1.f, *d1), f_mid);
    *d3 = f3_sub(f3_scale(-1.f, *d4), f_mid);
    return atan2f(f3_dot(C, G), f3_dot(A, B) * Gmag);
}

/* ============================================================================================
 * bonded potentials  (src/bonds.cpp)
 * ========================================================================================== */
typedef struct { int n; int* id; float* equil; float* k; } SpringData;

static void dist_spring_value(Engine* e, Node* n, int mode) {   /* src/bonds.cpp:297-318 */
    SpringData* d = (SpringData*)n->data; Node* pos = parent(e, n, 0);
    float pot = 0.f;
    for (int nt = 0; nt < d->n; ++nt) {
        f3 x1 = f3_load(pos->output, d->id[nt * 2]), x2 = f3_load(pos->output, d->id[nt * 2 + 1]);
        f3 disp = f3_sub(x1, x2);
        float inv_mag = rsqrtf_(f3_mag2(disp));
        f3 deriv = f3_scale(d->k[nt] * (1.f - d->equil[nt] * inv_mag), disp);
        pot += 0.5f * d->k[nt] * sqr(sqrtf(f3_mag2(disp)) - d->equil[nt]);
        f3_update(pos->sens, d->id[nt * 2], deriv);
        f3_update(pos->sens, d->id[nt * 2 + 1], f3_scale(-1.f, deriv));
    }
    if (mode == PotentialAndDerivMode) n->potential = pot;
}

static void angle_spring_value(Engine* e, Node* n, int mode) {   /* src/bonds.cpp:457-487 */
    SpringData* d = (SpringData*)n->data; Node* pos = parent(e, n, 0);
    float pot = 0.f;
    for (int nt = 0; nt < d->n; ++nt) {
        f3 a1 = f3_load(pos->output, d->id[nt * 3]), a2 = f3_load(pos->output, d->id[nt * 3 + 1]), a3 = f3_load(pos->output, d->id[nt * 3 + 2]);
        f3 x1 = f3_sub(a1, a3); float inv_d1 = rsqrtf_(f3_mag2(x1)); f3 x1h = f3_scale(inv_d1, x1);
        f3 x2 = f3_sub(a2, a3); float inv_d2 = rsqrtf_(f3_mag2(x2)); f3 x2h = f3_scale(inv_d2, x2);
        float dp = f3_dot(x1h, x2h);
        float pref = d->k[nt] * (dp - d->equil[nt]);
        f3 d1 = f3_scale(pref * inv_d1, f3_sub(x2h, f3_scale(dp, x1h)));
        f3 d2 = f3_scale(pref * inv_d2, f3_sub(x1h, f3_scale(dp, x2h)));
        f3 d3 = f3_scale(-1.f, f3_add(d1, d2));
        f3_update(pos->sens, d->id[nt * 3], d1); f3_update(pos->sens, d->id[nt * 3 + 1], d2); f3_update(pos->sens, d->id[nt * 3 + 2], d3);
        pot += 0.5f * d->k[nt] * sqr(dp - d->equil[nt]);
    }
    if (mode == PotentialAndDerivMode) n->potential = pot;
}

static void dihedral_spring_value(Engine* e, Node* n, int mode) {   /* src/bonds.cpp:519-545 */
    SpringData* d = (SpringData*)n->data; Node* pos = parent(e, n, 0);
    float pot = 0.f;
    for (int nt = 0; nt < d->n; ++nt) {
        f3 x[4], dd[4];
        for (int a = 0; a < 4; ++a) x[a] = f3_load(pos->output, d->id[nt * 4 + a]);
        float dihedral = dihedral_germ(x[0], x[1], x[2], x[3], &dd[0], &dd[1], &dd[2], &dd[3]);
        float disp = dihedral - d->equil[nt];
        disp = (disp > M_PI_F) ? disp - 2.f * M_PI_F : disp;
        disp = (disp < -M_PI_F) ? disp + 2.f * M_PI_F : disp;
        float s = d->k[nt] * disp;
        for (int a = 0; a < 4; ++a) f3_update(pos->sens, d->id[nt * 4 + a], f3_scale(s, dd[a]));
        pot += 0.5f * d->k[nt] * sqr(disp);
    }
    if (mode == PotentialAndDerivMode) n->potential = pot;
}

typedef struct { int n; int* id; float* radius; float* k; } CavityData;
static void cavity_radial_value(Engine* e, Node* n, int mode) {   /* src/bonds.cpp:350-372 */
    CavityData* d = (CavityData*)n->data; Node* pos = parent(e, n, 0);
    float pot = 0.f;
    for (int nt = 0; nt < d->n; ++nt) {
        f3 x = f3_load(pos->output, d->id[nt]);
        float r2 = f3_mag2(x);
        if (r2 > sqr(d->radius[nt])) {
            float inv_r = rsqrtf_(r2), r = r2 * inv_r, excess = r - d->radius[nt];
            pot += 0.5f * d->k[nt] * sqr(excess);
            f3_update(pos->sens, d->id[nt], f3_scale(d->k[nt] * excess * inv_r, x));
        }
    }
    if (mode == PotentialAndDerivMode) n->potential = pot;
}

/* ---- rama_coord (src/bonds.cpp:171-249) ---- */
typedef struct { int n; int* atom; int* dummy; f3* jac; } RamaCoordData;

static void rama_coord_value(Engine* e, Node* n, int mode) {
    (void)mode;
    RamaCoordData* d = (RamaCoordData*)n->data; Node* pos = parent(e, n, 0);
    for (int nt = 0; nt < d->n; ++nt) {
        f3 x[5];
        for (int a = 0; a < 5; ++a) x[a] = f3_load(pos->output, d->atom[nt * 5 + a]);
        for (int pp = 0; pp < 2; ++pp) {
            f3 dd[5]; for (int a = 0; a < 5; ++a) dd[a] = f3_make(0.f, 0.f, 0.f);
            if (d->dummy[nt * 2 + pp]) VA(n->output, pp, nt) = -1.3963f;
            else VA(n->output, pp, nt) = dihedral_germ(x[0 + pp], x[1 + pp], x[2 + pp], x[3 + pp], &dd[0 + pp], &dd[1 + pp], &dd[2 + pp], &dd[3 + pp]);
            for (int a = 0; a < 5; ++a) d->jac[(nt * 2 + pp) * 5 + a] = dd[a];
        }
    }
}
static void rama_coord_deriv(Engine* e, Node* n) {
    RamaCoordData* d = (RamaCoordData*)n->data; Node* pos = parent(e, n, 0);
    for (int nt = 0; nt < d->n; ++nt) {
        float s0 = VA(n->sens, 0, nt), s1 = VA(n->sens, 1, nt);
        for (int a = 0; a < 5; ++a) {
            f3 j0 = d->jac[(nt * 2 + 0) * 5 + a], j1 = d->jac[(nt * 2 + 1) * 5 + a];
            int at = d->atom[nt * 5 + a];
            for (int k = 0; k < 3; ++k) VA(pos->sens, k, at) = s0 * j0.v[k] + (s1 * j1.v[k] + VA(pos->sens, k, at));
        }
    }
}

/* ============================================================================================
 * affine_alignment (src/eig.cpp).  The reference runs the 4x4 symmetric QR on 4 residues at a time with
 * control flow decided by any()/none() over the 4 SIMD lanes (eig.cpp:255-267); that grouping is
 * reproduced here with LANES-wide arrays because it changes how many QR sweeps a residue receives.
 * ========================================================================================== */
#define LANES 4
typedef struct { float v[LANES]; } S4;

static int r_idx(int i, int j) {   /* src/eig.cpp:86-102 */
    int ii = i < j ? i : j, jj = i < j ? j : i;
    switch (ii) { case 0: return jj; case 1: return 3 + jj; case 2: return 5 + jj; case 3: return 6 + jj; }
    return 1000;
}
#define FORL for (int l = 0; l < LANES; ++l)

static S4 house(int n, S4* x) {   /* src/eig.cpp:56-73 */
    S4 beta;
    FORL {
        float sigma2 = 1e-20f;
        for (int i = 1; i < n; ++i) sigma2 += x[i].v[l] * x[i].v[l];
        /* A vector that is zero to within 1e-9 (a residue lying EXACTLY in a coordinate plane of its reference frame: the first residue
         * of an ideal chain built at the origin) needs no reflection.  The formulas below are not a reflection there -- the 1e-20 added
         * to sigma2 dominates, beta comes out as 1 instead of 2 / |v|^2 and the transformed matrix loses an eigenvalue; the reference is
         * spared by the rounding noise of its fused arithmetic (it returns the true eigenvector, as this branch does). */
        if (x[0].v[l] * x[0].v[l] + sigma2 < 1e-18f) { for (int i = 1; i < n; ++i) x[i].v[l] = 0.f; beta.v[l] = 0.f; continue; }
        float mu = sqrtf(x[0].v[l] * x[0].v[l] + sigma2);
        float s = (0.f < x[0].v[l]) ? -sigma2 * rcpf(x[0].v[l] + mu) : x[0].v[l] - mu;
        beta.v[l] = 2.f * s * s * rcpf(sigma2 + s * s);
        x[0].v[l] = mu;
        for (int i = 1; i < n; ++i) x[i].v[l] *= rcpf(s);
    }
    return beta;
}

static void symmetric_tridiagonalize_4x4(S4* beta, S4* A) {   /* src/eig.cpp:105-134 */
    for (int k = 0; k < 2; ++k) {
        int m = 4 - (k + 1);
        beta[k] = house(m, A + r_idx(k, k + 1));
        FORL {
            float p[3], w[3];
#define vv(j) ((j) == 0 ? 1.f : A[r_idx(k, k + 1 + (j))].v[l])
            for (int i = 0; i < m; ++i) {
                p[i] = 0.f;
                for (int j = 0; j < m; ++j) p[i] += A[r_idx(i + k + 1, j + k + 1)].v[l] * vv(j);
                p[i] *= beta[k].v[l];
            }
            float p_dot_v = 0.f;
            for (int i = 0; i < m; ++i) p_dot_v += p[i] * vv(i);
            for (int i = 0; i < m; ++i) w[i] = p[i] - (0.5f * beta[k].v[l] * p_dot_v) * vv(i);
            for (int i = 0; i < m; ++i) for (int j = i; j < m; ++j) A[r_idx(i + k + 1, j + k + 1)].v[l] -= vv(i) * w[j] + vv(j) * w[i];
#undef vv
        }
    }
}

static void unpack_tridiagonalize_4x4(S4* d, S4* u, S4* rot_, S4* beta, S4* A) {   /* src/eig.cpp:137-178 */
#define rot(i, j) (rot_[(i) * 4 + (j)].v[l])
    FORL {
        for (int i = 0; i < 4; ++i) d[i].v[l] = A[r_idx(i, i)].v[l];
        for (int i = 0; i < 3; ++i) u[i].v[l] = A[r_idx(i, i + 1)].v[l];
        for (int i = 0; i < 4; ++i) for (int j = i; j < 4; ++j) rot(i, j) = (i == j);
#define v0(j) ((j) == 0 ? 1.f : A[r_idx(0, 1 + (j))].v[l])
#define v1(j) ((j) == 0 ? 1.f : A[r_idx(1, 2 + (j))].v[l])
        for (int i = 1; i < 4; ++i) for (int j = i; j < 4; ++j) rot(i, j) -= beta[0].v[l] * (v0(i - 1) * v0(j - 1));
        for (int i = 2; i < 4; ++i) for (int j = i; j < 4; ++j) rot(i, j) -= beta[1].v[l] * (v1(i - 2) * v1(j - 2));
        for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) rot(j, i) = rot(i, j);
        float coeff = beta[0].v[l] * beta[1].v[l] * (v0(1) * v1(0) + v0(2) * v1(1));
        for (int i = 2; i < 4; ++i) for (int j = 1; j < 4; ++j) rot(i, j) += coeff * v1(i - 2) * v0(j - 1);
#undef v0
#undef v1
    }
#undef rot
}

static void implicit_symm_QR_step_4x4(int n, S4* d, S4* u, S4* rot_) {   /* src/eig.cpp:183-228 */
#define rot(i, j) (rot_[(i) * 4 + (j)].v[l])
    FORL {
        float dval = 0.5f * (d[n - 2].v[l] - d[n - 1].v[l]) + 1e-20f;
        float un = u[n - 2].v[l];
        float mu = d[n - 1].v[l] - un * un * rcpf(dval + copysignf(sqrtf(dval * dval + un * un), dval));
        float x = d[0].v[l] - mu, z = u[0].v[l];
        for (int k = 0; k < n - 1; ++k) {
            float inv_r = rsqrtf_(x * x + z * z);     /* givens, src/eig.cpp:75-82 */
            int trivial = (z == 0.f);
            float c = trivial ? 1.f : x * inv_r;
            float s = trivial ? 0.f : -z * inv_r;
            for (int j = 0; j < 4; ++j) {
                float t1 = rot(k, j), t2 = rot(k + 1, j);
                rot(k, j) = c * t1 - s * t2;
                rot(k + 1, j) = s * t1 + c * t2;
            }
            if (k > 0) u[k - 1].v[l] = c * x - s * z;
            float T00 = d[k].v[l], T11 = d[k + 1].v[l], T01 = u[k].v[l];
            d[k].v[l] = T00 * c * c - T01 * 2.f * c * s + T11 * s * s;
            d[k + 1].v[l] = T00 * s * s + T01 * 2.f * c * s + T11 * c * c;
            u[k].v[l] = (T00 - T11) * c * s + T01 * (c * c - s * s);
            x = u[k].v[l];
            if (k < n - 2) { z = -u[k + 1].v[l] * s; u[k + 1].v[l] *= c; }
        }
    }
#undef rot
}

static int symm_QR_4x4(S4* d, S4* rot, S4* A, float tol, int max_iter) {   /* src/eig.cpp:232-273 */
    const int n = 4;
    S4 beta[2], u[3];
    symmetric_tridiagonalize_4x4(beta, A);
    unpack_tridiagonalize_4x4(d, u, rot, beta, A);
    for (int k = 0; k < max_iter; ++k) {
        for (int i = 0; i < n - 1; ++i) FORL {
            if (fabsf(u[i].v[l]) <= tol * (fabsf(d[i].v[l]) + fabsf(d[i + 1].v[l]))) u[i].v[l] = 0.f; }
        int any_u[3] = {0, 0, 0};
        for (int i = 0; i < 3; ++i) FORL if (u[i].v[l] != 0.f) any_u[i] = 1;
        int q;
        if (any_u[2]) q = 0; else if (any_u[1]) q = 1; else if (any_u[0]) q = 2; else return k;
        int p;
        for (p = n - q - 1; p > 0; --p) if (!any_u[p - 1]) break;
        implicit_symm_QR_step_4x4(n - q - p, d + p, u + p, rot + 4 * p);
    }
    return -1;
}

typedef struct { int n_res, n_group; int* atoms; float* ref_geom; S4* evals; S4* evecs; } AffineData;

static void affine_alignment_value(Engine* e, Node* n, int mode) {   /* src/eig.cpp:317-386 */
    (void)mode;
    AffineData* d = (AffineData*)n->data; Node* pos = parent(e, n, 0);
    for (int ng = 0; ng < d->n_group; ++ng) {
        S4 F[10]; f3 center[LANES];
        FORL {
            int nr = ng * 4 + l; if (nr >= d->n_res) nr = ng * 4;   /* padding duplicates lane 0 (eig.cpp:307-314) */
            f3 a1 = f3_load(pos->output, d->atoms[nr * 3 + 0]), a2 = f3_load(pos->output, d->atoms[nr * 3 + 1]), a3 = f3_load(pos->output, d->atoms[nr * 3 + 2]);
            f3 c = f3_scale(1.f / 3.f, f3_add(f3_add(a1, a2), a3));
            a1 = f3_sub(a1, c); a2 = f3_sub(a2, c); a3 = f3_sub(a3, c);
            const float* g = d->ref_geom + nr * 9;
            float R[3][3];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = a1.v[j] * g[0 + i] + a2.v[j] * g[3 + i] + a3.v[j] * g[6 + i];
            float Fv[10] = {R[0][0] + R[1][1] + R[2][2], R[1][2] - R[2][1], R[2][0] - R[0][2], R[0][1] - R[1][0],
                            R[0][0] - R[1][1] - R[2][2], R[0][1] + R[1][0], R[0][2] + R[2][0],
                            -R[0][0] + R[1][1] - R[2][2], R[1][2] + R[2][1], -R[0][0] - R[1][1] + R[2][2]};
            for (int i = 0; i < 10; ++i) F[i].v[l] = Fv[i];
            center[l] = c;
        }
        S4* evals = d->evals + ng * 4; S4* evecs = d->evecs + ng * 16;
        symm_QR_4x4(evals, evecs, F, 1e-5f, 100);
        FORL {
            for (int i = 1; i < 4; ++i) if (evals[0].v[l] < evals[i].v[l]) {
                float t = evals[0].v[l]; evals[0].v[l] = evals[i].v[l]; evals[i].v[l] = t;
                for (int k = 0; k < 4; ++k) { float tt = evecs[k].v[l]; evecs[k].v[l] = evecs[i * 4 + k].v[l]; evecs[i * 4 + k].v[l] = tt; }
            }
            int nr = ng * 4 + l;
            if (nr < round_up(d->n_res, 4)) {
                for (int j = 0; j < 3; ++j) VA(n->output, j, nr) = center[l].v[j];
                for (int j = 0; j < 4; ++j) VA(n->output, 3 + j, nr) = evecs[j].v[l];
            }
        }
    }
}

static void affine_alignment_deriv(Engine* e, Node* n) {   /* src/eig.cpp:388-470 */
    AffineData* d = (AffineData*)n->data; Node* pos = parent(e, n, 0);
    for (int nr = 0; nr < d->n_res; ++nr) {
        int ng = nr / 4, l = nr % 4;
        const S4* evals = d->evals + ng * 4; const S4* evecs = d->evecs + ng * 16;
#define EV(k, i) (evecs[(k) * 4 + (i)].v[l])
        float inv_evals[4];
        for (int j = 1; j < 4; ++j) inv_evals[j] = rcpf(evals[0].v[l] - evals[j].v[l]);
        float sens3[3] = {VA(n->sens, 0, nr), VA(n->sens, 1, nr), VA(n->sens, 2, nr)};
        float torque[3] = {VA(n->sens, 3, nr), VA(n->sens, 4, nr), VA(n->sens, 5, nr)};
        float quat_sens[4] = {
            2.f * (-torque[0] * EV(0, 1) - torque[1] * EV(0, 2) - torque[2] * EV(0, 3)),
            2.f * (torque[0] * EV(0, 0) + torque[1] * EV(0, 3) - torque[2] * EV(0, 2)),
            2.f * (torque[1] * EV(0, 0) + torque[2] * EV(0, 1) - torque[0] * EV(0, 3)),
            2.f * (torque[2] * EV(0, 0) + torque[0] * EV(0, 2) - torque[1] * EV(0, 1))};
        float qsdb[4] = {0.f, 0.f, 0.f, 0.f};
        for (int dd = 0; dd < 4; ++dd) for (int i = 0; i < 4; ++i) qsdb[dd] += quat_sens[i] * EV(dd, i);
        for (int na = 0; na < 3; ++na) {
            float deriv[3] = {(1.f / 3.f) * sens3[0], (1.f / 3.f) * sens3[1], (1.f / 3.f) * sens3[2]};
            const float* g = d->ref_geom + nr * 9 + na * 3;
            /* the three symmetric 4x4 perturbation matrices dF/d(atom coordinate) (eig.cpp:452-465) */
            float f[3][10] = {
                {g[0], 0.f, g[2], -g[1], g[0], g[1], g[2], -g[0], 0.f, -g[0]},
                {g[1], -g[2], 0.f, g[0], -g[1], g[0], 0.f, g[1], g[2], -g[1]},
                {g[2], g[1], -g[0], 0.f, -g[2], 0.f, g[0], -g[2], g[1], g[2]}};
            for (int c = 0; c < 3; ++c)
                for (int k = 1; k < 4; ++k) {
                    float acc = 0.f;
                    for (int i = 0; i < 4; ++i) for (int j = i; j < 4; ++j) {
                        float t = (i == j) ? EV(k, i) * EV(0, j) : EV(k, i) * EV(0, j) + EV(k, j) * EV(0, i);
                        acc += f[c][r_idx(i, j)] * t;
                    }
                    deriv[c] += (inv_evals[k] * acc) * qsdb[k];
                }
            int at = d->atoms[nr * 3 + na];
            for (int c = 0; c < 3; ++c) VA(pos->sens, c, at) += deriv[c];
        }
#undef EV
    }
}

/* ============================================================================================
 * infer_H_O (src/hbond.cpp:14-121)
 * ========================================================================================== */
typedef struct { int n_virtual; int* atom; float* bond_length; float* dfd; } InferData;

static void infer_value(Engine* e, Node* n, int mode) {
    (void)mode;
    InferData* d = (InferData*)n->data; Node* pos = parent(e, n, 0);
    for (int nv = 0; nv < d->n_virtual; ++nv) {
        f3 prev_c = f3_load(pos->output, d->atom[nv * 3]), curr_c = f3_load(pos->output, d->atom[nv * 3 + 1]), next_c = f3_load(pos->output, d->atom[nv * 3 + 2]);
        f3 prev = f3_sub(prev_c, curr_c); float prev_im = rsqrtf_(f3_mag2(prev)); prev = f3_scale(prev_im, prev);
        f3 next = f3_sub(next_c, curr_c); float next_im = rsqrtf_(f3_mag2(next)); next = f3_scale(next_im, next);
        f3 disp = f3_add(prev, next); float disp_im = rsqrtf_(f3_mag2(disp)); disp = f3_scale(disp_im, disp);
        f3 dir = f3_scale(-1.f, disp);
        f3 hp = f3_add(f3_scale(d->bond_length[nv], dir), curr_c);
        float* s = d->dfd + nv * 12;
        for (int k = 0; k < 3; ++k) { s[k] = prev.v[k]; s[4 + k] = next.v[k]; s[8 + k] = disp.v[k]; }
        s[3] = prev_im; s[7] = next_im; s[11] = disp_im;
        for (int k = 0; k < 3; ++k) { VA(n->output, k, nv) = hp.v[k]; VA(n->output, 3 + k, nv) = dir.v[k]; }
    }
}
static void infer_deriv(Engine* e, Node* n) {
    InferData* d = (InferData*)n->data; Node* pos = parent(e, n, 0);
    for (int nv = 0; nv < d->n_virtual; ++nv) {
        f3 sens_pos = f3_load(n->sens, nv);
        f3 sens_dir = f3_make(VA(n->sens, 3, nv), VA(n->sens, 4, nv), VA(n->sens, 5, nv));
        f3 snu = f3_add(sens_dir, f3_scale(d->bond_length[nv], sens_pos));   /* sens_neg_unitdisp */
        const float* s = d->dfd + nv * 12;
        f3 prev = f3_make(s[0], s[1], s[2]), next = f3_make(s[4], s[5], s[6]), disp = f3_make(s[8], s[9], s[10]);
        float prev_im = s[3], next_im = s[7], disp_im = s[11];
        /* fmsub(a,b,c) = a*b - c */
        f3 sn_disp = f3_scale(disp_im, f3_sub(f3_scale(f3_dot(disp, snu), disp), snu));
        f3 sn_prev = f3_scale(-prev_im, f3_sub(f3_scale(f3_dot(prev, sn_disp), prev), sn_disp));
        f3 sn_next = f3_scale(-next_im, f3_sub(f3_scale(f3_dot(next, sn_disp), next), sn_disp));
        f3_update(pos->sens, d->atom[nv * 3], sn_prev);
        f3_update(pos->sens, d->atom[nv * 3 + 1], f3_sub(f3_sub(sens_pos, sn_prev), sn_next));
        f3_update(pos->sens, d->atom[nv * 3 + 2], sn_next);
    }
}

/* ============================================================================================
 * interaction graph  (src/interaction_graph.h)
 * ========================================================================================== */
enum { IT_ROTAMER = 0, IT_HBOND_COVERAGE = 1, IT_ENVIRONMENT = 2, IT_PROTEIN_HBOND = 3, IT_RADIAL = 4, IT_HBOND_SC_RADIAL = 5 };

typedef struct {
    int itype, symmetric;
    int n_dim1, n_dim2, n_param;
    int n_knot, n_knot_angular; float inv_dx, inv_dtheta;   /* quadspline shape (bead_interaction.h:12-27) */
    int n_elem1, n_elem2, n_type1, n_type2;
    int *loc1, *loc2, *types1, *types2, *id1, *id2;
    float* param;
    float cutoff;
    float *pos1, *pos2;             /* packed copies, stride 8 */
    int max_n_edge, n_edge;
    int *edge_i1, *edge_i2;
    float* edge_value; float* edge_deriv;   /* (n_dim1+n_dim2) per edge */
    float* edge_sens;
    float *pos1_deriv, *pos2_deriv;
} IGraph;

static int acceptable_id_pair(int itype, int id1, int id2) {
    switch (itype) {
        case IT_ROTAMER: return ((unsigned)id1 >> N_BIT_ROTAMER) != ((unsigned)id2 >> N_BIT_ROTAMER);   /* bead_interaction.h:195-197 */
        case IT_HBOND_COVERAGE:                                                                           /* hbond.cpp:254-259 */
        case IT_RADIAL: case IT_HBOND_SC_RADIAL:                                                          /* sidechain_radial.cpp:41-44 */
        case IT_ENVIRONMENT: return (2 < id1 - id2) || (2 < id2 - id1);                                   /* environment.cpp:22-25 */
        default: return 1;                                                                                /* hbond.cpp:162-164 */
    }
}

static float igraph_type_cutoff(const IGraph* g, const float* p) {
    switch (g->itype) {
        case IT_ROTAMER: case IT_HBOND_COVERAGE:
            return (float)((g->n_knot - 2 - 1e-6) / g->inv_dx);         /* bead_interaction.h:191-193, hbond.cpp:250-252 */
        case IT_ENVIRONMENT: return p[0] + 1.f / p[1];                  /* environment.cpp:18-20, vector_math.h:660-666 */
        case IT_RADIAL: case IT_HBOND_SC_RADIAL: return (float)((16 - 2 - 1e-6) / p[0]);   /* sidechain_radial.cpp:31-34 */
        default: return sqrtf(3.5f * 3.5f);                             /* hbond.cpp:124,158-160 */
    }
}

static void igraph_update_cutoffs(IGraph* g) {   /* interaction_graph.h:383-398 */
    g->cutoff = 0.f;
    for (int t1 = 0; t1 < g->n_type1; ++t1) for (int t2 = 0; t2 < g->n_type2; ++t2)
        g->cutoff = maxf(g->cutoff, igraph_type_cutoff(g, g->param + (t1 * g->n_type2 + t2) * g->n_param));
}

static int igraph_init(IGraph* g, hid_t grp, int itype) {   /* interaction_graph.h:305-381 */
    memset(g, 0, sizeof(*g));
    g->itype = itype; g->symmetric = (itype == IT_ROTAMER || itype == IT_RADIAL);
    hsize_t dims[3];
    g->param = h5_read_f(grp, "interaction_param", 3, dims);
    if (!g->param) return -1;
    g->n_type1 = (int)dims[0]; g->n_type2 = (int)dims[1]; g->n_param = (int)dims[2];
    switch (itype) {
        case IT_ROTAMER: g->n_dim1 = 6; g->n_dim2 = 6; break;
        case IT_HBOND_COVERAGE: g->n_dim1 = 7; g->n_dim2 = 6; break;
        case IT_ENVIRONMENT: g->n_dim1 = 6; g->n_dim2 = 4; if (g->n_param != 4) return -1; break;
        case IT_PROTEIN_HBOND: g->n_dim1 = 6; g->n_dim2 = 6; if (g->n_param != 8) return -1; break;
        case IT_RADIAL: case IT_HBOND_SC_RADIAL: g->n_dim1 = 3; g->n_dim2 = 3; if (g->n_param != 17) return -1; break;
    }
    if (itype == IT_ROTAMER || itype == IT_HBOND_COVERAGE) {
        /* compile-time knot counts of the reference become run-time: n_param = 2*ka + 2*k
         * (bead_interaction.h:12-27,187-189; hbond.cpp:246-248) */
        int np = g->n_param;
        if (itype == IT_ROTAMER) {
            if (np == 34) { g->n_knot_angular = 8; g->n_knot = 9; g->inv_dx = 1.f; }
            else if (np == 40) { g->n_knot_angular = 8; g->n_knot = 12; g->inv_dx = 1.f; }
            else if (np == 62) { g->n_knot_angular = 15; g->n_knot = 16; g->inv_dx = 2.f; }
            else return -1;
        } else {
            if (np == 30) { g->n_knot_angular = 8; g->n_knot = 7; g->inv_dx = 1.f; }
            else if (np == 40) { g->n_knot_angular = 8; g->n_knot = 12; g->inv_dx = 1.f; }
            else if (np == 54) { g->n_knot_angular = 15; g->n_knot = 12; g->inv_dx = 2.f; }
            else return -1;
        }
        g->inv_dtheta = (g->n_knot_angular - 3) / 2.f;
    }
    igraph_update_cutoffs(g);
    const char* s1 = g->symmetric ? "" : "1";
    char nm[32];
    snprintf(nm, 32, "index%s", s1); g->loc1 = h5_read_i(grp, nm, 1, dims); if (!g->loc1) return -1; g->n_elem1 = (int)dims[0];
    snprintf(nm, 32, "type%s", s1);  g->types1 = h5_read_i(grp, nm, 1, dims);
    snprintf(nm, 32, "id%s", s1);    g->id1 = h5_read_i(grp, nm, 1, dims);
    if (!g->types1 || !g->id1) return -1;
    if (!g->symmetric) {
        g->loc2 = h5_read_i(grp, "index2", 1, dims); if (!g->loc2) return -1; g->n_elem2 = (int)dims[0];
        g->types2 = h5_read_i(grp, "type2", 1, dims); g->id2 = h5_read_i(grp, "id2", 1, dims);
        if (!g->types2 || !g->id2) return -1;
    } else { g->n_elem2 = g->n_elem1; g->loc2 = g->loc1; g->types2 = g->types1; g->id2 = g->id1; }
    g->max_n_edge = round_up(g->n_elem1 * g->n_elem2 / (g->symmetric ? 2 : 1), 16) + 16;
    g->pos1 = (float*)xcalloc((size_t)g->n_elem1 * 8, sizeof(float));
    g->pos2 = g->symmetric ? g->pos1 : (float*)xcalloc((size_t)g->n_elem2 * 8, sizeof(float));
    g->pos1_deriv = (float*)xcalloc((size_t)g->n_elem1 * 8, sizeof(float));
    g->pos2_deriv = g->symmetric ? g->pos1_deriv : (float*)xcalloc((size_t)g->n_elem2 * 8, sizeof(float));
    g->edge_i1 = (int*)xcalloc(g->max_n_edge, sizeof(int)); g->edge_i2 = (int*)xcalloc(g->max_n_edge, sizeof(int));
    g->edge_value = (float*)xcalloc(g->max_n_edge, sizeof(float));
    g->edge_sens = (float*)xcalloc(g->max_n_edge, sizeof(float));
    g->edge_deriv = (float*)xcalloc((size_t)g->max_n_edge * 16, sizeof(float));
    return 0;
}

/* bead_interaction.h:30-84 quadspline: coverage value, d1 (n_dim1 first 6), d2 (6) */
static float quadspline(const IGraph* g, float* d1, float* d2, const float* p, const float* x1, const float* x2) {
    int ka = g->n_knot_angular, k = g->n_knot; float inv_dx = g->inv_dx, inv_dtheta = g->inv_dtheta;
    f3 displace = f3_make(x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]);
    f3 rvec1 = f3_make(x1[3], x1[4], x1[5]), rvec2 = f3_make(x2[3], x2[4], x2[5]);
    float dist2 = f3_mag2(displace), inv_dist = rsqrtf_(dist2);
    float dist_coord = dist2 * (inv_dist * inv_dx);
    f3 u = f3_scale(inv_dist, displace);
    float cos1 = f3_dot(rvec1, u), cos2 = f3_dot(rvec2, f3_scale(-1.f, u));
    float a1[2], a2[2], wide[2], narrow[2];
    deBoor_vd(a1, p, (cos1 + 1.f) * inv_dtheta + 1.f);
    deBoor_vd(a2, p + ka, (cos2 + 1.f) * inv_dtheta + 1.f);
    clamped_deBoor_vd(wide, p + 2 * ka, dist_coord, k);
    clamped_deBoor_vd(narrow, p + 2 * ka + k, dist_coord, k);
    float angular_weight = a1[0] * a2[0];
    float radial_deriv = inv_dx * (wide[1] + angular_weight * narrow[1]);
    float angular_deriv1 = inv_dtheta * a1[1] * a2[0] * narrow[0];
    float angular_deriv2 = inv_dtheta * a1[0] * a2[1] * narrow[0];
    f3 rXX = f3_sub(f3_scale(angular_deriv1, rvec1), f3_scale(angular_deriv2, rvec2));
    f3 deriv_dir = f3_scale(inv_dist, f3_sub(rXX, f3_scale(f3_dot(u, rXX), u)));
    f3 d_displace = f3_add(f3_scale(radial_deriv, u), deriv_dir);
    f3 d_rvec1 = f3_scale(angular_deriv1, u), d_rvec2 = f3_scale(-angular_deriv2, u);
    float coverage = wide[0] + angular_weight * narrow[0];
    for (int c = 0; c < 3; ++c) { d1[c] = -d_displace.v[c]; d1[3 + c] = d_rvec1.v[c]; d2[c] = d_displace.v[c]; d2[3 + c] = d_rvec2.v[c]; }
    return coverage;
}

/* sidechain_radial.cpp:46-61 */
static float radial_edge(float* d1, float* d2, const float* p, const float* x1, const float* x2) {
    float inv_dx = p[0];
    f3 disp = f3_make(x1[0] - x2[0], x1[1] - x2[1], x1[2] - x2[2]);
    float dist2 = f3_mag2(disp), inv_dist = rsqrtf_(dist2 + 1e-7f);
    float dist_coord = dist2 * (inv_dist * inv_dx);
    float en[2]; clamped_deBoor_vd(en, p + 1, dist_coord, 16);
    for (int c = 0; c < 3; ++c) { d1[c] = disp.v[c] * (inv_dist * inv_dx * en[1]); d2[c] = -d1[c]; }
    return en[0];
}

/* hbond.cpp:261-276 */
static float hbond_coverage_edge(const IGraph* g, float* d1, float* d2, const float* p, const float* x1, const float* x2) {
    float coverage = quadspline(g, d1, d2, p, x1, x2);
    float prefactor = sqr(1.f - x1[6]);
    for (int c = 0; c < 6; ++c) { d1[c] *= prefactor; d2[c] *= prefactor; }
    d1[6] = -coverage * (1.f - x1[6]) * 2.f;
    return prefactor * coverage;
}

/* environment.cpp:27-60 */
static float environment_edge(float* d1, float* d2, const float* p, const float* cb, const float* sc) {
    f3 displace = f3_make(sc[0] - cb[0], sc[1] - cb[1], sc[2] - cb[2]);
    f3 rvec1 = f3_make(cb[3], cb[4], cb[5]);
    float prob = sc[3];
    float dist2 = f3_mag2(displace), inv_dist = rsqrtf_(dist2), dist = dist2 * inv_dist;
    f3 u = f3_scale(inv_dist, displace);
    float r0 = p[0], r_sharp = p[1], dot0 = p[2], dot_sharp = p[3];
    float dp = f3_dot(u, rvec1);
    float rs[2], as[2];
    compact_sigmoid(rs, dist - r0, r_sharp);
    compact_sigmoid(as, dot0 - dp, dot_sharp);
    f3 dd = f3_scale(prob, f3_sub(f3_scale(rs[1] * as[0], u), f3_scale(rs[0] * as[1] * inv_dist, f3_sub(rvec1, f3_scale(dp, u)))));
    for (int c = 0; c < 3; ++c) { d1[3 + c] = -prob * rs[0] * as[1] * u.v[c]; d1[c] = -dd.v[c]; d2[c] = dd.v[c]; }
    float score = rs[0] * as[0];
    d2[3] = score;
    return prob * score;
}

/* hbond.cpp:128-148,166-230.  `group_active` reproduces the reference's SIMD quirk: the angular cut-off is
 * applied to a group of 4 consecutive edges at once -- `if(none(within_angular_cutoff))` -- so an edge
 * outside the angular cone still gets the (tiny) sigmoid value when a group-mate is inside it. */
static float protein_hbond_edge(float* d1, float* d2, const float* p, const float* x1, const float* x2, int group_active) {
    f3 H = f3_make(x1[0], x1[1], x1[2]), O = f3_make(x2[0], x2[1], x2[2]);
    f3 rHN = f3_make(x1[3], x1[4], x1[5]), rOC = f3_make(x2[3], x2[4], x2[5]);
    f3 HO = f3_sub(H, O);
    float magHO2 = f3_mag2(HO) + 1e-6f, invHOmag = rsqrtf_(magHO2), magHO = magHO2 * invHOmag;
    f3 rHO = f3_scale(invHOmag, HO);
    float dotHOC = f3_dot(rHO, rOC), dotOHN = -f3_dot(rHO, rHN);
    f3 dH = f3_make(0, 0, 0), dO = dH, drHN = dH, drOC = dH;
    float hb = 0.f;
    if (group_active) {
        float os[2], is[2], radial[2], ang1[2], ang2[2], v[2];
        sigmoid(os, (p[2] - magHO) * p[3]);
        sigmoid(is, (magHO - p[0]) * p[1]);
        radial[0] = os[0] * is[0];
        radial[1] = -p[3] * os[1] * is[0] + p[1] * is[1] * os[0];
        sigmoid(v, (dotHOC - p[4]) * p[5]); ang1[0] = v[0]; ang1[1] = p[5] * v[1];
        sigmoid(v, (dotOHN - p[4]) * p[5]); ang2[0] = v[0]; ang2[1] = p[5] * v[1];
        hb = radial[0] * ang1[0] * ang2[0];
        float c0 = radial[1] * ang1[0] * ang2[0];
        float c1 = radial[0] * ang1[1] * ang2[0];
        float c2 = -radial[0] * ang1[0] * ang2[1];
        drOC = f3_scale(c1, rHO);
        drHN = f3_scale(c2, rHO);
        dH = f3_add(f3_add(f3_scale(c0, rHO), f3_scale(c1 * invHOmag, f3_sub(rOC, f3_scale(dotHOC, rHO)))),
                    f3_scale(c2 * invHOmag, f3_add(rHN, f3_scale(dotOHN, rHO))));
        dO = f3_scale(-1.f, dH);
    }
    float hb_log = (1.f <= hb) ? 100.f : -logf(1.f - hb);
    float pref = minf(rcpf(1.f - hb), 1e5f);
    for (int c = 0; c < 3; ++c) { d1[c] = dH.v[c] * pref; d1[3 + c] = drHN.v[c] * pref; d2[c] = dO.v[c] * pref; d2[3 + c] = drOC.v[c] * pref; }
    return hb_log;
}
static int protein_hbond_within_cone(const float* x1, const float* x2) {
    f3 HO = f3_make(x1[0] - x2[0], x1[1] - x2[1], x1[2] - x2[2]);
    float invHOmag = rsqrtf_(f3_mag2(HO) + 1e-6f);
    f3 rHO = f3_scale(invHOmag, HO);
    float dotHOC = f3_dot(rHO, f3_make(x2[3], x2[4], x2[5])), dotOHN = -f3_dot(rHO, f3_make(x1[3], x1[4], x1[5]));
    return (0.f < dotHOC) && (0.f < dotOHN);
}

/* interaction_graph.h:443-504 compute_edges; the pair list is rebuilt from scratch every call, which yields
 * the same membership and order as the reference's cached list (interaction_graph.h:116-158, 211-256):
 * lexicographic in (i1>>2, i2, i1&3) with dist2 < cutoff^2 and acceptable_id_pair. */
static void igraph_compute_edges(IGraph* g, Node* n1, Node* n2) {
    for (int i = 0; i < g->n_elem1; ++i) for (int c = 0; c < g->n_dim1; ++c) g->pos1[i * 8 + c] = VA(n1->output, c, g->loc1[i]);
    if (!g->symmetric) for (int i = 0; i < g->n_elem2; ++i) for (int c = 0; c < g->n_dim2; ++c) g->pos2[i * 8 + c] = VA(n2->output, c, g->loc2[i]);
    float cutoff2 = sqr(g->cutoff);
    int ne = 0;
    for (int i1b = 0; i1b < g->n_elem1; i1b += 4)
        for (int i2 = g->symmetric ? i1b + 1 : 0; i2 < g->n_elem2; ++i2)
            for (int i1 = i1b; i1 < i1b + 4 && i1 < g->n_elem1; ++i1) {
                if (g->symmetric && !(i1 < i2)) continue;
                const float* a = g->pos1 + i1 * 8; const float* b = g->pos2 + i2 * 8;
                float dist2 = sqr(a[0] - b[0]) + sqr(a[1] - b[1]) + sqr(a[2] - b[2]);
                if (!(dist2 < cutoff2)) continue;
                if (!acceptable_id_pair(g->itype, g->id1[i1], g->id2[i2])) continue;
                g->edge_i1[ne] = i1; g->edge_i2[ne] = i2; ++ne;
            }
    g->n_edge = ne;
    int nd = g->n_dim1 + g->n_dim2;
    for (int gb = 0; gb < ne; gb += 4) {
        int group_active = 0;
        if (g->itype == IT_PROTEIN_HBOND)
            for (int e = gb; e < gb + 4 && e < ne; ++e)
                group_active |= protein_hbond_within_cone(g->pos1 + g->edge_i1[e] * 8, g->pos2 + g->edge_i2[e] * 8);
        for (int e = gb; e < gb + 4 && e < ne; ++e) {
            int i1 = g->edge_i1[e], i2 = g->edge_i2[e];
            const float* p = g->param + (g->types1[i1] * g->n_type2 + g->types2[i2]) * g->n_param;
            float* d1 = g->edge_deriv + (size_t)e * nd; float* d2 = d1 + g->n_dim1;
            const float* x1 = g->pos1 + i1 * 8; const float* x2 = g->pos2 + i2 * 8;
            switch (g->itype) {
                case IT_ROTAMER: g->edge_value[e] = quadspline(g, d1, d2, p, x1, x2); break;
                case IT_HBOND_COVERAGE: g->edge_value[e] = hbond_coverage_edge(g, d1, d2, p, x1, x2); break;
                case IT_ENVIRONMENT: g->edge_value[e] = environment_edge(d1, d2, p, x1, x2); break;
                case IT_RADIAL: case IT_HBOND_SC_RADIAL: g->edge_value[e] = radial_edge(d1, d2, p, x1, x2); break;
                default: g->edge_value[e] = protein_hbond_edge(d1, d2, p, x1, x2, group_active); break;
            }
        }
    }
}

/* interaction_graph.h:507-556 */
static void igraph_propagate(IGraph* g, Node* n1, Node* n2) {
    memset(g->pos1_deriv, 0, (size_t)g->n_elem1 * 8 * sizeof(float));
    if (!g->symmetric) memset(g->pos2_deriv, 0, (size_t)g->n_elem2 * 8 * sizeof(float));
    int nd = g->n_dim1 + g->n_dim2;
    for (int e = 0; e < g->n_edge; ++e) {
        float s = g->edge_sens[e];
        const float* d1 = g->edge_deriv + (size_t)e * nd; const float* d2 = d1 + g->n_dim1;
        float* t1 = g->pos1_deriv + g->edge_i1[e] * 8; float* t2 = g->pos2_deriv + g->edge_i2[e] * 8;
        for (int c = 0; c < g->n_dim1; ++c) t1[c] += s * d1[c];
        for (int c = 0; c < g->n_dim2; ++c) t2[c] += s * d2[c];
    }
    for (int i = 0; i < g->n_elem1; ++i) for (int c = 0; c < g->n_dim1; ++c) VA(n1->sens, c, g->loc1[i]) += g->pos1_deriv[i * 8 + c];
    if (!g->symmetric) for (int i = 0; i < g->n_elem2; ++i) for (int c = 0; c < g->n_dim2; ++c) VA(n2->sens, c, g->loc2[i]) += g->pos2_deriv[i * 8 + c];
}

/* bead_interaction.h:86-130: derivative of the quadspline value w.r.t. the n_param spline coefficients */
static void quadspline_param_deriv(const IGraph* g, float* d_param, const float* p, const float* x1, const float* x2) {
    int ka = g->n_knot_angular, k = g->n_knot; float inv_dx = g->inv_dx, inv_dtheta = g->inv_dtheta;
    for (int i = 0; i < g->n_param; ++i) d_param[i] = 0.f;
    f3 displace = f3_make(x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]);
    f3 rvec1 = f3_make(x1[3], x1[4], x1[5]), rvec2 = f3_make(x2[3], x2[4], x2[5]);
    float dist2 = f3_mag2(displace), inv_dist = rsqrtf_(dist2);
    float dist_coord = dist2 * (inv_dist * inv_dx);
    f3 u = f3_scale(inv_dist, displace);
    float cos1 = f3_dot(rvec1, u), cos2 = f3_dot(rvec2, f3_scale(-1.f, u));
    float a1[2], a2[2], narrow[2], result[4]; int sb;
    deBoor_vd(a1, p, (cos1 + 1.f) * inv_dtheta + 1.f);
    deBoor_vd(a2, p + ka, (cos2 + 1.f) * inv_dtheta + 1.f);
    clamped_deBoor_coeff_deriv(&sb, result, dist_coord, k);
    for (int i = 0; i < 4; ++i) d_param[2 * ka + sb + i] = result[i];                           /* wide_cover */
    for (int i = 0; i < 4; ++i) d_param[2 * ka + k + sb + i] = a1[0] * a2[0] * result[i];       /* narrow_cover */
    clamped_deBoor_vd_scalar(narrow, p + 2 * ka + k, dist_coord, k);
    deBoor_coeff_deriv(&sb, result, (cos1 + 1.f) * inv_dtheta + 1.f);
    for (int i = 0; i < 4; ++i) d_param[sb + i] = a2[0] * narrow[0] * result[i];
    deBoor_coeff_deriv(&sb, result, (cos2 + 1.f) * inv_dtheta + 1.f);
    for (int i = 0; i < 4; ++i) d_param[ka + sb + i] = a1[0] * narrow[0] * result[i];
}

/* interaction_graph.h:404-416 get_param_deriv: the edges and edge sensitivities of the last evaluation
 * (compute_edges<true> re-derives the same edges from the same positions), :497-503 and :537-543 */
static void igraph_param_deriv(IGraph* g, float* out) {
    int n = g->n_type1 * g->n_type2 * g->n_param;
    for (int i = 0; i < n; ++i) out[i] = 0.f;
    float* dp = (float*)xcalloc((size_t)g->n_param, sizeof(float));
    for (int e = 0; e < g->n_edge; ++e) {
        int i1 = g->edge_i1[e], i2 = g->edge_i2[e];
        int t = g->types1[i1] * g->n_type2 + g->types2[i2];
        const float* p = g->param + t * g->n_param;
        const float* x1 = g->pos1 + i1 * 8; const float* x2 = g->pos2 + i2 * 8;
        switch (g->itype) {
            case IT_ROTAMER: quadspline_param_deriv(g, dp, p, x1, x2); break;                    /* bead_interaction.h:204-207 */
            case IT_HBOND_COVERAGE: {                                                             /* hbond.cpp:278-283 */
                quadspline_param_deriv(g, dp, p, x1, x2);
                float prefactor = sqr(1.f - x1[6]);
                for (int d = 0; d < g->n_param; ++d) dp[d] *= prefactor;
            } break;
            case IT_ENVIRONMENT: for (int d = 0; d < g->n_param; ++d) dp[d] = 0.f; break;        /* environment.cpp:62-65 */
            case IT_RADIAL: case IT_HBOND_SC_RADIAL: {                                            /* sidechain_radial.cpp:63-77 */
                for (int d = 0; d < g->n_param; ++d) dp[d] = 0.f;
                float dist = sqrtf(sqr(x1[0] - x2[0]) + sqr(x1[1] - x2[1]) + sqr(x1[2] - x2[2]));
                float r[2]; clamped_deBoor_vd_scalar(r, p + 1, p[0] * dist, 16);
                dp[0] = r[1] * dist;
                int sb; float w[4]; clamped_deBoor_coeff_deriv(&sb, w, p[0] * dist, 16);
                for (int k = 0; k < 4; ++k) dp[1 + sb + k] = w[k];
            } break;
            default: for (int d = 0; d < g->n_param; ++d) dp[d] = -1.f; break;                   /* hbond.cpp:232-235 */
        }
        for (int d = 0; d < g->n_param; ++d) out[t * g->n_param + d] += g->edge_sens[e] * dp[d];
    }
    free(dp);
}

static void igraph_count_edges_by_type(IGraph* g, float* out) {   /* interaction_graph.h:427-441 */
    for (int i = 0; i < g->n_type1 * g->n_type2; ++i) out[i] = 0.f;
    for (int e = 0; e < g->n_edge; ++e) out[g->types1[g->edge_i1[e]] * g->n_type2 + g->types2[g->edge_i2[e]]] += 1.f;
}

/* ---- protein_hbond node (hbond.cpp:290-368) ---- */
typedef struct { IGraph g; int n_donor, n_acceptor; float* sens_scaled; } ProteinHBondData;

static void protein_hbond_value(Engine* e, Node* n, int mode) {
    (void)mode;
    ProteinHBondData* d = (ProteinHBondData*)n->data; Node* infer = parent(e, n, 0);
    int nv = d->n_donor + d->n_acceptor;
    for (int i = 0; i < nv; ++i) { for (int c = 0; c < 6; ++c) VA(n->output, c, i) = VA(infer->output, c, i); VA(n->output, 6, i) = 0.f; }
    igraph_compute_edges(&d->g, infer, infer);
    for (int ne = 0; ne < d->g.n_edge; ++ne) {
        float hb_log = d->g.edge_value[ne];
        VA(n->output, 6, d->g.edge_i1[ne]) += hb_log;
        VA(n->output, 6, d->g.edge_i2[ne] + d->n_donor) += hb_log;
    }
    for (int i = 0; i < nv; ++i) VA(n->output, 6, i) = 1.f - expf(-VA(n->output, 6, i));
}
static void protein_hbond_deriv(Engine* e, Node* n) {
    ProteinHBondData* d = (ProteinHBondData*)n->data; Node* infer = parent(e, n, 0);
    int nv = d->n_donor + d->n_acceptor;
    for (int i = 0; i < nv; ++i) d->sens_scaled[i] = VA(n->sens, 6, i) * (1.f - VA(n->output, 6, i));
    for (int ne = 0; ne < d->g.n_edge; ++ne) d->g.edge_sens[ne] = d->sens_scaled[d->g.edge_i1[ne]] + d->sens_scaled[d->g.edge_i2[ne] + d->n_donor];
    igraph_propagate(&d->g, infer, infer);
    for (int nd = 0; nd < d->n_donor; ++nd) for (int c = 0; c < 6; ++c) VA(infer->sens, c, d->g.loc1[nd]) += VA(n->sens, c, nd);
    for (int na = 0; na < d->n_acceptor; ++na) for (int c = 0; c < 6; ++c) VA(infer->sens, c, d->g.loc2[na]) += VA(n->sens, c, na + d->n_donor);
}

/* ---- hbond_coverage node (hbond.cpp:371-414) ---- */
static void hbond_coverage_value(Engine* e, Node* n, int mode) {
    (void)mode;
    IGraph* g = (IGraph*)n->data;
    igraph_compute_edges(g, parent(e, n, 0), parent(e, n, 1));
    va_fill(n->output, 0.f);
    for (int ne = 0; ne < g->n_edge; ++ne) VA(n->output, 0, g->edge_i2[ne]) += g->edge_value[ne];
}
static void hbond_coverage_deriv(Engine* e, Node* n) {
    IGraph* g = (IGraph*)n->data;
    for (int ne = 0; ne < g->n_edge; ++ne) g->edge_sens[ne] = VA(n->sens, 0, g->edge_i2[ne]);
    igraph_propagate(g, parent(e, n, 0), parent(e, n, 1));
}

/* ---- environment_coverage node (environment.cpp:71-108) ---- */
static void environment_coverage_value(Engine* e, Node* n, int mode) {
    (void)mode;
    IGraph* g = (IGraph*)n->data;
    igraph_compute_edges(g, parent(e, n, 0), parent(e, n, 1));
    va_fill(n->output, 0.f);
    for (int ne = 0; ne < g->n_edge; ++ne) VA(n->output, 0, g->edge_i1[ne]) += g->edge_value[ne];
}
static void environment_coverage_deriv(Engine* e, Node* n) {
    IGraph* g = (IGraph*)n->data;
    for (int ne = 0; ne < g->n_edge; ++ne) g->edge_sens[ne] = VA(n->sens, 0, g->edge_i1[ne]);
    igraph_propagate(g, parent(e, n, 0), parent(e, n, 1));
}

/* ---- hbond_energy (hbond.cpp:417-456) ---- */
/* radial (sidechain_radial.cpp:81-104) and hbond_sc_radial (:107-136): unit edge sensitivities, potential = sum of edge values */
static void radial_pairs_value(Engine* e, Node* n, int mode) {
    IGraph* g = (IGraph*)n->data; Node* n1 = parent(e, n, 0); Node* n2 = g->symmetric ? NULL : parent(e, n, 1);
    igraph_compute_edges(g, n1, n2);
    for (int ne = 0; ne < g->n_edge; ++ne) g->edge_sens[ne] = 1.f;
    igraph_propagate(g, n1, n2);
    if (mode == PotentialAndDerivMode) { float pot = 0.f; for (int ne = 0; ne < g->n_edge; ++ne) pot += g->edge_value[ne]; n->potential = pot; }
}

typedef struct { float E_protein; float n_hbond; } HBondEnergyData;
static void hbond_energy_value(Engine* e, Node* n, int mode) {
    (void)mode;
    HBondEnergyData* d = (HBondEnergyData*)n->data; Node* ph = parent(e, n, 0);
    float tot = 0.f;
    for (int nv = 0; nv < ph->n_elem; ++nv) { tot += VA(ph->output, 6, nv); VA(ph->sens, 6, nv) += d->E_protein; }
    n->potential = tot * d->E_protein;
    d->n_hbond = tot;
}

/* ---- weighted_pos (environment.cpp:112-156) ---- */
typedef struct { int* index_pos; int* index_weight; } WeightedPosData;
static void weighted_pos_value(Engine* e, Node* n, int mode) {
    (void)mode;
    WeightedPosData* d = (WeightedPosData*)n->data; Node* pos = parent(e, n, 0); Node* en = parent(e, n, 1);
    for (int i = 0; i < n->n_elem; ++i) {
        for (int c = 0; c < 3; ++c) VA(n->output, c, i) = VA(pos->output, c, d->index_pos[i]);
        VA(n->output, 3, i) = expf(-VA(en->output, 0, d->index_weight[i]));
    }
}
static void weighted_pos_deriv(Engine* e, Node* n) {
    WeightedPosData* d = (WeightedPosData*)n->data; Node* pos = parent(e, n, 0); Node* en = parent(e, n, 1);
    for (int i = 0; i < n->n_elem; ++i) {
        for (int c = 0; c < 3; ++c) VA(pos->sens, c, d->index_pos[i]) += VA(n->sens, c, i);
        VA(en->sens, 0, d->index_weight[i]) -= VA(n->output, 3, i) * VA(n->sens, 3, i);
    }
}

/* ---- nonlinear_coupling (environment.cpp:324-397) ---- */
typedef struct { int n_restype, n_coeff; float offset, inv_dx; float* coeff; int* types; } NonlinearData;
static void nonlinear_coupling_value(Engine* e, Node* n, int mode) {
    (void)mode;
    NonlinearData* d = (NonlinearData*)n->data; Node* in = parent(e, n, 0);
    float pot = 0.f;
    for (int i = 0; i < in->n_elem; ++i) {
        float coord = (VA(in->output, 0, i) - d->offset) * d->inv_dx;
        float v[2]; clamped_deBoor_vd_scalar(v, d->coeff + d->types[i] * d->n_coeff, coord, d->n_coeff);
        pot += v[0];
        VA(in->sens, 0, i) += v[1] * d->inv_dx;
    }
    n->potential = pot;
}

/* ---- rama_map_pot (rama_map_pot.cpp:15-82) ---- */
typedef struct { int n_residue; int* residue; int* map_id; PeriodicSpline2D s; } RamaMapData;
static void rama_map_pot_value(Engine* e, Node* n, int mode) {
    RamaMapData* d = (RamaMapData*)n->data; Node* rama = parent(e, n, 0);
    const float scale = d->s.nx * (0.5f / M_PI_F - 1e-7f);
    const float shift = M_PI_F;
    float pot = 0.f;
    for (int nr = 0; nr < d->n_residue; ++nr) {
        int r = d->residue[nr];
        float value, dx, dy;
        spline2d_eval(&d->s, &value, &dx, &dy, d->map_id[nr], (VA(rama->output, 0, r) + shift) * scale, (VA(rama->output, 1, r) + shift) * scale);
        pot += value;
        VA(rama->sens, 0, r) += dx * scale;
        VA(rama->sens, 1, r) += dy * scale;
    }
    if (mode == PotentialAndDerivMode) n->potential = pot;
}

/* ---- placement nodes (placement.cpp) ---- */
enum { PL_SCALAR = 0, PL_VECTOR = 1, PL_POINT = 2 };
typedef struct {
    int n_elem, n_pos_dim, n_sig; int sig[3];
    int* affine_residue; int* layer; int* rama_residue;
    int is_rama; int n_layer; float* fixed_data;       /* (n_layer, n_pos_dim) */
    PeriodicSpline2D s; float* rama_deriv;             /* (n_elem, 2*n_pos_dim) */
    float* param_deriv;                                /* (n_layer, n_pos_dim), fixed placements (placement.cpp:113-148) */
} PlacementData;

static void placement_value(Engine* e, Node* n, int mode) {   /* placement.cpp:264-281, 60-78, 139-141, 183-201 */
    (void)mode;
    PlacementData* d = (PlacementData*)n->data; Node* aff = parent(e, n, 0);
    Node* rama = d->is_rama ? parent(e, n, 1) : NULL;
    if (d->param_deriv) memset(d->param_deriv, 0, sizeof(float) * (size_t)d->n_layer * d->n_pos_dim);   /* placement.cpp:133-137 reset() */
    for (int ne = 0; ne < d->n_elem; ++ne) {
        int ar = d->affine_residue[ne];
        f3 t = f3_load(aff->output, ar);
        float q[4] = {VA(aff->output, 3, ar), VA(aff->output, 4, ar), VA(aff->output, 5, ar), VA(aff->output, 6, ar)};
        float U[9]; quat_to_rot(U, q);
        float val[8];
        if (d->is_rama) {
            const float scale_x = d->s.nx * (0.5f / M_PI_F - 1e-7f), scale_y = d->s.ny * (0.5f / M_PI_F - 1e-7f);
            int rr = d->rama_residue[ne];
            spline2d_eval(&d->s, val, d->rama_deriv + ne * 2 * d->n_pos_dim, d->rama_deriv + ne * 2 * d->n_pos_dim + d->n_pos_dim,
                          d->layer[ne], (VA(rama->output, 0, rr) + M_PI_F) * scale_x, (VA(rama->output, 1, rr) + M_PI_F) * scale_y);
        } else for (int c = 0; c < d->n_pos_dim; ++c) val[c] = d->fixed_data[d->layer[ne] * d->n_pos_dim + c];
        int off = 0;
        for (int s = 0; s < d->n_sig; ++s) {
            if (d->sig[s] == PL_SCALAR) { VA(n->output, off, ne) = val[off]; off += 1; }
            else {
                f3 v = f3_make(val[off], val[off + 1], val[off + 2]);
                f3 r = d->sig[s] == PL_VECTOR ? apply_rotation(U, v) : apply_affine(U, t, v);
                for (int c = 0; c < 3; ++c) VA(n->output, off + c, ne) = r.v[c];
                off += 3;
            }
        }
    }
}
static void placement_deriv(Engine* e, Node* n) {   /* placement.cpp:283-307, 80-92, 204-230 */
    PlacementData* d = (PlacementData*)n->data; Node* aff = parent(e, n, 0);
    Node* rama = d->is_rama ? parent(e, n, 1) : NULL;
    for (int ne = 0; ne < d->n_elem; ++ne) {
        int ar = d->affine_residue[ne];
        f3 t = f3_load(aff->output, ar);
        float q[4] = {VA(aff->output, 3, ar), VA(aff->output, 4, ar), VA(aff->output, 5, ar), VA(aff->output, 6, ar)};
        float U[9]; quat_to_rot(U, q);
        float ref_sens[8]; f3 com = f3_make(0, 0, 0), torque = f3_make(0, 0, 0);
        int off = 0;
        for (int s = 0; s < d->n_sig; ++s) {
            if (d->sig[s] == PL_SCALAR) { ref_sens[off] = VA(n->sens, off, ne); off += 1; }
            else {
                f3 sv = f3_make(VA(n->sens, off, ne), VA(n->sens, off + 1, ne), VA(n->sens, off + 2, ne));
                f3 xv = f3_make(VA(n->output, off, ne), VA(n->output, off + 1, ne), VA(n->output, off + 2, ne));
                f3 rs = apply_inverse_rotation(U, sv);
                for (int c = 0; c < 3; ++c) ref_sens[off + c] = rs.v[c];
                if (d->sig[s] == PL_POINT) { com = f3_add(com, sv); torque = f3_add(torque, f3_cross(f3_sub(xv, t), sv)); }
                else torque = f3_add(torque, f3_cross(xv, sv));
                off += 3;
            }
        }
        if (d->param_deriv) for (int c = 0; c < d->n_pos_dim; ++c) d->param_deriv[d->layer[ne] * d->n_pos_dim + c] += ref_sens[c];   /* placement.cpp:143-148 */
        if (d->is_rama) {
            const float scale_x = d->s.nx * (0.5f / M_PI_F - 1e-7f), scale_y = d->s.ny * (0.5f / M_PI_F - 1e-7f);
            const float* rd = d->rama_deriv + ne * 2 * d->n_pos_dim;
            float a = 0.f, b = 0.f;
            for (int c = 0; c < d->n_pos_dim; ++c) { a += ref_sens[c] * rd[c]; b += ref_sens[c] * rd[d->n_pos_dim + c]; }
            VA(rama->sens, 0, d->rama_residue[ne]) += scale_x * a;
            VA(rama->sens, 1, d->rama_residue[ne]) += scale_y * b;
        }
        for (int c = 0; c < 3; ++c) { VA(aff->sens, c, ar) += com.v[c]; VA(aff->sens, 3 + c, ar) += torque.v[c]; }
    }
}

/* ---- optional restraint / external-field nodes (bonds.cpp, environment.cpp, sidechain_radial.cpp, membrane_potential.cpp) ---- */
typedef struct { int kind, n; int* id; float* x0; float* k; float* v3; float* a; float* b; float time_initial, time_step; int round_num; } PointData;
static void point_potential_value(Engine* e, Node* n, int mode) {
    PointData* d = (PointData*)n->data; Node* pos = parent(e, n, 0);
    float pot = 0.f;
    if (d->kind == 2 && mode == DerivMode) d->round_num += 1;                     /* bonds.cpp:150 */
    float time_estimate = d->time_initial + d->time_step * d->round_num;          /* bonds.cpp:151 */
    for (int nt = 0; nt < d->n; ++nt) {
        f3 x = f3_load(pos->output, d->id[nt]);
        if (d->kind == 0) {                                                        /* atom_pos_spring, bonds.cpp:36-48 */
            f3 disp = f3_sub(x, f3_make(d->x0[nt * 3], d->x0[nt * 3 + 1], d->x0[nt * 3 + 2]));
            pot += 0.5f * d->k[nt] * f3_mag2(disp);
            f3_update(pos->sens, d->id[nt], f3_scale(d->k[nt], disp));
        } else if (d->kind == 1) {                                                 /* tension, bonds.cpp:73-88 */
            f3 c = f3_make(d->v3[nt * 3], d->v3[nt * 3 + 1], d->v3[nt * 3 + 2]);
            pot -= f3_dot(x, c);
            f3_update(pos->sens, d->id[nt], f3_scale(-1.f, c));
        } else if (d->kind == 2) {                                                 /* AFM, bonds.cpp:147-166 */
            f3 tip = f3_add(f3_make(d->x0[nt * 3], d->x0[nt * 3 + 1], d->x0[nt * 3 + 2]),
                            f3_scale(time_estimate, f3_make(d->v3[nt * 3], d->v3[nt * 3 + 1], d->v3[nt * 3 + 2])));
            f3 diff = f3_sub(x, tip);
            pot += 0.5 * d->k[nt] * f3_mag2(diff);
            f3_update(pos->sens, d->id[nt], f3_scale(d->k[nt], diff));
        } else {                                                                   /* z_flat_bottom, bonds.cpp:406-425 */
            float z = x.v[2], z0 = d->a[nt], radius = d->b[nt];
            float excess = z - z0 > radius ? z - z0 - radius : (z - z0 < -radius ? z - z0 + radius : 0.f);
            VA(pos->sens, 2, d->id[nt]) += d->k[nt] * excess;
            pot += 0.5f * d->k[nt] * sqr(excess);
        }
    }
    if (mode == PotentialAndDerivMode || d->kind == 1 || d->kind == 2) n->potential = pot;
}

typedef struct { int n; int* id; float* energy; float* dist; float* scale; float* cutoff; } ContactData;
static void contact_value(Engine* e, Node* n, int mode) {   /* sidechain_radial.cpp:187-204 */
    (void)mode;
    ContactData* d = (ContactData*)n->data; Node* bead = parent(e, n, 0);
    float pot = 0.f;
    for (int nc = 0; nc < d->n; ++nc) {
        f3 disp = f3_sub(f3_load(bead->output, d->id[nc * 2]), f3_load(bead->output, d->id[nc * 2 + 1]));
        float dist = sqrtf(f3_mag2(disp));
        if (dist >= d->cutoff[nc]) continue;
        float c[2]; compact_sigmoid(c, dist - d->dist[nc], d->scale[nc]);
        pot += d->energy[nc] * c[0];
        f3 deriv = f3_scale(d->energy[nc] * c[1] * rcpf(dist), disp);
        f3_update(bead->sens, d->id[nc * 2], deriv);
        f3_update(bead->sens, d->id[nc * 2 + 1], f3_scale(-1.f, deriv));
    }
    n->potential = pot;
}

typedef struct { float* value; } ConstantData;
static void constant_value(Engine* e, Node* n, int mode) {   /* bonds.cpp:562-564 */
    (void)e; (void)mode;
    ConstantData* d = (ConstantData*)n->data;
    for (int ne = 0; ne < n->n_elem; ++ne) for (int c = 0; c < n->elem_width; ++c) VA(n->output, c, ne) = d->value[ne * n->elem_width + c];
}
static void no_deriv(Engine* e, Node* n) { (void)e; (void)n; }

typedef struct { int* id; } SliceData;
static void slice_value(Engine* e, Node* n, int mode) {   /* bonds.cpp:605-611 */
    (void)mode;
    SliceData* d = (SliceData*)n->data; Node* pos = parent(e, n, 0);
    for (int na = 0; na < n->n_elem; ++na) for (int c = 0; c < n->elem_width; ++c) VA(n->output, c, na) = VA(pos->output, c, d->id[na]);
}
static void slice_deriv(Engine* e, Node* n) {   /* bonds.cpp:613-619 */
    SliceData* d = (SliceData*)n->data; Node* pos = parent(e, n, 0);
    for (int na = 0; na < n->n_elem; ++na) for (int c = 0; c < n->elem_width; ++c) VA(pos->sens, c, d->id[na]) += VA(n->sens, c, na);
}

typedef struct { int n_coeff; float offset, inv_dx; float* coeff; float* jac; } UniformTransformData;
static void uniform_transform_value(Engine* e, Node* n, int mode) {   /* environment.cpp:180-188 */
    (void)mode;
    UniformTransformData* d = (UniformTransformData*)n->data; Node* in = parent(e, n, 0);
    for (int ne = 0; ne < n->n_elem; ++ne) {
        float v[2]; clamped_deBoor_vd_scalar(v, d->coeff, (VA(in->output, 0, ne) - d->offset) * d->inv_dx, d->n_coeff);
        VA(n->output, 0, ne) = v[0];
        d->jac[ne] = v[1] * d->inv_dx;
    }
}
static void uniform_transform_deriv(Engine* e, Node* n) {   /* environment.cpp:190-194 */
    UniformTransformData* d = (UniformTransformData*)n->data; Node* in = parent(e, n, 0);
    for (int ne = 0; ne < n->n_elem; ++ne) VA(in->sens, 0, ne) += d->jac[ne] * VA(n->sens, 0, ne);
}

typedef struct { int n_coupling; float* couplings; int* types; int has_inact, inact_dim; } LinearCouplingData;
static void linear_coupling_value(Engine* e, Node* n, int mode) {   /* environment.cpp:286-300 */
    (void)mode;
    LinearCouplingData* d = (LinearCouplingData*)n->data; Node* in = parent(e, n, 0);
    Node* inact = d->has_inact ? parent(e, n, 1) : NULL;
    float pot = 0.f;
    for (int ne = 0; ne < in->n_elem; ++ne) {
        float c = d->couplings[d->types[ne]];
        float act = inact ? sqr(1.f - VA(inact->output, d->inact_dim, ne)) : 1.f;
        float val = VA(in->output, 0, ne);
        pot += c * val * act;
        VA(in->sens, 0, ne) += c * act;
        if (inact) VA(inact->sens, d->inact_dim, ne) -= c * val;
    }
    n->potential = pot;
}

/* spline.cpp:192-259 and spline.h:456-515 (LayeredClampedSpline1D<1>) */
static void solve_clamped_1d_spline(int n, double* coefficients, const double* data, double* ts) {
    double *a = ts, *b = ts + n, *c = ts + 2 * n, *solution = ts + 3 * n;
    for (int i = 0; i < n; ++i) { a[i] = 1. / 6.; b[i] = 2. / 3.; c[i] = 1. / 6.; solution[i] = data[i]; }
    a[n - 1] *= 2.; c[0] *= 2.;
    solve_tridiagonal_system(n, solution, a + 1, b, c);
    for (int i = 0; i < 4 * (n - 1); ++i) coefficients[i] = 0.;
    for (int i = 0; i < n; ++i)
        for (int inc = 0; inc < 4; ++inc) {
            int idx = i + inc - 2;
            if (idx < 0 || idx >= n - 1) continue;
            for (int k = 0; k < 4; ++k) coefficients[idx * 4 + k] += solution[i] * bspline_coeffs[inc][k];
        }
    for (int k = 0; k < 4; ++k) coefficients[k] += solution[1] * bspline_coeffs[3][k];
    for (int k = 0; k < 4; ++k) coefficients[(n - 2) * 4 + k] += solution[n - 2] * bspline_coeffs[0][k];
}
typedef struct { int n_layer, nx; float* coeff; float* left; float* right; } ClampedSpline1D;
static void clamped1d_fit(ClampedSpline1D* s, const double* data) {
    s->coeff = (float*)xcalloc((size_t)s->n_layer * (s->nx - 1) * 4, sizeof(float));
    s->left = (float*)xcalloc(s->n_layer, sizeof(float)); s->right = (float*)xcalloc(s->n_layer, sizeof(float));
    double* ct = (double*)xcalloc((size_t)(s->nx - 1) * 4, sizeof(double)); double* ts = (double*)xcalloc((size_t)4 * s->nx, sizeof(double));
    for (int il = 0; il < s->n_layer; ++il) {
        s->left[il] = (float)data[il * s->nx]; s->right[il] = (float)data[il * s->nx + s->nx - 1];
        solve_clamped_1d_spline(s->nx, ct, data + (size_t)il * s->nx, ts);
        for (int i = 0; i < (s->nx - 1) * 4; ++i) s->coeff[(size_t)il * (s->nx - 1) * 4 + i] = (float)ct[i];
    }
    free(ct); free(ts);
}
static void clamped1d_eval(const ClampedSpline1D* s, float result[2], int layer, float x) {   /* result = (deriv, value) */
    if (x >= s->nx - 1) { result[0] = 0.f; result[1] = s->right[layer]; }
    else if (x <= 0) { result[0] = 0.f; result[1] = s->left[layer]; }
    else {
        int x_bin = (int)x; float fx = x - x_bin, fx2 = fx * fx, fx3 = fx * fx2;
        const float* c = s->coeff + ((size_t)layer * (s->nx - 1) + x_bin) * 4;
        result[0] = c[1] + 2.f * fx * c[2] + 3.f * fx2 * c[3];
        result[1] = c[0] + fx * c[1] + fx2 * c[2] + fx3 * c[3];
    }
}
typedef struct {
    int n_elem, n_donor, n_acceptor; int *cb_index, *env_index, *restype; float *cov_midpoint, *cov_sharpness;
    ClampedSpline1D cb, uhb; float cb_z_shift, cb_z_scale, uhb_z_shift, uhb_z_scale;
} MembraneData;
static void membrane_value(Engine* e, Node* n, int mode) {   /* membrane_potential.cpp:104-151 */
    (void)mode;
    MembraneData* d = (MembraneData*)n->data;
    Node* cb = parent(e, n, 0); Node* env = parent(e, n, 1); Node* hb = parent(e, n, 2);
    float pot = 0.f;
    for (int nr = 0; nr < d->n_elem; ++nr) {
        float r[2]; clamped1d_eval(&d->cb, r, d->restype[nr], (VA(cb->output, 2, d->cb_index[nr]) + d->cb_z_shift) * d->cb_z_scale);
        float spline_value = r[1], spline_deriv = r[0] * d->cb_z_scale;
        float sg[2]; compact_sigmoid(sg, VA(env->output, 0, d->env_index[nr]) - d->cov_midpoint[d->restype[nr]], d->cov_sharpness[d->restype[nr]]);
        pot += spline_value * sg[0];
        VA(cb->sens, 2, d->cb_index[nr]) += spline_deriv * sg[0];
        VA(env->sens, 0, d->env_index[nr]) += spline_value * sg[1];
    }
    for (int nv = 0; nv < d->n_donor + d->n_acceptor; ++nv) {
        float r[2]; clamped1d_eval(&d->uhb, r, nv >= d->n_donor, (VA(hb->output, 2, nv) + d->uhb_z_shift) * d->uhb_z_scale);
        float spline_value = r[1], spline_deriv = r[0] * d->uhb_z_scale, uhb_prob = 1.f - VA(hb->output, 6, nv);
        pot += spline_value * sqr(uhb_prob);
        VA(hb->sens, 2, nv) += spline_deriv * sqr(uhb_prob);
        VA(hb->sens, 6, nv) += -2.f * spline_value * uhb_prob;
    }
    n->potential = pot;
}

/* ---- backbone_pairs (backbone_steric.cpp) ---- */
typedef struct { int n_res; int* residue; int* id; int* n_atom; float* ref_pos; float dist_cutoff; } BackboneData;
static float nonbonded_kernel(int return_deriv, float r_mag2) {   /* backbone_steric.cpp:18-30 */
    const float energy_scale = 4.f, wall = 3.0f, wall_squared = wall * wall, width = 0.10f, sharpness = 1.f / (wall * width);
    float v[2]; compact_sigmoid(v, r_mag2 - wall_squared, sharpness);
    return return_deriv ? 2.f * (energy_scale * v[1]) : energy_scale * v[0];
}
static void backbone_pairs_value(Engine* e, Node* n, int mode) {   /* backbone_steric.cpp:81-145 */
    BackboneData* d = (BackboneData*)n->data; Node* aff = parent(e, n, 0);
    const float cutoff2_atom = 3.f * 3.f + 0.1f * 3.f;
    int nres = d->n_res;
    f3* coords = (f3*)xcalloc(nres, sizeof(f3)); f3* rp = (f3*)xcalloc((size_t)nres * 4, sizeof(f3));
    for (int nr = 0; nr < nres; ++nr) {
        int ar = d->residue[nr];
        float q[4] = {VA(aff->output, 3, ar), VA(aff->output, 4, ar), VA(aff->output, 5, ar), VA(aff->output, 6, ar)};
        float U[9]; quat_to_rot(U, q);
        f3 t = f3_load(aff->output, ar);
        coords[nr] = t;
        for (int na = 0; na < 4; ++na) rp[nr * 4 + na] = apply_affine(U, t, f3_make(d->ref_pos[(nr * 4 + na) * 3], d->ref_pos[(nr * 4 + na) * 3 + 1], d->ref_pos[(nr * 4 + na) * 3 + 2]));
    }
    float pot = 0.f, cut2 = sqr(d->dist_cutoff);
    /* canonical pair order as the PairlistComputation<true> would emit it */
    for (int i1b = 0; i1b < nres; i1b += 4)
        for (int nr2 = i1b + 1; nr2 < nres; ++nr2)
            for (int nr1 = i1b; nr1 < i1b + 4 && nr1 < nres; ++nr1) {
                if (!(nr1 < nr2)) continue;
                if (!(f3_mag2(f3_sub(coords[nr1], coords[nr2])) < cut2)) continue;
                int i1 = d->id[nr1], i2 = d->id[nr2];
                if (!((1 < i1 - i2) || (1 < i2 - i1))) continue;   /* backbone_steric.cpp:32-35 */
                f3 d1 = f3_make(0, 0, 0), tq1 = d1, d2 = d1, tq2 = d1;
                int hit = 0;
                for (int a1 = 0; a1 < d->n_atom[nr1]; ++a1) for (int a2 = 0; a2 < d->n_atom[nr2]; ++a2) {
                    f3 x1 = rp[nr1 * 4 + a1], x2 = rp[nr2 * 4 + a2];
                    f3 r = f3_sub(x1, x2);
                    float r2 = f3_mag2(r);
                    if (r2 > cutoff2_atom) continue;
                    hit = 1;
                    float dor = nonbonded_kernel(1, r2);
                    pot += nonbonded_kernel(0, r2);
                    f3 g = f3_scale(dor, r);
                    d1 = f3_add(d1, g); tq1 = f3_add(tq1, f3_cross(f3_sub(x1, coords[nr1]), g));
                    f3 mg = f3_scale(-1.f, g);
                    d2 = f3_add(d2, mg); tq2 = f3_add(tq2, f3_cross(f3_sub(x2, coords[nr2]), mg));
                }
                if (hit) for (int c = 0; c < 3; ++c) {
                    VA(aff->sens, c, d->residue[nr1]) += d1.v[c]; VA(aff->sens, 3 + c, d->residue[nr1]) += tq1.v[c];
                    VA(aff->sens, c, d->residue[nr2]) += d2.v[c]; VA(aff->sens, 3 + c, d->residue[nr2]) += tq2.v[c];
                }
            }
    if (mode == PotentialAndDerivMode) n->potential = pot;
    free(coords); free(rp);
}

/* ============================================================================================
 * rotamer node (src/rotamer.cpp)
 * ========================================================================================== */
typedef struct {
    int n_rot, n_elem;
    float* prob; float* cur; float* old; float* energy_offset;   /* (n_elem, n_rot) */
} NodeHolder;
typedef struct { int edge_num, dim, ne; } EdgeLoc;
typedef struct {
    int n_rot1, n_rot2;
    NodeHolder *nodes1, *nodes2;
    int max_n_edge, n_edge, prev_n_edge;
    float* prob;                      /* (edge, n_rot1*n_rot2) */
    float* cur; float* old;           /* (edge, n_rot1+n_rot2) */
    float* marginal;                  /* (edge, n_rot1*n_rot2) */
    int *ei1, *ei2;
    int* locator;                     /* dense (n_elem1, n_elem2) slot table, replaces EdgeLocator (rotamer.cpp:134-206) */
    EdgeLoc* edge_loc; int n_edge_loc, cap_edge_loc;
} EdgeHolder;

typedef struct {
    IGraph g;
    int n_prob_nodes;
    int n_elem_rot[7];
    NodeHolder nodes[7];       /* index by n_rot: 1,3,6 used */
    EdgeHolder edges[7][7];    /* [1][1],[1][3],[1][6],[3][3],[3][6],[6][6] */
    float damping, tol; int max_iter, iteration_chunk_size;
    long n_bad_solve;
} RotamerData;

static void node_holder_init(NodeHolder* h, int n_rot, int n_elem) {
    h->n_rot = n_rot; h->n_elem = n_elem;
    h->prob = (float*)xcalloc((size_t)n_rot * n_elem, sizeof(float));
    h->cur = (float*)xcalloc((size_t)n_rot * n_elem, sizeof(float));
    h->old = (float*)xcalloc((size_t)n_rot * n_elem, sizeof(float));
    h->energy_offset = (float*)xcalloc(n_elem, sizeof(float));
    for (int i = 0; i < n_rot * n_elem; ++i) h->cur[i] = h->old[i] = 1.f;
}
static void edge_holder_init(EdgeHolder* h, NodeHolder* n1, NodeHolder* n2, int max_n_edge) {
    memset(h, 0, sizeof(*h));
    h->n_rot1 = n1->n_rot; h->n_rot2 = n2->n_rot; h->nodes1 = n1; h->nodes2 = n2; h->max_n_edge = max_n_edge;
    int r12 = h->n_rot1 * h->n_rot2, rs = h->n_rot1 + h->n_rot2;
    h->prob = (float*)xcalloc((size_t)(max_n_edge + 3) * r12, sizeof(float));
    h->marginal = (float*)xcalloc((size_t)(max_n_edge + 3) * r12, sizeof(float));
    h->cur = (float*)xcalloc((size_t)(max_n_edge + 3) * rs, sizeof(float));
    h->old = (float*)xcalloc((size_t)(max_n_edge + 3) * rs, sizeof(float));
    h->ei1 = (int*)xcalloc(max_n_edge + 4, sizeof(int)); h->ei2 = (int*)xcalloc(max_n_edge + 4, sizeof(int));
    h->locator = (int*)xcalloc((size_t)n1->n_elem * n2->n_elem + 1, sizeof(int));
    for (int i = 0; i < n1->n_elem * n2->n_elem; ++i) h->locator[i] = -1;
    for (int i = 0; i < (max_n_edge + 3) * r12; ++i) h->prob[i] = 1.f;
    h->cap_edge_loc = 1024; h->edge_loc = (EdgeLoc*)xcalloc(h->cap_edge_loc, sizeof(EdgeLoc));
}
static void edge_holder_reset(EdgeHolder* h) {   /* rotamer.cpp:351-360 */
    int r12 = h->n_rot1 * h->n_rot2;
    for (int idx = 0; idx < h->n_edge; ++idx) {
        for (int j = 0; j < r12; ++j) h->prob[idx * r12 + j] = 1.f;
        h->locator[h->ei1[idx] * h->nodes2->n_elem + h->ei2[idx]] = -1;
    }
    h->n_edge = 0; h->n_edge_loc = 0;
}
static void edge_holder_add(EdgeHolder* h, int ne, float prob_val, int id1, int rot1, int id2, int rot2) {   /* rotamer.cpp:363-376 */
    int* slot = &h->locator[id1 * h->nodes2->n_elem + id2];
    if (*slot < 0) { *slot = h->n_edge++; h->ei1[*slot] = id1; h->ei2[*slot] = id2; }
    int idx = *slot;
    h->prob[idx * h->n_rot1 * h->n_rot2 + rot1 * h->n_rot2 + rot2] *= prob_val;
    if (h->n_edge_loc == h->cap_edge_loc) { h->cap_edge_loc *= 2; h->edge_loc = (EdgeLoc*)realloc(h->edge_loc, h->cap_edge_loc * sizeof(EdgeLoc)); }
    EdgeLoc el = {ne, rot1 * h->n_rot2 + rot2, idx};
    h->edge_loc[h->n_edge_loc++] = el;
}

/* rotamer.cpp:453-522 update_beliefs: sequential over edges, node beliefs multiplied in place and
 * re-normalised after every edge; edge beliefs L1-normalised afterwards (exact reciprocal here, 12-bit
 * approx_rcp in the reference -- only a message scale, which the next normalisation removes). */
static void update_beliefs(EdgeHolder* h) {
    int n1 = h->n_rot1, n2 = h->n_rot2, rs = n1 + n2;
    for (int ne = 0; ne < h->n_edge; ++ne) {
        float* on1 = h->nodes1->old + h->ei1[ne] * n1; float* on2 = h->nodes2->old + h->ei2[ne] * n2;
        float* cn1 = h->nodes1->cur + h->ei1[ne] * n1; float* cn2 = h->nodes2->cur + h->ei2[ne] * n2;
        const float* oe = h->old + ne * rs; float* ce = h->cur + ne * rs;
        const float* P = h->prob + ne * n1 * n2;
        float v1[6], v2[6], e1[6], e2[6];
        for (int i = 0; i < n1; ++i) v1[i] = on1[i] * rcpf(1e-10f + oe[i]);
        for (int j = 0; j < n2; ++j) v2[j] = on2[j] * rcpf(1e-10f + oe[n1 + j]);
        for (int i = 0; i < n1; ++i) { float s = 0.f; for (int j = 0; j < n2; ++j) s += P[i * n2 + j] * v2[j]; e1[i] = s; }
        for (int j = 0; j < n2; ++j) { float s = 0.f; for (int i = 0; i < n1; ++i) s += v1[i] * P[i * n2 + j]; e2[j] = s; }
        float s1 = 0.f, s2 = 0.f, t1[6], t2[6];
        for (int i = 0; i < n1; ++i) { t1[i] = e1[i] * cn1[i]; s1 += t1[i]; }
        for (int j = 0; j < n2; ++j) { t2[j] = e2[j] * cn2[j]; s2 += t2[j]; }
        float r1 = rcpf(s1), r2 = rcpf(s2);
        for (int i = 0; i < n1; ++i) { ce[i] = e1[i]; }
        for (int j = 0; j < n2; ++j) { ce[n1 + j] = e2[j]; }
        /* store node beliefs last: for a 33/66 edge both nodes live in the same holder but are distinct */
        for (int i = 0; i < n1; ++i) cn1[i] = t1[i] * r1;
        for (int j = 0; j < n2; ++j) cn2[j] = t2[j] * r2;
    }
    for (int ne = 0; ne < h->n_edge; ++ne) {
        float* ce = h->cur + ne * rs; float s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < n1; ++i) s1 += ce[i];
        for (int j = 0; j < n2; ++j) s2 += ce[n1 + j];
        float r1 = rcpf(s1), r2 = rcpf(s2);
        for (int i = 0; i < n1; ++i) ce[i] *= r1;
        for (int j = 0; j < n2; ++j) ce[n1 + j] *= r2;
    }
}

static void standardize_belief_update(NodeHolder* h, float damping) {   /* rotamer.cpp:258-273 */
    for (int ne = 0; ne < h->n_elem; ++ne) {
        float* b = h->cur + ne * h->n_rot; const float* o = h->old + ne * h->n_rot;
        float m = b[0]; for (int i = 1; i < h->n_rot; ++i) m = maxf(b[i], m);
        float r = rcpf(m);
        if (damping != 0.f) for (int i = 0; i < h->n_rot; ++i) b[i] = (1.f - damping) * r * b[i] + damping * o[i];
        else for (int i = 0; i < h->n_rot; ++i) b[i] = r * b[i];
    }
}
static float max_deviation(NodeHolder* h) {   /* rotamer.cpp:275-281 (signed, no abs) */
    float dev = 0.f;
    for (int i = 0; i < h->n_rot * h->n_elem; ++i) dev = maxf(h->cur[i] - h->old[i], dev);
    return dev;
}
static void swap_ptr(float** a, float** b) { float* t = *a; *a = *b; *b = t; }

static void calculate_new_beliefs(RotamerData* d, float damping, int do_swap_for_initial) {   /* rotamer.cpp:988-1002 */
    memcpy(d->nodes[3].cur, d->nodes[3].prob, sizeof(float) * 3 * d->nodes[3].n_elem);
    memcpy(d->nodes[6].cur, d->nodes[6].prob, sizeof(float) * 6 * d->nodes[6].n_elem);
    update_beliefs(&d->edges[3][3]); update_beliefs(&d->edges[3][6]); update_beliefs(&d->edges[6][6]);
    if (do_swap_for_initial) { swap_ptr(&d->nodes[3].cur, &d->nodes[3].old); swap_ptr(&d->nodes[6].cur, &d->nodes[6].old); }
    standardize_belief_update(&d->nodes[3], damping); standardize_belief_update(&d->nodes[6], damping);
}

static void node_calculate_marginals(NodeHolder* h) {   /* rotamer.cpp:283-290 */
    for (int nn = 0; nn < h->n_elem; ++nn) {
        float* b = h->cur + nn * h->n_rot; float s = 0.f;
        for (int i = 0; i < h->n_rot; ++i) s += b[i];
        float r = rcpf(s);
        for (int i = 0; i < h->n_rot; ++i) b[i] *= r;
    }
}
static void edge_calculate_marginals(EdgeHolder* h) {   /* rotamer.cpp:405-429 */
    int n1 = h->n_rot1, n2 = h->n_rot2, rs = n1 + n2;
    for (int ne = 0; ne < h->n_edge; ++ne) {
        const float* b1 = h->nodes1->cur + h->ei1[ne] * n1; const float* b2 = h->nodes2->cur + h->ei2[ne] * n2;
        const float* b = h->cur + ne * rs; const float* p = h->prob + ne * n1 * n2; float* marg = h->marginal + ne * n1 * n2;
        float bc1[6], bc2[6], s = 0.f;
        for (int i = 0; i < n1; ++i) bc1[i] = b1[i] * rcpf(1e-10f + b[i]);
        for (int j = 0; j < n2; ++j) bc2[j] = b2[j] * rcpf(1e-10f + b[n1 + j]);
        for (int i = 0; i < n1; ++i) for (int j = 0; j < n2; ++j) { marg[i * n2 + j] = p[i * n2 + j] * bc1[i] * bc2[j]; s += marg[i * n2 + j]; }
        float r = rcpf(s);
        for (int i = 0; i < n1 * n2; ++i) marg[i] *= r;
    }
}
static float node_free_energy(NodeHolder* h, int nn) {   /* rotamer.cpp:292-302 */
    float b[6], s = 0.f;
    for (int i = 0; i < h->n_rot; ++i) { b[i] = h->cur[nn * h->n_rot + i]; s += b[i]; }
    float r = rcpf(s);
    float en = h->energy_offset[nn];
    for (int i = 0; i < h->n_rot; ++i) { b[i] *= r; en += b[i] * logf((1e-10f + b[i]) * rcpf(1e-10f + h->prob[nn * h->n_rot + i])); }
    return en;
}
static float edge_free_energy(EdgeHolder* h, int ne) {   /* rotamer.cpp:431-451 */
    int n1 = h->n_rot1, n2 = h->n_rot2;
    const float* b1 = h->nodes1->cur + h->ei1[ne] * n1; const float* b2 = h->nodes2->cur + h->ei2[ne] * n2;
    const float* p = h->marginal + ne * n1 * n2; const float* pr = h->prob + ne * n1 * n2;
    float en = 0.f;
    for (int i = 0; i < n1; ++i) for (int j = 0; j < n2; ++j) {
        int k = i * n2 + j;
        en += p[k] * logf((1e-10f + p[k]) * rcpf(1e-10f + pr[k] * b1[i] * b2[j]));
    }
    return en;
}

static void rotamer_fill_holders(Engine* e, Node* n) {   /* rotamer.cpp:793-852 */
    RotamerData* d = (RotamerData*)n->data;
    static const int rots[3] = {1, 3, 6};
    for (int a = 0; a < 3; ++a) for (int b = a; b < 3; ++b) edge_holder_reset(&d->edges[rots[a]][rots[b]]);
    for (int a = 0; a < 3; ++a) memset(d->nodes[rots[a]].prob, 0, sizeof(float) * rots[a] * d->nodes[rots[a]].n_elem);
    const unsigned selector = (1u << N_BIT_ROTAMER) - 1u;
    for (int i = 0; i < d->g.n_elem1; ++i) {
        unsigned id = (unsigned)d->g.id1[i];
        unsigned rot = id & selector; id >>= N_BIT_ROTAMER;
        unsigned n_rot = id & selector; id >>= N_BIT_ROTAMER;
        int index = d->g.loc1[i];
        float energy = 0.f;
        for (int k = 0; k < d->n_prob_nodes; ++k) energy += VA(parent(e, n, 1 + k)->output, 0, index);
        d->nodes[n_rot].prob[id * n_rot + rot] += energy;
    }
    for (int a = 0; a < 3; ++a) {   /* convert_energy_to_prob, rotamer.cpp:239-256 */
        NodeHolder* h = &d->nodes[rots[a]];
        for (int ne = 0; ne < h->n_elem; ++ne) {
            float* p = h->prob + ne * h->n_rot; float off = p[0];
            for (int k = 1; k < h->n_rot; ++k) off = minf(off, p[k]);
            for (int k = 0; k < h->n_rot; ++k) p[k] = expf(off - p[k]);
            h->energy_offset[ne] = off;
        }
    }
    igraph_compute_edges(&d->g, parent(e, n, 0), NULL);
    for (int ne = 0; ne < d->g.n_edge; ++ne) {
        int id1 = d->g.id1[d->g.edge_i1[ne]], id2 = d->g.id2[d->g.edge_i2[ne]];
        float prob = expf(-d->g.edge_value[ne]);
        if ((id1 & (selector << N_BIT_ROTAMER)) > (id2 & (selector << N_BIT_ROTAMER))) { int t = id1; id1 = id2; id2 = t; }
        unsigned rot1 = id1 & selector; id1 >>= N_BIT_ROTAMER;
        unsigned rot2 = id2 & selector; id2 >>= N_BIT_ROTAMER;
        unsigned n_rot1 = id1 & selector; id1 >>= N_BIT_ROTAMER;
        unsigned n_rot2 = id2 & selector; id2 >>= N_BIT_ROTAMER;
        edge_holder_add(&d->edges[n_rot1][n_rot2], ne, prob, id1, rot1, id2, rot2);
    }
    for (int b = 1; b < 3; ++b) {   /* move_edge_prob_to_node2, rotamer.cpp:378-385 */
        EdgeHolder* h = &d->edges[1][rots[b]];
        for (int ne = 0; ne < h->n_edge; ++ne) for (int k = 0; k < h->n_rot2; ++k) h->nodes2->prob[h->ei2[ne] * h->n_rot2 + k] *= h->prob[ne * h->n_rot2 + k];
    }
}

static int rotamer_solve(RotamerData* d, float* final_dev) {   /* rotamer.cpp:1005-1061 */
    static const int rots[3] = {1, 3, 6};
    for (int a = 0; a < 3; ++a) memcpy(d->nodes[rots[a]].old, d->nodes[rots[a]].prob, sizeof(float) * rots[a] * d->nodes[rots[a]].n_elem);
    EdgeHolder* eh[3] = {&d->edges[3][3], &d->edges[3][6], &d->edges[6][6]};
    for (int k = 0; k < 3; ++k) for (int i = 0; i < eh[k]->n_edge * (eh[k]->n_rot1 + eh[k]->n_rot2); ++i) eh[k]->old[i] = 1.f;
    calculate_new_beliefs(d, 0.f, 1);
    float maxdev = 1e10f; int iter = 0;
    for (; maxdev > d->tol && iter < d->max_iter; iter += d->iteration_chunk_size) {
        for (int j = 0; j < d->iteration_chunk_size; ++j) {
            swap_ptr(&d->nodes[3].cur, &d->nodes[3].old); swap_ptr(&d->nodes[6].cur, &d->nodes[6].old);
            for (int k = 0; k < 3; ++k) swap_ptr(&eh[k]->cur, &eh[k]->old);
            calculate_new_beliefs(d, d->damping, 0);
        }
        maxdev = maxf(max_deviation(&d->nodes[3]), max_deviation(&d->nodes[6]));
    }
    node_calculate_marginals(&d->nodes[1]); node_calculate_marginals(&d->nodes[3]); node_calculate_marginals(&d->nodes[6]);
    /* edges11 marginal is trivially 1 (rotamer.cpp:1056) */
    for (int ne = 0; ne < d->edges[1][1].n_edge; ++ne) d->edges[1][1].marginal[ne] = 1.f;
    for (int k = 0; k < 3; ++k) edge_calculate_marginals(eh[k]);
    *final_dev = maxdev;
    return iter;
}

static void rotamer_propagate(Engine* e, Node* n) {   /* rotamer.cpp:956-985 */
    RotamerData* d = (RotamerData*)n->data;
    EdgeHolder* h;
    h = &d->edges[1][1]; for (int k = 0; k < h->n_edge_loc; ++k) d->g.edge_sens[h->edge_loc[k].edge_num] = 1.f;
    h = &d->edges[1][3]; for (int k = 0; k < h->n_edge_loc; ++k) d->g.edge_sens[h->edge_loc[k].edge_num] = d->nodes[3].cur[h->ei2[h->edge_loc[k].ne] * 3 + h->edge_loc[k].dim];
    h = &d->edges[1][6]; for (int k = 0; k < h->n_edge_loc; ++k) d->g.edge_sens[h->edge_loc[k].edge_num] = d->nodes[6].cur[h->ei2[h->edge_loc[k].ne] * 6 + h->edge_loc[k].dim];
    EdgeHolder* eh[3] = {&d->edges[3][3], &d->edges[3][6], &d->edges[6][6]};
    for (int s = 0; s < 3; ++s) { h = eh[s];
        for (int k = 0; k < h->n_edge_loc; ++k) d->g.edge_sens[h->edge_loc[k].edge_num] = h->marginal[h->edge_loc[k].ne * h->n_rot1 * h->n_rot2 + h->edge_loc[k].dim]; }
    igraph_propagate(&d->g, parent(e, n, 0), NULL);
    const unsigned selector = (1u << N_BIT_ROTAMER) - 1u;
    for (int i = 0; i < d->g.n_elem1; ++i) {
        unsigned id = (unsigned)d->g.id1[i];
        unsigned rot = id & selector; id >>= N_BIT_ROTAMER;
        unsigned n_rot = id & selector; id >>= N_BIT_ROTAMER;
        for (int k = 0; k < d->n_prob_nodes; ++k) VA(parent(e, n, 1 + k)->sens, 0, d->g.loc1[i]) += d->nodes[n_rot].cur[id * n_rot + rot];
    }
}

static float rotamer_energy(RotamerData* d) {   /* rotamer.cpp:854-866 */
    float en = 0.f;
    for (int nn = 0; nn < d->nodes[1].n_elem; ++nn) en += node_free_energy(&d->nodes[1], nn);
    for (int nn = 0; nn < d->nodes[3].n_elem; ++nn) en += node_free_energy(&d->nodes[3], nn);
    for (int nn = 0; nn < d->nodes[6].n_elem; ++nn) en += node_free_energy(&d->nodes[6], nn);
    for (int ne = 0; ne < d->edges[1][1].n_edge; ++ne) en += -logf(d->edges[1][1].prob[ne]);
    for (int ne = 0; ne < d->edges[3][3].n_edge; ++ne) en += edge_free_energy(&d->edges[3][3], ne);
    for (int ne = 0; ne < d->edges[3][6].n_edge; ++ne) en += edge_free_energy(&d->edges[3][6], ne);
    for (int ne = 0; ne < d->edges[6][6].n_edge; ++ne) en += edge_free_energy(&d->edges[6][6], ne);
    return en;
}

static void rotamer_value(Engine* e, Node* n, int mode) {   /* rotamer.cpp:779-789 */
    RotamerData* d = (RotamerData*)n->data;
    rotamer_fill_holders(e, n);
    float dev; int iter = rotamer_solve(d, &dev);
    e->rotamer_iterations = iter;
    if (iter >= d->max_iter - d->iteration_chunk_size - 1) d->n_bad_solve++;
    rotamer_propagate(e, n);
    if (mode == PotentialAndDerivMode) n->potential = rotamer_energy(d);
}

/* rotamer.cpp:928-954 */
static void rotamer_arrange(RotamerData* d, const float* e1, const float* e3, const float* e6, float* out) {
    const unsigned selector = (1u << N_BIT_ROTAMER) - 1u;
    int k = 0;
    for (int i = 0; i < d->g.n_elem1; ++i) {
        unsigned id = (unsigned)d->g.id1[i];
        if (id & selector) continue;
        int seen = 0; for (int j = 0; j < i; ++j) if (d->g.id1[j] == d->g.id1[i]) { seen = 1; break; }
        if (seen) continue;
        id >>= N_BIT_ROTAMER; unsigned n_rot = id & selector; id >>= N_BIT_ROTAMER;
        out[k++] = n_rot == 1 ? e1[id] : (n_rot == 3 ? e3[id] : e6[id]);
    }
}

static int rotamer_get_value(Engine* e, Node* n, const char* log_name, int n_output, float* out) {   /* rotamer.cpp:675-773 */
    RotamerData* d = (RotamerData*)n->data;
    int n1 = d->nodes[1].n_elem, n3 = d->nodes[3].n_elem, n6 = d->nodes[6].n_elem, n_node = n1 + n3 + n6;
    static const int rots[3] = {1, 3, 6};
    if (!strcmp(log_name, "n_node")) { if (n_output != 1) return 1; out[0] = (float)n_node; return 0; }
    if (!strcmp(log_name, "count_edges_by_type")) { if (n_output != d->g.n_type1 * d->g.n_type2) return 1; igraph_count_edges_by_type(&d->g, out); return 0; }
    if (!strcmp(log_name, "node_energy")) {
        if (n_output != n_node * 6) return 1;
        int nn = 0;
        for (int a = 0; a < 3; ++a) { NodeHolder* h = &d->nodes[rots[a]];
            for (int ne = 0; ne < h->n_elem; ++ne, ++nn) for (int nr = 0; nr < 6; ++nr) out[nn * 6 + nr] = nr < h->n_rot ? -logf(h->prob[ne * h->n_rot + nr]) : 1e5f; }
        return 0;
    }
    if (!strcmp(log_name, "rotamer_free_energy") || !strcmp(log_name, "rotamer_1body_energy")) {
        int is_free = !strcmp(log_name, "rotamer_free_energy");
        int npn = is_free ? 1 : d->n_prob_nodes;
        if (n_output != n_node * npn) return 1;
        float* e1 = (float*)xcalloc(n1 + 1, sizeof(float)); float* e3 = (float*)xcalloc(n3 + 1, sizeof(float)); float* e6 = (float*)xcalloc(n6 + 1, sizeof(float));
        float* tmp = (float*)xcalloc(n_node, sizeof(float));
        for (int ip = 0; ip < npn; ++ip) {
            memset(e1, 0, sizeof(float) * n1); memset(e3, 0, sizeof(float) * n3); memset(e6, 0, sizeof(float) * n6);
            float* ee[7] = {0, e1, 0, e3, 0, 0, e6};
            if (is_free) {   /* rotamer.cpp:868-902 */
                for (int a = 0; a < 3; ++a) for (int nn = 0; nn < d->nodes[rots[a]].n_elem; ++nn) ee[rots[a]][nn] += node_free_energy(&d->nodes[rots[a]], nn);
                EdgeHolder* h = &d->edges[1][1];
                for (int ne = 0; ne < h->n_edge; ++ne) { float en = -logf(h->prob[ne]); e1[h->ei1[ne]] += 0.5 * en; e1[h->ei2[ne]] += 0.5 * en; }
                EdgeHolder* eh[3] = {&d->edges[3][3], &d->edges[3][6], &d->edges[6][6]};
                for (int s = 0; s < 3; ++s) { h = eh[s];
                    for (int ne = 0; ne < h->n_edge; ++ne) { float en = edge_free_energy(h, ne); ee[h->n_rot1][h->ei1[ne]] += 0.5 * en; ee[h->n_rot2][h->ei2[ne]] += 0.5 * en; } }
            } else {         /* rotamer.cpp:904-926 */
                const unsigned selector = (1u << N_BIT_ROTAMER) - 1u;
                Node* pn = parent(e, n, 1 + ip);
                for (int i = 0; i < d->g.n_elem1; ++i) {
                    unsigned id = (unsigned)d->g.id1[i];
                    unsigned rot = id & selector; id >>= N_BIT_ROTAMER; unsigned n_rot = id & selector; id >>= N_BIT_ROTAMER;
                    ee[n_rot][id] += d->nodes[n_rot].cur[id * n_rot + rot] * VA(pn->output, 0, d->g.loc1[i]);
                }
            }
            rotamer_arrange(d, e1, e3, e6, tmp);
            for (int i = 0; i < n_node; ++i) out[i * npn + ip] = tmp[i];
        }
        free(e1); free(e3); free(e6); free(tmp);
        return 0;
    }
    if (!strcmp(log_name, "edge_marginal_in_graph_order") || !strcmp(log_name, "edge_energy")) {   /* rotamer.cpp:712-763 */
        int do_marginal = !strcmp(log_name, "edge_marginal_in_graph_order");
        if ((long)n_output != (long)n_node * n_node * 36) return 1;
        memset(out, 0, sizeof(float) * n_output);
        int starts[7] = {0, 0, 0, n1, 0, 0, n1 + n3};
        if (do_marginal) {
            float* nm = (float*)xcalloc((size_t)n_node * 6, sizeof(float));
            int nn = 0;
            for (int a = 0; a < 3; ++a) { NodeHolder* h = &d->nodes[rots[a]];
                for (int ne = 0; ne < h->n_elem; ++ne, ++nn) for (int nr = 0; nr < h->n_rot; ++nr) nm[nn * 6 + nr] = h->cur[ne * h->n_rot + nr]; }
            for (int i1 = 0; i1 < n_node; ++i1) for (int i2 = 0; i2 < n_node; ++i2) for (int r1 = 0; r1 < 6; ++r1) for (int r2 = 0; r2 < 6; ++r2)
                out[(((size_t)i1 * n_node + i2) * 6 + r1) * 6 + r2] = (i1 == i2) ? nm[i1 * 6 + r1] * (r1 == r2) : nm[i1 * 6 + r1] * nm[i2 * 6 + r2];
            free(nm);
        }
        EdgeHolder* eh[4] = {&d->edges[1][1], &d->edges[3][3], &d->edges[3][6], &d->edges[6][6]};
        for (int s = 0; s < 4; ++s) { EdgeHolder* h = eh[s];
            for (int ne = 0; ne < h->n_edge; ++ne) {
                int i1 = starts[h->n_rot1] + h->ei1[ne], i2 = starts[h->n_rot2] + h->ei2[ne];
                for (int r1 = 0; r1 < h->n_rot1; ++r1) for (int r2 = 0; r2 < h->n_rot2; ++r2) {
                    float v = do_marginal ? h->marginal[ne * h->n_rot1 * h->n_rot2 + r1 * h->n_rot2 + r2] : -logf(h->prob[ne * h->n_rot1 * h->n_rot2 + r1 * h->n_rot2 + r2]);
                    out[(((size_t)i1 * n_node + i2) * 6 + r1) * 6 + r2] = v;
                    out[(((size_t)i2 * n_node + i1) * 6 + r2) * 6 + r1] = v;
                } } }
        return 0;
    }
    if (!strcmp(log_name, "read n_bad_solve")) { if (n_output != 1) return 1; out[0] = (float)d->n_bad_solve; return 0; }
    fprintf(stderr, "ERROR: Value %s not implemented\n", log_name);
    return 1;
}

/* ============================================================================================
 * node construction (registry by name prefix: src/deriv_engine.cpp:231-241)
 * ========================================================================================== */
static int is_prefix(const char* a, const char* b) { return strncmp(a, b, strlen(a)) == 0; }

static void coord_node(Node* n, int n_elem, int width) {
    n->potential_term = 0; n->n_elem = n_elem; n->elem_width = width;
    n->output = va_alloc(width, round_up(n_elem, 4)); n->sens = va_alloc(width, round_up(n_elem, 4));
}

static int build_node(Engine* e, Node* n, hid_t grp, const char* name) {
    hsize_t dims[4];
    n->potential_term = 1; n->n_elem = 1; n->elem_width = 1;
    if (is_prefix("dist_spring", name) || is_prefix("angle_spring", name) || is_prefix("dihedral_spring", name)) {
        int w = is_prefix("dist_spring", name) ? 2 : (is_prefix("angle_spring", name) ? 3 : 4);
        SpringData* d = (SpringData*)xcalloc(1, sizeof(*d));
        d->id = h5_read_i(grp, "id", 2, dims); if (!d->id || (int)dims[1] != w) return -1; d->n = (int)dims[0];
        d->equil = h5_read_f(grp, "equil_dist", 1, dims); d->k = h5_read_f(grp, "spring_const", 1, dims);
        if (!d->equil || !d->k) return -1;
        n->data = d; n->compute_value = w == 2 ? dist_spring_value : (w == 3 ? angle_spring_value : dihedral_spring_value);
        return 0;
    }
    if (is_prefix("atom_pos_spring", name) || is_prefix("tension", name) || is_prefix("AFM", name) || is_prefix("z_flat_bottom", name)) {
        PointData* d = (PointData*)xcalloc(1, sizeof(*d));
        d->kind = is_prefix("atom_pos_spring", name) ? 0 : is_prefix("tension", name) ? 1 : is_prefix("AFM", name) ? 2 : 3;
        d->id = h5_read_i(grp, d->kind == 0 ? "id" : "atom", 1, dims); if (!d->id) return -1; d->n = (int)dims[0];
        if (d->kind == 0) { d->x0 = h5_read_f(grp, "x0", 2, dims); d->k = h5_read_f(grp, "spring_const", 1, dims); if (!d->x0 || !d->k) return -1; }
        else if (d->kind == 1) { d->v3 = h5_read_f(grp, "tension_coeff", 2, dims); if (!d->v3) return -1; }
        else if (d->kind == 2) {
            d->k = h5_read_f(grp, "spring_const", 1, dims); d->x0 = h5_read_f(grp, "starting_tip_pos", 2, dims); d->v3 = h5_read_f(grp, "pulling_vel", 2, dims);
            if (!d->k || !d->x0 || !d->v3) return -1;
            d->time_initial = h5_attr_f(grp, "pulling_vel", "time_initial"); d->time_step = h5_attr_f(grp, "pulling_vel", "time_step");
        } else {
            d->a = h5_read_f(grp, "z0", 1, dims); d->b = h5_read_f(grp, "radius", 1, dims); d->k = h5_read_f(grp, "spring_constant", 1, dims);
            if (!d->a || !d->b || !d->k) return -1;
        }
        n->data = d; n->compute_value = point_potential_value; return 0;
    }
    if (is_prefix("radial", name) || is_prefix("hbond_sc_radial", name)) {
        IGraph* g = (IGraph*)xcalloc(1, sizeof(*g));
        if (igraph_init(g, grp, is_prefix("radial", name) ? IT_RADIAL : IT_HBOND_SC_RADIAL)) return -1;
        n->data = g; n->compute_value = radial_pairs_value; return 0;
    }
    if (is_prefix("contact", name)) {
        ContactData* d = (ContactData*)xcalloc(1, sizeof(*d));
        d->id = h5_read_i(grp, "id", 2, dims); if (!d->id || dims[1] != 2) return -1; d->n = (int)dims[0];
        d->energy = h5_read_f(grp, "energy", 1, dims); d->dist = h5_read_f(grp, "distance", 1, dims); d->scale = h5_read_f(grp, "width", 1, dims);
        if (!d->energy || !d->dist || !d->scale) return -1;
        d->cutoff = (float*)xcalloc(d->n, sizeof(float));
        for (int i = 0; i < d->n; ++i) { d->scale[i] = 1.f / d->scale[i]; d->cutoff[i] = d->dist[i] + 1.f / d->scale[i]; }   /* sidechain_radial.cpp:170-171 */
        n->data = d; n->compute_value = contact_value; return 0;
    }
    if (is_prefix("constant", name)) {
        ConstantData* d = (ConstantData*)xcalloc(1, sizeof(*d));
        d->value = h5_read_f(grp, "value", 2, dims); if (!d->value) return -1;
        coord_node(n, (int)dims[0], (int)dims[1]); n->data = d; n->compute_value = constant_value; n->propagate_deriv = no_deriv; return 0;
    }
    if (is_prefix("slice", name)) {
        SliceData* d = (SliceData*)xcalloc(1, sizeof(*d));
        d->id = h5_read_i(grp, "id", 1, dims); if (!d->id) return -1;
        coord_node(n, (int)dims[0], e->nodes[n->parents[0]].elem_width); n->data = d; n->compute_value = slice_value; n->propagate_deriv = slice_deriv; return 0;
    }
    if (is_prefix("uniform_transform", name)) {
        UniformTransformData* d = (UniformTransformData*)xcalloc(1, sizeof(*d));
        d->coeff = h5_read_f(grp, "bspline_coeff", 1, dims); if (!d->coeff) return -1; d->n_coeff = (int)dims[0];
        d->offset = h5_attr_f(grp, "bspline_coeff", "spline_offset"); d->inv_dx = h5_attr_f(grp, "bspline_coeff", "spline_inv_dx");
        int ne = e->nodes[n->parents[0]].n_elem;
        d->jac = (float*)xcalloc(ne, sizeof(float));
        coord_node(n, ne, 1); n->data = d; n->compute_value = uniform_transform_value; n->propagate_deriv = uniform_transform_deriv; return 0;
    }
    if (is_prefix("linear_coupling_uniform", name) || is_prefix("linear_coupling_with_inactivation", name)) {
        LinearCouplingData* d = (LinearCouplingData*)xcalloc(1, sizeof(*d));
        d->has_inact = is_prefix("linear_coupling_with_inactivation", name);
        if (d->has_inact) d->inact_dim = h5_attr_i(grp, ".", "inactivation_dim", 0);
        d->couplings = h5_read_f(grp, "couplings", 1, dims); if (!d->couplings) return -1; d->n_coupling = (int)dims[0];
        d->types = h5_read_i(grp, "coupling_types", 1, dims); if (!d->types) return -1;
        n->data = d; n->compute_value = linear_coupling_value; return 0;
    }
    if (is_prefix("membrane_potential", name)) {
        MembraneData* d = (MembraneData*)xcalloc(1, sizeof(*d));
        d->cb_index = h5_read_i(grp, "cb_index", 1, dims); if (!d->cb_index) return -1; d->n_elem = (int)dims[0];
        d->env_index = h5_read_i(grp, "env_index", 1, dims); d->restype = h5_read_i(grp, "residue_type", 1, dims);
        d->cov_midpoint = h5_read_f(grp, "cov_midpoint", 1, dims); d->cov_sharpness = h5_read_f(grp, "cov_sharpness", 1, dims);
        if (!d->env_index || !d->restype || !d->cov_midpoint || !d->cov_sharpness) return -1;
        if (h5_dims(grp, "donor_residue_ids", 1, dims)) return -1;
        d->n_donor = (int)dims[0];
        if (h5_dims(grp, "acceptor_residue_ids", 1, dims)) return -1;
        d->n_acceptor = (int)dims[0];
        double* cbe = h5_read_d(grp, "cb_energy", 2, dims); if (!cbe) return -1;
        d->cb.n_layer = (int)dims[0]; d->cb.nx = (int)dims[1]; clamped1d_fit(&d->cb, cbe); free(cbe);
        double* uhe = h5_read_d(grp, "uhb_energy", 2, dims); if (!uhe || dims[0] != 2) return -1;
        d->uhb.n_layer = 2; d->uhb.nx = (int)dims[1]; clamped1d_fit(&d->uhb, uhe); free(uhe);
        d->cb_z_shift = -h5_attr_f(grp, "cb_energy", "z_min");
        d->cb_z_scale = (d->cb.nx - 1) / (h5_attr_f(grp, "cb_energy", "z_max") + d->cb_z_shift);
        d->uhb_z_shift = -h5_attr_f(grp, "uhb_energy", "z_min");
        d->uhb_z_scale = (d->uhb.nx - 1) / (h5_attr_f(grp, "uhb_energy", "z_max") + d->uhb_z_shift);
        n->data = d; n->compute_value = membrane_value; return 0;
    }
    if (is_prefix("cavity_radial", name)) {
        CavityData* d = (CavityData*)xcalloc(1, sizeof(*d));
        d->id = h5_read_i(grp, "id", 1, dims); if (!d->id) return -1; d->n = (int)dims[0];
        d->radius = h5_read_f(grp, "radius", 1, dims); d->k = h5_read_f(grp, "spring_constant", 1, dims);
        n->data = d; n->compute_value = cavity_radial_value; return 0;
    }
    if (is_prefix("rama_coord", name)) {
        RamaCoordData* d = (RamaCoordData*)xcalloc(1, sizeof(*d));
        d->atom = h5_read_i(grp, "id", 2, dims); if (!d->atom || dims[1] != 5) return -1; d->n = (int)dims[0];
        d->dummy = (int*)xcalloc(d->n * 2, sizeof(int)); d->jac = (f3*)xcalloc((size_t)d->n * 10, sizeof(f3));
        for (int i = 0; i < d->n; ++i) {
            d->dummy[i * 2] = d->atom[i * 5] == -1; d->dummy[i * 2 + 1] = d->atom[i * 5 + 4] == -1;
            if (d->dummy[i * 2]) d->atom[i * 5] = 0;
            if (d->dummy[i * 2 + 1]) d->atom[i * 5 + 4] = 0;
        }
        coord_node(n, d->n, 2); n->data = d; n->compute_value = rama_coord_value; n->propagate_deriv = rama_coord_deriv; return 0;
    }
    if (is_prefix("affine_alignment", name)) {
        AffineData* d = (AffineData*)xcalloc(1, sizeof(*d));
        d->atoms = h5_read_i(grp, "atoms", 2, dims); if (!d->atoms || dims[1] != 3) return -1; d->n_res = (int)dims[0];
        d->ref_geom = h5_read_f(grp, "ref_geom", 3, dims); if (!d->ref_geom) return -1;
        d->n_group = round_up(d->n_res, 4) / 4;
        d->evals = (S4*)xcalloc((size_t)d->n_group * 4, sizeof(S4)); d->evecs = (S4*)xcalloc((size_t)d->n_group * 16, sizeof(S4));
        coord_node(n, d->n_res, 7); n->data = d; n->compute_value = affine_alignment_value; n->propagate_deriv = affine_alignment_deriv; return 0;
    }
    if (is_prefix("infer_H_O", name)) {
        InferData* d = (InferData*)xcalloc(1, sizeof(*d));
        hsize_t dd[2], da[2];
        int* don = h5_read_i(grp, "donors/id", 2, dd); int* acc = h5_read_i(grp, "acceptors/id", 2, da);
        float* bl_d = h5_read_f(grp, "donors/bond_length", 1, dims); float* bl_a = h5_read_f(grp, "acceptors/bond_length", 1, dims);
        if (!don || !acc || !bl_d || !bl_a) return -1;
        int nd = (int)dd[0], na = (int)da[0];
        d->n_virtual = nd + na;
        d->atom = (int*)xcalloc((size_t)d->n_virtual * 3, sizeof(int)); d->bond_length = (float*)xcalloc(d->n_virtual, sizeof(float));
        memcpy(d->atom, don, sizeof(int) * nd * 3); memcpy(d->atom + nd * 3, acc, sizeof(int) * na * 3);
        memcpy(d->bond_length, bl_d, sizeof(float) * nd); memcpy(d->bond_length + nd, bl_a, sizeof(float) * na);
        d->dfd = (float*)xcalloc((size_t)d->n_virtual * 12, sizeof(float));
        coord_node(n, d->n_virtual, 6); n->data = d; n->compute_value = infer_value; n->propagate_deriv = infer_deriv; return 0;
    }
    if (is_prefix("protein_hbond", name)) {
        ProteinHBondData* d = (ProteinHBondData*)xcalloc(1, sizeof(*d));
        if (igraph_init(&d->g, grp, IT_PROTEIN_HBOND)) return -1;
        d->n_donor = d->g.n_elem1; d->n_acceptor = d->g.n_elem2;
        d->sens_scaled = (float*)xcalloc(d->n_donor + d->n_acceptor + 4, sizeof(float));
        coord_node(n, d->n_donor + d->n_acceptor, 7); n->data = d; n->compute_value = protein_hbond_value; n->propagate_deriv = protein_hbond_deriv; return 0;
    }
    if (is_prefix("hbond_coverage", name)) {
        IGraph* g = (IGraph*)xcalloc(1, sizeof(*g));
        if (igraph_init(g, grp, IT_HBOND_COVERAGE)) return -1;
        coord_node(n, g->n_elem2, 1); n->data = g; n->compute_value = hbond_coverage_value; n->propagate_deriv = hbond_coverage_deriv; return 0;
    }
    if (is_prefix("hbond_energy", name)) {
        HBondEnergyData* d = (HBondEnergyData*)xcalloc(1, sizeof(*d));
        d->E_protein = h5_attr_f(grp, ".", "protein_hbond_energy");
        n->data = d; n->compute_value = hbond_energy_value; return 0;
    }
    if (is_prefix("environment_coverage", name)) {
        IGraph* g = (IGraph*)xcalloc(1, sizeof(*g));
        if (igraph_init(g, grp, IT_ENVIRONMENT)) return -1;
        coord_node(n, g->n_elem1, 1); n->data = g; n->compute_value = environment_coverage_value; n->propagate_deriv = environment_coverage_deriv; return 0;
    }
    if (is_prefix("weighted_pos", name)) {
        WeightedPosData* d = (WeightedPosData*)xcalloc(1, sizeof(*d));
        d->index_pos = h5_read_i(grp, "index_pos", 1, dims); d->index_weight = h5_read_i(grp, "index_weight", 1, dims);
        if (!d->index_pos || !d->index_weight) return -1;
        coord_node(n, (int)dims[0], 4); n->data = d; n->compute_value = weighted_pos_value; n->propagate_deriv = weighted_pos_deriv; return 0;
    }
    if (is_prefix("nonlinear_coupling", name)) {
        NonlinearData* d = (NonlinearData*)xcalloc(1, sizeof(*d));
        d->coeff = h5_read_f(grp, "coeff", 2, dims); if (!d->coeff) return -1;
        d->n_restype = (int)dims[0]; d->n_coeff = (int)dims[1];
        d->offset = h5_attr_f(grp, "coeff", "spline_offset"); d->inv_dx = h5_attr_f(grp, "coeff", "spline_inv_dx");
        d->types = h5_read_i(grp, "coupling_types", 1, dims); if (!d->types) return -1;
        n->data = d; n->compute_value = nonlinear_coupling_value; return 0;
    }
    if (is_prefix("rama_map_pot", name)) {
        RamaMapData* d = (RamaMapData*)xcalloc(1, sizeof(*d));
        d->residue = h5_read_i(grp, "residue_id", 1, dims); if (!d->residue) return -1; d->n_residue = (int)dims[0];
        d->map_id = h5_read_i(grp, "rama_map_id", 1, dims);
        double* raw = h5_read_d(grp, "rama_pot", 3, dims); if (!raw || !d->map_id) return -1;
        d->s.n_layer = (int)dims[0]; d->s.nx = (int)dims[1]; d->s.ny = (int)dims[2]; d->s.ndim = 1;
        if (d->s.nx != d->s.ny) return -1;
        spline2d_fit(&d->s, raw); free(raw);
        n->data = d; n->compute_value = rama_map_pot_value; return 0;
    }
    if (is_prefix("placement_", name)) {
        PlacementData* d = (PlacementData*)xcalloc(1, sizeof(*d));
        /* registry of placement.cpp:319-325 (the four variants the README force field uses) */
        if (is_prefix("placement_fixed_point_vector_only", name)) { d->n_sig = 2; d->sig[0] = PL_POINT; d->sig[1] = PL_VECTOR; d->n_pos_dim = 6; }
        else if (is_prefix("placement_fixed_point_vector_scalar", name)) { d->n_sig = 3; d->sig[0] = PL_POINT; d->sig[1] = PL_VECTOR; d->sig[2] = PL_SCALAR; d->n_pos_dim = 7; }
        else if (is_prefix("placement_fixed_point_only", name)) { d->n_sig = 1; d->sig[0] = PL_POINT; d->n_pos_dim = 3; }
        else if (is_prefix("placement_fixed_scalar", name)) { d->n_sig = 1; d->sig[0] = PL_SCALAR; d->n_pos_dim = 1; }
        else if (is_prefix("placement_scalar", name)) { d->n_sig = 1; d->sig[0] = PL_SCALAR; d->n_pos_dim = 1; d->is_rama = 1; }
        else return -1;
        d->layer = h5_read_i(grp, "layer_index", 1, dims); if (!d->layer) return -1; d->n_elem = (int)dims[0];
        d->affine_residue = h5_read_i(grp, "affine_residue", 1, dims); if (!d->affine_residue) return -1;
        if (d->is_rama) {
            d->rama_residue = h5_read_i(grp, "rama_residue", 1, dims); if (!d->rama_residue) return -1;
            double* raw = h5_read_d(grp, "placement_data", 4, dims); if (!raw || (int)dims[3] != d->n_pos_dim) return -1;
            d->s.n_layer = (int)dims[0]; d->s.nx = (int)dims[1]; d->s.ny = (int)dims[2]; d->s.ndim = d->n_pos_dim;
            spline2d_fit(&d->s, raw); free(raw);
            d->rama_deriv = (float*)xcalloc((size_t)d->n_elem * 2 * d->n_pos_dim, sizeof(float));
        } else {
            d->fixed_data = h5_read_f(grp, "placement_data", 2, dims); if (!d->fixed_data || (int)dims[1] != d->n_pos_dim) return -1;
            d->n_layer = (int)dims[0];
            d->param_deriv = (float*)xcalloc((size_t)d->n_layer * d->n_pos_dim, sizeof(float));
        }
        coord_node(n, d->n_elem, d->n_pos_dim); n->data = d; n->compute_value = placement_value; n->propagate_deriv = placement_deriv; return 0;
    }
    if (is_prefix("backbone_pairs", name)) {
        BackboneData* d = (BackboneData*)xcalloc(1, sizeof(*d));
        d->id = h5_read_i(grp, "id", 1, dims); if (!d->id) return -1; d->n_res = (int)dims[0];
        d->residue = d->id;
        d->n_atom = h5_read_i(grp, "n_atom", 1, dims); d->ref_pos = h5_read_f(grp, "ref_pos", 3, dims);
        if (!d->n_atom || !d->ref_pos) return -1;
        float max_dev = 0.f;
        for (int nr = 0; nr < d->n_res; ++nr) for (int na = 0; na < d->n_atom[nr]; ++na) {
            const float* p = d->ref_pos + (nr * 4 + na) * 3; max_dev = maxf(sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]), max_dev); }
        d->dist_cutoff = 2 * max_dev + sqrtf(3.f * 3.f + 0.1f * 3.f);
        n->data = d; n->compute_value = backbone_pairs_value; return 0;
    }
    if (is_prefix("rotamer", name)) {
        RotamerData* d = (RotamerData*)xcalloc(1, sizeof(*d));
        hid_t pg = H5Gopen2(grp, "pair_interaction", H5P_DEFAULT); if (pg < 0) return -1;
        int err = igraph_init(&d->g, pg, IT_ROTAMER); H5Gclose(pg); if (err) return -1;
        d->n_prob_nodes = n->n_parent - 1;
        d->damping = h5_attr_f(grp, ".", "damping"); d->tol = h5_attr_f(grp, ".", "tol");
        d->max_iter = h5_attr_i(grp, ".", "max_iter", 1000); d->iteration_chunk_size = h5_attr_i(grp, ".", "iteration_chunk_size", 2);
        /* calculate_n_elem, rotamer.cpp:559-578 */
        const unsigned selector = (1u << N_BIT_ROTAMER) - 1u;
        for (int i = 0; i < d->g.n_elem1; ++i) {
            unsigned id = (unsigned)d->g.id1[i]; unsigned rot = id & selector; id >>= N_BIT_ROTAMER; unsigned n_rot = id & selector; id >>= N_BIT_ROTAMER;
            if (rot >= n_rot || (n_rot != 1 && n_rot != 3 && n_rot != 6)) { fprintf(stderr, "ERROR: invalid rotamer number\n"); return -1; }
            if ((int)id + 1 > d->n_elem_rot[n_rot]) d->n_elem_rot[n_rot] = (int)id + 1;
        }
        static const int rots[3] = {1, 3, 6};
        for (int a = 0; a < 3; ++a) node_holder_init(&d->nodes[rots[a]], rots[a], d->n_elem_rot[rots[a]]);
        for (int a = 0; a < 3; ++a) for (int b = a; b < 3; ++b) {
            int na = d->n_elem_rot[rots[a]], nb = d->n_elem_rot[rots[b]];
            edge_holder_init(&d->edges[rots[a]][rots[b]], &d->nodes[rots[a]], &d->nodes[rots[b]], a == b ? na * (na + 1) / 2 : na * nb);
        }
        n->data = d; n->compute_value = rotamer_value; return 0;
    }
    fprintf(stderr, "ERROR: No node type found for name '%s'\n", name);
    return -1;
}

/* ============================================================================================
 * engine: construction and execution
 * ========================================================================================== */
static int find_node(Engine* e, const char* name) { for (int i = 0; i < e->n_node; ++i) if (!strcmp(e->nodes[i].name, name)) return i; return -1; }

typedef struct { char names[MAX_NODES][64]; int n; } NameList;
static herr_t list_cb(hid_t g, const char* name, const H5L_info_t* info, void* data) {
    (void)g; (void)info; NameList* nl = (NameList*)data;
    if (nl->n < MAX_NODES) { strncpy(nl->names[nl->n], name, 63); nl->n++; }
    return 0;
}
static int cmp_names(const void* a, const void* b) { return strcmp((const char*)a, (const char*)b); }

DerivEngine* construct_deriv_engine(int n_atom, const char* potential_file, bool quiet) {   /* engine_c_library.cpp:9-20 */
    (void)quiet;
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t file = H5Fopen(potential_file, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (file < 0) { fprintf(stderr, "\n\nERROR: cannot open %s\n", potential_file); return NULL; }
    hid_t pg = H5Gopen2(file, "/input/potential", H5P_DEFAULT);
    if (pg < 0) { H5Fclose(file); fprintf(stderr, "\n\nERROR: no /input/potential\n"); return NULL; }
    Engine* e = (Engine*)xcalloc(1, sizeof(Engine));
    e->n_atom = n_atom;
    Node* pos = &e->nodes[0]; strcpy(pos->name, "pos"); coord_node(pos, n_atom, 3); e->n_node = 1;

    /* deriv_engine.cpp:200-226: alphabetical rounds of "all arguments already placed" */
    NameList nl; nl.n = 0;
    hsize_t idx = 0;
    H5Literate(pg, H5_INDEX_NAME, H5_ITER_INC, &idx, list_cb, &nl);
    strcpy(nl.names[nl.n++], "pos");     /* dep_graph["pos"] is an ordinary std::map entry (deriv_engine.cpp:201) */
    qsort(nl.names, nl.n, 64, cmp_names);
    static char args[MAX_NODES][MAX_ARGS][64]; int n_args[MAX_NODES]; int placed[MAX_NODES];
    for (int i = 0; i < nl.n; ++i) {
        placed[i] = 0;
        n_args[i] = strcmp(nl.names[i], "pos") ? h5_attr_strings(pg, nl.names[i], "arguments", args[i], MAX_ARGS) : 0;
        if (n_args[i] < 0) n_args[i] = 0;
    }
    int order[MAX_NODES], n_order = 0;
    for (int round = 0; round <= nl.n; ++round)
        for (int i = 0; i < nl.n; ++i) {
            if (placed[i]) continue;
            int ok = 1;
            for (int a = 0; a < n_args[i]; ++a) {
                int found = 0;   /* in_topo: placed earlier, possibly earlier in this same sweep */
                for (int k = 0; k < n_order && !found; ++k) if (!strcmp(nl.names[order[k]], args[i][a])) found = 1;
                if (!found) { ok = 0; break; }
            }
            if (ok) { order[n_order++] = i; placed[i] = 1; }
        }
    if (n_order != nl.n) { fprintf(stderr, "\n\nERROR: Unsatisfiable dependency in potential computation\n"); goto fail; }

    for (int k = 0; k < n_order; ++k) {
        int i = order[k];
        if (!strcmp(nl.names[i], "pos")) continue;   /* added specially (deriv_engine.cpp:233) */
        Node* n = &e->nodes[e->n_node];
        strncpy(n->name, nl.names[i], 63);
        n->n_parent = n_args[i];
        for (int a = 0; a < n_args[i]; ++a) {
            int p = find_node(e, args[i][a]);
            if (p < 0 || e->nodes[p].potential_term) { fprintf(stderr, "\n\nERROR: %s is not an intermediate value, but it is an argument of %s\n", args[i][a], n->name); goto fail; }
            n->parents[a] = p;
        }
        hid_t grp = H5Gopen2(pg, nl.names[i], H5P_DEFAULT);
        int err = build_node(e, n, grp, nl.names[i]);
        H5Gclose(grp);
        if (err) { fprintf(stderr, "\n\nERROR: while adding '%s'\n", nl.names[i]); goto fail; }
        for (int a = 0; a < n->n_parent; ++a) { Node* p = &e->nodes[n->parents[a]]; p->children[p->n_child++] = e->n_node; }
        e->n_node++;
    }
    H5Gclose(pg); H5Fclose(file);
    return e;
fail:
    H5Gclose(pg); H5Fclose(file); free(e);
    return NULL;
}

void free_deriv_engine(DerivEngine* engine) { free(engine); /* test infrastructure: node payloads are leaked on purpose */ }

/* deriv_engine.cpp:124-169 */
static void engine_compute(Engine* e, int mode) {
    for (int i = 0; i < e->n_node; ++i) e->nodes[i].germ_level = e->nodes[i].deriv_level = -1;
    if (mode == PotentialAndDerivMode) e->potential = 0.f;
    for (int lvl = 0, not_finished = 1;; ++lvl, not_finished = 0) {
        for (int i = 0; i < e->n_node; ++i) {
            Node* n = &e->nodes[i];
            if (n->germ_level == -1) {
                not_finished = 1;
                int all = 1;
                for (int a = 0; a < n->n_parent; ++a) { int l = e->nodes[n->parents[a]].germ_level; if (l == -1 || l == lvl) all = 0; }
                if (all) {
                    if (n->compute_value) n->compute_value(e, n, mode);
                    n->germ_level = lvl;
                    if (mode == PotentialAndDerivMode && n->potential_term) e->potential += n->potential;
                    if (!n->potential_term) memset(n->sens.x, 0, sizeof(float) * n->sens.row_width * n->sens.n_elem);
                }
            }
            if (n->deriv_level == -1 && n->germ_level != -1) {
                not_finished = 1;
                int all = 1;
                for (int c = 0; c < n->n_child; ++c) { int l = e->nodes[n->children[c]].deriv_level; if (l == -1 || l == lvl) all = 0; }
                if (all) {
                    if (n->propagate_deriv && !n->potential_term) n->propagate_deriv(e, n);
                    n->deriv_level = lvl;
                }
            }
        }
        if (!not_finished) break;
    }
}

int evaluate_energy(float* energy, DerivEngine* e, const float* pos) {   /* engine_c_library.cpp:29-46 */
    for (int na = 0; na < e->n_atom; ++na) for (int d = 0; d < 3; ++d) VA(e->nodes[0].output, d, na) = pos[na * 3 + d];
    engine_compute(e, PotentialAndDerivMode);
    *energy = e->potential;
    return 0;
}
int evaluate_deriv(float* deriv, DerivEngine* e, const float* pos) {   /* engine_c_library.cpp:48-64 */
    for (int na = 0; na < e->n_atom; ++na) for (int d = 0; d < 3; ++d) VA(e->nodes[0].output, d, na) = pos[na * 3 + d];
    engine_compute(e, PotentialAndDerivMode);
    for (int na = 0; na < e->n_atom; ++na) for (int d = 0; d < 3; ++d) deriv[na * 3 + d] = VA(e->nodes[0].sens, d, na);
    return 0;
}
int set_param(int n_param, const float* param, DerivEngine* e, const char* node_name) {
    int i = find_node(e, node_name); if (i < 0) { fprintf(stderr, "ERROR: name not found\n"); return 1; }
    Node* n = &e->nodes[i];
    IGraph* g = NULL;
    if (is_prefix("rotamer", n->name)) g = &((RotamerData*)n->data)->g;
    else if (is_prefix("hbond_coverage", n->name) || is_prefix("environment_coverage", n->name) || is_prefix("hbond_sc_radial", n->name)) g = (IGraph*)n->data;
    if (!g || n_param != g->n_type1 * g->n_type2 * g->n_param) { fprintf(stderr, "ERROR: Bad param size\n"); return 1; }
    memcpy(g->param, param, sizeof(float) * n_param); igraph_update_cutoffs(g);
    return 0;
}
int get_param(int n_param, float* param, DerivEngine* e, const char* node_name) {
    int i = find_node(e, node_name); if (i < 0) return 1;
    Node* n = &e->nodes[i];
    IGraph* g = NULL;
    if (is_prefix("rotamer", n->name)) g = &((RotamerData*)n->data)->g;
    else if (is_prefix("hbond_coverage", n->name) || is_prefix("environment_coverage", n->name) || is_prefix("hbond_sc_radial", n->name)) g = (IGraph*)n->data;
    if (!g || n_param != g->n_type1 * g->n_type2 * g->n_param) { fprintf(stderr, "ERROR: Wrong number of parameters\n"); return 1; }
    memcpy(param, g->param, sizeof(float) * n_param);
    return 0;
}
/* engine_c_library.cpp:93-100 as compiled with -DPARAM_DERIV; nodes without an override have an empty vector
 * (deriv_engine.h:71-74) */
int get_param_deriv(int n_param, float* deriv, DerivEngine* e, const char* node_name) {
    int i = find_node(e, node_name); if (i < 0) { fprintf(stderr, "ERROR: name not found\n"); return 1; }
    Node* n = &e->nodes[i];
    int expected = 0;
    IGraph* g = NULL;
    if (is_prefix("rotamer", n->name)) g = &((RotamerData*)n->data)->g;                                  /* rotamer.cpp:1064-1066 */
    else if (is_prefix("hbond_coverage", n->name) || is_prefix("environment_coverage", n->name) || is_prefix("hbond_sc_radial", n->name)) g = (IGraph*)n->data;   /* hbond.cpp:401-402, environment.cpp:104-105, sidechain_radial.cpp:131-135 */
    if (g) expected = g->n_type1 * g->n_type2 * g->n_param;
    else if (is_prefix("hbond_energy", n->name)) expected = 1;
    else if (is_prefix("nonlinear_coupling", n->name)) { NonlinearData* d = (NonlinearData*)n->data; expected = d->n_restype * d->n_coeff; }
    else if (is_prefix("placement_fixed", n->name)) { PlacementData* d = (PlacementData*)n->data; expected = d->n_layer * d->n_pos_dim; }
    else if (is_prefix("linear_coupling", n->name)) expected = ((LinearCouplingData*)n->data)->n_coupling;
    else if (is_prefix("uniform_transform", n->name)) expected = 2 + ((UniformTransformData*)n->data)->n_coeff;
    if (n_param != expected) { fprintf(stderr, "ERROR: Wrong number of parameters, expected %i but got %i\n", expected, n_param); return 1; }
    if (g) igraph_param_deriv(g, deriv);
    else if (is_prefix("hbond_energy", n->name)) deriv[0] = ((HBondEnergyData*)n->data)->n_hbond;       /* hbond.cpp:447-448 */
    else if (is_prefix("nonlinear_coupling", n->name)) {                                                 /* environment.cpp:375-389 */
        NonlinearData* d = (NonlinearData*)n->data; Node* in = &e->nodes[n->parents[0]];
        for (int k = 0; k < expected; ++k) deriv[k] = 0.f;
        for (int ne = 0; ne < in->n_elem; ++ne) {
            int sb; float result[4];
            clamped_deBoor_coeff_deriv(&sb, result, (VA(in->output, 0, ne) - d->offset) * d->inv_dx, d->n_coeff);
            for (int k = 0; k < 4; ++k) deriv[d->types[ne] * d->n_coeff + sb + k] += result[k];
        }
    } else if (is_prefix("linear_coupling", n->name)) {                                                  /* environment.cpp:301-312 */
        LinearCouplingData* d = (LinearCouplingData*)n->data; Node* in = &e->nodes[n->parents[0]];
        Node* inact = d->has_inact ? &e->nodes[n->parents[1]] : NULL;
        for (int k = 0; k < expected; ++k) deriv[k] = 0.f;
        for (int ne = 0; ne < in->n_elem; ++ne) {
            float act = inact ? 1.f - VA(inact->output, d->inact_dim, ne) : 1.f;   /* sic: not squared there */
            deriv[d->types[ne]] += VA(in->output, 0, ne) * act;
        }
    } else if (is_prefix("uniform_transform", n->name)) {                                                /* environment.cpp:205-221 */
        UniformTransformData* d = (UniformTransformData*)n->data; Node* in = &e->nodes[n->parents[0]];
        for (int k = 0; k < expected; ++k) deriv[k] = 0.f;
        for (int ne = 0; ne < in->n_elem; ++ne) {
            float coord = (VA(in->output, 0, ne) - d->offset) * d->inv_dx;
            float v[2]; clamped_deBoor_vd_scalar(v, d->coeff, coord, d->n_coeff);
            int sb; float result[4]; clamped_deBoor_coeff_deriv(&sb, result, coord, d->n_coeff);
            deriv[0] += v[1]; deriv[1] += v[1] * (VA(in->output, 0, ne) - d->offset);
            for (int k = 0; k < 4; ++k) deriv[2 + sb + k] += result[k];
        }
    } else if (is_prefix("placement_fixed", n->name)) memcpy(deriv, ((PlacementData*)n->data)->param_deriv, sizeof(float) * (size_t)expected);   /* placement.cpp:156-160 */
    return 0;
}

int get_output_dims(int* n_elem, int* elem_width, DerivEngine* e, const char* node_name) {
    int i = find_node(e, node_name); if (i < 0) { fprintf(stderr, "ERROR: name not found\n"); return 1; }
    *n_elem = e->nodes[i].n_elem; *elem_width = e->nodes[i].elem_width; return 0;
}
static int get_array(int n_output, float* out, DerivEngine* e, const char* node_name, int want_sens) {
    int i = find_node(e, node_name); if (i < 0) { fprintf(stderr, "ERROR: name not found\n"); return 1; }
    Node* n = &e->nodes[i];
    if (n->potential_term) { if (n_output != 1) return 1; *out = n->potential; return 0; }
    if (n_output != n->n_elem * n->elem_width) { fprintf(stderr, "ERROR: wrong size for CoordNode\n"); return 1; }
    VecArray a = want_sens ? n->sens : n->output;
    for (int ne = 0; ne < n->n_elem; ++ne) for (int d = 0; d < n->elem_width; ++d) out[ne * n->elem_width + d] = VA(a, d, ne);
    return 0;
}
int get_output(int n_output, float* output, DerivEngine* e, const char* node_name) { return get_array(n_output, output, e, node_name, 0); }
int get_sens(int n_output, float* output, DerivEngine* e, const char* node_name) { return get_array(n_output, output, e, node_name, 1); }

int get_value_by_name(int n_output, float* output, DerivEngine* e, const char* node_name, const char* log_name) {
    int i = find_node(e, node_name); if (i < 0) { fprintf(stderr, "ERROR: name not found\n"); return 1; }
    Node* n = &e->nodes[i];
    if (is_prefix("rotamer", n->name)) return rotamer_get_value(e, n, log_name, n_output, output);
    if (is_prefix("hbond_coverage", n->name) && !strcmp(log_name, "count_edges_by_type")) {
        IGraph* g = (IGraph*)n->data; if (n_output != g->n_type1 * g->n_type2) return 1; igraph_count_edges_by_type(g, output); return 0; }
    fprintf(stderr, "ERROR: No values implemented\n");
    return 1;
}

int oracle_get_pairlist(DerivEngine* e, const char* node_name, int max_edge, int* i1, int* i2) {
    int i = find_node(e, node_name); if (i < 0) return -1;
    Node* n = &e->nodes[i]; IGraph* g = NULL;
    if (is_prefix("rotamer", n->name)) g = &((RotamerData*)n->data)->g;
    else if (is_prefix("protein_hbond", n->name)) g = &((ProteinHBondData*)n->data)->g;
    else if (is_prefix("hbond_coverage", n->name) || is_prefix("environment_coverage", n->name)) g = (IGraph*)n->data;
    if (!g) return -1;
    for (int k = 0; k < g->n_edge && k < max_edge; ++k) { i1[k] = g->edge_i1[k]; i2[k] = g->edge_i2[k]; }
    return g->n_edge;
}
int oracle_rotamer_iterations(DerivEngine* e) { return e->rotamer_iterations; }

/* ---- spline helpers of the C-ABI (engine_c_library.cpp:196-276) ---- */
int clamped_spline_solve(int N, float* bspline_coeff, const float* values) {
    double* tmp = (double*)xcalloc(3 * N, sizeof(double)); double* c = (double*)xcalloc(N, sizeof(double)); double* v = (double*)xcalloc(N, sizeof(double));
    for (int i = 0; i < N - 2; ++i) v[i] = values[i];
    solve_clamped_1d_spline_for_bsplines(N, c, v, tmp);
    for (int i = 0; i < N; ++i) bspline_coeff[i] = (float)c[i];
    free(tmp); free(c); free(v); return 0;
}
int clamped_spline_value(int N, float* result, const float* bspline_coeff, int nx, float* x) {
    for (int i = 0; i < nx; ++i) { float r[2]; clamped_deBoor_vd(r, bspline_coeff, x[i], N); result[i] = r[0]; } return 0; }
int get_clamped_value_and_deriv(int N, float* result, const float* bspline_coeff, int nx, float* x) {
    for (int i = 0; i < nx; ++i) { float r[2]; clamped_deBoor_vd_scalar(r, bspline_coeff, x[i], N); result[i * 2] = r[0]; result[i * 2 + 1] = r[1]; } return 0; }
int get_clamped_coeff_deriv(int N, float* result, const float* bspline_coeff, float x) {
    (void)bspline_coeff; int sb; float data[4]; clamped_deBoor_coeff_deriv(&sb, data, x, N);
    for (int i = 0; i < N; ++i) result[i] = 0.f;
    for (int i = 0; i < 4; ++i) result[sb + i] = data[i];
    return 0;
}
int upside_main(int argc, const char* const* argv, int verbose) { (void)argc; (void)argv; (void)verbose; fprintf(stderr, "oracle: upside_main is not restated (CLI is out of scope)\n"); return 1; }

/* ============================================================================================
 * RNG, thermostat, integrator
 * ========================================================================================== */
static uint32_t rotl32(uint32_t x, unsigned n) { return (x << (n & 31)) | (x >> ((32 - n) & 31)); }

void oracle_threefry4x32(uint32_t out[4], const uint32_t ctr[4], const uint32_t key[4]) {   /* Random123/threefry.h:296-430 */
    static const unsigned R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
    uint32_t ks[5], X[4];
    ks[4] = 0x1BD11BDA;
    for (int i = 0; i < 4; ++i) { ks[i] = key[i]; X[i] = ctr[i]; ks[4] ^= key[i]; }
    for (int i = 0; i < 4; ++i) X[i] += ks[i];
    for (int r = 0; r < 20; ++r) {
        if (r % 2 == 0) { X[0] += X[1]; X[1] = rotl32(X[1], R[r % 8][0]); X[1] ^= X[0]; X[2] += X[3]; X[3] = rotl32(X[3], R[r % 8][1]); X[3] ^= X[2]; }
        else            { X[0] += X[3]; X[3] = rotl32(X[3], R[r % 8][0]); X[3] ^= X[0]; X[2] += X[1]; X[1] = rotl32(X[1], R[r % 8][1]); X[1] ^= X[2]; }
        if (r % 4 == 3) { int k = r / 4 + 1; for (int i = 0; i < 4; ++i) X[i] += ks[(k + i) % 5]; X[3] += k; }
    }
    for (int i = 0; i < 4; ++i) out[i] = X[i];
}
static float u01f(uint32_t in) { const float factor = 1.f / 4294967296.f; return (float)in * factor + 0.5f * factor; }           /* uniform.hpp:145-153 */
static float uneg11f(uint32_t in) { const float factor = 1.f / 2147483648.f; return (float)(int32_t)in * factor + 0.5f * factor; } /* uniform.hpp:171-179 */
static void boxmuller(float out[2], uint32_t u0, uint32_t u1) {   /* boxmuller.hpp boxmuller(uint32_t,uint32_t) */
    const float PIf = 3.1415926535897932f;
    float a = PIf * uneg11f(u0);
    float r = sqrtf(-2.f * logf(u01f(u1)));
    out[0] = sinf(a) * r; out[1] = cosf(a) * r;
}
static void rng_bits(uint32_t bits[4], uint32_t seed, uint32_t stream, uint32_t atom, uint64_t timestep, uint32_t draw) {   /* random.h:19-44 */
    uint32_t k[4] = {seed, stream, 0u, 0u};
    uint32_t c[4] = {(uint32_t)(timestep & 0xffffffffu), (uint32_t)((timestep >> 32) & 0xffffffffu), atom, draw};
    oracle_threefry4x32(bits, c, k);
}
void oracle_random_normal4(float out[4], uint32_t seed, uint32_t stream, uint32_t atom, uint64_t timestep) {   /* random.h:55-60 */
    uint32_t b[4]; rng_bits(b, seed, stream, atom, timestep, 0u);
    boxmuller(out, b[0], b[1]); boxmuller(out + 2, b[2], b[3]);
}
void oracle_random_uniform4(float out[4], uint32_t seed, uint32_t stream, uint32_t atom, uint64_t timestep, uint32_t draw) {   /* random.h:46-53 */
    uint32_t b[4]; rng_bits(b, seed, stream, atom, timestep, draw);
    for (int i = 0; i < 4; ++i) out[i] = u01f(b[i]);
}

typedef struct { uint64_t n_invocations; uint32_t seed; float timescale, delta_t, mom_scale, temp, noise_scale; } Thermostat;
static void thermostat_update(Thermostat* t) {   /* thermostat.h:9-12 */
    t->mom_scale = (float)exp(-t->delta_t / t->timescale);
    t->noise_scale = sqrtf(t->temp * (1 - t->mom_scale * t->mom_scale));
}
static void thermostat_apply(Thermostat* t, float* mom, int n_atom) {   /* thermostat.cpp:9-18 */
    for (int na = 0; na < n_atom; ++na) {
        float nrm[4]; oracle_random_normal4(nrm, t->seed, 0u /* THERMOSTAT_RANDOM_STREAM */, (uint32_t)na, t->n_invocations);
        for (int d = 0; d < 3; ++d) mom[na * 3 + d] = t->mom_scale * mom[na * 3 + d] + t->noise_scale * nrm[d];
    }
    t->n_invocations++;
}

/* integrator_type: 0 = Verlet, 1 = Predescu (deriv_engine.h:230, deriv_engine.cpp:173-180) */
int oracle_run_md_integrator(DerivEngine* e, float* pos, float* mom, int n_round, float dt, float temperature, uint32_t seed,
                             float thermostat_timescale, int thermostat_interval_rounds, int integrator_type);
int oracle_run_md(DerivEngine* e, float* pos, float* mom, int n_round, float dt, float temperature, uint32_t seed,
                  float thermostat_timescale, int thermostat_interval_rounds) {
    return oracle_run_md_integrator(e, pos, mom, n_round, dt, temperature, seed, thermostat_timescale, thermostat_interval_rounds, 0);
}
int oracle_run_md_integrator(DerivEngine* e, float* pos, float* mom, int n_round, float dt, float temperature, uint32_t seed,
                             float thermostat_timescale, int thermostat_interval_rounds, int integrator_type) {
    int n_atom = e->n_atom;
    Node* p = &e->nodes[0];
    for (int na = 0; na < n_atom; ++na) for (int d = 0; d < 3; ++d) { VA(p->output, d, na) = pos[na * 3 + d]; mom[na * 3 + d] = 0.f; }
    /* main.cpp:515-523 */
    Thermostat t; memset(&t, 0, sizeof(t));
    t.seed = seed; t.timescale = thermostat_timescale; t.temp = 1.f; t.delta_t = 1e8f; thermostat_update(&t);
    t.temp = temperature; thermostat_update(&t);
    thermostat_apply(&t, mom, n_atom);
    t.delta_t = thermostat_interval_rounds * 3 * dt; thermostat_update(&t);
    for (int round = 0; round < n_round; ++round) {   /* main.cpp:621-666 */
        if (!(round % thermostat_interval_rounds)) thermostat_apply(&t, mom, n_atom);
        for (int stage = 0; stage < 3; ++stage) {   /* deriv_engine.cpp:172-192: Verlet weights {1,1,1}, or Predescu et al. 2012 */
            engine_compute(e, DerivMode);
            float a = integrator_type == 1 ? (float)0.108991425403425322 : (float)(1. / 6.);   /* deriv_engine.cpp:176-177: double constants narrowed */
            float b = integrator_type == 1 ? (float)0.290485609075128726 : (float)(1. / 3.);
            float mom_update[3] = {1.5f - 3.f * a, 1.5f - 3.f * a, 6.f * a};
            float pos_update[3] = {3.f * b, 3.0f - 6.f * b, 3.f * b};
            float vel_factor = dt * mom_update[stage], pos_factor = dt * pos_update[stage];
            for (int na = 0; na < n_atom; ++na) for (int d = 0; d < 3; ++d) {   /* deriv_engine.cpp:11-35 (max_force = 0) */
                float pp = mom[na * 3 + d] - vel_factor * VA(p->sens, d, na);
                mom[na * 3 + d] = pp;
                VA(p->output, d, na) += pos_factor * pp;
            }
        }
    }
    for (int na = 0; na < n_atom; ++na) for (int d = 0; d < 3; ++d) pos[na * 3 + d] = VA(p->output, d, na);
    return 0;
}
