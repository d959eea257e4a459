// Host-side, construction-time spline fitting (double precision), as in the reference where tables are
// fitted once while the engine is built: /root/reference/src/spline.cpp:7-156, 158-189, 262-292 and
// spline.h:396-431 (LayeredPeriodicSpline2D::fit_spline).  Also the four scalar spline helpers the C-ABI
// exports (engine_c_library.cpp:196-276), which never touch an engine.
#pragma once
#include <cstring>
#include <vector>

namespace splinefit {

inline void solve_tridiagonal_system(int n, double* d, double* a, double* b, double* c) {   // Thomas algorithm
    for (int k = 1; k < n; ++k) { const double m = a[k - 1] / b[k - 1]; b[k] -= m * c[k - 1]; d[k] -= m * d[k - 1]; }
    d[n - 1] = d[n - 1] / b[n - 1];
    for (int k = n - 2; k >= 0; --k) d[k] = (d[k] - c[k] * d[k + 1]) / b[k];
}

// periodic tridiagonal system through the Sherman-Morrison formula
inline void solve_periodic_tridiagonal_system(int n, double* solution, double* d, double* a, double* b, double* c, double* tmp) {
    const double b1 = b[0], cn = c[n - 1], ratio = a[0] / b1;
    b[0] += b1; b[n - 1] += ratio * cn;
    std::memcpy(tmp, a, n * sizeof(double)); std::memcpy(tmp + n, b, n * sizeof(double)); std::memcpy(tmp + 2 * n, c, n * sizeof(double));
    solution[0] = -b1;
    for (int i = 1; i < n - 1; ++i) solution[i] = 0.;
    solution[n - 1] = cn;
    double* q = solution;
    solve_tridiagonal_system(n, q, a + 1, b, c);
    std::memcpy(a, tmp, n * sizeof(double)); std::memcpy(b, tmp + n, n * sizeof(double)); std::memcpy(c, tmp + 2 * n, n * sizeof(double));
    double* y = d;
    solve_tridiagonal_system(n, d, a + 1, b, c);
    const double q_prefactor = (y[0] - y[n - 1] * ratio) / (1. + q[0] - q[n - 1] * ratio);
    for (int i = 0; i < n; ++i) solution[i] = y[i] - q_prefactor * q[i];
}

// cubic B-spline pieces in the monomial basis on [0,1)
static const double kBsplineMonomial[4][4] = {
    {0. / 6., 0. / 6., 0. / 6., 1. / 6.}, {1. / 6., 3. / 6., 3. / 6., -3. / 6.},
    {4. / 6., 0. / 6., -6. / 6., 3. / 6.}, {1. / 6., -3. / 6., 3. / 6., -1. / 6.}};

inline void solve_periodic_1d_spline(int n, double* coefficients, const double* data, double* ts) {
    double *a = ts, *b = ts + n, *c = ts + 2 * n, *d = ts + 3 * n, *solution = ts + 4 * n, *later = ts + 5 * n;
    for (int i = 0; i < n; ++i) { a[i] = 1. / 6.; b[i] = 2. / 3.; c[i] = 1. / 6.; d[i] = data[i]; }
    solve_periodic_tridiagonal_system(n, solution, d, a, b, c, later);
    for (int i = 0; i < 4 * n; ++i) coefficients[i] = 0.;
    for (int i = 0; i < n; ++i)
        for (int inc = 0; inc < 4; ++inc) {
            int idx = i + inc - 2;
            if (idx < 0) idx += n;
            if (idx >= n) idx -= n;
            for (int k = 0; k < 4; ++k) coefficients[idx * 4 + k] += solution[i] * kBsplineMonomial[inc][k];
        }
}

inline void solve_periodic_2d_spline(int nx, int ny, double* coefficients, const double* data, double* ts) {
    const int sum_dim = nx + ny;
    double* splines_1d = ts;
    double* scratch = splines_1d + (size_t)nx * ny * 4;
    double* values_temp = scratch + sum_dim * 8;
    double* coeffs_temp = values_temp + sum_dim * 4;
    for (int ix = 0; ix < nx; ++ix) solve_periodic_1d_spline(ny, splines_1d + (size_t)ix * ny * 4, data + (size_t)ix * ny, scratch);
    for (int iy = 0; iy < ny; ++iy)
        for (int py = 0; py < 4; ++py) {
            for (int ix = 0; ix < nx; ++ix) values_temp[ix] = splines_1d[(size_t)ix * ny * 4 + iy * 4 + py];
            solve_periodic_1d_spline(nx, coeffs_temp, values_temp, scratch);
            for (int ix = 0; ix < nx; ++ix)
                for (int px = 0; px < 4; ++px) coefficients[(size_t)ix * ny * 16 + iy * 16 + px * 4 + py] = coeffs_temp[ix * 4 + px];
        }
}

inline void solve_clamped_1d_spline_for_bsplines(int n_coeff, double* coefficients, const double* data, double* ts) {
    const int n = n_coeff - 2;
    double *a = ts, *b = ts + n_coeff, *c = ts + 2 * n_coeff;
    for (int i = 0; i < n; ++i) { a[i] = 1. / 6.; b[i] = 2. / 3.; c[i] = 1. / 6.; coefficients[i + 1] = data[i]; }
    a[n - 1] *= 2.; c[0] *= 2.;
    solve_tridiagonal_system(n_coeff - 2, coefficients + 1, a + 1, b, c);
    coefficients[0] = coefficients[2];
    coefficients[n_coeff - 1] = coefficients[n_coeff - 3];
}

// spline.cpp:192-259: clamped (zero end-slope) interpolating cubic through n points on the integer grid, as monomial
// coefficients of its n-1 intervals
inline void solve_clamped_1d_spline(int n, double* coefficients, const double* data, double* ts) {
    double *a = ts, *b = ts + n, *c = ts + 2 * n, *solution = ts + 3 * n;
    for (int i = 0; i < n; ++i) { a[i] = 1. / 6.; b[i] = 2. / 3.; c[i] = 1. / 6.; solution[i] = data[i]; }
    a[n - 1] *= 2.; c[0] *= 2.;
    solve_tridiagonal_system(n, solution, a + 1, b, c);
    for (int i = 0; i < 4 * (n - 1); ++i) coefficients[i] = 0.;
    for (int i = 0; i < n; ++i)
        for (int inc = 0; inc < 4; ++inc) {
            const int idx = i + inc - 2;
            if (idx < 0 || idx >= n - 1) continue;
            for (int k = 0; k < 4; ++k) coefficients[idx * 4 + k] += solution[i] * kBsplineMonomial[inc][k];
        }
    for (int k = 0; k < 4; ++k) coefficients[k] += solution[1] * kBsplineMonomial[3][k];                      // left wing
    for (int k = 0; k < 4; ++k) coefficients[(n - 2) * 4 + k] += solution[n - 2] * kBsplineMonomial[0][k];    // right wing
}

}  // namespace splinefit

// spline.h:456-493 LayeredClampedSpline1D<1>::fit_spline: data (n_layer,nx) -> fp32 (n_layer,nx-1,4)
inline std::vector<float> fit_layered_clamped_spline1d(const std::vector<double>& data, int n_layer, int nx) {
    std::vector<float> out((size_t)n_layer * (nx - 1) * 4);
    std::vector<double> coeff((size_t)(nx - 1) * 4), ts((size_t)4 * nx);
    for (int il = 0; il < n_layer; ++il) {
        splinefit::solve_clamped_1d_spline(nx, coeff.data(), data.data() + (size_t)il * nx, ts.data());
        for (size_t i = 0; i < coeff.size(); ++i) out[(size_t)il * (nx - 1) * 4 + i] = (float)coeff[i];
    }
    return out;
}

// data (n_layer,nx,ny,ndim) -> fp32 bicubic patch coefficients (n_layer,nx,ny,ndim,16)
inline std::vector<float> fit_layered_periodic_spline2d(const std::vector<double>& data, int n_layer, int nx, int ny, int ndim) {
    std::vector<float> out((size_t)n_layer * nx * ny * ndim * 16);
    std::vector<double> coeff_tmp((size_t)nx * ny * 16), data_tmp((size_t)nx * ny), ts((size_t)(nx + 8) * (ny + 8) * 4 + 64 * (size_t)(nx + ny));
    for (int il = 0; il < n_layer; ++il)
        for (int id = 0; id < ndim; ++id) {
            for (int ix = 0; ix < nx; ++ix) for (int iy = 0; iy < ny; ++iy)
                data_tmp[(size_t)ix * ny + iy] = data[(((size_t)il * nx + ix) * ny + iy) * ndim + id];
            splinefit::solve_periodic_2d_spline(nx, ny, coeff_tmp.data(), data_tmp.data(), ts.data());
            for (int ix = 0; ix < nx; ++ix) for (int iy = 0; iy < ny; ++iy) for (int ic = 0; ic < 16; ++ic)
                out[((((size_t)il * nx + ix) * ny + iy) * ndim + id) * 16 + ic] = (float)coeff_tmp[((size_t)ix * ny + iy) * 16 + ic];
        }
    return out;
}
