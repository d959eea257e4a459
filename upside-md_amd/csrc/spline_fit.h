// Host-side, construction-time fitting of interpolating cubic splines (double precision) for the device tables: the
// Ramachandran maps and the rotamer placement maps (periodic, bicubic patches) and the membrane profiles (clamped ends).
// What has to come out is fixed by the reference -- the same interpolant in the same storage layout
// (/root/reference/src/spline.h:396-431, 456-493 read the tables as per-cell monomial coefficients; the maths is the
// uniform cubic B-spline through the samples, spline.cpp) -- how it is computed is not: here one generic routine solves the
// constant-coefficient interpolation system  (c[i-1] + 4 c[i] + c[i+1]) / 6 = y[i]  for the B-spline control values with
// either boundary rule (wrap-around or mirrored ghosts = zero end slope) by plain elimination that carries the wrap-around
// column along, and one 4x4 matrix turns four consecutive control values into the monomial coefficients of a cell.
#pragma once
#include <cstddef>
#include <vector>
using std::size_t;

namespace tablefit {

enum class Ends { Periodic, ZeroSlope };

// control values c[0..n) of the uniform cubic B-spline that interpolates y[0..n) on the integer grid.
// Periodic: c[-1] = c[n-1], c[n] = c[0].  ZeroSlope: mirrored ghosts c[-1] = c[1], c[n] = c[n-2] (first derivative 0 at both ends).
inline std::vector<double> control_values(const double* y, int n, Ends ends) {
    std::vector<double> c(y, y + n);
    if (n == 1) { c[0] = y[0]; return c; }            // a constant
    // rows: lower[i] c[i-1] + diag[i] c[i] + upper[i] c[i+1] (+ wrap[i] c[n-1] for the periodic corner) = rhs[i]
    std::vector<double> diag(n, 4. / 6.), upper(n, 1. / 6.), lower(n, 1. / 6.), wrap(n, 0.);
    double last_row_first = 0.;                        // coefficient of c[0] in the last row (periodic corner)
    if (ends == Ends::ZeroSlope) { upper[0] = 2. / 6.; lower[n - 1] = 2. / 6.; }
    else if (n == 2) { upper[0] = 2. / 6.; lower[1] = 2. / 6.; }      // both neighbours of a point are the other point
    else { wrap[0] = 1. / 6.; last_row_first = 1. / 6.; }
    // forward elimination of the sub-diagonal; the last row is kept apart and reduced against every pivot row as well, so
    // its c[0] entry (the second corner) travels right until it lands on the last two columns
    double lr_coef = last_row_first;                   // entry of the last row in the current pivot column
    double lr_diag = diag[n - 1], lr_rhs = c[n - 1];   // its c[n-1] entry and right-hand side
    double lr_prev = lower[n - 1];                     // its genuine c[n-2] entry
    for (int i = 0; i + 1 < n; ++i) {
        const bool last_pivot = i == n - 2;
        if (last_pivot) lr_coef += lr_prev;            // the wandering entry has reached column n-2
        // reduce the last row against pivot row i (entries of row i: diag[i] at i, upper[i] at i+1, wrap[i] at n-1)
        const double f = lr_coef / diag[i];
        lr_rhs -= f * c[i];
        lr_diag -= f * (last_pivot ? upper[i] + wrap[i] : wrap[i]);
        lr_coef = last_pivot ? 0. : -f * upper[i];     // becomes the entry in column i+1
        if (!last_pivot) {                             // reduce row i+1 (its sub-diagonal) against row i
            const double g = lower[i + 1] / diag[i];
            diag[i + 1] -= g * upper[i];
            wrap[i + 1] -= g * wrap[i];
            c[i + 1] -= g * c[i];
        }
    }
    // back substitution (row n-2's "upper" and "wrap" are the same column)
    c[n - 1] = lr_rhs / lr_diag;
    for (int i = n - 2; i >= 0; --i) {
        const double tail = i == n - 2 ? (upper[i] + wrap[i]) * c[n - 1] : upper[i] * c[i + 1] + wrap[i] * c[n - 1];
        c[i] = (c[i] - tail) / diag[i];
    }
    return c;
}

// monomial coefficients (powers 0..3 of the offset t in [0,1) from the cell's left sample) of the cell whose four
// relevant control values are c[-1], c[0], c[1], c[2]
static const double kCellFromControl[4][4] = {
    { 1. / 6.,  4. / 6.,  1. / 6., 0.      },
    {-3. / 6.,  0.,       3. / 6., 0.      },
    { 3. / 6., -6. / 6.,  3. / 6., 0.      },
    {-1. / 6.,  3. / 6., -3. / 6., 1. / 6. }};

inline int wrap_index(int i, int n) { i %= n; return i < 0 ? i + n : i; }
inline int mirror_index(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// (n-1 cells) x 4 monomial coefficients of the zero-end-slope interpolant through y[0..n)
inline std::vector<double> clamped_cells(const double* y, int n) {
    const std::vector<double> c = control_values(y, n, Ends::ZeroSlope);
    std::vector<double> cell((size_t)(n - 1) * 4);
    for (int i = 0; i + 1 < n; ++i)
        for (int p = 0; p < 4; ++p) {
            double v = 0.;
            for (int k = 0; k < 4; ++k) v += kCellFromControl[p][k] * c[mirror_index(i - 1 + k, n)];
            cell[(size_t)i * 4 + p] = v;
        }
    return cell;
}

// nx x ny cells x 16 (power of x major, power of y minor) of the doubly periodic interpolant through y[ix*ny + iy]
inline std::vector<double> periodic_patches(const double* y, int nx, int ny) {
    // control values of the tensor-product spline: the 1-d solve along y for every x, then along x for every y
    std::vector<double> ctl((size_t)nx * ny), line(nx > ny ? nx : ny);
    for (int ix = 0; ix < nx; ++ix) {
        const std::vector<double> c = control_values(y + (size_t)ix * ny, ny, Ends::Periodic);
        for (int iy = 0; iy < ny; ++iy) ctl[(size_t)ix * ny + iy] = c[iy];
    }
    for (int iy = 0; iy < ny; ++iy) {
        for (int ix = 0; ix < nx; ++ix) line[ix] = ctl[(size_t)ix * ny + iy];
        const std::vector<double> c = control_values(line.data(), nx, Ends::Periodic);
        for (int ix = 0; ix < nx; ++ix) ctl[(size_t)ix * ny + iy] = c[ix];
    }
    std::vector<double> patch((size_t)nx * ny * 16);
    for (int ix = 0; ix < nx; ++ix)
        for (int iy = 0; iy < ny; ++iy) {
            double nb[4][4], half[4][4];       // the cell's 4x4 control values; after the x transform
            for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) nb[a][b] = ctl[(size_t)wrap_index(ix - 1 + a, nx) * ny + wrap_index(iy - 1 + b, ny)];
            for (int px = 0; px < 4; ++px) for (int b = 0; b < 4; ++b) {
                double v = 0.;
                for (int a = 0; a < 4; ++a) v += kCellFromControl[px][a] * nb[a][b];
                half[px][b] = v;
            }
            double* out = &patch[((size_t)ix * ny + iy) * 16];
            for (int px = 0; px < 4; ++px) for (int py = 0; py < 4; ++py) {
                double v = 0.;
                for (int b = 0; b < 4; ++b) v += kCellFromControl[py][b] * half[px][b];
                out[px * 4 + py] = v;
            }
        }
    return patch;
}

// the N + 2 control values (ghosts included) of the zero-end-slope interpolant through N samples: what the C-ABI's
// clamped_spline_solve returns (engine_c_library.cpp:196-212)
inline std::vector<double> clamped_control_values_with_ghosts(const double* y, int n) {
    const std::vector<double> c = control_values(y, n, Ends::ZeroSlope);
    std::vector<double> out((size_t)n + 2);
    for (int i = 0; i < n; ++i) out[i + 1] = c[i];
    out[0] = c[mirror_index(-1, n)]; out[n + 1] = c[mirror_index(n, n)];
    return out;
}

}  // namespace tablefit

// data (n_layer,nx) -> fp32 (n_layer,nx-1,4): the membrane profiles (reference storage: spline.h:456-493)
inline std::vector<float> fit_layered_clamped_spline1d(const std::vector<double>& data, int n_layer, int nx) {
    std::vector<float> out;
    out.reserve((size_t)n_layer * (nx - 1) * 4);
    for (int il = 0; il < n_layer; ++il)
        for (double v : tablefit::clamped_cells(data.data() + (size_t)il * nx, nx)) out.push_back((float)v);
    return out;
}

// data (n_layer,nx,ny,ndim) -> fp32 bicubic patch coefficients (n_layer,nx,ny,ndim,16) (reference storage: spline.h:396-431)
inline std::vector<float> fit_layered_periodic_spline2d(const std::vector<double>& data, int n_layer, int nx, int ny, int ndim) {
    std::vector<float> out((size_t)n_layer * nx * ny * ndim * 16);
    std::vector<double> plane((size_t)nx * ny);
    for (int il = 0; il < n_layer; ++il)
        for (int id = 0; id < ndim; ++id) {
            for (size_t cell = 0; cell < plane.size(); ++cell) plane[cell] = data[((size_t)il * nx * ny + cell) * ndim + id];
            const std::vector<double> patch = tablefit::periodic_patches(plane.data(), nx, ny);
            for (size_t cell = 0; cell < plane.size(); ++cell)
                for (int k = 0; k < 16; ++k) out[(((size_t)il * nx * ny + cell) * ndim + id) * 16 + k] = (float)patch[cell * 16 + k];
        }
    return out;
}
