// slam_weyl.hpp -- batched Weyl-chamber coordinates on the device (gfx950 only).
//
// weylchamber.c1c2c3 (called at src/slam/basis_abc.py:80-84 and src/slam/optimizer.py:85,103,224; algorithm
// restated in SURVEY.md Appendix A-4): eigenvalues of U U~ / sqrt(det U), U~ = (sy x sy) U^T (sy x sy); their
// phases are sorted, folded into the Weyl chamber and rounded to `ndigits` decimals.
//
// The reference obtains the eigenvalues from LAPACK's general complex eigensolver.  Here one thread handles one
// 4x4 unitary and uses the structure instead: in the magic basis Q, U_B = Q^+ U Q and the spectrum of U U~ is
// that of m = U_B U_B^T, a *symmetric* unitary matrix.  m = X + iY with X, Y real symmetric and commuting, so one
// real orthogonal matrix diagonalises both: cyclic Jacobi sweeps with the joint-diagonalisation angle
// (Cardoso-Souloumiac; it degenerates to the classical Jacobi angle for a single matrix) -- robust for the
// degenerate spectra of CX / iSWAP / SWAP-like gates, where a characteristic-polynomial solver loses half the
// digits.  Everything is fp64; this is bookkeeping (two calls per target), not a hot loop.
#pragma once
#include "slam_device.hpp"

namespace slamdev {

struct cplx { double re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }

// det of a 3x3 complex minor of U (rows r0 r1 r2, columns c0 c1 c2)
__device__ __forceinline__ cplx det3(const cplx (&U)[4][4], int r0, int r1, int r2, int c0, int c1, int c2) {
    const cplx a = cmul(U[r0][c0], csub(cmul(U[r1][c1], U[r2][c2]), cmul(U[r1][c2], U[r2][c1])));
    const cplx b = cmul(U[r0][c1], csub(cmul(U[r1][c0], U[r2][c2]), cmul(U[r1][c2], U[r2][c0])));
    const cplx c = cmul(U[r0][c2], csub(cmul(U[r1][c0], U[r2][c1]), cmul(U[r1][c1], U[r2][c0])));
    return cadd(csub(a, b), c);
}

// out[3] = (c1, c2, c3) of the 4x4 unitary at U (row-major re,im), in units of pi.
// ndigits >= 0: rounded like numpy.round(v, ndigits); < 0: unrounded.
__device__ inline void weyl_c1c2c3(const double* __restrict__ Uin, int ndigits, double* __restrict__ out) {
    cplx U[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) U[r][c] = {Uin[(r * 4 + c) * 2], Uin[(r * 4 + c) * 2 + 1]};
    // det U (Laplace expansion along row 0)
    cplx det = cmul(U[0][0], det3(U, 1, 2, 3, 1, 2, 3));
    det = csub(det, cmul(U[0][1], det3(U, 1, 2, 3, 0, 2, 3)));
    det = cadd(det, cmul(U[0][2], det3(U, 1, 2, 3, 0, 1, 3)));
    det = csub(det, cmul(U[0][3], det3(U, 1, 2, 3, 0, 1, 2)));
    // magic basis Q = 1/sqrt2 [[1,0,0,i],[0,i,1,0],[0,i,-1,0],[1,0,0,-i]]:  UB = Q^+ U Q
    const double h = 0.70710678118654752440;
    cplx T[4][4];  // T = U Q : column j of T = sum_k U[.][k] Q[k][j]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const cplx u0 = U[r][0], u1 = U[r][1], u2 = U[r][2], u3 = U[r][3];
        T[r][0] = {h * (u0.re + u3.re), h * (u0.im + u3.im)};                      // Q[0][0] = Q[3][0] = 1
        T[r][1] = {h * (-u1.im - u2.im), h * (u1.re + u2.re)};                     // Q[1][1] = Q[2][1] = i
        T[r][2] = {h * (u1.re - u2.re), h * (u1.im - u2.im)};                      // Q[1][2] = 1, Q[2][2] = -1
        T[r][3] = {h * (-u0.im + u3.im), h * (u0.re - u3.re)};                     // Q[0][3] = i, Q[3][3] = -i
    }
    cplx B[4][4];  // B = Q^+ T : row i of B = sum_k conj(Q[k][i]) T[k][.]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const cplx t0 = T[0][c], t1 = T[1][c], t2 = T[2][c], t3 = T[3][c];
        B[0][c] = {h * (t0.re + t3.re), h * (t0.im + t3.im)};
        B[1][c] = {h * (t1.im + t2.im), h * (-t1.re - t2.re)};                     // conj(i) = -i
        B[2][c] = {h * (t1.re - t2.re), h * (t1.im - t2.im)};
        B[3][c] = {h * (t0.im - t3.im), h * (-t0.re + t3.re)};                     // conj(i) = -i on row 0, conj(-i) = i on row 3
    }
    // m = B B^T = X + iY (symmetric)
    double X[4][4], Y[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) {
            cplx s = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 4; ++k) s = cadd(s, cmul(B[i][k], B[j][k]));
            X[i][j] = X[j][i] = s.re;
            Y[i][j] = Y[j][i] = s.im;
        }
    // joint Jacobi diagonalisation of the commuting real symmetric pair (X, Y)
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i + 1; j < 4; ++j) off += X[i][j] * X[i][j] + Y[i][j] * Y[i][j];
        if (off < 1e-31) break;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const double h1x = X[p][p] - X[q][q], h1y = Y[p][p] - Y[q][q];
                const double h2x = 2.0 * X[p][q], h2y = 2.0 * Y[p][q];
                const double ton = (h1x * h1x + h1y * h1y) - (h2x * h2x + h2y * h2y);
                const double toff = 2.0 * (h1x * h2x + h1y * h2y);
                if (toff == 0.0 && ton >= 0.0) continue;  // nothing to rotate (also h1 = h2 = 0)
                const double theta = 0.25 * atan2(toff, ton);
                double sn, cs;
                sincos(theta, &sn, &cs);
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    double (&A)[4][4] = w ? Y : X;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {  // columns p, q
                        const double ap = A[r][p], aq = A[r][q];
                        A[r][p] = cs * ap + sn * aq;
                        A[r][q] = -sn * ap + cs * aq;
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {  // rows p, q
                        const double ap = A[p][c], aq = A[q][c];
                        A[p][c] = cs * ap + sn * aq;
                        A[q][c] = -sn * ap + cs * aq;
                    }
                }
            }
    }
    // eigenvalues of U U~ / sqrt(det U): m_kk / sqrt(det), principal square root
    const double dabs = sqrt(sqrt(det.re * det.re + det.im * det.im));
    const double dphi = 0.5 * atan2(det.im, det.re);
    double sdn, sdc;
    sincos(dphi, &sdn, &sdc);
    const cplx isq = {sdc / dabs, -sdn / dabs};  // 1 / sqrt(det)
    double S[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const cplx ev = cmul({X[k][k], Y[k][k]}, isq);
        double two_s = atan2(ev.im, ev.re) * 0.31830988618379067154;  // angle / pi
        // "two_S <= -0.5 -> += 2"; the 1e-12 keeps SWAP-like spectra (all four phases exactly at -pi/2, where
        // the reference's own result depends on LAPACK's last bit) on one side of the knife edge
        if (two_s <= -0.5 + 1e-12) two_s += 2.0;
        S[k] = 0.5 * two_s;
    }
    // sort descending (5-comparator network)
#define SLAM_CSWAP(a, b) { const double hi_ = fmax(S[a], S[b]), lo_ = fmin(S[a], S[b]); S[a] = hi_; S[b] = lo_; }
    SLAM_CSWAP(0, 1) SLAM_CSWAP(2, 3) SLAM_CSWAP(0, 2) SLAM_CSWAP(1, 3) SLAM_CSWAP(1, 2)
#undef SLAM_CSWAP
    int n = (int)rint(S[0] + S[1] + S[2] + S[3]);
    n = n < 0 ? 0 : (n > 3 ? 3 : n);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n) S[i] -= 1.0;
    double R[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) R[i] = S[(i + n) & 3];  // numpy.roll(S, -n)[:3]
    double c1 = R[0] + R[1], c2 = R[0] + R[2], c3 = R[1] + R[2];
    if (c3 < 0.0) {
        c1 = 1.0 - c1;
        c3 = -c3;
    }
    if (ndigits >= 0) {
        double scale = 1.0;
        for (int i = 0; i < ndigits && i < 15; ++i) scale *= 10.0;
        c1 = rint(c1 * scale) / scale;
        c2 = rint(c2 * scale) / scale;
        c3 = rint(c3 * scale) / scale;
    }
    out[0] = c1 + 0.0;
    out[1] = c2 + 0.0;
    out[2] = c3 + 0.0;
}

__global__ void c1c2c3_kernel(const double* __restrict__ U, int64_t M, int ndigits, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) weyl_c1c2c3(U + i * 32, ndigits, out + i * 3);
}

// ---------------------------------------------------------------------------------
// Span predictor (replaces monodromy_range_from_target, src/slam/utils/polytopes/polytope_wrap.py:39-94, for resident targets): the
// smallest number of leading gates of a template whose coverage set contains the target.  The host (coverage.py) turns each prefix
// g_1 .. g_k into 14 half-spaces in the target's alcove coordinates -- the inequalities of the multiplicative eigenvalue problem for
// SU(4) collapse to one bound per subset K -- and this kernel evaluates them: Weyl coordinates of the target (weyl_c1c2c3, 8 digits as
// on the host), both alcove points of its class, the 14 order-statistic sums, compare.  One thread per target, HBM-bound on the 256 B
// of the target.
// ---------------------------------------------------------------------------------
constexpr int kSpanPatterns = 14;
struct SpanRegions {
    int32_t k_max;                 // 1 .. SLAM_MAX_SPAN_EVAL
    double tol;                    // widens (> 0) the regions, alcove units (= units of pi)
    double point[4];               // k = 1: the alcove point of the first gate
    double bounds[16][kSpanPatterns];  // [SLAM_MAX_SPAN_EVAL]; k = 2 .. : bounds[k - 1][p] <= sum_p(gamma)  (-inf: no constraint)
};

// alcove point (decreasing, sum 0, a_1 - a_4 <= 1) of i^{2 shift} CAN(c1, c2, c3): coverage._alcove_columns
__device__ inline void alcove_point(double c1, double c2, double c3, double shift, double (&a)[4]) {
    double v[4] = {0.5 * (c1 + c2 - c3) + shift, 0.5 * (c1 - c2 + c3) + shift, 0.5 * (-c1 + c2 + c3) + shift, 0.5 * (-c1 - c2 - c3) + shift};
    for (int j = 0; j < 4; ++j) v[j] -= floor(v[j]);
    double t;
#define SLAM_CE(i, j) { t = fmax(v[i], v[j]); v[j] = fmin(v[i], v[j]); v[i] = t; }
    SLAM_CE(0, 1) SLAM_CE(2, 3) SLAM_CE(0, 2) SLAM_CE(1, 3) SLAM_CE(1, 2)
#undef SLAM_CE
    const int s = (int)rint(v[0] + v[1] + v[2] + v[3]);
    const double e[7] = {v[0], v[1], v[2], v[3], v[0] - 1.0, v[1] - 1.0, v[2] - 1.0};
    for (int j = 0; j < 4; ++j) a[j] = e[j + s];
}

__global__ void span_predict_kernel(const double* __restrict__ U, int64_t M, SpanRegions r, int32_t* __restrict__ spans) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    double c[3];
    weyl_c1c2c3(U + i * 32, 8, c);
    int best = r.k_max + 1;
    bool local = false;
    for (int sh = 0; sh < 2; ++sh) {
        double a[4];
        alcove_point(c[0], c[1], c[2], sh ? 0.5 : 0.0, a);
        local = local || (fabs(a[0]) <= 1e-8 && fabs(a[3]) <= 1e-8);
        // sums over the subsets K (coverage._PATTERNS order): gamma_{5 - k}, k in K  ->  zero-based a[4 - k]
        const double g1 = a[3], g2 = a[2], g3 = a[1], g4 = a[0];  // g_k = gamma_{5 - k}
        const double sums[kSpanPatterns] = {g1, g2, g3, g4, g1 + g2, g1 + g3, g1 + g4, g2 + g3, g2 + g4, g3 + g4,
                                            g1 + g2 + g3, g1 + g2 + g4, g1 + g3 + g4, g2 + g3 + g4};
        const double t1 = (r.tol > 0.0 ? r.tol : 0.0) + 1e-12;
        if (fabs(a[0] - r.point[0]) <= t1 && fabs(a[1] - r.point[1]) <= t1 && fabs(a[2] - r.point[2]) <= t1 && fabs(a[3] - r.point[3]) <= t1)
            best = 1;
        for (int k = 2; k <= r.k_max && k < best; ++k) {
            bool ok = true;
            for (int p = 0; p < kSpanPatterns; ++p) ok = ok && (sums[p] >= r.bounds[k - 1][p] - r.tol);
            if (ok) best = k;
        }
    }
    spans[i] = local ? 0 : best;
}

}  // namespace slamdev
