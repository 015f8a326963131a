// slam_long.hpp -- templates of 6 .. 16 two-qubit gates: ONE WAVEFRONT per (target, restart) item (gfx950 only).
//
// The quad kernels of slam_kernels.hpp keep an item's n x n quasi-Newton metric in the registers of four lanes; at five gates
// (n = 36) that fills the register file.  The reference's own use of MixedOrderBasisCircuitTemplate (src/slam/basis.py:213-359,
// scripts/haar_improvements.ipynb) builds circuits of 12 .. 26 weak gates, n = 6 (k + 1) up to 102 and beyond.  Here the work
// decomposition turns round:
//   * a wavefront owns one item; QUAD j (lanes 4j .. 4j + 3) owns LAYER j of the template, lane c of the quad column c (forward) and
//     row c (backward) of that layer's 4x4 matrices -- 16 layers at once, a 17th (k = 16) in a second pass of quad 0;
//   * W = M_L-1 ... M_1 M_0, M_0 = K_0, M_j = K_j G_j.  Every quad builds its M_j, then all prefix products Pre_j = M_j ... M_0 and all
//     suffix products Q_j = M_L-1 ... M_j come from a parallel SCAN over the quads (Hillis-Steele: 4 steps of 4x4 complex products,
//     exchanged through the wave's LDS) instead of a chain of 16 dependent layers;
//   * the loss needs Tr(T^+ Pre_L-1); the six derivatives of layer j need row c of Suf_j = z T^+ Q_j+1 and column c of Pre_j -- the
//     quad computes them with the same U3 derivative formulas as eval_quad (slam_device.hpp) and the gradient goes back through LDS;
//   * the optimizer is the SAME quasi-Newton iteration as minimize_body (oracle/bfgs_port.py is its NumPy form): BFGS with the inverse
//     Hessian approximation in fp32 -- here an n x n matrix per wavefront in DEVICE MEMORY (at most 102 x 128 floats = 52 KB per
//     wavefront; the slices of the resident wavefronts sit in L2 / Infinity Cache), lane l owning columns 2l and 2l + 1 --, one
//     matrix-vector product and one rank-2 update per accepted step, Armijo backtracking from a first step capped at 2 rad, cautious
//     update (steps violating the weak-Wolfe curvature condition do not touch the metric, the next first step grows), periodic restart,
//     the same stopping tests.  (A first version kept 8 L-BFGS pairs in LDS: on an 8-gate template of a weak gate SciPy's L-BFGS-B with
//     8 pairs needs 3.4 x the evaluations of its BFGS -- mean 621 against 183, maximum 1788 against 396 --, and one item per wavefront
//     turns every straggler into stage time.)  Control flow is wave-uniform: one item per wavefront, no lock-step state machine.
// Loss, gradient, parameters, steps: fp64.
//
// Reference behaviour: CircuitTemplate.eval src/slam/basis.py:102-104,124-169; BasicCost src/slam/cost_function.py:140-145; the
// restart loop src/slam/optimizer.py:253-295.
#pragma once
#include "slam_kernels.hpp"

namespace slamdev {

constexpr int kLongMaxSpan = 16;
constexpr int kLongMaxLayers = kLongMaxSpan + 1;  // 17: quads 0..15 in pass 0, quad 0 again in pass 1
constexpr int kLongN = 6 * kLongMaxLayers;        // 102 parameters at most
constexpr int kLongNP = 128;                      // vector length in LDS: component i = 2 lane + a, a = 0, 1
constexpr int kLongSlots = 2;                     // parameter slots per lane (an adjacent pair: one Philox block, one float2 of a metric row)
constexpr int kLongHStride = kLongNP;             // floats per row of the inverse Hessian in device memory
constexpr int kLongRowBatch = 16;                 // rows of it requested together (mat-vec / update passes)

// LDS of one wavefront, in doubles
constexpr int kLongOffTbl = 0;                                  // sincos table (64 double2)
constexpr int kLongOffX = kLongOffTbl + kSincosLdsDoubles;      // trial point x[NP]
constexpr int kLongOffG = kLongOffX + kLongNP;                  // gradient g[NP]
constexpr int kLongOffTrig = kLongOffG + kLongNP;               // (cos, sin) of the 6 angles of every layer: [L][6] double2
constexpr int kLongOffPre = kLongOffTrig + 12 * kLongMaxLayers; // Pre_j, column-major 4x4 complex: [L][16] double2
constexpr int kLongOffQ = kLongOffPre + 32 * kLongMaxLayers;    // Q_j
constexpr int kLongOffF32 = kLongOffQ + 32 * kLongMaxLayers;    // three fp32 vectors [NP]: broadcast operands of the metric's mat-vec / update
constexpr int kLongLdsDoubles = kLongOffF32 + 3 * kLongNP / 2;
constexpr size_t kLongLdsBytes = (size_t)kLongLdsDoubles * sizeof(double);
static_assert(8 * kLongLdsBytes <= 160 * 1024, "two wavefronts per SIMD need eight wavefronts' LDS per CU");

struct LongArgs {
    const double* targets;    // [n_active][32]: target of stage slot s
    const int32_t* orig;      // original target index of slot s, or nullptr = first_target + s
    int32_t first_target;
    const double* x0;         // [M][n] or nullptr (Philox)
    StageCtl* ctl;
    int32_t restarts;
    int32_t maxiter;
    double gtol, stop_loss, gtol_far, far_loss, exit_loss;
    uint64_t seed;
    int64_t target_base;
    uint32_t flags;           // SLAM_FLAG_EARLY_EXIT | SLAM_FLAG_ORDERED | kFlagNoExterior
    int32_t cost_kind;
    int32_t* solved;
    ItemRec* item_rec;        // [M]
    double* item_x;           // [M][n]
    const double* gates;      // [k][32]
    int32_t k;
    float* hmem;              // [gridDim.x][n][kLongHStride]: the wavefronts' inverse Hessian approximations
    // optional per-iteration trace (use_callback, optimizer.py:217-224), as in MinimizeArgs: after accepted step number it >= 1 of item m,
    // trace_loss[m][it - 1] = loss and trace_x[m][it - 1][:] = parameters; nullptr = off
    double* trace_loss;       // [M][trace_cap]
    double* trace_x;          // [M][trace_cap][n]
    int32_t trace_cap;
};

struct LongEvalArgs {
    const double* targets;
    const double* x;          // [M][n]
    const int32_t* target_of; // [M]
    int64_t n_items;
    double* loss;
    double* grad;             // [M][n] or nullptr
    double* unitary;          // [M][32] or nullptr
    int32_t cost_kind;
    const double* gates;      // [k][32]
    int32_t k;
};

__device__ __forceinline__ double long_readlane(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, result in every lane.  quad_perm x 2, row_half_mirror, row_mirror: every lane holds its 16-lane row's
// sum; the four rows meet through scalar registers
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return (long_readlane(v, 0) + long_readlane(v, 16)) + (long_readlane(v, 32) + long_readlane(v, 48));
}
__device__ __forceinline__ double wave_max_abs(double v) {  // v >= 0
    v = max_abs(v, dpp_f64<0xB1>(v));
    v = max_abs(v, dpp_f64<0x4E>(v));
    v = max_abs(v, dpp_f64<0x141>(v));
    v = max_abs(v, dpp_f64<0x140>(v));
    return max_abs(max_abs(long_readlane(v, 0), long_readlane(v, 16)), max_abs(long_readlane(v, 32), long_readlane(v, 48)));
}

// (f0, f1)^T <- U3^+ (f0, f1)^T,  U3^+ = D_lam^* R^T D_phi^*
__device__ __forceinline__ void u3_col_inv(const U3t& t, double& f0r, double& f0i, double& f1r, double& f1i) {
    const double g1r = t.cp * f1r + t.sp * f1i;  // e^{-i phi} f1
    const double g1i = t.cp * f1i - t.sp * f1r;
    const double y0r = t.c * f0r + t.s * g1r;
    const double y0i = t.c * f0i + t.s * g1i;
    const double tr = t.c * g1r - t.s * f0r;
    const double ti = t.c * g1i - t.s * f0i;
    f0r = y0r; f0i = y0i;
    f1r = t.cl * tr + t.sl * ti;  // e^{-i lam} t
    f1i = t.cl * ti - t.sl * tr;
}

// C[:, c] = A B[:, c]: A = the full 4x4 at `a` (column-major double2[16]), b = column c of B
__device__ __forceinline__ void mat_col(const double2* a, const double (&br)[4], const double (&bi)[4], double (&cr)[4], double (&ci)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { cr[r] = 0.0; ci[r] = 0.0; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double2 e = a[s * 4 + r];  // A[r][s]
            cr[r] = fma(e.x, br[s], fma(-e.y, bi[s], cr[r]));
            ci[r] = fma(e.x, bi[s], fma(e.y, br[s], ci[r]));
        }
    }
}
// C[c, :] = A[c, :] B: a = row c of A, B = the full 4x4 at `b`
__device__ __forceinline__ void row_mat(const double (&ar)[4], const double (&ai)[4], const double2* b, double (&cr)[4], double (&ci)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) { cr[s] = 0.0; ci[s] = 0.0; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double2 e = b[s * 4 + r];  // B[r][s]
            cr[s] = fma(ar[r], e.x, fma(-ai[r], e.y, cr[s]));
            ci[s] = fma(ar[r], e.y, fma(ai[r], e.x, ci[s]));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Fused loss + gradient of one item by the whole wavefront.  In: the trial point in lds[kLongOffX ..], the target column of this lane
// (tcol = T[0][c]), the gate table.  Out: the loss (wave-uniform), the gradient in lds[kLongOffG ..], W = Pre_{L-1} in LDS.
// `pin_exterior`: layers 0 and k contribute no gradient (SLAM_FLAG_NO_EXTERIOR).
// ---------------------------------------------------------------------------------------------------------------------------
// The item's target column and the lane's gate columns (column c of G_j for the layers j = quad, quad + 16 this lane's quad owns: constant
// for the whole launch) are loaded by the caller ONCE -- per evaluation they were two rounds of global loads with two wavefronts per SIMD
// to hide them.
struct LongGateCols {
    double r[2][4], i[2][4];
};
__device__ __forceinline__ LongGateCols load_gate_cols(const double* gates, int k) {
    LongGateCols g;
    const int c = threadIdx.x & 3, quad = threadIdx.x >> 2;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int j = quad + 16 * ps;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double2 e = make_double2(r == c ? 1.0 : 0.0, 0.0);  // G_0 = 1
            if (j >= 1 && j <= k) e = *reinterpret_cast<const double2*>(gates + 32 * (j - 1) + (r * 4 + c) * 2);
            g.r[ps][r] = e.x;
            g.i[ps][r] = e.y;
        }
    }
    return g;
}
template <bool HUGE_ARGS>
__device__ __forceinline__ double eval_long(double* lds, const double (&tre)[4], const double (&tim)[4], const LongGateCols& gcol, int k, int cost_kind, bool pin_exterior) {
    const int lane = threadIdx.x;
    const int c = lane & 3;
    const int quad = lane >> 2;
    const int L = k + 1;
    const double2* tbl = reinterpret_cast<const double2*>(lds + kLongOffTbl);
    const double* xs = lds + kLongOffX;
    double* gs = lds + kLongOffG;
    double2* trig = reinterpret_cast<double2*>(lds + kLongOffTrig);
    double2* Pre = reinterpret_cast<double2*>(lds + kLongOffPre);
    double2* Q = reinterpret_cast<double2*>(lds + kLongOffQ);
    const int npass = L > 16 ? 2 : 1;

    // ---- 1. trig entries: lane c of quad j handles parameters c and c + 4 of layer j
    for (int ps = 0; ps < npass; ++ps) {
        const int j = quad + 16 * ps;
        if (j < L) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = c + 4 * h;
                if (m < 6) {
                    const double xv = xs[6 * j + m];
                    const double arg = (m == 0 || m == 3) ? 0.5 * xv : xv;
                    double s, co;
                    if constexpr (HUGE_ARGS) sincos_any(arg, tbl, s, co);
                    else sincos_tbl(arg, tbl, s, co);  // (the optimizer's angles stay far below the table path's limit)
                    trig[6 * j + m] = make_double2(co, s);
                }
            }
        }
    }
    lds_fence();
    // ---- 2. M_j = K_j G_j (G_0 = 1): column c into Pre_j and Q_j
    for (int ps = 0; ps < npass; ++ps) {
        const int j = quad + 16 * ps;
        if (j < L) {
            const U3t B = load_u3(reinterpret_cast<const double*>(trig), 6 * j), A = load_u3(reinterpret_cast<const double*>(trig), 6 * j + 3);
            double Fr[4], Fi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Fr[r] = ps == 0 ? gcol.r[0][r] : gcol.r[1][r];
                Fi[r] = ps == 0 ? gcol.i[0][r] : gcol.i[1][r];
            }
            u3_col(B, Fr[0], Fi[0], Fr[1], Fi[1]);
            u3_col(B, Fr[2], Fi[2], Fr[3], Fi[3]);
            u3_col(A, Fr[0], Fi[0], Fr[2], Fi[2]);
            u3_col(A, Fr[1], Fi[1], Fr[3], Fi[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Pre[j * 16 + c * 4 + r] = make_double2(Fr[r], Fi[r]);
                Q[j * 16 + c * 4 + r] = make_double2(Fr[r], Fi[r]);
            }
        }
    }
    lds_fence();
    // ---- 3. scans inside pass 0 (layers 0..15): Pre_j <- Pre_j Pre_{j-s}, Q_j <- Q_{j+s} Q_j, s = 1, 2, 4, 8
    {
        const int j = quad;
        const int top = L < 16 ? L : 16;  // layers of pass 0
#pragma unroll 1
        for (int s = 1; s < top; s <<= 1) {  // (three levels up to 8 layers, four beyond)
            const bool dp = j < top && j >= s;
            const bool dq = j + s < top;
            double pr[4], pi[4], qr[4], qi[4];
            if (dp) {
                double br[4], bi[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 e = Pre[(j - s) * 16 + c * 4 + r];
                    br[r] = e.x; bi[r] = e.y;
                }
                mat_col(Pre + j * 16, br, bi, pr, pi);
            }
            if (dq) {
                double ar[4], ai[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 e = Q[(j + s) * 16 + r * 4 + c];  // row c of Q_{j+s}
                    ar[r] = e.x; ai[r] = e.y;
                }
                row_mat(ar, ai, Q + j * 16, qr, qi);
            }
            lds_fence();
            if (dp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Pre[j * 16 + c * 4 + r] = make_double2(pr[r], pi[r]);
            }
            if (dq) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Q[j * 16 + r * 4 + c] = make_double2(qr[r], qi[r]);
            }
            lds_fence();
        }
        if (npass == 2) {
            // layer 16 (one layer in pass 1): Pre_16 = M_16 Pre_15; Q_j <- M_16 Q_j for j < 16 (Q_16 = M_16 as stored)
            double pr[4], pi[4], qr[4], qi[4];
            const bool dp = quad == 0;
            if (dp) {
                double br[4], bi[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 e = Pre[15 * 16 + c * 4 + r];
                    br[r] = e.x; bi[r] = e.y;
                }
                mat_col(Pre + 16 * 16, br, bi, pr, pi);
            }
            {
                double ar[4], ai[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 e = Q[16 * 16 + r * 4 + c];
                    ar[r] = e.x; ai[r] = e.y;
                }
                row_mat(ar, ai, Q + j * 16, qr, qi);
            }
            lds_fence();
            if (dp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Pre[16 * 16 + c * 4 + r] = make_double2(pr[r], pi[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Q[j * 16 + r * 4 + c] = make_double2(qr[r], qi[r]);
            lds_fence();
        }
    }
    // ---- 4. t = Tr(T^+ W), W = Pre_{L-1}: every quad computes it (no broadcast needed)
    double pr = 0.0, pi = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double2 w = Pre[(L - 1) * 16 + c * 4 + r];
        pr = fma(tre[r], w.x, fma(tim[r], w.y, pr));
        pi = fma(tre[r], w.y, fma(-tim[r], w.x, pi));
    }
    quad_sum2(pr, pi);
    const double at2 = pr * pr + pi * pi;
    const double rat = (at2 > 1e-300) ? fast_rsqrt(at2) : 0.0;
    const double at = at2 * rat;
    const double basic = 1.0 - 0.25 * at;  // BasicCost, cost_function.py:140-145
    const bool sq = (cost_kind == 1);      // SquareCost = 0.8 (2 L - L^2) of BasicCost L (cost_function.py:169-173)
    const double c0 = sq ? 1.6 : 1.0, c1 = sq ? -0.8 : 0.0, d1 = sq ? -1.6 : 0.0;
    const double fout = basic * fma(c1, basic, c0);
    const double inv = (0.25 * rat) * fma(d1, basic, c0);
    const double zr = -pr * inv, zi = pi * inv;
    // row c of Z = z T^+
    double Zr[4], Zi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Zr[r] = zr * tre[r] + zi * tim[r];
        Zi[r] = zi * tre[r] - zr * tim[r];
    }
    // ---- 5. the six derivatives of every layer
    for (int ps = 0; ps < npass; ++ps) {
        const int j = quad + 16 * ps;
        if (j < L) {
            const U3t B = load_u3(reinterpret_cast<const double*>(trig), 6 * j), A = load_u3(reinterpret_cast<const double*>(trig), 6 * j + 3);
            double Ur[4], Ui[4];
            if (j == L - 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { Ur[r] = Zr[r]; Ui[r] = Zi[r]; }
            } else {
                row_mat(Zr, Zi, Q + (j + 1) * 16, Ur, Ui);  // row c of Suf_j = Z Q_{j+1}
            }
            double Hr[4], Hi[4], fr[4], fi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = Pre[j * 16 + c * 4 + r];
                Hr[r] = e.x; Hi[r] = e.y;
                fr[r] = e.x; fi[r] = e.y;
            }
            // f = input of layer j's 1Q gates = K_j^+ h
            u3_col_inv(A, fr[0], fi[0], fr[2], fi[2]);
            u3_col_inv(A, fr[1], fi[1], fr[3], fi[3]);
            u3_col_inv(B, fr[0], fi[0], fr[1], fi[1]);
            u3_col_inv(B, fr[2], fi[2], fr[3], fi[3]);
            double part[6];
            const double m1 = im_mul(Ur[1], Ui[1], Hr[1], Hi[1]);
            const double m2 = im_mul(Ur[2], Ui[2], Hr[2], Hi[2]);
            const double m3 = im_mul(Ur[3], Ui[3], Hr[3], Hi[3]);
            part[1] = -(m1 + m3);
            part[4] = -(m2 + m3);
            double thA2 = 0.0;
            u3_row_dtheta_first(A, Ur[0], Ui[0], Ur[2], Ui[2], Hr[0], Hi[0], Hr[2], Hi[2], thA2);
            u3_row_dtheta_first(A, Ur[1], Ui[1], Ur[3], Ui[3], Hr[1], Hi[1], Hr[3], Hi[3], thA2);
            double tB01r, tB01i, tB23r, tB23i;
            u3_row_t(B, Ur[0], Ui[0], Ur[1], Ui[1], tB01r, tB01i);
            u3_row_t(B, Ur[2], Ui[2], Ur[3], Ui[3], tB23r, tB23i);
            double th01, la01, th23, la23;
            dtheta_dlam_last(B, Ur[0], Ui[0], tB01r, tB01i, fr[0], fi[0], fr[1], fi[1], th01, la01);
            dtheta_dlam_last(B, Ur[2], Ui[2], tB23r, tB23i, fr[2], fi[2], fr[3], fi[3], th23, la23);
            part[0] = th01 + th23;
            part[2] = la01 + la23;
            part[5] = -(im_mul(Ur[2], Ui[2], fr[2], fi[2]) + im_mul(Ur[3], Ui[3], fr[3], fi[3]));
            part[3] = 0.5 * thA2;
            quad_sum2(part[0], part[1]);
            quad_sum2(part[2], part[3]);
            quad_sum2(part[4], part[5]);
            const bool pinned = pin_exterior && (j == 0 || j == k);
            const double v0 = c == 0 ? part[0] : (c == 1 ? part[1] : (c == 2 ? part[2] : part[3]));
            gs[6 * j + c] = pinned ? 0.0 : v0;
            if (c < 2) gs[6 * j + 4 + c] = pinned ? 0.0 : (c == 0 ? part[4] : part[5]);
        }
    }
    lds_fence();
    return fout;
}

// the wavefront's LDS prologue
__device__ __forceinline__ void long_prologue(double* lds) {
    load_sincos_table(reinterpret_cast<double2*>(lds + kLongOffTbl), threadIdx.x);
    lds_fence();
}

// ---------------------------------------------------------------------------------------------------------------------------
// slam_eval_loss_grad / slam_eval_unitary for 6 .. 16 gates: one item per wavefront
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kWave, 2) eval_long_kernel(LongEvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int n = 6 * (a.k + 1);
    long_prologue(lds);
    const LongGateCols gcol = load_gate_cols(a.gates, a.k);
    for (int64_t item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        const int64_t tgt = a.target_of[item];
#pragma unroll
        for (int s = 0; s < kLongSlots; ++s) {
            const int i = 2 * lane + s;
            lds[kLongOffX + i] = (i < n) ? a.x[item * n + i] : 0.0;
        }
        lds_fence();
        double tre[4], tim[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double2 t = *reinterpret_cast<const double2*>(a.targets + tgt * 32 + (lane & 3) * 2 + 8 * r);
            tre[r] = t.x;
            tim[r] = t.y;
        }
        const double f = eval_long<true>(lds, tre, tim, gcol, a.k, a.cost_kind, false);
        if (lane == 0) a.loss[item] = f;
        if (a.grad) {
#pragma unroll
            for (int s = 0; s < kLongSlots; ++s) {
                const int i = 2 * lane + s;
                if (i < n) a.grad[item * n + i] = lds[kLongOffG + i];
            }
        }
        if (a.unitary && lane < 16) {
            // W = Pre_{L-1}, column-major in LDS -> row-major (re, im) out
            const double2 e = reinterpret_cast<const double2*>(lds + kLongOffPre)[a.k * 16 + lane];  // element (r = lane & 3, col = lane >> 2)
            const int r = lane & 3, cc = lane >> 2;
            a.unitary[item * 32 + (r * 4 + cc) * 2] = e.x;
            a.unitary[item * 32 + (r * 4 + cc) * 2 + 1] = e.y;
        }
        lds_fence();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Quasi-Newton minimisation (the iteration of minimize_body / oracle/bfgs_port.py), one item per wavefront, persistent wavefronts
// pulling (restart-major) queue positions.  Lane l holds components 2l and 2l + 1 of x, g, p; the fp32 inverse Hessian approximation
// H (n rows of kLongHStride floats in device memory, this wavefront's slice) is read row by row as float2 per lane: since H is
// symmetric, lane l's two COLUMNS give its two components of H v, and a row is one coalesced 8 n-byte access of the wavefront.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kWave, 2) minimize_long_kernel(LongArgs args) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int k = args.k;
    const int n = 6 * (k + 1);
    float* f32a = reinterpret_cast<float*>(lds + kLongOffF32);  // [NP] broadcast vector of the mat-vec: g'
    float* f32b = f32a + kLongNP;                               // [NP] pending update: s
    float* f32c = f32b + kLongNP;                               // [NP] pending update: v
    float2* const Hm = reinterpret_cast<float2*>(args.hmem + (size_t)blockIdx.x * (size_t)n * kLongHStride) + lane;  // + j * 64: row j
    long_prologue(lds);
    const LongGateCols gcol = load_gate_cols(args.gates, k);
    const unsigned n_act = (unsigned)args.ctl->n_active;
    const unsigned n_items = n_act * (unsigned)args.restarts;
    const bool early = args.flags & 1u, ordered = args.flags & 2u, pin = args.flags & kFlagNoExterior;
    unsigned rounds = 0;
    bool valid[kLongSlots];
#pragma unroll
    for (int s = 0; s < kLongSlots; ++s) {
        const int i = 2 * lane + s;
        valid[s] = i < n && !(pin && (i < 6 || i >= 6 * k));
    }
    const bool lane_in = 2 * lane < n;  // this lane holds components of the problem (n is even)

    while (true) {
        unsigned pos = 0;
        if (lane == 0) pos = atomicAdd(&args.ctl->work_counter, 1u);
        pos = (unsigned)__builtin_amdgcn_readfirstlane((int)pos);
        if (pos >= n_items) break;
        const unsigned rs = pos / n_act;         // restart (restart-major queue: every target's restart r before any r + 1)
        const unsigned sl = pos - rs * n_act;    // stage slot
        const unsigned item = sl * (unsigned)args.restarts + rs;
        const int mine = args.restarts - (int)rs;
        if (early) {
            const int fl = __hip_atomic_load(&args.solved[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ordered ? (fl > mine) : (fl != 0)) {  // a sibling restart has already succeeded (ordered: one with a lower index)
                if (lane == 0) item_rec_store_dropped(args.item_rec + item, ST_PREEMPTED);
                continue;
            }
        }
        const int tgt = args.orig ? args.orig[sl] : args.first_target + (int)sl;
        double tre[4], tim[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double2 t = *reinterpret_cast<const double2*>(args.targets + (int64_t)sl * 32 + (lane & 3) * 2 + 8 * r);
            tre[r] = t.x;
            tim[r] = t.y;
        }
        double x[kLongSlots], g[kLongSlots], p[kLongSlots];
        {
            // start point: explicit, or Philox block `lane` = the parameter pair (2 lane, 2 lane + 1) -- the numbers of oracle.x0_philox
            double x0v[2] = {0.0, 0.0};
            if (lane_in) {
                if (args.x0) {
                    x0v[0] = args.x0[(int64_t)item * n + 2 * lane];
                    x0v[1] = args.x0[(int64_t)item * n + 2 * lane + 1];
                } else {
                    uint32_t w[4];
                    philox4x32_10((uint32_t)lane, rs, (uint32_t)(tgt + (int)args.target_base), (uint32_t)k, (uint32_t)args.seed, (uint32_t)(args.seed >> 32), w);
                    x0v[0] = x0_from_words(w[0], w[1]);
                    x0v[1] = x0_from_words(w[2], w[3]);
                }
            }
#pragma unroll
            for (int s = 0; s < kLongSlots; ++s) {
                x[s] = valid[s] ? x0v[s] : 0.0;
                lds[kLongOffX + 2 * lane + s] = x[s];
            }
        }
        lds_fence();
        double f = eval_long<false>(lds, tre, tim, gcol, k, args.cost_kind, pin);
        ++rounds;
        int nev = 1, nacc = 0, iters = 0, nback = 0, nstall = 0, status = ST_MAXITER;
        bool ident = true, scaled = false;  // ident: H is the identity (nothing of it is in memory yet)
        // The rank-2 update of an accepted step is applied by the NEXT step's mat-vec pass (one read + one write of the matrix per
        // accepted step instead of two reads + one write: at 16 gates the passes are bound by the memory system): pend = an update is
        // waiting; its s and v in LDS (f32b, f32c), this lane's components of w and s in registers
        bool pend = false;
        float pw0 = 0.0f, pw1 = 0.0f, ps0 = 0.0f, ps1 = 0.0f;
        double hs1 = 0.0, grow = 1.0;       // the effective metric is H + hs1 I (the one-off scaling of the initial metric)
        double gnorm = 0.0, gp = 0.0, pp = 0.0, alpha = 0.0;
        bool done = false;
        if (!isfinite(f)) {
            status = ST_NONFINITE;
            done = true;
        } else {
            nacc = 1;
            double gg = 0.0, gm = 0.0;
#pragma unroll
            for (int s = 0; s < kLongSlots; ++s) {
                g[s] = valid[s] ? lds[kLongOffG + 2 * lane + s] : 0.0;
                // (the first direction is -H g with H = 1 in fp32, as in the quad kernels and the NumPy port)
                p[s] = -(double)(float)g[s];
                gm = max_abs(gm, g[s]);
            }
#pragma unroll
            for (int s = 0; s < kLongSlots; ++s) gg = fma(g[s], p[s], gg);
            gp = wave_sum(gg);
            double d2 = 0.0;
#pragma unroll
            for (int s = 0; s < kLongSlots; ++s) d2 = fma(p[s], p[s], d2);
            pp = wave_sum(d2);
            gnorm = wave_max_abs(gm);
            alpha = (pp > 1e-300) ? fmin(grow, kStepMax * fast_rsqrt(pp)) : grow;
            if (f < args.stop_loss || gnorm < args.gtol || (gnorm < args.gtol_far && f > args.far_loss)) { status = ST_CONVERGED; done = true; }
            else if (args.maxiter <= 0) { status = ST_MAXITER; done = true; }
        }
        while (!done) {
            // ---- trial point
            double xt[kLongSlots];
#pragma unroll
            for (int s = 0; s < kLongSlots; ++s) {
                xt[s] = fma(alpha, p[s], x[s]);
                lds[kLongOffX + 2 * lane + s] = xt[s];
            }
            lds_fence();
            const double ft = eval_long<false>(lds, tre, tim, gcol, k, args.cost_kind, pin);
            ++rounds;
            ++nev;
            const bool finite = isfinite(ft);
            const bool armijo = finite && (ft <= f + kArmijoC1 * alpha * gp);
            if (armijo) {
                ++nacc;
                double gt[kLongSlots], qv[kLongSlots];
                double pgt = 0.0, yy = 0.0, gm = 0.0;
#pragma unroll
                for (int s = 0; s < kLongSlots; ++s) {
                    gt[s] = valid[s] ? lds[kLongOffG + 2 * lane + s] : 0.0;
                    const double ya = gt[s] - g[s];
                    pgt = fma(p[s], gt[s], pgt);
                    yy = fma(ya, ya, yy);
                    gm = max_abs(gm, gt[s]);
                }
                pgt = wave_sum(pgt);
                yy = wave_sum(yy);
                // ---- q = H g' (fp32), H = the matrix in memory (or the identity) + the pending update: one pass over the rows -- row j is
                // read (float2 per lane: this lane's two columns), updated with s_j w + v_j s, written back, and multiplied by g'_j
                if (ident && !pend) {
#pragma unroll
                    for (int s = 0; s < kLongSlots; ++s) qv[s] = (double)(float)gt[s];
                } else {
                    f32a[2 * lane] = (float)gt[0];
                    f32a[2 * lane + 1] = (float)gt[1];
                    lds_fence();
                    float a0 = 0.0f, a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;  // two accumulator pairs: even / odd rows
                    if (lane_in) {
                        // kLongRowBatch rows requested before the first one is used: a row is one 8 n-byte access of the wavefront, and with
                        // one or two in flight the pass ran at the memory's latency (45 us per iteration at 12 gates)
                        for (int j0 = 0; j0 < n; j0 += kLongRowBatch) {
                            float2 h[kLongRowBatch];
#pragma unroll
                            for (int u = 0; u < kLongRowBatch; ++u) {
                                const int j = j0 + u;
                                if (ident) h[u] = make_float2(j == 2 * lane ? 1.0f : 0.0f, j == 2 * lane + 1 ? 1.0f : 0.0f);
                                else if (j < n) h[u] = Hm[(size_t)j * 64];
                                else h[u] = make_float2(0.0f, 0.0f);
                            }
                            if (pend) {
#pragma unroll
                                for (int u = 0; u < kLongRowBatch; ++u) {
                                    const int j = j0 + u;
                                    const float sj = f32b[j], vj = f32c[j];
                                    h[u].x = fmaf(vj, ps0, h[u].x); h[u].y = fmaf(vj, ps1, h[u].y);
                                    h[u].x = fmaf(sj, pw0, h[u].x); h[u].y = fmaf(sj, pw1, h[u].y);
                                    if (j < n) Hm[(size_t)j * 64] = h[u];
                                }
                            }
#pragma unroll
                            for (int u = 0; u < kLongRowBatch; u += 2) {
                                const float g0 = f32a[j0 + u], g1 = f32a[j0 + u + 1];  // (rows beyond n: g' = 0 there; the vectors have kLongNP entries)
                                a0 = fmaf(h[u].x, g0, a0); a1 = fmaf(h[u].y, g0, a1);
                                b0 = fmaf(h[u + 1].x, g1, b0); b1 = fmaf(h[u + 1].y, g1, b1);
                            }
                        }
                    }
                    if (pend) ident = false;  // the matrix is in memory now
                    pend = false;
                    qv[0] = (double)(a0 + b0);
                    qv[1] = (double)(a1 + b1);
                    lds_fence();
                }
                const double sg = alpha * pgt;
                const double sy = alpha * (pgt - gp);
                const double ss = (alpha * alpha) * pp;
                const bool too_short = sy < (1.0 - kWolfeC2) * alpha * (-gp);  // weak-Wolfe curvature condition violated
                const bool curv = !too_short && sy > 0.0 && (sy * sy > (kCurvEps * kCurvEps) * (ss * yy));
                const bool first = curv && !scaled;
                scaled = scaled || curv;
                // first update of an item: scale the initial metric (the identity) by s.y / y.y -- as the scalar hs1
                const double fac = first ? (sy * fast_rcp(yy)) : 1.0;
                hs1 = first ? fac - 1.0 : hs1;
                double yu = 0.0;
#pragma unroll
                for (int s = 0; s < kLongSlots; ++s) {
                    qv[s] = fma(hs1, gt[s], qv[s]);  // q = H_eff g'
                    const double ua = fma(fac, p[s], qv[s]);  // u = H_eff y = q + fac p   (p = -H_eff g before this round's scaling)
                    yu = fma(gt[s] - g[s], ua, yu);
                }
                yu = wave_sum(yu);
                const double rho = curv ? fast_rcp(sy) : 0.0;
                const double cf = rho * (1.0 + rho * yu);
                double wg = 0.0;
                double sa[kLongSlots], wa[kLongSlots], va[kLongSlots];
#pragma unroll
                for (int s = 0; s < kLongSlots; ++s) {
                    sa[s] = alpha * p[s];
                    const double ua = fma(fac, p[s], qv[s]);
                    wa[s] = cf * sa[s] - rho * ua;
                    va[s] = -rho * ua;
                    wg = fma(wa[s], gt[s], wg);
                }
                wg = wave_sum(wg);
                // ---- H += s w^T + v s^T (fp32; row j gets s_j w + v_j s): left pending for the next accepted step's pass
                if (curv) {
                    f32b[2 * lane] = (float)sa[0];
                    f32b[2 * lane + 1] = (float)sa[1];
                    f32c[2 * lane] = (float)va[0];
                    f32c[2 * lane + 1] = (float)va[1];
                    pw0 = (float)wa[0]; pw1 = (float)wa[1];
                    ps0 = (float)sa[0]; ps1 = (float)sa[1];
                    pend = true;
                    lds_fence();
                }
                nstall = (f - ft <= kStallDf) ? nstall + 1 : 0;
                f = ft;
                ++iters;
                nback = 0;
                grow = too_short ? fmin(grow * kGrowFactor, kGrowMax) : 1.0;
                double d1 = 0.0, d2 = 0.0;
#pragma unroll
                for (int s = 0; s < kLongSlots; ++s) {
                    x[s] = xt[s];
                    g[s] = gt[s];
                    p[s] = -(qv[s] + sa[s] * wg + va[s] * sg);
                    d1 = fma(g[s], p[s], d1);
                    d2 = fma(p[s], p[s], d2);
                }
                if (args.trace_loss && iters <= args.trace_cap) {  // (wave-uniform; nothing when off)
                    const int64_t row = (int64_t)item * args.trace_cap + (iters - 1);
                    if (lane == 0) args.trace_loss[row] = f;
#pragma unroll
                    for (int s = 0; s < kLongSlots; ++s)
                        if (valid[s]) args.trace_x[row * n + 2 * lane + s] = x[s];
                }
                gnorm = wave_max_abs(gm);
                gp = wave_sum(d1);
                pp = wave_sum(d2);
                alpha = (pp > 1e-300) ? fmin(grow, kStepMax * fast_rsqrt(pp)) : grow;
                if (f < args.stop_loss || gnorm < args.gtol || (gnorm < args.gtol_far && f > args.far_loss)) { status = ST_CONVERGED; done = true; }
                else if (nstall >= 2) { status = ST_STALLED; done = true; }
                else if (iters >= args.maxiter) { status = ST_MAXITER; done = true; }
                // not a descent direction (H lost positive definiteness numerically), or the periodic restart: steepest descent again
                const bool periodic = !done && ((iters & (kRestartPeriod - 1)) == 0);
                if (!done && (!(gp < 0.0) || periodic)) {
                    ident = true;
                    pend = false;
                    hs1 = 0.0;
                    scaled = periodic ? false : scaled;
                    double gg = 0.0;
#pragma unroll
                    for (int s = 0; s < kLongSlots; ++s) {
                        p[s] = -g[s];
                        gg = fma(g[s], g[s], gg);
                    }
                    gg = wave_sum(gg);
                    gp = -gg;
                    pp = gg;
                    alpha = periodic ? ((gg > 1e-300) ? fmin(grow, kStepMax * fast_rsqrt(gg)) : grow) : alpha;
                }
            } else {
                // safeguarded quadratic interpolation backtrack
                const double denom = 2.0 * (ft - f - gp * alpha);
                const double anew = (finite && denom > 0.0 && isfinite(denom)) ? (-gp * alpha * alpha * fast_rcp(denom)) : 0.5 * alpha;
                alpha = fmin(fmax(anew, 0.1 * alpha), 0.5 * alpha);
                grow = 1.0;
                ++nback;
                if (nback > kMaxBacktrack) { status = (gnorm < kStallGnorm) ? ST_STALLED : ST_LINESEARCH; done = true; }
            }
            // ---- early exit across the restarts of one target (optimizer.py:287-295)
            if (early && !done) {
                const int fl = __hip_atomic_load(&args.solved[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ordered ? (fl > mine) : (fl != 0)) { status = ST_PREEMPTED; done = true; }
            }
        }
        if (early && status != ST_PREEMPTED && f < args.exit_loss && lane == 0)
            __hip_atomic_fetch_max(&args.solved[sl], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) item_rec_store(args.item_rec + item, f, iters, status, nev, nacc);
#pragma unroll
        for (int s = 0; s < kLongSlots; ++s) {
            const int i = 2 * lane + s;
            if (i < n) args.item_x[(int64_t)item * n + i] = x[s];
        }
    }
    if (lane == 0 && rounds) atomicAdd(&args.ctl->rounds, (unsigned long long)rounds);
}

}  // namespace slamdev
