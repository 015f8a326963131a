// slam_v2.hpp -- templates whose 2Q gates carry their own optimisable parameters (gfx950 only).
//
// Reference: CircuitTemplateV2 (src/slam/basisv2.py:27-299): base_gates are gate CLASSES / lambdas; every gate instance of
// the circuit gets its own "Q" parameters next to the "P" parameters of the U gates, optionally box-bounded
// (add_bound, basisv2.py:174-190 -> SciPy L-BFGS-B, src/slam/optimizer.py:255-268).  The gates the reference's callers
// pass are members of the conversion-gain family (RiSwapGate(alpha) = CG(0, 0, -pi alpha / 2, 0, 1);
// ConversionGainGate(phi_c, phi_g, gc, gg, t), custom_gates.py:163-212, 534-606; parallel_drive_volume.py:91-96):
//     G(a, phi_c, b, phi_g):   {|01>,|10>} block [[cos a, -conj(w)], [w, cos a]],  w  = -i e^{i phi_c} sin a,  a = gc t
//                              {|00>,|11>} block [[cos b, -conj(w')], [w', cos b]], w' = -i e^{i phi_g} sin b,  b = gg t
// (closed form of exp(-i t H), src/slam/hamiltonian.py:84-111).  Each of the four raw angles of a gate instance is an
// affine function of at most one of the gate's QN parameters: raw[r] = scale[r] * q[sel[r]] + offset[r]  (sel = -1:
// constant); the host derives that map from the Python callable (basisv2.py).
//
// Parameter vector of a span-k template, index order: P0 .. P{6(k+1)-1} (as in slam_device.hpp), then the QN
// parameters of gate 1, of gate 2, ...  n = 6 (k + 1) + QN k.
//
// Kernels: the same work decomposition as the fixed-gate path (a quad of lanes per (target, seed) item, lane c owns
// column c of the running product and row c of the backward vector, fp32 packed inverse Hessian in registers), in a
// plain form -- static item assignment, both layer inputs and outputs stored, per-item gate entries read from LDS --
// plus the gradient with respect to the gate angles and a projected quasi-Newton step for the box bounds.  This path is
// a first correct one, measured but not tuned like minimize_kernel<K>.
#pragma once
#include "slam_device.hpp"
#include "slam_kernels.hpp"

namespace slamdev {

struct V2GateMap {  // raw angle order: 0 a, 1 phi_c, 2 b, 3 phi_g
    double scale[4];
    double offset[4];
    int32_t sel[4];
    int32_t pad[4];
};
static_assert(sizeof(V2GateMap) == 96, "V2GateMap layout");

template <int K, int QN>
struct CfgV2 {
    static constexpr int NP = 6 * (K + 1);
    static constexpr int NQ = QN * K;
    static constexpr int N = NP + NQ;
    static constexpr int NA = (N + 3) / 4;
    // per-quad LDS area (doubles): trig of the P parameters [2 NP], gate trig [8 K], trial Q values [NQ -> padded],
    // summed 1Q partials [NP], summed raw-angle partials [4 K]; the fp32 mat-vec / update exchanges of slam_device.hpp
    // (10 NA - 8 doubles) reuse the front of it between evaluations
    static constexpr int OFF_GTRIG = 2 * NP;
    static constexpr int OFF_QVAL = OFF_GTRIG + 8 * K;
    static constexpr int OFF_GP = OFF_QVAL + ((NQ + 1) / 2) * 2;
    static constexpr int OFF_DQ = OFF_GP + NP;
    static constexpr int XNEED0 = OFF_DQ + 4 * K;
    static constexpr int XNEED = XNEED0 > 10 * NA - 8 ? XNEED0 : 10 * NA - 8;
    static constexpr int XSTRIDE = (XNEED - 8 + 15) / 16 * 16 + 8;
    static constexpr int LDS_XCHG = kQuadsPerWave * XSTRIDE;
    static constexpr int LDS_FH = 2 * K * 4 * kRow * 2;          // f_1..f_K and h_0..h_{K-1}: [vector][row][lane + pad] double2
    static constexpr int LDS_BOUNDS = 2 * NA * 4;                // (lo, hi) per parameter, padded to 4 NA
    static constexpr int LDS_DOUBLES = LDS_XCHG + LDS_FH + LDS_BOUNDS + kSincosLdsDoubles;
};

// ---- conversion-gain gate actions from the (cos, sin) table entries of the four raw angles -------------------------
struct CGt {
    double ca, sa, cpc, spc, cb, sb, cpg, spg;
};
__device__ __forceinline__ CGt load_cg(const double* gt) {
    const double2* t = reinterpret_cast<const double2*>(gt);
    const double2 a = t[0], pc = t[1], b = t[2], pg = t[3];
    return CGt{a.x, a.y, pc.x, pc.y, b.x, b.y, pg.x, pg.y};
}
// (x_lo, x_hi) <- [[c, -conj(w)], [w, c]] (x_lo, x_hi),  w = (wr, wi)
__device__ __forceinline__ void cg_block_col(double c, double wr, double wi, double& lr, double& li, double& hr, double& hi) {
    const double nlr = c * lr - (wr * hr + wi * hi);  // -conj(w) x_hi = -(wr - i wi)(hr + i hi)
    const double nli = c * li - (wr * hi - wi * hr);
    const double nhr = (wr * lr - wi * li) + c * hr;
    const double nhi = (wr * li + wi * lr) + c * hi;
    lr = nlr; li = nli; hr = nhr; hi = nhi;
}
// (u_lo, u_hi) <- (u_lo, u_hi) [[c, -conj(w)], [w, c]]
__device__ __forceinline__ void cg_block_row(double c, double wr, double wi, double& lr, double& li, double& hr, double& hi) {
    const double nlr = c * lr + (wr * hr - wi * hi);
    const double nli = c * li + (wr * hi + wi * hr);
    const double nhr = c * hr - (wr * lr + wi * li);  // u_lo (-conj(w))
    const double nhi = c * hi - (wr * li - wi * lr);
    lr = nlr; li = nli; hr = nhr; hi = nhi;
}
__device__ __forceinline__ void cg_col(const CGt& g, double (&Fr)[4], double (&Fi)[4]) {
    cg_block_col(g.ca, g.sa * g.spc, -g.sa * g.cpc, Fr[1], Fi[1], Fr[2], Fi[2]);  // w = -i e^{i phi} s = s (sin phi - i cos phi)
    cg_block_col(g.cb, g.sb * g.spg, -g.sb * g.cpg, Fr[0], Fi[0], Fr[3], Fi[3]);
}
__device__ __forceinline__ void cg_row(const CGt& g, double (&Ur)[4], double (&Ui)[4]) {
    cg_block_row(g.ca, g.sa * g.spc, -g.sa * g.cpc, Ur[1], Ui[1], Ur[2], Ui[2]);
    cg_block_row(g.cb, g.sb * g.spg, -g.sb * g.cpg, Ur[0], Ui[0], Ur[3], Ui[3]);
}
// Re( u_lo v_lo + u_hi v_hi ) with (v_lo, v_hi) = [[d, -conj(e)], [e, d]] (h_lo, h_hi): the derivative of a block is
// again of that form -- d/d(angle): d = -sin, e = dw/d(angle) = -i e^{i phi} cos;  d/d(phi): d = 0, e = i w
__device__ __forceinline__ double cg_block_dot(double d, double er, double ei, double ulr, double uli, double uhr, double uhi,
                                               double hlr, double hli, double hhr, double hhi) {
    const double vlr = d * hlr - (er * hhr + ei * hhi), vli = d * hli - (er * hhi - ei * hhr);
    const double vhr = (er * hlr - ei * hli) + d * hhr, vhi = (er * hli + ei * hlr) + d * hhi;
    return (ulr * vlr - uli * vli) + (uhr * vhr - uhi * vhi);
}

// ---------------------------------------------------------------------------------------------------------------
// loss + gradient with respect to all n parameters for the quad's item
//   xd     this lane's parameter slots (component 4a + q)
//   maps   gate maps of G_1..G_K (wave-uniform, global memory)
// ---------------------------------------------------------------------------------------------------------------
template <int K, int QN>
__device__ __forceinline__ void eval_quad_v2(const double (&xd)[CfgV2<K, QN>::NA], const double* tcol, const V2GateMap* maps, double* xq,
                                             double2* fh, const double2* tbl, int q, int cost_kind, double& fout,
                                             double (&gd)[CfgV2<K, QN>::NA], double (&Wr)[4], double (&Wi)[4]) {
    using C = CfgV2<K, QN>;
    double tre[4], tim[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double2 t = *reinterpret_cast<const double2*>(tcol + 8 * r);
        tre[r] = t.x;
        tim[r] = t.y;
    }
    // ---- 1. trig of the 1Q parameters (owner lanes); the Q values go to LDS for the gates' raw angles
    {
        double2* t2 = reinterpret_cast<double2*>(xq);
#pragma unroll
        for (int a = 0; a < C::NA; ++a) {
            const int i = 4 * a + q;
            if (i < C::NP) {
                const int i3 = i - 3 * ((i * 43) >> 7);
                const double arg = (i3 == 0) ? 0.5 * xd[a] : xd[a];
                double s, c;
                sincos_any(arg, tbl, s, c);
                t2[i] = make_double2(c, s);
            } else if (i < C::N) {
                xq[C::OFF_QVAL + (i - C::NP)] = xd[a];
            }
        }
    }
    lds_fence();
    // lane r of the quad evaluates raw angle r of every gate:  raw = scale q[sel] + offset
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int sel = maps[j].sel[q];
        const double qv = xq[C::OFF_QVAL + QN * j + (sel < 0 ? 0 : sel)];
        const double raw = (sel < 0) ? maps[j].offset[q] : fma(maps[j].scale[q], qv, maps[j].offset[q]);
        double s, c;
        sincos_any(raw, tbl, s, c);
        reinterpret_cast<double2*>(xq + C::OFF_GTRIG)[4 * j + q] = make_double2(c, s);
    }
    lds_fence();

    // ---- 2. forward
    double Fr[4], Fi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { Fr[r] = (r == q) ? 1.0 : 0.0; Fi[r] = 0.0; }
#pragma unroll
    for (int j = 0; j <= K; ++j) {
        if (j > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) fh[((2 * (j - 1)) * 4 + r) * kRow] = make_double2(Fr[r], Fi[r]);  // f_j
        }
        const U3t B = load_u3(xq, 6 * j);
        const U3t A = load_u3(xq, 6 * j + 3);
        u3_col(B, Fr[0], Fi[0], Fr[1], Fi[1]);
        u3_col(B, Fr[2], Fi[2], Fr[3], Fi[3]);
        u3_col(A, Fr[0], Fi[0], Fr[2], Fi[2]);
        u3_col(A, Fr[1], Fi[1], Fr[3], Fi[3]);
        if (j < K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) fh[((2 * j + 1) * 4 + r) * kRow] = make_double2(Fr[r], Fi[r]);  // h_j
            cg_col(load_cg(xq + C::OFF_GTRIG + 8 * j), Fr, Fi);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { Wr[r] = Fr[r]; Wi[r] = Fi[r]; }

    // ---- 3. loss (BasicCost / SquareCost as in eval_quad)
    double pr = 0.0, pi = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pr = fma(tre[r], Fr[r], fma(tim[r], Fi[r], pr));
        pi = fma(tre[r], Fi[r], fma(-tim[r], Fr[r], pi));
    }
    pr = quad_sum(pr);
    pi = quad_sum(pi);
    const double at2 = pr * pr + pi * pi;
    const double rat = (at2 > 1e-300) ? fast_rsqrt(at2) : 0.0;
    const double at = at2 * rat;
    const double basic = 1.0 - 0.25 * at;
    fout = (cost_kind == 1) ? 0.8 * basic * (2.0 - basic) : basic;
    const double inv = (cost_kind == 1) ? 0.25 * rat * 1.6 * (1.0 - basic) : 0.25 * rat;
    const double zr = -pr * inv, zi = pi * inv;

    // ---- 4. backward
    double Ur[4], Ui[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Ur[r] = zr * tre[r] + zi * tim[r];
        Ui[r] = zi * tre[r] - zr * tim[r];
    }
    double Hr[4], Hi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { Hr[r] = Fr[r]; Hi[r] = Fi[r]; }
#pragma unroll
    for (int j = K; j >= 0; --j) {
        if (j < K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 v = fh[((2 * j + 1) * 4 + r) * kRow];
                Hr[r] = v.x; Hi[r] = v.y;
            }
        }
        const U3t B = load_u3(xq, 6 * j);
        const U3t A = load_u3(xq, 6 * j + 3);
        double part[6];
        const double m1 = im_mul(Ur[1], Ui[1], Hr[1], Hi[1]);
        const double m2 = im_mul(Ur[2], Ui[2], Hr[2], Hi[2]);
        const double m3 = im_mul(Ur[3], Ui[3], Hr[3], Hi[3]);
        part[1] = -(m1 + m3);
        part[4] = -(m2 + m3);
        u3_row(A, Ur[0], Ui[0], Ur[2], Ui[2]);
        u3_row(A, Ur[1], Ui[1], Ur[3], Ui[3]);
        double tB01r, tB01i, tB23r, tB23i;
        u3_row_t(B, Ur[0], Ui[0], Ur[1], Ui[1], tB01r, tB01i);
        u3_row_t(B, Ur[2], Ui[2], Ur[3], Ui[3], tB23r, tB23i);
        double fr[4], fi[4];
        if (j > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 v = fh[((2 * (j - 1)) * 4 + r) * kRow];
                fr[r] = v.x; fi[r] = v.y;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { fr[r] = (r == q) ? 1.0 : 0.0; fi[r] = 0.0; }
        }
        double th01, la01, th23, la23;
        dtheta_dlam_last(B, Ur[0], Ui[0], tB01r, tB01i, fr[0], fi[0], fr[1], fi[1], th01, la01);
        dtheta_dlam_last(B, Ur[2], Ui[2], tB23r, tB23i, fr[2], fi[2], fr[3], fi[3], th23, la23);
        part[0] = th01 + th23;
        part[2] = la01 + la23;
        const double l2 = im_mul(Ur[2], Ui[2], fr[2], fi[2]);
        const double l3 = im_mul(Ur[3], Ui[3], fr[3], fi[3]);
        part[5] = -(l2 + l3);
        part[3] = dtheta_pair(A, Ur[0], Ui[0], Ur[2], Ui[2], fr[0], fi[0], fr[2], fi[2]) +
                  dtheta_pair(A, Ur[1], Ui[1], Ur[3], Ui[3], fr[1], fi[1], fr[3], fi[3]);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const double sum = quad_sum(part[m]);
            if (q == 0) xq[C::OFF_GP + 6 * j + m] = sum;
        }
        if (j > 0) {
            // gate j: d loss / d raw angle = Re( u~ (dG / d angle) h_{j-1} ), summed over the four columns
            double hr[4], hi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 v = fh[((2 * (j - 1) + 1) * 4 + r) * kRow];
                hr[r] = v.x; hi[r] = v.y;
            }
            const CGt g = load_cg(xq + C::OFF_GTRIG + 8 * (j - 1));
            double d[4];
            // a:     d = -sin a,  e = -i e^{i phi_c} cos a = cos a (sin phi_c, -cos phi_c)
            d[0] = cg_block_dot(-g.sa, g.ca * g.spc, -g.ca * g.cpc, Ur[1], Ui[1], Ur[2], Ui[2], hr[1], hi[1], hr[2], hi[2]);
            // phi_c: d = 0,       e = i w,  w = sin a (sin phi_c, -cos phi_c)  ->  i w = sin a (cos phi_c, sin phi_c)
            d[1] = cg_block_dot(0.0, g.sa * g.cpc, g.sa * g.spc, Ur[1], Ui[1], Ur[2], Ui[2], hr[1], hi[1], hr[2], hi[2]);
            d[2] = cg_block_dot(-g.sb, g.cb * g.spg, -g.cb * g.cpg, Ur[0], Ui[0], Ur[3], Ui[3], hr[0], hi[0], hr[3], hi[3]);
            d[3] = cg_block_dot(0.0, g.sb * g.cpg, g.sb * g.spg, Ur[0], Ui[0], Ur[3], Ui[3], hr[0], hi[0], hr[3], hi[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double sum = quad_sum(d[r]);
                if (q == 0) xq[C::OFF_DQ + 4 * (j - 1) + r] = sum;
            }
            cg_row(g, Ur, Ui);  // u <- u~ G_j
        }
    }
    // ---- 5. gradient in the owners' slots
    lds_fence();
#pragma unroll
    for (int a = 0; a < C::NA; ++a) {
        const int i = 4 * a + q;
        double v = 0.0;
        if (i < C::NP) {
            v = xq[C::OFF_GP + i];
        } else if (i < C::N) {
            const int j = (i - C::NP) / QN, m = (i - C::NP) - QN * j;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (maps[j].sel[r] == m) v = fma(maps[j].scale[r], xq[C::OFF_DQ + 4 * j + r], v);
        }
        gd[a] = v;
    }
    lds_fence();
}

template <int K, int QN>
struct EvalV2Args {
    const double* targets;
    const double* x;          // [M][n]
    const int32_t* target_of; // [M]
    int64_t n_items;
    double* loss;
    double* grad;             // [M][n] or nullptr
    double* unitary;          // [M][32] or nullptr
    int32_t cost_kind;
    const V2GateMap* maps;    // [K]
};

template <int K, int QN>
__global__ void __launch_bounds__(kWave, 1) eval_v2_kernel(EvalV2Args<K, QN> args) {
    using C = CfgV2<K, QN>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x, q = lane & 3, quad = lane >> 2;
    double2* fh = reinterpret_cast<double2*>(lds + C::LDS_XCHG) + lane;
    double2* tbl = reinterpret_cast<double2*>(lds + C::LDS_XCHG + C::LDS_FH + C::LDS_BOUNDS);
    load_sincos_table(tbl, lane);
    lds_fence();
    const int64_t item = (int64_t)blockIdx.x * kQuadsPerWave + quad;
    const bool live = item < args.n_items;
    const int64_t it = live ? item : 0;
    const double* tcol = args.targets + (int64_t)args.target_of[it] * 32 + q * 2;
    double xd[C::NA], gd[C::NA];
#pragma unroll
    for (int a = 0; a < C::NA; ++a) {
        const int i = 4 * a + q;
        xd[a] = (i < C::N) ? args.x[it * C::N + i] : 0.0;
    }
    double f, Wr[4], Wi[4];
    eval_quad_v2<K, QN>(xd, tcol, args.maps, lds + quad * C::XSTRIDE, fh, tbl, q, args.cost_kind, f, gd, Wr, Wi);
    if (live) {
        if (q == 0) args.loss[item] = f;
        if (args.unitary) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                args.unitary[item * 32 + (r * 4 + q) * 2] = Wr[r];
                args.unitary[item * 32 + (r * 4 + q) * 2 + 1] = Wi[r];
            }
        }
        if (args.grad) {
#pragma unroll
            for (int a = 0; a < C::NA; ++a) {
                const int i = 4 * a + q;
                if (i < C::N) args.grad[item * C::N + i] = gd[a];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// projected quasi-Newton minimisation: one item per quad, static assignment (the V2 callers run a few targets)
// ---------------------------------------------------------------------------------------------------------------
template <int K, int QN>
struct MinimizeV2Args {
    const double* targets;     // resident targets
    const int32_t* active;     // [n_active] target index per slot, or nullptr = identity
    int32_t n_active;
    int32_t restarts;
    const double* x0;          // [M][n] or nullptr
    const double* init_lo;     // [n] start points ~ U[init_lo, init_hi) (parameter_guess, basisv2.py:150-172)
    const double* init_hi;
    const double* bound_lo;    // [n] box bounds (-inf / +inf: none)
    const double* bound_hi;
    int32_t maxiter;
    double gtol, stop_loss, gtol_far, far_loss;
    uint64_t seed;
    int64_t target_base;
    int32_t cost_kind;
    const V2GateMap* maps;
    double* item_loss;         // [M] outputs, [slot][restart]
    double* item_x;            // [M][n]
    int32_t* item_iters;
    int32_t* item_status;
    int32_t* item_evals;
    int32_t* item_acc;
    StageCtl* ctl;
};

template <int K, int QN>
__global__ void __launch_bounds__(kWave, 1) minimize_v2_kernel(MinimizeV2Args<K, QN> args) {
    using C = CfgV2<K, QN>;
    constexpr int NA = C::NA;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x, q = lane & 3, quad = lane >> 2;
    double* xq = lds + quad * C::XSTRIDE;
    float* xq32 = reinterpret_cast<float*>(xq);
    double2* fh = reinterpret_cast<double2*>(lds + C::LDS_XCHG) + lane;
    double2* bnd = reinterpret_cast<double2*>(lds + C::LDS_XCHG + C::LDS_FH);  // [4 NA] (lo, hi), index = parameter
    double2* tbl = reinterpret_cast<double2*>(lds + C::LDS_XCHG + C::LDS_FH + C::LDS_BOUNDS);
    load_sincos_table(tbl, lane);
    for (int i = lane; i < 4 * NA; i += kWave)
        bnd[i] = (i < C::N) ? make_double2(args.bound_lo[i], args.bound_hi[i]) : make_double2(0.0, 0.0);
    lds_fence();
    const int64_t n_items = (int64_t)args.n_active * args.restarts;
    const int64_t item = (int64_t)blockIdx.x * kQuadsPerWave + quad;
    bool live = item < n_items;
    const int64_t it0 = live ? item : 0;
    const int slot = (int)(it0 / args.restarts);
    const int restart = (int)(it0 - (int64_t)slot * args.restarts);
    const int tgt = args.active ? args.active[slot] : slot;
    const double* tcol = args.targets + (int64_t)tgt * 32 + q * 2;

    double x[NA], g[NA], p[NA], lo[NA], hi[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int i = 4 * a + q;
        const double2 b = bnd[i];
        lo[a] = b.x; hi[a] = b.y;
        double xv = 0.0;
        if (i < C::N) {
            if (args.x0) {
                xv = args.x0[it0 * C::N + i];
            } else {
                // same Philox stream layout as the fixed-gate path, span tagged with 0x100 (V2), mapped onto the
                // parameter's start range
                const double u = x0_philox(args.seed, (uint32_t)(tgt + (int)args.target_base), (uint32_t)restart, (uint32_t)(K | 0x100), (uint32_t)i) *
                                 (1.0 / 6.283185307179586476925286766559);
                xv = fma(u, args.init_hi[i] - args.init_lo[i], args.init_lo[i]);
            }
            xv = fmin(fmax(xv, lo[a]), hi[a]);
        }
        x[a] = xv; g[a] = 0.0; p[a] = 0.0;
    }
    HMat<NA> H;
    h_set_identity_where<NA>(H, q, true);
    bool fresh = true, scaled = false;
    int nev = 0, iters = 0, nback = 0, nstall = 0, status = ST_MAXITER;
    double f = 0.0, alpha = 0.0, gs = 0.0, gp = 0.0, gnorm = 0.0, grow = 1.0;

    while (__any(live)) {
        // ---- trial point: projection of x + alpha p onto the box; s = actual step
        double xt[NA], sv[NA], gt[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            xt[a] = fmin(fmax(fma(alpha, p[a], x[a]), lo[a]), hi[a]);
            sv[a] = xt[a] - x[a];
        }
        gs = qdot<NA>(g, sv);  // slope of the linear model along the projected step
        double ft, Wr[4], Wi[4];
        eval_quad_v2<K, QN>(xt, tcol, args.maps, xq, fh, tbl, q, args.cost_kind, ft, gt, Wr, Wi);
        const bool active = live;
        const bool finite = isfinite(ft);
        // The quasi-Newton algebra lives in the subspace of the variables that can move: the gradient component of a
        // parameter that is fixed (lo == hi: a bounded-to-a-point Q, the theta / phi of an rz layer) or that sits on a
        // bound with the descent direction pointing out of the box is dropped.  (Left in, such a component -- d loss /
        // d alpha of a fixed gate is large -- enters y = g' - g with s = 0 there and poisons H.)
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const bool blocked = (xt[a] <= lo[a] && gt[a] > 0.0) || (xt[a] >= hi[a] && gt[a] < 0.0);
            gt[a] = (finite && !blocked) ? gt[a] : 0.0;
        }
        const bool armijo = finite && (ft <= f + kArmijoC1 * gs);
        const bool acc = active && (fresh ? finite : armijo);
        const bool step = acc && !fresh;
        nev += active ? (acc ? 0x100001 : 1) : 0;

        // ---- quasi-Newton update with s = the projected step (zero for quads that do not step)
        double qv[NA];
        h_matvec<NA>(H, gt, xq32, q, qv);
        double sy = 0.0, yy = 0.0, ss = 0.0;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const double sa = step ? sv[a] : 0.0;
            const double ya = gt[a] - g[a];
            sy = fma(sa, ya, sy);
            yy = fma(ya, ya, yy);
            ss = fma(sa, sa, ss);
        }
        sy = quad_sum(sy); yy = quad_sum(yy); ss = quad_sum(ss);
        const bool curv = step && sy > 0.0 && (sy * sy > (kCurvEps * kCurvEps) * (ss * yy));
        const bool too_short = sy < (1.0 - kWolfeC2) * (-gs);
        const bool first = curv && !scaled;
        scaled = scaled || curv;
        const double fac = first ? (sy * fast_rcp(yy)) : 1.0;
        {
            const float f32 = (float)fac;
            const f32x2 f2 = f32x2{f32, f32};
#pragma unroll
            for (int b = 0; b < NA * (NA + 1) / 2; ++b) {
                H.h[b][0] *= f2;
                H.h[b][1] *= f2;
            }
        }
        // u = H y = H g' - H g.  With projected directions p is not -H g any more, so H g is formed explicitly:
        // hg = H g  (second mat-vec; this path is not the tuned one)
        double hg[NA];
        h_matvec<NA>(H, g, xq32, q, hg);
        double yu = 0.0;
        double ua[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            qv[a] *= fac;  // H was scaled after q = H g' was formed; hg already uses the scaled H
            ua[a] = qv[a] - hg[a];
            yu = fma(gt[a] - g[a], ua[a], yu);
        }
        yu = quad_sum(yu);
        const double rho = curv ? fast_rcp(sy) : 0.0;
        const double cf = rho * (1.0 + rho * yu);
        double wg = 0.0, sg = 0.0;
        {
            float s32[NA], w32[NA], v32[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const double sa = step ? sv[a] : 0.0;
                const double wa = cf * sa - rho * ua[a];
                const double va = -rho * ua[a];
                s32[a] = (float)sa; w32[a] = (float)wa; v32[a] = (float)va;
                wg = fma(wa, gt[a], wg);
                sg = fma(sa, gt[a], sg);
            }
            h_update<NA>(H, s32, w32, v32, xq32, q);
        }
        wg = quad_sum(wg);
        sg = quad_sum(sg);

        bool done = false;
        if (acc) {
            nstall = (step && (f - ft) <= kStallDf) ? nstall + 1 : 0;
            f = ft;
            if (step) ++iters;
            nback = 0;
            grow = (step && too_short) ? fmin(grow * kGrowFactor, kGrowMax) : 1.0;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const double sa = step ? sv[a] : 0.0;
                const double va = -rho * ua[a];
                x[a] = xt[a];
                g[a] = gt[a];
                // new direction -H' g' = -(q + s (w.g') + v (s.g')), then projected: no component may point out of the box
                double pn = -(qv[a] + sa * wg + va * sg);
                const bool at_lo = x[a] <= lo[a], at_hi = x[a] >= hi[a];
                if ((at_lo && pn < 0.0) || (at_hi && pn > 0.0)) pn = 0.0;
                p[a] = pn;
            }
        } else if (active) {
            if (fresh) {
                f = ft; status = ST_NONFINITE; done = true;
            } else {
                const double denom = 2.0 * (ft - f - gs);
                const double anew = (finite && denom > 0.0 && isfinite(denom)) ? (-gs * alpha / denom) : 0.5 * alpha;
                alpha = fmin(fmax(anew, 0.1 * alpha), 0.5 * alpha);
                grow = 1.0;
                ++nback;
            }
        }
        {
            // projected gradient norm: components whose descent direction -g leaves the box do not count
            double m = 0.0;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const bool at_lo = x[a] <= lo[a], at_hi = x[a] >= hi[a];
                const bool blocked = (at_lo && g[a] > 0.0) || (at_hi && g[a] < 0.0);
                m = fmax(m, blocked ? 0.0 : fabs(g[a]));
            }
            gnorm = quad_max(m);
        }
        gp = qdot<NA>(g, p);
        const double pp = qdot<NA>(p, p);
        if (acc) {
            alpha = (pp > 1e-300) ? fmin(grow, kStepMax * fast_rsqrt(pp)) : grow;
            if (f < args.stop_loss || gnorm < args.gtol || (gnorm < args.gtol_far && f > args.far_loss)) { status = ST_CONVERGED; done = true; }
            else if (nstall >= 2) { status = ST_STALLED; done = true; }
            else if (iters >= args.maxiter) { status = ST_MAXITER; done = true; }
        } else if (active && !fresh) {
            if (nback > kMaxBacktrack) { status = (gnorm < kStallGnorm) ? ST_STALLED : ST_LINESEARCH; done = true; }
        }
        fresh = false;
        const bool reset = active && !done && !(gp < 0.0);
        if (__any(reset)) {
            h_set_identity_where<NA>(H, q, reset);
            double gg2 = 0.0;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const bool at_lo = x[a] <= lo[a], at_hi = x[a] >= hi[a];
                double pn = -g[a];
                if ((at_lo && pn < 0.0) || (at_hi && pn > 0.0)) pn = 0.0;
                p[a] = reset ? pn : p[a];
                gg2 = fma(g[a], pn, gg2);
            }
            gg2 = quad_sum(gg2);
            gp = reset ? gg2 : gp;
            // nothing left to move along: a KKT point of the box-constrained problem
            if (reset && !(gg2 < 0.0)) { status = ST_CONVERGED; done = true; }
        }
        if (active && done) {
            if (q == 0) {
                args.item_loss[item] = f;
                args.item_iters[item] = iters;
                args.item_status[item] = status;
                args.item_evals[item] = nev & 0xFFFFF;
                args.item_acc[item] = (int)((unsigned)nev >> 20);
            }
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int i = 4 * a + q;
                if (i < C::N) args.item_x[item * C::N + i] = x[a];
            }
            live = false;
            alpha = 0.0;
#pragma unroll
            for (int a = 0; a < NA; ++a) p[a] = 0.0;
        }
    }
}

}  // namespace slamdev
