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
// Kernels (round 3: off the first-correct path): the work decomposition of the fixed-gate path -- a quad of lanes per
// (target, seed) item, lane c owns column c of the running product and row c of the backward vector, fp32 packed inverse
// Hessian in registers -- with
//   * the LEAN layout: only the layer outputs h_j go through LDS, the layer inputs f_j = G_j h_{j-1} are recomputed;
//   * table-driven sincos with all lookups requested ahead (no out-of-line call in the loop, no scratch);
//   * gradient partials pair-summed by one DPP stage and added by their owner lane (was: a full quad reduction each);
//   * persistent wavefronts: a quad that finishes pulls the next (target, restart) item from the stage's RESTART-MAJOR
//     queue; a restart whose target already has a successful lower-index restart is dropped when pulled (the reference's
//     sequential break, optimizer.py:287-295);
//   * ONE symmetric mat-vec per iteration: H g is carried from the previous iteration, so H y = H g' - H g needs H g' only
//     (the projected direction is no longer -H g, which is why the first version formed both);
//   * the one-off scaling of the initial inverse Hessian as the scalar hs1 (metric H + hs1 I), as in minimize_kernel;
//   * the queue taken in wave-private chunks and scanned 64 positions per flag load (as minimize_kernel: one atomic per
//     pulled position bound the stages in which most positions are void);
//   * the quad-lane index re-materialised every round for spans >= 2, the general gate class and the bounded kernels: LDS
//     addresses derived from it are not hoisted and held across the loop (no scratch, fewer AGPRs, span 2 of the RiSwap
//     class at two wavefronts per SIMD).
// plus the gradient with respect to the gate angles, the projected quasi-Newton step for the box bounds (a failed line
// search restarts the metric instead of ending the item) and the cost constraint of set_constraint (basisv2.py:192-200)
// as a multiplier method around that loop (slam_v2_set_constraint).
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
    static constexpr int NAP = (NP + 3) / 4;  // slots that can hold a 1Q parameter
    // per-quad LDS area (doubles): trig of the P parameters [2 NP], gate trig [8 K], trial Q values [NQ -> even],
    // pair sums of the 1Q partials [2 NP], pair sums of the raw-angle partials [8 K]; the fp32 mat-vec / update exchanges of
    // slam_device.hpp (10 NA - 8 doubles) reuse the front of it between evaluations
    static constexpr int OFF_GTRIG = 2 * NP;
    static constexpr int OFF_QVAL = OFF_GTRIG + 8 * K;
    static constexpr int OFF_GP = OFF_QVAL + ((NQ + 1) / 2) * 2;
    static constexpr int OFF_DQ = OFF_GP + 2 * NP;
    static constexpr int XNEED0 = OFF_DQ + 8 * K;
    static constexpr int XNEED = XNEED0 > 10 * NA - 8 ? XNEED0 : 10 * NA - 8;
    static constexpr int XSTRIDE = (XNEED - 8 + 15) / 16 * 16 + 8;
    static constexpr int LDS_XCHG = kQuadsPerWave * XSTRIDE;
    static constexpr int LDS_FH = K * 4 * kRow * 2;              // h_0..h_{K-1}: [vector][row][lane + pad] double2
    static constexpr int LDS_BOUNDS = 2 * NA * 4;                // (lo, hi) per parameter, padded to 4 NA
    static constexpr int LDS_CW = NA * 4 + kQuadsPerWave * 8;    // cost constraint (after the trig table): one weight per parameter, then per quad
                                                                 // (mu, loss, c, multiplier updates) at x and (loss, c) at the trial point
    // span 1: the raw start values (x0_philox) of kRing consecutive queue positions, generated one position per lane when the
    // wavefront reaches them -- a refill then copies.  (Until round 5 every taking lane ran its NA Philox blocks at the refill and the
    // whole wavefront waited: ~460 vector instructions per refill event, one event every two or three rounds of ~1000 at span 1.)
    static constexpr int kRing = (K == 1) ? 32 : 0;
    static constexpr int RN = (N + 1) / 2 * 2;  // ring row: N values, padded to a double2 boundary
    static constexpr int LDS_RING = kRing * RN;
    static constexpr int LDS_DOUBLES = LDS_XCHG + LDS_FH + LDS_BOUNDS + kSincosLdsDoubles + LDS_CW + LDS_RING;
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

// Gate sub-class GQ = 1 ("a only"): phi_c = 0, b = 0, phi_g irrelevant -- RiSwapGate(alpha), the gate class of most V2 callers
// (decomp_trajectory.ipynb cell 5, basisv2.py:31).  The gate is the identity on |00>, |11> and [[c, -i s'], [-i s', c]] ... with
// w = -i sin a on {|01>, |10>}: 8 real products per application instead of 32, one raw-angle derivative instead of four.
__device__ __forceinline__ void ra_apply(double c, double sa, double (&Fr)[4], double (&Fi)[4]) {
    const double lr = Fr[1], li = Fi[1], hr = Fr[2], hi = Fi[2];  // w = (0, -sa): the block is symmetric, column and row action agree
    Fr[1] = fma(c, lr, sa * hi);
    Fi[1] = fma(c, li, -(sa * hr));
    Fr[2] = fma(c, hr, sa * li);
    Fi[2] = fma(c, hi, -(sa * lr));
}
// Re( u (dG/da) h ) on the (1, 2) block: d = -sin a, e = -i cos a
__device__ __forceinline__ double ra_dot(double c, double sa, double ulr, double uli, double uhr, double uhi, double hlr, double hli, double hhr,
                                         double hhi) {
    const double vlr = fma(c, hhi, -(sa * hlr)), vli = -fma(c, hhr, sa * hli);
    const double vhr = fma(c, hli, -(sa * hhr)), vhi = -fma(c, hlr, sa * hhi);
    return (ulr * vlr - uli * vli) + (uhr * vhr - uhi * vhi);
}

// bit a set: slot a of lane q (parameter 4a + q) is a theta of a U gate (index < NP and divisible by 3)
template <int K, int QN>
__device__ __forceinline__ int theta_slot_bits_v2(int q) {
    int bits = 0;
#pragma unroll
    for (int a = 0; a < CfgV2<K, QN>::NAP; ++a) {
        const int i = 4 * a + q;
        bits |= ((i < CfgV2<K, QN>::NP) && ((i - 3 * ((i * 43) >> 7)) == 0)) << a;  // i % 3 for i < 128
    }
    return bits;
}

// ---------------------------------------------------------------------------------------------------------------
// loss + gradient with respect to all n parameters for the quad's item
//   xd     this lane's parameter slots (component 4a + q); |x| < 2e8 (table-driven sincos), checked by the host for explicit
//          seeds and guaranteed by the optimizer for its own points (HUGE_ARGS = true: the evaluation entry point, any x)
//   maps   gate maps of G_1..G_K (wave-uniform, global memory)
// ---------------------------------------------------------------------------------------------------------------
template <int K, int QN, bool HUGE_ARGS, int GQ = 0>
__device__ __forceinline__ void eval_quad_v2(const double (&xd)[CfgV2<K, QN>::NA], const double* tcol, const V2GateMap* maps, double* xq,
                                             double2* fh, const double2* tbl, int q, int theta_bits, int cost_kind, double& fout,
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
        if constexpr (HUGE_ARGS) {
#pragma unroll
            for (int a = 0; a < C::NAP; ++a) {
                const int i = 4 * a + q;
                const int i3 = i - 3 * ((i * 43) >> 7);
                const double arg = (i3 == 0) ? 0.5 * xd[a] : xd[a];
                double s, c;
                sincos_any(arg, tbl, s, c);
                if (i < C::NP) t2[i] = make_double2(c, s);
            }
        } else {
            // table lookups requested four slots ahead of their polynomials (all of them at once would keep 7 registers
            // per slot alive: too many next to the long templates' vectors)
            constexpr int kChunk = 4;
            const SincosLits L = sincos_lits_device();
#pragma unroll
            for (int a0 = 0; a0 < C::NAP; a0 += kChunk) {
                double rr[kChunk];
                int kk[kChunk];
                double2 tt[kChunk];
#pragma unroll
                for (int c = 0; c < kChunk; ++c) {
                    const int a = a0 + c;
                    if (a < C::NAP) {
                        const double arg = __builtin_amdgcn_ldexp(xd[a], __builtin_amdgcn_sbfe(theta_bits, a, 1));
                        sincos_tbl_lookup<true>(arg, tbl, L, rr[c], kk[c], tt[c]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < kChunk; ++c) {
                    const int a = a0 + c;
                    if (a < C::NAP) {
                        double sn, cs;
                        sincos_tbl_finish<true>(rr[c], kk[c], tt[c], L, sn, cs);
                        if (4 * a + q < C::NP) t2[4 * a + q] = make_double2(cs, sn);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int a = C::NP / 4; a < C::NA; ++a) {
            const int i = 4 * a + q;
            if (i >= C::NP && i < C::N) xq[C::OFF_QVAL + (i - C::NP)] = xd[a];
        }
    }
    lds_fence();
    // lane r of the quad evaluates raw angle r of every gate:  raw = scale q[sel] + offset
    {
        double raw[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int sel = maps[j].sel[q];
            const double qv = xq[C::OFF_QVAL + QN * j + (sel < 0 ? 0 : sel)];
            raw[j] = (sel < 0) ? maps[j].offset[q] : fma(maps[j].scale[q], qv, maps[j].offset[q]);
        }
        if constexpr (HUGE_ARGS) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                double s, c;
                sincos_any(raw[j], tbl, s, c);
                reinterpret_cast<double2*>(xq + C::OFF_GTRIG)[4 * j + q] = make_double2(c, s);
            }
        } else {
            double rr[K];
            int kk[K];
            double2 tt[K];
            const SincosLits L = sincos_lits_device();
#pragma unroll
            for (int j = 0; j < K; ++j) sincos_tbl_lookup<true>(raw[j], tbl, L, rr[j], kk[j], tt[j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                double s, c;
                sincos_tbl_finish<true>(rr[j], kk[j], tt[j], L, s, c);
                reinterpret_cast<double2*>(xq + C::OFF_GTRIG)[4 * j + q] = make_double2(c, s);
            }
        }
    }
    lds_fence();

    // ---- 2. forward; only the layer outputs h_j (j < K) are stored
    double Fr[4], Fi[4];
    // this lane's unit vector e_q as real pairs per qubit: (1 - bit, bit)
    const double e0[2] = {(q & 1) ? 0.0 : 1.0, (q & 1) ? 1.0 : 0.0};
    const double e1[2] = {(q & 2) ? 0.0 : 1.0, (q & 2) ? 1.0 : 0.0};
#pragma unroll
    for (int j = 0; j <= K; ++j) {
        const U3t B = load_u3(xq, 6 * j);
        const U3t A = load_u3(xq, 6 * j + 3);
        if (j == 0) {
            // layer 0 acts on the unit vector e_q:  K_0 e_q = (A e_{q >> 1}) (x) (B e_{q & 1}) -- 40 operations instead of 64
            double b0r, b0i, b1r, b1i, a0r, a0i, a1r, a1i;
            u3_unit(B, e0[0], e0[1], b0r, b0i, b1r, b1i);
            u3_unit(A, e1[0], e1[1], a0r, a0i, a1r, a1i);
            Fr[0] = a0r * b0r - a0i * b0i; Fi[0] = a0r * b0i + a0i * b0r;
            Fr[1] = a0r * b1r - a0i * b1i; Fi[1] = a0r * b1i + a0i * b1r;
            Fr[2] = a1r * b0r - a1i * b0i; Fi[2] = a1r * b0i + a1i * b0r;
            Fr[3] = a1r * b1r - a1i * b1i; Fi[3] = a1r * b1i + a1i * b1r;
        } else {
            u3_col(B, Fr[0], Fi[0], Fr[1], Fi[1]);
            u3_col(B, Fr[2], Fi[2], Fr[3], Fi[3]);
            u3_col(A, Fr[0], Fi[0], Fr[2], Fi[2]);
            u3_col(A, Fr[1], Fi[1], Fr[3], Fi[3]);
        }
        if (j < K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) fh[(j * 4 + r) * kRow] = make_double2(Fr[r], Fi[r]);  // h_j
            if constexpr (GQ == 1) {
                const double2 ta = *reinterpret_cast<const double2*>(xq + C::OFF_GTRIG + 8 * j);
                ra_apply(ta.x, ta.y, Fr, Fi);
            } else {
                cg_col(load_cg(xq + C::OFF_GTRIG + 8 * j), Fr, Fi);
            }
        }
        // long templates: keep the scheduler from pulling every layer's table reads to the front (12 registers per layer)
        if constexpr (K >= 3) __builtin_amdgcn_sched_barrier(0);
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
    quad_sum2(pr, pi);
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
    double Hr[4], Hi[4];  // h_j: the current layer's output
#pragma unroll
    for (int r = 0; r < 4; ++r) { Hr[r] = Fr[r]; Hi[r] = Fi[r]; }
    // pair sums (lanes q, q ^ 1) of a partial; the even lane of the pair parks it at [2 i + (q >> 1)] for the owner to add
    auto park = [&](int off, int i, double v) {
        const double ps = v + dpp_f64<0xB1>(v);
        if ((q & 1) == 0) xq[off + 2 * i + (q >> 1)] = ps;
    };
#pragma unroll
    for (int j = K; j >= 1; --j) {
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
        // h_{j-1} (the gate's input; also the next iteration's layer output) and f_j = G_j h_{j-1}
        double Pr[4], Pi[4], fr[4], fi[4];
        CGt g;
        if (j > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 v = fh[((j - 1) * 4 + r) * kRow];
                Pr[r] = v.x; Pi[r] = v.y;
                fr[r] = v.x; fi[r] = v.y;
            }
            g = load_cg(xq + C::OFF_GTRIG + 8 * (j - 1));
            if constexpr (GQ == 1) ra_apply(g.ca, g.sa, fr, fi);
            else cg_col(g, fr, fi);
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
        for (int m = 0; m < 6; ++m) park(C::OFF_GP, 6 * j + m, part[m]);
        if (j > 0) {
            // gate j: d loss / d raw angle = Re( u~ (dG / d angle) h_{j-1} ), summed over the four columns
            double d[4];
            if constexpr (GQ == 1) {
                park(C::OFF_DQ, 4 * (j - 1), ra_dot(g.ca, g.sa, Ur[1], Ui[1], Ur[2], Ui[2], Pr[1], Pi[1], Pr[2], Pi[2]));
                ra_apply(g.ca, g.sa, Ur, Ui);  // u <- u~ G_j
            } else {
            // a:     d = -sin a,  e = -i e^{i phi_c} cos a = cos a (sin phi_c, -cos phi_c)
            d[0] = cg_block_dot(-g.sa, g.ca * g.spc, -g.ca * g.cpc, Ur[1], Ui[1], Ur[2], Ui[2], Pr[1], Pi[1], Pr[2], Pi[2]);
            // phi_c: d = 0,       e = i w,  w = sin a (sin phi_c, -cos phi_c)  ->  i w = sin a (cos phi_c, sin phi_c)
            d[1] = cg_block_dot(0.0, g.sa * g.cpc, g.sa * g.spc, Ur[1], Ui[1], Ur[2], Ui[2], Pr[1], Pi[1], Pr[2], Pi[2]);
            d[2] = cg_block_dot(-g.sb, g.cb * g.spg, -g.cb * g.cpg, Ur[0], Ui[0], Ur[3], Ui[3], Pr[0], Pi[0], Pr[3], Pi[3]);
            d[3] = cg_block_dot(0.0, g.sb * g.cpg, g.sb * g.spg, Ur[0], Ui[0], Ur[3], Ui[3], Pr[0], Pi[0], Pr[3], Pi[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) park(C::OFF_DQ, 4 * (j - 1) + r, d[r]);
            cg_row(g, Ur, Ui);  // u <- u~ G_j
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { Hr[r] = Pr[r]; Hi[r] = Pi[r]; }
        }
        if constexpr (K >= 3) __builtin_amdgcn_sched_barrier(0);
    }
    {
        // layer 0 from its output side (slam_device.hpp:l0_gate_partials): Ur/Ui is the backward vector behind G_1
        const U3t B = load_u3(xq, 0);
        const U3t A = load_u3(xq, 3);
        double b0r, b0i, b1r, b1i, tbr, tbi, a0r, a0i, a1r, a1i, tar, tai;
        u3_unit_t(B, e0[0], e0[1], b0r, b0i, b1r, b1i, tbr, tbi);
        u3_unit_t(A, e1[0], e1[1], a0r, a0i, a1r, a1i, tar, tai);
        const double wB0r = (Ur[0] * a0r - Ui[0] * a0i) + (Ur[2] * a1r - Ui[2] * a1i), wB0i = (Ur[0] * a0i + Ui[0] * a0r) + (Ur[2] * a1i + Ui[2] * a1r);
        const double wB1r = (Ur[1] * a0r - Ui[1] * a0i) + (Ur[3] * a1r - Ui[3] * a1i), wB1i = (Ur[1] * a0i + Ui[1] * a0r) + (Ur[3] * a1i + Ui[3] * a1r);
        const double wA0r = (Ur[0] * b0r - Ui[0] * b0i) + (Ur[1] * b1r - Ui[1] * b1i), wA0i = (Ur[0] * b0i + Ui[0] * b0r) + (Ur[1] * b1i + Ui[1] * b1r);
        const double wA1r = (Ur[2] * b0r - Ui[2] * b0i) + (Ur[3] * b1r - Ui[3] * b1i), wA1i = (Ur[2] * b0i + Ui[2] * b0r) + (Ur[3] * b1i + Ui[3] * b1r);
        double part[6];
        l0_gate_partials(B, e0[0], e0[1], wB0r, wB0i, wB1r, wB1i, b0r, b0i, b1r, b1i, tbr, tbi, part[0], part[1], part[2]);
        l0_gate_partials(A, e1[0], e1[1], wA0r, wA0i, wA1r, wA1i, a0r, a0i, a1r, a1i, tar, tai, part[3], part[4], part[5]);
#pragma unroll
        for (int m = 0; m < 6; ++m) park(C::OFF_GP, m, part[m]);
    }
    // ---- 5. gradient in the owners' slots
    lds_fence();
#pragma unroll
    for (int a = 0; a < C::NA; ++a) {
        const int i = 4 * a + q;
        double v = 0.0;
        if (i < C::NP) {
            const double2 ps = *reinterpret_cast<const double2*>(xq + C::OFF_GP + 2 * i);
            v = ps.x + ps.y;
        } else if (i < C::N) {
            const int j = (i - C::NP) / QN, m = (i - C::NP) - QN * j;
#pragma unroll
            for (int r = 0; r < (GQ == 1 ? 1 : 4); ++r) {  // (GQ = 1: only the angle a depends on a parameter)
                const double2 ps = *reinterpret_cast<const double2*>(xq + C::OFF_DQ + 2 * (4 * j + r));
                if (maps[j].sel[r] == m) v = fma(maps[j].scale[r], ps.x + ps.y, v);
            }
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
    eval_quad_v2<K, QN, true>(xd, tcol, args.maps, lds + quad * C::XSTRIDE, fh, tbl, q, 0, args.cost_kind, f, gd, Wr, Wi);
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
// Where the packed fp32 inverse Hessian of a wavefront's 16 items lives.  Up to 32 parameters (NA <= 8) it fits the
// registers next to the evaluation's working set (HMat, slam_device.hpp).  Beyond -- spans 4 and 5, span 3 with four
// parameters per gate: up to n = 56, 105 blocks = 420 registers per lane -- it cannot, and LDS has no room either (a block is
// 1 KiB per wavefront): those instantiations keep it in a per-wavefront slice of device memory, block b as one float4 per
// lane at [b][lane] (every access a fully coalesced 1 KiB line), read once by the mat-vec and read + written once by the
// rank-2 update per iteration.  Same arithmetic, same block order: the two variants give identical results.
// ---------------------------------------------------------------------------------------------------------------
template <int K, int QN>
__host__ __device__ constexpr bool v2_h_in_memory() { return CfgV2<K, QN>::NA > 8; }
template <int K, int QN>
__host__ __device__ constexpr int v2_h_floats_per_wave() { return CfgV2<K, QN>::NA * (CfgV2<K, QN>::NA + 1) / 2 * 4 * kWave; }

template <int NA>
__device__ __forceinline__ void hm_set_identity_where(f32x4* Hm, int q, bool where) {
    if (!where) return;
#pragma unroll 1
    for (int b = 0; b < NA; ++b)
#pragma unroll 1
        for (int a = 0; a <= b; ++a) {
            f32x4 v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (a == b) v = f32x4{q == 0 ? 1.0f : 0.0f, q == 1 ? 1.0f : 0.0f, q == 2 ? 1.0f : 0.0f, q == 3 ? 1.0f : 0.0f};
            Hm[blk(a, b) * kWave] = v;
        }
}
// out = H v, the loops of h_matvec with the blocks loaded from memory (column block b's NA.. loads are independent of the
// LDS broadcasts: they are in flight together)
template <int NA>
__device__ __forceinline__ void hm_matvec(const f32x4* Hm, const double (&vd)[NA], float* xq32, int q, double (&out)[NA]) {
    float v32[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        v32[a] = (float)vd[a];
        xq32[4 * a + q] = v32[a];
    }
    lds_fence();
    float* xt = xq32 + 4 * NA - 16;
    f32x2 acc[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) acc[a] = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int b = 0; b < NA; ++b) {
        const f32x4 vb = *reinterpret_cast<const f32x4*>(xq32 + 4 * b);
        const f32x2 vb0 = f32x2{vb.x, vb.y}, vb1 = f32x2{vb.z, vb.w};
        f32x2 t0 = f32x2{0.0f, 0.0f}, t1 = f32x2{0.0f, 0.0f};
#pragma unroll
        for (int a = 0; a <= b; ++a) {
            const f32x4 h = Hm[blk(a, b) * kWave];
            const f32x2 h0 = f32x2{h.x, h.y}, h1 = f32x2{h.z, h.w};
            acc[a] = __builtin_elementwise_fma(h0, vb0, acc[a]);
            acc[a] = __builtin_elementwise_fma(h1, vb1, acc[a]);
            if (a < b) {
                const f32x2 va = f32x2{v32[a], v32[a]};
                t0 = __builtin_elementwise_fma(h0, va, t0);
                t1 = __builtin_elementwise_fma(h1, va, t1);
            }
        }
        if (b >= 1) {
            xt[(4 * b + 0) * 4 + q] = t0.x;
            xt[(4 * b + 1) * 4 + q] = t0.y;
            xt[(4 * b + 2) * 4 + q] = t1.x;
            xt[(4 * b + 3) * 4 + q] = t1.y;
        }
        __builtin_amdgcn_sched_barrier(0);  // one column block's loads in flight at a time: hoisted, all NBLK cost 4 registers each
    }
    lds_fence();
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        float o = acc[a].x + acc[a].y;
        if (a >= 1) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(xt + (4 * a + q) * 4);
            o += (t.x + t.y) + (t.z + t.w);
        }
        out[a] = (double)o;
    }
    lds_fence();
}
// H += s w^T + v s^T
template <int NA>
__device__ __forceinline__ void hm_update(f32x4* Hm, const float (&s32)[NA], const float (&w32)[NA], const float (&v32)[NA], float* xq32, int q) {
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        xq32[4 * a + q] = w32[a];
        xq32[4 * NA + 4 * a + q] = s32[a];
    }
    lds_fence();
#pragma unroll
    for (int b = 0; b < NA; ++b) {
        const f32x4 wb = *reinterpret_cast<const f32x4*>(xq32 + 4 * b);
        const f32x4 sb = *reinterpret_cast<const f32x4*>(xq32 + 4 * NA + 4 * b);
        const f32x2 wb0 = f32x2{wb.x, wb.y}, wb1 = f32x2{wb.z, wb.w};
        const f32x2 sb0 = f32x2{sb.x, sb.y}, sb1 = f32x2{sb.z, sb.w};
#pragma unroll
        for (int a = 0; a <= b; ++a) {
            const f32x2 sa = f32x2{s32[a], s32[a]};
            const f32x2 va = f32x2{v32[a], v32[a]};
            const f32x4 h = Hm[blk(a, b) * kWave];
            f32x2 h0 = f32x2{h.x, h.y}, h1 = f32x2{h.z, h.w};
            h0 = __builtin_elementwise_fma(va, sb0, h0);
            h1 = __builtin_elementwise_fma(va, sb1, h1);
            h0 = __builtin_elementwise_fma(sa, wb0, h0);
            h1 = __builtin_elementwise_fma(sa, wb1, h1);
            Hm[blk(a, b) * kWave] = f32x4{h0.x, h0.y, h1.x, h1.y};
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    lds_fence();
}

template <int NA, bool MEM>
struct HStore {
    HMat<NA> regs;
    __device__ __forceinline__ void bind(float*, int) {}
    __device__ __forceinline__ void set_identity_where(int q, bool w) { h_set_identity_where<NA>(regs, q, w); }
    __device__ __forceinline__ void matvec(const double (&v)[NA], float* xq32, int q, double (&out)[NA]) { h_matvec<NA>(regs, v, xq32, q, out); }
    __device__ __forceinline__ void update(const float (&s)[NA], const float (&w)[NA], const float (&v)[NA], float* xq32, int q) { h_update<NA>(regs, s, w, v, xq32, q); }
};
template <int NA>
struct HStore<NA, true> {
    f32x4* mem;  // this lane's first block; block b at mem[b * 64]
    __device__ __forceinline__ void bind(float* wave_slice, int lane) { mem = reinterpret_cast<f32x4*>(wave_slice) + lane; }
    __device__ __forceinline__ void set_identity_where(int q, bool w) { hm_set_identity_where<NA>(mem, q, w); }
    __device__ __forceinline__ void matvec(const double (&v)[NA], float* xq32, int q, double (&out)[NA]) { hm_matvec<NA>(mem, v, xq32, q, out); }
    __device__ __forceinline__ void update(const float (&s)[NA], const float (&w)[NA], const float (&v)[NA], float* xq32, int q) { hm_update<NA>(mem, s, w, v, xq32, q); }
};

// ---------------------------------------------------------------------------------------------------------------
// projected quasi-Newton minimisation, persistent wavefronts over the stage's restart-major work queue
// ---------------------------------------------------------------------------------------------------------------
#ifndef SLAM_V2_REFILL_BATCH
#define SLAM_V2_REFILL_BATCH ((K == 1) ? 2 : 1)
#endif
#ifndef SLAM_V2_REMAT_Q
#define SLAM_V2_REMAT_Q 1
#endif
#ifndef SLAM_V2_REMAT_Q_COND
#define SLAM_V2_REMAT_Q_COND (K >= 2 || GQ == 0 || !FREE)
#endif
#ifndef SLAM_V2_WAVES
#define SLAM_V2_WAVES(K, QN, GQ, FREE) (((K) == 1 && (FREE)) ? 2 : 1)
#endif
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
    double exit_loss;          // SLAM_FLAG_EARLY_EXIT: a finished restart below this stops the target's LATER restarts
    uint32_t flags;
    uint64_t seed;
    int64_t target_base;
    int32_t cost_kind;
    const V2GateMap* maps;
    int32_t* solved;           // [n_active], zeroed before launch: restarts - r of the lowest-index successful restart r, 0 = none
    ItemRec* item_rec;         // [M] outputs, [slot][restart]
    double* item_x;            // [M][n]
    StageCtl* ctl;             // work queue head, round counter
    float* hmem;               // v2_h_in_memory<K, QN>(): [gridDim.x][v2_h_floats_per_wave] inverse Hessians; else unused
    // optional per-iteration trace (use_callback, optimizer.py:217-224), as MinimizeArgs: after accepted step number it >= 1 of
    // item m, trace_loss[m][it - 1] = loss and trace_x[m][it - 1][:] = parameters; nullptr = off
    double* trace_loss;        // [M][trace_cap]
    double* trace_x;           // [M][trace_cap][n]
    int32_t trace_cap;
    int32_t bounded;           // 0: every bound is (-inf, +inf) -- plain BFGS (optimizer.py:255): no projection, no blocked components
    // cost constraint of the template (CircuitTemplateV2.set_constraint, basisv2.py:192-200: circuit_cost(x) <= param_max_cost),
    // for circuit costs that are affine in the parameters over the box: sum_i cons_w[i] x_i <= cons_max.  nullptr = none.
    const double* cons_w;      // [n]
    double cons_max;           // (lowered by the feasibility tolerance by the host side)
    double cons_rho;           // penalty parameter of the augmented Lagrangian
    double cons_tol;           // feasibility / complementarity tolerance of the multiplier loop
};

// FREE = true: instantiated for launches without any finite bound (plain BFGS): the projection code and the trial point kept
// across the evaluation (2 NA registers) are compiled out; FREE = false handles both (args.bounded, wave-uniform)
constexpr int kMaxMultiplierUpdates = 24;

template <int K, int QN, int GQ = 0, bool FREE = false>
__global__ void __launch_bounds__(kWave, SLAM_V2_WAVES(K, QN, GQ, FREE)) minimize_v2_kernel(MinimizeV2Args<K, QN> args) {
    using C = CfgV2<K, QN>;
    constexpr int NA = C::NA;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x, quad = lane >> 2;
    int q = lane & 3;
    double* xq = lds + quad * C::XSTRIDE;
    float* xq32 = reinterpret_cast<float*>(xq);
    double2* fh = reinterpret_cast<double2*>(lds + C::LDS_XCHG) + lane;
    double2* bnd = reinterpret_cast<double2*>(lds + C::LDS_XCHG + C::LDS_FH);  // [4 NA] (lo, hi), index = parameter
    double2* tbl = reinterpret_cast<double2*>(lds + C::LDS_XCHG + C::LDS_FH + C::LDS_BOUNDS);
    load_sincos_table(tbl, lane);
    for (int i = lane; i < 4 * NA; i += kWave)
        bnd[i] = (i < C::N) ? make_double2(args.bound_lo[i], args.bound_hi[i]) : make_double2(0.0, 0.0);
    // cost constraint (FREE = false only): weights staged next to the bounds
    const bool cons = !FREE && args.cons_w != nullptr;  // wave-uniform
    double* cw = lds + C::LDS_XCHG + C::LDS_FH + C::LDS_BOUNDS + kSincosLdsDoubles;
    if (cons)
        for (int i = lane; i < 4 * NA; i += kWave) cw[i] = (i < C::N) ? args.cons_w[i] : 0.0;
    // the item's multiplier state lives in LDS (all four lanes of a quad write the same values): the kernels that handle bounds
    // without a constraint keep their register budget
    double* cst = cw + 4 * NA + quad * 8;  // [0] mu [1] loss(x) [2] c(x) [3] updates [4] loss(trial) [5] c(trial)
    double* const ring = lds + C::LDS_XCHG + C::LDS_FH + C::LDS_BOUNDS + kSincosLdsDoubles + C::LDS_CW;  // [kRing][RN]
    unsigned ring_base = 0, ring_end = 0;  // wave-uniform: queue positions the ring holds
    const bool use_ring = C::kRing > 0 && args.x0 == nullptr;  // wave-uniform
    lds_fence();
    const int theta_bits = theta_slot_bits_v2<K, QN>(q);
    // n_active < 0: the stage's target count is produced on the device by the previous stage's compaction (span loop enqueued
    // without host round trips, as for the fixed-gate path)
    const unsigned n_act = (unsigned)(args.n_active < 0 ? args.ctl->n_active : args.n_active);
    const unsigned n_items = n_act * (unsigned)args.restarts;
    const bool early = (args.flags & 1u) != 0;

    // ---- per-quad state
    bool live = false, fresh = false, scaled = false;
    unsigned item = 0;
    int slot = 0, restart = 0;
    int nev = 0, iters = 0, nback = 0, nstall = 0, status = ST_MAXITER;
    double f = 0.0, alpha = 0.0, gp = 0.0, gnorm = 0.0, grow = 1.0, hs1 = 0.0;
    const double* tcol = args.targets + q * 2;
    // hg = (H + hs1 I) g: the unprojected direction, negated.  Without bounds (FREE) the direction is never masked, p = -hg
    // throughout, and hg is not kept (HG(a) reads -p[a]): 2 NA registers
    double x[NA], g[NA], p[NA], hg[FREE ? 1 : NA];
#define HG(a) (FREE ? -p[a] : hg[FREE ? 0 : (a)])
#pragma unroll
    for (int a = 0; a < NA; ++a) { x[a] = 0.0; g[a] = 0.0; p[a] = 0.0; if constexpr (!FREE) hg[a] = 0.0; }
    HStore<NA, v2_h_in_memory<K, QN>()> H;
    H.bind(args.hmem + (size_t)blockIdx.x * v2_h_floats_per_wave<K, QN>(), lane);
    H.set_identity_where(q, true);
    bool exhausted = false;  // wave-uniform
    unsigned rounds = 0;
    // chunk of queue positions a wave takes per atomicAdd: big chunks mean few atomics, small batches need every wave busy
    const unsigned per_wave = n_items / gridDim.x;
    const unsigned kChunk = per_wave >= 256u ? 64u : (per_wave >= 64u ? 32u : 16u);
    unsigned cur_next = 0, cur_end = 0;  // wave-uniform
    unsigned pre_base = 0;               // lane 0: base of the prefetched chunk
    if (lane == 0) pre_base = atomicAdd(&args.ctl->work_counter, kChunk);

    while (true) {
#if SLAM_V2_REMAT_Q
        // q re-materialised every iteration (as minimize_kernel at k = 2): keeps the lane-dependent LDS addresses derived from it
        // from being hoisted out of the loop and held in registers across it
        if constexpr (SLAM_V2_REMAT_Q_COND) {
            asm volatile("" : "+v"(q));
            __builtin_assume((unsigned)q < 4u);
        }
#endif
        // ---- 1. idle quads pull the next queue positions (restart-major: position = restart * n_active + slot).
        // As in minimize_kernel: positions come in wave-private chunks (one atomicAdd per chunk, the next one requested
        // while the current one is consumed) and are scanned 64 at a time, so that the positions a successful sibling
        // restart has made void -- 15 of 16 at k = 2, 3 for the RiSwap class -- cost one flag load per window instead of
        // one atomic round trip each (first version: 0.5 G evaluations/s at k = 2 against 3.8 at k = 1: the stage was
        // bound by its single counter).
        bool taken = false;
        // (the refill code is wave-wide: at span 1, where items last ~ 30 rounds, it pays to let two quads go idle before running
        // it, as in minimize_kernel)
        constexpr int kRefillBatch = SLAM_V2_REFILL_BATCH;
        const int n_idle0 = __popcll(__ballot(!live && q == 0));
        const bool go = n_idle0 >= kRefillBatch || n_idle0 == kQuadsPerWave;
        while (go && !exhausted && __any(!live)) {
            if (cur_next >= cur_end) {
                const unsigned b = (unsigned)__builtin_amdgcn_readfirstlane((int)pre_base);
                if (b >= n_items) { exhausted = true; break; }
                cur_next = b;
                cur_end = (b + kChunk < n_items) ? b + kChunk : n_items;
                if (lane == 0) pre_base = atomicAdd(&args.ctl->work_counter, kChunk);
            }
            if constexpr (C::kRing > 0) {
                if (use_ring && cur_next >= ring_end) {
                    // the next kRing positions' start values: lane l runs the (N + 1) / 2 Philox blocks of position cur_next + l
                    ring_base = cur_next;
                    ring_end = (cur_next + (unsigned)C::kRing < cur_end) ? cur_next + (unsigned)C::kRing : cur_end;
                    if ((unsigned)lane < ring_end - ring_base) {
                        const unsigned gp = ring_base + (unsigned)lane;
                        const unsigned gr = gp / n_act, gsl = gp - gr * n_act;
                        const int gt_ = args.active ? args.active[gsl] : (int)gsl;
                        const uint32_t tw = (uint32_t)(gt_ + (int)args.target_base);
#pragma unroll 1
                        for (int m = 0; m < C::RN / 2; ++m) {
                            uint32_t w[4];
                            philox4x32_10((uint32_t)m, gr, tw, (uint32_t)(K | 0x100), (uint32_t)args.seed, (uint32_t)(args.seed >> 32), w);
                            *reinterpret_cast<double2*>(ring + lane * C::RN + 2 * m) = make_double2(x0_from_words(w[0], w[1]), x0_from_words(w[2], w[3]));
                        }
                    }
                    lds_fence();
                }
            }
            unsigned wlen = (cur_end - cur_next < (unsigned)kWave) ? cur_end - cur_next : (unsigned)kWave;
            if constexpr (C::kRing > 0) {
                if (use_ring && ring_end - cur_next < wlen) wlen = ring_end - cur_next;  // hand out only what the ring holds
            }
            const bool valid = (unsigned)lane < wlen;
            const unsigned pos = cur_next + (unsigned)lane;
            const unsigned prs = pos / n_act, psl = pos - prs * n_act;
            bool skipv = false;
            if (early && valid) {
                // ordered early exit: dropped iff a LOWER-index restart of the target has already succeeded
                const int fl = __hip_atomic_load(&args.solved[psl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                skipv = fl > args.restarts - (int)prs;
            }
            const unsigned long long lt = (1ull << lane) - 1ull;
            const unsigned long long avail = __ballot(valid && !skipv);
            const unsigned long long idle = __ballot(!live && q == 0);
            const int n_av = __popcll(avail), n_idle_now = __popcll(idle);
            const int n_take = n_av < n_idle_now ? n_av : n_idle_now;
            const int myrank = __popcll(avail & lt);
            const bool handed = valid && !skipv && myrank < n_take;
            const unsigned long long handed_mask = __ballot(handed);
            // the window is consumed up to the last position handed out (all of it when every position with work found a
            // quad); the rest is looked at again next time
            const unsigned consumed = (n_take == n_av) ? wlen : (unsigned)(64 - __builtin_clzll(handed_mask));
            if (valid && skipv && (unsigned)lane < consumed) {
                const unsigned o = psl * (unsigned)args.restarts + prs;
                item_rec_store_dropped(args.item_rec + o, ST_PREEMPTED);
            }
            int* wp = reinterpret_cast<int*>(lds);  // quad 0's exchange area, dead between rounds: [16] slots, [16] restarts
            if (handed) {
                wp[myrank] = (int)psl;
                wp[16 + myrank] = (int)prs;
                if constexpr (C::kRing > 0) wp[32 + myrank] = (int)(pos - ring_base);
            }
            lds_fence();
            const int qrank = __popcll(idle & ((1ull << (lane & ~3)) - 1ull));  // rank of this quad among the idle ones
            const bool get = !live && qrank < n_take;
            const unsigned sl = get ? (unsigned)wp[qrank] : 0u;
            const unsigned rs = get ? (unsigned)wp[16 + qrank] : 0u;
            const int rslot = (C::kRing > 0 && get) ? wp[32 + qrank] : 0;
            lds_fence();
            cur_next += consumed;
            if (get) {
                const unsigned oidx = sl * (unsigned)args.restarts + rs;
                item = oidx;
                slot = (int)sl;
                restart = (int)rs;
                const int tgt = args.active ? args.active[sl] : (int)sl;
                tcol = args.targets + (int64_t)tgt * 32 + q * 2;
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const int i = 4 * a + q;
                    double xv = 0.0;
                    if (i < C::N) {
                        if (args.x0) {
                            xv = args.x0[(int64_t)oidx * C::N + i];
                        } else {
                            // same Philox stream layout as the fixed-gate path, span tagged with 0x100 (V2), mapped onto the
                            // parameter's start range
                            const double raw = (C::kRing > 0) ? ring[rslot * C::RN + i]
                                                              : x0_philox(args.seed, (uint32_t)(tgt + (int)args.target_base), (uint32_t)rs, (uint32_t)(K | 0x100), (uint32_t)i);
                            const double u = raw * (1.0 / 6.283185307179586476925286766559);
                            xv = fma(u, args.init_hi[i] - args.init_lo[i], args.init_lo[i]);
                        }
                        const double2 b = bnd[i];
                        xv = fmin(fmax(xv, b.x), b.y);
                    }
                    x[a] = xv; g[a] = 0.0; p[a] = 0.0; if constexpr (!FREE) hg[a] = 0.0;
                }
                alpha = 0.0; gp = 0.0; f = 0.0; grow = 1.0; hs1 = 0.0;
                nev = 0; iters = 0; nback = 0; nstall = 0; status = ST_MAXITER;
                scaled = false; fresh = true; live = true; taken = true;
                if (!FREE) { cst[0] = 0.0; cst[3] = 0.0; }  // multiplier estimate; metric restarts + multiplier updates
            }
        }
        if (__any(taken)) H.set_identity_where(q, taken);
        if (!__any(live)) break;
        ++rounds;

        // ---- 2. trial point: projection of x + alpha p onto the box; s = actual step
        double xt[NA], gt[NA];
        double gs = 0.0;  // slope of the linear model along the projected step s = xt - x
        const bool bounded = !FREE && args.bounded != 0;  // wave-uniform
        if (bounded) {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const double2 b = bnd[4 * a + q];
                xt[a] = fmin(fmax(fma(alpha, p[a], x[a]), b.x), b.y);
                gs = fma(g[a], xt[a] - x[a], gs);
            }
            gs = quad_sum(gs);
        } else {
#pragma unroll
            for (int a = 0; a < NA; ++a) xt[a] = fma(alpha, p[a], x[a]);
            gs = alpha * gp;  // s = alpha p
        }
        double ft, Wr[4], Wi[4];
        eval_quad_v2<K, QN, false, GQ>(xt, tcol, args.maps, xq, fh, tbl, q, theta_bits, args.cost_kind, ft, gt, Wr, Wi);
        // Cost constraint (set_constraint, basisv2.py:192-200; the reference hands it to SLSQP): augmented Lagrangian around this
        // box-constrained loop.  The function minimised is  L(x) = loss(x) + rho / 2 max(0, c(x) + mu / rho)^2  with
        // c(x) = w.x - cmax: smooth, same box -- one more quad reduction per evaluation.  When the loop below has converged on L
        // the item's multiplier estimate moves, mu <- max(0, mu + rho c), and the loop goes on from the same point with the same
        // metric until c <= tol and mu c = 0 within tol (first-order multiplier method; the violation shrinks by ~ rho / curvature
        // per update: 5..7 updates to 1e-8 with rho = 30 / max_i w_i^2 -- i.e. rho w.w = 30 for ONE weighted parameter, k times that for k
        // gates of equal weight: the penalty is stiffer, never softer, than that figure).  cmax is handed in lowered by that tolerance: results are feasible.
        if (cons) {
            double ct = 0.0;
#pragma unroll
            for (int a = 0; a < NA; ++a) ct = fma(cw[4 * a + q], xt[a], ct);
            ct = quad_sum(ct) - args.cons_max;
            cst[4] = ft;
            cst[5] = ct;
            const double tpen = args.cons_rho * fmax(0.0, fma(cst[0], 1.0 / args.cons_rho, ct));  // rho max(0, c + mu / rho)
            ft = fma(0.5 * tpen, tpen * (1.0 / args.cons_rho), ft);
#pragma unroll
            for (int a = 0; a < NA; ++a) gt[a] = fma(tpen, cw[4 * a + q], gt[a]);
        }
        const bool active = live;
        const bool finite = isfinite(ft);
        // The quasi-Newton algebra lives in the subspace of the variables that can move: the gradient component of a
        // parameter that is fixed (lo == hi: a bounded-to-a-point Q, the theta / phi of an rz layer) or that sits on a
        // bound with the descent direction pointing out of the box is dropped.  (Left in, such a component -- d loss /
        // d alpha of a fixed gate is large -- enters y = g' - g with s = 0 there and poisons H.)
        if (bounded) {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const double2 b = bnd[4 * a + q];
                const bool blocked = (xt[a] <= b.x && gt[a] > 0.0) || (xt[a] >= b.y && gt[a] < 0.0);
                gt[a] = (finite && !blocked) ? gt[a] : 0.0;
            }
        } else {
#pragma unroll
            for (int a = 0; a < NA; ++a) gt[a] = finite ? gt[a] : 0.0;
        }
        const bool armijo = finite && (ft <= f + kArmijoC1 * gs);
        const bool acc = active && (fresh ? finite : armijo);
        const bool step = acc && !fresh;
        nev += active ? (acc ? 0x100001 : 1) : 0;

        // ---- 3. quasi-Newton update with s = the projected step (zero for quads that do not step).  ONE mat-vec:
        //         q = H g'; with hg = (H + hs1 I) g carried over, u = (H + hs1 I) y = q + hs1 g' - fac hg
        double qv[NA];
        H.matvec(gt, xq32, q, qv);
        double sy = 0.0, yy = 0.0, ss = 0.0, sg = 0.0;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const double sa = FREE ? (step ? alpha : 0.0) * p[a] : (step ? xt[a] - x[a] : 0.0);
            const double ya = gt[a] - g[a];
            sy = fma(sa, ya, sy);
            yy = fma(ya, ya, yy);
            ss = fma(sa, sa, ss);
            sg = fma(sa, gt[a], sg);
        }
        quad_sum2(sy, yy);
        quad_sum2(ss, sg);
        const bool too_short = sy < (1.0 - kWolfeC2) * (-gs);
        const bool curv = step && !too_short && sy > 0.0 && (sy * sy > (kCurvEps * kCurvEps) * (ss * yy));
        const bool first = curv && !scaled;
        scaled = scaled || curv;
        // first update of an item: the initial inverse Hessian (the identity then) scaled by s.y / y.y, as the scalar hs1
        const double fac = first ? (sy * fast_rcp(yy)) : 1.0;
        hs1 = first ? fac - 1.0 : hs1;
        double yu = 0.0;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            qv[a] = fma(hs1, gt[a], qv[a]);              // (H + hs1 I) g'
            const double ua = fma(-fac, HG(a), qv[a]);   // (H + hs1 I) y
            yu = fma(gt[a] - g[a], ua, yu);
        }
        yu = quad_sum(yu);
        const double rho = curv ? fast_rcp(sy) : 0.0;
        const double cf = rho * (1.0 + rho * yu);
        double wg = 0.0;
        {
            float s32[NA], w32[NA], v32[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const double sa = FREE ? (step ? alpha : 0.0) * p[a] : (step ? xt[a] - x[a] : 0.0);
                const double ua = fma(-fac, HG(a), qv[a]);
                const double wa = cf * sa - rho * ua;  // rho = cf = 0 unless curv: w = v = 0, H unchanged
                const double va = -rho * ua;
                s32[a] = (float)sa; w32[a] = (float)wa; v32[a] = (float)va;
                wg = fma(wa, gt[a], wg);
            }
            H.update(s32, w32, v32, xq32, q);
        }
        wg = quad_sum(wg);

        // ---- 4. per-quad state machine
        bool done = false;
        if (acc) {
            nstall = (step && (f - ft) <= kStallDf) ? nstall + 1 : 0;
            f = ft;
            if (cons) { cst[1] = cst[4]; cst[2] = cst[5]; }
            if (step) ++iters;
            nback = 0;
            grow = (step && too_short) ? fmin(grow * kGrowFactor, kGrowMax) : 1.0;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const double sa = FREE ? (step ? alpha : 0.0) * p[a] : (step ? xt[a] - x[a] : 0.0);
                const double ua = fma(-fac, HG(a), qv[a]);
                const double va = -rho * ua;
                x[a] = FREE ? fma(step ? alpha : 0.0, p[a], x[a]) : xt[a];  // (FREE: the same fma that formed the trial point)
                g[a] = gt[a];
                // H' g' = (H + hs1 I) g' + s (w.g') + v (s.g'); the direction is its negative, projected: no component may
                // point out of the box
                const double hn = qv[a] + sa * wg + va * sg;
                if constexpr (!FREE) hg[a] = hn;
                p[a] = -hn;
            }
            if (bounded) {
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const double2 b = bnd[4 * a + q];
                    const bool out = (x[a] <= b.x && HG(a) > 0.0) || (x[a] >= b.y && HG(a) < 0.0);
                    p[a] = out ? 0.0 : p[a];
                }
            }
            if (args.trace_loss) {  // wave-uniform: nothing when off
                if (step && iters <= args.trace_cap) {
                    const int64_t row = (int64_t)item * args.trace_cap + (iters - 1);
                    if (q == 0) args.trace_loss[row] = cons ? cst[1] : f;
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        const int i = 4 * a + q;
                        if (i < C::N) args.trace_x[row * C::N + i] = x[a];
                    }
                }
            }
        } else if (active) {
            if (fresh) {
                f = ft; status = ST_NONFINITE; done = true;
            } else {
                const double denom = 2.0 * (ft - f - gs);
                const double anew = (finite && denom > 0.0 && isfinite(denom)) ? (-gs * alpha * fast_rcp(denom)) : 0.5 * alpha;
                alpha = fmin(fmax(anew, 0.1 * alpha), 0.5 * alpha);
                grow = 1.0;
                ++nback;
            }
        }
        {
            // projected gradient norm: components whose descent direction -g leaves the box do not count
            double m = 0.0;
            if (bounded) {
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const double2 b = bnd[4 * a + q];
                    const bool blocked = (x[a] <= b.x && g[a] > 0.0) || (x[a] >= b.y && g[a] < 0.0);
                    m = max_abs(m, blocked ? 0.0 : g[a]);
                }
            } else {
#pragma unroll
                for (int a = 0; a < NA; ++a) m = max_abs(m, g[a]);
            }
            gnorm = quad_max(m);
        }
        gp = qdot<NA>(g, p);
        const double pp = qdot<NA>(p, p);
        if (acc) {
            alpha = (pp > 1e-300) ? fmin(grow, kStepMax * fast_rsqrt(pp)) : grow;
            if (f < args.stop_loss || gnorm < args.gtol || (!cons && gnorm < args.gtol_far && f > args.far_loss)) { status = ST_CONVERGED; done = true; }
            else if (nstall >= 2) { status = ST_STALLED; done = true; }
            else if (iters >= args.maxiter) { status = ST_MAXITER; done = true; }
        } else if (active && !fresh) {
            if (nback > kMaxBacktrack) { status = (gnorm < kStallGnorm) ? ST_STALLED : ST_LINESEARCH; done = true; }
        }
        fresh = false;
        const bool periodic = step && !done && ((kRestartPeriod & (kRestartPeriod - 1)) == 0 ? ((iters & (kRestartPeriod - 1)) == 0) : (iters % kRestartPeriod == 0));  // periodic restart of the metric (slam_kernels.hpp)
        const bool reset = active && !done && (!(gp < 0.0) || periodic);
        if (__any(reset)) {
            H.set_identity_where(q, reset);
            hs1 = reset ? 0.0 : hs1;
            scaled = periodic ? false : scaled;
            double gg2 = 0.0;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const double2 b = bnd[4 * a + q];
                double pn = -g[a];
                if ((x[a] <= b.x && pn < 0.0) || (x[a] >= b.y && pn > 0.0)) pn = 0.0;
                p[a] = reset ? pn : p[a];
                if constexpr (!FREE) hg[a] = reset ? g[a] : hg[a];
                gg2 = fma(g[a], pn, gg2);
            }
            gg2 = quad_sum(gg2);
            gp = reset ? gg2 : gp;
            alpha = periodic ? ((gg2 < -1e-300) ? fmin(grow, kStepMax * fast_rsqrt(-gg2)) : grow) : alpha;  // (-gg2 = |projected g|^2)
            // nothing left to move along: a KKT point of the box-constrained problem
            if (reset && !(gg2 < 0.0)) { status = ST_CONVERGED; done = true; }
        }
        if (bounded) {
            // A failed line search is not a point of rest.  With bounds the projected step of a quasi-Newton direction need not
            // be a descent direction (a variable about to reach its bound is clamped mid-step, and the metric couples it to the
            // others); the projected step of the steepest-descent direction always is: the metric starts over from the identity
            // and the loop goes on (the variable lands ON its bound and is left out from then on).  With a cost constraint: the
            // loop has come to rest on L(.; mu) -- multiplier update, or the end of the item (no update from a failed search).
            const double nouter = cst[3];  // metric restarts + multiplier updates of the item so far
            const bool lsfail = active && done && status == ST_LINESEARCH && nouter < (double)kMaxMultiplierUpdates;
            bool update = false;
            double mu_next = 0.0;
            if (cons) {
                const bool rest = active && done && (status == ST_CONVERGED || status == ST_STALLED);
                const double mu = cst[0], cc = cst[2];
                mu_next = fmax(0.0, fma(args.cons_rho, cc, mu));
                const bool kkt = cc <= args.cons_tol && (mu_next == 0.0 || fabs(cc) <= args.cons_tol);
                update = rest && !kkt && nouter < (double)kMaxMultiplierUpdates;
                mu_next = update ? mu_next : mu;
            }
            if (lsfail || update) {
                if (cons) cst[0] = mu_next;
                cst[3] = nouter + 1.0;
                hs1 = lsfail ? 0.0 : hs1;
                scaled = lsfail ? false : scaled;
                done = false; fresh = true; alpha = 0.0; nstall = 0; nback = 0; status = ST_MAXITER;  // re-evaluated at x (with the new mu)
            }
            if (__any(lsfail)) H.set_identity_where(q, lsfail);
            // with a constraint, what leaves the kernel is the plain loss -- of a feasible point: an item that ran out of multiplier
            // updates with the constraint still violated reports +inf and never wins its target
            if (cons) f = (active && done && status != ST_NONFINITE) ? ((cst[2] <= args.cons_tol) ? cst[1] : INFINITY) : f;
        }
        if (active && done) {
            if (early && f < args.exit_loss && q == 0)
                __hip_atomic_fetch_max(&args.solved[slot], args.restarts - restart, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (q == 0) item_rec_store(args.item_rec + item, f, iters, status, nev & 0xFFFFF, (int)((unsigned)nev >> 20));
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int i = 4 * a + q;
                if (i < C::N) args.item_x[(int64_t)item * C::N + i] = x[a];
            }
            live = false;
            alpha = 0.0;
#pragma unroll
            for (int a = 0; a < NA; ++a) p[a] = 0.0;
        }
    }
    if (lane == 0 && rounds) atomicAdd(&args.ctl->rounds, (unsigned long long)rounds);
}

#undef HG

}  // namespace slamdev
