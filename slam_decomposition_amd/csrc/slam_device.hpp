// slam_device.hpp -- device-side building blocks of the SLAM template optimizer for gfx950.
//
// Work decomposition:
//   * a QUAD (4 adjacent lanes) owns one (target, seed) work item; a 64-lane wavefront
//     carries 16 quads -- with R = 16 restarts that is one target per wavefront;
//   * lane c of the quad owns COLUMN c of the running 4x4 product: columns of
//     W = K_k G_k ... G_1 K_0 evolve independently under left multiplication, so the
//     forward chain needs no cross-lane traffic; rows of (z T^+)(suffix) evolve
//     independently under right multiplication, so the backward chain needs none either;
//   * the only exchanges are quad reductions (DPP quad_perm shuffles) and small
//     transposes through a wave-private LDS exchange area (trig table, gradient partials, fp32 broadcasts);
//   * the 2Q gate matrices are wave-uniform device data: read with scalar loads, they sit in SGPRs and feed
//     the fp64 FMAs as scalar operands (no VGPR, no LDS read);
//   * memory latencies are kept off the critical path by hand: every LDS / scalar read is requested half a layer
//     to a layer before its first use (sched_barriers keep the compiler from sinking it back);
//   * the n x n inverse-Hessian approximation (the quasi-Newton metric, a preconditioner) is kept
//     in fp32 VGPRs as packed symmetric 4x4 blocks -- lane q holds row q of every upper-triangle
//     block -- and is applied / updated with packed fp32 FMAs.  Loss, gradient, parameters,
//     steps and all scalars of the iteration are fp64.
//
// Reference behaviour being computed (paths relative to the reference checkout):
//   CircuitTemplate.eval            src/slam/basis.py:102-104,124-169
//   BasicCost.unitary_fidelity      src/slam/cost_function.py:140-145
//   scipy BFGS restart loop         src/slam/optimizer.py:253-295
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slam_sincos.hpp"

namespace slamdev {

constexpr int kQuadsPerWave = 16;
constexpr int kWave = 64;
// row stride (in double2) of the stored column vectors: 64 lanes + 1 pad slot, so that consecutive rows start
// 4 banks apart and the gradient gather (different rows per lane) does not pile onto one bank column
constexpr int kRow = kWave + 1;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// PSQ: the quad's exchange area also holds the four lanes' gradient partials of every parameter (structured gate
// classes; the dense class keeps the pair sums in the dead trig entries: its stored vectors leave no LDS for more)
template <int K, bool PSQ = false>
struct Cfg {
    static constexpr int L = K + 1;                    // 1Q layers
    static constexpr int N = 6 * L;                    // parameters (basis.py:152-169)
    static constexpr int NA = (N + 3) / 4;             // parameter slots per lane
    static constexpr int NP = NA * 4;                  // padded parameter count
    static constexpr int NBLK = NA * (NA + 1) / 2;     // upper-triangle 4x4 blocks of H
    // doubles per quad in the exchange area.  Users: trig table (2N doubles); the gradient partials (PSQ: four planes,
    // below; else pair sums parked in the layers' dead trig entries); fp32 mat-vec broadcast (4 NA floats) + transposed
    // partial sums (16 (NA - 1) floats); fp32 rank-2 update broadcasts (8 NA floats); the start points dealt out at a refill.
    // The fp32 exchanges of the quasi-Newton algebra (h_matvec / h_update: 20 NA - 16 floats per quad, live only BETWEEN
    // evaluations) overlay the whole wave's exchange area with their OWN quad stride, FSTRIDE floats == 4 (mod 32): their
    // traffic is mostly ds_write_b32 of 4 consecutive dwords per quad, serviced in groups of 32 lanes = 8 quads on 32
    // banks -- with the double area's stride (== 16 mod 32 dwords) the 8 quads fell on 2 bank groups, a 4-way conflict on
    // every such store (31 per round at k = 2: most of the measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 33 %).
    static constexpr int FNEED = 20 * NA - 16;
    static constexpr int FSTRIDE = (FNEED - 4 + 31) / 32 * 32 + 4;
    // trig table [0, 2N).  PSQ: four planes of gradient partials, plane q' (the partials computed by lane q') at
    // PS0 + q' PSP, parameter i at [i] -- plane stride == 1 (mod 4) doubles, quad stride an odd multiple of 32 bytes: the 16
    // lanes of a ds_write_b64 group and the 32 lanes of a ds_read_b64 group then hit distinct banks (measured: the
    // parameter-major layout [i][q'] read with ds_read_b128 put SQ_LDS_BANK_CONFLICT at 32 % of SQ_LDS_IDX_ACTIVE).
    // The planes start over the TOP layer's trig entries, which are dead once the forward pass has loaded them (K <= 3).
    static constexpr int PSP = (N + 3) / 4 * 4 + 1;
    // PSQ: the trig table of quad number 4g + w starts (0, 4, 2, 6)[w] doubles into the quad's area (TOFF = the largest).  A
    // ds_write_b128 is serviced in groups of 8 lanes = 2 quads on a 128-byte bank window, each quad storing 64 contiguous
    // bytes, a ds_read_b128 in groups of 16 lanes = the quads {0,3,5,6}, {1,2,4,7}, ... on a 256-byte window, each quad one
    // broadcast 16-byte slot.  With a uniform quad stride == 32 (mod 128) bytes the two store footprints overlap by half
    // (2-way conflict on every trig store: 9 % of all LDS cycles at k = 1); these offsets put the stores 64 bytes apart and
    // keep the four broadcast slots of a read group distinct (exhaustive search over offsets per quad mod 4 with the bank
    // model of MI355X_MICROARCH.md; an offset for the odd quads alone moved the conflicts to the reads: measured 16 -> 26 %)
    static constexpr int TOFF = PSQ ? 6 : 0;
    static constexpr int PS0 = 12 * K + TOFF;
    static constexpr int XNEED_D = PSQ ? PS0 + 3 * PSP + N : 2 * N;
    static constexpr int XNEED_F = (kQuadsPerWave * FSTRIDE + 2 * kQuadsPerWave - 1) / (2 * kQuadsPerWave);  // doubles per quad so that the float overlay fits
    static constexpr int XNEED = XNEED_D > XNEED_F ? XNEED_D : XNEED_F;
    // quad stride of the double area, == 4 (mod 16) doubles = 8 (mod 32) dwords: the four quads of a ds_write_b64 group
    // (pair sums, two adjacent doubles per quad) and of a ds_read_b128 group (trig entries, one broadcast double2 per quad)
    // land on different banks; the per-lane double2 gather of the gradient keeps a partial 2-way overlap between two quads
    // (== 4 (mod 8) is enough for that: 4 and 12 (mod 16) both put four consecutive quads on four different 32-byte segments)
    // PSQ: == 4 (mod 16), i.e. 32 (mod 128) bytes, which the trig-store offset above assumes
    static constexpr int XSTRIDE = PSQ ? (XNEED - 4 + 15) / 16 * 16 + 4 : (XNEED - 4 + 7) / 8 * 8 + 4;
    static constexpr int LDS_XCHG = kQuadsPerWave * XSTRIDE;
    static_assert(kQuadsPerWave * FSTRIDE * 4 <= LDS_XCHG * 8, "float overlay must fit the exchange area");
    static constexpr int LDS_FH = 2 * K * 4 * kRow * 2;  // 2K column vectors x 4 rows x (64 lanes + pad) x (re,im)
    static constexpr int LDS_DOUBLES = LDS_XCHG + LDS_FH;
    // LEAN layout (structured gate classes): only the K layer outputs h_j are stored; the layer inputs
    // f_j = G_j h_{j-1} are recomputed in the backward pass (0 / 8 / 16 products for CX / XRI / XGEN gates)
    // (h_0 .. h_{K-2}: the last one, h_{K-1}, stays in registers)
    static constexpr int LDS_FH_LEAN = (K - 1) * 4 * kRow * 2;
    static constexpr int LDS_DOUBLES_LEAN = LDS_XCHG + LDS_FH_LEAN;
};

template <int K, int GC>
__host__ __device__ constexpr bool lean_layout() { return GC != 0; }  // every class but GC_DENSE
// the quad partial-sum area (Cfg<K, true>): structured classes up to span 3 (beyond, its LDS would cost a wavefront per CU)
template <int K, int GC>
__host__ __device__ constexpr bool psq_layout() { return lean_layout<K, GC>() && K <= 3; }
template <int K, int GC>
__host__ __device__ constexpr int lds_work_doubles() {
    return lean_layout<K, GC>() ? Cfg<K, psq_layout<K, GC>()>::LDS_DOUBLES_LEAN : Cfg<K, false>::LDS_DOUBLES;
}
// ... plus the wave's copy of the 32-entry (cos, sin) table of sincos_tbl, after the working areas
// ... plus the convergence thresholds of the launch (gtol, stop_loss, gtol_far, far_loss): tested every round, and as
// kernel arguments they were re-read from the kernarg segment right in front of the test (a scalar-cache latency exposed
// twice per round); an LDS read is requested ahead and counted exactly by lgkmcnt
constexpr int kColdDoubles = 4;
constexpr int kSincosLdsDoubles = 2 * kSincosTableDoubles;  // the LDS copy of the sincos table covers the whole circle
// Span 1, structured gate classes: the START POINTS of the next kSeedRing queue positions of the wavefront's chunk are generated
// together -- one Philox block per lane, kSeedRing N / 2 = 48 lanes busy -- and parked in the SPARE doubles [70, 82) of the exchange
// areas of quads 8 .. 15 (XNEED = 69 of XSTRIDE = 84 at span 1; the fp32 overlay of the quasi-Newton algebra ends inside quad 6's area):
// no LDS is added, a k = 1 wavefront keeps sharing its SIMD's CU with a k = 3 one (4 x (11.5 + 27.7) KB = 157 of 160 KB).  A refill then
// only copies.  Round 4 ran the blocks at the refill for the quads that took an item -- 12 .. 18 of 64 lanes busy for ~110 vector
// instructions (20 quarter-rate 32 x 32 -> 64-bit multiplies among them) every four or five rounds.  (A 64-position ring in 6 KB of
// extra LDS ran the k = 1 launch 3.4 % faster alone -- 8.73 -> 8.43 ms, 885 -> 860 vector instructions per round -- and the driver's
// command 0.3 % SLOWER: with 17.5 KB per k = 1 wavefront the k = 1 / k = 3 pairing no longer fits four times into a CU.)
constexpr int kSeedRing = 8;
constexpr int kSeedRingOff = 70;
template <int K, int GC>
__host__ __device__ constexpr bool seed_ring() { return K == 1 && psq_layout<K, GC>(); }
template <int K, int GC>
__host__ __device__ constexpr int lds_doubles() { return lds_work_doubles<K, GC>() + kSincosLdsDoubles + kColdDoubles; }

__device__ const double kSinCosTable[kSincosTableDoubles] = SLAM_SINCOS_TABLE;

// the wave's LDS copy covers the whole circle: entry j = (cos, sin)(j pi / 32), j = 0..63 (second half = minus the first), so
// the lookup needs no sign fix-up (three vector instructions per sincos).  One double2 per lane; the caller fences
__device__ __forceinline__ void load_sincos_table(double2* tbl, int lane) {
    const double2 t = reinterpret_cast<const double2*>(kSinCosTable)[lane & 31];
    tbl[lane] = (lane & 32) ? make_double2(-t.x, -t.y) : t;
}

// Gate matrices G_1..G_K of the launch: K x 32 doubles, row-major (re, im), in device memory.
// They are wave-uniform, so they are read through a constant-address-space pointer with scalar
// loads (s_load_dwordx16) and feed the fp64 FMAs as SGPR operands: no VGPRs, no LDS reads.
typedef const __attribute__((address_space(4))) double* gate_ptr;

__device__ __forceinline__ gate_ptr gate_matrix(const double* gates, int j) {
    // opaque to the optimiser: without it the loop-invariant loads of all K matrices are hoisted
    // out of the iteration loop, 64 K SGPRs do not fit, and every use becomes a v_readlane.
    unsigned long long a = (unsigned long long)(gates + 32 * j);
    asm volatile("" : "+s"(a));
    return (gate_ptr)a;
}

// ---------------------------------------------------------------------------------
// quad (4-lane) cross-lane primitives: DPP quad_perm on the two 32-bit halves
// ---------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
// quad_perm [1,0,3,2] = 0xB1 (xor 1), [2,3,0,1] = 0x4E (xor 2), [0,0,0,0] = 0x00 (broadcast lane 0)
// (the same permutation through the LDS unit -- ds_swizzle_b32 in quad-perm mode, no vector-ALU slot -- was measured in
// round 3: -40 vector instructions per round, +3 % time: HISTORY.md)
template <int CTRL>
__device__ __forceinline__ double quad_shuffle(double v) { return dpp_f64<CTRL>(v); }
__device__ __forceinline__ double quad_sum(double v) {
    v += quad_shuffle<0xB1>(v);
    v += quad_shuffle<0x4E>(v);
    return v;
}
// two sums at once: both values' shuffles of a stage are in flight together (one LDS round trip per stage, not two)
__device__ __forceinline__ void quad_sum2(double& a, double& b) {
    const double a1 = quad_shuffle<0xB1>(a), b1 = quad_shuffle<0xB1>(b);
    a += a1; b += b1;
    const double a2 = quad_shuffle<0x4E>(a), b2 = quad_shuffle<0x4E>(b);
    a += a2; b += b2;
}
// max(|a|, |b|) in ONE instruction: fmax() canonicalises its operands first (IEEE sNaN semantics), an extra v_max_f64
// x, x, x each; the callers' values are finite by construction
__device__ __forceinline__ double max_abs(double a, double b) {
    double m;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ double quad_max(double v) {  // v >= 0
    v = max_abs(v, quad_shuffle<0xB1>(v));
    v = max_abs(v, quad_shuffle<0x4E>(v));
    return v;
}

// All LDS traffic is wave-private (one wavefront per workgroup) and the LDS unit executes one
// wave's instructions in order, so a write is visible to every later read of the same wave without
// any wait: only the compiler has to be kept from reordering.  (__syncthreads() here would add
// s_waitcnt vmcnt(0): every fence would then wait for the global result stores and target loads.)
__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------
// Philox4x32-10 (must match oracle/slam_oracle.py:philox4x32 / x0_philox bit for bit)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// two Philox words -> x0 ~ U[0, 2pi) with 53 random bits
__device__ __forceinline__ double x0_from_words(uint32_t a, uint32_t b) {
    const uint64_t m = ((uint64_t)(a >> 5) << 26) + (uint64_t)(b >> 6);
    return (double)m * (1.0 / 9007199254740992.0) * 6.283185307179586476925286766559;
}
// x0[i]: parameter i of (seed, target index, restart, span k); words (0,1) of the Philox block of
// parameter pair i >> 1 for even i, words (2,3) for odd i
__device__ __forceinline__ double x0_philox(uint64_t seed, uint32_t target, uint32_t restart, uint32_t k,
                                            uint32_t i) {
    uint32_t w[4];
    philox4x32_10(i >> 1, restart, target, k, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    return (i & 1) ? x0_from_words(w[2], w[3]) : x0_from_words(w[0], w[1]);
}

// ---------------------------------------------------------------------------------
// U3 action on a pair of complex amplitudes
// U3 = diag(1, e^{i phi}) R(theta/2) diag(1, e^{i lam}),  R = [[c, -s], [s, c]]
// ---------------------------------------------------------------------------------
struct U3t {
    double c, s;    // cos(theta/2), sin(theta/2)
    double cp, sp;  // cos(phi), sin(phi)
    double cl, sl;  // cos(lam), sin(lam)
};

// trig table entry i of this quad: (cos, sin) of the (half-)angle of parameter i
__device__ __forceinline__ U3t load_u3(const double* xq, int base_param) {
    const double2* t = reinterpret_cast<const double2*>(xq) + base_param;
    const double2 a = t[0], b = t[1], c = t[2];
    U3t u;
    u.c = a.x; u.s = a.y; u.cp = b.x; u.sp = b.y; u.cl = c.x; u.sl = c.y;
    return u;
}

// column action: (f0, f1)^T <- U3 (f0, f1)^T
__device__ __forceinline__ void u3_col(const U3t& t, double& f0r, double& f0i, double& f1r, double& f1i) {
    const double g1r = t.cl * f1r - t.sl * f1i;
    const double g1i = t.cl * f1i + t.sl * f1r;
    const double y0r = t.c * f0r - t.s * g1r;
    const double y0i = t.c * f0i - t.s * g1i;
    const double tr = t.s * f0r + t.c * g1r;
    const double ti = t.s * f0i + t.c * g1i;
    f0r = y0r; f0i = y0i;
    f1r = t.cp * tr - t.sp * ti;
    f1i = t.cp * ti + t.sp * tr;
}

// column e of U3, e = 0 or 1 given as the real pair (f0, f1) = (1 - e, e): the column action on a real unit vector
// (12 operations instead of 16; bit for bit what u3_col computes from (f0, 0, f1, 0))
__device__ __forceinline__ void u3_unit(const U3t& t, double f0, double f1, double& y0r, double& y0i, double& y1r, double& y1i) {
    const double g1r = t.cl * f1, g1i = t.sl * f1;
    y0r = t.c * f0 - t.s * g1r;
    y0i = -(t.s * g1i);
    const double tr = t.s * f0 + t.c * g1r;
    const double ti = t.c * g1i;
    y1r = t.cp * tr - t.sp * ti;
    y1i = t.cp * ti + t.sp * tr;
}

// u3_unit that also returns t = (s f0 + c e^{i lam} f1), the second component before the phi phase.  With it the column of
// 2 dU3/dtheta (= U3 with (c, s) -> (-s, c)) applied to the same unit vector is (-t, e^{i phi} y0): no second rotation.
__device__ __forceinline__ void u3_unit_t(const U3t& t, double f0, double f1, double& y0r, double& y0i, double& y1r, double& y1i,
                                          double& tr, double& ti) {
    const double g1r = t.cl * f1, g1i = t.sl * f1;
    y0r = t.c * f0 - t.s * g1r;
    y0i = -(t.s * g1i);
    tr = t.s * f0 + t.c * g1r;
    ti = t.c * g1i;
    y1r = t.cp * tr - t.sp * ti;
    y1i = t.cp * ti + t.sp * tr;
}
// Layer 0 seen from its OUTPUT side.  Its input is the unit vector e_q, so K_0 e_q = a (x) b with a = A e_{q1}, b = B e_{q0}, and
// the derivative with respect to a parameter of one gate X (column x = X e, w = the backward vector contracted with the OTHER
// gate's column) is Re(w . dx):
//   d/dphi x = (0, i x1)                 ->  -Im(w1 x1)
//   d/dlam x = i f1 x   (f1 = the bit)   ->  -f1 Im(w0 x0 + w1 x1)
//   d/dtheta x = 1/2 (-t, e^{i phi} x0)  ->  1/2 ( Re(w1 e^{i phi} x0) - Re(w0 t) )
// 42 operations per gate including the column itself, against ~60 for the row action with its intermediates.
__device__ __forceinline__ void l0_gate_partials(const U3t& g, double f0, double f1, double w0r, double w0i, double w1r, double w1i,
                                                 double x0r, double x0i, double x1r, double x1i, double tr, double ti, double& dth, double& dph,
                                                 double& dla) {
    const double d1r = g.cp * x0r - g.sp * x0i, d1i = g.cp * x0i + g.sp * x0r;  // e^{i phi} x0
    dth = 0.5 * ((w1r * d1r - w1i * d1i) - (w0r * tr - w0i * ti));
    const double im1 = w1r * x1i + w1i * x1r;
    dph = -im1;
    dla = -f1 * (im1 + (w0r * x0i + w0i * x0r));
}

// row action: (u0, u1) <- (u0, u1) U3
__device__ __forceinline__ void u3_row(const U3t& t, double& u0r, double& u0i, double& u1r, double& u1i) {
    const double g1r = t.cp * u1r - t.sp * u1i;
    const double g1i = t.cp * u1i + t.sp * u1r;
    const double n0r = t.c * u0r + t.s * g1r;
    const double n0i = t.c * u0i + t.s * g1i;
    const double tr = t.c * g1r - t.s * u0r;
    const double ti = t.c * g1i - t.s * u0i;
    u0r = n0r; u0i = n0i;
    u1r = t.cl * tr - t.sl * ti;
    u1i = t.cl * ti + t.sl * tr;
}

// row action that also returns t = (c e^{i phi} u1 - s u0), the second component after the rotation and before the
// lambda phase: with n0 (= the new u0) it gives the theta derivative of the gate applied LAST to the row vector as
//   dU3/dtheta = 1/2 D_phi R J D_lam,  J = [[0,-1],[1,0]]   =>   Re(u dU f) = 1/2 Re( t f0 - n0 e^{i lam} f1 )
// (8 real operations per pair instead of 13 for the generic form below).
__device__ __forceinline__ void u3_row_t(const U3t& t, double& u0r, double& u0i, double& u1r, double& u1i, double& tr,
                                         double& ti) {
    const double g1r = t.cp * u1r - t.sp * u1i;
    const double g1i = t.cp * u1i + t.sp * u1r;
    const double n0r = t.c * u0r + t.s * g1r;
    const double n0i = t.c * u0i + t.s * g1i;
    tr = t.c * g1r - t.s * u0r;
    ti = t.c * g1i - t.s * u0i;
    u0r = n0r; u0i = n0i;
    u1r = t.cl * tr - t.sl * ti;
    u1i = t.cl * ti + t.sl * tr;
}
// 1/2 Re( t f0 - n0 (e^{i lam} f1) ) and the lambda derivative -Im( t e^{i lam} f1 ) of the same gate
__device__ __forceinline__ void dtheta_dlam_last(const U3t& g, double n0r, double n0i, double tr, double ti, double f0r, double f0i,
                                                 double f1r, double f1i, double& dth, double& dlam) {
    const double gr = g.cl * f1r - g.sl * f1i, gi = g.cl * f1i + g.sl * f1r;  // e^{i lam} f1
    dth = 0.5 * ((tr * f0r - ti * f0i) - (n0r * gr - n0i * gi));
    dlam = -(tr * gi + ti * gr);
}

// the same for a REAL input pair (f0, f1) -- layer 0, whose input is the unit vector e_q (8 operations instead of 13)
__device__ __forceinline__ void dtheta_dlam_last_real(const U3t& g, double n0r, double n0i, double tr, double ti, double f0, double f1,
                                                      double& dth, double& dlam) {
    const double gr = g.cl * f1, gi = g.sl * f1;  // e^{i lam} f1
    dth = 0.5 * (tr * f0 - (n0r * gr - n0i * gi));
    dlam = -(tr * gi + ti * gr);
}

// Row action of the gate applied FIRST to the row vector u (u = the backward vector at the layer's output), plus
// that gate's theta derivative taken on the output side:  (dU/dtheta) U^-1 = 1/2 D_phi J D_phi^-1, so with h = the
// layer's output column   Re(u dU f) = 1/2 Re( (e^{i phi} u1) h0 - e^{-i phi} (u0 h1) )  -- e^{i phi} u1 is the first
// thing the row action computes anyway.  dth accumulates 2x the derivative (the caller halves once).
__device__ __forceinline__ void u3_row_dtheta_first(const U3t& t, double& u0r, double& u0i, double& u1r, double& u1i, double h0r,
                                                    double h0i, double h1r, double h1i, double& dth2) {
    const double g1r = t.cp * u1r - t.sp * u1i;
    const double g1i = t.cp * u1i + t.sp * u1r;
    const double pr = u0r * h1r - u0i * h1i, pi = u0r * h1i + u0i * h1r;  // u0 h1
    dth2 += (g1r * h0r - g1i * h0i) - (t.cp * pr + t.sp * pi);
    const double n0r = t.c * u0r + t.s * g1r;
    const double n0i = t.c * u0i + t.s * g1i;
    const double tr = t.c * g1r - t.s * u0r;
    const double ti = t.c * g1i - t.s * u0i;
    u0r = n0r; u0i = n0i;
    u1r = t.cl * tr - t.sl * ti;
    u1i = t.cl * ti + t.sl * tr;
}

// generic theta derivative on the input side: 1/2 Re( ut1 e^{-i lam} f0 - ut0 e^{i lam} f1 ) with ut = u K (the fully
// transformed row vector) and f = the layer's input  (dU/dtheta = 1/2 U M, M = [[0, -e^{i lam}], [e^{-i lam}, 0]])
__device__ __forceinline__ double dtheta_pair(const U3t& t, double ut0r, double ut0i, double ut1r, double ut1i,
                                              double f0r, double f0i, double f1r, double f1i) {
    const double ar = ut1r * f0r - ut1i * f0i, ai = ut1r * f0i + ut1i * f0r;
    const double br = ut0r * f1r - ut0i * f1i, bi = ut0r * f1i + ut0i * f1r;
    return 0.5 * ((t.cl * ar + t.sl * ai) - (t.cl * br - t.sl * bi));
}

__device__ __forceinline__ double im_mul(double ar, double ai, double br, double bi) { return ar * bi + ai * br; }

// 1/x and 1/sqrt(x) for normal positive x: hardware seed + two Newton steps (<= 1 ulp typical).  The
// IEEE-correct division the compiler emits (v_div_scale / v_div_fmas / v_div_fixup) is ~3x the instructions
// and its special-case handling is not needed: every caller guards x <= 0 / non-finite by a select.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = fma(0.5 * y, fma(-x * y, y, 1.0), y);
    y = fma(0.5 * y, fma(-x * y, y, 1.0), y);
    return y;
}

// "these values are needed here": pins the s_waitcnt of the LDS reads that produced them to this point of the program
__device__ __forceinline__ void u3_arrived(const U3t& a, const U3t& b) {
    asm volatile("" ::"v"(a.c), "v"(a.s), "v"(a.cp), "v"(a.sp), "v"(a.cl), "v"(a.sl), "v"(b.c), "v"(b.s), "v"(b.cp), "v"(b.sp),
                 "v"(b.cl), "v"(b.sl));
}
__device__ __forceinline__ void vec_arrived(const double (&r)[4], const double (&i)[4]) {
    asm volatile("" ::"v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(i[0]), "v"(i[1]), "v"(i[2]), "v"(i[3]));
}

// cold path, kept out of line so that its register appetite (ocml's Payne-Hanek reduction) does not
// shape the register allocation of the optimizer loop
__device__ __attribute__((noinline)) void sincos_slow(double x, double* s, double* c) { sincos(x, s, c); }

__device__ __forceinline__ void sincos_any(double x, const double2* tbl, double& s, double& c) {
    if (__builtin_expect(fabs(x) < kSincosTblLimit, 1)) {
        sincos_tbl(x, tbl, s, c);
    } else {
        sincos_slow(x, &s, &c);  // huge arguments, NaN/inf propagate
    }
}

// ---------------------------------------------------------------------------------
// 2Q gate application, specialised by the structure of the launch's gates (classified on the host,
// slam_hip.hip:classify_gates).  All classes read the same dense row-major (re, im) layout; a structured
// class simply never touches the entries that are zero by structure.
//   GC_DENSE  any 4x4 matrix                                            16 complex mul-adds
//   GC_XGEN   X-shaped: two complex 2x2 blocks on index pairs (0,3), (1,2)       8
//   GC_XRI    X-shaped with real diagonal and imaginary off-diagonal block entries:
//             RiSwapGate(alpha) (sqrt-iSWAP, iSWAP), canonical gates with c3 = 0 (B),
//             ConversionGainGate with zero phases                                 8 real-scalar products
//   GC_CX     qiskit CXGate: a permutation of amplitudes 1 <-> 3                 0
// ---------------------------------------------------------------------------------
//   GC_XRI1   GC_XRI whose block on (0,3) is the identity: RiSwapGate(alpha) = sqrt-iSWAP, iSWAP  4 real-scalar products
enum : int { GC_DENSE = 0, GC_XGEN = 1, GC_XRI = 2, GC_CX = 3, GC_XRI1 = 4 };
constexpr int kGateClasses = 5;

#define SLAM_GRE(r, s) G[((r) * 4 + (s)) * 2]
#define SLAM_GIM(r, s) G[((r) * 4 + (s)) * 2 + 1]

// (a, b) <- [[m00, m01], [m10, m11]] (a, b), complex entries read from G at (r0,r0) (r0,r1) (r1,r0) (r1,r1)
template <int R0, int R1>
__device__ __forceinline__ void block_col_gen(gate_ptr G, double (&Fr)[4], double (&Fi)[4]) {
    const double ar = Fr[R0], ai = Fi[R0], br = Fr[R1], bi = Fi[R1];
    Fr[R0] = fma(SLAM_GRE(R0, R0), ar, fma(-SLAM_GIM(R0, R0), ai, fma(SLAM_GRE(R0, R1), br, -SLAM_GIM(R0, R1) * bi)));
    Fi[R0] = fma(SLAM_GRE(R0, R0), ai, fma(SLAM_GIM(R0, R0), ar, fma(SLAM_GRE(R0, R1), bi, SLAM_GIM(R0, R1) * br)));
    Fr[R1] = fma(SLAM_GRE(R1, R0), ar, fma(-SLAM_GIM(R1, R0), ai, fma(SLAM_GRE(R1, R1), br, -SLAM_GIM(R1, R1) * bi)));
    Fi[R1] = fma(SLAM_GRE(R1, R0), ai, fma(SLAM_GIM(R1, R0), ar, fma(SLAM_GRE(R1, R1), bi, SLAM_GIM(R1, R1) * br)));
}
// (a, b) <- (a, b) [[m00, m01], [m10, m11]]   (row vector times block)
template <int R0, int R1>
__device__ __forceinline__ void block_row_gen(gate_ptr G, double (&Ur)[4], double (&Ui)[4]) {
    const double ar = Ur[R0], ai = Ui[R0], br = Ur[R1], bi = Ui[R1];
    Ur[R0] = fma(ar, SLAM_GRE(R0, R0), fma(-ai, SLAM_GIM(R0, R0), fma(br, SLAM_GRE(R1, R0), -bi * SLAM_GIM(R1, R0))));
    Ui[R0] = fma(ar, SLAM_GIM(R0, R0), fma(ai, SLAM_GRE(R0, R0), fma(br, SLAM_GIM(R1, R0), bi * SLAM_GRE(R1, R0))));
    Ur[R1] = fma(ar, SLAM_GRE(R0, R1), fma(-ai, SLAM_GIM(R0, R1), fma(br, SLAM_GRE(R1, R1), -bi * SLAM_GIM(R1, R1))));
    Ui[R1] = fma(ar, SLAM_GIM(R0, R1), fma(ai, SLAM_GRE(R0, R1), fma(br, SLAM_GIM(R1, R1), bi * SLAM_GRE(R1, R1))));
}
// the same with block = [[d0, i o01], [i o10, d1]], d*, o* real
template <int R0, int R1>
__device__ __forceinline__ void block_col_ri(gate_ptr G, double (&Fr)[4], double (&Fi)[4]) {
    const double d0 = SLAM_GRE(R0, R0), o01 = SLAM_GIM(R0, R1), o10 = SLAM_GIM(R1, R0), d1 = SLAM_GRE(R1, R1);
    const double ar = Fr[R0], ai = Fi[R0], br = Fr[R1], bi = Fi[R1];
    Fr[R0] = fma(d0, ar, -o01 * bi);
    Fi[R0] = fma(d0, ai, o01 * br);
    Fr[R1] = fma(d1, br, -o10 * ai);
    Fi[R1] = fma(d1, bi, o10 * ar);
}
template <int R0, int R1>
__device__ __forceinline__ void block_row_ri(gate_ptr G, double (&Ur)[4], double (&Ui)[4]) {
    const double d0 = SLAM_GRE(R0, R0), o01 = SLAM_GIM(R0, R1), o10 = SLAM_GIM(R1, R0), d1 = SLAM_GRE(R1, R1);
    const double ar = Ur[R0], ai = Ui[R0], br = Ur[R1], bi = Ui[R1];
    Ur[R0] = fma(d0, ar, -o10 * bi);
    Ui[R0] = fma(d0, ai, o10 * br);
    Ur[R1] = fma(d1, br, -o01 * ai);
    Ui[R1] = fma(d1, bi, o01 * ar);
}

// F <- G F
template <int GC>
__device__ __forceinline__ void gate_col(gate_ptr G, double (&Fr)[4], double (&Fi)[4]) {
    if constexpr (GC == GC_CX) {
        const double tr = Fr[1], ti = Fi[1];
        Fr[1] = Fr[3]; Fi[1] = Fi[3];
        Fr[3] = tr; Fi[3] = ti;
    } else if constexpr (GC == GC_XRI1) {
        block_col_ri<1, 2>(G, Fr, Fi);
    } else if constexpr (GC == GC_XRI) {
        block_col_ri<0, 3>(G, Fr, Fi);
        block_col_ri<1, 2>(G, Fr, Fi);
    } else if constexpr (GC == GC_XGEN) {
        block_col_gen<0, 3>(G, Fr, Fi);
        block_col_gen<1, 2>(G, Fr, Fi);
    } else {
        double nr[4], ni[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double ar = 0.0, ai = 0.0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double gx = SLAM_GRE(r, s), gy = SLAM_GIM(r, s);
                ar = fma(gx, Fr[s], fma(-gy, Fi[s], ar));
                ai = fma(gx, Fi[s], fma(gy, Fr[s], ai));
            }
            nr[r] = ar; ni[r] = ai;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { Fr[r] = nr[r]; Fi[r] = ni[r]; }
    }
}

// u <- u G
template <int GC>
__device__ __forceinline__ void gate_row(gate_ptr G, double (&Ur)[4], double (&Ui)[4]) {
    if constexpr (GC == GC_CX) {
        const double tr = Ur[1], ti = Ui[1];
        Ur[1] = Ur[3]; Ui[1] = Ui[3];
        Ur[3] = tr; Ui[3] = ti;
    } else if constexpr (GC == GC_XRI1) {
        block_row_ri<1, 2>(G, Ur, Ui);
    } else if constexpr (GC == GC_XRI) {
        block_row_ri<0, 3>(G, Ur, Ui);
        block_row_ri<1, 2>(G, Ur, Ui);
    } else if constexpr (GC == GC_XGEN) {
        block_row_gen<0, 3>(G, Ur, Ui);
        block_row_gen<1, 2>(G, Ur, Ui);
    } else {
        double nr[4], ni[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { nr[s] = 0.0; ni[s] = 0.0; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double gx = SLAM_GRE(r, s), gy = SLAM_GIM(r, s);
                nr[s] = fma(Ur[r], gx, fma(-Ui[r], gy, nr[s]));
                ni[s] = fma(Ur[r], gy, fma(Ui[r], gx, ni[s]));
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) { Ur[s] = nr[s]; Ui[s] = ni[s]; }
    }
}

// The entries of one gate matrix that a layer needs, fetched ONCE per use site group and early: for the XRI class the 8
// real numbers are loaded into SGPRs at the top of the layer (forward) / once per backward layer for both of its uses
// (f_j = G_j h_{j-1} and u <- u G_j), so the scalar-cache latency (~200 cycles, previously exposed three to five times
// per layer right in front of the first use) runs under the layer's 1Q-gate arithmetic.  The other classes keep the
// load-at-use form (32 / 64 SGPRs per matrix do not fit next to the loop's other scalars).
template <int GC>
struct GateRegs {
    gate_ptr G;
};
template <>
struct GateRegs<GC_XRI> {
    double a_d0, a_o01, a_o10, a_d1;  // block on index pair (0, 3)
    double b_d0, b_o01, b_o10, b_d1;  // block on index pair (1, 2)
};
template <>
struct GateRegs<GC_XRI1> {
    double b_d0, b_o01, b_o10, b_d1;  // block on index pair (1, 2); the (0, 3) block is the identity
};
template <int GC>
__device__ __forceinline__ GateRegs<GC> load_gate(const double* gates, int j) {
    gate_ptr G = gate_matrix(gates, j);
    if constexpr (GC == GC_XRI1) {
        GateRegs<GC> r;
        r.b_d0 = SLAM_GRE(1, 1); r.b_o01 = SLAM_GIM(1, 2); r.b_o10 = SLAM_GIM(2, 1); r.b_d1 = SLAM_GRE(2, 2);
        return r;
    } else if constexpr (GC == GC_XRI) {
        GateRegs<GC> r;
        r.a_d0 = SLAM_GRE(0, 0); r.a_o01 = SLAM_GIM(0, 3); r.a_o10 = SLAM_GIM(3, 0); r.a_d1 = SLAM_GRE(3, 3);
        r.b_d0 = SLAM_GRE(1, 1); r.b_o01 = SLAM_GIM(1, 2); r.b_o10 = SLAM_GIM(2, 1); r.b_d1 = SLAM_GRE(2, 2);
        return r;
    } else {
        return GateRegs<GC>{G};
    }
}
template <int GC>
__device__ __forceinline__ void gate_col(const GateRegs<GC>& g, double (&Fr)[4], double (&Fi)[4]) {
    if constexpr (GC == GC_XRI1) {
        const double ar = Fr[1], ai = Fi[1], br = Fr[2], bi = Fi[2];
        Fr[1] = fma(g.b_d0, ar, -g.b_o01 * bi);
        Fi[1] = fma(g.b_d0, ai, g.b_o01 * br);
        Fr[2] = fma(g.b_d1, br, -g.b_o10 * ai);
        Fi[2] = fma(g.b_d1, bi, g.b_o10 * ar);
    } else if constexpr (GC == GC_XRI) {
        {
            const double ar = Fr[0], ai = Fi[0], br = Fr[3], bi = Fi[3];
            Fr[0] = fma(g.a_d0, ar, -g.a_o01 * bi);
            Fi[0] = fma(g.a_d0, ai, g.a_o01 * br);
            Fr[3] = fma(g.a_d1, br, -g.a_o10 * ai);
            Fi[3] = fma(g.a_d1, bi, g.a_o10 * ar);
        }
        {
            const double ar = Fr[1], ai = Fi[1], br = Fr[2], bi = Fi[2];
            Fr[1] = fma(g.b_d0, ar, -g.b_o01 * bi);
            Fi[1] = fma(g.b_d0, ai, g.b_o01 * br);
            Fr[2] = fma(g.b_d1, br, -g.b_o10 * ai);
            Fi[2] = fma(g.b_d1, bi, g.b_o10 * ar);
        }
    } else {
        gate_col<GC>(g.G, Fr, Fi);
    }
}
template <int GC>
__device__ __forceinline__ void gate_row(const GateRegs<GC>& g, double (&Ur)[4], double (&Ui)[4]) {
    if constexpr (GC == GC_XRI1) {
        const double ar = Ur[1], ai = Ui[1], br = Ur[2], bi = Ui[2];
        Ur[1] = fma(g.b_d0, ar, -g.b_o10 * bi);
        Ui[1] = fma(g.b_d0, ai, g.b_o10 * br);
        Ur[2] = fma(g.b_d1, br, -g.b_o01 * ai);
        Ui[2] = fma(g.b_d1, bi, g.b_o01 * ar);
    } else if constexpr (GC == GC_XRI) {
        {
            const double ar = Ur[0], ai = Ui[0], br = Ur[3], bi = Ui[3];
            Ur[0] = fma(g.a_d0, ar, -g.a_o10 * bi);
            Ui[0] = fma(g.a_d0, ai, g.a_o10 * br);
            Ur[3] = fma(g.a_d1, br, -g.a_o01 * ai);
            Ui[3] = fma(g.a_d1, bi, g.a_o01 * ar);
        }
        {
            const double ar = Ur[1], ai = Ui[1], br = Ur[2], bi = Ui[2];
            Ur[1] = fma(g.b_d0, ar, -g.b_o10 * bi);
            Ui[1] = fma(g.b_d0, ai, g.b_o10 * br);
            Ur[2] = fma(g.b_d1, br, -g.b_o01 * ai);
            Ui[2] = fma(g.b_d1, bi, g.b_o01 * ar);
        }
    } else {
        gate_row<GC>(g.G, Ur, Ui);
    }
}

// bit a set: slot a of lane q (parameter 4a + q) is a theta (parameter index divisible by 3).  Computed once per kernel.
template <int K>
__device__ __forceinline__ int theta_slot_bits(int q) {
    int bits = 0;
#pragma unroll
    for (int a = 0; a < Cfg<K>::NA; ++a) {
        const int i = 4 * a + q;
        bits |= ((i - 3 * ((i * 43) >> 7)) == 0) << a;  // i % 3 for i < 128
    }
    return bits;
}

// ---------------------------------------------------------------------------------
// Fused forward chain + BasicCost + analytic gradient for the quad's item.
//   xd    this lane's parameter slots: xd[a] = x[4a + q]
//   tcol  global pointer to T[0][q] of the item's target (row-major (re, im): T[r][q] is 8 r doubles on);
//         the column is re-read every evaluation (L1/L2 hits) instead of living in 16 registers
//   gates gate matrices G_1..G_K in device memory (scalar loads -> SGPR operands)
//   theta_bits  theta_slot_bits<K>(q)
//   xq    LDS: this quad's exchange area (trig table, then gradient transpose)
//   fh    LDS: this lane's slice of the stored column vectors (stride 64 double2 per row)
//   cost_kind  0 = BasicCost, 1 = SquareCost (wave-uniform)
// Returns loss (replicated over the quad), gd[a] = dloss/dx[4a + q] and column q of W.
// ---------------------------------------------------------------------------------
// HUGE_ARGS: also handle |x| >= 2e9 (out-of-line ocml path).  The optimizer kernel keeps |x| far
// below that (x0 in [0, 2 pi) or validated by the host, steps <= 2 rad) and instantiates false, so no
// function call -- and none of the register save/restore traffic a call site drags in -- sits in its loop.
template <int K, bool HUGE_ARGS, int GC>
__device__ __forceinline__ void eval_quad(const double (&xd)[Cfg<K>::NA], const double* tcol,
                                          const double* gates, double* xq, double2* fh, const double2* tbl,
                                          int q, int theta_bits, int cost_kind, double& fout, double (&gd)[Cfg<K>::NA],
                                          double (&Wr)[4], double (&Wi)[4]) {
    constexpr bool LEAN = lean_layout<K, GC>();
    constexpr bool PSQ = psq_layout<K, GC>();
    using C = Cfg<K, PSQ>;
    asm volatile("" : "+v"(theta_bits));  // one register, not NA hoisted lane masks
    // this quad's trig table (Cfg::TOFF): offsets (0, 4, 2, 6) doubles for quad mod 4 = 0..3
    double* const xt = xq + (PSQ ? (int)((threadIdx.x & 4) + ((threadIdx.x >> 2) & 2)) : 0);
    auto HS = [](int j) constexpr { return LEAN ? j : 2 * j + 1; };  // fh slot of the layer output h_j
    // the target column is requested first and consumed after the forward pass
    double tre[4], tim[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double2 t = *reinterpret_cast<const double2*>(tcol + 8 * r);
        tre[r] = t.x;
        tim[r] = t.y;
    }
    // ---- 1. trig table: each lane handles its own parameter slots
    {
        double2* t2 = reinterpret_cast<double2*>(xt);
        if constexpr (HUGE_ARGS) {
#pragma unroll
            for (int a = 0; a < C::NA; ++a) {
                const int i = 4 * a + q;
                const int i3 = i - 3 * ((i * 43) >> 7);  // i % 3 for i < 128
                const double arg = (i3 == 0) ? 0.5 * xd[a] : xd[a];
                double s, c;
                sincos_any(arg, tbl, s, c);
                t2[i] = make_double2(c, s);
            }
        } else {
            // all table entries are requested before the first polynomial: one exposed LDS latency instead of NA (the
            // compiler cannot move a table read above the previous slot's store -- both live in the same LDS array)
            double rr[C::NA];
            int kk[C::NA];
            double2 tt[C::NA];
            const SincosLits L = sincos_lits_device();
#pragma unroll
            for (int a = 0; a < C::NA; ++a) {
                // theta slots take the half angle: bit a of theta_bits is set when parameter 4a + q is a theta -> exponent
                // -1 (sign-extended bit-field extract) for v_ldexp_f64, which gets zeros and denormals right.  (The
                // select form cost a multiplication, two v_cndmask and a lane mask in a scalar register pair per slot.)
                const double arg = __builtin_amdgcn_ldexp(xd[a], __builtin_amdgcn_sbfe(theta_bits, a, 1));
                sincos_tbl_lookup<true>(arg, tbl, L, rr[a], kk[a], tt[a]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < C::NA; ++a) {
                double s, c;
                sincos_tbl_finish<true>(rr[a], kk[a], tt[a], L, s, c);
                t2[4 * a + q] = make_double2(c, s);
            }
        }
    }
    lds_fence();

    // ---- 2. forward: F = column q of the running product
    // LEAN: P = h_{j-1} while the backward pass is at layer j.  h_{K-1} never goes through LDS; the others are read ONCE,
    // at the top of layer j -- a hundred operations before their first use (f_j = G_j h_{j-1}) -- and then stay in
    // registers as the next layer's h
    double Pr[4], Pi[4];
    // this lane's unit vector e_q as real pairs per qubit: (1 - bit, bit)
    // (from an opaque copy of q: four selects per evaluation instead of eight registers live across the optimizer loop)
    int qo = q;
    asm volatile("" : "+v"(qo));
    const double e0[2] = {(qo & 1) ? 0.0 : 1.0, (qo & 1) ? 1.0 : 0.0};
    const double e1[2] = {(qo & 2) ? 0.0 : 1.0, (qo & 2) ? 1.0 : 0.0};
    double Fr[4], Fi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Fr[r] = (r == q) ? 1.0 : 0.0;
        Fi[r] = 0.0;
    }
    // the next layer's trig entries are requested half a layer ahead (before the qubit-1 gate is applied)
    U3t Bn = load_u3(xt, 0), An = load_u3(xt, 3);
#pragma unroll
    for (int j = 0; j <= K; ++j) {
        if (j > 0 && !LEAN) {
#pragma unroll
            for (int r = 0; r < 4; ++r) fh[((2 * (j - 1)) * 4 + r) * kRow] = make_double2(Fr[r], Fi[r]);
        }
        const U3t B = Bn;  // qubit 0 gate
        const U3t A = An;  // qubit 1 gate
        GateRegs<GC> Gf;
        if (j < K) {
            // requested as soon as the trig entries have arrived (an LDS wait with scalar loads in flight would have to
            // wait for those as well), consumed after the layer's 64 operations
            if constexpr (GC == GC_XRI || GC == GC_XRI1) u3_arrived(B, A);
            Gf = load_gate<GC>(gates, j);
            if constexpr (GC == GC_XRI || GC == GC_XRI1) __builtin_amdgcn_sched_barrier(0);
        }
        double b0r, b0i, b1r, b1i;
        if (j == 0) {
            // layer 0 acts on the unit vector e_q:  K_0 e_q = (A e_{q >> 1}) (x) (B e_{q & 1}) -- 40 operations instead of 64
            u3_unit(B, e0[0], e0[1], b0r, b0i, b1r, b1i);
        } else {
            u3_col(B, Fr[0], Fi[0], Fr[1], Fi[1]);
            u3_col(B, Fr[2], Fi[2], Fr[3], Fi[3]);
        }
        if (j < K) {
            Bn = load_u3(xt, 6 * (j + 1));
            An = load_u3(xt, 6 * (j + 1) + 3);
            __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the reads to their first use)
        }
        if (j == 0) {
            double a0r, a0i, a1r, a1i;
            u3_unit(A, e1[0], e1[1], a0r, a0i, a1r, a1i);
            Fr[0] = a0r * b0r - a0i * b0i; Fi[0] = a0r * b0i + a0i * b0r;
            Fr[1] = a0r * b1r - a0i * b1i; Fi[1] = a0r * b1i + a0i * b1r;
            Fr[2] = a1r * b0r - a1i * b0i; Fi[2] = a1r * b0i + a1i * b0r;
            Fr[3] = a1r * b1r - a1i * b1i; Fi[3] = a1r * b1i + a1i * b1r;
        } else {
            u3_col(A, Fr[0], Fi[0], Fr[2], Fi[2]);
            u3_col(A, Fr[1], Fi[1], Fr[3], Fi[3]);
        }
        if (j < K) {
            if (LEAN && j == K - 1) {
                // the last stored vector is needed again a layer and a half later: it stays in registers
#pragma unroll
                for (int r = 0; r < 4; ++r) { Pr[r] = Fr[r]; Pi[r] = Fi[r]; }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) fh[(HS(j) * 4 + r) * kRow] = make_double2(Fr[r], Fi[r]);
            }
            // F <- G_{j+1} F
            gate_col<GC>(Gf, Fr, Fi);
        }
    }
    // column q of W = template unitary (CircuitTemplate.eval)
#pragma unroll
    for (int r = 0; r < 4; ++r) { Wr[r] = Fr[r]; Wi[r] = Fi[r]; }

    // ---- 3. t = Tr(T^+ W), loss, z = -conj(t) / (4|t|)
    double pr = 0.0, pi = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pr = fma(tre[r], Fr[r], fma(tim[r], Fi[r], pr));
        pi = fma(tre[r], Fi[r], fma(-tim[r], Fr[r], pi));
    }
    quad_sum2(pr, pi);
    const double at2 = pr * pr + pi * pi;
    const double rat = (at2 > 1e-300) ? fast_rsqrt(at2) : 0.0;  // 1 / |t|
    const double at = at2 * rat;
    const double basic = 1.0 - 0.25 * at;  // BasicCost, cost_function.py:140-145
    // SquareCost (cost_function.py:169-173): 1 - (|t|^2 + d) / (d (d + 1)), d = 4, is the monotone map
    // 0.8 (2 L - L^2) of BasicCost L, so its gradient is 1.6 (1 - L) times BasicCost's (wave-uniform select)
    // written as L (c0 + c1 L) and its derivative factor d0 + d1 L with wave-uniform coefficients ((1, 0), (1, 0) for
    // BasicCost -- exact --, (1.6, -0.8), (1.6, -1.6) for SquareCost): scalar selects instead of vector ones
    // (opaque: hoisted out of the optimizer loop the three coefficients were spilled into VGPR lanes and came back through six
    // v_readlane -- vector-ALU slots -- per evaluation; re-selected here they are three s_cselect_b64)
    asm volatile("" : "+s"(cost_kind));
    const bool sq = (cost_kind == 1);
    const double c0 = sq ? 1.6 : 1.0, c1 = sq ? -0.8 : 0.0, d1 = sq ? -1.6 : 0.0;
    fout = basic * fma(c1, basic, c0);
    const double inv = (0.25 * rat) * fma(d1, basic, c0);
    const double zr = -pr * inv, zi = pi * inv;

    // ---- 4. backward: u = row q of (z T^+)(suffix); accumulate this column's partials
    double Ur[4], Ui[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Ur[r] = zr * tre[r] + zi * tim[r];
        Ui[r] = zi * tre[r] - zr * tim[r];
    }
    constexpr bool kL0Out = true;  // layer 0's partials from its output side (l0_gate_partials)
    constexpr bool kBwdTrigAhead = (K == 1);
    constexpr bool kEarlyP = (K <= 4);
    constexpr bool kKeepTopTrig = (K <= 3);  // layer K's trig entries stay in registers from the forward pass
    static_assert(!PSQ || kKeepTopTrig, "the partial-sum planes start over the top layer's trig entries: nobody may read those in the backward pass");
    double Hr[4], Hi[4];  // h = output of the current layer (registers for j = K)
#pragma unroll
    for (int r = 0; r < 4; ++r) { Hr[r] = Fr[r]; Hi[r] = Fi[r]; }
#pragma unroll
    for (int j = K; j >= (kL0Out ? 1 : 0); --j) {
        if (!LEAN && j < K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 v = fh[(HS(j) * 4 + r) * kRow];
                Hr[r] = v.x; Hi[r] = v.y;
            }
        }
        // layer K's trig entries are still in registers from the forward pass; layer j - 1's are requested below, before
        // this layer's pair sums are stored
        // (K = 1 only: at longer spans the 24 registers are not there)
        if constexpr (!kKeepTopTrig) {
            if (j == K) {
                An = load_u3(xt, 6 * j + 3);
                Bn = load_u3(xt, 6 * j);
            }
        }
        const U3t B = Bn;
        const U3t A = An;
        if (LEAN && kEarlyP && j > 0 && j < K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 v = fh[(HS(j - 1) * 4 + r) * kRow];
                Pr[r] = v.x; Pi[r] = v.y;
            }
        }
        GateRegs<GC> Gb;
        if (j > 0) {  // one request for both uses below, issued once this layer's trig entries have arrived
            if constexpr (GC == GC_XRI || GC == GC_XRI1) u3_arrived(B, A);
            Gb = load_gate<GC>(gates, j - 1);
            if constexpr (GC == GC_XRI || GC == GC_XRI1) __builtin_amdgcn_sched_barrier(0);
        }
        // phi: dU/dphi = i diag(0,1) U  ->  -Im( sum over rows with that qubit's bit set of u*h )
        const double m1 = im_mul(Ur[1], Ui[1], Hr[1], Hi[1]);
        const double m2 = im_mul(Ur[2], Ui[2], Hr[2], Hi[2]);
        const double m3 = im_mul(Ur[3], Ui[3], Hr[3], Hi[3]);
        double part[6];
        part[1] = -(m1 + m3);  // qubit 0: rows 1, 3
        part[4] = -(m2 + m3);  // qubit 1: rows 2, 3
        // u~ = u K_j
        // A is applied first: its theta derivative comes from the output side (u, h) -- at k <= 2; at k = 3 keeping
        // h alive through the row actions costs more in spills than the shorter formula saves (measured)
        constexpr bool kThetaOut = (K <= 2);
        double thA2 = 0.0;
        if constexpr (kThetaOut) {
            u3_row_dtheta_first(A, Ur[0], Ui[0], Ur[2], Ui[2], Hr[0], Hi[0], Hr[2], Hi[2], thA2);
            u3_row_dtheta_first(A, Ur[1], Ui[1], Ur[3], Ui[3], Hr[1], Hi[1], Hr[3], Hi[3], thA2);
        } else {
            u3_row(A, Ur[0], Ui[0], Ur[2], Ui[2]);
            u3_row(A, Ur[1], Ui[1], Ur[3], Ui[3]);
        }
        double tB01r, tB01i, tB23r, tB23i;  // B is applied last: its theta / lambda derivatives come from the intermediates
        u3_row_t(B, Ur[0], Ui[0], Ur[1], Ui[1], tB01r, tB01i);
        u3_row_t(B, Ur[2], Ui[2], Ur[3], Ui[3], tB23r, tB23i);
        // f = input of layer j
        double fr[4], fi[4];
        if (j > 0) {
            if (LEAN) {
                if (!kEarlyP && j < K) {  // span 5: no registers for the early request
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double2 v = fh[(HS(j - 1) * 4 + r) * kRow];
                        Pr[r] = v.x; Pi[r] = v.y;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { fr[r] = Pr[r]; fi[r] = Pi[r]; }
                gate_col<GC>(Gb, fr, fi);  // f_j = G_j h_{j-1}
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 v = fh[(2 * (j - 1) * 4 + r) * kRow];
                    fr[r] = v.x; fi[r] = v.y;
                }
            }
        } else {
            fr[0] = e1[0] * e0[0]; fr[1] = e1[0] * e0[1]; fr[2] = e1[1] * e0[0]; fr[3] = e1[1] * e0[1];
#pragma unroll
            for (int r = 0; r < 4; ++r) fi[r] = 0.0;
        }
        // qubit 0 (B, applied last to u): theta and lambda from the rotation intermediates
        double th01, la01, th23, la23;
        if (j > 0) {
            dtheta_dlam_last(B, Ur[0], Ui[0], tB01r, tB01i, fr[0], fi[0], fr[1], fi[1], th01, la01);
            dtheta_dlam_last(B, Ur[2], Ui[2], tB23r, tB23i, fr[2], fi[2], fr[3], fi[3], th23, la23);
        } else {  // the input of layer 0 is real (e_q)
            dtheta_dlam_last_real(B, Ur[0], Ui[0], tB01r, tB01i, fr[0], fr[1], th01, la01);
            dtheta_dlam_last_real(B, Ur[2], Ui[2], tB23r, tB23i, fr[2], fr[3], th23, la23);
        }
        part[0] = th01 + th23;
        part[2] = la01 + la23;
        // qubit 1 (A): lambda: dU/dlam = U i diag(0,1); theta: generic form
        if (j > 0) {
            const double l2 = im_mul(Ur[2], Ui[2], fr[2], fi[2]);
            const double l3 = im_mul(Ur[3], Ui[3], fr[3], fi[3]);
            part[5] = -(l2 + l3);
        } else {
            part[5] = -(Ui[2] * fr[2] + Ui[3] * fr[3]);
        }
        if constexpr (kThetaOut)
            part[3] = 0.5 * thA2;
        else
            part[3] = dtheta_pair(A, Ur[0], Ui[0], Ur[2], Ui[2], fr[0], fi[0], fr[2], fi[2]) +
                      dtheta_pair(A, Ur[1], Ui[1], Ur[3], Ui[3], fr[1], fi[1], fr[3], fi[3]);
        // sum the partials of lane pairs (q, q ^ 1) and park the 6 x 2 pair sums in this layer's own trig-table
        // entries, which are dead from here on (every lane of the quad has loaded them above): parameter i = 6 j + m
        // ends up as the double2 at xq + 2 i, so its owner lane reads it back with one ds_read_b128 at a fixed offset
        // from a per-lane base -- no lane-dependent address arithmetic, no bank conflicts (the old gather read four
        // b64 words per parameter from the stored-vector slots: ~20 integer instructions each and a 2-way conflict)
        if (kBwdTrigAhead && j > 0) {
            Bn = load_u3(xt, 6 * (j - 1));
            An = load_u3(xt, 6 * (j - 1) + 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PSQ) {
            // every lane stores its partial into its own plane; the owner of parameter i reads the four planes' entry i and
            // adds them up (no DPP moves, no additions on the producing side)
#pragma unroll
            for (int m = 0; m < 6; ++m) xq[C::PS0 + q * C::PSP + 6 * j + m] = part[m];
        } else {
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                const double ps = part[m] + dpp_f64<0xB1>(part[m]);
                if ((q & 1) == 0) xq[12 * j + 2 * m + (q >> 1)] = ps;
            }
        }
        if (!kBwdTrigAhead && j > 0) {
            // K >= 2: requested here, behind the pair-sum stores (no registers to spare earlier): the gate's row action and
            // the next layer's phi partials run under the LDS latency.  The qubit-1 gate is applied first: its entries first
            An = load_u3(xt, 6 * (j - 1) + 3);
            Bn = load_u3(xt, 6 * (j - 1));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (j > 0) {
            // u <- u~ G_j
            gate_row<GC>(Gb, Ur, Ui);
            if (LEAN) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { Hr[r] = Pr[r]; Hi[r] = Pi[r]; }
            }
        }
    }

    if constexpr (kL0Out) {
        // layer 0 from its output side: Ur/Ui is the backward vector behind G_1, Bn/An hold layer 0's trig entries
        const U3t B = Bn;
        const U3t A = An;
        double b0r, b0i, b1r, b1i, tbr, tbi, a0r, a0i, a1r, a1i, tar, tai;
        u3_unit_t(B, e0[0], e0[1], b0r, b0i, b1r, b1i, tbr, tbi);
        u3_unit_t(A, e1[0], e1[1], a0r, a0i, a1r, a1i, tar, tai);
        // wB[r0] = u[r0] a0 + u[2 + r0] a1,   wA[r1] = u[2 r1] b0 + u[2 r1 + 1] b1
        const double wB0r = (Ur[0] * a0r - Ui[0] * a0i) + (Ur[2] * a1r - Ui[2] * a1i), wB0i = (Ur[0] * a0i + Ui[0] * a0r) + (Ur[2] * a1i + Ui[2] * a1r);
        const double wB1r = (Ur[1] * a0r - Ui[1] * a0i) + (Ur[3] * a1r - Ui[3] * a1i), wB1i = (Ur[1] * a0i + Ui[1] * a0r) + (Ur[3] * a1i + Ui[3] * a1r);
        const double wA0r = (Ur[0] * b0r - Ui[0] * b0i) + (Ur[1] * b1r - Ui[1] * b1i), wA0i = (Ur[0] * b0i + Ui[0] * b0r) + (Ur[1] * b1i + Ui[1] * b1r);
        const double wA1r = (Ur[2] * b0r - Ui[2] * b0i) + (Ur[3] * b1r - Ui[3] * b1i), wA1i = (Ur[2] * b0i + Ui[2] * b0r) + (Ur[3] * b1i + Ui[3] * b1r);
        double part[6];
        l0_gate_partials(B, e0[0], e0[1], wB0r, wB0i, wB1r, wB1i, b0r, b0i, b1r, b1i, tbr, tbi, part[0], part[1], part[2]);
        l0_gate_partials(A, e1[0], e1[1], wA0r, wA0i, wA1r, wA1i, a0r, a0i, a1r, a1i, tar, tai, part[3], part[4], part[5]);
        if constexpr (PSQ) {
#pragma unroll
            for (int m = 0; m < 6; ++m) xq[C::PS0 + q * C::PSP + m] = part[m];
        } else {
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                const double ps = part[m] + dpp_f64<0xB1>(part[m]);
                if ((q & 1) == 0) xq[2 * m + (q >> 1)] = ps;
            }
        }
    }

    // ---- 5. the owner of parameter i (lane i & 3, slot i >> 2) adds the two pair sums
    lds_fence();
    {
        if constexpr (PSQ) {
            const double* ps = xq + C::PS0 + q;
            double p0[C::NA], p1[C::NA], p2[C::NA], p3[C::NA];
#pragma unroll
            for (int a = 0; a < C::NA; ++a) {
                p0[a] = ps[4 * a];
                p1[a] = ps[4 * a + C::PSP];
                p2[a] = ps[4 * a + 2 * C::PSP];
                p3[a] = ps[4 * a + 3 * C::PSP];
            }
#pragma unroll
            for (int a = 0; a < C::NA; ++a) gd[a] = (4 * a + q < C::N) ? (p0[a] + p1[a]) + (p2[a] + p3[a]) : 0.0;
        } else {
            const double2* ps2 = reinterpret_cast<const double2*>(xq) + q;
            double2 ps[C::NA];
#pragma unroll
            for (int a = 0; a < C::NA; ++a) ps[a] = ps2[4 * a];
#pragma unroll
            for (int a = 0; a < C::NA; ++a) gd[a] = (4 * a + q < C::N) ? ps[a].x + ps[a].y : 0.0;
        }
    }
    lds_fence();
}

// ---------------------------------------------------------------------------------
// Packed symmetric fp32 inverse Hessian in registers: block (a, b), a <= b, index b(b+1)/2 + a;
// lane q holds H[4a + q][4b + e], e = 0..3, as two f32x2.
// ---------------------------------------------------------------------------------
__host__ __device__ constexpr int blk(int a, int b) { return b * (b + 1) / 2 + a; }

template <int NA>
struct HMat {
    f32x2 h[NA * (NA + 1) / 2][2];
};

template <int NA>
__device__ __forceinline__ void h_set_identity_where(HMat<NA>& H, int q, bool where) {
#pragma unroll
    for (int b = 0; b < NA; ++b)
#pragma unroll
        for (int a = 0; a <= b; ++a)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                f32x2 cur = H.h[blk(a, b)][hh];
                const float id0 = (a == b && 2 * hh == q) ? 1.0f : 0.0f;
                const float id1 = (a == b && 2 * hh + 1 == q) ? 1.0f : 0.0f;
                cur.x = where ? id0 : cur.x;
                cur.y = where ? id1 : cur.y;
                H.h[blk(a, b)][hh] = cur;
            }
}

// out = H v  (v, out distributed: slot a of lane q = component 4a + q); fp32 arithmetic.
// Lane q holds row q of each upper block (a, b): it adds H[4a+q][4b+e] v[4b+e] to its own out[a]
// and owes H[4a+q][4b+e] v[4a+q] to lane e's out[b]; those "transposed" sums go through LDS as soon
// as column block b is finished, so only one 4-float accumulator is live at a time.
template <int NA>
__device__ __forceinline__ void h_matvec(const HMat<NA>& H, const double (&vd)[NA], float* xq32, int q,
                                         double (&out)[NA]) {
    float v32[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        v32[a] = (float)vd[a];
        xq32[4 * a + q] = v32[a];
    }
    lds_fence();
    float* xt = xq32 + 4 * NA - 16;  // [b >= 1][e][q] transposed partial sums, stored from b = 1 (disjoint from the v area)
    f32x2 acc[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) acc[a] = f32x2{0.0f, 0.0f};
    // every column block's broadcast is requested before the first product (one exposed LDS latency instead of NA)
    // (not at span 5: the 36 + 32 registers of the two batches are not there)
    constexpr bool kBatch = NA <= 8;
    f32x4 vall[NA];
    if constexpr (kBatch) {
#pragma unroll
        for (int b = 0; b < NA; ++b) vall[b] = *reinterpret_cast<const f32x4*>(xq32 + 4 * b);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int b = 0; b < NA; ++b) {
        const f32x4 vb = kBatch ? vall[b] : *reinterpret_cast<const f32x4*>(xq32 + 4 * b);
        const f32x2 vb0 = f32x2{vb.x, vb.y}, vb1 = f32x2{vb.z, vb.w};
        f32x2 t0 = f32x2{0.0f, 0.0f}, t1 = f32x2{0.0f, 0.0f};
#pragma unroll
        for (int a = 0; a <= b; ++a) {
            acc[a] = __builtin_elementwise_fma(H.h[blk(a, b)][0], vb0, acc[a]);
            acc[a] = __builtin_elementwise_fma(H.h[blk(a, b)][1], vb1, acc[a]);
            if (a < b) {
                const f32x2 va = f32x2{v32[a], v32[a]};
                t0 = __builtin_elementwise_fma(H.h[blk(a, b)][0], va, t0);
                t1 = __builtin_elementwise_fma(H.h[blk(a, b)][1], va, t1);
            }
        }
        if (b >= 1) {
            xt[(4 * b + 0) * 4 + q] = t0.x;
            xt[(4 * b + 1) * 4 + q] = t0.y;
            xt[(4 * b + 2) * 4 + q] = t1.x;
            xt[(4 * b + 3) * 4 + q] = t1.y;
        }
    }
    lds_fence();
    f32x4 tall[NA];
    if constexpr (kBatch) {
#pragma unroll
        for (int a = 1; a < NA; ++a) tall[a] = *reinterpret_cast<const f32x4*>(xt + (4 * a + q) * 4);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        float o = acc[a].x + acc[a].y;
        if (a >= 1) {
            const f32x4 t = kBatch ? tall[a] : *reinterpret_cast<const f32x4*>(xt + (4 * a + q) * 4);
            o += (t.x + t.y) + (t.z + t.w);
        }
        out[a] = (double)o;
    }
    lds_fence();
}

// H += s w^T + v s^T  (rank-2 BFGS inverse update), s, w, v distributed (already fp32)
template <int NA>
__device__ __forceinline__ void h_update(HMat<NA>& H, const float (&s32)[NA], const float (&w32)[NA],
                                         const float (&v32)[NA], float* xq32, int q) {
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
            f32x2 h0 = H.h[blk(a, b)][0], h1 = H.h[blk(a, b)][1];
            h0 = __builtin_elementwise_fma(va, sb0, h0);
            h1 = __builtin_elementwise_fma(va, sb1, h1);
            h0 = __builtin_elementwise_fma(sa, wb0, h0);
            h1 = __builtin_elementwise_fma(sa, wb1, h1);
            H.h[blk(a, b)][0] = h0;
            H.h[blk(a, b)][1] = h1;
        }
    }
    lds_fence();
}

template <int NA>
__device__ __forceinline__ double qdot(const double (&a)[NA], const double (&b)[NA]) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NA; ++i) s = fma(a[i], b[i], s);
    return quad_sum(s);
}

}  // namespace slamdev
