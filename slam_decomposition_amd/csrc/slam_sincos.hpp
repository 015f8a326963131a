// slam_sincos.hpp -- branch-free fp64 sincos for |x| < 2e9 (FMA Cody-Waite reduction by pi/2 in
// double-double, fdlibm kernel polynomials on [-pi/4, pi/4]); <= 2 ulp, |abs err| < 3e-16.
// Compiles for the device (hipcc) and for the host (g++, tests/test_sincos_host.py).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define SLAM_HD __host__ __device__ __forceinline__
#else
#define SLAM_HD inline
#endif

namespace slamdev {

constexpr double kSincosFastLimit = 2.0e9;   // sincos_fast
constexpr double kSincosTblLimit = 2.0e8;    // sincos_tbl: n = x * 32/pi must fit an int32

SLAM_HD void sincos_fast(double x, double& s, double& c) {
    const double n = rint(x * 0.63661977236758134308);  // x * 2/pi
    double r = fma(-n, 1.5707963267948965580, x);        // pi/2 high part
    r = fma(-n, 6.1232339957367658860e-17, r);           // pi/2 low part
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double sr = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)n;
    const bool swap = q & 1;
    const double s0 = swap ? cr : sr;
    const double c0 = swap ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

// (cos, sin)(j pi / 32), j = 0..31, correctly rounded (hex float literals).  The kernels keep a copy in LDS.
#define SLAM_SINCOS_TABLE \
    { 0x1.0000000000000p+0, 0x0.0p+0, 0x1.fd88da3d12526p-1, 0x1.917a6bc29b42cp-4, \
    0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60bp-3, 0x1.e9f4156c62ddap-1, 0x1.294062ed59f06p-2, \
    0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2, 0x1.c38b2f180bdb1p-1, 0x1.e2b5d3806f63bp-2, \
    0x1.a9b66290ea1a3p-1, 0x1.1c73b39ae68c8p-1, 0x1.8bc806b151741p-1, 0x1.44cf325091dd6p-1, \
    0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1, 0x1.44cf325091dd6p-1, 0x1.8bc806b151741p-1, \
    0x1.1c73b39ae68c8p-1, 0x1.a9b66290ea1a3p-1, 0x1.e2b5d3806f63bp-2, 0x1.c38b2f180bdb1p-1, \
    0x1.87de2a6aea963p-2, 0x1.d906bcf328d46p-1, 0x1.294062ed59f06p-2, 0x1.e9f4156c62ddap-1, \
    0x1.8f8b83c69a60bp-3, 0x1.f6297cff75cb0p-1, 0x1.917a6bc29b42cp-4, 0x1.fd88da3d12526p-1, \
    -0x1.d9cceba3f91f2p-66, 0x1.0000000000000p+0, -0x1.917a6bc29b42cp-4, 0x1.fd88da3d12526p-1, \
    -0x1.8f8b83c69a60bp-3, 0x1.f6297cff75cb0p-1, -0x1.294062ed59f06p-2, 0x1.e9f4156c62ddap-1, \
    -0x1.87de2a6aea963p-2, 0x1.d906bcf328d46p-1, -0x1.e2b5d3806f63bp-2, 0x1.c38b2f180bdb1p-1, \
    -0x1.1c73b39ae68c8p-1, 0x1.a9b66290ea1a3p-1, -0x1.44cf325091dd6p-1, 0x1.8bc806b151741p-1, \
    -0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1, -0x1.8bc806b151741p-1, 0x1.44cf325091dd6p-1, \
    -0x1.a9b66290ea1a3p-1, 0x1.1c73b39ae68c8p-1, -0x1.c38b2f180bdb1p-1, 0x1.e2b5d3806f63bp-2, \
    -0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2, -0x1.e9f4156c62ddap-1, 0x1.294062ed59f06p-2, \
    -0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60bp-3, -0x1.fd88da3d12526p-1, 0x1.917a6bc29b42cp-4 }
constexpr int kSincosTableDoubles = 64;

// Table-driven variant: x = n pi/32 + r, |r| <= pi/64; (cos, sin)(n pi/32) from the 32-entry half-circle
// table (sign from bit 5 of n), (cos, sin)(r) from degree-7/8 Taylor polynomials, combined by the angle
// addition formulas.  26 instructions on the device against 41 for sincos_fast (no quadrant selects, short
// polynomials); same accuracy (|abs err| < 3e-16) for |x| < 2e8.
// The evaluation is split in two so that a caller with several arguments can request all table entries first and
// run the polynomials while they are on their way from LDS (sincos_tbl_lookup for every argument, then
// sincos_tbl_finish for every argument); sincos_tbl is the two in sequence.
//
// The nine fp64 literals travel in a struct: a 64-bit literal needs a scalar register pair on gfx950, and inside the
// optimizer loop the compiler hoisted them out of the loop and then spilled them into VGPR lanes (a v_readlane -- a
// vector-ALU slot -- per use).  sincos_lits_device() hands them out through an empty asm, so they are materialised
// (two s_mov_b32 each) once per evaluation and die with it.
struct SincosLits {
    double inv;         // 32/pi
    double hi, lo;      // pi/32 in two parts (= pi/2 high / 16, pi/2 low / 16)
    double s2, s1, s0;  // sin(r) = r + r^3 (s0 + z (s1 + z s2)), z = r^2
    double c2, c1, c0;  // cos(r) = 1 - z/2 + z^2 (c0 + z (c1 + z c2))
};
SLAM_HD SincosLits sincos_lits() {
    return SincosLits{10.185916357881301489,    9.8174770424681034876e-02, 3.8270212473354786788e-18,
                      -1.98412698412698412698e-04, 8.33333333333333333333e-03, -1.66666666666666666667e-01,
                      2.48015873015873015873e-05, -1.38888888888888888889e-03, 4.16666666666666666667e-02};
}
#if defined(__HIPCC__)
__device__ __forceinline__ double sincos_opaque(double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(c));
#endif
    return c;
}
__device__ __forceinline__ SincosLits sincos_lits_device() {
    SincosLits l = sincos_lits();
    l.inv = sincos_opaque(l.inv); l.hi = sincos_opaque(l.hi); l.lo = sincos_opaque(l.lo);
    l.s2 = sincos_opaque(l.s2); l.s1 = sincos_opaque(l.s1); l.s0 = sincos_opaque(l.s0);
    l.c2 = sincos_opaque(l.c2); l.c1 = sincos_opaque(l.c1); l.c0 = sincos_opaque(l.c0);
    return l;
}
#endif
// FULL: the table holds the whole circle (64 entries, the second half = minus the first): no sign fix-up in the finish
template <bool FULL = false, class Tbl>
SLAM_HD void sincos_tbl_lookup(double x, const Tbl* tbl /* double2-like {x = cos, y = sin} [32 or 64] */, const SincosLits& L, double& r,
                               int& k, Tbl& t) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    // round to nearest by adding 1.5 * 2^52: the integer lands in the low dword, no v_rndne / v_cvt (|x * 32/pi| < 2^31)
    const double nm = fma(x, L.inv, 6755399441055744.0);
    k = __double2loint(nm);
    const double n = nm - 6755399441055744.0;
#else
    const double n = rint(x * L.inv);
    k = (int)n;
#endif
    t = tbl[k & (FULL ? 63 : 31)];
    r = fma(-n, L.hi, x);
    r = fma(-n, L.lo, r);
}
template <bool FULL = false, class Tbl>
SLAM_HD void sincos_tbl_finish(double r, int k, const Tbl& t, const SincosLits& L, double& s, double& c) {
    const double z = r * r;
    double ps = fma(z, L.s2, L.s1);
    ps = fma(z, ps, L.s0);
    const double sr = fma(r * z, ps, r);
    double pc = fma(z, L.c2, L.c1);
    pc = fma(z, pc, L.c0);
    const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
    const double c0 = fma(t.x, cr, -(t.y * sr));
    const double s0 = fma(t.y, cr, t.x * sr);
    if (FULL) {
        c = c0;
        s = s0;
        return;
    }
    // odd half-turns: (cos, sin)(a + pi) = -(cos, sin)(a)
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    const int sgn = k << 26 & (int)0x80000000;
    c = __hiloint2double(__double2hiint(c0) ^ sgn, __double2loint(c0));
    s = __hiloint2double(__double2hiint(s0) ^ sgn, __double2loint(s0));
#else
    c = (k & 32) ? -c0 : c0;
    s = (k & 32) ? -s0 : s0;
#endif
}
template <class Tbl>
SLAM_HD void sincos_tbl(double x, const Tbl* tbl, double& s, double& c) {
    const SincosLits L = sincos_lits();
    double r;
    int k;
    Tbl t;
    sincos_tbl_lookup(x, tbl, L, r, k, t);
    sincos_tbl_finish(r, k, t, L, s, c);
}

}  // namespace slamdev
