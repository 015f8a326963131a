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

constexpr double kSincosFastLimit = 2.0e9;

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

}  // namespace slamdev
