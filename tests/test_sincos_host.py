"""CPU: the kernels' branch-free sincos routines (slam_decomposition_amd/csrc/slam_sincos.hpp: the
table-driven sincos_tbl the kernels use, and the table-free sincos_fast) compiled for the host with g++ and
compared with long-double libm over |x| up to 1e8 / 1e9."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include "slam_sincos.hpp"
#include <cstdio>
#include <cstdlib>
#include <cmath>
struct D2 { double x, y; };
int main() {
    static const double raw[slamdev::kSincosTableDoubles] = SLAM_SINCOS_TABLE;
    const D2* tbl = reinterpret_cast<const D2*>(raw);
    srand48(1);
    const double ranges[] = {1, 10, 100, 1e4, 1e6, 1e9};
    double worst = 0;
    for (double R : ranges)
        for (int i = 0; i < 400000; ++i) {
            const double x = (drand48() * 2 - 1) * (R < 1e8 ? R : 1e8);
            double s, c;
            slamdev::sincos_tbl(x, tbl, s, c);
            const double es = fabs((double)(s - sinl((long double)x)));
            const double ec = fabs((double)(c - cosl((long double)x)));
            if (es > worst) worst = es;
            if (ec > worst) worst = ec;
        }
    for (int j = 0; j < 32; ++j) {  // the table itself: correctly rounded
        const long double a = j * 3.14159265358979323846264338327950288L / 32;
        if (fabs((double)(tbl[j].x - cosl(a))) > 1.2e-16 || fabs((double)(tbl[j].y - sinl(a))) > 1.2e-16) return 3;
    }
    for (double R : ranges)
        for (int i = 0; i < 400000; ++i) {
            const double x = (drand48() * 2 - 1) * R;
            double s, c;
            slamdev::sincos_fast(x, s, c);
            const double es = fabs((double)(s - sinl((long double)x)));
            const double ec = fabs((double)(c - cosl((long double)x)));
            if (es > worst) worst = es;
            if (ec > worst) worst = ec;
        }
    const double xs[] = {0.0, -0.0, 1.5707963267948966, 3.141592653589793, -7.853981633974483, 1e-300};
    for (double x : xs) {
        double s, c;
        slamdev::sincos_fast(x, s, c);
        if (fabs(s - sin(x)) > 3e-16 || fabs(c - cos(x)) > 3e-16) return 2;
        slamdev::sincos_tbl(x, tbl, s, c);
        if (fabs(s - sin(x)) > 3e-16 || fabs(c - cos(x)) > 3e-16) return 2;
    }
    printf("%.3e\n", worst);
    return worst < 3e-16 ? 0 : 1;
}
"""


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_sincos_fast_accuracy(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    inc = os.path.join(ROOT, "slam_decomposition_amd", "csrc")
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-I", inc, "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
