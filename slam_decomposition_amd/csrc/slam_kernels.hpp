// slam_kernels.hpp -- __global__ kernels of libslamhip (gfx950 only).
#pragma once
#include "slam_device.hpp"

namespace slamdev {

constexpr double kArmijoC1 = 1e-4;
constexpr int kMaxBacktrack = 20;
constexpr double kStepMax = 2.0;  // cap on |alpha p|_2 of the first trial step (parameters are angles)
constexpr double kCurvEps = 1e-10;
constexpr double kStallDf = 1e-15;
constexpr double kStallGnorm = 1e-5;

enum : int { ST_CONVERGED = 0, ST_MAXITER = 1, ST_LINESEARCH = 2, ST_NONFINITE = 3, ST_STALLED = 4, ST_PREEMPTED = 5 };

struct MinimizeArgs {
    const double* targets;    // [n_targets][32]
    const double* gates;      // [n_gates][32]
    const int32_t* active;    // [n_active] or nullptr
    const double* x0;         // [M][n] or nullptr
    int64_t n_items;          // M = n_active * restarts
    int32_t restarts;
    int32_t maxiter;
    double gtol;
    double stop_loss;
    uint64_t seed;
    uint32_t flags;
    int32_t gate_seq[8];
    // per-item outputs
    double* item_loss;        // [M]
    double* item_x;           // [M][n]
    int32_t* item_iters;      // [M]
    int32_t* item_status;     // [M]
    int32_t* item_evals;      // [M]
};

struct EvalArgs {
    const double* targets;
    const double* gates;
    const double* x;          // [M][n]
    const int32_t* target_of; // [M]
    int64_t n_items;
    int32_t gate_seq[8];
    double* loss;             // [M]
    double* grad;             // [M][n] or nullptr
    double* unitary;          // [M][4][4][2] or nullptr: W = CircuitTemplate.eval(x)
};

template <int K>
__device__ __forceinline__ void stage_gates(const double* gates, const int32_t (&gate_seq)[8], double* gl, int lane) {
    // K * 32 doubles: lane l copies doubles l, l + 64, ...
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (lane < 32) gl[j * 32 + lane] = gates[(int64_t)gate_seq[j] * 32 + lane];
    }
}

// ---------------------------------------------------------------------------------
// loss + gradient for explicit parameter vectors (slam_eval_loss_grad)
// ---------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(kWave) eval_kernel(EvalArgs args) {
    using C = Cfg<K>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* gl = lds;
    double* xchg = lds + C::LDS_GATES;
    double2* fhbase = reinterpret_cast<double2*>(lds + C::LDS_GATES + C::LDS_XCHG);
    const int lane = threadIdx.x;
    const int q = lane & 3;
    const int quad = lane >> 2;
    stage_gates<K>(args.gates, args.gate_seq, gl, lane);
    lds_fence();
    const int64_t item = (int64_t)blockIdx.x * kQuadsPerWave + quad;
    const bool live = item < args.n_items;
    const int64_t it = live ? item : 0;
    const int64_t tgt = args.target_of[it];
    double tre[4], tim[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        tre[r] = args.targets[tgt * 32 + (r * 4 + q) * 2];
        tim[r] = args.targets[tgt * 32 + (r * 4 + q) * 2 + 1];
    }
    double xd[C::NA], gd[C::NA];
#pragma unroll
    for (int a = 0; a < C::NA; ++a) {
        const int i = 4 * a + q;
        xd[a] = (i < C::N) ? args.x[it * C::N + i] : 0.0;
    }
    double f, Wr[4], Wi[4];
    eval_quad<K>(xd, tre, tim, gl, xchg + quad * C::XSTRIDE, fhbase + lane, q, f, gd, Wr, Wi);
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

// ---------------------------------------------------------------------------------
// batched quasi-Newton minimisation: one quad per (target, seed)
// ---------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(kWave, 1) minimize_kernel(MinimizeArgs args) {
    using C = Cfg<K>;
    constexpr int NA = C::NA;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* gl = lds;
    double* xchg = lds + C::LDS_GATES;
    double2* fhbase = reinterpret_cast<double2*>(lds + C::LDS_GATES + C::LDS_XCHG);
    const int lane = threadIdx.x;
    const int q = lane & 3;
    const int quad = lane >> 2;
    double* xq = xchg + quad * C::XSTRIDE;
    double2* fh = fhbase + lane;
    stage_gates<K>(args.gates, args.gate_seq, gl, lane);
    lds_fence();

    const int64_t item = (int64_t)blockIdx.x * kQuadsPerWave + quad;
    const bool live = item < args.n_items;
    const int64_t itc = live ? item : 0;
    const int64_t slot = itc / args.restarts;
    const int32_t restart = (int32_t)(itc - slot * args.restarts);
    const int32_t tgt = args.active ? args.active[slot] : (int32_t)slot;

    double tre[4], tim[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        tre[r] = args.targets[(int64_t)tgt * 32 + (r * 4 + q) * 2];
        tim[r] = args.targets[(int64_t)tgt * 32 + (r * 4 + q) * 2 + 1];
    }

    double x[NA], g[NA], p[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int i = 4 * a + q;
        if (i < C::N) {
            x[a] = args.x0 ? args.x0[itc * C::N + i]
                           : x0_philox(args.seed, (uint32_t)tgt, (uint32_t)restart, (uint32_t)K, (uint32_t)i);
        } else {
            x[a] = 0.0;
        }
    }

    double H[C::NBLK][4];
    h_set_identity<NA>(H, q);

    double f, Wr[4], Wi[4];
    eval_quad<K>(x, tre, tim, gl, xq, fh, q, f, g, Wr, Wi);
    int nev = 1, iters = 0, nback = 0, nstall = 0;
    bool scaled = false;
    int status = ST_MAXITER;
    bool done = !live;
    double gg = qdot<NA>(g, g);
    double gnorm;
    {
        double m = 0.0;
#pragma unroll
        for (int a = 0; a < NA; ++a) m = fmax(m, fabs(g[a]));
        gnorm = quad_max(m);
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) p[a] = -g[a];
    double gp = -gg;
    double alpha = fmin(1.0, 1.0 / fmax(sqrt(gg), 1e-300));
    if (!isfinite(f)) { status = ST_NONFINITE; done = true; }
    else if (f < args.stop_loss || gnorm < args.gtol) { status = ST_CONVERGED; done = true; }
    if (args.maxiter <= 0 && !done) { done = true; }

    // every quad evaluates in lock-step; the wave leaves when all its quads are done
    while (!__all(done)) {
        const bool was_done = done;
        double xt[NA], gt[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) xt[a] = fma(alpha, p[a], x[a]);
        double ft;
        eval_quad<K>(xt, tre, tim, gl, xq, fh, q, ft, gt, Wr, Wi);
        const bool active = !done;
        if (active) ++nev;
        const bool finite = isfinite(ft);
        const bool armijo = finite && (ft <= f + kArmijoC1 * alpha * gp);
        const bool acc = active && armijo;

        // ---- quasi-Newton update (masked by acc through zeroed s, y)
        double s[NA], y[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            s[a] = acc ? (xt[a] - x[a]) : 0.0;
            y[a] = acc ? (gt[a] - g[a]) : 0.0;
        }
        double qv[NA];
        h_matvec<NA>(H, gt, xq, q, qv);
        const double sy = qdot<NA>(s, y);
        const double yy = qdot<NA>(y, y);
        const double ss = qdot<NA>(s, s);
        const bool curv = acc && (sy > kCurvEps * sqrt(ss * yy));
        const bool first = curv && !scaled;
        scaled = scaled || curv;
        const double fac = first ? (sy / yy) : 1.0;
        if (__any(first)) {
#pragma unroll
            for (int b = 0; b < C::NBLK; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) H[b][e] *= fac;
        }
        double u[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            qv[a] *= fac;
            u[a] = fma(fac, p[a], qv[a]);  // H y = H g' - H g = q + p  (p = -H g)
        }
        const double yu = qdot<NA>(y, u);
        const double rho = curv ? 1.0 / sy : 0.0;
        const double cf = rho * (1.0 + rho * yu);
        double w[NA], v[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            // selects, not products with rho = 0: u may be non-finite after a rejected trial point
            w[a] = curv ? (cf * s[a] - rho * u[a]) : 0.0;
            v[a] = curv ? (-rho * u[a]) : 0.0;
        }
        h_update<NA>(H, s, w, v, xq, q);
        const double wg = qdot<NA>(w, gt);
        const double sg = qdot<NA>(s, gt);

        // ---- per-quad state machine
        if (acc) {
            nstall = ((f - ft) <= kStallDf) ? nstall + 1 : 0;
            f = ft;
            ++iters;
            nback = 0;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                x[a] = xt[a];
                g[a] = gt[a];
                p[a] = -(qv[a] + s[a] * wg + v[a] * sg);
            }
        } else if (active) {
            const double denom = 2.0 * (ft - f - gp * alpha);
            const double anew = (finite && denom > 0.0 && isfinite(denom)) ? (-gp * alpha * alpha / denom) : 0.5 * alpha;
            alpha = fmin(fmax(anew, 0.1 * alpha), 0.5 * alpha);
            ++nback;
        }
        // quad-uniform reductions must be executed by all lanes
        {
            double m = 0.0;
#pragma unroll
            for (int a = 0; a < NA; ++a) m = fmax(m, fabs(g[a]));
            gnorm = quad_max(m);
        }
        gp = qdot<NA>(g, p);
        const double pp = qdot<NA>(p, p);
        if (acc) alpha = fmin(1.0, kStepMax / fmax(sqrt(pp), 1e-300));
        if (acc) {
            if (f < args.stop_loss || gnorm < args.gtol) { status = ST_CONVERGED; done = true; }
            else if (nstall >= 2) { status = ST_STALLED; done = true; }
            else if (iters >= args.maxiter) { status = ST_MAXITER; done = true; }
        } else if (active) {
            if (nback > kMaxBacktrack) { status = (gnorm < kStallGnorm) ? ST_STALLED : ST_LINESEARCH; done = true; }
        }
        // not a descent direction (H lost positive definiteness numerically): restart from steepest descent
        const bool reset = !done && !(gp < 0.0);
        if (__any(reset)) {
#pragma unroll
            for (int b = 0; b < NA; ++b)
#pragma unroll
                for (int a = 0; a <= b; ++a)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        H[blk(a, b)][e] = reset ? ((a == b && e == q) ? 1.0 : 0.0) : H[blk(a, b)][e];
#pragma unroll
            for (int a = 0; a < NA; ++a) p[a] = reset ? -g[a] : p[a];
            const double gg2 = qdot<NA>(g, g);
            gp = reset ? -gg2 : gp;
        }
        // a sibling restart of the same target reached stop_loss: stop working on that target
        if (args.flags & 1u) {
            const bool succ = done && !was_done && status == ST_CONVERGED && f < args.stop_loss;
            if (__any(succ)) {
#pragma unroll
                for (int qd = 0; qd < kQuadsPerWave; ++qd) {
                    const int t_qd = __builtin_amdgcn_readlane(tgt, 4 * qd);
                    const int s_qd = __builtin_amdgcn_readlane((int)succ, 4 * qd);
                    if (s_qd && t_qd == tgt && !done) { done = true; status = ST_PREEMPTED; }
                }
            }
        }
    }

    if (live) {
        if (q == 0) {
            args.item_loss[item] = f;
            args.item_iters[item] = iters;
            args.item_status[item] = status;
            args.item_evals[item] = nev;
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int i = 4 * a + q;
            if (i < C::N) args.item_x[item * C::N + i] = x[a];
        }
    }
}

// ---------------------------------------------------------------------------------
// per-target reduction over restarts: argmin of item_loss (ties -> lowest restart)
// ---------------------------------------------------------------------------------
struct ReduceArgs {
    const double* item_loss;   // [n_active * R]
    const double* item_x;      // [n_active * R][n]
    const int32_t* item_evals; // [n_active * R]
    int64_t n_active;
    int32_t restarts;
    int32_t n;                 // parameters at this span
    double* best_loss;         // [n_active]
    double* best_x;            // [n_active][n]
    int32_t* best_restart;     // [n_active]
    unsigned long long* eval_counter;  // += sum of evals
};

__global__ void reduce_best_kernel(ReduceArgs a) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long ev = 0;
    if (t < a.n_active) {
        double best = INFINITY;
        int br = 0;
        for (int r = 0; r < a.restarts; ++r) {
            const double l = a.item_loss[t * a.restarts + r];
            ev += (unsigned long long)a.item_evals[t * a.restarts + r];
            if (l < best) { best = l; br = r; }   // NaN never wins
        }
        a.best_loss[t] = best;
        a.best_restart[t] = br;
        const double* src = a.item_x + (t * a.restarts + br) * a.n;
        for (int i = 0; i < a.n; ++i) a.best_x[t * a.n + i] = src[i];
    }
    // wave-level sum, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) ev += __shfl_down(ev, off);
    if ((threadIdx.x & 63) == 0 && ev) atomicAdd(a.eval_counter, ev);
}

// ---------------------------------------------------------------------------------
// span-loop bookkeeping on the device (TemplateOptimizer._run, optimizer.py:281-303)
// ---------------------------------------------------------------------------------
struct MergeArgs {
    const int32_t* active;      // [n_active] target index of each stage slot (nullptr = identity)
    const double* stage_loss;   // [n_active]
    const double* stage_x;      // [n_active][n]
    int64_t n_active;
    int32_t n;
    int32_t nmax;
    int32_t k;
    double* best_loss;          // [n_targets]
    double* best_x;             // [n_targets][nmax]
    int32_t* best_cycles;       // [n_targets]
};

__global__ void merge_stage_kernel(MergeArgs a) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.n_active) return;
    const int64_t t = a.active ? a.active[s] : s;
    const double l = a.stage_loss[s];
    // "if best_result is None or result.fun < best_result" (optimizer.py:281)
    if (a.best_cycles[t] < 0 || l < a.best_loss[t]) {
        a.best_loss[t] = l;
        a.best_cycles[t] = a.k;
        for (int i = 0; i < a.n; ++i) a.best_x[t * a.nmax + i] = a.stage_x[s * a.n + i];
    }
}

// reset the results of targets [first, first + n) and (optionally) write their indices as the
// initial active list
__global__ void init_results_kernel(double* best_loss, int32_t* best_cycles, int32_t* active, int64_t first,
                                    int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        best_loss[first + t] = INFINITY;
        best_cycles[first + t] = -1;
        if (active) active[t] = (int32_t)(first + t);
    }
}

// Ordered compaction of the targets that still need a longer template:
// keep t iff !(best_loss[t] < threshold)   (optimizer.py:301: break when best < threshold).
// Single workgroup, chunked scan: n is at most a few million.
__global__ void __launch_bounds__(1024) compact_active_kernel(const int32_t* active_in, int64_t n_in,
                                                             const double* best_loss, double threshold,
                                                             int32_t* active_out, int32_t* n_out) {
    __shared__ int32_t counts[1024];
    __shared__ int32_t offs[1025];
    const int tid = threadIdx.x;
    const int64_t chunk = (n_in + 1023) / 1024;
    const int64_t lo = tid * chunk;
    const int64_t hi = (lo + chunk < n_in) ? lo + chunk : n_in;
    int32_t c = 0;
    for (int64_t s = lo; s < hi; ++s) {
        const int32_t t = active_in ? active_in[s] : (int32_t)s;
        c += !(best_loss[t] < threshold);
    }
    counts[tid] = c;
    __syncthreads();
    if (tid == 0) {
        int32_t acc = 0;
        for (int i = 0; i < 1024; ++i) { offs[i] = acc; acc += counts[i]; }
        offs[1024] = acc;
        *n_out = acc;
    }
    __syncthreads();
    int32_t o = offs[tid];
    for (int64_t s = lo; s < hi; ++s) {
        const int32_t t = active_in ? active_in[s] : (int32_t)s;
        if (!(best_loss[t] < threshold)) active_out[o++] = t;
    }
}

}  // namespace slamdev
