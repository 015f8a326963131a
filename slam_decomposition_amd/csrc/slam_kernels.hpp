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
// An accepted step whose slope along p fell by less than (1 - c2) was too short (weak-Wolfe curvature condition violated).  Such a
// pair (s, y) carries a tiny s.y relative to the slope: it is NOT used to update the metric (round 3: SciPy's Wolfe search never
// produces one; used, it pollutes the metric -- mean evaluations per item -11 .. -17 % at equal minima, oracle study in HISTORY.md),
// and the next first trial step is longer.  c2 = 0.7, growth 8 (oracle grid, HISTORY.md; round 2: 0.9 / 4, growth only).
constexpr double kWolfeC2 = 0.7;      // too short:
constexpr double kGrowFactor = 8.0;   // the next first trial step is this much longer (compounding while it keeps
constexpr double kGrowMax = 1048576.0;  // happening).  Covers negative curvature, where the update is skipped.
// Every kRestartPeriod accepted iterations an item's quasi-Newton metric starts over from the identity.  One item in 1e3..1e5
// ends up with a metric that has stopped learning (steps nearly orthogonal to the gradient on a plateau of the loss): 400..1300
// evaluations where SciPy's BFGS needs 50..170 from the same start, and ONE such item sets the duration of its whole stage
// (CNOT k = 2, 1 M items: pct 99.99 of the evaluation counts 197, maximum 1309).  Restarted, it is through in ~40 more.  128 is
// past the 99th percentile of the iteration counts at every span: the mean does not notice.  (oracle/bfgs_port.py: RESTART_PERIOD)
constexpr int kRestartPeriod = 128;
static_assert((kRestartPeriod & (kRestartPeriod - 1)) == 0, "the periodic restart tests iters & (period - 1)");

// One finished (or dropped) work item: the five scalars of its result as ONE 32-byte record = one sector, written by one lane as two
// 16-byte stores (round 3 wrote them into five arrays: five partial-sector stores per item, WRITE_SIZE 2.2x the payload)
struct __attribute__((aligned(32))) ItemRec {
    double loss;
    int32_t iters;
    int32_t status;
    int32_t evals;   // all loss+gradient evaluations of the item
    int32_t acc;     // those whose point was accepted
    int32_t pad[2];
};
static_assert(sizeof(ItemRec) == 32, "ItemRec layout");
__device__ __forceinline__ unsigned long long pack2(int lo, int hi) { return (unsigned long long)(unsigned)lo | ((unsigned long long)(unsigned)hi << 32); }
__device__ __forceinline__ void item_rec_store(ItemRec* r, double loss, int iters, int status, int evals, int acc) {
    // (field by field, all into the one sector: paired into 8- or 16-byte stores the operands need aligned register pairs / quads, and
    // the k = 2 kernel -- at its 256-register limit -- spilled two registers for them)
    r->loss = loss;
    asm volatile("" ::: "memory");  // (keeps the store vectoriser from pairing them again)
    r->iters = iters;
    asm volatile("" ::: "memory");
    r->status = status;
    asm volatile("" ::: "memory");
    r->evals = evals;
    asm volatile("" ::: "memory");
    r->acc = acc;
    asm volatile("" ::: "memory");
    *reinterpret_cast<int2*>(&r->pad[0]) = make_int2(0, 0);
}
// positions dropped at a refill: a sibling restart has already succeeded
__device__ __forceinline__ void item_rec_store_dropped(ItemRec* r, int status) { item_rec_store(r, (double)INFINITY, 0, status, 0, 0); }

enum : int { ST_CONVERGED = 0, ST_MAXITER = 1, ST_LINESEARCH = 2, ST_NONFINITE = 3, ST_STALLED = 4, ST_PREEMPTED = 5 };

// Device-side control block of one span stage.  The span loop (optimizer.py:233-303) is enqueued as one
// chain of kernels without host round trips: the number of targets a stage works on is produced on the
// device by the previous stage's compaction, and every kernel of the stage reads it from here.
struct StageCtl {
    unsigned long long evals;    // += fused loss+gradient evaluations (reduce kernel)
    unsigned long long rounds;   // += lock-step evaluation rounds of every wavefront
    unsigned int work_counter;   // work queue of the optimizer kernel
    int32_t n_active;            // targets of this stage (written by init / the previous stage's compaction)
    unsigned long long evals_accepted;   // += evaluations whose point was accepted (initial point or Armijo step)
    unsigned long long evals_preempted;  // += evaluations of items that ended pre-empted by a sibling restart
    int32_t pad[6];
};
static_assert(sizeof(StageCtl) == 64, "StageCtl layout");

// internal bit of MinimizeArgs::flags (beside SLAM_FLAG_*): the launch records per-iteration traces
constexpr uint32_t kFlagTrace = 0x100u;
// SLAM_FLAG_NO_EXTERIOR (CircuitTemplate(no_exterior_1q=True), basis.py:154,165): the template is G_k K_{k-1} ... K_1 G_1 -- layers 0 and
// K are identities.  Their 12 parameters are pinned at zero (U3(0, 0, 0) = 1): start values and gradient components are zeroed, and a
// quasi-Newton iteration from the identity metric then never moves them (their rows of the metric stay unit rows: s_i = y_i = 0), so the
// run IS the run of the 6 (k - 1)-parameter problem, addition of zeros aside.
constexpr uint32_t kFlagNoExterior = 0x200u;

template <int K>
struct MinimizeArgs {
    const double* targets;    // [n_active][32]: target of stage slot s (gathered, or the resident array itself)
    const int32_t* orig;      // [n_active] original target index of slot s, or nullptr = first_target + s
    int32_t first_target;
    const double* x0;         // [M][n] or nullptr
    StageCtl* ctl;            // n_active (-> M = n_active * restarts work items), work queue, round counter
    int32_t restarts;
    int32_t maxiter;
    double gtol;
    double stop_loss;
    double gtol_far;
    double far_loss;
    double exit_loss;            // a finished restart below this pre-empts its siblings (SLAM_FLAG_EARLY_EXIT)
    uint64_t seed;
    int64_t target_base;         // added to the target index in the Philox key (slam_opt_params.target_base)
    uint32_t flags;
    uint32_t items_per_quad;     // launch shaping (slam_opt_params.items_per_quad)
    int32_t cost_kind;           // 0 BasicCost, 1 SquareCost
    // [n_active], zeroed before launch (SLAM_FLAG_EARLY_EXIT): 0 = no restart of the target has succeeded yet,
    // else restarts - r for the lowest-index successful restart r so far (atomic max)
    int32_t* solved;
    // per-item outputs
    ItemRec* item_rec;        // [M]
    double* item_x;           // [M][n]
    const double* gates;      // [K][32]: G_1..G_K of this span
    // optional per-iteration trace (use_callback, optimizer.py:217-224): after accepted quasi-Newton step number
    // it >= 1 of item m, trace_loss[m][it - 1] = loss and trace_x[m][it - 1][:] = parameters; nullptr = off
    double* trace_loss;       // [M][trace_cap]
    double* trace_x;          // [M][trace_cap][n]
    int32_t trace_cap;
};

template <int K>
struct EvalArgs {
    const double* targets;
    const double* x;          // [M][n]
    const int32_t* target_of; // [M]
    int64_t n_items;
    double* loss;             // [M]
    double* grad;             // [M][n] or nullptr
    double* unitary;          // [M][4][4][2] or nullptr: W = CircuitTemplate.eval(x)
    int32_t cost_kind;        // 0 BasicCost, 1 SquareCost
    const double* gates;      // [K][32]: G_1..G_K of this span
};

// ---------------------------------------------------------------------------------
// loss + gradient (+ template unitary) for explicit parameter vectors
// ---------------------------------------------------------------------------------
template <int K, int GC>
__global__ void __launch_bounds__(kWave) eval_kernel(EvalArgs<K> args) {
    using C = Cfg<K, psq_layout<K, GC>()>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* xchg = lds;
    double2* fhbase = reinterpret_cast<double2*>(lds + C::LDS_XCHG);
    const int lane = threadIdx.x;
    const int q = lane & 3;
    const int quad = lane >> 2;
    double2* tbl = reinterpret_cast<double2*>(lds + lds_work_doubles<K, GC>());
    load_sincos_table(tbl, lane);
    lds_fence();
    const int64_t item = (int64_t)blockIdx.x * kQuadsPerWave + quad;
    const bool live = item < args.n_items;
    const int64_t it = live ? item : 0;
    const int64_t tgt = args.target_of[it];
    const double* tcol = args.targets + tgt * 32 + q * 2;
    double xd[C::NA], gd[C::NA];
#pragma unroll
    for (int a = 0; a < C::NA; ++a) {
        const int i = 4 * a + q;
        xd[a] = (i < C::N) ? args.x[it * C::N + i] : 0.0;
    }
    double f, Wr[4], Wi[4];
    eval_quad<K, true, GC>(xd, tcol, args.gates, xchg + quad * C::XSTRIDE, fhbase + lane, tbl, q, theta_slot_bits<K>(q), args.cost_kind, f, gd, Wr, Wi);
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

// Kernel arguments that are only needed when an item starts or finishes (result pointers, seeds, the work queue) are
// re-read from the kernarg segment where they are used instead of living in SGPRs across the evaluation: the optimizer
// loop is short of scalar registers (the thresholds it tests every round were being spilled into VGPR lanes and read back
// with v_readlane -- a vector-ALU slot each).  The pointer is laundered through an empty asm so that the loads stay at
// their use sites (as for the gate matrices, slam_device.hpp:gate_matrix).
template <int K>
__device__ __forceinline__ const __attribute__((address_space(4))) MinimizeArgs<K>* cold_args() {
    unsigned long long a = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a));
    return (const __attribute__((address_space(4))) MinimizeArgs<K>*)a;
}
// Multi-queue launches (MQ: several independent sub-problems -- their own targets, gates, work queue, outputs -- behind ONE
// launch, slam_decompose_multi): the argument block of the sub-problem the wavefront is working on lives in device memory
// (MinimizeArgs<K>[n_sub]); `cur` is its address, wave-uniform.
template <int K, bool MQ>
__device__ __forceinline__ const __attribute__((address_space(4))) MinimizeArgs<K>* cold_args(unsigned long long cur) {
    if constexpr (MQ) {
        // (wave-uniform by construction; the compiler cannot see it through the loop: two v_readfirstlane per use)
        unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cur), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur >> 32));
        unsigned long long u = ((unsigned long long)hi << 32) | lo;
        asm volatile("" : "+s"(u));
        return (const __attribute__((address_space(4))) MinimizeArgs<K>*)u;
    } else {
        return cold_args<K>();
    }
}

// ---------------------------------------------------------------------------------
// batched quasi-Newton minimisation.  Persistent wavefronts: each of the 16 quads of a wave
// owns one (target, seed) item at a time and pulls the next one from a global counter when
// it finishes, so lanes stay busy although items need very different iteration counts.
// All quads of a wave evaluate in lock-step (one fused loss+gradient per round).
// ---------------------------------------------------------------------------------
// Wave-local stage (WL, span_wave_kernel): ONE wavefront runs all restarts of ONE target -- the queue is the restart range, the
// early-exit flag an LDS word, and instead of per-item records in HBM the wavefront keeps the stage's winner (ordered rule: the
// lowest-index restart below the exit level, else the lowest loss, ties to the lower index) as it goes.
struct WlStage {
    int t;               // resident index of the wavefront's target
    int* flag;           // LDS: restarts - r of the lowest-index successful restart so far (0: none)
    double* win_x;       // LDS: the winner's parameters [n]
    double win_loss;     // wave-uniform
    int win_r;           // wave-uniform; -1: no finite restart yet
    bool hit;            // wave-uniform: a restart below the exit level has finished
    unsigned ev_all, ev_acc, ev_pre;  // per lane (lane q = 0 of a quad): evaluations of the items it finished (32 bits: a quad's share of one
                                      // target's restarts -- the 64-bit form cost the k = 3 stage the three registers that let it share a SIMD with k = 1)
    unsigned rounds;
    unsigned round_cap;  // exit condition every wavefront reaches whatever the state machine does (never met by a sane run)
};
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <int K, int GC, bool MQ, bool WL>
__device__ __forceinline__ void minimize_body(const MinimizeArgs<K>& args, const MinimizeArgs<K>* subs, int n_sub, WlStage& wl) {
    static_assert(!WL || MQ, "the wave-local stage reads its argument block from device memory");
    using C = Cfg<K, psq_layout<K, GC>()>;
    constexpr int NA = C::NA;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* xchg = lds;
    double2* fhbase = reinterpret_cast<double2*>(lds + C::LDS_XCHG);
    const int lane = threadIdx.x;
    int q = lane & 3;
    const int quad = lane >> 2;
    const int theta_bits = theta_slot_bits<K>(q);
    double* xq = xchg + quad * C::XSTRIDE;
    float* xq32 = reinterpret_cast<float*>(xchg) + quad * C::FSTRIDE;  // fp32 overlay with its own conflict-free stride
    double2* fh = fhbase + lane;
    double2* tbl = reinterpret_cast<double2*>(lds + lds_work_doubles<K, GC>());
    load_sincos_table(tbl, lane);
    double2* cold = tbl + kSincosLdsDoubles / 2;  // (gtol, stop_loss), (gtol_far, far_loss)
    if (lane == 0) {
        cold[0] = make_double2(args.gtol, args.stop_loss);
        cold[1] = make_double2(args.gtol_far, args.far_loss);
    }
    lds_fence();
    // ---- launch shape from the device-side target count (the grid is sized for the host's upper bound)
    // MQ: the wavefront starts on sub-problem blockIdx.x mod n_sub and moves on to the next one (cyclically) when that one's queue
    // is exhausted AND its own quads have drained -- gates, targets and outputs are wave-uniform, so a wavefront never holds items of
    // two sub-problems at once; it leaves after n_sub exhausted queues in a row
    constexpr bool kRing = seed_ring<K, GC>();
    if constexpr (kRing) static_assert(kSeedRingOff + C::N <= C::XSTRIDE && kSeedRingOff >= C::XNEED && (kSeedRingOff & 1) == 0, "the ring slots sit in the spare part of a quad's exchange area");
    // ring slot r = the spare doubles of quad 8 + r's exchange area: the start point of queue position ring_base + r
    double* const ring = xchg + 8 * C::XSTRIDE + kSeedRingOff;
    unsigned ring_base = 0, ring_end = 0;  // wave-uniform: the queue positions the ring holds
    unsigned sub_idx = MQ ? (unsigned)blockIdx.x % (unsigned)n_sub : 0u;
    unsigned long long cur = MQ ? (unsigned long long)(subs + sub_idx) : 0ull;  // wave-uniform
    int sub_tries = 0;
    unsigned n_act = WL ? 1u : (unsigned)(MQ ? cold_args<K, MQ>(cur)->ctl->n_active : args.ctl->n_active);
    unsigned n_items = n_act * (unsigned)args.restarts;
    unsigned n_waves = (n_items + kQuadsPerWave - 1) / kQuadsPerWave;
    if constexpr (WL) {
        n_waves = 1;  // this wavefront is the stage
    } else if constexpr (!MQ) {
        if (args.items_per_quad > 1) {
            n_waves = (n_items + kQuadsPerWave * args.items_per_quad - 1) / (kQuadsPerWave * args.items_per_quad);
            if (n_waves < 1 && n_items) n_waves = 1;
        }
        if (n_waves > gridDim.x) n_waves = gridDim.x;
        if (blockIdx.x >= n_waves) return;
    } else {
        n_waves = gridDim.x / (unsigned)n_sub + 1u;  // the sub-problems share the grid (chunk size: from the first one a wavefront sees)
    }
    // The work queue is RESTART-MAJOR: position idx holds restart idx / n_active of stage slot idx % n_active, so every
    // target's restart 0 is handed out before any restart 1, and so on -- the order of the reference's sequential
    // loop (optimizer.py:253).  By the time a target's restart r + 1 comes up, its restart r has usually finished:
    // if it succeeded, r + 1 and all later ones are dropped when pulled (one flag load) instead of running beside
    // it and being pre-empted half-way.  (Target-major order -- a wave holding 16 restarts of one target at once --
    // spent 47 % of the k = 2 and 90 % of the k = 3 evaluations of the 65 536 x 32 sqrt(iSWAP) batch on restarts
    // that a sibling then pre-empted.)  A wave takes `chunk` consecutive positions at a time: big chunks mean few
    // atomics, small batches need every wave busy.
    unsigned kChunkV;
    {
        const unsigned per_wave = n_items / n_waves;
        kChunkV = per_wave >= 256u ? 64u : (per_wave >= 64u ? 32u : 16u);
    }

    // ---- per-quad state (replicated over the quad's 4 lanes unless distributed)
    bool live = false, fresh = false, scaled = false;
    unsigned item = 0;
    int slot = 0, tgt = 0;
    int nev = 0, iters = 0, nback = 0, nstall = 0, status = ST_MAXITER;
    double f = 0.0, alpha = 0.0, gp = 0.0, gnorm = 0.0, grow = 1.0;
    // pp = p.p of the current direction; hs1 = (scale of the identity part of the inverse Hessian) - 1: the effective
    // metric is H + hs1 I, so the one-off scaling of the initial inverse Hessian (Nocedal & Wright eq. 6.20) is a scalar
    // update instead of a multiplication of every stored block
    double pp = 0.0, hs1 = 0.0;
    const double* tcol = cold_args<K, MQ>(cur)->targets + q * 2;  // this lane's column of the quad's target
    double x[NA], g[NA], p[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) { x[a] = 0.0; g[a] = 0.0; p[a] = 0.0; }
    HMat<NA> H;
    h_set_identity_where<NA>(H, q, true);
    bool exhausted = false;  // wave-uniform
    const unsigned kChunk = kChunkV;  // wave-uniform: 16 (small batches: spread over all waves) .. 64
    unsigned cur_next = 0, cur_end = WL ? n_items : 0u;  // wave-uniform (WL: the whole restart range is this wavefront's chunk)
    unsigned pre_base = 0;               // lane 0: base of the prefetched chunk
    unsigned rounds = 0;                 // wave-uniform
    if constexpr (!WL) {
        if (lane == 0) pre_base = atomicAdd(&cold_args<K, MQ>(cur)->ctl->work_counter, kChunk);
    }

    while (true) {
        if constexpr (MQ && !WL) {
            // this sub-problem's queue is exhausted and the wavefront's quads have drained: on to the next one
            if (exhausted && !__any(live)) {
                if (lane == 0 && rounds) atomicAdd(&cold_args<K, MQ>(cur)->ctl->rounds, (unsigned long long)rounds);
                rounds = 0;
                if (++sub_tries >= n_sub) break;
                sub_idx = (sub_idx + 1u == (unsigned)n_sub) ? 0u : sub_idx + 1u;
                cur = (unsigned long long)(subs + sub_idx);
                n_act = (unsigned)cold_args<K, MQ>(cur)->ctl->n_active;
                n_items = n_act * (unsigned)args.restarts;
                exhausted = false;
                cur_next = 0;
                cur_end = 0;
                ring_base = 0;
                ring_end = 0;  // (another queue: the parked start points are not its positions')
                if (lane == 0) pre_base = atomicAdd(&cold_args<K, MQ>(cur)->ctl->work_counter, kChunk);
            }
        }
        // q is re-materialised every iteration: otherwise the lane-dependent LDS addresses derived from it
        // (gradient gather, stash slots) are hoisted out of the loop, kept live across it, spilled to scratch
        // and reloaded -- one exposed memory latency each -- in every round
        if constexpr (K == 2) {
            asm volatile("" : "+v"(q));  // (measured: pays at k = 2 only)
            __builtin_assume((unsigned)q < 4u);  // keeps the slot-validity tests 4 a + q < N compile-time for a < NA - 1
        }
        // ---- 1. idle quads pull work until every quad has an item or the queue is empty
        //         (single exit, single back edge: the loop-carried state is large).
        // Items are handed out from a wave-private chunk [cur_next, cur_end) of kChunk consecutive
        // items; the next chunk is requested (one atomicAdd per wave) as soon as the current one is
        // entered, and its base is only waited for when the current chunk runs dry.
        bool taken = false;
        {
            // the refill code is wave-wide (everyone waits while it runs), so at short spans -- where
            // items last only ~40 rounds -- it pays to let kRefillBatch quads go idle before running it
            // (3 / 2 / 1 before the seeds were shared out over the wave; measured again since: k = 1 10.96 -> 10.72 ms, k = 2 7.95 -> 7.87)
#ifndef SLAM_REFILL_BATCH_K1
#define SLAM_REFILL_BATCH_K1 2
#endif
#ifndef SLAM_REFILL_BATCH_K2
#define SLAM_REFILL_BATCH_K2 1
#endif
            constexpr int kRefillBatch = (K == 1) ? SLAM_REFILL_BATCH_K1 : (K == 2 ? SLAM_REFILL_BATCH_K2 : 1);
            const int n_idle = __popcll(__ballot(!live && q == 0));
            const bool go = n_idle >= kRefillBatch || n_idle == __popcll(__ballot(q == 0));
            while (go && !exhausted && __any(!live)) {
                if constexpr (WL) {
                    if (cur_next >= cur_end) { exhausted = true; break; }
                } else if (cur_next >= cur_end) {
                    const unsigned b = (unsigned)__builtin_amdgcn_readfirstlane((int)pre_base);
                    cur_next = b;
                    cur_end = (b + kChunk < n_items) ? b + kChunk : n_items;
                    if (b >= n_items) { exhausted = true; break; }
                    if constexpr (MQ) sub_tries = 0;  // this queue still had work
                    if (lane == 0) pre_base = atomicAdd(&cold_args<K, MQ>(cur)->ctl->work_counter, kChunk);
                }
                // ---- scan up to 64 queue positions at once: lane l looks at position cur_next + l.  Positions whose
                // target already has a successful restart are dropped here (one flag load for the whole window, their
                // outputs written by the scanning lanes); the idle quads get the first positions that still need work.
                bool use_ring = false;  // wave-uniform
                if constexpr (kRing) {
                    use_ring = cold_args<K, MQ>(cur)->x0 == nullptr;
                    if (use_ring && cur_next >= ring_end) {
                        // the next kSeedRing positions' start points: lane l runs Philox block l % (N / 2) of position cur_next + l / (N / 2)
                        constexpr int kPairs = C::N / 2;
                        ring_base = cur_next;
                        ring_end = (cur_next + (unsigned)kSeedRing < cur_end) ? cur_next + (unsigned)kSeedRing : cur_end;
                        const int gr_ = lane / kPairs, gm = lane - gr_ * kPairs;
                        if ((unsigned)gr_ < ring_end - ring_base) {
                            const unsigned gp = ring_base + (unsigned)gr_;
                            const unsigned grs = gp / n_act, gsl = gp - grs * n_act;
                            int gt_;
                            if constexpr (WL) gt_ = wl.t;
                            else gt_ = cold_args<K, MQ>(cur)->orig ? cold_args<K, MQ>(cur)->orig[gsl] : cold_args<K, MQ>(cur)->first_target + (int)gsl;
                            const uint64_t gseed = cold_args<K, MQ>(cur)->seed;
                            uint32_t w[4];
                            philox4x32_10((uint32_t)gm, grs, (uint32_t)(gt_ + (int)cold_args<K, MQ>(cur)->target_base), (uint32_t)K, (uint32_t)gseed,
                                          (uint32_t)(gseed >> 32), w);
                            *reinterpret_cast<double2*>(ring + gr_ * C::XSTRIDE + 2 * gm) = make_double2(x0_from_words(w[0], w[1]), x0_from_words(w[2], w[3]));
                        }
                        lds_fence();
                    }
                }
                unsigned wlen = (cur_end - cur_next < (unsigned)kWave) ? cur_end - cur_next : (unsigned)kWave;
                if (kRing && use_ring && ring_end - cur_next < wlen) wlen = ring_end - cur_next;  // hand out only what the ring holds
                const bool valid = (unsigned)lane < wlen;
                const unsigned pos = cur_next + (unsigned)lane;   // queue position
                const unsigned prs = pos / n_act;                 // restart
                const unsigned psl = pos - prs * n_act;           // stage slot (target)
                bool skipv = false;
                if ((args.flags & 1u) && valid) {
                    // flag = restarts - r of the lowest-index successful restart r (0: none).  Ordered mode
                    // (SLAM_FLAG_ORDERED): only restarts with a HIGHER index than a successful one are dropped, so the
                    // winner is the lowest-index successful restart whatever the scheduling -- the restart the
                    // reference's sequential loop stops at (optimizer.py:287-295).
                    int fl;
                    if constexpr (WL) fl = *reinterpret_cast<volatile int*>(wl.flag);
                    else fl = __hip_atomic_load(&(MQ ? cold_args<K, MQ>(cur)->solved : args.solved)[psl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int mine = args.restarts - (int)prs;
                    skipv = (args.flags & 2u) ? (fl > mine) : (fl != 0);
                }
                const unsigned long long lt = (1ull << lane) - 1ull;
                const unsigned long long avail = __ballot(valid && !skipv);
                const unsigned long long idle = __ballot(!live && q == 0);
                const int n_av = __popcll(avail), n_idle_now = __popcll(idle);
                const int n_take = n_av < n_idle_now ? n_av : n_idle_now;
                const int myrank = __popcll(avail & lt);
                const bool handed = valid && !skipv && myrank < n_take;
                const unsigned long long handed_mask = __ballot(handed);
                // the window is consumed up to the last position handed out (all of it when every position with work
                // found a quad); the rest is looked at again next time
                const unsigned consumed = (n_take == n_av) ? wlen : (unsigned)(64 - __builtin_clzll(handed_mask));
                if constexpr (!WL) {
                    if (valid && skipv && (unsigned)lane < consumed) {
                        // a sibling restart already succeeded: nothing to do for this item
                        // (outputs and explicit seeds keep the [slot][restart] layout whatever the processing order)
                        const unsigned o = psl * (unsigned)args.restarts + prs;
                        item_rec_store_dropped(cold_args<K, MQ>(cur)->item_rec + o, ST_PREEMPTED);
                    }
                }
                int* wp = reinterpret_cast<int*>(xchg);  // wave-private, dead between rounds: [16] slots, [16] restarts
                if (handed) {
                    wp[myrank] = (int)psl;
                    wp[16 + myrank] = (int)prs;
                    if constexpr (kRing) wp[32 + myrank] = (int)(pos - ring_base);  // where this position's start point sits
                }
                lds_fence();
                const int qrank = __popcll(idle & ((1ull << (lane & ~3)) - 1ull));  // rank of this quad among the idle ones
                const bool get = !live && qrank < n_take;
                const unsigned sl = get ? (unsigned)wp[qrank] : 0u;
                const unsigned rs = get ? (unsigned)wp[16 + qrank] : 0u;
                const int rslot = (kRing && get) ? wp[32 + qrank] : 0;
                lds_fence();
                cur_next += consumed;
                {
                    // Start points.  Each Philox block yields the parameter pair (2m, 2m + 1) of one item; computed per lane
                    // for its own slots, a lane runs NA blocks (half of them twice within the quad) whether or not its quad
                    // takes an item.  With the wide exchange area (kSharedSeeds) the blocks of ALL taking quads are dealt
                    // over the 64 lanes instead -- n_take N / 2 blocks, one pass for up to 10 / 7 / 5 quads at k = 1 / 2 / 3 --
                    // and handed to their owners through LDS: the same numbers, a third to a sixth of the instructions.
                    constexpr bool kSharedSeeds = psq_layout<K, GC>();
                    const bool shared = kSharedSeeds && cold_args<K, MQ>(cur)->x0 == nullptr;  // wave-uniform
                    if (get) {
                        const unsigned oidx = sl * (unsigned)args.restarts + rs;
                        item = oidx;
                        slot = (int)sl;
                        const unsigned restart = rs;
                        // three independent loads (no load feeds another's address)
                        if constexpr (WL) {
                            tgt = wl.t;
                            tcol = cold_args<K, MQ>(cur)->targets + (int64_t)wl.t * 32 + q * 2;
                        } else {
                            tgt = cold_args<K, MQ>(cur)->orig ? cold_args<K, MQ>(cur)->orig[sl] : cold_args<K, MQ>(cur)->first_target + (int)sl;
                            tcol = cold_args<K, MQ>(cur)->targets + (int64_t)sl * 32 + q * 2;
                        }
                        if (!shared) {
#pragma unroll
                            for (int a = 0; a < NA; ++a) {
                                const int i = 4 * a + q;
                                double xv = 0.0;
                                if (i < C::N) {
                                    if constexpr (kSharedSeeds)  // (not shared: explicit start points)
                                        xv = cold_args<K, MQ>(cur)->x0[(int64_t)oidx * C::N + i];
                                    else
                                        xv = cold_args<K, MQ>(cur)->x0 ? cold_args<K, MQ>(cur)->x0[(int64_t)oidx * C::N + i]
                                                     : x0_philox(cold_args<K, MQ>(cur)->seed, (uint32_t)(tgt + (int)cold_args<K, MQ>(cur)->target_base), restart, (uint32_t)K, (uint32_t)i);
                                }
                                x[a] = xv;
                            }
                        }
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
                            p[a] = 0.0;
                            g[a] = 0.0;
                        }
                        alpha = 0.0; gp = 0.0; f = 0.0; grow = 1.0; pp = 0.0; hs1 = 0.0;
                        nev = 0; iters = 0; nback = 0; nstall = 0; status = ST_MAXITER;
                        scaled = false; fresh = true; live = true; taken = true;
                    }
                    if constexpr (kRing) {
                        if (shared && get) {
#pragma unroll
                            for (int a = 0; a < NA; ++a) x[a] = (4 * a + q < C::N) ? ring[rslot * C::XSTRIDE + 4 * a + q] : 0.0;
                        }
                    } else if constexpr (kSharedSeeds) {
                        if (shared) {
                            constexpr int kPairs = C::N / 2;
                            static_assert(5 * C::N <= C::XSTRIDE && 4 * C::N * 8 >= 64 * 4, "staging area [4N, 5N) inside the quad's exchange area and clear of wp");
                            // wp[32 + r]: Philox target word of the r-th taking quad, wp[48 + r]: its quad index
                            if (get && q == 0) {
                                wp[32 + qrank] = tgt + (int)cold_args<K, MQ>(cur)->target_base;
                                wp[48 + qrank] = quad;
                            }
                            lds_fence();
                            const uint64_t seed = cold_args<K, MQ>(cur)->seed;
                            const int jobs = n_take * kPairs;
                            for (int j0 = 0; j0 < jobs; j0 += kWave) {
                                const int jb = j0 + lane;
                                if (jb < jobs) {
                                    const int r = jb / kPairs, m = jb - r * kPairs;
                                    uint32_t w[4];
                                    philox4x32_10((uint32_t)m, (uint32_t)wp[16 + r], (uint32_t)wp[32 + r], (uint32_t)K, (uint32_t)seed,
                                                  (uint32_t)(seed >> 32), w);
                                    // staged behind the trig table and the first partial sums of the quad's exchange area
                                    // (doubles [4N, 5N): clear of the 64 ints of wp at the head of quad 0's)
                                    *reinterpret_cast<double2*>(xchg + wp[48 + r] * C::XSTRIDE + 4 * C::N + 2 * m) =
                                        make_double2(x0_from_words(w[0], w[1]), x0_from_words(w[2], w[3]));
                                }
                            }
                            lds_fence();
                            if (get) {
#pragma unroll
                                for (int a = 0; a < NA; ++a) x[a] = (4 * a + q < C::N) ? xq[4 * C::N + 4 * a + q] : 0.0;
                            }
                            lds_fence();
                        }
                    }
                    if ((args.flags & kFlagNoExterior) && get) {
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
                            const int i = 4 * a + q;
                            x[a] = (i < 6 || i >= 6 * K) ? 0.0 : x[a];
                        }
                    }
                }
            }
            if (__any(taken)) h_set_identity_where<NA>(H, q, taken);
        }
        if (!__any(live)) {
            if constexpr (MQ && !WL) continue;  // (nothing is live, hence nothing loop-carried to copy: the next sub-problem, or out)
            else break;
        }
        if constexpr (WL) {
            if (rounds >= wl.round_cap) break;
        }
        ++rounds;

        // early-exit flag of this quad's target (consumed at the end of the round)
        // (loaded for every quad -- slot 0 for idle ones -- and only looked at after the evaluation, so that
        // the round does not start with a wait for global memory)
        int sflag = 0;
        if (args.flags & 1u) {
            if constexpr (WL) sflag = *reinterpret_cast<volatile int*>(wl.flag);
            else sflag = __hip_atomic_load(&(MQ ? cold_args<K, MQ>(cur)->solved : args.solved)[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }

        // ---- 2. one fused loss + gradient evaluation at the trial point x + alpha p (x itself when
        //         fresh: alpha = 0, p = 0)
        double gt[NA];
        double ft, Wr[4], Wi[4];
        {
            double xt[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) xt[a] = fma(alpha, p[a], x[a]);
            eval_quad<K, false, GC>(xt, tcol, MQ ? cold_args<K, MQ>(cur)->gates : args.gates, xq, fh, tbl, q, theta_bits,
                                    WL ? __builtin_amdgcn_readfirstlane(args.cost_kind) : args.cost_kind, ft, gt, Wr, Wi);
        }
        const bool active = live;
        const bool finite = isfinite(ft);
        // a non-finite trial point is never accepted; its gradient is zeroed here once so that everything below
        // stays finite without per-element guards (0 * NaN would otherwise leak into H through w and v)
#pragma unroll
        for (int a = 0; a < NA; ++a) gt[a] = finite ? gt[a] : 0.0;
        if (args.flags & kFlagNoExterior) {  // wave-uniform
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int i = 4 * a + q;
                gt[a] = (i < 6 || i >= 6 * K) ? 0.0 : gt[a];
            }
        }
        const bool armijo = finite && (ft <= f + kArmijoC1 * alpha * gp);
        const bool acc = active && (fresh ? finite : armijo);
        const bool step = acc && !fresh;  // a real quasi-Newton step (not the initial evaluation)
        // evaluation counters of the item, packed: bits 0..19 all evaluations, bits 20..31 the accepted ones
        // (at most maxiter + 1 <= 4095: the host caps maxiter)
        nev += active ? (acc ? 0x100001 : 1) : 0;

        // ---- 3. quasi-Newton update.  s = am p and y = g' - g are formed on the fly; for quads that do not
        //         step, am = 0 and curv = false, hence rho = cf = 0 and w = v = 0: H is left unchanged (their
        //         y-dependent scalars are finite garbage that is multiplied by zero).
        const double am = step ? alpha : 0.0;
        double qv[NA];
        __builtin_amdgcn_sched_barrier(0);  // keep the phases apart: interleaving them only adds live registers
        h_matvec<NA>(H, gt, xq32, q, qv);
        __builtin_amdgcn_sched_barrier(0);
        // s = am p, so every product with s follows from p.g' (one reduction) and the direction's own p.g, p.p, which
        // the previous round left in gp and pp:  s.g' = am p.g',  s.y = am (p.g' - p.g),  s.s = am^2 p.p
        double pgt = 0.0, yy = 0.0;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const double ya = gt[a] - g[a];
            pgt = fma(p[a], gt[a], pgt);
            yy = fma(ya, ya, yy);
        }
        quad_sum2(pgt, yy);
        const double sg = am * pgt;
        const double sy = am * (pgt - gp);
        const double ss = (am * am) * pp;
        const bool too_short = sy < (1.0 - kWolfeC2) * alpha * (-gp);  // weak-Wolfe curvature condition violated
        const bool curv = step && !too_short && sy > 0.0 && (sy * sy > (kCurvEps * kCurvEps) * (ss * yy));
        const bool first = curv && !scaled;
        scaled = scaled || curv;
        // first update of an item: scale the initial inverse Hessian (= the identity then) by s.y / y.y -- as the scalar
        // hs1 (round 1 multiplied all 30 / 42 stored blocks by a factor that is 1 in every other round)
        const double fac = first ? (sy * fast_rcp(yy)) : 1.0;
        hs1 = first ? fac - 1.0 : hs1;
        // u = H_eff y = H_eff g' - H_eff g = q + fac p   (p = -H_eff g with the metric before this round's scaling)
        double yu = 0.0;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            qv[a] = fma(hs1, gt[a], qv[a]);  // q = H_eff g' = H g' + hs1 g'
            const double ua = fma(fac, p[a], qv[a]);
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
                const double sa = am * p[a];
                const double ua = fma(fac, p[a], qv[a]);
                const double wa = cf * sa - rho * ua;  // rho = cf = 0 unless curv: w = v = 0, H unchanged
                const double va = -rho * ua;
                s32[a] = (float)sa;
                w32[a] = (float)wa;
                v32[a] = (float)va;
                wg = fma(wa, gt[a], wg);
            }
            __builtin_amdgcn_sched_barrier(0);
            h_update<NA>(H, s32, w32, v32, xq32, q);
            __builtin_amdgcn_sched_barrier(0);
        }
        wg = quad_sum(wg);
        // convergence thresholds: requested here, tested at the end of the state machine
        const double2 th0 = cold[0], th1 = cold[1];
        __builtin_amdgcn_sched_barrier(0);

        // ---- 4. per-quad state machine
        // x <- x + s outside the branch: s = am p is zero unless the step was accepted, so no per-component select
#pragma unroll
        for (int a = 0; a < NA; ++a) x[a] += am * p[a];
        bool done = false;
        if (acc) {
            nstall = (step && (f - ft) <= kStallDf) ? nstall + 1 : 0;
            f = ft;
            if (step) ++iters;
            nback = 0;
            grow = (step && too_short) ? fmin(grow * kGrowFactor, kGrowMax) : 1.0;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const double sa = am * p[a];
                const double ua = fma(fac, p[a], qv[a]);
                const double va = -rho * ua;
                g[a] = gt[a];
                p[a] = -(qv[a] + sa * wg + va * sg);
            }
            if (args.flags & kFlagTrace) {  // wave-uniform flag test: nothing when off
                const auto* ca = cold_args<K, MQ>(cur);
                const int t_cap = ca->trace_cap;
                double* const t_loss = ca->trace_loss;
                double* const t_x = ca->trace_x;
                if (step && iters <= t_cap) {
                const int64_t row = (int64_t)item * t_cap + (iters - 1);
                if (q == 0) t_loss[row] = f;
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const int i = 4 * a + q;
                    if (i < C::N) t_x[row * C::N + i] = x[a];
                }
                }
            }
        } else if (active) {
            if (fresh) {
                f = ft;
                status = ST_NONFINITE;
                done = true;
            } else {
                const double denom = 2.0 * (ft - f - gp * alpha);
                const double anew = (finite && denom > 0.0 && isfinite(denom)) ? (-gp * alpha * alpha * fast_rcp(denom)) : 0.5 * alpha;
                alpha = fmin(fmax(anew, 0.1 * alpha), 0.5 * alpha);
                grow = 1.0;
                ++nback;
            }
        }
        // quad reductions are executed by all lanes
        {
            double m = max_abs(g[0], g[NA > 1 ? 1 : 0]);
#pragma unroll
            for (int a = 2; a < NA; ++a) m = max_abs(m, g[a]);
            gnorm = quad_max(m);
        }
        gp = qdot<NA>(g, p);
        pp = qdot<NA>(p, p);
        if (acc) {
            alpha = (pp > 1e-300) ? fmin(grow, kStepMax * fast_rsqrt(pp)) : grow;
            if (f < th0.y || gnorm < th0.x || (gnorm < th1.x && f > th1.y)) {
                status = ST_CONVERGED; done = true;
            } else if (nstall >= 2) { status = ST_STALLED; done = true; }
            else if (iters >= args.maxiter) { status = ST_MAXITER; done = true; }
        } else if (active && !fresh) {
            if (nback > kMaxBacktrack) { status = (gnorm < kStallGnorm) ? ST_STALLED : ST_LINESEARCH; done = true; }
        }
        fresh = false;
        // not a descent direction (H lost positive definiteness numerically), or the periodic restart: steepest descent again
        const bool periodic = step && !done && ((iters & (kRestartPeriod - 1)) == 0);
        const bool reset = active && !done && (!(gp < 0.0) || periodic);
        if (__any(reset)) {
            h_set_identity_where<NA>(H, q, reset);
            hs1 = reset ? 0.0 : hs1;
            scaled = periodic ? false : scaled;  // the restarted metric gets its initial scaling again
#pragma unroll
            for (int a = 0; a < NA; ++a) p[a] = reset ? -g[a] : p[a];
            const double gg2 = qdot<NA>(g, g);
            gp = reset ? -gg2 : gp;
            pp = reset ? gg2 : pp;
            alpha = periodic ? ((gg2 > 1e-300) ? fmin(grow, kStepMax * fast_rsqrt(gg2)) : grow) : alpha;
        }
        // ---- 5. early exit across the restarts of one target (optimizer.py:287-295)
        if (args.flags & 1u) {
            asm volatile("" : "+v"(sflag));  // not to be tested (= waited for) any earlier than here
            const int mine = args.restarts - (int)(item - (unsigned)slot * (unsigned)args.restarts);
            const bool beaten = (args.flags & 2u) ? (sflag > mine) : (sflag != 0);
            if (active && !done && beaten) { status = ST_PREEMPTED; done = true; }
            if (active && done && status != ST_PREEMPTED && f < args.exit_loss && q == 0) {
                if constexpr (WL) __hip_atomic_fetch_max(wl.flag, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_fetch_max(&(MQ ? cold_args<K, MQ>(cur)->solved : args.solved)[slot], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ---- 6. finished items leave; their quads pull new work next round
        if constexpr (WL) {
            // the stage's winner, kept up to date as restarts finish (those of one round one after the other; item == restart here)
            const bool fin = active && done;
            if (fin && q == 0) {
                const unsigned e = (unsigned)(nev & 0xFFFFF);
                wl.ev_all += e;
                if (status == ST_PREEMPTED) wl.ev_pre += e;
                else wl.ev_acc += (unsigned)nev >> 20;
            }
            unsigned long long m = __ballot(fin && q == 0 && status != ST_PREEMPTED);
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1ull;
                const double lf = readlane_f64(f, l);
                const int lr = __builtin_amdgcn_readlane((int)item, l);
                const bool below = lf < args.exit_loss;
                const bool take = wl.hit ? (below && lr < wl.win_r) : (below || lf < wl.win_loss || (lf == wl.win_loss && lr < wl.win_r));
                if (take) {  // wave-uniform
                    wl.win_loss = lf;
                    wl.win_r = lr;
                    wl.hit = wl.hit || below;
                    if ((lane >> 2) == (l >> 2)) {
#pragma unroll
                        for (int a = 0; a < NA; ++a)
                            if (4 * a + q < C::N) wl.win_x[4 * a + q] = x[a];
                    }
                }
            }
            if (fin) {
                live = false;
                alpha = 0.0;
            }
        } else if (active && done) {
            if (q == 0) item_rec_store(cold_args<K, MQ>(cur)->item_rec + item, f, iters, status, nev & 0xFFFFF, (int)((unsigned)nev >> 20));
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int i = 4 * a + q;
                if (i < C::N) cold_args<K, MQ>(cur)->item_x[(int64_t)item * C::N + i] = x[a];
            }
            live = false;
            alpha = 0.0;  // an idle quad keeps evaluating x + 0 p (its direction is reset when it takes the next item)
        }
    }
    if constexpr (WL) {
        wl.rounds = rounds;
        lds_fence();
    } else if constexpr (!MQ) {
        if (lane == 0 && rounds) atomicAdd(&cold_args<K>()->ctl->rounds, (unsigned long long)rounds);
    }
}

template <int K, int GC, bool MQ = false>
__global__ void __launch_bounds__(kWave, (K <= 2 ? 2 : 1)) minimize_kernel(MinimizeArgs<K> args, const MinimizeArgs<K>* subs, int n_sub) {
    WlStage none{};
    minimize_body<K, GC, MQ, false>(args, subs, n_sub, none);
}

// ---------------------------------------------------------------------------------
// The WHOLE span loop of a target in one wavefront (small batches: at most a few targets per SIMD).  Wavefront <-> target: all R
// restarts of span k run in its 16 quads (wave-local queue, LDS early-exit flag), the winner is reduced as restarts finish, merged
// into the target's running best (optimizer.py:281-284), and if that is still above the threshold (optimizer.py:301) the same
// wavefront goes straight on to span k + 1 -- no stage barrier across targets, no bookkeeping launch, no per-item records.  A lone
// 1024 x 16 batch then takes the longest PER-TARGET chain (sum over spans of that target's slowest needed restart) instead of the sum
// of the three stages' slowest items.  Same items, same seeds, same quasi-Newton loop (minimize_body): the results are those of the
// per-span kernels bit for bit.  One wavefront per SIMD (the k = 3 body's registers).
// ---------------------------------------------------------------------------------
struct WaveLoopArgs {
    const MinimizeArgs<1>* stage_args;  // [SLAM_MAX_SPAN_EVAL + 1]: block k = the argument block of span k (one layout for all spans)
    StageCtl* ctl;                      // [k]: counters; ctl[0].work_counter hands out the targets
    int32_t k_min, k_max;
    int32_t first, count;               // target window
    double threshold;
    double* best_loss;
    double* best_x;
    int32_t* best_cycles;
    double* span_loss;
    int32_t nmax;
    uint32_t round_cap;                 // rounds after which a wavefront leaves a stage whatever happens
    // speculative spans (span_spec_kernel): stage k of target i leaves its result in row (k - 1) * count + i
    double* spec_loss;                  // [3 * count]: the stage's winner loss (INFINITY: no finite restart)
    double* spec_x;                     // [3 * count][nmax]
    unsigned long long* spec_ev;        // [3 * count][3]: evaluations (all, accepted, pre-empted) of the stage
};
constexpr int kSpanLossStride = 16;  // = SLAM_MAX_SPAN_EVAL
constexpr int kWlLdsDoubles = 40;  // winner row (<= 36 parameters) + the flag word

template <int K, int GC, int KL = 3, bool SPEC = false>
__device__ __forceinline__ void wave_stage(const WaveLoopArgs& a, int t, double* lds, double& best_loss, int& best_cycles) {
    using C = Cfg<K, psq_layout<K, GC>()>;
    const int lane = threadIdx.x;
    const MinimizeArgs<K>* blk = reinterpret_cast<const MinimizeArgs<K>*>(a.stage_args) + K;
    // the stage's argument block through scalar loads (wave-uniform: it feeds scalar operands and uniform branches)
    MinimizeArgs<K> args;
    {
        unsigned long long pa = (unsigned long long)blk;
        asm volatile("" : "+s"(pa));
        __builtin_memcpy(&args, (const __attribute__((address_space(4))) void*)pa, sizeof(args));
    }
    WlStage wl{};
    wl.t = t;
    wl.win_x = lds + lds_doubles<KL, GC>();
    wl.flag = reinterpret_cast<int*>(wl.win_x + 36);
    wl.win_loss = INFINITY;
    wl.win_r = -1;
    wl.hit = false;
    wl.round_cap = a.round_cap;
    if (lane == 0) *wl.flag = 0;
    if (lane < 36) wl.win_x[lane] = 0.0;
    lds_fence();
    minimize_body<K, GC, true, true>(args, blk, 1, wl);
    // counters of the stage: one atomic each per wavefront
    unsigned long long e0 = wl.ev_all, e1 = wl.ev_acc, e2 = wl.ev_pre;
    for (int off = 32; off > 0; off >>= 1) {
        e0 += __shfl_down(e0, off);
        e1 += __shfl_down(e1, off);
        e2 += __shfl_down(e2, off);
    }
    const double stage_loss = wl.win_r >= 0 ? wl.win_loss : (double)INFINITY;
    if constexpr (SPEC) {
        // a stage run ahead of the span loop's decision: its result and counters go to the staging rows; span_merge_kernel
        // applies the loop's bookkeeping (and books the evaluations of stages the loop would not have run as pre-empted)
        const int64_t row = (int64_t)(K - 1) * a.count + (t - a.first);
        if (lane == 0) {
            a.spec_loss[row] = stage_loss;
            a.spec_ev[row * 3 + 0] = e0;
            a.spec_ev[row * 3 + 1] = e1;
            a.spec_ev[row * 3 + 2] = e2;
            atomicAdd(&a.ctl[K].rounds, (unsigned long long)wl.rounds);
        }
        if (lane < a.nmax) a.spec_x[row * a.nmax + lane] = (lane < C::N) ? wl.win_x[lane] : 0.0;
        return;
    }
    if (lane == 0) {
        StageCtl* c = a.ctl + K;
        atomicAdd(&c->evals, e0);
        atomicAdd(&c->evals_accepted, e1);
        atomicAdd(&c->evals_preempted, e2);
        atomicAdd(&c->rounds, (unsigned long long)wl.rounds);
        atomicAdd(&c->n_active, 1);
    }
    // merge into the running best (optimizer.py:281-284), record the "Cycle (k =...)" value
    if (best_cycles < 0 || stage_loss < best_loss) {
        best_loss = stage_loss;
        best_cycles = K;
        if (lane < a.nmax) a.best_x[(int64_t)t * a.nmax + lane] = (lane < C::N) ? wl.win_x[lane] : 0.0;
    }
    if (lane == 0) a.span_loss[(int64_t)t * kSpanLossStride + (K - 1)] = best_loss;
    lds_fence();
}

template <int GC>
__global__ void __launch_bounds__(kWave, 1) span_wave_kernel(WaveLoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    // one workgroup (= wavefront) per target: the grid IS the batch (at most a few wavefronts per SIMD by the host's eligibility rule).
    // (A persistent form -- wavefronts pulling targets from a counter in a loop around the stages -- hung on the GPU: the compiler
    // structurises that outer loop with per-lane exit masks around three inlined optimizer loops; a grid of independent wavefronts
    // needs no loop and lets the hardware hand out the targets.)
    const int i = (int)blockIdx.x;
    if (i >= a.count) return;
    const int t = a.first + i;
    double best_loss = INFINITY;
    int best_cycles = -1;
    // "span not run" for every span first (what init_results_kernel does for the per-span launches); the stages overwrite theirs
    if (lane < kSpanLossStride) a.span_loss[(int64_t)t * kSpanLossStride + lane] = NAN;
    if (a.k_min <= 1 && a.k_max >= 1) wave_stage<1, GC>(a, t, lds, best_loss, best_cycles);
    if (a.k_min <= 2 && a.k_max >= 2 && !(best_loss < a.threshold)) wave_stage<2, GC>(a, t, lds, best_loss, best_cycles);
    if (a.k_min <= 3 && a.k_max >= 3 && !(best_loss < a.threshold)) wave_stage<3, GC>(a, t, lds, best_loss, best_cycles);
    if (lane == 0) {
        a.best_loss[t] = best_loss;
        a.best_cycles[t] = best_cycles;
    }
}

// ---------------------------------------------------------------------------------
// Speculative spans (small batches that leave the chip mostly empty): the stages of the span loop do not depend on each other's
// RESULTS -- only on whether they are needed (optimizer.py:301-303 breaks at the first span below the threshold) -- so all spans of
// all targets start at once, one wavefront per (target, span), one launch per span on its own stream (the k = 1, 2 kernels keep their
// own register budget: two wavefronts per SIMD beside one of k = 3), and span_merge_kernel then walks every target's stage
// results in span order exactly as the loop would: running best with strict "<" (optimizer.py:281-284), stop at the first success.
// Same items, same Philox start points (keyed by span), same winner rule per stage => the per-span launches' results bit for bit; a
// lone batch takes the longest single stage instead of the sum of three.  Stages the loop would not have run are wasted work: their
// evaluations are booked as pre-empted.  (For CNOT -- BASELINE configs[1] -- no Haar target is solved before k = 3: nothing is wasted.)
// ---------------------------------------------------------------------------------
template <int K, int GC>
__global__ void __launch_bounds__(kWave, K >= 3 ? 1 : 2) span_spec_kernel(WaveLoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int i = (int)blockIdx.x;
    if (i >= a.count) return;
    double best_loss = INFINITY;
    int best_cycles = -1;
    wave_stage<K, GC, K, true>(a, a.first + i, lds, best_loss, best_cycles);
}

// The span loop's bookkeeping over stage results that were produced side by side (speculative spans of small batches:
// span_spec_kernel; overlapped spans of medium ones: one per-span optimizer launch + reduction per helper context).  Per target, in
// span order: running best with strict "<" (optimizer.py:281-284), "Cycle (k =...)" value, stop at the first success (:301-303); the
// evaluations of stages the loop would not have run are booked as pre-empted.
struct SpanMergeArgs {
    int32_t k_min, k_max;
    int32_t first, count;            // target window; stage k's row of target first + i is i
    int32_t nmax;
    double threshold;
    const double* loss[4];           // [k]: stage k's winner loss per target (INFINITY: no finite restart)
    const double* x[4];              // [k]: its parameters, row stride xstride[k]
    int32_t xstride[4], xn[4];       // xn[k]: valid parameters per row (6 (k + 1))
    const unsigned long long* ev[4]; // [k]: (all, accepted, pre-empted) evaluations per target
    const StageCtl* src_ctl[4];      // [k]: the producing stage's control block (rounds), or nullptr
    StageCtl* ctl;                   // the call's control blocks [k]
    double* best_loss;
    double* best_x;
    int32_t* best_cycles;
    double* span_loss;
};

__global__ void __launch_bounds__(kWave) span_merge_kernel(SpanMergeArgs a) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const bool live = i < a.count;
    const int t = a.first + (live ? i : 0);
    double best = INFINITY;
    int cyc = -1;
    bool done = false;
    if (live)
        for (int j = 0; j < kSpanLossStride; ++j) a.span_loss[(int64_t)t * kSpanLossStride + j] = NAN;  // "span not run"
    for (int k = 1; k <= 3; ++k) {
        const bool in = k >= a.k_min && k <= a.k_max;
        unsigned long long e_all = 0, e_acc = 0, e_pre = 0;
        int act = 0;
        if (live && in) {
            const unsigned long long e0 = a.ev[k][(int64_t)i * 3 + 0];
            e_all = e0;
            if (!done) {
                const double sl = a.loss[k][i];
                if (cyc < 0 || sl < best) {
                    best = sl;
                    cyc = k;
                }
                a.span_loss[(int64_t)t * kSpanLossStride + (k - 1)] = best;
                e_acc = a.ev[k][(int64_t)i * 3 + 1];
                e_pre = a.ev[k][(int64_t)i * 3 + 2];
                act = 1;
                done = best < a.threshold;
            } else {
                e_pre = e0;  // the span loop had already stopped: all of this stage was speculation
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            e_all += __shfl_down(e_all, off);
            e_acc += __shfl_down(e_acc, off);
            e_pre += __shfl_down(e_pre, off);
            act += __shfl_down(act, off);
        }
        if (threadIdx.x == 0 && in) {
            StageCtl* c = a.ctl + k;
            atomicAdd(&c->evals, e_all);
            atomicAdd(&c->evals_accepted, e_acc);
            atomicAdd(&c->evals_preempted, e_pre);
            atomicAdd(&c->n_active, act);
            if (blockIdx.x == 0 && a.src_ctl[k]) atomicAdd(&c->rounds, a.src_ctl[k]->rounds);
        }
    }
    if (!live) return;
    a.best_loss[t] = best;
    a.best_cycles[t] = cyc;
    const int nv = cyc >= 1 ? a.xn[cyc] : 0;
    const double* src = cyc >= 1 ? a.x[cyc] + (int64_t)i * a.xstride[cyc] : nullptr;
    for (int j = 0; j < a.nmax; ++j) a.best_x[(int64_t)t * a.nmax + j] = j < nv ? src[j] : 0.0;
}

// ---------------------------------------------------------------------------------
// per-target reduction over restarts: argmin of item_loss (ties -> lowest restart), then the span loop's
// bookkeeping (TemplateOptimizer._run, optimizer.py:281-303): "if best_result is None or result.fun <
// best_result" the stage result replaces the target's best (loss, parameters, cycles).
// ---------------------------------------------------------------------------------
struct ReduceArgs {
    const ItemRec* item_rec;   // [n_active * R]
    const double* item_x;      // [n_active * R][n]
    double exit_loss;          // ordered == 1: the winner is the lowest-index restart below exit_loss (else the argmin)
    int32_t ordered;
    StageCtl* ctl;             // n_active; evals += sum of item_evals
    int32_t restarts;
    int32_t n;                 // parameters at this span
    double* stage_loss;        // [n_active]
    double* stage_x;           // [n_active][n]
    int32_t* stage_restart;    // [n_active]
    // merge into the resident results (best_loss == nullptr: single-stage call, no merge)
    const int32_t* active;     // [n_active] target index of each stage slot (nullptr = identity)
    int32_t nmax;
    int32_t k;
    double* best_loss;         // [n_targets]
    double* best_x;            // [n_targets][nmax]
    int32_t* best_cycles;      // [n_targets]
    double* span_loss;         // [n_targets][kSpanLossStride]: running best after span k at [k - 1] ("Cycle (k =...), Best Loss")
    // overlapped spans: the slot's evaluation counts (all, accepted, pre-empted) go here instead of into ctl -- the merge books them
    unsigned long long* slot_ev;  // [n_active][3] or nullptr
};

struct EvalCounts {
    unsigned long long all = 0, accepted = 0, preempted = 0;
};

__device__ __forceinline__ void reduce_merge_slot(const ReduceArgs& a, int64_t s, EvalCounts& ev) {
    {
        double best = INFINITY;
        int br = 0;
        bool hit = false;  // ordered mode: a restart below exit_loss has been seen (the lowest index wins)
        // (unrolled: four restarts' loads in flight -- the trip count is a run-time value, and one thread walks several
        // targets one after the other: the single-workgroup epilogue of a small batch is a chain of memory latencies)
#pragma unroll 4
        for (int r = 0; r < a.restarts; ++r) {
            // one 32-byte record per restart: two 16-byte loads
            const ulonglong2* rw = reinterpret_cast<const ulonglong2*>(a.item_rec + (s * a.restarts + r));
            const ulonglong2 w0 = rw[0], w1 = rw[1];
            const double l = __longlong_as_double((long long)w0.x);
            const bool pre = (int)(w0.y >> 32) == ST_PREEMPTED;
            const unsigned long long e = w1.x & 0xffffffffull;
            const unsigned long long ac = w1.x >> 32;
            ev.all += e;
            ev.preempted += pre ? e : 0ull;
            ev.accepted += pre ? 0ull : ac;
            // branch-free (the loads of the next restarts must not wait for this one's verdict):
            //   not hit yet: an ordered-mode restart below exit_loss takes the stage and closes it; else the lower loss takes
            const bool below = a.ordered && l < a.exit_loss;
            const bool take = !hit && (below || l < best);   // NaN / +inf (pre-empted) never win
            best = take ? l : best;
            br = take ? r : br;
            hit = hit || below;
        }
        a.stage_loss[s] = best;
        a.stage_restart[s] = br;
        const double* src = a.item_x + (s * a.restarts + br) * a.n;
        // the winner's parameters are loaded six at a time (n = 6 (k + 1), plus the gate parameters of a V2 template: tail loop)
        int i6 = 0;
        for (; i6 + 6 <= a.n; i6 += 6) {
            double v[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = src[i6 + j];
#pragma unroll
            for (int j = 0; j < 6; ++j) a.stage_x[s * a.n + i6 + j] = v[j];
        }
        for (int i = i6; i < a.n; ++i) a.stage_x[s * a.n + i] = src[i];
        if (a.best_loss) {
            const int64_t t = a.active ? a.active[s] : s;
            if (a.best_cycles[t] < 0 || best < a.best_loss[t]) {
                a.best_loss[t] = best;
                a.best_cycles[t] = a.k;
                int j6 = 0;
                for (; j6 + 6 <= a.n; j6 += 6) {
                    double v[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) v[j] = src[j6 + j];
#pragma unroll
                    for (int j = 0; j < 6; ++j) a.best_x[t * a.nmax + j6 + j] = v[j];
                }
                for (int i = j6; i < a.n; ++i) a.best_x[t * a.nmax + i] = src[i];
                for (int i = a.n; i < a.nmax; ++i) a.best_x[t * a.nmax + i] = 0.0;  // defined rows: nothing beyond 6 (best_cycles + 1)
            }
            if (a.span_loss) a.span_loss[t * kSpanLossStride + (a.k - 1)] = a.best_loss[t];
        }
    }
}

// wave-level sums of the evaluation counters, one atomic each per wave
__device__ __forceinline__ void publish_eval_counts(StageCtl* ctl, EvalCounts ev, int tid) {
    for (int off = 32; off > 0; off >>= 1) {
        ev.all += __shfl_down(ev.all, off);
        ev.accepted += __shfl_down(ev.accepted, off);
        ev.preempted += __shfl_down(ev.preempted, off);
    }
    if ((tid & 63) == 0 && ev.all) {
        atomicAdd(&ctl->evals, ev.all);
        atomicAdd(&ctl->evals_accepted, ev.accepted);
        atomicAdd(&ctl->evals_preempted, ev.preempted);
    }
}

__global__ void reduce_merge_kernel(ReduceArgs a) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    EvalCounts ev;
    const bool live = s < a.ctl->n_active;
    if (live) reduce_merge_slot(a, s, ev);
    if (a.slot_ev) {
        if (live) {
            a.slot_ev[s * 3 + 0] = ev.all;
            a.slot_ev[s * 3 + 1] = ev.accepted;
            a.slot_ev[s * 3 + 2] = ev.preempted;
        }
        return;
    }
    publish_eval_counts(a.ctl, ev, threadIdx.x);
}

// stage inputs: clear the early-exit flags and gather the active targets into a dense array so that the
// optimizer kernel addresses a slot's target directly (no active[] -> targets[] dependent load on the
// refill path).  active == nullptr: the stage works on the resident array itself, nothing to gather.
__global__ void stage_prepare_kernel(const double* targets, const int32_t* active, const StageCtl* ctl, double* out,
                                     int32_t* solved) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per double2
    const int64_t n_active = ctl->n_active;
    if (i < n_active) solved[i] = 0;
    if (active && i < n_active * 16) {
        const int64_t s = i >> 4;
        const int e = (int)(i & 15);
        reinterpret_cast<double2*>(out)[i] = reinterpret_cast<const double2*>(targets)[(int64_t)active[s] * 16 + e];
    }
}

// single-stage calls: the host knows the target count
__global__ void set_n_active_kernel(StageCtl* ctl, int32_t n) { ctl->n_active = n; }

// ---------------------------------------------------------------------------------
// slam_decompose_predicted: template sizes from the coverage lookup (span_predict_kernel), per-size target lists built on the device.
//   counts[k - 1]   targets whose size is k (1 .. k_max);  counts[k_max] local targets (size 0), counts[k_max + 1] out of reach
//   lists[(k - 1) * n + i]  the i-th target of size k (resident index; order = arrival order of the wavefronts -- a target's result
//                           does not depend on its place in a list)
// Also resets the window's results: (+inf, -1) -- (0, 0) for a local target, which needs no gate.
// ---------------------------------------------------------------------------------
__global__ void span_bucket_kernel(const int32_t* __restrict__ spans, int64_t first, int64_t n, int32_t k_max, int32_t* __restrict__ lists,
                                   int32_t* __restrict__ counts, double* best_loss, int32_t* best_cycles, double* span_loss) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = t < n;
    const int32_t s = in ? spans[t] : -1;
    if (in) {
        const int64_t tgt = first + t;
        best_loss[tgt] = s == 0 ? 0.0 : (double)INFINITY;
        best_cycles[tgt] = s == 0 ? 0 : -1;
        for (int j = 0; j < kSpanLossStride; ++j) span_loss[tgt * kSpanLossStride + j] = NAN;
    }
    const int32_t slot = !in ? -1 : (s == 0 ? k_max : (s > k_max ? k_max + 1 : s - 1));
    const unsigned long long lt = (1ull << (threadIdx.x & 63)) - 1ull;
    for (int32_t b = 0; b <= k_max + 1; ++b) {  // one atomic per wavefront and class
        const unsigned long long m = __ballot(slot == b);
        if (m == 0ull) continue;
        int32_t base = 0;
        if ((threadIdx.x & 63) == __builtin_ctzll(m)) base = atomicAdd(&counts[b], __popcll(m));
        base = __shfl(base, __builtin_ctzll(m));
        if (slot == b && b < k_max) lists[(int64_t)b * n + base + __popcll(m & lt)] = (int32_t)(first + t);
    }
}

// the targets of size k join the stage's active list behind the ones carried over from the previous stage (single workgroup: the
// base offset is read before anybody moves it)
__global__ void stage_append_kernel(const int32_t* __restrict__ bucket, const int32_t* __restrict__ count, StageCtl* ctl, int32_t* active) {
    __shared__ int32_t base_s;
    if (threadIdx.x == 0) base_s = ctl->n_active;
    __syncthreads();
    const int32_t base = base_s, n = *count;
    for (int32_t i = threadIdx.x; i < n; i += blockDim.x) active[base + i] = bucket[i];
    __syncthreads();
    if (threadIdx.x == 0) ctl->n_active = base + n;
}
__global__ void clear_ctl_kernel(StageCtl* ctl_all, int32_t n_words, int32_t* counts, int32_t n_counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) reinterpret_cast<unsigned long long*>(ctl_all)[i] = 0ull;
    if (i < n_counts) counts[i] = 0;
}

// reset the results of targets [first, first + n), (optionally) write their indices as the initial
// active list, and publish n as the first stage's target count
// ... and prepare the first stage's inputs (early-exit flags; the window's targets as a dense array when
// the batch is a window of the resident targets).  One thread per double2 of the window's targets.
// list_mode: the batch is an explicit list of target indices already in `active` (slam_decompose_list)
// instead of the window [first, first + n).
__global__ void init_results_kernel(double* best_loss, int32_t* best_cycles, double* span_loss, int32_t* active, int64_t first,
                                    int64_t n, StageCtl* first_stage, const double* targets, double* stage_targets,
                                    int32_t* solved, StageCtl* ctl_all, int32_t n_ctl_words, int32_t list_mode) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        // one thread clears all stages' control blocks, then publishes the first stage's target count
        for (int w = 0; w < n_ctl_words; ++w) reinterpret_cast<unsigned long long*>(ctl_all)[w] = 0ull;
        first_stage->n_active = (int32_t)n;
    }
    if (t < n) {
        const int64_t tgt = list_mode ? (int64_t)active[t] : first + t;
        best_loss[tgt] = INFINITY;
        best_cycles[tgt] = -1;
        for (int j = 0; j < kSpanLossStride; ++j) span_loss[tgt * kSpanLossStride + j] = NAN;
        solved[t] = 0;
        if (active && !list_mode) active[t] = (int32_t)tgt;
    }
    if (active && t < n * 16) {
        const int64_t src = list_mode ? (int64_t)active[t >> 4] * 16 + (t & 15) : first * 16 + t;
        reinterpret_cast<double2*>(stage_targets)[t] = reinterpret_cast<const double2*>(targets)[src];
    }
}

// freshly (re)allocated resident results: "nothing found yet" for every target, so that windows no call has
// decomposed read as best_loss = +inf, best_cycles = -1 instead of uninitialised memory
__global__ void fill_results_kernel(double* best_loss, int32_t* best_cycles, double* span_loss, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        best_loss[t] = INFINITY;
        best_cycles[t] = -1;
        for (int j = 0; j < kSpanLossStride; ++j) span_loss[t * kSpanLossStride + j] = NAN;
    }
}

// Ordered compaction of the targets that still need a longer template:
// keep t iff !(best_loss[t] < threshold)   (optimizer.py:301: break when best < threshold).
// Single workgroup, chunked scan: n is at most a few million.
template <int NT>
__device__ __forceinline__ int32_t compact_block(const int32_t* active_in, int64_t n_in, const double* best_loss,
                                                 double threshold, int32_t* active_out, int32_t* counts, int32_t* offs) {
    const int tid = threadIdx.x;
    const int64_t chunk = (n_in + NT - 1) / NT;
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
        for (int i = 0; i < NT; ++i) { offs[i] = acc; acc += counts[i]; }
        offs[NT] = acc;
    }
    __syncthreads();
    int32_t o = offs[tid];
    for (int64_t s = lo; s < hi; ++s) {
        const int32_t t = active_in ? active_in[s] : (int32_t)s;
        if (!(best_loss[t] < threshold)) active_out[o++] = t;
    }
    return offs[NT];
}

// Small batches (at most kEpilogueMaxTargets targets): everything between two optimizer launches in ONE
// single-workgroup kernel -- reduction over restarts, span-loop bookkeeping, compaction of the unsolved
// targets and the next stage's inputs (gathered targets, cleared early-exit flags).  With several batches in
// flight every extra kernel of the chain waits for wavefront slots held by other batches' persistent
// optimizer waves (hundreds of microseconds each under load), so the chain is kept as short as possible.
constexpr int64_t kEpilogueMaxTargets = 8192;

struct EpilogueArgs {
    ReduceArgs r;
    int32_t has_next;
    double threshold;
    int32_t* active_out;     // [n_upper] next stage's active list
    StageCtl* next;
    const double* targets;   // resident targets
    double* stage_targets;   // next stage's gathered targets
    int32_t* solved;
};

// NT = 256 for the smallest batches: a 4-wave workgroup finds room on a busy GPU much sooner than a 16-wave one.
template <int NT>
__global__ void __launch_bounds__(NT) stage_epilogue_kernel(EpilogueArgs a) {
    __shared__ int32_t counts[NT];
    __shared__ int32_t offs[NT + 1];
    const int tid = threadIdx.x;
    const int64_t n_in = a.r.ctl->n_active;
    EvalCounts ev;
    for (int64_t s = tid; s < n_in; s += NT) reduce_merge_slot(a.r, s, ev);
    publish_eval_counts(a.r.ctl, ev, tid);
    if (!a.has_next) return;
    __syncthreads();  // this workgroup wrote every best_loss the compaction reads
    const int32_t total = compact_block<NT>(a.r.active, n_in, a.r.best_loss, a.threshold, a.active_out, counts, offs);
    __syncthreads();  // active_out complete
    if (tid == 0) a.next->n_active = total;
    for (int64_t i = tid; i < (int64_t)total * 16; i += NT)
        reinterpret_cast<double2*>(a.stage_targets)[i] =
            reinterpret_cast<const double2*>(a.targets)[(int64_t)a.active_out[i >> 4] * 16 + (i & 15)];
    for (int64_t i = tid; i < total; i += NT) a.solved[i] = 0;
}

// Big batches: the same single launch per stage, as a GRID of 256-thread workgroups -- reduction over restarts + span-loop
// bookkeeping per target, then the compaction of the targets that still need a longer template by one atomicAdd per
// workgroup on the next stage's target count (ordered inside a workgroup, workgroups in arrival order: the active list's
// order only decides which wavefront works on what, never a result), then the next stage's inputs for the kept targets.
// Round 2 ran three kernels here (reduce_merge, a single-workgroup compact_active, stage_prepare): with several batches in
// flight each of them waited its turn behind other batches' persistent wavefronts.
__device__ __forceinline__ void stage_epilogue_grid_body(const EpilogueArgs& a) {
    __shared__ int32_t wave_counts[4];
    __shared__ int32_t s_base;
    __shared__ int32_t kept_t[256];  // the workgroup's kept targets, in order
    const int tid = threadIdx.x;
    const int64_t n_in = a.r.ctl->n_active;
    const int64_t s = (int64_t)blockIdx.x * 256 + tid;
    EvalCounts ev;
    bool keep = false;
    int32_t t = -1;
    if (s < n_in) {
        reduce_merge_slot(a.r, s, ev);
        t = a.r.active ? a.r.active[s] : (int32_t)s;
        keep = a.has_next && !(a.r.best_loss[t] < a.threshold);  // optimizer.py:301: break when best < threshold
    }
    publish_eval_counts(a.r.ctl, ev, tid);
    if (!a.has_next) return;
    const unsigned long long m = __ballot(keep);
    const int lane = tid & 63, w = tid >> 6;
    if (lane == 0) wave_counts[w] = __popcll(m);
    __syncthreads();
    const int32_t tot = wave_counts[0] + wave_counts[1] + wave_counts[2] + wave_counts[3];
    if (tid == 0) s_base = tot ? atomicAdd(&a.next->n_active, tot) : 0;
    if (keep) {
        int32_t r = __popcll(m & ((1ull << lane) - 1ull));
        for (int i = 0; i < w; ++i) r += wave_counts[i];
        kept_t[r] = t;
    }
    __syncthreads();
    const int32_t base = s_base;
    if (tid < tot) {
        a.active_out[base + tid] = kept_t[tid];
        a.solved[base + tid] = 0;
    }
    // the kept targets' matrices, copied by the whole workgroup: 16 double2 per target, destination contiguous
    const double2* src = reinterpret_cast<const double2*>(a.targets);
    double2* dst = reinterpret_cast<double2*>(a.stage_targets) + (int64_t)base * 16;
    for (int e = tid; e < tot * 16; e += 256) dst[e] = src[(int64_t)kept_t[e >> 4] * 16 + (e & 15)];
}
__global__ void __launch_bounds__(256) stage_epilogue_grid_kernel(EpilogueArgs a) { stage_epilogue_grid_body(a); }
// slam_decompose_multi: the same for n_sub sub-problems in one launch (blockIdx.y = sub-problem, its arguments in device memory)
__global__ void __launch_bounds__(256) stage_epilogue_multi_kernel(const EpilogueArgs* arr) {
    const EpilogueArgs a = arr[blockIdx.y];
    stage_epilogue_grid_body(a);
}

}  // namespace slamdev
