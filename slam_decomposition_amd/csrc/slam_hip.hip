// slam_hip.hip -- host side of libslamhip.so: the C ABI declared in include/slam_hip.h.
// gfx950 only; no torch, no CUDA compatibility layer.
#include "../../include/slam_hip.h"
#include "slam_kernels.hpp"
#include "slam_sampler.hpp"
#include "slam_weyl.hpp"
#include "slam_v2.hpp"
#include "slam_long.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <algorithm>
#include <vector>

using namespace slamdev;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace

// slam_comm.hip reports its errors through the same thread-local message (library-internal, not exported)
extern "C" __attribute__((visibility("hidden"))) void slam_set_last_error(const char* msg) { g_err = msg ? msg : ""; }

namespace {

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(SLAM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// growable device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return e;
        }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            want = bytes;
            e = hipMalloc(&p, want);
        }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() { return static_cast<T*>(p); }
};

}  // namespace

struct slam_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_a[SLAM_MAX_SPAN_EVAL + 1] = {}, ev_b[SLAM_MAX_SPAN_EVAL + 1] = {};  // optimizer-kernel bracket per span
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr; // whole-call bracket
    // The host waits for a finished span loop on a blocking-sync event: the waiting thread sleeps instead of
    // spinning, so that many contexts (one host thread each) can be in flight without the threads fighting
    // over cores.  (Measured: with spinning waits, 32 batches in flight run 30 % slower than 16.)
    hipEvent_t ev_done = nullptr;
    StageCtl* h_ctl = nullptr;                   // pinned: the stages' control blocks, copied back once per call
    double* h_gates = nullptr;                   // pinned mirror of span_gates
    void* h_stage = nullptr;                     // pinned staging for result fetches (same reason: no spinning
    size_t h_stage_cap = 0;                      // inside the runtime's pageable-copy path)
    int64_t n_targets = 0;
    int32_t n_gates = 0;
    DevBuf targets, gates;
    // stage work buffers
    DevBuf active, active2, x0;
    DevBuf item_rec, item_x;  // per work item: one 32-byte result record (slam_kernels.hpp: ItemRec), the parameter row
    DevBuf stage_loss, stage_x, stage_restart;
    // decompose results
    DevBuf best_loss, best_x, best_cycles, span_loss;
    DevBuf v2_hmem;              // inverse Hessians of the long parametrised-gate templates (slam_v2.hpp: v2_h_in_memory)
    DevBuf v2_maps, v2_bounds;   // slam_v2_*: staged gate maps [SLAM_MAX_SPAN_EVAL], (init_lo, init_hi, bound_lo, bound_hi)[n]
    std::vector<V2GateMap> v2_gates_host;
    int v2_qn = 0;
    // cost constraint per span (slam_v2_set_constraint): weights [n(k)] on the device, right-hand side; empty = none
    DevBuf v2_cons_w[SLAM_V2_MAX_SPAN + 1];
    int v2_cons_n[SLAM_V2_MAX_SPAN + 1] = {0};
    double v2_cons_max[SLAM_V2_MAX_SPAN + 1] = {0};
    double v2_cons_rho[SLAM_V2_MAX_SPAN + 1] = {0};  // penalty parameter of the multiplier method: 30 / max_i w_i^2
    DevBuf trace_loss, trace_x;  // slam_minimize_stage_trace
    int32_t trace_cap = 0;       // > 0 only inside slam_minimize_stage_trace
    double stage_exit_loss = -1.0;  // single-stage calls: >= 0 overrides stop_loss as the ordered early-exit level
    int32_t result_nmax = 0;
    int64_t result_filled = 0;  // targets whose resident results have been initialised (+inf / -1) for result_nmax
    DevBuf counters;  // StageCtl[SLAM_MAX_SPAN_EVAL + 2]: one control block per span stage (slam_kernels.hpp)
    DevBuf long_hmem;  // inverse Hessian approximations of the wavefront-per-item kernels: [resident wavefronts][n][128] floats
    DevBuf bucket_lists, bucket_counts;  // slam_decompose_predicted: per-size target lists [k_max][count], their sizes
    int32_t* h_bucket_counts = nullptr;  // pinned mirror of bucket_counts
    DevBuf solved;
    DevBuf stage_targets;
    DevBuf span_gates;  // 64 slots x [SLAM_MAX_SPAN_EVAL][32] doubles
    struct StagedSeq { bool valid = false; int32_t seq[SLAM_MAX_SPAN_EVAL] = {}; } staged[SLAM_MAX_SPAN_EVAL + 1];
    int cost_kind = 0;  // SLAM_COST_*
    std::vector<double> gates_host;
    int compute_units = 0;
    int reserve_waves = 0;  // wavefront slots the persistent optimizer grid leaves free for the span loop's bookkeeping kernels
    int64_t resident_waves[SLAM_MAX_SPAN_EVAL + 1][kGateClasses] = {};
    // eval buffers
    DevBuf ev_x, ev_tof, ev_loss, ev_grad, ev_unitary, ev_weyl;
    slam_stats stats{};
    bool max_lds_set[SLAM_MAX_SPAN_EVAL + 1][kGateClasses][3] = {};  // [.][.][0] eval kernel, [1] optimizer kernel, [2] its multi-queue form
    int64_t resident_waves_mq[SLAM_MAX_SPAN_EVAL + 1][kGateClasses] = {};
    int64_t resident_waves_long = 0;  // wavefront-per-item kernels (slam_long.hpp): resident wavefronts; 0 = not asked yet
    bool long_eval_ready = false;
    int64_t resident_waves_wl[kGateClasses] = {};  // span_wave_kernel<GC>: resident wavefronts (0 = not asked yet)
    // speculative spans (span_spec_kernel): staging rows, two side streams, fork / join events
    DevBuf spec_loss, spec_x, spec_ev;
    hipStream_t spec_stream[2] = {nullptr, nullptr};
    hipEvent_t spec_fork = nullptr, spec_join[2] = {nullptr, nullptr};
    bool spec_attr_set[4][kGateClasses] = {};
    // overlapped spans (decompose_overlapped): one helper context per span (own stream, own stage buffers; targets borrowed)
    slam_ctx* helper[SLAM_MAX_SPAN_EVAL + 1] = {};
    hipEvent_t ov_fork = nullptr, ov_join[SLAM_MAX_SPAN_EVAL + 1] = {};
    DevBuf slot_ev;                 // (helper side) per-slot evaluation counts of its stage
    bool slot_ev_on = false;        // (helper side) single-stage reductions write slot_ev instead of the stage's counters
    uint64_t gates_version = 1;     // bumped by slam_set_gates
    uint64_t helper_gates_version = 0;  // (helper side) the owner's gates_version its gate table is a copy of
    // slam_decompose_multi (this context leads the call): the sub-problems' argument blocks / epilogue arguments per span, staged
    // through pinned memory
    DevBuf mq_args;
    void* h_mq_args = nullptr;
    size_t h_mq_cap = 0;
    int v2_per_cu[SLAM_V2_MAX_SPAN + 1][3][2][2] = {};  // resident workgroups per CU of minimize_v2_kernel<K, QN, GQ, FREE> (0 = not asked yet)

    ~slam_ctx() {
        DevBuf* all[] = {&targets, &gates, &active, &active2, &x0, &item_rec, &item_x, &stage_loss, &stage_x, &stage_restart, &best_loss,
                         &best_x, &best_cycles, &span_loss, &trace_loss, &trace_x, &v2_maps, &v2_bounds, &v2_hmem, &long_hmem, &bucket_lists, &bucket_counts, &v2_cons_w[0], &v2_cons_w[1], &v2_cons_w[2], &v2_cons_w[3], &v2_cons_w[4], &v2_cons_w[5], &counters, &solved, &stage_targets, &span_gates, &ev_x, &ev_tof, &ev_loss, &ev_grad, &ev_unitary, &ev_weyl};
        for (DevBuf* b : all) b->release();
        for (hipEvent_t e : ev_a) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_b) if (e) (void)hipEventDestroy(e);
        if (ev_t0) (void)hipEventDestroy(ev_t0);
        if (ev_t1) (void)hipEventDestroy(ev_t1);
        if (ev_done) (void)hipEventDestroy(ev_done);
        if (h_ctl) (void)hipHostFree(h_ctl);
        if (h_stage) (void)hipHostFree(h_stage);
        if (h_gates) (void)hipHostFree(h_gates);
        if (h_bucket_counts) (void)hipHostFree(h_bucket_counts);
        if (h_mq_args) (void)hipHostFree(h_mq_args);
        mq_args.release();
        spec_loss.release();
        spec_x.release();
        spec_ev.release();
        slot_ev.release();
        if (ov_fork) (void)hipEventDestroy(ov_fork);
        for (hipEvent_t e : ov_join) if (e) (void)hipEventDestroy(e);
        for (slam_ctx* h : helper) {
            if (!h) continue;
            h->targets.p = nullptr;  // borrowed from this context
            h->targets.cap = 0;
            delete h;
        }
        if (spec_fork) (void)hipEventDestroy(spec_fork);
        for (int j = 0; j < 2; ++j) {
            if (spec_join[j]) (void)hipEventDestroy(spec_join[j]);
            if (spec_stream[j]) (void)hipStreamDestroy(spec_stream[j]);
        }
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

// Every API call leaves the context's stream drained: the per-span gate slots, the pinned staging buffers and
// DevBuf::reserve's hipFree rely on it.  A call that fails after it has enqueued work therefore waits for that work
// (ignoring the result: the error being reported is the first one) and forgets the cached gate slots.
int drained(slam_ctx* c, int rc) {
    if (rc != SLAM_OK && c && c->stream) {
        const std::string keep = g_err;
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        for (hipStream_t st : c->spec_stream)
            if (st) (void)hipStreamSynchronize(st);
        for (slam_ctx* h : c->helper)
            if (h && h->stream) (void)hipStreamSynchronize(h->stream);
        (void)hipGetLastError();
        for (auto& st : c->staged) st.valid = false;
        g_err = keep;
    }
    return rc;
}

template <int K, int GC>
constexpr size_t lds_bytes() { return sizeof(double) * lds_doubles<K, GC>(); }

// copy G_1..G_K of this span, in order, into the context's small device buffer (stream-ordered)
int stage_gates(slam_ctx* c, int k, const int32_t* gate_seq, const double** d_out) {
    // One device slot per span k, cached: as long as the gate table and the span's sequence do not change (the
    // normal case: one basis, many batches) nothing is copied -- a copy inside the span loop's chain of kernels
    // costs as much as a kernel there (it waits its turn on a busy GPU).  Every API call ends with the stream
    // drained, so a slot is never rewritten under a running kernel.  The host side of a slot is pinned: a copy
    // from pageable memory would make the calling thread wait (spinning) for everything enqueued before it.
    double* dst = c->span_gates.as<double>() + (size_t)k * SLAM_MAX_SPAN_EVAL * 32;
    *d_out = dst;
    slam_ctx::StagedSeq& st = c->staged[k];
    if (st.valid && std::memcmp(st.seq, gate_seq, (size_t)k * sizeof(int32_t)) == 0) return SLAM_OK;
    double* tmp = c->h_gates + (size_t)k * SLAM_MAX_SPAN_EVAL * 32;
    for (int j = 0; j < k; ++j) std::memcpy(tmp + 32 * j, c->gates_host.data() + (size_t)gate_seq[j] * 32, 32 * sizeof(double));
    HIP_TRY(hipMemcpyAsync(dst, tmp, (size_t)k * 32 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    std::memcpy(st.seq, gate_seq, (size_t)k * sizeof(int32_t));
    st.valid = true;
    return SLAM_OK;
}

// Structure class of the gates of one span (see slam_device.hpp: GC_*).  A launch uses the most
// general class any of its gates needs; entries below 1e-15 count as structural zeros.
int classify_gates(slam_ctx* c, int k, const int32_t* gate_seq) {
    const double tol = 1e-15;
    bool all_cx = true, all_x = true, all_xri = true, all_xri1 = true;
    for (int j = 0; j < k; ++j) {
        const double* g = c->gates_host.data() + (size_t)gate_seq[j] * 32;
        auto re = [&](int r, int s) { return g[(r * 4 + s) * 2]; };
        auto im = [&](int r, int s) { return g[(r * 4 + s) * 2 + 1]; };
        auto mag = [&](int r, int s) { return std::fabs(re(r, s)) + std::fabs(im(r, s)); };
        // qiskit CXGate: ones at (0,0), (1,3), (2,2), (3,1)
        bool cx = true;
        for (int r = 0; r < 4; ++r)
            for (int s = 0; s < 4; ++s) {
                const bool one = (r == 0 && s == 0) || (r == 1 && s == 3) || (r == 2 && s == 2) || (r == 3 && s == 1);
                if (std::fabs(re(r, s) - (one ? 1.0 : 0.0)) > tol || std::fabs(im(r, s)) > tol) cx = false;
            }
        all_cx = all_cx && cx;
        // X shape: non-zeros only inside the blocks on index pairs (0,3) and (1,2)
        bool x = true;
        for (int r = 0; r < 4; ++r)
            for (int s = 0; s < 4; ++s) {
                const bool inside = ((r == 0 || r == 3) && (s == 0 || s == 3)) || ((r == 1 || r == 2) && (s == 1 || s == 2));
                if (!inside && mag(r, s) > tol) x = false;
            }
        all_x = all_x && x;
        const bool xri = x && std::fabs(im(0, 0)) <= tol && std::fabs(im(3, 3)) <= tol && std::fabs(im(1, 1)) <= tol &&
                         std::fabs(im(2, 2)) <= tol && std::fabs(re(0, 3)) <= tol && std::fabs(re(3, 0)) <= tol &&
                         std::fabs(re(1, 2)) <= tol && std::fabs(re(2, 1)) <= tol;
        all_xri = all_xri && xri;
        // ... and the identity on the (0,3) block: RiSwapGate(alpha) (sqrt-iSWAP, iSWAP) leaves |00> and |11> alone
        all_xri1 = all_xri1 && xri && std::fabs(re(0, 0) - 1.0) <= tol && std::fabs(re(3, 3) - 1.0) <= tol && mag(0, 3) <= tol &&
                   mag(3, 0) <= tol;
    }
    if (all_cx) return GC_CX;
    if (all_xri1) return GC_XRI1;
    if (all_xri) return GC_XRI;
    if (all_x) return GC_XGEN;
    return GC_DENSE;
}

template <int K, int GC>
int launch_eval(slam_ctx* c, const int32_t* gate_seq, const double* d_x, const int32_t* d_tof, int64_t M,
                double* d_loss, double* d_grad, double* d_unitary) {
    const size_t lds = lds_bytes<K, GC>();
    if (!c->max_lds_set[K][GC][0]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&eval_kernel<K, GC>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->max_lds_set[K][GC][0] = true;
    }
    EvalArgs<K> a{};
    a.targets = c->targets.as<double>();
    a.x = d_x;
    a.target_of = d_tof;
    a.n_items = M;
    a.loss = d_loss;
    a.grad = d_grad;
    a.unitary = d_unitary;
    a.cost_kind = c->cost_kind;
    { int rc = stage_gates(c, K, gate_seq, &a.gates); if (rc) return rc; }
    const int64_t blocks = (M + kQuadsPerWave - 1) / kQuadsPerWave;
    hipLaunchKernelGGL((eval_kernel<K, GC>), dim3((unsigned)blocks), dim3(kWave), lds, c->stream, a);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

struct StageLaunch {
    double exit_loss;               // a finished restart below this pre-empts its siblings
    const int32_t* gate_seq;
    const double* d_stage_targets;  // [n_active][32]
    const int32_t* d_active;        // original target index per slot, or nullptr = first_target + slot
    int32_t first_target;
    const double* d_x0;
    int64_t n_items_max;            // upper bound (grid sizing); the kernel reads the real count from ctl
    const slam_opt_params* prm;
    StageCtl* ctl;
};

// the optimizer kernel's argument block for one stage of one context
template <int K>
int build_minimize_args(slam_ctx* c, const StageLaunch& sl, MinimizeArgs<K>& a) {
    const slam_opt_params* prm = sl.prm;
    a = MinimizeArgs<K>{};
    a.targets = sl.d_stage_targets;
    a.orig = sl.d_active;
    a.first_target = sl.first_target;
    a.x0 = sl.d_x0;
    a.ctl = sl.ctl;
    a.restarts = prm->restarts;
    a.maxiter = prm->maxiter;
    a.gtol = prm->gtol;
    a.stop_loss = prm->stop_loss;
    a.gtol_far = prm->gtol_far;
    a.far_loss = prm->far_loss;
    a.exit_loss = sl.exit_loss;
    a.seed = prm->seed;
    a.target_base = prm->target_base;
    a.flags = prm->flags & (SLAM_FLAG_EARLY_EXIT | SLAM_FLAG_ORDERED);  // (the upper bits are internal: kFlagTrace, kFlagNoExterior)
    if (prm->flags & SLAM_FLAG_NO_EXTERIOR) a.flags |= kFlagNoExterior;
    a.items_per_quad = prm->items_per_quad;
    a.cost_kind = c->cost_kind;
    a.solved = c->solved.as<int32_t>();
    a.item_rec = c->item_rec.as<ItemRec>();
    a.item_x = c->item_x.as<double>();
    a.trace_cap = c->trace_cap;
    a.trace_loss = c->trace_cap > 0 ? c->trace_loss.as<double>() : nullptr;
    if (a.trace_loss) a.flags |= kFlagTrace;
    a.trace_x = c->trace_cap > 0 ? c->trace_x.as<double>() : nullptr;
    { int rc = stage_gates(c, K, sl.gate_seq, &a.gates); if (rc) return rc; }
    return SLAM_OK;
}

template <int K, int GC, bool MQ = false>
int prepare_minimize_kernel(slam_ctx* c) {
    const size_t lds = lds_bytes<K, GC>();
    if (!c->max_lds_set[K][GC][MQ ? 2 : 1]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&minimize_kernel<K, GC, MQ>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&minimize_kernel<K, GC, MQ>),
                                                             kWave, lds));
        if (per_cu < 1) per_cu = 1;
        (MQ ? c->resident_waves_mq : c->resident_waves)[K][GC] = (int64_t)per_cu * c->compute_units;
        c->max_lds_set[K][GC][MQ ? 2 : 1] = true;
    }
    return SLAM_OK;
}

template <int K, int GC>
int launch_minimize(slam_ctx* c, const StageLaunch& sl) {
    const size_t lds = lds_bytes<K, GC>();
    { int rc = prepare_minimize_kernel<K, GC>(c); if (rc) return rc; }
    const slam_opt_params* prm = sl.prm;
    MinimizeArgs<K> a;
    { int rc = build_minimize_args<K>(c, sl, a); if (rc) return rc; }
    // persistent wavefronts: never more blocks than can be resident, every quad pulls items.  The grid is
    // sized for the upper bound of the item count; the kernel derives the real launch shape (waves that
    // take part, chunk size) from the device-side target count and surplus waves exit at once.
    int64_t blocks = (sl.n_items_max + kQuadsPerWave - 1) / kQuadsPerWave;
    if (prm->items_per_quad > 1) {
        blocks = (sl.n_items_max + (int64_t)kQuadsPerWave * prm->items_per_quad - 1) / ((int64_t)kQuadsPerWave * prm->items_per_quad);
        if (blocks < 1) blocks = 1;
    }
    // leave a few wavefront slots free: with several batches in flight the bookkeeping kernels of the other streams
    // (reduce / compaction / epilogue) otherwise wait for the tail of this stage before they can even start
    const int64_t cap = c->resident_waves[K][GC] - c->reserve_waves > 0 ? c->resident_waves[K][GC] - c->reserve_waves : 1;
    if (blocks > cap) blocks = cap;
    HIP_TRY(hipEventRecord(c->ev_a[K], c->stream));
    hipLaunchKernelGGL((minimize_kernel<K, GC>), dim3((unsigned)blocks), dim3(kWave), lds, c->stream, a, (const MinimizeArgs<K>*)nullptr, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_b[K], c->stream));
    return SLAM_OK;
}

// templates of SLAM_MAX_SPAN_QUAD + 1 .. SLAM_MAX_SPAN_MINIMIZE gates: one wavefront per item (slam_long.hpp)
int launch_minimize_long(slam_ctx* c, const StageLaunch& sl, int k) {
    if (c->resident_waves_long == 0) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&minimize_long_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLongLdsBytes));
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&minimize_long_kernel), kWave, kLongLdsBytes));
        c->resident_waves_long = (int64_t)(per_cu < 1 ? 1 : per_cu) * c->compute_units;
    }
    const slam_opt_params* prm = sl.prm;
    LongArgs a{};
    a.targets = sl.d_stage_targets;
    a.orig = sl.d_active;
    a.first_target = sl.first_target;
    a.x0 = sl.d_x0;
    a.ctl = sl.ctl;
    a.restarts = prm->restarts;
    a.maxiter = prm->maxiter;
    a.gtol = prm->gtol;
    a.stop_loss = prm->stop_loss;
    a.gtol_far = prm->gtol_far;
    a.far_loss = prm->far_loss;
    a.exit_loss = sl.exit_loss;
    a.seed = prm->seed;
    a.target_base = prm->target_base;
    a.flags = prm->flags & (SLAM_FLAG_EARLY_EXIT | SLAM_FLAG_ORDERED);
    if (prm->flags & SLAM_FLAG_NO_EXTERIOR) a.flags |= kFlagNoExterior;
    a.cost_kind = c->cost_kind;
    a.solved = c->solved.as<int32_t>();
    a.item_rec = c->item_rec.as<ItemRec>();
    a.item_x = c->item_x.as<double>();
    a.k = k;
    a.trace_cap = c->trace_cap;
    a.trace_loss = c->trace_cap > 0 ? c->trace_loss.as<double>() : nullptr;
    a.trace_x = c->trace_cap > 0 ? c->trace_x.as<double>() : nullptr;
    { int rc = stage_gates(c, k, sl.gate_seq, &a.gates); if (rc) return rc; }
    int64_t blocks = sl.n_items_max;  // one item per wavefront at a time
    if (blocks > c->resident_waves_long) blocks = c->resident_waves_long;
    if (blocks < 1) blocks = 1;
    // (sized for the whole grid and the longest template once: growing it between two stages of a chain would free it under the stage in flight)
    HIP_TRY(c->long_hmem.reserve((size_t)c->resident_waves_long * (size_t)(6 * (SLAM_MAX_SPAN_MINIMIZE + 1)) * kLongHStride * sizeof(float)));
    a.hmem = c->long_hmem.as<float>();
    HIP_TRY(hipEventRecord(c->ev_a[k], c->stream));
    hipLaunchKernelGGL(minimize_long_kernel, dim3((unsigned)blocks), dim3(kWave), kLongLdsBytes, c->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_b[k], c->stream));
    return SLAM_OK;
}

int launch_eval_long(slam_ctx* c, int k, const int32_t* gate_seq, const double* d_x, const int32_t* d_tof, int64_t M, double* d_loss, double* d_grad,
                     double* d_unitary) {
    if (!c->long_eval_ready) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&eval_long_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLongLdsBytes));
        c->long_eval_ready = true;
    }
    LongEvalArgs a{};
    a.targets = c->targets.as<double>();
    a.x = d_x;
    a.target_of = d_tof;
    a.n_items = M;
    a.loss = d_loss;
    a.grad = d_grad;
    a.unitary = d_unitary;
    a.cost_kind = c->cost_kind;
    a.k = k;
    { int rc = stage_gates(c, k, gate_seq, &a.gates); if (rc) return rc; }
    int64_t blocks = M;
    const int64_t cap = (int64_t)8 * c->compute_units;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(eval_long_kernel, dim3((unsigned)blocks), dim3(kWave), kLongLdsBytes, c->stream, a);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

int check_gate_seq(slam_ctx* c, int k, const int32_t* gate_seq) {
    if (!gate_seq) return fail(SLAM_ERR_INVALID, "gate_seq is NULL");
    for (int j = 0; j < k; ++j)
        if (gate_seq[j] < 0 || gate_seq[j] >= c->n_gates)
            return fail(SLAM_ERR_INVALID, "gate_seq[%d] = %d outside the gate table (n_gates = %d)", j,
                        gate_seq[j], c->n_gates);
    return SLAM_OK;
}

int check_params(const slam_opt_params* p) {
    if (!p) return fail(SLAM_ERR_INVALID, "params is NULL");
    if (p->restarts <= 0) return fail(SLAM_ERR_INVALID, "restarts must be > 0 (got %d)", p->restarts);
    if (p->maxiter < 0 || p->maxiter > SLAM_MAX_MAXITER)
        return fail(SLAM_ERR_INVALID, "maxiter must be in 0..%d (got %d)", SLAM_MAX_MAXITER, p->maxiter);
    if (p->target_base < 0 || p->target_base > 0x7fffffffLL) return fail(SLAM_ERR_INVALID, "target_base out of range");
    return SLAM_OK;
}

StageCtl* stage_ctl(slam_ctx* c, int k) { return c->counters.as<StageCtl>() + k; }

// Work buffers of a stage with at most n_upper targets at span <= k_max.
int reserve_stage_buffers(slam_ctx* c, int64_t n_upper, int k_max, const slam_opt_params* prm) {
    const int n = 6 * (k_max + 1);
    const int64_t M = n_upper * (int64_t)prm->restarts;
    if (M > 0x7fff0000LL) return fail(SLAM_ERR_INVALID, "too many work items in one stage (%lld)", (long long)M);
    HIP_TRY(c->item_rec.reserve(M * sizeof(ItemRec)));
    HIP_TRY(c->item_x.reserve(M * n * sizeof(double)));
    HIP_TRY(c->stage_loss.reserve(n_upper * sizeof(double)));
    HIP_TRY(c->stage_x.reserve(n_upper * n * sizeof(double)));
    HIP_TRY(c->stage_restart.reserve(n_upper * sizeof(int32_t)));
    HIP_TRY(c->solved.reserve(n_upper * sizeof(int32_t)));
    HIP_TRY(c->stage_targets.reserve((size_t)n_upper * 32 * sizeof(double)));
    return SLAM_OK;
}

// Enqueues one span stage for the device-resident active list (d_active may be nullptr = identity): stage
// inputs, the optimizer kernel, the per-target reduction and (merge) the span loop's bookkeeping.  No host
// synchronisation: the stage's target count lives in its control block (stage_ctl(c, k)->n_active, at most
// n_upper), which must have been zeroed and published on the stream before.  Leaves per-slot results in
// ctx->stage_loss / stage_x / stage_restart and the per-item arrays.
// What follows the optimizer kernel of a stage inside the span loop (nullptr: single-stage call, reduce only).
struct SpanLoopStep {
    double exit_loss;        // ordered early exit: the span loop's success threshold; otherwise params->stop_loss
    bool inputs_ready;       // the previous kernel of the chain already prepared this stage's inputs
    bool has_next;           // a longer span follows: compact the unsolved targets into active_out
    double threshold;
    int32_t* active_out;
};

ReduceArgs build_reduce_args(slam_ctx* c, int k, const int32_t* d_active, const slam_opt_params* prm, double exit_loss, bool merge) {
    ReduceArgs r{};
    r.slot_ev = (!merge && c->slot_ev_on) ? c->slot_ev.as<unsigned long long>() : nullptr;
    r.item_rec = c->item_rec.as<ItemRec>();
    r.item_x = c->item_x.as<double>();
    r.exit_loss = exit_loss;
    r.ordered = ((prm->flags & SLAM_FLAG_EARLY_EXIT) && (prm->flags & SLAM_FLAG_ORDERED)) ? 1 : 0;
    r.ctl = stage_ctl(c, k);
    r.restarts = prm->restarts;
    r.n = 6 * (k + 1);
    r.stage_loss = c->stage_loss.as<double>();
    r.stage_x = c->stage_x.as<double>();
    r.stage_restart = c->stage_restart.as<int32_t>();
    if (merge) {
        r.active = d_active;
        r.nmax = c->result_nmax;
        r.k = k;
        r.best_loss = c->best_loss.as<double>();
        r.best_x = c->best_x.as<double>();
        r.best_cycles = c->best_cycles.as<int32_t>();
        r.span_loss = c->span_loss.as<double>();
    }
    return r;
}

EpilogueArgs build_epilogue_args(slam_ctx* c, int k, const ReduceArgs& r, const SpanLoopStep& loop) {
    EpilogueArgs e{};
    e.r = r;
    e.has_next = loop.has_next ? 1 : 0;
    e.threshold = loop.threshold;
    e.active_out = loop.active_out;
    e.next = stage_ctl(c, k + 1);
    e.targets = c->targets.as<double>();
    e.stage_targets = c->stage_targets.as<double>();
    e.solved = c->solved.as<int32_t>();
    return e;
}

int enqueue_stage(slam_ctx* c, int k, const int32_t* gate_seq, const int32_t* d_active, int64_t n_upper,
                  const double* d_x0, const slam_opt_params* prm, const SpanLoopStep* loop) {
    const int n = 6 * (k + 1);
    const int64_t M = n_upper * (int64_t)prm->restarts;
    if (M <= 0) return SLAM_OK;
    StageCtl* ctl = stage_ctl(c, k);
    const bool merge = loop != nullptr;
    if (!(loop && loop->inputs_ready)) {
        const int64_t nt = n_upper * 16;
        hipLaunchKernelGGL(stage_prepare_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, c->stream,
                           c->targets.as<double>(), d_active, ctl, c->stage_targets.as<double>(), c->solved.as<int32_t>());
        HIP_TRY(hipGetLastError());
    }
    const double* d_stage_targets = d_active ? c->stage_targets.as<double>() : c->targets.as<double>();
    // which finished restart stops its siblings: in ordered mode the one the reference's loop breaks at -- best
    // result below the success threshold (optimizer.py:287) -- otherwise one that reached stop_loss
    const double exit_loss = ((prm->flags & SLAM_FLAG_ORDERED) && loop) ? loop->exit_loss
                             : (((prm->flags & SLAM_FLAG_ORDERED) && c->stage_exit_loss >= 0.0) ? c->stage_exit_loss : prm->stop_loss);
    StageLaunch sl{exit_loss, gate_seq, d_stage_targets, d_active, 0, d_x0, M, prm, ctl};
    int rc;
    const int gc = classify_gates(c, k, gate_seq);
#define SLAM_MIN_CASE(KK)                                                   \
    case KK:                                                                \
        if (gc == GC_CX) rc = launch_minimize<KK, GC_CX>(c, sl);            \
        else if (gc == GC_XRI1) rc = launch_minimize<KK, GC_XRI1>(c, sl);   \
        else if (gc == GC_XRI) rc = launch_minimize<KK, GC_XRI>(c, sl);     \
        else if (gc == GC_XGEN) rc = launch_minimize<KK, GC_XGEN>(c, sl);   \
        else rc = launch_minimize<KK, GC_DENSE>(c, sl);                     \
        break;
    switch (k) {
        SLAM_MIN_CASE(1)
        SLAM_MIN_CASE(2)
        SLAM_MIN_CASE(3)
        SLAM_MIN_CASE(4)
        SLAM_MIN_CASE(5)
        default:
            if (k < 1 || k > SLAM_MAX_SPAN_MINIMIZE) return fail(SLAM_ERR_UNSUPPORTED, "minimize supports spans 1..%d (got %d)", SLAM_MAX_SPAN_MINIMIZE, k);
            rc = launch_minimize_long(c, sl, k);  // SLAM_MAX_SPAN_QUAD < k: one wavefront per item
            break;
    }
    if (rc != SLAM_OK) return rc;

    const ReduceArgs r = build_reduce_args(c, k, d_active, prm, exit_loss, merge);
    if (loop) {
        // reduction, bookkeeping, compaction and the next stage's inputs in ONE launch: a single workgroup for small
        // batches (ordered compaction without atomics), a grid of 256-thread workgroups beyond
        const EpilogueArgs e = build_epilogue_args(c, k, r, *loop);
        if (n_upper <= 2048) hipLaunchKernelGGL(stage_epilogue_kernel<256>, dim3(1), dim3(256), 0, c->stream, e);
        else if (n_upper <= kEpilogueMaxTargets) hipLaunchKernelGGL(stage_epilogue_kernel<1024>, dim3(1), dim3(1024), 0, c->stream, e);
        else hipLaunchKernelGGL(stage_epilogue_grid_kernel, dim3((unsigned)((n_upper + 255) / 256)), dim3(256), 0, c->stream, e);
        HIP_TRY(hipGetLastError());
        return SLAM_OK;
    }
    // single-stage call: per-target reduction only
    const int rb = 256;
    hipLaunchKernelGGL(reduce_merge_kernel, dim3((unsigned)((n_upper + rb - 1) / rb)), dim3(rb), 0, c->stream, r);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

// After the stream has drained: fold the stages' control blocks and kernel brackets into the statistics.
int collect_stats(slam_ctx* c, int k_min, int k_max, const StageCtl* h_ctl, int restarts) {
    for (int k = k_min; k <= k_max; ++k) {
        if (h_ctl[k].n_active <= 0) continue;
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_a[k], c->ev_b[k]));
        c->stats.kernel_ms += ms;
        c->stats.kernel_ms_span[k] += ms;
        c->stats.kernel_launches += 1;
        c->stats.evals[k] += (int64_t)h_ctl[k].evals;
        c->stats.evals_accepted[k] += (int64_t)h_ctl[k].evals_accepted;
        c->stats.evals_preempted[k] += (int64_t)h_ctl[k].evals_preempted;
        c->stats.wave_rounds[k] += (int64_t)h_ctl[k].rounds;
        c->stats.items[k] += (int64_t)h_ctl[k].n_active * restarts;
    }
    return SLAM_OK;
}

int ensure_results_n(slam_ctx* c, int nmax);
int ensure_results(slam_ctx* c, int k_max) { return ensure_results_n(c, 6 * (k_max + 1)); }
int ensure_results_n(slam_ctx* c, int nmax) {
    const void* p0 = c->best_loss.p;
    const void* p1 = c->best_cycles.p;
    HIP_TRY(c->best_loss.reserve(c->n_targets * sizeof(double)));
    HIP_TRY(c->best_x.reserve(c->n_targets * (size_t)nmax * sizeof(double)));
    HIP_TRY(c->best_cycles.reserve(c->n_targets * sizeof(int32_t)));
    HIP_TRY(c->span_loss.reserve(c->n_targets * (size_t)kSpanLossStride * sizeof(double)));
    if (c->result_nmax != nmax || c->result_filled != c->n_targets || p0 != c->best_loss.p || p1 != c->best_cycles.p) {
        // new batch, new row width or new allocation: every target starts as "nothing found yet", so that a
        // window no call has decomposed reads as (+inf, -1) instead of uninitialised memory
        hipLaunchKernelGGL(fill_results_kernel, dim3((unsigned)((c->n_targets + 255) / 256)), dim3(256), 0, c->stream,
                           c->best_loss.as<double>(), c->best_cycles.as<int32_t>(), c->span_loss.as<double>(), c->n_targets);
        HIP_TRY(hipGetLastError());
        c->result_filled = c->n_targets;
    }
    c->result_nmax = nmax;
    return SLAM_OK;
}

// Results of the window [first, first + count) on their way to the host: small windows go through pinned
// staging (asynchronous copies, no spinning inside the runtime's pageable path), big ones straight into the
// caller's arrays.  enqueue_fetch only enqueues; finish_fetch runs after the stream has drained.
struct FetchReq {
    double* best_loss;
    double* best_x;
    int32_t* best_cycles;
    size_t b_loss = 0, b_x = 0, b_cyc = 0;
    bool staged = false;
};

int enqueue_fetch_n(slam_ctx* ctx, int nmax, int64_t first, int64_t count, FetchReq& fr);
int enqueue_fetch(slam_ctx* ctx, int k_layout, int64_t first, int64_t count, FetchReq& fr) {
    return enqueue_fetch_n(ctx, 6 * (k_layout + 1), first, count, fr);
}
int enqueue_fetch_n(slam_ctx* ctx, int nmax, int64_t first, int64_t count, FetchReq& fr) {
    if (ctx->result_nmax != nmax || ctx->n_targets <= 0)
        return fail(SLAM_ERR_STATE, "no resident results with rows of %d parameters", nmax);
    if (first < 0 || count < 0 || first + count > ctx->n_targets)
        return fail(SLAM_ERR_INVALID, "target window outside the resident batch");
    const size_t N = (size_t)count;
    const size_t o = (size_t)first;
    fr.b_loss = N * sizeof(double);
    fr.b_x = N * nmax * sizeof(double);
    fr.b_cyc = N * sizeof(int32_t);
    if (N == 0) return SLAM_OK;
    const size_t need = fr.b_loss + fr.b_x + fr.b_cyc;
    char* h = nullptr;
    if (need <= (size_t)1 << 20) {
        if (need > ctx->h_stage_cap) {
            if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
            ctx->h_stage = nullptr;
            ctx->h_stage_cap = 0;
            HIP_TRY(hipHostMalloc(&ctx->h_stage, need + need / 4, hipHostMallocDefault));
            ctx->h_stage_cap = need + need / 4;
        }
        h = static_cast<char*>(ctx->h_stage);
        fr.staged = true;
    }
    // (big windows: an extra host copy of tens of MB costs more than the runtime's own staging -- round 5 A/B with everything staged:
    //  the driver's command 14.8 -> 15.2 ms per step, 327 680 targets through the API 81.8 -> 83.3 ms)
    if (fr.best_loss)
        HIP_TRY(hipMemcpyAsync(h ? (void*)h : (void*)fr.best_loss, ctx->best_loss.as<double>() + o, fr.b_loss, hipMemcpyDeviceToHost, ctx->stream));
    if (fr.best_x)
        HIP_TRY(hipMemcpyAsync(h ? (void*)(h + fr.b_loss) : (void*)fr.best_x, ctx->best_x.as<double>() + o * nmax, fr.b_x, hipMemcpyDeviceToHost, ctx->stream));
    if (fr.best_cycles)
        HIP_TRY(hipMemcpyAsync(h ? (void*)(h + fr.b_loss + fr.b_x) : (void*)fr.best_cycles, ctx->best_cycles.as<int32_t>() + o, fr.b_cyc, hipMemcpyDeviceToHost, ctx->stream));
    return SLAM_OK;
}

void finish_fetch(slam_ctx* ctx, const FetchReq& fr) {
    if (!fr.staged) return;
    const char* h = static_cast<const char*>(ctx->h_stage);
    if (fr.best_loss) std::memcpy(fr.best_loss, h, fr.b_loss);
    if (fr.best_x) std::memcpy(fr.best_x, h + fr.b_loss, fr.b_x);
    if (fr.best_cycles) std::memcpy(fr.best_cycles, h + fr.b_loss + fr.b_x, fr.b_cyc);
}

// -----------------------------------------------------------------------------------------------------------------------
// Small batches: the whole span loop of a target in ONE wavefront (span_wave_kernel, slam_kernels.hpp) -- one launch per call, no
// stage barrier across targets, no bookkeeping launches.  Used when the batch is at most kWaveLoopTargetsPerSimd targets per SIMD,
// spans <= 3, ordered early exit, one gate structure class for all spans; SLAM_FLAG_STAGED forces the per-span launches.
// -----------------------------------------------------------------------------------------------------------------------
constexpr int kWaveLoopTargetsPerSimd = 1;

template <int GC>
int launch_span_wave(slam_ctx* c, const WaveLoopArgs& a, int64_t count) {
    const size_t lds = sizeof(double) * (size_t)(lds_doubles<3, GC>() + kWlLdsDoubles);
    if (c->resident_waves_wl[GC] == 0) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&span_wave_kernel<GC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&span_wave_kernel<GC>), kWave, lds));
        c->resident_waves_wl[GC] = (int64_t)(per_cu < 1 ? 1 : per_cu) * c->compute_units;
    }
    const int64_t blocks = count;  // one wavefront per target
    hipLaunchKernelGGL((span_wave_kernel<GC>), dim3((unsigned)blocks), dim3(kWave), lds, c->stream, a);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

// speculative spans: one launch per span, each on its own stream, then the merge (span_spec_kernel / span_merge_kernel)
template <int K, int GC>
int launch_span_spec(slam_ctx* c, const WaveLoopArgs& a, int64_t count, hipStream_t stream) {
    const size_t lds = sizeof(double) * (size_t)(lds_doubles<K, GC>() + kWlLdsDoubles);
    if (!c->spec_attr_set[K][GC]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&span_spec_kernel<K, GC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->spec_attr_set[K][GC] = true;
    }
    hipLaunchKernelGGL((span_spec_kernel<K, GC>), dim3((unsigned)count), dim3(kWave), lds, stream, a);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

template <int K>
int launch_span_spec_gc(slam_ctx* c, int gc, const WaveLoopArgs& a, int64_t count, hipStream_t stream) {
    if (gc == GC_CX) return launch_span_spec<K, GC_CX>(c, a, count, stream);
    if (gc == GC_XRI1) return launch_span_spec<K, GC_XRI1>(c, a, count, stream);
    if (gc == GC_XRI) return launch_span_spec<K, GC_XRI>(c, a, count, stream);
    if (gc == GC_XGEN) return launch_span_spec<K, GC_XGEN>(c, a, count, stream);
    return launch_span_spec<K, GC_DENSE>(c, a, count, stream);
}

constexpr int64_t kOverlapMaxItems = 1 << 17;  // four times what the chip holds at once (2048 wavefronts x 16 items)

// would decompose_overlapped (below) take this call?
bool overlap_eligible(const slam_ctx* c, int64_t count, int k_min, int k_max, const slam_opt_params* prm) {
    static const bool env_staged = std::getenv("SLAM_STAGED") != nullptr;
    static const bool env_off = []{ const char* e = std::getenv("SLAM_OVERLAP"); return e && e[0] == '0'; }();
    if ((prm->flags & SLAM_FLAG_STAGED) || env_staged || env_off) return false;
    if (!(prm->flags & SLAM_FLAG_EARLY_EXIT) || !(prm->flags & SLAM_FLAG_ORDERED)) return false;
    if (k_max <= k_min || k_max > 3 || c->trace_cap > 0) return false;
    if (prm->flags & SLAM_FLAG_OVERLAP) return true;
    return !(prm->flags & SLAM_FLAG_NO_OVERLAP) && count * (int64_t)prm->restarts <= kOverlapMaxItems;
}

// returns SLAM_OK and *taken = true when the call was served by the wave-loop kernel; *taken = false: not eligible (nothing enqueued)
int decompose_wave_loop(slam_ctx* c, int64_t first, int64_t count, int k_min, int k_max, const int32_t* gate_seqs, const slam_opt_params* prm,
                        double success_threshold, FetchReq* fetch, bool* taken) {
    *taken = false;
    if (prm->flags & SLAM_FLAG_STAGED) return SLAM_OK;
    static const bool env_staged = std::getenv("SLAM_STAGED") != nullptr;  // (test runs: the whole suite through the per-span launches)
    static const bool env_no_wave = []{ const char* e = std::getenv("SLAM_WAVE_LOOP"); return e && e[0] == '0'; }();  // (A/B runs)
    if (env_staged || env_no_wave) return SLAM_OK;
    if (!(prm->flags & SLAM_FLAG_EARLY_EXIT) || !(prm->flags & SLAM_FLAG_ORDERED)) return SLAM_OK;
    if (k_max > 3 || c->trace_cap > 0) return SLAM_OK;
    if (count > (int64_t)kWaveLoopTargetsPerSimd * 4 * c->compute_units) return SLAM_OK;
    // a wavefront runs its target's restarts 16 at a time: with many restarts and few targets the per-span launches, which spread the
    // items over the whole chip, are faster (measured, CNOT, profiles/r4_wave_probe.txt: R = 32: 1024 targets 1.34 vs 1.63 ms, 16
    // targets 1.00 vs 0.81; R = 64: 1024 targets 2.01 vs 2.41, 256 targets 1.96 vs 1.44; R = 128: 1024 targets 3.52 vs 3.25)
    if (prm->restarts > 16 && !(prm->restarts <= 64 && count * (int64_t)prm->restarts >= 32768)) return SLAM_OK;
    // ... and with more than 16 restarts the spans side by side on the whole chip (decompose_overlapped) beat it where they may run
    // (tools/r4_wave_vs_overlap.py, 1024 targets: R = 32 1.29 -> 1.15 ms CNOT, 1.47 -> 1.31 sqrt(iSWAP); R = 64 2.01 -> 1.67, 2.26 -> 1.57;
    // with 16 restarts the wave kernels win: 0.90 vs 1.08, 512 targets 0.67 vs 0.75)
    if (prm->restarts > 16 && overlap_eligible(c, count, k_min, k_max, prm)) return SLAM_OK;
    int gc = -1;
    {
        const int32_t* gs = gate_seqs;
        for (int k = k_min; k <= k_max; ++k) {
            const int g = classify_gates(c, k, gs);
            if (gc >= 0 && g != gc) return SLAM_OK;  // (per-span classes differ: the per-span launches use each span's own class)
            gc = g;
            gs += k;
        }
    }
    // argument blocks of the spans (the kernels' own MinimizeArgs layout), staged through pinned memory
    const size_t need = sizeof(MinimizeArgs<1>) * (size_t)(SLAM_MAX_SPAN_EVAL + 1);
    HIP_TRY(c->mq_args.reserve(need));
    if (need > c->h_mq_cap) {
        if (c->h_mq_args) (void)hipHostFree(c->h_mq_args);
        c->h_mq_args = nullptr;
        c->h_mq_cap = 0;
        HIP_TRY(hipHostMalloc(&c->h_mq_args, need, hipHostMallocDefault));
        c->h_mq_cap = need;
    }
    MinimizeArgs<1>* h = static_cast<MinimizeArgs<1>*>(c->h_mq_args);
    const int32_t* gs = gate_seqs;
    for (int k = k_min; k <= k_max; ++k) {
        StageLaunch sl{success_threshold, gs, c->targets.as<double>(), nullptr, 0, nullptr, count * (int64_t)prm->restarts, prm, stage_ctl(c, k)};
        int rc;
        switch (k) {
            case 1: rc = build_minimize_args<1>(c, sl, *reinterpret_cast<MinimizeArgs<1>*>(&h[k])); break;
            case 2: rc = build_minimize_args<2>(c, sl, *reinterpret_cast<MinimizeArgs<2>*>(&h[k])); break;
            default: rc = build_minimize_args<3>(c, sl, *reinterpret_cast<MinimizeArgs<3>*>(&h[k])); break;
        }
        if (rc) return rc;
        gs += k;
    }
    *taken = true;
    HIP_TRY(hipEventRecord(c->ev_t0, c->stream));
    HIP_TRY(hipMemcpyAsync(c->mq_args.p, h, need, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->counters.p, 0, sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), c->stream));
    WaveLoopArgs a{};
    a.stage_args = c->mq_args.as<MinimizeArgs<1>>();
    a.ctl = c->counters.as<StageCtl>();
    a.k_min = k_min;
    a.k_max = k_max;
    a.first = (int32_t)first;
    a.count = (int32_t)count;
    a.threshold = success_threshold;
    a.best_loss = c->best_loss.as<double>();
    a.best_x = c->best_x.as<double>();
    a.best_cycles = c->best_cycles.as<int32_t>();
    a.span_loss = c->span_loss.as<double>();
    a.nmax = c->result_nmax;
    // every wavefront leaves a stage after this many rounds whatever happens: far above what R restarts of maxiter iterations
    // (each at most 1 + 21 line-search evaluations) can need when they run 16 at a time
    {
        const double need_rounds = ((double)prm->restarts / 16.0 + 1.0) * ((double)prm->maxiter + 2.0) * 22.0;
        a.round_cap = need_rounds > 4.0e9 ? 0xffffffffu : (uint32_t)need_rounds;
    }
    // speculative spans (see span_spec_kernel): all spans of all targets at once when the loop has more than one span.
    // SLAM_SPECULATE=0 keeps the one-wavefront-per-target loop (A/B runs, tests of that kernel).
    static const bool env_no_spec = []{ const char* e = std::getenv("SLAM_SPECULATE"); return e && e[0] == '0'; }();
    // Worth it while every (target, span) wavefront finds a SIMD of its own (measured, tools/r4_spec_probe.sh, CNOT x 16 restarts: 1
    // target 0.51 -> 0.28 ms, 256 targets 0.80 -> 0.55; at 1024 targets the k = 2 and k = 3 stages no longer fit beside each other
    // -- 256 + 304 registers -- and the two forms tie: 0.94 / 0.89 ms CNOT, 1.14 / 1.18 ms sqrt(iSWAP), where k = 3 is mostly wasted)
    const bool spec = k_max > k_min && !env_no_spec && count <= 2 * (int64_t)c->compute_units;
    int n_launch = 1;
    HIP_TRY(hipEventRecord(c->ev_a[0], c->stream));  // bracket 0: the whole call's optimizer work (kernel_ms)
    int rc;
    if (spec) {
        HIP_TRY(c->spec_loss.reserve(sizeof(double) * 3 * (size_t)count));
        HIP_TRY(c->spec_x.reserve(sizeof(double) * 3 * (size_t)count * (size_t)c->result_nmax));
        HIP_TRY(c->spec_ev.reserve(sizeof(unsigned long long) * 9 * (size_t)count));
        a.spec_loss = c->spec_loss.as<double>();
        a.spec_x = c->spec_x.as<double>();
        a.spec_ev = c->spec_ev.as<unsigned long long>();
        if (!c->spec_stream[0]) {
            for (int j = 0; j < 2; ++j) {
                HIP_TRY(hipStreamCreateWithFlags(&c->spec_stream[j], hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&c->spec_join[j], hipEventDisableTiming));
            }
            HIP_TRY(hipEventCreateWithFlags(&c->spec_fork, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(c->spec_fork, c->stream));
        // the longest stage first, on the call's own stream; the others beside it
        int side = 0;
        for (int k = k_max; k >= k_min; --k) {
            hipStream_t st = c->stream;
            if (k != k_max) {
                st = c->spec_stream[side];
                HIP_TRY(hipStreamWaitEvent(st, c->spec_fork, 0));
            }
            HIP_TRY(hipEventRecord(c->ev_a[k], st));  // span k's own launch (the spans overlap in time: kernel_ms_span[k])
            switch (k) {
                case 1: rc = launch_span_spec_gc<1>(c, gc, a, count, st); break;
                case 2: rc = launch_span_spec_gc<2>(c, gc, a, count, st); break;
                default: rc = launch_span_spec_gc<3>(c, gc, a, count, st); break;
            }
            if (rc) return rc;
            HIP_TRY(hipEventRecord(c->ev_b[k], st));
            if (k != k_max) {
                HIP_TRY(hipEventRecord(c->spec_join[side], st));
                HIP_TRY(hipStreamWaitEvent(c->stream, c->spec_join[side], 0));
                ++side;
            }
        }
        SpanMergeArgs mg{};
        mg.k_min = k_min;
        mg.k_max = k_max;
        mg.first = (int32_t)first;
        mg.count = (int32_t)count;
        mg.nmax = c->result_nmax;
        mg.threshold = success_threshold;
        for (int k = k_min; k <= k_max; ++k) {
            mg.loss[k] = a.spec_loss + (size_t)(k - 1) * count;
            mg.x[k] = a.spec_x + (size_t)(k - 1) * count * c->result_nmax;
            mg.xstride[k] = c->result_nmax;
            mg.xn[k] = 6 * (k + 1);
            mg.ev[k] = a.spec_ev + (size_t)(k - 1) * count * 3;
            mg.src_ctl[k] = nullptr;  // (the stage kernels add their rounds themselves)
        }
        mg.ctl = a.ctl;
        mg.best_loss = a.best_loss;
        mg.best_x = a.best_x;
        mg.best_cycles = a.best_cycles;
        mg.span_loss = a.span_loss;
        hipLaunchKernelGGL(span_merge_kernel, dim3((unsigned)((count + kWave - 1) / kWave)), dim3(kWave), 0, c->stream, mg);
        HIP_TRY(hipGetLastError());
        n_launch = (k_max - k_min + 1) + 1;
    } else {
        if (gc == GC_CX) rc = launch_span_wave<GC_CX>(c, a, count);
        else if (gc == GC_XRI1) rc = launch_span_wave<GC_XRI1>(c, a, count);
        else if (gc == GC_XRI) rc = launch_span_wave<GC_XRI>(c, a, count);
        else if (gc == GC_XGEN) rc = launch_span_wave<GC_XGEN>(c, a, count);
        else rc = launch_span_wave<GC_DENSE>(c, a, count);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(c->ev_b[0], c->stream));
    HIP_TRY(hipEventRecord(c->ev_t1, c->stream));
    if (fetch) {
        rc = enqueue_fetch_n(c, c->result_nmax, first, count, *fetch);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(c->h_ctl, c->counters.p, sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->ev_done, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev_done));
    if (fetch) finish_fetch(c, *fetch);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
    c->stats.total_ms = ms;
    float kms = 0.f;
    HIP_TRY(hipEventElapsedTime(&kms, c->ev_a[0], c->ev_b[0]));
    c->stats.kernel_ms += kms;
    if (spec) {
        // one launch per span, side by side: every span's own bracket (they overlap: their sum exceeds kernel_ms)
        for (int k = k_min; k <= k_max; ++k) {
            float sk = 0.f;
            HIP_TRY(hipEventElapsedTime(&sk, c->ev_a[k], c->ev_b[k]));
            c->stats.kernel_ms_span[k] += sk;
        }
    } else {
        c->stats.kernel_ms_span[0] += kms;  // ONE launch for all spans: no per-span split exists -- booked under index 0
    }
    c->stats.kernel_launches += n_launch;
    for (int k = k_min; k <= k_max; ++k) {
        if (c->h_ctl[k].n_active <= 0 && c->h_ctl[k].evals == 0) continue;
        c->stats.evals[k] += (int64_t)c->h_ctl[k].evals;
        c->stats.evals_accepted[k] += (int64_t)c->h_ctl[k].evals_accepted;
        c->stats.evals_preempted[k] += (int64_t)c->h_ctl[k].evals_preempted;
        c->stats.wave_rounds[k] += (int64_t)c->h_ctl[k].rounds;
        c->stats.items[k] += (int64_t)c->h_ctl[k].n_active * prm->restarts;
    }
    return SLAM_OK;
}

// -----------------------------------------------------------------------------------------------------------------------
// Medium batches: the spans of the loop side by side.  The stages of a target do not use each other's results, only the decision
// whether they are needed (optimizer.py:301-303), and their start points are keyed by span -- so every span runs for ALL targets of
// the window at once, each as the ordinary per-span pipeline (stage inputs, optimizer kernel, per-target reduction: what
// slam_minimize_stage runs) on a helper context with its own stream and stage buffers, and span_merge_kernel then applies the loop's
// bookkeeping in span order.  Results are those of the staged launches bit for bit; the work of stages the loop would not have reached
// is wasted (booked as pre-empted evaluations), which is why this form is taken only while one call cannot fill the chip for long
// (targets x restarts <= kOverlapMaxItems and no SLAM_FLAG_NO_OVERLAP; measured, tools/r4_overlap_probe.py: CNOT 4096 x 16 2.84 -> 1.86 ms, 20 480 x 16 5.1 -> 4.9,
// sqrt(iSWAP) 65 536 x 32 18.1 -> 16.5) or when the caller asks for it (SLAM_FLAG_OVERLAP).
// -----------------------------------------------------------------------------------------------------------------------
__global__ void iota_kernel(int32_t* out, int32_t first, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = first + (int32_t)i;
}

int decompose_overlapped(slam_ctx* c, int64_t first, int64_t count, int k_min, int k_max, const int32_t* gate_seqs, const slam_opt_params* prm,
                         double success_threshold, FetchReq* fetch, bool* taken) {
    *taken = false;
    if (!overlap_eligible(c, count, k_min, k_max, prm)) return SLAM_OK;
    // helper contexts (one per span), created on first use; their targets are this context's (borrowed for the call)
    for (int k = k_min; k <= k_max; ++k) {
        if (!c->helper[k]) {
            int rc = slam_ctx_create(c->device, &c->helper[k]);
            if (rc) return rc;
            HIP_TRY(hipEventCreateWithFlags(&c->ov_join[k], hipEventDisableTiming));
        }
    }
    if (!c->ov_fork) HIP_TRY(hipEventCreateWithFlags(&c->ov_fork, hipEventDisableTiming));
    struct Unborrow {
        slam_ctx* c;
        ~Unborrow() {
            for (slam_ctx* h : c->helper)
                if (h) {
                    h->targets.p = nullptr;
                    h->targets.cap = 0;
                    h->n_targets = 0;
                }
        }
    } unborrow{c};
    *taken = true;
    HIP_TRY(hipEventRecord(c->ev_t0, c->stream));
    HIP_TRY(hipMemsetAsync(c->counters.p, 0, sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), c->stream));
    HIP_TRY(hipEventRecord(c->ev_a[0], c->stream));  // bracket 0: the whole call's optimizer work (kernel_ms)
    HIP_TRY(hipEventRecord(c->ov_fork, c->stream));
    SpanMergeArgs mg{};
    const int32_t* gs = gate_seqs;
    // the spans are enqueued longest first (k_max ... k_min): each on its helper's stream
    const int32_t* seq_of[SLAM_MAX_SPAN_EVAL + 1] = {};
    for (int k = k_min; k <= k_max; ++k) {
        seq_of[k] = gs;
        gs += k;
    }
    for (int k = k_max; k >= k_min; --k) {
        slam_ctx* h = c->helper[k];
        h->targets.p = c->targets.p;
        h->targets.cap = c->targets.cap;
        h->n_targets = c->n_targets;
        if (h->helper_gates_version != c->gates_version) {
            h->gates_host = c->gates_host;
            h->n_gates = c->n_gates;
            for (auto& st : h->staged) st.valid = false;
            h->helper_gates_version = c->gates_version;
        }
        h->cost_kind = c->cost_kind;
        h->reserve_waves = c->reserve_waves;
        h->stage_exit_loss = success_threshold;  // ordered early exit at the loop's threshold (optimizer.py:287)
        h->slot_ev_on = true;
        HIP_TRY(h->active.reserve((size_t)count * sizeof(int32_t)));
        HIP_TRY(h->slot_ev.reserve((size_t)count * 3 * sizeof(unsigned long long)));
        int rc = reserve_stage_buffers(h, count, k, prm);
        if (rc) return rc;
        HIP_TRY(hipStreamWaitEvent(h->stream, c->ov_fork, 0));
        hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->stream, h->active.as<int32_t>(), (int32_t)first, count);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemsetAsync(h->counters.p, 0, sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), h->stream));
        hipLaunchKernelGGL(set_n_active_kernel, dim3(1), dim3(1), 0, h->stream, stage_ctl(h, k), (int32_t)count);
        HIP_TRY(hipGetLastError());
        rc = enqueue_stage(h, k, seq_of[k], h->active.as<int32_t>(), count, nullptr, prm, nullptr);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(c->ov_join[k], h->stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ov_join[k], 0));
        mg.loss[k] = h->stage_loss.as<double>();
        mg.x[k] = h->stage_x.as<double>();
        mg.xstride[k] = 6 * (k + 1);
        mg.xn[k] = 6 * (k + 1);
        mg.ev[k] = h->slot_ev.as<unsigned long long>();
        mg.src_ctl[k] = stage_ctl(h, k);
    }
    mg.k_min = k_min;
    mg.k_max = k_max;
    mg.first = (int32_t)first;
    mg.count = (int32_t)count;
    mg.nmax = c->result_nmax;
    mg.threshold = success_threshold;
    mg.ctl = c->counters.as<StageCtl>();
    mg.best_loss = c->best_loss.as<double>();
    mg.best_x = c->best_x.as<double>();
    mg.best_cycles = c->best_cycles.as<int32_t>();
    mg.span_loss = c->span_loss.as<double>();
    hipLaunchKernelGGL(span_merge_kernel, dim3((unsigned)((count + kWave - 1) / kWave)), dim3(kWave), 0, c->stream, mg);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_b[0], c->stream));
    HIP_TRY(hipEventRecord(c->ev_t1, c->stream));
    int rc = SLAM_OK;
    if (fetch) {
        rc = enqueue_fetch_n(c, c->result_nmax, first, count, *fetch);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(c->h_ctl, c->counters.p, sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->ev_done, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev_done));
    if (fetch) finish_fetch(c, *fetch);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
    c->stats.total_ms = ms;
    float kms = 0.f;
    HIP_TRY(hipEventElapsedTime(&kms, c->ev_a[0], c->ev_b[0]));
    c->stats.kernel_ms += kms;
    for (int k = k_min; k <= k_max; ++k) {
        // every span's own optimizer launch, bracketed on its helper's stream by enqueue_stage (the spans overlap in time: the sum of
        // kernel_ms_span exceeds kernel_ms -- ADVICE r4: the whole used to be booked on the first span)
        float sk = 0.f;
        HIP_TRY(hipEventElapsedTime(&sk, c->helper[k]->ev_a[k], c->helper[k]->ev_b[k]));
        c->stats.kernel_ms_span[k] += sk;
    }
    c->stats.kernel_launches += 2 * (k_max - k_min + 1) + 1;
    for (int k = k_min; k <= k_max; ++k) {
        c->stats.evals[k] += (int64_t)c->h_ctl[k].evals;
        c->stats.evals_accepted[k] += (int64_t)c->h_ctl[k].evals_accepted;
        c->stats.evals_preempted[k] += (int64_t)c->h_ctl[k].evals_preempted;
        c->stats.wave_rounds[k] += (int64_t)c->h_ctl[k].rounds;
        c->stats.items[k] += (int64_t)c->h_ctl[k].n_active * prm->restarts;
    }
    return SLAM_OK;
}

// h_list != nullptr: the batch is the explicit list of resident-target indices h_list[0..count) (first ignored)
int decompose_body(slam_ctx* c, int64_t first, int64_t count, int k_min, int k_max, const int32_t* gate_seqs,
                   const slam_opt_params* prm, double success_threshold, const int32_t* h_list, int k_layout,
                   FetchReq* fetch) {
    if (!c) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (c->n_targets <= 0) return fail(SLAM_ERR_STATE, "no targets: call slam_set_targets first");
    if (c->n_gates <= 0) return fail(SLAM_ERR_STATE, "no gates: call slam_set_gates first");
    if (k_min < 1 || k_max < k_min) return fail(SLAM_ERR_INVALID, "bad span range [%d, %d]", k_min, k_max);
    if (k_max > SLAM_MAX_SPAN_MINIMIZE)
        return fail(SLAM_ERR_UNSUPPORTED, "minimize supports spans 1..%d (got k_max = %d)", SLAM_MAX_SPAN_MINIMIZE, k_max);
    int rc = check_params(prm);
    if (rc) return rc;
    {
        const int32_t* gs = gate_seqs;
        for (int k = k_min; k <= k_max; ++k) {
            rc = check_gate_seq(c, k, gs);
            if (rc) return rc;
            gs += k;
        }
    }
    if (h_list) {
        if (count <= 0) return fail(SLAM_ERR_INVALID, "empty target list");
        for (int64_t i = 0; i < count; ++i)
            if (h_list[i] < 0 || h_list[i] >= c->n_targets)
                return fail(SLAM_ERR_INVALID, "targets[%lld] = %d outside [0, %lld)", (long long)i, h_list[i], (long long)c->n_targets);
        first = 0;
    } else if (first < 0 || count <= 0 || first + count > c->n_targets)
        return fail(SLAM_ERR_INVALID, "target window [%lld, %lld) outside [0, %lld)", (long long)first,
                    (long long)(first + count), (long long)c->n_targets);
    if (k_layout == 0) k_layout = k_max;
    if (k_layout < k_max || k_layout > SLAM_MAX_SPAN_MINIMIZE) return fail(SLAM_ERR_INVALID, "k_layout must be in [k_max, %d]", SLAM_MAX_SPAN_MINIMIZE);
    if (c->result_nmax != 0 && c->result_nmax != 6 * (k_layout + 1) && (h_list || !(first == 0 && count == c->n_targets)))
        return fail(SLAM_ERR_STATE, "resident results were produced with a different k_max");
    rc = ensure_results(c, k_layout);
    if (rc) return rc;
    if (!h_list && k_layout == k_max) {
        bool taken = false;
        rc = decompose_wave_loop(c, first, count, k_min, k_max, gate_seqs, prm, success_threshold, fetch, &taken);
        if (rc || taken) return rc;
        rc = decompose_overlapped(c, first, count, k_min, k_max, gate_seqs, prm, success_threshold, fetch, &taken);
        if (rc || taken) return rc;
    }
    const int64_t N = count;
    HIP_TRY(c->active.reserve(N * sizeof(int32_t)));
    HIP_TRY(c->active2.reserve(N * sizeof(int32_t)));
    rc = reserve_stage_buffers(c, N, k_max, prm);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev_t0, c->stream));
    const bool whole = !h_list && (first == 0 && count == c->n_targets);
    if (h_list) HIP_TRY(hipMemcpyAsync(c->active.p, h_list, (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(init_results_kernel, dim3((unsigned)(((whole ? N : N * 16) + 255) / 256)), dim3(256), 0, c->stream,
                       c->best_loss.as<double>(), c->best_cycles.as<int32_t>(), c->span_loss.as<double>(),
                       whole ? (int32_t*)nullptr : c->active.as<int32_t>(), first, N, stage_ctl(c, k_min),
                       c->targets.as<double>(), c->stage_targets.as<double>(), c->solved.as<int32_t>(),
                       c->counters.as<StageCtl>(), (int32_t)(sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2) / 8), h_list ? 1 : 0);
    HIP_TRY(hipGetLastError());

    // The whole span loop is enqueued at once: a stage's target count is produced on the device by the
    // previous stage's compaction (every stage is sized for N on the host; a stage without targets costs a
    // few empty launches), so the only host synchronisation is the one at the end.
    const int32_t* d_active = whole ? nullptr : c->active.as<int32_t>();  // nullptr = identity
    DevBuf* cur = &c->active;
    DevBuf* nxt = &c->active2;
    const int32_t* gs = gate_seqs;
    for (int k = k_min; k <= k_max; ++k) {
        // the first stage's inputs come from init_results_kernel, the later ones from the previous stage's epilogue
        SpanLoopStep step{success_threshold, true, k < k_max, success_threshold, nxt->as<int32_t>()};
        rc = enqueue_stage(c, k, gs, d_active, N, nullptr, prm, &step);
        if (rc) return rc;
        gs += k;
        if (k < k_max) {
            d_active = nxt->as<int32_t>();
            DevBuf* t = cur; cur = nxt; nxt = t;
        }
    }
    HIP_TRY(hipEventRecord(c->ev_t1, c->stream));
    if (fetch) {  // the results ride on the same wait as the span loop
        rc = enqueue_fetch(c, k_layout, h_list ? 0 : first, h_list ? c->n_targets : count, *fetch);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(c->h_ctl, c->counters.p, sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->ev_done, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev_done));
    if (fetch) finish_fetch(c, *fetch);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
    c->stats.total_ms = ms;
    return collect_stats(c, k_min, k_max, c->h_ctl, prm->restarts);
}

int decompose_impl(slam_ctx* c, int64_t first, int64_t count, int k_min, int k_max, const int32_t* gate_seqs,
                   const slam_opt_params* prm, double success_threshold, const int32_t* h_list = nullptr, int k_layout = 0,
                   FetchReq* fetch = nullptr) {
    return drained(c, decompose_body(c, first, count, k_min, k_max, gate_seqs, prm, success_threshold, h_list, k_layout, fetch));
}


// -----------------------------------------------------------------------------------------------------------------------
// slam_decompose_multi: the span loops of SEVERAL contexts -- same device, same target window, each with its own gate table
// (e.g. the bases of a parametric-Hamiltonian sweep, BASELINE configs[4]) -- enqueued as ONE chain of kernels on the first
// context's stream: per span ONE multi-queue optimizer launch (minimize_kernel<K, GC, true>: a wavefront works on one
// sub-problem at a time, so gates stay scalar operands) and ONE epilogue launch for all of them.  Every context ends with the
// results its own slam_decompose_range call would have left (ordered early exit: bit for bit) -- which is why contexts whose gates
// fall into different structure classes get one optimizer launch PER CLASS (measured: the same item run through the GC_XRI1 and the
// GC_XRI instantiation ends 1e-14 apart -- the compiler contracts the surrounding products differently -- so a launch in the most
// general class of the group would not reproduce the per-context calls).
// -----------------------------------------------------------------------------------------------------------------------
template <int K, int GC>
int launch_minimize_multi(slam_ctx* lead, const MinimizeArgs<K>& common, const MinimizeArgs<K>* d_subs, int n, int64_t items_max_total) {
    const size_t lds = lds_bytes<K, GC>();
    { int rc = prepare_minimize_kernel<K, GC, true>(lead); if (rc) return rc; }
    int64_t blocks = (items_max_total + kQuadsPerWave - 1) / kQuadsPerWave;
    const int64_t cap = lead->resident_waves_mq[K][GC] > 0 ? lead->resident_waves_mq[K][GC] : 1;
    if (blocks > cap) blocks = cap;
    if (blocks < n) blocks = n;  // every sub-problem has a wavefront that starts on it
    hipLaunchKernelGGL((minimize_kernel<K, GC, true>), dim3((unsigned)blocks), dim3(kWave), lds, lead->stream, common, d_subs, n);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

int decompose_multi_body(slam_ctx** cs, int n, int64_t first, int64_t count, int k_min, int k_max, const int32_t* gate_seqs,
                         const slam_opt_params* prm, double success_threshold) {
    if (!cs || n <= 0) return fail(SLAM_ERR_INVALID, "no contexts");
    slam_ctx* lead = cs[0];
    for (int i = 0; i < n; ++i) {
        if (!cs[i]) return fail(SLAM_ERR_INVALID, "ctxs[%d] is NULL", i);
        if (cs[i]->device != lead->device) return fail(SLAM_ERR_INVALID, "ctxs[%d] lives on device %d, ctxs[0] on %d", i, cs[i]->device, lead->device);
        for (int j = 0; j < i; ++j)
            if (cs[j] == cs[i]) return fail(SLAM_ERR_INVALID, "ctxs[%d] and ctxs[%d] are the same context", j, i);
        if (cs[i]->n_targets <= 0) return fail(SLAM_ERR_STATE, "ctxs[%d]: no targets", i);
        if (cs[i]->n_gates <= 0) return fail(SLAM_ERR_STATE, "ctxs[%d]: no gates", i);
        if (first < 0 || count <= 0 || first + count > cs[i]->n_targets)
            return fail(SLAM_ERR_INVALID, "ctxs[%d]: target window [%lld, %lld) outside [0, %lld)", i, (long long)first, (long long)(first + count), (long long)cs[i]->n_targets);
        if (cs[i]->cost_kind != lead->cost_kind) return fail(SLAM_ERR_INVALID, "ctxs[%d]: another cost function than ctxs[0]", i);
    }
    HIP_TRY(hipSetDevice(lead->device));
    if (k_min < 1 || k_max < k_min) return fail(SLAM_ERR_INVALID, "bad span range [%d, %d]", k_min, k_max);
    if (k_max > 3) return fail(SLAM_ERR_UNSUPPORTED, "slam_decompose_multi runs spans 1..3 (got k_max = %d)", k_max);
    int rc = check_params(prm);
    if (rc) return rc;
    if (!(prm->flags & SLAM_FLAG_EARLY_EXIT) || !(prm->flags & SLAM_FLAG_ORDERED))
        return fail(SLAM_ERR_INVALID, "slam_decompose_multi needs SLAM_FLAG_EARLY_EXIT | SLAM_FLAG_ORDERED (results independent of scheduling)");
    const int64_t N = count;
    for (int i = 0; i < n; ++i) {
        slam_ctx* c = cs[i];
        const int32_t* gs = gate_seqs;
        for (int k = k_min; k <= k_max; ++k) {
            rc = check_gate_seq(c, k, gs);
            if (rc) return rc;
            gs += k;
        }
        if (c->result_nmax != 0 && c->result_nmax != 6 * (k_max + 1) && !(first == 0 && count == c->n_targets))
            return fail(SLAM_ERR_STATE, "ctxs[%d]: resident results were produced with a different k_max", i);
        // (allocation and the first fill run on the context's own, idle stream; the lead's stream waits for them below)
        rc = ensure_results(c, k_max);
        if (rc) return rc;
        HIP_TRY(c->active.reserve(N * sizeof(int32_t)));
        HIP_TRY(c->active2.reserve(N * sizeof(int32_t)));
        rc = reserve_stage_buffers(c, N, k_max, prm);
        if (rc) return rc;
        if (c != lead) HIP_TRY(hipStreamSynchronize(c->stream));
    }
    // staging: per span n optimizer argument blocks + n epilogue argument blocks
    const size_t sz_ma = sizeof(MinimizeArgs<1>), sz_ep = sizeof(EpilogueArgs);
    static_assert(sizeof(MinimizeArgs<1>) == sizeof(MinimizeArgs<3>), "argument blocks of all spans have one layout");
    const size_t slot = (size_t)n * (sz_ma + sz_ep);
    const size_t need = slot * (size_t)(k_max + 1);
    HIP_TRY(lead->mq_args.reserve(need));
    if (need > lead->h_mq_cap) {
        if (lead->h_mq_args) (void)hipHostFree(lead->h_mq_args);
        lead->h_mq_args = nullptr;
        lead->h_mq_cap = 0;
        HIP_TRY(hipHostMalloc(&lead->h_mq_args, need, hipHostMallocDefault));
        lead->h_mq_cap = need;
    }
    HIP_TRY(hipEventRecord(lead->ev_t0, lead->stream));
    for (int i = 0; i < n; ++i) {
        slam_ctx* c = cs[i];
        const bool whole = (first == 0 && count == c->n_targets);
        hipLaunchKernelGGL(init_results_kernel, dim3((unsigned)(((whole ? N : N * 16) + 255) / 256)), dim3(256), 0, lead->stream,
                           c->best_loss.as<double>(), c->best_cycles.as<int32_t>(), c->span_loss.as<double>(),
                           whole ? (int32_t*)nullptr : c->active.as<int32_t>(), first, N, stage_ctl(c, k_min), c->targets.as<double>(),
                           c->stage_targets.as<double>(), c->solved.as<int32_t>(), c->counters.as<StageCtl>(),
                           (int32_t)(sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2) / 8), 0);
        HIP_TRY(hipGetLastError());
    }
    std::vector<const int32_t*> d_active((size_t)n);
    std::vector<DevBuf*> cur((size_t)n), nxt((size_t)n);
    for (int i = 0; i < n; ++i) {
        const bool whole = (first == 0 && count == cs[i]->n_targets);
        d_active[(size_t)i] = whole ? nullptr : cs[i]->active.as<int32_t>();
        cur[(size_t)i] = &cs[i]->active;
        nxt[(size_t)i] = &cs[i]->active2;
    }
    const int32_t* gs = gate_seqs;
    for (int k = k_min; k <= k_max; ++k) {
        char* h = static_cast<char*>(lead->h_mq_args) + slot * (size_t)k;
        char* d = static_cast<char*>(lead->mq_args.p) + slot * (size_t)k;
        MinimizeArgs<1>* h_ma = reinterpret_cast<MinimizeArgs<1>*>(h);  // (one layout for every span)
        EpilogueArgs* h_ep = reinterpret_cast<EpilogueArgs*>(h + (size_t)n * sz_ma);
        // contexts ordered by the structure class of their gates: one contiguous run of argument blocks, one launch, per class
        std::vector<int> order((size_t)n), cls((size_t)n);
        for (int i = 0; i < n; ++i) { order[(size_t)i] = i; cls[(size_t)i] = classify_gates(cs[i], k, gs); }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cls[(size_t)a] < cls[(size_t)b]; });
        for (int slot_i = 0; slot_i < n; ++slot_i) {
            const int i = order[(size_t)slot_i];
            slam_ctx* c = cs[i];
            const double* d_stage_targets = d_active[(size_t)i] ? c->stage_targets.as<double>() : c->targets.as<double>();
            StageLaunch sl{success_threshold, gs, d_stage_targets, d_active[(size_t)i], 0, nullptr, N * (int64_t)prm->restarts, prm, stage_ctl(c, k)};
            switch (k) {
                case 1: rc = build_minimize_args<1>(c, sl, *reinterpret_cast<MinimizeArgs<1>*>(&h_ma[slot_i])); break;
                case 2: rc = build_minimize_args<2>(c, sl, *reinterpret_cast<MinimizeArgs<2>*>(&h_ma[slot_i])); break;
                default: rc = build_minimize_args<3>(c, sl, *reinterpret_cast<MinimizeArgs<3>*>(&h_ma[slot_i])); break;
            }
            if (rc) return rc;
            if (c != lead) HIP_TRY(hipStreamSynchronize(c->stream));  // (its gate slot may just have been staged on its own stream)
            SpanLoopStep step{success_threshold, true, k < k_max, success_threshold, nxt[(size_t)i]->as<int32_t>()};
            const ReduceArgs r = build_reduce_args(c, k, d_active[(size_t)i], prm, success_threshold, true);
            h_ep[slot_i] = build_epilogue_args(c, k, r, step);
        }
        HIP_TRY(hipMemcpyAsync(d, h, slot, hipMemcpyHostToDevice, lead->stream));
#define SLAM_MQ_CASE(KK)                                                                                                                         \
    case KK: {                                                                                                                                   \
        const MinimizeArgs<KK>& a0 = *reinterpret_cast<const MinimizeArgs<KK>*>(&h_ma[lo]);                                                      \
        const MinimizeArgs<KK>* dsub = reinterpret_cast<const MinimizeArgs<KK>*>(d) + lo;                                                        \
        if (gc == GC_CX) rc = launch_minimize_multi<KK, GC_CX>(lead, a0, dsub, cnt, items_total);                                                \
        else if (gc == GC_XRI1) rc = launch_minimize_multi<KK, GC_XRI1>(lead, a0, dsub, cnt, items_total);                                       \
        else if (gc == GC_XRI) rc = launch_minimize_multi<KK, GC_XRI>(lead, a0, dsub, cnt, items_total);                                         \
        else if (gc == GC_XGEN) rc = launch_minimize_multi<KK, GC_XGEN>(lead, a0, dsub, cnt, items_total);                                       \
        else rc = launch_minimize_multi<KK, GC_DENSE>(lead, a0, dsub, cnt, items_total);                                                         \
    } break;
        HIP_TRY(hipEventRecord(lead->ev_a[k], lead->stream));
        for (int lo = 0; lo < n;) {
            const int gc = cls[(size_t)order[(size_t)lo]];
            int hi = lo;
            while (hi < n && cls[(size_t)order[(size_t)hi]] == gc) ++hi;
            const int cnt = hi - lo;
            const int64_t items_total = (int64_t)cnt * N * prm->restarts;
            switch (k) {
                SLAM_MQ_CASE(1)
                SLAM_MQ_CASE(2)
                SLAM_MQ_CASE(3)
                default: return fail(SLAM_ERR_UNSUPPORTED, "slam_decompose_multi runs spans 1..3");
            }
            if (rc) return rc;
            lead->stats.kernel_launches += 1;
            lo = hi;
        }
        HIP_TRY(hipEventRecord(lead->ev_b[k], lead->stream));
#undef SLAM_MQ_CASE
        hipLaunchKernelGGL(stage_epilogue_multi_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)n), dim3(256), 0, lead->stream,
                           reinterpret_cast<const EpilogueArgs*>(d + (size_t)n * sz_ma));
        HIP_TRY(hipGetLastError());
        gs += k;
        if (k < k_max)
            for (int i = 0; i < n; ++i) {
                d_active[(size_t)i] = nxt[(size_t)i]->as<int32_t>();
                DevBuf* t = cur[(size_t)i]; cur[(size_t)i] = nxt[(size_t)i]; nxt[(size_t)i] = t;
            }
    }
    HIP_TRY(hipEventRecord(lead->ev_t1, lead->stream));
    for (int i = 0; i < n; ++i)
        HIP_TRY(hipMemcpyAsync(cs[i]->h_ctl, cs[i]->counters.p, sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), hipMemcpyDeviceToHost, lead->stream));
    HIP_TRY(hipEventRecord(lead->ev_done, lead->stream));
    HIP_TRY(hipEventSynchronize(lead->ev_done));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, lead->ev_t0, lead->ev_t1));
    lead->stats.total_ms = ms;
    // statistics: every context gets the evaluations / items of its own queues; the kernels' time goes to the leading context
    for (int i = 0; i < n; ++i) {
        slam_ctx* c = cs[i];
        for (int k = k_min; k <= k_max; ++k) {
            if (c->h_ctl[k].n_active <= 0) continue;
            c->stats.evals[k] += (int64_t)c->h_ctl[k].evals;
            c->stats.evals_accepted[k] += (int64_t)c->h_ctl[k].evals_accepted;
            c->stats.evals_preempted[k] += (int64_t)c->h_ctl[k].evals_preempted;
            c->stats.wave_rounds[k] += (int64_t)c->h_ctl[k].rounds;
            c->stats.items[k] += (int64_t)c->h_ctl[k].n_active * prm->restarts;
        }
    }
    for (int k = k_min; k <= k_max; ++k) {
        float kms = 0.f;
        HIP_TRY(hipEventElapsedTime(&kms, lead->ev_a[k], lead->ev_b[k]));
        lead->stats.kernel_ms += kms;
        lead->stats.kernel_ms_span[k] += kms;
    }
    return SLAM_OK;
}

int decompose_multi_impl(slam_ctx** cs, int n, int64_t first, int64_t count, int k_min, int k_max, const int32_t* gate_seqs,
                         const slam_opt_params* prm, double success_threshold) {
    const int rc = decompose_multi_body(cs, n, first, count, k_min, k_max, gate_seqs, prm, success_threshold);
    if (rc != SLAM_OK && cs)
        for (int i = 0; i < n; ++i)
            if (cs[i]) (void)drained(cs[i], rc);  // (the work sits on ctxs[0]'s stream; every context forgets its cached gate slots)
    return rc;
}

}  // namespace

extern "C" {

const char* slam_last_error(void) { return g_err.c_str(); }

const char* slam_version(void) { return "slamhip 0.4.0 (gfx950)"; }

int slam_abi_version(void) { return SLAM_ABI_VERSION; }

int slam_device_count(int* count) {
    if (!count) return fail(SLAM_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(SLAM_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *count = n;
    return SLAM_OK;
}

int slam_ctx_create(int device, slam_ctx** out) {
    if (!out) return fail(SLAM_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(SLAM_ERR_INVALID, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    if (const char* e = std::getenv("SLAM_HOST_WAIT")) {
        // tuning knob: how the HIP runtime's own waits behave on this device (block / yield / spin)
        const unsigned fl = e[0] == 'b' ? hipDeviceScheduleBlockingSync : (e[0] == 'y' ? hipDeviceScheduleYield : hipDeviceScheduleSpin);
        if (hipSetDeviceFlags(fl) != hipSuccess) (void)hipGetLastError();
    }
    slam_ctx* c = new (std::nothrow) slam_ctx();
    if (!c) return fail(SLAM_ERR_NOMEM, "out of host memory");
    c->device = device;
    if (const char* e = std::getenv("SLAM_RESERVE_WAVES")) c->reserve_waves = std::atoi(e) > 0 ? std::atoi(e) : 0;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int k = 0; k <= SLAM_MAX_SPAN_EVAL && e == hipSuccess; ++k) {
        e = hipEventCreate(&c->ev_a[k]);
        if (e == hipSuccess) e = hipEventCreate(&c->ev_b[k]);
    }
    if (e == hipSuccess) e = hipEventCreate(&c->ev_t0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_t1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventBlockingSync | hipEventDisableTiming);

    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->h_ctl), sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), hipHostMallocDefault);
    if (e == hipSuccess) e = c->counters.reserve(sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2));
    if (e == hipSuccess) e = c->span_gates.reserve((size_t)64 * SLAM_MAX_SPAN_EVAL * 32 * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->h_gates), (size_t)64 * SLAM_MAX_SPAN_EVAL * 32 * sizeof(double), hipHostMallocDefault);
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device);
        if (e == hipSuccess) c->compute_units = prop.multiProcessorCount;
    }
    if (e != hipSuccess) {
        delete c;
        return fail(SLAM_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return SLAM_OK;
}

int slam_ctx_destroy(slam_ctx* ctx) {
    if (!ctx) return SLAM_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    delete ctx;
    return SLAM_OK;
}

int slam_ctx_device_info(slam_ctx* ctx, char* name, int name_len, int* compute_units, int* clock_khz) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (clock_khz) *clock_khz = prop.clockRate;
    return SLAM_OK;
}

int slam_set_targets(slam_ctx* ctx, const double* targets, int64_t n_targets) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    if (!targets || n_targets <= 0) return fail(SLAM_ERR_INVALID, "targets must be non-empty");
    if (n_targets > 0x7fffffffLL) return fail(SLAM_ERR_INVALID, "too many targets");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)n_targets * 32 * sizeof(double);
    HIP_TRY(ctx->targets.reserve(bytes));
    HIP_TRY(hipMemcpyAsync(ctx->targets.p, targets, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->n_targets = n_targets;
    ctx->result_nmax = 0;
    ctx->result_filled = 0;
    return SLAM_OK;
}

int slam_set_gates(slam_ctx* ctx, const double* gates, int32_t n_gates) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    if (!gates || n_gates <= 0 || n_gates > SLAM_MAX_GATES)
        return fail(SLAM_ERR_INVALID, "n_gates must be in 1..%d", SLAM_MAX_GATES);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)n_gates * 32 * sizeof(double);
    HIP_TRY(ctx->gates.reserve(bytes));
    HIP_TRY(hipMemcpyAsync(ctx->gates.p, gates, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->gates_host.assign(gates, gates + (size_t)n_gates * 32);
    ctx->n_gates = n_gates;
    ++ctx->gates_version;
    for (auto& st : ctx->staged) st.valid = false;
    return SLAM_OK;
}

static int eval_body(slam_ctx* ctx, int k, const int32_t* gate_seq, const double* x, const int32_t* target_of,
                     int64_t M, double* loss, double* grad, double* unitary, double* weyl, int ndigits) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->n_targets <= 0) return fail(SLAM_ERR_STATE, "no targets: call slam_set_targets first");
    if (ctx->n_gates <= 0) return fail(SLAM_ERR_STATE, "no gates: call slam_set_gates first");
    if (k < 1 || k > SLAM_MAX_SPAN_EVAL) return fail(SLAM_ERR_UNSUPPORTED, "span must be in 1..%d (got %d)", SLAM_MAX_SPAN_EVAL, k);
    if (M < 0) return fail(SLAM_ERR_INVALID, "M < 0");
    if (M == 0) return SLAM_OK;
    if (!x || !target_of || !loss) return fail(SLAM_ERR_INVALID, "x, target_of and loss must be non-NULL");
    int rc = check_gate_seq(ctx, k, gate_seq);
    if (rc) return rc;
    for (int64_t m = 0; m < M; ++m)
        if (target_of[m] < 0 || target_of[m] >= ctx->n_targets)
            return fail(SLAM_ERR_INVALID, "target_of[%lld] = %d outside [0, %lld)", (long long)m, target_of[m], (long long)ctx->n_targets);
    const int n = 6 * (k + 1);
    HIP_TRY(ctx->ev_x.reserve((size_t)M * n * sizeof(double)));
    HIP_TRY(ctx->ev_tof.reserve((size_t)M * sizeof(int32_t)));
    HIP_TRY(ctx->ev_loss.reserve((size_t)M * sizeof(double)));
    if (grad) HIP_TRY(ctx->ev_grad.reserve((size_t)M * n * sizeof(double)));
    if (unitary || weyl) HIP_TRY(ctx->ev_unitary.reserve((size_t)M * 32 * sizeof(double)));
    if (weyl) HIP_TRY(ctx->ev_weyl.reserve((size_t)M * 3 * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(ctx->ev_x.p, x, (size_t)M * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->ev_tof.p, target_of, (size_t)M * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    double* d_loss = ctx->ev_loss.as<double>();
    double* d_grad = grad ? ctx->ev_grad.as<double>() : nullptr;
    double* d_unit = (unitary || weyl) ? ctx->ev_unitary.as<double>() : nullptr;
    const double* d_x = ctx->ev_x.as<double>();
    const int32_t* d_tof = ctx->ev_tof.as<int32_t>();
    const int gc = classify_gates(ctx, k, gate_seq);
#define SLAM_EVAL_CASE(KK)                                                                                  \
    case KK:                                                                                                \
        if (gc == GC_CX) rc = launch_eval<KK, GC_CX>(ctx, gate_seq, d_x, d_tof, M, d_loss, d_grad, d_unit);   \
        else if (gc == GC_XRI1) rc = launch_eval<KK, GC_XRI1>(ctx, gate_seq, d_x, d_tof, M, d_loss, d_grad, d_unit); \
        else if (gc == GC_XRI) rc = launch_eval<KK, GC_XRI>(ctx, gate_seq, d_x, d_tof, M, d_loss, d_grad, d_unit); \
        else if (gc == GC_XGEN) rc = launch_eval<KK, GC_XGEN>(ctx, gate_seq, d_x, d_tof, M, d_loss, d_grad, d_unit); \
        else rc = launch_eval<KK, GC_DENSE>(ctx, gate_seq, d_x, d_tof, M, d_loss, d_grad, d_unit);          \
        break;
    switch (k) {
        SLAM_EVAL_CASE(1)
        SLAM_EVAL_CASE(2)
        SLAM_EVAL_CASE(3)
        SLAM_EVAL_CASE(4)
        default:
            if (k > SLAM_MAX_SPAN_QUAD) { rc = launch_eval_long(ctx, k, gate_seq, d_x, d_tof, M, d_loss, d_grad, d_unit); break; }
            if (gc == GC_CX) rc = launch_eval<5, GC_CX>(ctx, gate_seq, d_x, d_tof, M, d_loss, d_grad, d_unit);
            else if (gc == GC_XRI1) rc = launch_eval<5, GC_XRI1>(ctx, gate_seq, d_x, d_tof, M, d_loss, d_grad, d_unit);
            else if (gc == GC_XRI) rc = launch_eval<5, GC_XRI>(ctx, gate_seq, d_x, d_tof, M, d_loss, d_grad, d_unit);
            else if (gc == GC_XGEN) rc = launch_eval<5, GC_XGEN>(ctx, gate_seq, d_x, d_tof, M, d_loss, d_grad, d_unit);
            else rc = launch_eval<5, GC_DENSE>(ctx, gate_seq, d_x, d_tof, M, d_loss, d_grad, d_unit);
            break;
    }
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(loss, ctx->ev_loss.p, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (grad) HIP_TRY(hipMemcpyAsync(grad, ctx->ev_grad.p, (size_t)M * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (unitary) HIP_TRY(hipMemcpyAsync(unitary, ctx->ev_unitary.p, (size_t)M * 32 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (weyl) {
        // Weyl coordinates of the template unitaries without bringing the unitaries back (optimizer.py:85,103)
        hipLaunchKernelGGL(c1c2c3_kernel, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, ctx->stream, ctx->ev_unitary.as<double>(), M,
                           ndigits, ctx->ev_weyl.as<double>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(weyl, ctx->ev_weyl.p, (size_t)M * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

static int eval_impl(slam_ctx* ctx, int k, const int32_t* gate_seq, const double* x, const int32_t* target_of,
                     int64_t M, double* loss, double* grad, double* unitary, double* weyl = nullptr, int ndigits = 8) {
    return drained(ctx, eval_body(ctx, k, gate_seq, x, target_of, M, loss, grad, unitary, weyl, ndigits));
}

// Weyl coordinates of `count` unitaries that are already in device memory
static int weyl_device(slam_ctx* ctx, const double* d_unitaries, int64_t count, int ndigits, double* out) {
    HIP_TRY(ctx->ev_weyl.reserve((size_t)count * 3 * sizeof(double)));
    hipLaunchKernelGGL(c1c2c3_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, ctx->stream, d_unitaries, count, ndigits,
                       ctx->ev_weyl.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, ctx->ev_weyl.p, (size_t)count * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

int slam_eval_loss_grad(slam_ctx* ctx, int k, const int32_t* gate_seq, const double* x, const int32_t* target_of,
                        int64_t M, double* loss, double* grad) {
    return eval_impl(ctx, k, gate_seq, x, target_of, M, loss, grad, nullptr);
}

int slam_eval_unitary(slam_ctx* ctx, int k, const int32_t* gate_seq, const double* x, const int32_t* target_of,
                      int64_t M, double* unitary, double* loss) {
    if (!unitary) return fail(SLAM_ERR_INVALID, "unitary is NULL");
    std::vector<double> tmp;
    if (!loss && M > 0) {
        tmp.resize((size_t)M);
        loss = tmp.data();
    }
    return eval_impl(ctx, k, gate_seq, x, target_of, M, loss, nullptr, unitary);
}

// Per-item results of a single-stage call for the host: the records come over as they are and are unpacked into the
// caller's arrays (any of which may be NULL).  The stream is idle (the caller has waited for the stage).
static int fetch_item_records(slam_ctx* c, int64_t M, double* item_loss, int32_t* item_iters, int32_t* item_status, int32_t* item_evals) {
    if (!(item_loss || item_iters || item_status || item_evals) || M <= 0) return SLAM_OK;
    std::vector<ItemRec> rec((size_t)M);
    HIP_TRY(hipMemcpyAsync(rec.data(), c->item_rec.p, (size_t)M * sizeof(ItemRec), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < M; ++i) {
        if (item_loss) item_loss[i] = rec[(size_t)i].loss;
        if (item_iters) item_iters[i] = rec[(size_t)i].iters;
        if (item_status) item_status[i] = rec[(size_t)i].status;
        if (item_evals) item_evals[i] = rec[(size_t)i].evals;
    }
    return SLAM_OK;
}

static int minimize_stage_body(slam_ctx* ctx, int k, const int32_t* gate_seq, const int32_t* active, int64_t n_active,
                        const double* x0, const slam_opt_params* params, double* best_loss, double* best_x,
                        int32_t* best_restart, double* item_loss, int32_t* item_iters, int32_t* item_status,
                        int32_t* item_evals) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->n_targets <= 0) return fail(SLAM_ERR_STATE, "no targets: call slam_set_targets first");
    if (ctx->n_gates <= 0) return fail(SLAM_ERR_STATE, "no gates: call slam_set_gates first");
    if (k < 1 || k > SLAM_MAX_SPAN_MINIMIZE)
        return fail(SLAM_ERR_UNSUPPORTED, "minimize supports spans 1..%d (got %d)", SLAM_MAX_SPAN_MINIMIZE, k);
    int rc = check_params(params);
    if (rc) return rc;
    rc = check_gate_seq(ctx, k, gate_seq);
    if (rc) return rc;
    if (!active) n_active = ctx->n_targets;
    if (n_active < 0) return fail(SLAM_ERR_INVALID, "n_active < 0");
    if (n_active == 0) return SLAM_OK;
    if (!best_loss || !best_x) return fail(SLAM_ERR_INVALID, "best_loss and best_x must be non-NULL");
    const int n = 6 * (k + 1);
    const int64_t M = n_active * (int64_t)params->restarts;
    const int32_t* d_active = nullptr;
    if (active) {
        for (int64_t s = 0; s < n_active; ++s)
            if (active[s] < 0 || active[s] >= ctx->n_targets)
                return fail(SLAM_ERR_INVALID, "active[%lld] = %d outside [0, %lld)", (long long)s, active[s], (long long)ctx->n_targets);
        HIP_TRY(ctx->active.reserve((size_t)n_active * sizeof(int32_t)));
        HIP_TRY(hipMemcpyAsync(ctx->active.p, active, (size_t)n_active * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        d_active = ctx->active.as<int32_t>();
    }
    const double* d_x0 = nullptr;
    if (x0) {
        // the optimizer kernel's range reduction covers |x| < 2e9; steps are <= 2 rad each
        for (int64_t i = 0; i < M * n; ++i)
            if (!(x0[i] > -1e8 && x0[i] < 1e8))
                return fail(SLAM_ERR_INVALID, "x0[%lld] = %g: explicit seeds must be finite with |x| < 1e8", (long long)i, x0[i]);
        HIP_TRY(ctx->x0.reserve((size_t)M * n * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(ctx->x0.p, x0, (size_t)M * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        d_x0 = ctx->x0.as<double>();
    }
    rc = reserve_stage_buffers(ctx, n_active, k, params);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_t0, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->counters.p, 0, sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), ctx->stream));
    hipLaunchKernelGGL(set_n_active_kernel, dim3(1), dim3(1), 0, ctx->stream, stage_ctl(ctx, k), (int32_t)n_active);
    HIP_TRY(hipGetLastError());
    rc = enqueue_stage(ctx, k, gate_seq, d_active, n_active, d_x0, params, nullptr);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_t1, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->h_ctl, ctx->counters.p, sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_done, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev_done));  // sleep until the stage is done; the copies below then find an idle stream
    HIP_TRY(hipMemcpyAsync(best_loss, ctx->stage_loss.p, (size_t)n_active * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(best_x, ctx->stage_x.p, (size_t)n_active * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (best_restart) HIP_TRY(hipMemcpyAsync(best_restart, ctx->stage_restart.p, (size_t)n_active * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    rc = fetch_item_records(ctx, M, item_loss, item_iters, item_status, item_evals);
    if (rc) return rc;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_t0, ctx->ev_t1));
    ctx->stats.total_ms = ms;
    return collect_stats(ctx, k, k, ctx->h_ctl, params->restarts);
}

int slam_minimize_stage(slam_ctx* ctx, int k, const int32_t* gate_seq, const int32_t* active, int64_t n_active,
                        const double* x0, const slam_opt_params* params, double* best_loss, double* best_x,
                        int32_t* best_restart, double* item_loss, int32_t* item_iters, int32_t* item_status,
                        int32_t* item_evals) {
    return drained(ctx, minimize_stage_body(ctx, k, gate_seq, active, n_active, x0, params, best_loss, best_x, best_restart,
                                            item_loss, item_iters, item_status, item_evals));
}

int slam_decompose_resident(slam_ctx* ctx, int k_min, int k_max, const int32_t* gate_seqs,
                            const slam_opt_params* params, double success_threshold) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    return decompose_impl(ctx, 0, ctx->n_targets, k_min, k_max, gate_seqs, params, success_threshold);
}

int slam_decompose_range(slam_ctx* ctx, int64_t first, int64_t count, int k_min, int k_max,
                         const int32_t* gate_seqs, const slam_opt_params* params, double success_threshold) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    return decompose_impl(ctx, first, count, k_min, k_max, gate_seqs, params, success_threshold);
}

int slam_decompose_list(slam_ctx* ctx, const int32_t* targets, int64_t count, int k_min, int k_max, int k_layout,
                        const int32_t* gate_seqs, const slam_opt_params* params, double success_threshold) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    if (!targets) return fail(SLAM_ERR_INVALID, "targets is NULL");
    return decompose_impl(ctx, 0, count, k_min, k_max, gate_seqs, params, success_threshold, targets, k_layout);
}

// -----------------------------------------------------------------------------------------------------------------------
// slam_decompose_predicted: CircuitTemplate(use_polytopes=True) for a window of RESIDENT targets without a host step in between --
// the coverage lookup (span_predict_kernel), the per-size target lists (span_bucket_kernel) and ONE span loop in which the targets of
// size k join the loop at stage k.  carry = 0: a target runs at its own size only (exact regions: basis.py:95-100 returns
// range(k, k + 1)); carry = 1: targets that miss the threshold go on to the next size (regions widened by `tol`, lower bounds).
// Replaces the host's np.nonzero + one slam_decompose_list per size (round 4).
// -----------------------------------------------------------------------------------------------------------------------
int decompose_predicted_body(slam_ctx* c, int64_t first, int64_t count, int k_max, const double* point, const double* bounds, double tol,
                             int carry, const int32_t* gate_seqs, const slam_opt_params* prm, double success_threshold,
                             int64_t* n_local, int64_t* n_unreachable) {
    HIP_TRY(hipSetDevice(c->device));
    if (c->n_targets <= 0) return fail(SLAM_ERR_STATE, "no targets");
    if (c->n_gates <= 0) return fail(SLAM_ERR_STATE, "no gates");
    if (first < 0 || count <= 0 || first + count > c->n_targets) return fail(SLAM_ERR_INVALID, "target window outside the resident batch");
    if (k_max < 1 || k_max > SLAM_MAX_SPAN_MINIMIZE) return fail(SLAM_ERR_INVALID, "k_max must be 1..%d (got %d)", SLAM_MAX_SPAN_MINIMIZE, k_max);
    if (!point || (k_max > 1 && !bounds) || !gate_seqs) return fail(SLAM_ERR_INVALID, "point / bounds / gate_seqs is NULL");
    int rc = check_params(prm);
    if (rc) return rc;
    {
        const int32_t* gs = gate_seqs;
        for (int k = 1; k <= k_max; ++k) {
            rc = check_gate_seq(c, k, gs);
            if (rc) return rc;
            gs += k;
        }
    }
    if (c->result_nmax != 0 && c->result_nmax != 6 * (k_max + 1) && !(first == 0 && count == c->n_targets))
        return fail(SLAM_ERR_STATE, "resident results were produced with a different k_max");
    rc = ensure_results(c, k_max);
    if (rc) return rc;
    const int64_t N = count;
    HIP_TRY(c->active.reserve(N * sizeof(int32_t)));
    HIP_TRY(c->active2.reserve(N * sizeof(int32_t)));
    HIP_TRY(c->bucket_lists.reserve((size_t)k_max * N * sizeof(int32_t)));
    HIP_TRY(c->bucket_counts.reserve((size_t)(SLAM_MAX_SPAN_EVAL + 2) * sizeof(int32_t)));
    HIP_TRY(c->ev_weyl.reserve((size_t)N * sizeof(int32_t)));
    if (!c->h_bucket_counts) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_bucket_counts), (SLAM_MAX_SPAN_EVAL + 2) * sizeof(int32_t), hipHostMallocDefault));
    rc = reserve_stage_buffers(c, N, k_max, prm);
    if (rc) return rc;
    SpanRegions r{};
    r.k_max = k_max;
    r.tol = tol;
    for (int j = 0; j < 4; ++j) r.point[j] = point[j];
    for (int k = 2; k <= k_max; ++k)
        for (int p = 0; p < kSpanPatterns; ++p) r.bounds[k - 1][p] = bounds[(size_t)(k - 1) * kSpanPatterns + p];
    HIP_TRY(hipEventRecord(c->ev_t0, c->stream));
    const int n_words = (int)(sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2) / 8);
    hipLaunchKernelGGL(clear_ctl_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, c->stream, c->counters.as<StageCtl>(), n_words,
                       c->bucket_counts.as<int32_t>(), SLAM_MAX_SPAN_EVAL + 2);
    HIP_TRY(hipGetLastError());
    int32_t* d_spans = c->ev_weyl.as<int32_t>();
    hipLaunchKernelGGL(span_predict_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, c->stream, c->targets.as<double>() + first * 32, N, r, d_spans);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(span_bucket_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, d_spans, first, N, (int32_t)k_max,
                       c->bucket_lists.as<int32_t>(), c->bucket_counts.as<int32_t>(), c->best_loss.as<double>(), c->best_cycles.as<int32_t>(),
                       c->span_loss.as<double>());
    HIP_TRY(hipGetLastError());
    DevBuf* cur = &c->active;
    DevBuf* nxt = &c->active2;
    const int32_t* gs = gate_seqs;
    for (int k = 1; k <= k_max; ++k) {
        // stage k's list: what the previous stage carried over (carry) + the targets of size k
        hipLaunchKernelGGL(stage_append_kernel, dim3(1), dim3(1024), 0, c->stream, c->bucket_lists.as<int32_t>() + (size_t)(k - 1) * N,
                           c->bucket_counts.as<int32_t>() + (k - 1), stage_ctl(c, k), cur->as<int32_t>());
        HIP_TRY(hipGetLastError());
        SpanLoopStep step{success_threshold, false, carry && k < k_max, success_threshold, nxt->as<int32_t>()};
        rc = enqueue_stage(c, k, gs, cur->as<int32_t>(), N, nullptr, prm, &step);
        if (rc) return rc;
        gs += k;
        DevBuf* t = cur; cur = nxt; nxt = t;
    }
    HIP_TRY(hipEventRecord(c->ev_t1, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_ctl, c->counters.p, sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_bucket_counts, c->bucket_counts.p, (SLAM_MAX_SPAN_EVAL + 2) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->ev_done, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev_done));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
    c->stats.total_ms = ms;
    if (n_local) *n_local = c->h_bucket_counts[k_max];
    if (n_unreachable) *n_unreachable = c->h_bucket_counts[k_max + 1];
    return collect_stats(c, 1, k_max, c->h_ctl, prm->restarts);
}

int slam_decompose_predicted(slam_ctx* ctx, int64_t first, int64_t count, int k_max, const double* point, const double* bounds, double tol,
                             int carry, const int32_t* gate_seqs, const slam_opt_params* params, double success_threshold,
                             int64_t* n_local, int64_t* n_unreachable) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    return drained(ctx, decompose_predicted_body(ctx, first, count, k_max, point, bounds, tol, carry, gate_seqs, params, success_threshold, n_local, n_unreachable));
}

int slam_fetch_results_range(slam_ctx* ctx, int k_max, int64_t first, int64_t count, double* best_loss,
                             double* best_x, int32_t* best_cycles) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    FetchReq fr{best_loss, best_x, best_cycles};
    int rc = enqueue_fetch(ctx, k_max, first, count, fr);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_done, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev_done));
    finish_fetch(ctx, fr);
    return SLAM_OK;
}

int slam_decompose_range_fetch(slam_ctx* ctx, int64_t first, int64_t count, int k_min, int k_max, const int32_t* gate_seqs,
                               const slam_opt_params* params, double success_threshold, double* best_loss, double* best_x,
                               int32_t* best_cycles) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    FetchReq fr{best_loss, best_x, best_cycles};
    return decompose_impl(ctx, first, count, k_min, k_max, gate_seqs, params, success_threshold, nullptr, 0, &fr);
}

int slam_decompose_multi(slam_ctx** ctxs, int32_t n_ctx, int64_t first, int64_t count, int k_min, int k_max, const int32_t* gate_seqs,
                         const slam_opt_params* params, double success_threshold) {
    return decompose_multi_impl(ctxs, n_ctx, first, count, k_min, k_max, gate_seqs, params, success_threshold);
}

int slam_fetch_results(slam_ctx* ctx, int k_max, double* best_loss, double* best_x, int32_t* best_cycles) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    return slam_fetch_results_range(ctx, k_max, 0, ctx->n_targets, best_loss, best_x, best_cycles);
}

int slam_decompose(slam_ctx* ctx, int k_min, int k_max, const int32_t* gate_seqs, const slam_opt_params* params,
                   double success_threshold, double* best_loss, double* best_x, int32_t* best_cycles) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    int rc = decompose_impl(ctx, 0, ctx->n_targets, k_min, k_max, gate_seqs, params, success_threshold);
    if (rc) return rc;
    return slam_fetch_results(ctx, k_max, best_loss, best_x, best_cycles);
}

int slam_c1c2c3(slam_ctx* ctx, const double* unitaries, int64_t count, int ndigits, double* out) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    if (count < 0) return fail(SLAM_ERR_INVALID, "count < 0");
    if (count == 0) return SLAM_OK;
    if (!unitaries || !out) return fail(SLAM_ERR_INVALID, "unitaries and out must be non-NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->ev_unitary.reserve((size_t)count * 32 * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(ctx->ev_unitary.p, unitaries, (size_t)count * 32 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return weyl_device(ctx, ctx->ev_unitary.as<double>(), count, ndigits, out);
}

int slam_targets_c1c2c3(slam_ctx* ctx, int64_t first, int64_t count, int ndigits, double* out) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    if (first < 0 || count < 0 || first + count > ctx->n_targets)
        return fail(SLAM_ERR_INVALID, "target window outside the resident batch");
    if (count == 0) return SLAM_OK;
    if (!out) return fail(SLAM_ERR_INVALID, "out is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    return weyl_device(ctx, ctx->targets.as<double>() + first * 32, count, ndigits, out);
}

int slam_predict_spans(slam_ctx* ctx, int64_t first, int64_t count, int k_max, const double* point, const double* bounds, double tol,
                       int32_t* spans_out) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    if (first < 0 || count < 0 || first + count > ctx->n_targets)
        return fail(SLAM_ERR_INVALID, "target window outside the resident batch");
    if (k_max < 1 || k_max > SLAM_MAX_SPAN_EVAL) return fail(SLAM_ERR_INVALID, "k_max must be 1..%d (got %d)", SLAM_MAX_SPAN_EVAL, k_max);
    if (!point || (k_max > 1 && !bounds)) return fail(SLAM_ERR_INVALID, "point / bounds is NULL");
    if (count == 0) return SLAM_OK;
    if (!spans_out) return fail(SLAM_ERR_INVALID, "spans_out is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    static_assert(sizeof(SpanRegions{}.bounds) / sizeof(SpanRegions{}.bounds[0]) == SLAM_MAX_SPAN_EVAL, "SpanRegions::bounds holds SLAM_MAX_SPAN_EVAL prefixes");
    SpanRegions r{};
    r.k_max = k_max;
    r.tol = tol;
    for (int j = 0; j < 4; ++j) r.point[j] = point[j];
    for (int k = 2; k <= k_max; ++k)
        for (int p = 0; p < kSpanPatterns; ++p) r.bounds[k - 1][p] = bounds[(size_t)(k - 1) * kSpanPatterns + p];
    HIP_TRY(ctx->ev_weyl.reserve((size_t)count * sizeof(int32_t)));
    hipLaunchKernelGGL(span_predict_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, ctx->stream,
                       ctx->targets.as<double>() + first * 32, count, r, ctx->ev_weyl.as<int32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(spans_out, ctx->ev_weyl.p, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

int slam_eval_c1c2c3(slam_ctx* ctx, int k, const int32_t* gate_seq, const double* x, int64_t M, int ndigits, double* out) {
    if (!out && M > 0) return fail(SLAM_ERR_INVALID, "out is NULL");
    std::vector<int32_t> tof((size_t)(M > 0 ? M : 0), 0);  // the loss is not wanted: any resident target will do
    std::vector<double> loss((size_t)(M > 0 ? M : 0));
    return eval_impl(ctx, k, gate_seq, x, tof.data(), M, loss.data(), nullptr, nullptr, out, ndigits);
}

int slam_sample_haar(slam_ctx* ctx, uint64_t seed, int64_t first_index, int64_t n_targets) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    if (n_targets <= 0 || n_targets > 0x7fffffffLL) return fail(SLAM_ERR_INVALID, "n_targets must be in 1..2^31-1");
    if (first_index < 0) return fail(SLAM_ERR_INVALID, "first_index < 0");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->targets.reserve((size_t)n_targets * 32 * sizeof(double)));
    hipLaunchKernelGGL(haar_targets_kernel, dim3((unsigned)((n_targets + 127) / 128)), dim3(128), 0, ctx->stream,
                       ctx->targets.as<double>(), first_index, n_targets, seed);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->n_targets = n_targets;
    ctx->result_nmax = 0;
    ctx->result_filled = 0;
    return SLAM_OK;
}

int slam_get_targets(slam_ctx* ctx, int64_t first, int64_t count, double* out) {
    if (!ctx || !out) return fail(SLAM_ERR_INVALID, "NULL argument");
    if (first < 0 || count < 0 || first + count > ctx->n_targets)
        return fail(SLAM_ERR_INVALID, "target window outside the resident batch");
    if (count == 0) return SLAM_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(out, ctx->targets.as<double>() + first * 32, (size_t)count * 32 * sizeof(double),
                           hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

int slam_fetch_span_losses(slam_ctx* ctx, int64_t first, int64_t count, double* out) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    if (ctx->result_nmax == 0 || ctx->n_targets <= 0) return fail(SLAM_ERR_STATE, "no resident results");
    if (first < 0 || count < 0 || first + count > ctx->n_targets) return fail(SLAM_ERR_INVALID, "target window outside the resident batch");
    if (count == 0) return SLAM_OK;
    if (!out) return fail(SLAM_ERR_INVALID, "out is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(out, ctx->span_loss.as<double>() + first * kSpanLossStride, (size_t)count * kSpanLossStride * sizeof(double),
                           hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

int slam_minimize_stage_trace(slam_ctx* ctx, int k, const int32_t* gate_seq, const int32_t* active, int64_t n_active,
                              const double* x0, const slam_opt_params* params, double exit_loss, int32_t trace_cap,
                              double* best_loss, double* best_x, int32_t* best_restart, double* item_loss,
                              int32_t* item_iters, int32_t* item_status, double* trace_loss, double* trace_x) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    if (!params) return fail(SLAM_ERR_INVALID, "params is NULL");
    if (trace_cap <= 0 || !trace_loss || !trace_x) return fail(SLAM_ERR_INVALID, "trace buffers and trace_cap > 0 are required");
    if (k < 1 || k > SLAM_MAX_SPAN_MINIMIZE) return fail(SLAM_ERR_UNSUPPORTED, "per-iteration traces are recorded for spans 1..%d (got %d)", SLAM_MAX_SPAN_MINIMIZE, k);
    if (!active) n_active = ctx->n_targets;
    if (n_active <= 0 || params->restarts <= 0) return fail(SLAM_ERR_INVALID, "nothing to trace");
    const int n = 6 * (k + 1);
    const int64_t M = n_active * (int64_t)params->restarts;
    const size_t rows = (size_t)M * (size_t)trace_cap;
    if (rows * (size_t)(n + 1) * sizeof(double) > ((size_t)4 << 30))
        return fail(SLAM_ERR_INVALID, "trace of %lld items x %d iterations exceeds 4 GiB: trace fewer targets at a time", (long long)M, trace_cap);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->trace_loss.reserve(rows * sizeof(double)));
    HIP_TRY(ctx->trace_x.reserve(rows * n * sizeof(double)));
    // rows that no iteration reaches read as NaN
    HIP_TRY(hipMemsetAsync(ctx->trace_loss.p, 0xFF, rows * sizeof(double), ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->trace_x.p, 0xFF, rows * n * sizeof(double), ctx->stream));
    ctx->trace_cap = trace_cap;
    ctx->stage_exit_loss = exit_loss;
    int rc = slam_minimize_stage(ctx, k, gate_seq, active, n_active, x0, params, best_loss, best_x, best_restart, item_loss, item_iters,
                                 item_status, nullptr);
    ctx->trace_cap = 0;
    ctx->stage_exit_loss = -1.0;
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(trace_loss, ctx->trace_loss.p, rows * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(trace_x, ctx->trace_x.p, rows * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

int slam_set_cost(slam_ctx* ctx, int cost) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    if (cost != SLAM_COST_BASIC && cost != SLAM_COST_SQUARE)
        return fail(SLAM_ERR_INVALID, "Unrecognized Cost Function (%d)", cost);
    ctx->cost_kind = cost;
    return SLAM_OK;
}

int slam_synchronize(slam_ctx* ctx) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

int slam_host_alloc(size_t bytes, void** ptr) {
    if (!ptr) return fail(SLAM_ERR_INVALID, "ptr is NULL");
    *ptr = nullptr;
    if (bytes == 0) return fail(SLAM_ERR_INVALID, "bytes == 0");
    HIP_TRY(hipHostMalloc(ptr, bytes, hipHostMallocPortable));
    return SLAM_OK;
}

int slam_host_free(void* ptr) {
    if (!ptr) return SLAM_OK;
    HIP_TRY(hipHostFree(ptr));
    return SLAM_OK;
}

int slam_get_stats(slam_ctx* ctx, slam_stats* out) {
    if (!ctx || !out) return fail(SLAM_ERR_INVALID, "NULL argument");
    *out = ctx->stats;
    return SLAM_OK;
}

int slam_reset_stats(slam_ctx* ctx) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    ctx->stats = slam_stats{};
    return SLAM_OK;
}

int slam_ctx_device(slam_ctx* ctx, int* device) {
    if (!ctx || !device) return fail(SLAM_ERR_INVALID, "NULL argument");
    *device = ctx->device;
    return SLAM_OK;
}

int slam_best_loss_device_ptr(slam_ctx* ctx, void** ptr, int64_t* n) {
    if (!ctx || !ptr || !n) return fail(SLAM_ERR_INVALID, "NULL argument");
    if (!ctx->best_loss.p || ctx->n_targets <= 0) return fail(SLAM_ERR_STATE, "no resident results");
    *ptr = ctx->best_loss.p;
    *n = ctx->n_targets;
    return SLAM_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// templates with parametrised 2Q gates (CircuitTemplateV2, src/slam/basisv2.py:27-299): slam_v2.hpp
// ------------------------------------------------------------------------------------------------------------------
namespace {

template <int K, int QN>
constexpr size_t v2_lds_bytes() { return sizeof(double) * CfgV2<K, QN>::LDS_DOUBLES; }

int v2_stage_maps(slam_ctx* c, int k, const int32_t* gate_seq, const V2GateMap** d_out) {
    if (c->v2_gates_host.empty()) return fail(SLAM_ERR_STATE, "no parametrised gates: call slam_v2_set_gates first");
    if (!gate_seq) return fail(SLAM_ERR_INVALID, "gate_seq is NULL");
    V2GateMap tmp[SLAM_MAX_SPAN_EVAL];
    for (int j = 0; j < k; ++j) {
        if (gate_seq[j] < 0 || gate_seq[j] >= (int)c->v2_gates_host.size())
            return fail(SLAM_ERR_INVALID, "gate_seq[%d] = %d outside the parametrised gate table (%d gates)", j, gate_seq[j], (int)c->v2_gates_host.size());
        tmp[j] = c->v2_gates_host[(size_t)gate_seq[j]];
    }
    HIP_TRY(c->v2_maps.reserve(sizeof(V2GateMap) * SLAM_MAX_SPAN_EVAL));
    HIP_TRY(hipMemcpyAsync(c->v2_maps.p, tmp, sizeof(V2GateMap) * (size_t)k, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // tmp is a stack buffer
    *d_out = c->v2_maps.as<V2GateMap>();
    return SLAM_OK;
}

template <int K, int QN>
int v2_launch_eval(slam_ctx* c, const V2GateMap* d_maps, const double* d_x, const int32_t* d_tof, int64_t M, double* d_loss, double* d_grad,
                   double* d_unitary) {
    const size_t lds = v2_lds_bytes<K, QN>();
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&eval_v2_kernel<K, QN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    EvalV2Args<K, QN> a{};
    a.targets = c->targets.as<double>();
    a.x = d_x;
    a.target_of = d_tof;
    a.n_items = M;
    a.loss = d_loss;
    a.grad = d_grad;
    a.unitary = d_unitary;
    a.cost_kind = c->cost_kind;
    a.maps = d_maps;
    hipLaunchKernelGGL((eval_v2_kernel<K, QN>), dim3((unsigned)((M + kQuadsPerWave - 1) / kQuadsPerWave)), dim3(kWave), lds, c->stream, a);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

struct V2Stage {
    bool bounded;        // some parameter has a finite bound (or is fixed): projected steps; else plain BFGS
    bool riswap_like;    // every gate of the span: only the angle a moves, phi_c = b = 0
    bool count_on_device = false;  // the kernel reads the stage's target count from its control block (n_active = upper bound)
    int k = 0;
    double exit_loss;
    const V2GateMap* d_maps;
    const int32_t* d_active;
    int32_t n_active;
    const double* d_x0;
    const double* d_bounds;  // init_lo | init_hi | bound_lo | bound_hi, n each
    const slam_opt_params* prm;
};

constexpr int kV2HmemWavesPerCu = 8;  // wavefronts per CU whose inverse Hessian lives in device memory (v2_hmem slices)

template <int K, int QN, int GQ, bool FREE>
int v2_launch_minimize_gq(slam_ctx* c, const V2Stage& sgt) {
    const size_t lds = v2_lds_bytes<K, QN>();
    // attribute + occupancy once per context and instantiation (both were runtime calls inside every stage launch of the
    // span loop's chain)
    int& per_cu = c->v2_per_cu[K][QN == 1 ? 0 : (QN == 2 ? 1 : 2)][GQ][FREE ? 1 : 0];
    if (per_cu == 0) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&minimize_v2_kernel<K, QN, GQ, FREE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int v = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, reinterpret_cast<const void*>(&minimize_v2_kernel<K, QN, GQ, FREE>), kWave, lds));
        per_cu = v < 1 ? 1 : v;
    }
    constexpr int n = CfgV2<K, QN>::N;
    MinimizeV2Args<K, QN> a{};
    a.targets = c->targets.as<double>();
    a.active = sgt.d_active;
    a.n_active = sgt.count_on_device ? -1 : sgt.n_active;
    a.restarts = sgt.prm->restarts;
    a.x0 = sgt.d_x0;
    a.init_lo = sgt.d_bounds;
    a.init_hi = sgt.d_bounds + n;
    a.bound_lo = sgt.d_bounds + 2 * n;
    a.bound_hi = sgt.d_bounds + 3 * n;
    a.maxiter = sgt.prm->maxiter;
    a.gtol = sgt.prm->gtol;
    a.stop_loss = sgt.prm->stop_loss;
    a.gtol_far = sgt.prm->gtol_far;
    a.far_loss = sgt.prm->far_loss;
    a.exit_loss = sgt.exit_loss;
    a.flags = sgt.prm->flags & (SLAM_FLAG_EARLY_EXIT | SLAM_FLAG_ORDERED);
    a.seed = sgt.prm->seed;
    a.target_base = sgt.prm->target_base;
    a.cost_kind = c->cost_kind;
    a.maps = sgt.d_maps;
    a.solved = c->solved.as<int32_t>();
    a.item_rec = c->item_rec.as<ItemRec>();
    a.item_x = c->item_x.as<double>();
    a.ctl = stage_ctl(c, K);
    a.trace_cap = c->trace_cap;
    a.trace_loss = c->trace_cap > 0 ? c->trace_loss.as<double>() : nullptr;
    a.trace_x = c->trace_cap > 0 ? c->trace_x.as<double>() : nullptr;
    a.bounded = sgt.bounded ? 1 : 0;
    if (c->v2_cons_n[K] > 0) {
        if (c->v2_cons_n[K] != n) return fail(SLAM_ERR_STATE, "the cost constraint of span %d was set for %d parameters, the template has %d (set the gates first)", K, c->v2_cons_n[K], n);
        if (FREE) return fail(SLAM_ERR_STATE, "internal: constrained stage dispatched to the unbounded kernel");
        a.cons_w = c->v2_cons_w[K].as<double>();
        // results are feasible: the multiplier loop ends with c <= tol against a right-hand side lowered by tol
        a.cons_tol = 1e-8 * (1.0 + std::fabs(c->v2_cons_max[K]));
        a.cons_max = c->v2_cons_max[K] - a.cons_tol;
        a.cons_rho = c->v2_cons_rho[K];
        a.bounded = 1;
    }
    // persistent wavefronts: never more than can be resident; every quad pulls items from the stage's queue
    const int64_t M = (int64_t)sgt.n_active * sgt.prm->restarts;
    int64_t blocks = (M + kQuadsPerWave - 1) / kQuadsPerWave;
    // (in-memory metric: at most kV2HmemWavesPerCu wavefronts per CU -- the bound v2_decompose_body sizes v2_hmem with up front, so the
    // per-stage reserve below can never grow the buffer in the middle of a chain: ADVICE r4)
    const int64_t cap = (int64_t)(v2_h_in_memory<K, QN>() && per_cu > kV2HmemWavesPerCu ? kV2HmemWavesPerCu : per_cu) * c->compute_units;
    if (blocks > cap) blocks = cap;
    if constexpr (v2_h_in_memory<K, QN>()) {
        HIP_TRY(c->v2_hmem.reserve((size_t)blocks * v2_h_floats_per_wave<K, QN>() * sizeof(float)));
        a.hmem = c->v2_hmem.as<float>();
    }
    HIP_TRY(hipEventRecord(c->ev_a[K], c->stream));
    hipLaunchKernelGGL((minimize_v2_kernel<K, QN, GQ, FREE>), dim3((unsigned)blocks), dim3(kWave), lds, c->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_b[K], c->stream));
    return SLAM_OK;
}

// gate sub-class of the stage (slam_v2.hpp): one parameter per gate that only moves the angle a, phi_c = b = 0 (RiSwapGate)
template <int K, int QN>
int v2_launch_minimize(slam_ctx* c, const V2Stage& sgt) {
    if constexpr (QN == 1 && K <= 3) {
        if (sgt.riswap_like && !sgt.bounded && c->v2_cons_n[K] == 0) return v2_launch_minimize_gq<K, QN, 1, true>(c, sgt);  // RiSwapGate class, plain BFGS
        if (sgt.riswap_like) return v2_launch_minimize_gq<K, QN, 1, false>(c, sgt);
    }
    if (!sgt.bounded && c->v2_cons_n[K] == 0) return v2_launch_minimize_gq<K, QN, 0, true>(c, sgt);  // general gates, plain BFGS
    return v2_launch_minimize_gq<K, QN, 0, false>(c, sgt);
}

// spans 1..3 with 1, 2 or 4 parameters per gate; spans 4 and 5 (the reference's default maximum_span_guess = 5,
// basisv2.py:35) where the packed inverse Hessian still fits one wavefront's 512 registers: n <= 41 parameters
#define SLAM_V2_DISPATCH(FN, ...)                                                                             \
    do {                                                                                                      \
        const int key = k * 10 + c->v2_qn;                                                                    \
        switch (key) {                                                                                        \
            case 11: rc = FN<1, 1>(__VA_ARGS__); break;                                                       \
            case 12: rc = FN<1, 2>(__VA_ARGS__); break;                                                       \
            case 14: rc = FN<1, 4>(__VA_ARGS__); break;                                                       \
            case 21: rc = FN<2, 1>(__VA_ARGS__); break;                                                       \
            case 22: rc = FN<2, 2>(__VA_ARGS__); break;                                                       \
            case 24: rc = FN<2, 4>(__VA_ARGS__); break;                                                       \
            case 31: rc = FN<3, 1>(__VA_ARGS__); break;                                                       \
            case 32: rc = FN<3, 2>(__VA_ARGS__); break;                                                       \
            case 34: rc = FN<3, 4>(__VA_ARGS__); break;                                                       \
            case 41: rc = FN<4, 1>(__VA_ARGS__); break;                                                       \
            case 42: rc = FN<4, 2>(__VA_ARGS__); break;                                                       \
            case 51: rc = FN<5, 1>(__VA_ARGS__); break;                                                       \
            default: rc = fail(SLAM_ERR_UNSUPPORTED, "parametrised-gate templates: spans 1..3 with 1, 2 or 4 parameters per gate, span 4 with 1 or 2, span 5 with 1 (got span %d, %d)", \
                               k, c->v2_qn);                                                                  \
        }                                                                                                     \
    } while (0)

int v2_eval_body(slam_ctx* c, int k, const int32_t* gate_seq, const double* x, const int32_t* target_of, int64_t M, double* loss,
                 double* grad, double* unitary) {
    if (!c) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (c->n_targets <= 0) return fail(SLAM_ERR_STATE, "no targets: call slam_set_targets first");
    if (k < 1 || k > SLAM_V2_MAX_SPAN) return fail(SLAM_ERR_UNSUPPORTED, "parametrised-gate templates support spans 1..%d (got %d)", SLAM_V2_MAX_SPAN, k);
    if (M < 0) return fail(SLAM_ERR_INVALID, "M < 0");
    if (M == 0) return SLAM_OK;
    if (!x || !target_of || !loss) return fail(SLAM_ERR_INVALID, "x, target_of and loss must be non-NULL");
    for (int64_t m = 0; m < M; ++m)
        if (target_of[m] < 0 || target_of[m] >= c->n_targets) return fail(SLAM_ERR_INVALID, "target_of[%lld] outside the resident batch", (long long)m);
    const V2GateMap* d_maps = nullptr;
    int rc = v2_stage_maps(c, k, gate_seq, &d_maps);
    if (rc) return rc;
    const int n = 6 * (k + 1) + c->v2_qn * k;
    HIP_TRY(c->ev_x.reserve((size_t)M * n * sizeof(double)));
    HIP_TRY(c->ev_tof.reserve((size_t)M * sizeof(int32_t)));
    HIP_TRY(c->ev_loss.reserve((size_t)M * sizeof(double)));
    if (grad) HIP_TRY(c->ev_grad.reserve((size_t)M * n * sizeof(double)));
    if (unitary) HIP_TRY(c->ev_unitary.reserve((size_t)M * 32 * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(c->ev_x.p, x, (size_t)M * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ev_tof.p, target_of, (size_t)M * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    double* d_grad = grad ? c->ev_grad.as<double>() : nullptr;
    double* d_unit = unitary ? c->ev_unitary.as<double>() : nullptr;
    SLAM_V2_DISPATCH(v2_launch_eval, c, d_maps, c->ev_x.as<double>(), c->ev_tof.as<int32_t>(), M, c->ev_loss.as<double>(), d_grad, d_unit);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(loss, c->ev_loss.p, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (grad) HIP_TRY(hipMemcpyAsync(grad, c->ev_grad.p, (size_t)M * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (unitary) HIP_TRY(hipMemcpyAsync(unitary, c->ev_unitary.p, (size_t)M * 32 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

int v2_minimize_body(slam_ctx* c, int k, const int32_t* gate_seq, const int32_t* active, int64_t n_active, const double* x0,
                     const double* init_lo, const double* init_hi, const double* bound_lo, const double* bound_hi,
                     const slam_opt_params* prm, double exit_loss, double* best_loss, double* best_x, int32_t* best_restart,
                     double* item_loss, int32_t* item_iters, int32_t* item_status, int32_t* item_evals) {
    if (!c) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (c->n_targets <= 0) return fail(SLAM_ERR_STATE, "no targets: call slam_set_targets first");
    if (k < 1 || k > SLAM_V2_MAX_SPAN) return fail(SLAM_ERR_UNSUPPORTED, "parametrised-gate templates support spans 1..%d (got %d)", SLAM_V2_MAX_SPAN, k);
    int rc = check_params(prm);
    if (rc) return rc;
    if (!active) n_active = c->n_targets;
    if (n_active <= 0) return n_active == 0 ? SLAM_OK : fail(SLAM_ERR_INVALID, "n_active < 0");
    if (!best_loss || !best_x) return fail(SLAM_ERR_INVALID, "best_loss and best_x must be non-NULL");
    if (!init_lo || !init_hi) return fail(SLAM_ERR_INVALID, "init_lo and init_hi must be non-NULL");
    const V2GateMap* d_maps = nullptr;
    rc = v2_stage_maps(c, k, gate_seq, &d_maps);
    if (rc) return rc;
    const int n = 6 * (k + 1) + c->v2_qn * k;
    const int64_t M = n_active * (int64_t)prm->restarts;
    std::vector<double> b((size_t)4 * n);
    for (int i = 0; i < n; ++i) {
        b[i] = init_lo[i];
        b[n + i] = init_hi[i];
        b[2 * n + i] = bound_lo ? bound_lo[i] : -INFINITY;
        b[3 * n + i] = bound_hi ? bound_hi[i] : INFINITY;
        if (!(b[i] <= b[n + i]) || !std::isfinite(b[i]) || !std::isfinite(b[n + i]))
            return fail(SLAM_ERR_INVALID, "start range of parameter %d must be finite with lo <= hi", i);
        if (!(b[2 * n + i] <= b[3 * n + i])) return fail(SLAM_ERR_INVALID, "bounds of parameter %d: lo > hi", i);
    }
    const int32_t* d_active = nullptr;
    if (active) {
        for (int64_t s2 = 0; s2 < n_active; ++s2)
            if (active[s2] < 0 || active[s2] >= c->n_targets) return fail(SLAM_ERR_INVALID, "active[%lld] outside the resident batch", (long long)s2);
        HIP_TRY(c->active.reserve((size_t)n_active * sizeof(int32_t)));
        HIP_TRY(hipMemcpyAsync(c->active.p, active, (size_t)n_active * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        d_active = c->active.as<int32_t>();
    }
    const double* d_x0 = nullptr;
    if (x0) {
        for (int64_t i = 0; i < M * n; ++i)
            if (!(x0[i] > -1e8 && x0[i] < 1e8)) return fail(SLAM_ERR_INVALID, "x0[%lld] = %g: explicit seeds must be finite with |x| < 1e8", (long long)i, x0[i]);
        HIP_TRY(c->x0.reserve((size_t)M * n * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(c->x0.p, x0, (size_t)M * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        d_x0 = c->x0.as<double>();
    }
    HIP_TRY(c->v2_bounds.reserve(b.size() * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(c->v2_bounds.p, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // b is a local buffer
    {
        // stage buffers sized for n parameters per item (reserve_stage_buffers sizes for 6 (k + 1))
        HIP_TRY(c->item_rec.reserve(M * sizeof(ItemRec)));
        HIP_TRY(c->item_x.reserve(M * n * sizeof(double)));
        HIP_TRY(c->stage_loss.reserve(n_active * sizeof(double)));
        HIP_TRY(c->stage_x.reserve(n_active * n * sizeof(double)));
        HIP_TRY(c->stage_restart.reserve(n_active * sizeof(int32_t)));
    }
    HIP_TRY(hipMemsetAsync(c->counters.p, 0, sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), c->stream));
    hipLaunchKernelGGL(set_n_active_kernel, dim3(1), dim3(1), 0, c->stream, stage_ctl(c, k), (int32_t)n_active);
    HIP_TRY(hipGetLastError());
    HIP_TRY(c->solved.reserve((size_t)n_active * sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(c->solved.p, 0, (size_t)n_active * sizeof(int32_t), c->stream));
    bool bounded = false;
    for (int i = 0; i < n; ++i) bounded = bounded || std::isfinite(b[2 * n + i]) || std::isfinite(b[3 * n + i]);
    bool riswap_like = c->v2_qn == 1;
    for (int j = 0; j < k && riswap_like; ++j) {
        const V2GateMap& gm = c->v2_gates_host[(size_t)gate_seq[j]];
        riswap_like = gm.sel[1] < 0 && gm.sel[2] < 0 && gm.offset[1] == 0.0 && gm.offset[2] == 0.0;  // phi_c = 0, b = 0 (phi_g: no effect then)
    }
    V2Stage sgt{bounded, riswap_like, false, k, exit_loss, d_maps, d_active, (int32_t)n_active, d_x0, c->v2_bounds.as<double>(), prm};
    SLAM_V2_DISPATCH(v2_launch_minimize, c, sgt);
    if (rc) return rc;
    ReduceArgs r{};
    r.item_rec = c->item_rec.as<ItemRec>();
    r.item_x = c->item_x.as<double>();
    r.exit_loss = exit_loss;
    r.ordered = 1;  // the winner is the restart the reference's sequential loop breaks at (restarts below it always run to their end)
    r.ctl = stage_ctl(c, k);
    r.restarts = prm->restarts;
    r.n = n;
    r.stage_loss = c->stage_loss.as<double>();
    r.stage_x = c->stage_x.as<double>();
    r.stage_restart = c->stage_restart.as<int32_t>();
    hipLaunchKernelGGL(reduce_merge_kernel, dim3((unsigned)((n_active + 255) / 256)), dim3(256), 0, c->stream, r);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(best_loss, c->stage_loss.p, (size_t)n_active * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(best_x, c->stage_x.p, (size_t)n_active * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (best_restart) HIP_TRY(hipMemcpyAsync(best_restart, c->stage_restart.p, (size_t)n_active * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_ctl, c->counters.p, sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    rc = fetch_item_records(c, M, item_loss, item_iters, item_status, item_evals);
    if (rc) return rc;
    return collect_stats(c, k, k, c->h_ctl, prm->restarts);
}

// The span loop of a CircuitTemplateV2 (optimizer.py:233-303) enqueued as ONE chain, like decompose_body: results reset +
// first stage's active list, then per span the V2 optimizer kernel (target count from the device) and the shared epilogue
// kernel (reduction over restarts, merge into the running best -- rows of n_k parameters into rows of nmax --, compaction).
int v2_decompose_body(slam_ctx* c, int64_t first, int64_t count, int k_min, int k_max, const int32_t* gate_seqs, const double* init_lo,
                      const double* init_hi, const double* bound_lo, const double* bound_hi, const slam_opt_params* prm,
                      double success_threshold, FetchReq* fetch) {
    if (!c) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (c->n_targets <= 0) return fail(SLAM_ERR_STATE, "no targets: call slam_set_targets first");
    if (c->v2_gates_host.empty()) return fail(SLAM_ERR_STATE, "no parametrised gates: call slam_v2_set_gates first");
    if (k_min < 1 || k_max < k_min || k_max > SLAM_V2_MAX_SPAN) return fail(SLAM_ERR_UNSUPPORTED, "parametrised-gate templates support spans 1..%d (got [%d, %d])", SLAM_V2_MAX_SPAN, k_min, k_max);
    int rc = check_params(prm);
    if (rc) return rc;
    if (!gate_seqs || !init_lo || !init_hi) return fail(SLAM_ERR_INVALID, "gate_seqs, init_lo and init_hi must be non-NULL");
    if (first < 0 || count <= 0 || first + count > c->n_targets) return fail(SLAM_ERR_INVALID, "target window outside the resident batch");
    const int qn = c->v2_qn;
    const int nmax = 6 * (k_max + 1) + qn * k_max;
    // per span: gate maps and (init_lo | init_hi | bound_lo | bound_hi), staged once
    std::vector<V2GateMap> maps;
    std::vector<double> b;
    std::vector<size_t> map_off, b_off;
    std::vector<bool> bounded_k, riswap_k;
    {
        const int32_t* gs = gate_seqs;
        size_t po = 0;  // offset into the caller's concatenated per-parameter arrays
        for (int k = k_min; k <= k_max; ++k) {
            const int n = 6 * (k + 1) + qn * k;
            map_off.push_back(maps.size());
            bool rl = qn == 1;
            for (int j = 0; j < k; ++j) {
                if (gs[j] < 0 || gs[j] >= (int)c->v2_gates_host.size()) return fail(SLAM_ERR_INVALID, "gate_seqs: index %d outside the parametrised gate table", gs[j]);
                const V2GateMap& gm = c->v2_gates_host[(size_t)gs[j]];
                maps.push_back(gm);
                rl = rl && gm.sel[1] < 0 && gm.sel[2] < 0 && gm.offset[1] == 0.0 && gm.offset[2] == 0.0;
            }
            riswap_k.push_back(rl);
            b_off.push_back(b.size());
            bool bd = false;
            for (int part = 0; part < 4; ++part)
                for (int i = 0; i < n; ++i) {
                    double v;
                    if (part == 0) v = init_lo[po + i];
                    else if (part == 1) v = init_hi[po + i];
                    else if (part == 2) v = bound_lo ? bound_lo[po + i] : -INFINITY;
                    else v = bound_hi ? bound_hi[po + i] : INFINITY;
                    if (part < 2 && !std::isfinite(v)) return fail(SLAM_ERR_INVALID, "start range of parameter %d at span %d must be finite", i, k);
                    if (part >= 2 && std::isfinite(v)) bd = true;
                    b.push_back(v);
                }
            for (int i = 0; i < n; ++i) {
                if (!(init_lo[po + i] <= init_hi[po + i])) return fail(SLAM_ERR_INVALID, "start range of parameter %d at span %d: lo > hi", i, k);
                if (bound_lo && bound_hi && !(bound_lo[po + i] <= bound_hi[po + i])) return fail(SLAM_ERR_INVALID, "bounds of parameter %d at span %d: lo > hi", i, k);
            }
            bounded_k.push_back(bd);
            po += (size_t)n;
            gs += k;
        }
    }
    HIP_TRY(c->v2_maps.reserve(sizeof(V2GateMap) * maps.size()));
    HIP_TRY(c->v2_bounds.reserve(b.size() * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(c->v2_maps.p, maps.data(), sizeof(V2GateMap) * maps.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->v2_bounds.p, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // local buffers
    const int64_t N = count, M = N * (int64_t)prm->restarts;
    if (M > 0x7fff0000LL) return fail(SLAM_ERR_INVALID, "too many work items in one stage (%lld)", (long long)M);
    rc = ensure_results_n(c, nmax);
    if (rc) return rc;
    HIP_TRY(c->active.reserve(N * sizeof(int32_t)));
    HIP_TRY(c->active2.reserve(N * sizeof(int32_t)));
    HIP_TRY(c->item_rec.reserve(M * sizeof(ItemRec)));
    HIP_TRY(c->item_x.reserve(M * nmax * sizeof(double)));
    {
        // inverse Hessians of the long templates (v2_h_in_memory: more than 8 parameter slots per lane) live in device memory, one slice
        // per resident wavefront: sized HERE for the longest span of the loop -- growing the buffer between two stages of the chain
        // would free it under the stage in flight (hipFree synchronises the device: ADVICE r3)
        size_t need = 0;
        for (int k = k_min; k <= k_max; ++k) {
            const int na = (6 * (k + 1) + c->v2_qn * k + 3) / 4;
            if (na <= 8) continue;
            int64_t blocks = (M + kQuadsPerWave - 1) / kQuadsPerWave;
            const int64_t cap = (int64_t)kV2HmemWavesPerCu * c->compute_units;  // the launch's own bound (v2_launch_minimize_gq)
            if (blocks > cap) blocks = cap;
            const size_t bytes = (size_t)blocks * (size_t)(na * (na + 1) / 2 * 4 * kWave) * sizeof(float);  // == v2_h_floats_per_wave<K, QN>()
            need = bytes > need ? bytes : need;
        }
        if (need) HIP_TRY(c->v2_hmem.reserve(need));
    }
    HIP_TRY(c->stage_loss.reserve(N * sizeof(double)));
    HIP_TRY(c->stage_x.reserve(N * nmax * sizeof(double)));
    HIP_TRY(c->stage_restart.reserve(N * sizeof(int32_t)));
    HIP_TRY(c->solved.reserve(N * sizeof(int32_t)));
    HIP_TRY(c->stage_targets.reserve((size_t)N * 32 * sizeof(double)));
    HIP_TRY(hipEventRecord(c->ev_t0, c->stream));
    hipLaunchKernelGGL(init_results_kernel, dim3((unsigned)((N * 16 + 255) / 256)), dim3(256), 0, c->stream, c->best_loss.as<double>(),
                       c->best_cycles.as<int32_t>(), c->span_loss.as<double>(), c->active.as<int32_t>(), first, N, stage_ctl(c, k_min),
                       c->targets.as<double>(), c->stage_targets.as<double>(), c->solved.as<int32_t>(), c->counters.as<StageCtl>(),
                       (int32_t)(sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2) / 8), 0);
    HIP_TRY(hipGetLastError());
    DevBuf* cur = &c->active;
    DevBuf* nxt = &c->active2;
    for (int k = k_min, si = 0; k <= k_max; ++k, ++si) {
        const int n = 6 * (k + 1) + qn * k;
        V2Stage sgt{bounded_k[(size_t)si], riswap_k[(size_t)si], true, k, success_threshold, c->v2_maps.as<V2GateMap>() + map_off[(size_t)si],
                    cur->as<int32_t>(), (int32_t)N, nullptr, c->v2_bounds.as<double>() + b_off[(size_t)si], prm};
        SLAM_V2_DISPATCH(v2_launch_minimize, c, sgt);
        if (rc) return rc;
        EpilogueArgs e{};
        e.r.item_rec = c->item_rec.as<ItemRec>();
        e.r.item_x = c->item_x.as<double>();
        e.r.exit_loss = success_threshold;
        e.r.ordered = 1;
        e.r.ctl = stage_ctl(c, k);
        e.r.restarts = prm->restarts;
        e.r.n = n;
        e.r.stage_loss = c->stage_loss.as<double>();
        e.r.stage_x = c->stage_x.as<double>();
        e.r.stage_restart = c->stage_restart.as<int32_t>();
        e.r.active = cur->as<int32_t>();
        e.r.nmax = nmax;
        e.r.k = k;
        e.r.best_loss = c->best_loss.as<double>();
        e.r.best_x = c->best_x.as<double>();
        e.r.best_cycles = c->best_cycles.as<int32_t>();
        e.r.span_loss = c->span_loss.as<double>();
        e.has_next = k < k_max ? 1 : 0;
        e.threshold = success_threshold;
        e.active_out = nxt->as<int32_t>();
        e.next = stage_ctl(c, k + 1);
        e.targets = c->targets.as<double>();
        e.stage_targets = c->stage_targets.as<double>();
        e.solved = c->solved.as<int32_t>();
        if (N <= 2048) hipLaunchKernelGGL(stage_epilogue_kernel<256>, dim3(1), dim3(256), 0, c->stream, e);
        else if (N <= kEpilogueMaxTargets) hipLaunchKernelGGL(stage_epilogue_kernel<1024>, dim3(1), dim3(1024), 0, c->stream, e);
        else hipLaunchKernelGGL(stage_epilogue_grid_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, e);
        HIP_TRY(hipGetLastError());
        DevBuf* t = cur; cur = nxt; nxt = t;
    }
    HIP_TRY(hipEventRecord(c->ev_t1, c->stream));
    if (fetch) {
        rc = enqueue_fetch_n(c, nmax, first, count, *fetch);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(c->h_ctl, c->counters.p, sizeof(StageCtl) * (SLAM_MAX_SPAN_EVAL + 2), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->ev_done, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev_done));
    if (fetch) finish_fetch(c, *fetch);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
    c->stats.total_ms = ms;
    return collect_stats(c, k_min, k_max, c->h_ctl, prm->restarts);
}

}  // namespace

extern "C" {

int slam_v2_decompose_range(slam_ctx* ctx, int64_t first, int64_t count, int k_min, int k_max, const int32_t* gate_seqs, const double* init_lo,
                            const double* init_hi, const double* bound_lo, const double* bound_hi, const slam_opt_params* params,
                            double success_threshold, double* best_loss, double* best_x, int32_t* best_cycles) {
    FetchReq fr{best_loss, best_x, best_cycles};
    return drained(ctx, v2_decompose_body(ctx, first, count, k_min, k_max, gate_seqs, init_lo, init_hi, bound_lo, bound_hi, params, success_threshold,
                                          (best_loss || best_x || best_cycles) ? &fr : nullptr));
}

int slam_v2_set_gates(slam_ctx* ctx, const slam_v2_gate* gates, int32_t n_gates) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    if (!gates || n_gates <= 0 || n_gates > SLAM_MAX_GATES) return fail(SLAM_ERR_INVALID, "n_gates must be in 1..%d", SLAM_MAX_GATES);
    const int qn = gates[0].n_params;
    if (qn != 1 && qn != 2 && qn != 4) return fail(SLAM_ERR_UNSUPPORTED, "parametrised gates take 1, 2 or 4 parameters (got %d)", qn);
    std::vector<V2GateMap> tmp((size_t)n_gates);
    for (int g = 0; g < n_gates; ++g) {
        if (gates[g].n_params != qn) return fail(SLAM_ERR_UNSUPPORTED, "all parametrised gates of a template must take the same number of parameters");
        for (int r = 0; r < 4; ++r) {
            if (gates[g].sel[r] < -1 || gates[g].sel[r] >= qn) return fail(SLAM_ERR_INVALID, "gate %d: sel[%d] = %d outside [-1, %d)", g, r, gates[g].sel[r], qn);
            if (!std::isfinite(gates[g].scale[r]) || !std::isfinite(gates[g].offset[r])) return fail(SLAM_ERR_INVALID, "gate %d: non-finite map", g);
            tmp[(size_t)g].sel[r] = gates[g].sel[r];
            tmp[(size_t)g].scale[r] = gates[g].sel[r] < 0 ? 0.0 : gates[g].scale[r];
            tmp[(size_t)g].offset[r] = gates[g].offset[r];
            tmp[(size_t)g].pad[r] = 0;
        }
    }
    ctx->v2_gates_host.swap(tmp);
    ctx->v2_qn = qn;
    for (int k = 0; k <= SLAM_V2_MAX_SPAN; ++k) ctx->v2_cons_n[k] = 0;  // a constraint belongs to the gate table it was set for
    return SLAM_OK;
}

int slam_v2_set_constraint(slam_ctx* ctx, int k, const double* weights, int n, double cost_max) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    if (k < 1 || k > SLAM_V2_MAX_SPAN) return fail(SLAM_ERR_UNSUPPORTED, "parametrised-gate templates support spans 1..%d (got %d)", SLAM_V2_MAX_SPAN, k);
    if (!weights || n == 0) {  // remove_constraint (basisv2.py:202-204)
        ctx->v2_cons_n[k] = 0;
        return SLAM_OK;
    }
    if (ctx->v2_gates_host.empty()) return fail(SLAM_ERR_STATE, "no parametrised gates: call slam_v2_set_gates first");
    const int want = 6 * (k + 1) + ctx->v2_qn * k;
    if (n != want) return fail(SLAM_ERR_INVALID, "span %d with %d parameters per gate has %d parameters (got %d weights)", k, ctx->v2_qn, want, n);
    if (!std::isfinite(cost_max)) return fail(SLAM_ERR_INVALID, "cost_max must be finite");
    double w2max = 0.0;
    for (int i = 0; i < n; ++i) {
        if (!std::isfinite(weights[i])) return fail(SLAM_ERR_INVALID, "weights[%d] is not finite", i);
        w2max = std::max(w2max, weights[i] * weights[i]);
    }
    if (!(w2max > 0.0)) return fail(SLAM_ERR_INVALID, "a cost constraint needs a non-zero weight");
    HIP_TRY(ctx->v2_cons_w[k].reserve((size_t)n * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(ctx->v2_cons_w[k].p, weights, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the caller's buffer
    ctx->v2_cons_n[k] = n;
    ctx->v2_cons_max[k] = cost_max;
    ctx->v2_cons_rho[k] = 30.0 / w2max;
    return SLAM_OK;
}

int slam_v2_eval_loss_grad(slam_ctx* ctx, int k, const int32_t* gate_seq, const double* x, const int32_t* target_of, int64_t M,
                           double* loss, double* grad, double* unitary) {
    return drained(ctx, v2_eval_body(ctx, k, gate_seq, x, target_of, M, loss, grad, unitary));
}

int slam_v2_minimize_stage(slam_ctx* ctx, int k, const int32_t* gate_seq, const int32_t* active, int64_t n_active, const double* x0,
                           const double* init_lo, const double* init_hi, const double* bound_lo, const double* bound_hi,
                           const slam_opt_params* params, double exit_loss, double* best_loss, double* best_x, int32_t* best_restart,
                           double* item_loss, int32_t* item_iters, int32_t* item_status, int32_t* item_evals) {
    return drained(ctx, v2_minimize_body(ctx, k, gate_seq, active, n_active, x0, init_lo, init_hi, bound_lo, bound_hi, params, exit_loss,
                                         best_loss, best_x, best_restart, item_loss, item_iters, item_status, item_evals));
}

int slam_v2_minimize_stage_trace(slam_ctx* ctx, int k, const int32_t* gate_seq, const int32_t* active, int64_t n_active, const double* x0,
                                 const double* init_lo, const double* init_hi, const double* bound_lo, const double* bound_hi,
                                 const slam_opt_params* params, double exit_loss, int32_t trace_cap, double* best_loss, double* best_x,
                                 int32_t* best_restart, double* item_loss, int32_t* item_iters, int32_t* item_status, double* trace_loss,
                                 double* trace_x) {
    if (!ctx) return fail(SLAM_ERR_INVALID, "ctx is NULL");
    if (!params) return fail(SLAM_ERR_INVALID, "params is NULL");
    if (trace_cap <= 0 || !trace_loss || !trace_x) return fail(SLAM_ERR_INVALID, "trace buffers and trace_cap > 0 are required");
    if (k < 1 || k > SLAM_V2_MAX_SPAN) return fail(SLAM_ERR_UNSUPPORTED, "parametrised-gate templates support spans 1..%d (got %d)", SLAM_V2_MAX_SPAN, k);
    if (!active) n_active = ctx->n_targets;
    if (n_active <= 0 || params->restarts <= 0) return fail(SLAM_ERR_INVALID, "nothing to trace");
    const int n = 6 * (k + 1) + ctx->v2_qn * k;
    const int64_t M = n_active * (int64_t)params->restarts;
    const size_t rows = (size_t)M * (size_t)trace_cap;
    if (rows * (size_t)(n + 1) * sizeof(double) > ((size_t)4 << 30))
        return fail(SLAM_ERR_INVALID, "trace of %lld items x %d iterations exceeds 4 GiB: trace fewer targets at a time", (long long)M, trace_cap);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->trace_loss.reserve(rows * sizeof(double)));
    HIP_TRY(ctx->trace_x.reserve(rows * n * sizeof(double)));
    HIP_TRY(hipMemsetAsync(ctx->trace_loss.p, 0xFF, rows * sizeof(double), ctx->stream));  // rows no iteration reaches read as NaN
    HIP_TRY(hipMemsetAsync(ctx->trace_x.p, 0xFF, rows * n * sizeof(double), ctx->stream));
    ctx->trace_cap = trace_cap;
    int rc = slam_v2_minimize_stage(ctx, k, gate_seq, active, n_active, x0, init_lo, init_hi, bound_lo, bound_hi, params, exit_loss, best_loss, best_x,
                                    best_restart, item_loss, item_iters, item_status, nullptr);
    ctx->trace_cap = 0;
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(trace_loss, ctx->trace_loss.p, rows * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(trace_x, ctx->trace_x.p, rows * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

}  // extern "C"
