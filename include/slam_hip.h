/*
 * slam_hip.h -- C ABI of libslamhip.so, the MI355X (gfx950) implementation of the
 * SLAM template-optimizer inner loop.
 *
 * The reference (Pitt-JonesLab/slam_decomposition) is pure Python: there is no FFI
 * today.  Each entry point below states the reference code it replaces
 * (paths relative to the reference checkout).  Python binds these with ctypes
 * (slam_decomposition_amd/_ffi.py); INTEGRATION.md shows the stub a maintainer of
 * the reference would add to src/slam/optimizer.py.
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is caller-allocated, C-contiguous,
 *     8-byte aligned host memory; the library never keeps a caller pointer after
 *     the call returns.  Device buffers are owned by the context.
 *   - complex matrices are row-major interleaved (re, im) doubles: a 4x4 matrix is
 *     double[4][4][2] = 32 doubles.
 *   - template parameters x are in index order P0..P{n-1}, n = 6(k+1): layer-major,
 *     within a layer (theta, phi, lambda) of the qubit-0 U gate then of the qubit-1
 *     U gate  (reference: src/slam/basis.py:152-169).
 *   - functions return 0 on success or a negative SLAM_ERR_* code;
 *     slam_last_error() gives the message for the calling thread.
 *   - one context = one GPU = one host thread at a time.  Multi-GPU = one context
 *     per device, one process per GPU, merged through slam_comm_* (RCCL) at the end.
 */
#ifndef SLAM_HIP_H
#define SLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM_OK 0
#define SLAM_ERR_INVALID (-1)     /* bad argument */
#define SLAM_ERR_HIP (-2)         /* HIP runtime error (message has the HIP string) */
#define SLAM_ERR_UNSUPPORTED (-3) /* e.g. span outside [1, SLAM_MAX_SPAN_MINIMIZE] */
#define SLAM_ERR_NOMEM (-4)
#define SLAM_ERR_STATE (-5)       /* call order: targets / basis not set */

#define SLAM_MAX_SPAN_QUAD 5      /* spans of the register-resident kernels (a quad of lanes per item; 1..3 are the tuned ones);
                                    the reference's default maximum_span_guess, src/slam/basis.py:59 */
#define SLAM_MAX_SPAN_EVAL 16     /* spans of slam_eval_*: beyond SLAM_MAX_SPAN_QUAD one wavefront per item (csrc/slam_long.hpp) */
#define SLAM_MAX_SPAN_MINIMIZE 16 /* spans of slam_minimize_stage / slam_decompose*: beyond SLAM_MAX_SPAN_QUAD a wavefront per item
                                    with its quasi-Newton metric (fp32) in device memory (the templates MixedOrderBasisCircuitTemplate builds from
                                    weak gates, src/slam/basis.py:213-359) */
#define SLAM_MAX_GATES 256
#define SLAM_MAX_MAXITER 4000    /* per-restart iteration cap accepted by the kernels (reference: 2500) */

/* per-item optimizer status (slam_minimize_stage: item_status) */
#define SLAM_ST_CONVERGED 0 /* loss < stop_loss or |g|_inf < gtol */
#define SLAM_ST_MAXITER 1   /* reference: options={"maxiter": 2500}, src/slam/optimizer.py:275 */
#define SLAM_ST_LINESEARCH 2
#define SLAM_ST_NONFINITE 3
#define SLAM_ST_STALLED 4   /* no representable decrease (fp64 noise floor at a non-zero minimum) */
#define SLAM_ST_PREEMPTED 5 /* a sibling restart of the same target reached stop_loss first
                               (reference: break over restarts, src/slam/optimizer.py:287-295) */

/* flags for slam_minimize_stage / slam_decompose */
#define SLAM_FLAG_EARLY_EXIT 1u /* stop a target's other restarts once one reaches stop_loss */
#define SLAM_FLAG_ORDERED 2u    /* with EARLY_EXIT: a successful restart (final loss < the span loop's success threshold;
                                   stop_loss in slam_minimize_stage) only stops the restarts with a HIGHER index, and the
                                   stage result is the lowest-index successful restart -- exactly the restart at which
                                   the reference's sequential loop breaks (src/slam/optimizer.py:287-295).  Results are
                                   then independent of scheduling, shard layout and items_per_quad (bitwise
                                   reproducible for a fixed seed); without it the first restart to FINISH wins, which
                                   runs fewer evaluations but depends on timing. */
#define SLAM_FLAG_STAGED 4u     /* span loops (slam_decompose*): always one optimizer launch + one bookkeeping launch per span.
                                   Without it a SMALL batch (at most one target per SIMD, at most 16 restarts -- or up to 64
                                   when targets x restarts fill the chip --, spans <= 3, ordered early exit, one gate structure
                                   class) runs through the one-wavefront-per-target kernel: a single launch in
                                   which every target goes from span to span on its own -- same items, same seeds, same
                                   results bit for bit, no stage barrier (round 4).  It also forces the span-by-span order
                                   where calls of at most two targets per CU (speculative spans) and calls of at most 2^17
                                   work items per span (overlapped spans) would run their spans side by side. */
#define SLAM_FLAG_OVERLAP 8u    /* span loops: run the spans side by side for ALL targets of the call whatever its size -- one
                                   optimizer launch per span on its own stream, the loop's bookkeeping afterwards in span order.
                                   Same results bit for bit; the stages the loop would not have reached are wasted work, so this
                                   pays when the early spans cannot succeed anyway (a CNOT basis: no generic target before three
                                   gates) and the call is alone on the device.  Needs EARLY_EXIT | ORDERED, spans <= 3. */
#define SLAM_FLAG_NO_OVERLAP 16u /* span loops: never run the spans of a MEDIUM call (at most 2^17 work items per span) side by side,
                                   which the library does by itself otherwise -- for callers that keep several calls in flight on
                                   the device: the chip is full then and the speculative stages only add work. */

#define SLAM_FLAG_NO_EXTERIOR 32u /* fixed-gate templates: CircuitTemplate(no_exterior_1q=True), src/slam/basis.py:57,154,165 -- the
                                   template is G_k K_{k-1} ... K_1 G_1: the six parameters of layer 0 and of layer k are pinned at zero
                                   (U3(0, 0, 0) = 1; start values and gradient components zeroed), so the optimisation runs over the
                                   6 (k - 1) interior parameters; x rows keep the 6 (k + 1) layout with zeros in the pinned places. */

typedef struct slam_ctx slam_ctx;

typedef struct slam_opt_params {
    int32_t restarts;     /* R: multi-start seeds per target per span (TRAINING_RESTARTS, optimizer.py:19) */
    int32_t maxiter;      /* per-restart iteration cap (optimizer.py:275: 2500) */
    double gtol;          /* stop when |g|_inf < gtol (SciPy BFGS default 1e-5; we default 1e-9) */
    double stop_loss;     /* stop when loss < stop_loss */
    uint64_t seed;        /* Philox key for x0 ~ U[0,2pi)^n (basis.py:106-111) */
    uint32_t flags;       /* SLAM_FLAG_* (bits above SLAM_FLAG_ORDERED are ignored) */
    uint32_t items_per_quad; /* launch shaping: at least this many work items per resident quad before more
                                wavefronts are launched.  0/1 = spread a small batch over as many wavefronts as
                                possible (lowest latency of one batch); 4..8 = keep quads refilled (highest
                                throughput when several batches are in flight on different streams).  With
                                SLAM_FLAG_ORDERED (or without early exit) results do not depend on it; with the plain
                                early-exit flag the winner among equally successful restarts depends on timing anyway. */
    double gtol_far;      /* with far_loss: also stop when |g|_inf < gtol_far and loss > far_loss, i.e. at a */
    double far_loss;      /* stationary point that is clearly not a zero of the loss.  gtol_far = 1e-5 is SciPy's
                             default gtol, which is what the reference runs with (optimizer.py:270-278);
                             set gtol_far = 0 to disable */
    int64_t target_base;  /* added to a target's index within the context's resident batch in the Philox key, so that a
                             shard holding targets [base, base + n) of a larger job draws the seeds the whole job would */
} slam_opt_params;

typedef struct slam_stats {
    double kernel_ms;         /* sum of HIP-event durations of the minimize kernel launches */
    int64_t kernel_launches;  /* number of minimize kernel launches */
    int64_t evals[SLAM_MAX_SPAN_EVAL + 1]; /* fused loss+grad evaluations per span k (index k) */
    int64_t items[SLAM_MAX_SPAN_EVAL + 1]; /* (target, seed) work items per span k */
    double total_ms;          /* HIP-event time of the last slam_decompose / slam_minimize_stage */
    double kernel_ms_span[SLAM_MAX_SPAN_EVAL + 1]; /* per span k (index k >= 1): HIP-event time of span k's optimizer launches.  Span-by-span
                                                      calls: they add up to kernel_ms.  Calls that run their spans side by side
                                                      (speculative / overlapped spans): each span's own launch -- overlapping in time, so
                                                      their sum exceeds kernel_ms.  Index 0: time of launches that cover SEVERAL spans (the
                                                      one-wavefront-per-target loop), for which no per-span split exists. */
    int64_t wave_rounds[SLAM_MAX_SPAN_EVAL + 1];   /* lock-step evaluation rounds summed over wavefronts, per span:
                                                      evals / (16 * wave_rounds) = fraction of quads that held an item */
    /* split of evals[k]: evaluations whose point was accepted (a restart's initial point or an Armijo-accepted
       step), evaluations of restarts that ended pre-empted by a successful sibling (all of them, accepted or not);
       the rest, evals - accepted - preempted, are rejected line-search trials */
    int64_t evals_accepted[SLAM_MAX_SPAN_EVAL + 1];
    int64_t evals_preempted[SLAM_MAX_SPAN_EVAL + 1];
} slam_stats;

/* Thread-local message of the last failing call on this thread. */
const char* slam_last_error(void);

/* Number of visible HIP devices. */
int slam_device_count(int* count);

/* Create / destroy a context bound to one device.  Creates a private HIP stream. */
int slam_ctx_create(int device, slam_ctx** out);
int slam_ctx_destroy(slam_ctx* ctx);

/* Name and CU count of the context's device (name buffer >= 256 bytes). */
int slam_ctx_device_info(slam_ctx* ctx, char* name, int name_len, int* compute_units, int* clock_khz);

/*
 * Upload the batch of target unitaries (resident in HBM until replaced).
 * Replaces iterating the sampler one target at a time:
 *   src/slam/optimizer.py:180-186 (approximate_from_distribution), src/slam/sampler.py:25-27.
 * targets: double[n_targets][4][4][2].
 */
int slam_set_targets(slam_ctx* ctx, const double* targets, int64_t n_targets);

/*
 * Fill the resident batch with n_targets Haar-random 4x4 unitaries generated on the device:
 * T_i = QR(Ginibre(Philox(seed, first_index + i))).Q with a positive diagonal of R -- the recipe of
 * HaarSample._get_unitary (src/slam/sampler.py:62-71 -> qiskit random_unitary -> SciPy unitary_group),
 * with Philox4x32-10 + Box-Muller in place of NumPy's generator: same distribution, different sample.
 * slam_get_targets copies resident targets [first, first + count) back: double[count][4][4][2].
 */
int slam_sample_haar(slam_ctx* ctx, uint64_t seed, int64_t first_index, int64_t n_targets);
int slam_get_targets(slam_ctx* ctx, int64_t first, int64_t count, double* out);

/*
 * Weyl-chamber coordinates (c1, c2, c3), in units of pi, of 4x4 unitaries -- weylchamber.c1c2c3 as called by
 * VariationalTemplate.target_invariant (src/slam/basis_abc.py:80-84) and on the optimizer's results
 * (src/slam/optimizer.py:85,103,224) -- computed on the device, one thread per unitary.  ndigits >= 0 rounds like
 * numpy.round (the reference uses 8); ndigits < 0 leaves the values unrounded.  out: double[count][3].
 *   slam_c1c2c3          unitaries from the host, double[count][4][4][2]
 *   slam_targets_c1c2c3  the resident targets [first, first + count) (nothing is uploaded)
 *   slam_eval_c1c2c3     the template unitaries CircuitTemplate.eval(x[m]) of M parameter vectors (the unitaries
 *                        stay on the device; needs resident targets and gates like slam_eval_unitary)
 */
int slam_c1c2c3(slam_ctx* ctx, const double* unitaries, int64_t count, int ndigits, double* out);
int slam_targets_c1c2c3(slam_ctx* ctx, int64_t first, int64_t count, int ndigits, double* out);
int slam_eval_c1c2c3(slam_ctx* ctx, int k, const int32_t* gate_seq, const double* x, int64_t M, int ndigits, double* out);

/*
 * Span predictor on the device: for every resident target of [first, first + count) the smallest number k of leading gates of a
 * template whose coverage set contains the target -- the lookup CircuitTemplate.get_spanning_range makes with use_polytopes=True
 * (src/slam/basis.py:95-100 -> monodromy_range_from_target, src/slam/utils/polytopes/polytope_wrap.py:39-94).  The coverage sets
 * come from the caller as half-spaces in the target's alcove coordinates (slam_decomposition_amd/coverage.py computes them from the
 * gates' Weyl coordinates):
 *   point   double[4]            alcove point of the first gate (k = 1: the target must be that class)
 *   bounds  double[k_max][14]    row k - 1 (k >= 2; row 0 unused): bounds[p] <= sum of the target's alcove coordinates over the
 *                                p-th subset ({4},{3},{2},{1},{4,3},{4,2},{4,1},{3,2},{3,1},{2,1},{4,3,2},{4,3,1},{4,2,1},{3,2,1} of the
 *                                decreasing coordinates a_1..a_4); -inf = no constraint
 *   tol                          widens the regions (units of pi)
 * spans_out: int32[count]: 0 = local target, 1 .. k_max, k_max + 1 = out of reach of the whole template.  k_max <= 5.
 */
int slam_predict_spans(slam_ctx* ctx, int64_t first, int64_t count, int k_max, const double* point, const double* bounds, double tol,
                       int32_t* spans_out);

/*
 * Upload the table of 2Q basis-gate matrices (CircuitTemplate(base_gates=...),
 * src/slam/basis.py:52-69; matrices from src/slam/utils/gates/custom_gates.py).
 * gates: double[n_gates][4][4][2].
 */
int slam_set_gates(slam_ctx* ctx, const double* gates, int32_t n_gates);

/*
 * Fused forward chain + BasicCost + analytic gradient for M independent items.
 * Replaces objective_func (src/slam/optimizer.py:191-214) = CircuitTemplate.eval
 * (src/slam/basis.py:102-104) + BasicCost.unitary_fidelity
 * (src/slam/cost_function.py:140-145), and SciPy's finite-difference gradient
 * (src/slam/optimizer.py:270-278, no jac) with the analytic one.
 *   k          span (number of 2Q gates), 1..SLAM_MAX_SPAN_EVAL
 *   gate_seq   int32[k]: index into the gate table for G_1..G_k
 *   x          double[M][6(k+1)]
 *   target_of  int32[M]: target index of each item
 *   loss       double[M]           (out)
 *   grad       double[M][6(k+1)]   (out, may be NULL)
 */
int slam_eval_loss_grad(slam_ctx* ctx, int k, const int32_t* gate_seq, const double* x,
                        const int32_t* target_of, int64_t M, double* loss, double* grad);

/*
 * CircuitTemplate.eval (src/slam/basis.py:102-104) for M parameter vectors: the 4x4 template
 * unitary W(x) = K_k G_k ... G_1 K_0, as qiskit's Operator(circuit).data returns it.
 *   unitary    double[M][4][4][2]  (out)
 *   loss       double[M]           (out, may be NULL) BasicCost against target_of[m]
 */
int slam_eval_unitary(slam_ctx* ctx, int k, const int32_t* gate_seq, const double* x,
                      const int32_t* target_of, int64_t M, double* unitary, double* loss);

/*
 * One span stage of TemplateOptimizer._run (src/slam/optimizer.py:233-303) for a
 * batch: for every active target run `restarts` independent quasi-Newton (BFGS)
 * minimisations from different seeds and return the best.
 * Replaces the `for r_i in range(training_restarts): opt.minimize(...)` loop
 * (optimizer.py:253-295) and parameter_guess (basis.py:106-111).
 *   active        int32[n_active] target indices (NULL = all targets 0..n_targets-1)
 *   x0            optional double[n_active][R][n] explicit seeds (NULL = Philox from params->seed,
 *                 counter = (pair index, restart, target index, k))
 *   best_loss     double[n_active]      (out)
 *   best_x        double[n_active][n]   (out)
 *   best_restart  int32[n_active]       (out, may be NULL)
 *   item_loss     double[n_active][R]   (out, may be NULL)  per-restart final loss
 *   item_iters    int32[n_active][R]    (out, may be NULL)
 *   item_status   int32[n_active][R]    (out, may be NULL)  SLAM_ST_*
 *   item_evals    int32[n_active][R]    (out, may be NULL)  loss+grad evaluations used
 */
int slam_minimize_stage(slam_ctx* ctx, int k, const int32_t* gate_seq, const int32_t* active,
                        int64_t n_active, const double* x0, const slam_opt_params* params,
                        double* best_loss, double* best_x, int32_t* best_restart,
                        double* item_loss, int32_t* item_iters, int32_t* item_status,
                        int32_t* item_evals);

/*
 * The whole span loop of TemplateOptimizer._run for the resident batch
 * (src/slam/optimizer.py:233-303): spans k = k_min..k_max; after each span the targets
 * whose best loss is < success_threshold leave the batch (optimizer.py:301-303), the
 * rest continue with the next span.  Compaction stays on the device; only the final
 * per-target results return.
 *   gate_seqs   int32[(k_max)(k_max+1)/2 ...] concatenated sequences for k = k_min..k_max
 *               (k_min entries, then k_min+1, ...)
 *   best_loss   double[n_targets]            (out)
 *   best_x      double[n_targets][6(k_max+1)] (out; first 6(best_cycles+1) entries valid)
 *   best_cycles int32[n_targets]             (out) span of the best result (optimizer.py:284)
 */
int slam_decompose(slam_ctx* ctx, int k_min, int k_max, const int32_t* gate_seqs,
                   const slam_opt_params* params, double success_threshold, double* best_loss,
                   double* best_x, int32_t* best_cycles);

/* Same as slam_decompose but leaves the results on the device (bench timing without D2H).
 * Fetch them afterwards with slam_fetch_results. */
int slam_decompose_resident(slam_ctx* ctx, int k_min, int k_max, const int32_t* gate_seqs,
                            const slam_opt_params* params, double success_threshold);
int slam_fetch_results(slam_ctx* ctx, int k_max, double* best_loss, double* best_x,
                       int32_t* best_cycles);

/* The same span loop restricted to the window [first, first + count) of the resident targets
 * (results of other targets are left untouched): lets a caller keep many batches resident and
 * process them one after the other, or overlap batches from several host threads' contexts. */
int slam_decompose_range(slam_ctx* ctx, int64_t first, int64_t count, int k_min, int k_max,
                         const int32_t* gate_seqs, const slam_opt_params* params,
                         double success_threshold);
/* slam_decompose_range followed by slam_fetch_results_range(ctx, k_max, first, count, ...) with one host wait
 * instead of two: the result copies are enqueued behind the last stage. */
int slam_decompose_range_fetch(slam_ctx* ctx, int64_t first, int64_t count, int k_min, int k_max, const int32_t* gate_seqs,
                               const slam_opt_params* params, double success_threshold, double* best_loss, double* best_x,
                               int32_t* best_cycles);

/* The span loops of SEVERAL contexts in one chain of kernels (round 4): n_ctx contexts on one device, each holding the target
 * window [first, first + count) and ITS OWN gate table (slam_set_gates) -- the bases of a basis-gate sweep
 * (src/slam/utils/gates/bare_candidates.py:47-69 builds such a family; the reference then runs one TemplateOptimizer per gate).
 * Per span ONE optimizer launch serves all of them -- a wavefront works on one context's queue at a time, so the gates stay scalar
 * operands -- and one bookkeeping launch.  Every context ends with exactly the resident results its own slam_decompose_range
 * call would have left (needs SLAM_FLAG_EARLY_EXIT | SLAM_FLAG_ORDERED: results independent of scheduling); fetch them per
 * context with slam_fetch_results_range.  Spans 1..3; the work is enqueued on ctxs[0]'s stream, whose statistics get the kernel
 * times (evaluation counts go to each context's own statistics).  The contexts must be idle and distinct. */
int slam_decompose_multi(slam_ctx** ctxs, int32_t n_ctx, int64_t first, int64_t count, int k_min, int k_max, const int32_t* gate_seqs,
                         const slam_opt_params* params, double success_threshold);

/* The same for an explicit list of resident-target indices (e.g. the targets a span predictor assigns to one
 * template size: CircuitTemplate.get_spanning_range with use_polytopes, src/slam/basis.py:95-100).  Results land
 * in the per-target resident arrays like those of slam_decompose_range; fetch them with
 * slam_fetch_results_range(ctx, k_layout, ...): rows of best_x are 6 (k_layout + 1) wide, so that lists with
 * different k_max can share one resident result set (k_layout = 0 means k_max). */
int slam_decompose_list(slam_ctx* ctx, const int32_t* targets, int64_t count, int k_min, int k_max, int k_layout,
                        const int32_t* gate_seqs, const slam_opt_params* params, double success_threshold);
int slam_fetch_results_range(slam_ctx* ctx, int k_max, int64_t first, int64_t count,
                             double* best_loss, double* best_x, int32_t* best_cycles);

/*
 * CircuitTemplate(use_polytopes=True) for the resident targets [first, first + count) in ONE chain of kernels (round 5): the template
 * size every target needs is looked up on the device (the half-spaces of slam_predict_spans: `point`, `bounds[k_max][14]`, `tol`), the
 * targets are sorted into per-size lists there, and one span loop runs in which the targets of size k join at stage k -- what
 * get_spanning_range + _run do per target in the reference (src/slam/basis.py:95-100 -> utils/polytopes/polytope_wrap.py:39-94,
 * src/slam/optimizer.py:233-303), without the host's per-size index lists and one slam_decompose_list call per size.
 *   carry = 0  exact regions: a target runs at its own size only (the reference's range(k, k + 1));
 *   carry = 1  a target that misses success_threshold at its size goes on to the next one (lower bounds, widened regions).
 * gate_seqs: the sequences of spans 1 .. k_max, concatenated.  Results stay resident (rows 6 (k_max + 1) wide: fetch with
 * slam_fetch_results_range(ctx, k_max, ...)); a local target (size 0) gets (loss 0, cycles 0), a target out of the template's reach
 * (+inf, -1); their numbers come back in n_local / n_unreachable (either may be NULL).
 */
int slam_decompose_predicted(slam_ctx* ctx, int64_t first, int64_t count, int k_max, const double* point, const double* bounds, double tol,
                             int carry, const int32_t* gate_seqs, const slam_opt_params* params, double success_threshold,
                             int64_t* n_local, int64_t* n_unreachable);

/*
 * Running best loss of resident targets [first, first + count) after every span the span loop ran for them:
 * out[t][k - 1] = best loss after span k, NaN where the target did not run span k (solved earlier, or k outside the
 * call's range) -- the value of the reference's "Cycle (k =...), Best Loss=..." log line (src/slam/optimizer.py:297).
 * out: double[count][SLAM_MAX_SPAN_EVAL].
 */
int slam_fetch_span_losses(slam_ctx* ctx, int64_t first, int64_t count, double* out);

/*
 * slam_minimize_stage with the per-iteration trajectories the reference collects through SciPy's callback when
 * use_callback=True (callbackF, src/slam/optimizer.py:217-224: loss and point after every BFGS iteration):
 *   trace_loss  double[n_active][R][trace_cap]      loss after accepted step number it at [it - 1], NaN beyond item_iters
 *   trace_x     double[n_active][R][trace_cap][n]   parameters at the same point (feed slam_eval_c1c2c3 for the
 *                                                   coordinate trajectory of optimizer.py:223-224)
 *   exit_loss   with SLAM_FLAG_EARLY_EXIT | SLAM_FLAG_ORDERED: a restart that ends below it stops the restarts of
 *               HIGHER index only, so every restart up to the first successful one runs to its end -- what the
 *               reference's sequential loop records (the span loop's success threshold, optimizer.py:287)
 * Meant for a few targets at a time (the trace must fit 4 GiB).  Spans 1..SLAM_MAX_SPAN_MINIMIZE (both kernel families, round 5).
 */
int slam_minimize_stage_trace(slam_ctx* ctx, int k, const int32_t* gate_seq, const int32_t* active, int64_t n_active,
                              const double* x0, const slam_opt_params* params, double exit_loss, int32_t trace_cap,
                              double* best_loss, double* best_x, int32_t* best_restart, double* item_loss,
                              int32_t* item_iters, int32_t* item_status, double* trace_loss, double* trace_x);

/*
 * Objective used by every later evaluation / minimisation of this context
 * (UnitaryCostFunction subclasses, src/slam/cost_function.py):
 *   SLAM_COST_BASIC   BasicCost   1 - |Tr(T^+ U)| / d                       (cost_function.py:140-145), default
 *   SLAM_COST_SQUARE  SquareCost  1 - (|Tr(T^+ U)|^2 + d) / (d (d + 1))     (cost_function.py:169-173)
 * Anything else fails like the reference's objective_func: "Unrecognized Cost Function" (optimizer.py:211).
 */
#define SLAM_COST_BASIC 0
#define SLAM_COST_SQUARE 1
int slam_set_cost(slam_ctx* ctx, int cost);

/* Block until all work queued on the context's stream has finished. */
int slam_synchronize(slam_ctx* ctx);

/* Page-locked host memory for result arrays (round 5).  The fetch entry points (slam_decompose_range_fetch, slam_fetch_results_range,
 * slam_v2_decompose_range ...) copy straight into the caller's arrays; when those are ordinary pageable memory the runtime pins and
 * unpins them around every copy, and FREEING such arrays later (munmap / heap trimming of tens of MB) was measured to stall every queue
 * of the process for 13-18 ms at the start of the next call (profiles/r5_api_timeline.txt).  Arrays from slam_host_alloc are visible to
 * every device, are copied into by DMA without staging, and are meant to be recycled by the caller (the Python binding keeps a pool).
 * Use them for ONE blocking call at a time (18.2 -> 16.9 ms for 65 536 x 32 sqrt(iSWAP) through the Python API); with several calls in
 * flight the device-side copy waits for wave slots behind the other calls' kernels and pageable arrays are faster (14.4 vs 14.9 ms per step).
 * Replaces nothing in the reference (its results are Python lists built on the host, src/slam/optimizer.py:113-119). */
int slam_host_alloc(size_t bytes, void** ptr);
int slam_host_free(void* ptr);

/* Accumulated kernel statistics since the last reset. */
int slam_get_stats(slam_ctx* ctx, slam_stats* out);
int slam_reset_stats(slam_ctx* ctx);

/* Device pointer + byte size of the resident per-target best-loss array
 * (double[n_targets]); used by the multi-GPU merge to hand the buffer to RCCL
 * without a host round trip. */
int slam_best_loss_device_ptr(slam_ctx* ctx, void** ptr, int64_t* n);

/* HIP device ordinal the context was created on (slam_comm_merge_add checks it against its communicator's). */
int slam_ctx_device(slam_ctx* ctx, int* device);

/*
 * Templates whose 2Q gates carry their own optimisable parameters -- CircuitTemplateV2 (src/slam/basisv2.py:27-299):
 * base_gates are gate classes / lambdas, every gate instance of the circuit gets its own "Q" parameters next to the "P"
 * parameters of the U gates, optionally box-bounded (add_bound, basisv2.py:174-190; the reference then switches SciPy to
 * L-BFGS-B, src/slam/optimizer.py:255-268).  Supported gate family: conversion-gain gates
 *     G(a, phi_c, b, phi_g) = exp(-i t H),  H = gc (e^{i phi_c} A B^+ + h.c.) + gg (e^{i phi_g} A B + h.c.),  a = gc t,  b = gg t
 * (ConversionGainGate, custom_gates.py:163-212; hamiltonian.py:84-111), which contains RiSwapGate(alpha) =
 * G(-pi alpha / 2, 0, 0, 0) (custom_gates.py:534-606).  A slam_v2_gate says how the four raw angles, in the order
 * (a, phi_c, b, phi_g), follow from the gate's n_params parameters q: raw[r] = scale[r] * q[sel[r]] + offset[r]
 * (sel[r] = -1: the constant offset[r]).  n_params is 1, 2 or 4 and the same for every gate of a template.
 *
 * Parameter vectors are in index order: P0 .. P{6(k+1)-1} as for fixed-gate templates, then the parameters of gate 1,
 * of gate 2, ...: n = 6 (k + 1) + n_params k.  Spans 1..3 for every n_params; span 4 for n_params <= 2 and span 5
 * (the reference's default maximum_span_guess, basisv2.py:35) for n_params = 1 -- as long as the packed inverse Hessian of
 * the n parameters fits one wavefront's registers (n <= 41); beyond that SLAM_ERR_UNSUPPORTED.
 *
 *   slam_v2_set_gates        the table of parametrised base gates (host side only; replaces nothing resident)
 *   slam_v2_eval_loss_grad   loss, gradient with respect to ALL n parameters (analytic, incl. the gate parameters) and
 *                            (optional) the template unitary of M parameter vectors -- objective_func
 *                            (optimizer.py:191-214) for a CircuitTemplateV2
 *   slam_v2_minimize_stage   one span stage: per active target `restarts` projected quasi-Newton minimisations
 *                            (BFGS metric, steps projected onto the box [bound_lo, bound_hi]; NULL bounds = none, which
 *                            is plain BFGS as in optimizer.py:255).  Start points: x0, or U[init_lo, init_hi) per
 *                            parameter from the Philox stream of params->seed (parameter_guess, basisv2.py:150-172:
 *                            the bound of a bounded parameter, else (-4 pi, 4 pi)).  The stage result is the
 *                            lowest-index restart below exit_loss, else the lowest loss -- the restart the reference's
 *                            sequential loop ends with (optimizer.py:281-295).  With SLAM_FLAG_EARLY_EXIT in
 *                            params->flags a restart is not started once a LOWER-index restart of its target has
 *                            ended below exit_loss (the loop's break); without it every restart runs to its end.
 *                            Persistent wavefronts over a restart-major queue, as slam_minimize_stage.
 *                            Outputs as in slam_minimize_stage with rows of n parameters.
 *   slam_v2_set_constraint   CircuitTemplateV2.set_constraint (basisv2.py:192-200: circuit_cost(x) <= param_max_cost, handed to
 *                            SciPy's SLSQP by optimizer.py:260-265) for circuit costs that are affine in the parameters over
 *                            the box: sum_i weights[i] x_i <= cost_max for the span-k template (n = its parameter count, index
 *                            order; RiSwapGate.cost() = alpha: weight 1 on every gate parameter).  Every later stage of that
 *                            span minimises the augmented Lagrangian loss + rho / 2 max(0, w.x - cost_max + mu / rho)^2 over the
 *                            box with the same projected quasi-Newton loop and updates the multiplier estimate mu per item
 *                            until w.x <= cost_max and mu (w.x - cost_max) = 0 to 1e-8; returned points are feasible, the
 *                            returned loss is the plain loss.
 *                            weights = NULL or n = 0 removes the constraint of span k (remove_constraint, basisv2.py:202-204);
 *                            slam_v2_set_gates removes all of them.
 */
#define SLAM_V2_MAX_SPAN 5 /* spans 4 and 5 for gates with few parameters: see slam_v2_minimize_stage */
typedef struct slam_v2_gate {
    int32_t n_params;
    int32_t sel[4];
    double scale[4];
    double offset[4];
} slam_v2_gate;
int slam_v2_set_gates(slam_ctx* ctx, const slam_v2_gate* gates, int32_t n_gates);
int slam_v2_set_constraint(slam_ctx* ctx, int k, const double* weights, int n, double cost_max);
int slam_v2_eval_loss_grad(slam_ctx* ctx, int k, const int32_t* gate_seq, const double* x, const int32_t* target_of, int64_t M,
                           double* loss, double* grad, double* unitary);
int slam_v2_minimize_stage(slam_ctx* ctx, int k, const int32_t* gate_seq, const int32_t* active, int64_t n_active, const double* x0,
                           const double* init_lo, const double* init_hi, const double* bound_lo, const double* bound_hi,
                           const slam_opt_params* params, double exit_loss, double* best_loss, double* best_x, int32_t* best_restart,
                           double* item_loss, int32_t* item_iters, int32_t* item_status, int32_t* item_evals);

/* The whole span loop of a CircuitTemplateV2 (TemplateOptimizer._run, src/slam/optimizer.py:233-303, with a V2 template) for the
 * resident targets [first, first + count), enqueued as one chain of kernels like slam_decompose_range: per span k = k_min..k_max the
 * V2 optimizer kernel over the targets still above success_threshold (ordered early exit per params->flags), the reduction over
 * restarts, the merge into the running best and the compaction -- no host round trip between the spans.
 *   gate_seqs                        concatenated gate sequences (k_min entries, then k_min + 1, ...), indices into slam_v2_set_gates' table
 *   init_lo/hi, bound_lo/hi          concatenated per span: n_k = 6 (k + 1) + n_params k entries for k = k_min, then k_min + 1, ...
 *                                    (bound_* may be NULL: no bounds)
 *   best_loss [count], best_cycles [count], best_x [count][n_kmax]: a target solved at span k has the n_k parameters of that
 *   span's layout in the front of its row, zeros behind (any of the three may be NULL). */
int slam_v2_decompose_range(slam_ctx* ctx, int64_t first, int64_t count, int k_min, int k_max, const int32_t* gate_seqs, const double* init_lo,
                            const double* init_hi, const double* bound_lo, const double* bound_hi, const slam_opt_params* params,
                            double success_threshold, double* best_loss, double* best_x, int32_t* best_cycles);

/* slam_v2_minimize_stage that also records every accepted iteration of every restart (use_callback=True for a
 * CircuitTemplateV2, src/slam/optimizer.py:217-224): trace_loss [M][trace_cap], trace_x [M][trace_cap][n] as
 * slam_minimize_stage_trace; rows no iteration reaches read as NaN. */
int slam_v2_minimize_stage_trace(slam_ctx* ctx, int k, const int32_t* gate_seq, const int32_t* active, int64_t n_active, const double* x0,
                                 const double* init_lo, const double* init_hi, const double* bound_lo, const double* bound_hi,
                                 const slam_opt_params* params, double exit_loss, int32_t trace_cap, double* best_loss, double* best_x,
                                 int32_t* best_restart, double* item_loss, int32_t* item_iters, int32_t* item_status, double* trace_loss,
                                 double* trace_x);

/*
 * Multi-GPU: one process per GPU, RCCL over xGMI, reached through this ABI (no torch, no MPI).  The path shards by
 * target, every rank keeps all restarts of its targets, so the only exchange is the FINAL min-all-reduce of the
 * best-loss vector -- the running minimum of TemplateOptimizer._run (src/slam/optimizer.py:281-284) taken over the
 * ranks -- plus a few scalars (barrier, slowest rank's time, counts).  librccl.so is dlopen'ed by the first of these
 * calls; a single-GPU process never loads it.
 *   slam_comm_get_unique_id   rank 0 creates the 128-byte id and hands it to the other ranks (file, env, socket:
 *                             the caller's business; bench.py / parallel.py use a file)
 *   slam_comm_init            ncclCommInitRank on `device` (collective: every rank calls it with the same id)
 *   slam_comm_allreduce_f64   in-place all-reduce of a small host array (staged through device memory)
 *   slam_comm_barrier         all ranks have arrived
 *   slam_comm_merge_begin     start a job-wide best-loss vector of n_global entries on this rank's GPU, all +inf
 *   slam_comm_merge_add       min-merge the context's RESIDENT best_loss[first_local, first_local + count) -- the
 *                             buffer slam_best_loss_device_ptr exposes -- into [first_global, first_global + count)
 *                             of that vector, device to device (call once per context / window of the rank).  The
 *                             context must live on the communicator's device (SLAM_ERR_INVALID otherwise); the read of
 *                             its buffer is complete when the call returns, so the context may be used again at once
 *   slam_comm_merge_add_host  the same from a host array (results that were fetched already)
 *   slam_allreduce_min        ncclAllReduce(min) of the vector in place over xGMI; n_below (may be NULL) = number of
 *                             entries < threshold counted on the device, merged (may be NULL) = host copy of the
 *                             n_global entries, merged_capacity = doubles it can hold (SLAM_ERR_INVALID if fewer)
 *   slam_comm_rank            rank / world size as RCCL ITSELF reports them (ncclCommUserRank / ncclCommCount);
 *                             SLAM_ERR_STATE if they differ from what slam_comm_init was given
 */
#define SLAM_COMM_ID_BYTES 128
#define SLAM_OP_SUM 0
#define SLAM_OP_MAX 2
#define SLAM_OP_MIN 3
typedef struct slam_comm slam_comm;
int slam_comm_get_unique_id(void* id);
int slam_comm_init(int device, int rank, int world, const void* id, slam_comm** out);
int slam_comm_destroy(slam_comm* comm);
int slam_comm_rank(slam_comm* comm, int* rank, int* world);
int slam_comm_allreduce_f64(slam_comm* comm, double* inout, int64_t n, int op);
int slam_comm_barrier(slam_comm* comm);
int slam_comm_merge_begin(slam_comm* comm, int64_t n_global);
int slam_comm_merge_add(slam_comm* comm, slam_ctx* ctx, int64_t first_local, int64_t count, int64_t first_global);
int slam_comm_merge_add_host(slam_comm* comm, const double* loss, int64_t count, int64_t first_global);
int slam_allreduce_min(slam_comm* comm, double threshold, int64_t* n_below, double* merged, int64_t merged_capacity);

/* Library version string. */
const char* slam_version(void);

/* Binary interface revision: bumped whenever an exported function changes its signature or a structure its layout.  History:
 *   4  slam_allreduce_min took its fifth argument, merged_capacity (round 3, without a new symbol: a caller built against an older
 *      header must check the revision before calling);
 *   5  round 4: slam_decompose_multi, slam_predict_spans, 32-byte item records;
 *   6  round 5: SLAM_MAX_SPAN_EVAL / SLAM_MAX_SPAN_MINIMIZE 5 -> 16 -- the per-span arrays of slam_stats and the rows of
 *      slam_fetch_span_losses grow with them --, SLAM_FLAG_NO_EXTERIOR, kernel_ms_span[0];
 *   7  round 5: slam_host_alloc / slam_host_free (new symbols only).
 * The Python binding refuses a library whose revision differs from the one it was written for. */
#define SLAM_ABI_VERSION 7
int slam_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SLAM_HIP_H */
