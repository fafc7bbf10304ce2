/*
 * ebo.h — C ABI of libebo_hip.so: the MI355X (gfx950) implementation of the
 * motion-compensated event-warping path of nurlanov-zh/event-based-odomety.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; no C++ types, no
 * torch types.  Nothing allocates across the ABI: the context owns every device
 * buffer, the caller owns every host pointer it passes in or gets results in.
 * Every function returns an int status (0 = ok, negative = error) and never
 * throws; ebo_last_error() gives the message.  A context belongs to one thread and
 * one device.  There is NO CPU fallback: without a gfx950 device ebo_create fails
 * with EBO_ERR_NO_DEVICE.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   R1  ceres::AutoDiffCostFunction<tracker::contrastFunctor,1,2>::Evaluate
 *       built at implementation/feature_tracker/src/feature_detector.cpp:359-363 on
 *       implementation/feature_tracker/include/feature_tracker/contrast_functor.h:10-292
 *   R2  tracker::FeatureDetector::compensateEventsContrast   feature_detector.cpp:298-464
 *   R3  tracker::FeatureDetector::integrateEvents            feature_detector.cpp:466-482
 *   R4  tracker::FeatureDetector::compensateEvents           feature_detector.cpp:243-296
 *   R5  tracker::Patch::integrateEvents                      implementation/feature_tracker/src/patch.cpp:65-85
 *   R6  tracker::Patch::integrateMotionCompensatedEvents     patch.cpp:87-130
 *   R7  tracker::DetectorParams                              include/feature_tracker/feature_detector.h:10-31
 *   R8  common::EventSample                                  common/include/common/data_types.h:12-38
 * The C++ façade with the reference's own names (tracker::FeatureDetector,
 * tracker::contrastFunctor, ...) sits on top of this header in
 * event-based-odomety_amd/include/; INTEGRATION.md shows the binding.
 */
#ifndef EBO_H
#define EBO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EBO_OK 0
#define EBO_ERR_ARG (-1)         /* null pointer, bad size, bad enum          */
#define EBO_ERR_HIP (-2)         /* a HIP runtime call failed                  */
#define EBO_ERR_RANGE (-3)       /* coordinate / timestamp outside packed range */
#define EBO_ERR_STATE (-4)       /* call order (no window set, ...)           */
#define EBO_ERR_UNSUPPORTED (-5) /* parameter combination not built            */
#define EBO_ERR_NO_DEVICE (-6)   /* no HIP device / not gfx950                 */
#define EBO_ERR_SOLVER (-7)      /* solver terminated with FAILURE             */
#define EBO_ERR_COMM (-8)        /* RCCL could not be loaded or a collective failed */

#define EBO_LOSS_EDGE 0     /* contrastFunctor::calculateEdgeLoss (reference default, :152-277) */
#define EBO_LOSS_VARIANCE 1 /* contrastFunctor::calculateVarianceLoss (:101-150, north-star objective) */

#define EBO_GRAD_JET 0     /* forward-mode dual numbers == ceres::Jet<double,2> (reference) */
#define EBO_GRAD_CENTRAL 1 /* central differences of the value-only objective */

#define EBO_SOLVE_GLOBAL 0      /* one problem over all patches incl. TV terms (reference, R2) */
#define EBO_SOLVE_INDEPENDENT 1 /* one 2-parameter problem per patch, solved on the device   */

#define EBO_COUNT_INTEGRATED 0 /* R3: un-warped counts                                  */
#define EBO_COUNT_WARPED 1     /* R2 final loop (:433-463): warp by per-patch flow, round() */
#define EBO_COUNT_FIELD 2      /* R4 warp loop (:270-295): warp by float32 per-pixel field  */

typedef struct ebo_ctx ebo_ctx;

/* R8: layout-compatible with common::EventSample on LP64
 * ({cv::Point2i{x,y}; enum EventPolarity sign; std::chrono::microseconds}). */
typedef struct ebo_event
{
	int32_t x;
	int32_t y;
	int32_t sign; /* -1 / +1 */
	int32_t reserved;
	int64_t t_us;
} ebo_event;

/* contrastFunctor's hard-coded members, contrast_functor.h:282-291. */
typedef struct ebo_functor_consts
{
	double max_possible_residual; /* 1e3 */
	double sigma_compensate;      /* 1   */
	int32_t kernel_compensate;    /* 3 (only 3 is built)  */
	int32_t kernel_st;            /* 3 (only 3 is built)  */
	double sigma_st;              /* 1.5 */
	int32_t kernel_nms;           /* 2 (only 2 is built)  */
	int32_t reserved;
} ebo_functor_consts;

/* R7: the DetectorParams fields this path reads, same defaults. */
typedef struct ebo_params
{
	int32_t device;           /* HIP device ordinal                         */
	int32_t image_w, image_h; /* imageSize {240,180}                        */
	int32_t patch_w, patch_h; /* patchCompensateSize {20,20}                */
	double tv_weight;         /* compensateTVweight 1e3                     */
	double tv_huber;          /* compensateTVHuberLoss 10                   */
	double scale;             /* compensateScale 1e-3                       */
	uint32_t min_events;      /* compensateMinNumEvents 100                 */
	int32_t loss;             /* EBO_LOSS_*  (default EDGE, as the reference) */
	int32_t grad;             /* EBO_GRAD_*  (default JET, as the reference)  */
	int32_t reserved;
	double fd_step;           /* step for EBO_GRAD_CENTRAL (default 1e-6)   */
	ebo_functor_consts k;
	uint64_t max_events;      /* device capacity, events over all windows of a batch (default 1<<20) */
	int32_t max_windows;      /* device capacity, windows per batch (default 1) */
	int32_t reserved2;
} ebo_params;

/* ceres::Solver::Options as set at feature_detector.cpp:401-410; unset fields
 * carry Ceres 2.0 defaults. */
typedef struct ebo_solver_opts
{
	int32_t max_num_iterations; /* 50    */
	int32_t use_nonmonotonic;   /* 1     */
	double function_tolerance;  /* 1e-12 */
	double gradient_tolerance;  /* 1e-12 */
	double parameter_tolerance; /* 1e-12 */
	double initial_radius;      /* 1e4   */
	double max_radius;          /* 1e16  */
	double min_radius;          /* 1e-32 */
	double min_relative_decrease; /* 1e-3 */
	double min_lm_diagonal;     /* 1e-6  */
	double max_lm_diagonal;     /* 1e32  */
	int32_t max_consecutive_nonmonotonic; /* 5 */
	int32_t max_consecutive_invalid;      /* 5 */
	int32_t jacobi_scaling;     /* 1 */
	int32_t mode;               /* EBO_SOLVE_* */
} ebo_solver_opts;

typedef struct ebo_summary
{
	int32_t iterations;     /* LM iterations (global) / max over patches (independent) */
	int32_t num_evals_cost; /* value-only evaluations of data terms     */
	int32_t num_evals_jac;  /* value+Jacobian evaluations of data terms */
	int32_t termination;    /* 0 convergence, 1 no convergence, 2 failure */
	double initial_cost;
	double final_cost;
} ebo_summary;

const char* ebo_version(void);
/* Number of HIP devices; 0 (and EBO_OK) when there is none. */
int ebo_device_count(int* n);
void ebo_default_params(ebo_params* p);
void ebo_default_solver(ebo_solver_opts* o);
/* Last error text of ctx (or of the calling thread's last failed ebo_create when ctx is NULL). */
const char* ebo_last_error(const ebo_ctx* ctx);

int ebo_create(const ebo_params* p, ebo_ctx** out);
void ebo_destroy(ebo_ctx* ctx);
/* Launch on this HIP stream (hipStream_t) instead of the context's own. */
int ebo_set_stream(ebo_ctx* ctx, void* hip_stream);
/* ---- HIP graphs over the asynchronous *_device calls ------------------------------------------
 * ebo_graph_begin starts recording the context's stream (hipStreamBeginCapture, thread-local mode);
 * the *_device calls made until ebo_graph_end (ebo_eval_device, ebo_solve_device,
 * ebo_count_image_device, ...) are recorded instead of run; ebo_graph_end instantiates the graph.
 * ebo_graph_launch replays it `times` times back to back on the context's stream, asynchronously,
 * with no host work between the steps.  Recorded calls must not allocate or synchronise: run the
 * same sequence once before recording (work tables are allocated on first use).  The graph reads
 * and writes the device pointers it was recorded with; ebo_set_windows / ebo_set_patches with the
 * same sizes reuse the same buffers, other changes need a new recording.  ebo_graph_end always
 * ends the recording, also after a failed call (EBO_ERR_HIP then, no graph).  Not in the reference:
 * the CPU path has no launch cost to hide. */
typedef struct ebo_graph ebo_graph;
int ebo_graph_begin(ebo_ctx* ctx);
int ebo_graph_end(ebo_ctx* ctx, ebo_graph** out);
int ebo_graph_launch(ebo_ctx* ctx, ebo_graph* graph, int times);
void ebo_graph_destroy(ebo_graph* graph);

int ebo_synchronize(ebo_ctx* ctx);

/* Patch grid of R2 (:301-346): npx*npy patches, last row/column absorb the remainder. */
int ebo_grid(const ebo_ctx* ctx, int* npx, int* npy);
int ebo_patch_rect(const ebo_ctx* ctx, int px, int py, int* x, int* y, int* w, int* h);

/* Load one window (the `events` list R2/R3/R4 receive), time-ordered as given.
 * The host buckets events by patch keeping their order (R2 :348-355), computes
 * the window's and each patch's reference time (:305-306, contrast_functor.h:18-20),
 * packs 8 B/event and uploads.  Replaces any previous window(s). */
int ebo_set_window(ebo_ctx* ctx, const ebo_event* ev, size_t n);
/* Batch of independent windows: window w = ev[offsets[w] .. offsets[w+1]). */
int ebo_set_windows(ebo_ctx* ctx, const ebo_event* ev, const size_t* offsets, int n_windows);
/* Same with the raw events already on the device (d_ev: ebo_event[] in device memory,
 * offsets: host, indices into d_ev): bucketing, reference times and packing all run on
 * the device; nothing but the 28-byte-per-patch table comes back.  ebo_set_window(s)
 * uses this path after one H2D copy of the raw events (EBO_BUCKET=host selects the host
 * counting sort instead). */
int ebo_set_windows_device(ebo_ctx* ctx, const ebo_event* d_ev, const size_t* offsets, int n_windows);
/* Compact raw event, 8 bytes -- a third of what common::EventSample moves over PCIe, which is most
 * of a window's set-up time: xy = x:15 | polarity:1 | y:15 | 0:1 (coordinates two's complement in
 * [-16384, 16383]; polarity 1 = +1), t_rel_us = t - t_base[window] in microseconds.  The packed
 * sidecar of a recording or a sensor driver can produce it directly; ebo_pack_events8 converts a
 * window of EventSamples (EBO_ERR_RANGE for a coordinate or a time that does not fit). */
typedef struct ebo_event8
{
	uint32_t xy;
	int32_t t_rel_us;
} ebo_event8;
int ebo_pack_events8(const ebo_event* ev, size_t n, int64_t t_base, ebo_event8* out);
/* ebo_set_windows on compact records: window w = ev[offsets[w] .. offsets[w+1]) with base time
 * t_base[w] (host arrays).  ev is host memory (page-locked memory -- hipHostMalloc /
 * hipHostRegister -- reaches the PCIe rate; the upload runs in groups of windows overlapped with
 * the bucketing of the groups before), or device memory for the _device form.  Results are
 * identical to ebo_set_windows on the same events. */
int ebo_set_windows8(ebo_ctx* ctx, const ebo_event8* ev, const int64_t* t_base, const size_t* offsets, int n_windows);
int ebo_set_windows8_device(ebo_ctx* ctx, const ebo_event8* d_ev, const int64_t* t_base, const size_t* offsets,
							int n_windows);
/* Arbitrary patches instead of a window: patch i = cv::Rect2i rects[i][4] = (x,y,w,h)
 * with its own event list ev[offsets[i]..offsets[i+1]) in list order, exactly what
 * tracker::contrastFunctor's constructor takes (contrast_functor.h:12-21; events
 * are NOT filtered by the rect, as there).  Afterwards ebo_eval / ebo_solve
 * (EBO_SOLVE_INDEPENDENT) / ebo_contrast_image address n_patches flows [n][2] with
 * window = 0.  n_patches <= max_windows * grid patches. */
int ebo_set_patches(ebo_ctx* ctx, const ebo_event* ev, const size_t* offsets, const int32_t* rects,
					int n_patches);
int ebo_num_windows(const ebo_ctx* ctx, int* n_windows);
int ebo_window_info(const ebo_ctx* ctx, int window, int64_t* t_ref_us, uint64_t* n_events);
/* n_events in the patch, active = (n_events > min_events), functor reference time. */
int ebo_patch_info(const ebo_ctx* ctx, int window, int patch, int32_t* n_events,
				   int32_t* active, int64_t* t_ref_us);

/* R1, batched over every patch of every loaded window: residual r and (if jac
 * != NULL) the 1x2 Jacobian at flows[w][p][0..1].  Inactive patches give 0.
 * Host pointers: flows [Wn][P][2], r [Wn][P], jac [Wn][P][2]. Synchronous.
 * Device memory: with EBO_LOSS_EDGE a Jacobian evaluation keeps a work table of 16 bytes per
 * LDS-resident canvas pixel and patch (reference defaults: 57.6 KB per patch; allocated on first
 * use, grown on demand, given back when later batches need under a quarter of it, freed by ebo_destroy).
 * It is never larger than 4 GiB (environment EBO_EDGE_CS_MB) nor than a quarter of the device memory that
 * is free at the time; beyond that, or when its allocation fails, the kernel re-derives what it would have
 * held -- same results to 1e-18 relative, ~7 % slower. */
int ebo_eval(ebo_ctx* ctx, const double* flows, double* r, double* jac);
/* Same on device pointers, asynchronous on the context's stream:
 * d_flows [Wn][P][2], d_out [Wn][P][3] = (r, J0, J1). */
int ebo_eval_device(ebo_ctx* ctx, const double* d_flows, int want_jac, double* d_out);
/* The image of warped events the functor builds for one patch at one flow
 * (contrast_functor.h:38-88): planar [channels][3h][3w], channels = 1 or 3. Diagnostic. */
int ebo_contrast_image(ebo_ctx* ctx, int window, int patch, const double* flow,
					   int channels, double* image);

/* Solve for the flows of every loaded window starting from 0 (R2 :316-414).
 * flows_out host [Wn][P][2]; summary [Wn] (may be NULL). */
int ebo_solve(ebo_ctx* ctx, const ebo_solver_opts* o, double* flows_out, ebo_summary* summary);
/* EBO_SOLVE_INDEPENDENT only: the whole solve in one launch, result left on the
 * device in d_flows_out [Wn][P][2]; asynchronous. d_stats (may be NULL) [Wn][P][4]
 * int32 = (iterations, value evals, jacobian evals, termination). */
int ebo_solve_device(ebo_ctx* ctx, const ebo_solver_opts* o, double* d_flows_out,
					 int32_t* d_stats);

/* The host solver of EBO_SOLVE_GLOBAL -- the trust-region LM over ONE problem per window (a contrast data
 * term per active patch + Huber-wrapped total-variation terms between grid neighbours, R2 :357-414) -- as the
 * resumable state machine ebo_solve drives internally, for a caller that must put something BETWEEN "evaluate"
 * and "step".  SURVEY 8(e), reference-faithful TV mode across GPUs: every rank evaluates the data terms of its
 * patch rows (ebo_set_patches + ebo_eval), ONE all-gather of (r, J0, J1) = 24 B per patch per evaluation, and
 * the same solver replicated on every rank takes the same step (tests/test_gpu_multiprocess.py: equal to the
 * one-process ebo_solve bit for bit).
 *   ebo_lm_create: npx x npy grid, active[p] != 0 iff patch p has a data term (n_events > min_events),
 *     tv_weight / tv_huber = DetectorParams::compensateTVweight / compensateTVHuberLoss, o as ebo_solve.
 *   ebo_lm_request(flows [P][2]): the point the data terms are wanted at; returns 1 = residuals AND Jacobians
 *     wanted, 2 = residuals only, 0 = the solve has finished (flows untouched), < 0 = EBO_ERR_*.
 *   ebo_lm_supply(r [P], jac [P][2]): the data terms at that point (inactive patches ignored; jac may be NULL
 *     when only residuals were wanted).  EBO_ERR_STATE after the solve has finished.
 *   ebo_lm_result: the solution (the lowest-cost point visited, as Ceres returns it) and the summary
 *     (num_evals_* count evaluation ROUNDS of the whole problem here).
 * Host only: no device is touched, no context is needed. */
typedef struct ebo_lm ebo_lm;
int ebo_lm_create(int npx, int npy, const uint8_t* active, double tv_weight, double tv_huber, const ebo_solver_opts* o,
				  ebo_lm** out);
int ebo_lm_request(ebo_lm* lm, double* flows);
int ebo_lm_supply(ebo_lm* lm, const double* r, const double* jac);
int ebo_lm_result(const ebo_lm* lm, double* flows, ebo_summary* summary);
void ebo_lm_destroy(ebo_lm* lm);

/* Integer-valued event-count images (CV_64F in the reference), host out
 * [Wn][image_h][image_w].  aux: EBO_COUNT_WARPED -> host flows [Wn][P][2];
 * EBO_COUNT_FIELD -> host float32 field [Wn][image_h][image_w][2]; else NULL. */
int ebo_count_image(ebo_ctx* ctx, int mode, const void* aux, double* image);
/* Same with aux and image on the device; asynchronous. */
int ebo_count_image_device(ebo_ctx* ctx, int mode, const void* d_aux, double* d_image);

/* tracker::FeatureDetector::initMotionField (feature_detector.cpp:53-142): per-pixel motion
 * field from the trajectories of the tracked feature patches.  Patch k has trajectory samples
 * traj_xy[traj_offsets[k]..traj_offsets[k+1])[2], traj_t[...] (us, ascending).  The velocity of
 * the segment at `timestamp` (std::lower_bound) is stored at the rounded trajectory point; the
 * other pixels get the average (use_average, DetectorParams::useAverageFlow) or the nearest
 * fixed point's value.  field_out: host float32 [image_h][image_w][2] (may be NULL);
 * fixed_xy: host [n_patches][2] (may be NULL).  The field also stays on the device:
 * ebo_count_image(ctx, EBO_COUNT_FIELD, NULL, image) then is R4 (compensateEvents) end to end.
 * ebo_interpolate_motion_field applies interpolateMotionField's TV smoothing to it. */
int ebo_init_motion_field(ebo_ctx* ctx, int64_t timestamp, int use_average, int n_patches,
						  const size_t* traj_offsets, const double* traj_xy, const int64_t* traj_t,
						  float* field_out, int32_t* n_fixed, int32_t* fixed_xy);

/* tracker::FeatureDetector::interpolateMotionField (feature_detector.cpp:144-241) after its
 * initMotionField call, i.e. on the field ebo_init_motion_field left on the device: if
 * cv::norm(field) > 0 (:152), the per-pixel problem of :154-214 (totalVarianceFunctor, weight 1,
 * between every pixel and its right / lower neighbour for x < w-1, y < h-1; HuberLoss(1e-5) when
 * use_l1 = DetectorParams::useL1; fixed points constant) is solved by the same trust-region LM
 * as ceres::Solve with the options of :216-222 (opts == NULL) and the field is overwritten with
 * the result rounded to float32 (:230-239).  All per-pixel work runs on the device; the linear
 * solve of each LM iteration is a multigrid-preconditioned conjugate-gradient run to 1e-13
 * relative residual.
 * field_out: host float32 [image_h][image_w][2] (may be NULL; the field stays resident for
 * ebo_count_image(EBO_COUNT_FIELD)).  cg_iterations (may be NULL): total CG iterations.
 * EBO_ERR_STATE without a prior ebo_init_motion_field; EBO_ERR_RANGE for a fixed point at pixel
 * (w-1, h-1), which is no parameter block of that problem (the reference aborts inside Ceres). */
int ebo_interpolate_motion_field(ebo_ctx* ctx, int use_l1, const ebo_solver_opts* opts, float* field_out,
								 ebo_summary* summary, int32_t* cg_iterations);

/* Packed binary sidecar of an events.txt: 32-byte header + 16 bytes per event
 * {int64 t_us (the reader's truncated microseconds), int16 x, int16 y, int8 sign, 3 x 0}; reading
 * it back yields exactly the events ebo_read_events_txt parsed, at memory speed instead of ~100 ns
 * of strtod per event.  Host only.  EBO_ERR_RANGE: coordinates beyond int16 / a sign other than
 * -1, +1 (write); bad magic, truncated file or bad sign (read; events before it are kept). */
int ebo_write_events_bin(const char* path, const ebo_event* ev, size_t n);
int ebo_read_events_bin(const char* path, ebo_event* out, size_t cap, size_t* n);

/* ---- per-feature tracker objective --------------------------------------
 * tracker::Optimizer::setGrad (optimizer.cpp:15-31): the image-gradient grid the tracker samples
 * with ceres::BiCubicInterpolator.  grad_x, grad_y: host [image_h][image_w] (CV_64F). */
int ebo_optimizer_set_grad(ebo_ctx* ctx, const double* grad_x, const double* grad_y);
/* ceres::Solver::Options as Optimizer::optimize sets them (optimizer.cpp:103-112;
 * OptimizerParams::maxNumIterations = 10). */
void ebo_optimizer_default_solver(ebo_solver_opts* o);
/* tracker::OptimizerCostFunctor (optimizer_cost.h:15-96) behind
 * ceres::AutoDiffCostFunction<OptimizerCostFunctor, ceres::DYNAMIC, 4, 1>::Evaluate
 * (optimizer.cpp:89-97), for n tracked patches in one launch.  rects [n][4] = cv::Rect2d
 * (x, y, width, height); patch i has m_i = int(width) * int(height) residuals, stored
 * consecutively; nabla: Patch::getNormalizedIntegratedNabla per patch ([m_i], row-major);
 * poses [n][4] = Sophus::SE2d::data() (cos, sin, tx, ty); flow_dirs [n].
 * residuals [sum m_i]; jac_pose [sum m_i][4] and jac_flow [sum m_i] are the Jacobians Ceres asks
 * for (row-major, w.r.t. the 4 stored SE2 parameters and the flow angle), both NULL = value only
 * (the double path of the functor, whose quotient rounds differently from the Jet path, as in
 * the reference).  Needs ebo_optimizer_set_grad (EBO_ERR_STATE). */
int ebo_optimizer_eval(ebo_ctx* ctx, int n, const double* rects, const double* nabla, const double* poses,
					   const double* flow_dirs, double* residuals, double* jac_pose, double* jac_flow);
/* The ceres::Solve of tracker::Optimizer::optimize (optimizer.cpp:83-119) for n tracked patches
 * in one launch: SE2 block with Sophus' LocalParameterizationSE2 + the flow angle, one residual
 * block with ceres::HuberLoss(huber) (OptimizerParams::huberLoss = 0.3), trust-region LM with the
 * options of :103-112 (opts == NULL; DENSE_QR there, a 4x4 Cholesky of the same normal equations
 * here).  normalize != 0: nabla is Patch::getIntegratedNabla and is normalised on the device as
 * Patch::getNormalizedIntegratedNabla does (patch.cpp:156-159; an all-zero patch gives NaN and
 * termination 2, parameters unchanged).  poses, flow_dirs: in = Patch::getWarp / getFlow, out =
 * the lowest-cost point visited.  summaries [n] may be NULL. */
int ebo_optimizer_solve(ebo_ctx* ctx, int n, const double* rects, const double* nabla, int normalize, double huber,
						const ebo_solver_opts* opts, double* poses, double* flow_dirs, ebo_summary* summaries);

/* tracker::Optimizer::drawCostMap (optimizer.cpp:33-60; OptimizerParams::drawCostMap, costMapWidth / costMapHeight) for n
 * tracked patches in one launch: cost_maps [n][map_h][map_w], cell (y + (map_h-1)/2, x + (map_w-1)/2) = cv::norm(image, NORM_L2)
 * of the functor's residual image (its double path, as `(*c)(poseNew.data(), &flowDir, image.data)` evaluates it) at
 * poseNew = SE2(pose.log().z(), (float(x) + tx, float(y) + ty)), x = -(map_w-1)/2 .. (map_w-1)/2, y likewise (cells an even
 * size leaves unvisited stay 0, as in cv::Mat::zeros).  rects / nabla / normalize / poses / flow_dirs as for
 * ebo_optimizer_solve; the reference calls it after the solve with the functor built BEFORE it (the rect and the
 * normalised nabla of the optimisation) and the solved pose and flow.  EBO_ERR_STATE without ebo_optimizer_set_grad. */
int ebo_optimizer_cost_map(ebo_ctx* ctx, int n, const double* rects, const double* nabla, int normalize, const double* poses,
						   const double* flow_dirs, int map_w, int map_h, double* cost_maps);

/* The event-count estimate of FeatureDetector::updateNumOfEvents (feature_detector.cpp:689-707) for n
 * tracked patches in one launch: the L1 norm over the patch rect of
 * 0.6 gradX' cos(flow) + 0.6 gradY' sin(flow), gradX' / gradY' = the gradient images of
 * ebo_optimizer_set_grad warped by cv::warpAffine(..., patch.getWarp().matrix2x3(),
 * cv::WARP_INVERSE_MAP) -- flags without interpolation bits: INTER_NEAREST through OpenCV's 10-bit
 * fixed-point map, BORDER_CONSTANT 0 -- truncated to an integer as `size_t sumPatch = cv::norm(..)`
 * does.  rects [n][4] = cv::Rect2d, poses [n][4] = Sophus::SE2d::data(), flow_dirs [n], out [n].
 * The two border branches of updateNumOfEvents (:668-687) are the caller's (the facade has them).
 * EBO_ERR_STATE without ebo_optimizer_set_grad. */
int ebo_estimate_num_events(ebo_ctx* ctx, int n, const double* rects, const double* poses, const double* flow_dirs,
							uint64_t* out);

/* tracker::Patch::warpImage (patch.cpp:132-154; called at feature_detector.cpp:508,615) for n tracked
 * patches in one launch: predictedNabla_ = -gradX'(patch_) cos(flowDir_) - gradY'(patch_) sin(flowDir_) with
 * gradX' / gradY' = the gradient images of ebo_optimizer_set_grad warped by cv::warpAffine(...,
 * warp_.matrix2x3(), cv::WARP_INVERSE_MAP) (INTER_NEAREST through OpenCV's 10-bit fixed-point map,
 * BORDER_CONSTANT 0, as ebo_estimate_num_events) and flowDir_ the patch's double member.  rects [n][4] =
 * cv::Rect2d, poses [n][4] = Sophus::SE2d::data(), flow_dirs [n]; patch i's [cvRound(h)][cvRound(w)] image
 * goes to predicted + nabla_offsets[i].  updated[i] = 0 and nothing is written for a patch whose rect
 * touches the image border (the reference returns early at :145-150 and keeps the old predictedNabla_).
 * EBO_ERR_STATE without ebo_optimizer_set_grad. */
int ebo_patch_warp_image(ebo_ctx* ctx, int n, const double* rects, const double* poses, const double* flow_dirs,
						 const size_t* nabla_offsets, double* predicted, int32_t* updated);

/* Event -> tracked-patch routing: what FeatureDetector::updatePatches does per event,
 * `if (patch.isInPatch(event.value.point)) patch.addEvent(event)` (feature_detector.cpp:585-596;
 * cv::Rect2d::contains on the integer point: x <= px < x + w, y <= py < y + h in double), for a
 * whole chunk of the stream and all tracked patches in one launch instead of one host test per
 * (event, patch).  ebo_route_set_events keeps the chunk's coordinates on the device (4 B/event);
 * ebo_route_events then gives, for patch i with rect rects[i][4], the indices (ascending = stream
 * order) of the first max_take[i] events at or after start[i] that fall inside it:
 * out_index[i * cap + k], k < out_count[i] <= min(max_take[i], cap), and out_next[i] = index after
 * the last event taken when the quota was reached, else n (chunk exhausted).  A patch's rect moves
 * when it is optimised, so a caller routes up to the event that makes a patch ready, optimises,
 * and routes again from out_next with the new rect (tracker::FeatureDetector::updatePatches(chunk)
 * in the facade does exactly that, all patches in lock step). */
int ebo_route_set_events(ebo_ctx* ctx, const ebo_event* ev, size_t n);
int ebo_route_events(ebo_ctx* ctx, int n_patches, const double* rects, const uint32_t* start,
					 const uint32_t* max_take, uint32_t cap, uint32_t* out_index, uint32_t* out_count,
					 uint32_t* out_next);

/* Diagnostic (bench.py's edge_roofline): ONE evaluation of the edge loss at d_flows (device, [Wn][P][2]) that
 * also counts the work it is made of, summed over the units that pass the empty-window test of
 * contrast_functor.h:159-165 -- out[0] units, [1] their events, [2] pixels of their bounding boxes, [3] pixels
 * of the eigenvalue region, [4] non-maximum-suppression windows, [5] argmax entries of the reverse pass
 * (0 without want_jac).  DESIGN.md 4.5 turns them into useful f64 operations.  Synchronous; the
 * evaluation's (r, J) go to the context's own result buffer. */
int ebo_edge_work_stats(ebo_ctx* ctx, const double* d_flows, int want_jac, uint64_t* out);

/* Diagnostic (bench.py's roofline.lds): the chip-wide rates of 64-bit LDS atomic adds (gops[0]) and single 64-bit LDS reads
 * (gops[1]) in the evaluation kernels' own access shape -- per lane a random base slot, then the 49 taps of a 7 x 7 footprint
 * at immediate offsets --, in 1e9 operations per second, measured NOW on the context's device (4 workgroups of 256 lanes
 * per CU, 42 footprints per lane, best of three; tools/microbench/lds_atomics.hip "7x7 taps, any base"): the rates the
 * scatter and the gather pass of the variance evaluation are priced against.  Synchronous. */
int ebo_lds_rates(ebo_ctx* ctx, double* gops);

/* Diagnostic (bench.py): the traffic of ebo_count_image_device with no work -- every packed event of the
 * loaded windows read once (16-byte loads), every pixel of d_image [Wn][image_h][image_w] written once
 * (16-byte stores, all 0.0) -- as the in-run yardstick of the HBM-bound count kernels: what a plain stream
 * of these bytes reaches on this GPU.  *bytes (may be NULL) receives the bytes moved.  Asynchronous. */
int ebo_stream_yardstick_device(ebo_ctx* ctx, double* d_image, uint64_t* bytes);

/* R2's final loop (:433-463) for windows whose PATCHES ARE SHARDED over ranks (BASELINE
 * config 4).  The context was loaded by ebo_set_patches with n_windows equal groups of units (group w =
 * this rank's patches of window w with the events inside them); flows_grid [n_windows][P][2] are the flows
 * of ALL P patches of the context's grid (every rank has them after the all-gather of the solved flows);
 * window_t_ref_us [n_windows] (host) is each WINDOW's reference time (:305-306), which a shard cannot
 * derive from its own events: ebo_window_ref_time(first, last) of the whole window's first / last event
 * time.  image [n_windows][image_h][image_w] receives the counts of THIS context's events only, each
 * warped by the flow of the grid patch its own coordinates select (:436-441); the sum of the ranks'
 * images (ebo_reduce_sum_device, or any reduce: integer-valued doubles add exactly) is the image
 * ebo_compensate_events_contrast returns for the whole window.  _device: pointers on the device,
 * asynchronous.  EBO_ERR_STATE without ebo_set_patches. */
int ebo_window_ref_time(int64_t t_first_us, int64_t t_last_us, int64_t* t_ref_us);
int ebo_count_image_shard(ebo_ctx* ctx, int n_windows, const int64_t* window_t_ref_us, const double* flows_grid,
						  double* image);
int ebo_count_image_shard_device(ebo_ctx* ctx, int n_windows, const int64_t* window_t_ref_us,
								 const double* d_flows_grid, double* d_image);

/* The same image BAND-LIMITED ("each GPU writes its own row-block; events warped across a
 * shard border need a halo"): a rank's events start in the image rows of its patch rows [own_row0,
 * own_row1) and leave them only by max|dt| * scale * |flow|, so the rank counts them into a band = its own
 * rows + `halo` rows above and below (clipped to the image), keeps the own rows and sends only the halo
 * rows to the two neighbouring ranks -- instead of reducing a full image per window from every rank onto one.
 *   ebo_band_plan            host only.  row_bounds [nranks + 1]: rank q owns image rows [row_bounds[q],
 *                            row_bounds[q + 1]) (0 ... image_h, rank order = row order).  Fills *out for `rank`.
 *                            EBO_ERR_UNSUPPORTED when some rank's halo would not fit inside its neighbour's rows
 *                            (the same verdict on every rank: all decide from the same numbers).
 *   ebo_count_image_band_device   counts this context's events (ebo_set_patches, as for ebo_count_image_shard)
 *                            into d_top [n_windows][own_row0 - band_row0][W], d_own [n_windows][own rows][W],
 *                            d_bottom [n_windows][band_row1 - own_row1][W] (uint32 counts; LDS tiles, no global
 *                            atomics).  *d_escaped (int32, device) becomes 1 when a unit's flow could carry an event
 *                            to an image row outside the band (then the caller falls back to
 *                            ebo_count_image_shard + reduce), else 0.  EBO_ERR_UNSUPPORTED when a unit's events select
 *                            more than one grid patch (arbitrary rects: use ebo_count_image_shard).  Asynchronous.
 *   ebo_band_exchange_device (ebo_comm_init) d_top -> rank - 1, d_bottom -> rank + 1, the neighbours' halos into
 *                            d_from_above [n_windows][recv_above][W] and d_from_below [n_windows][recv_below][W]
 *                            (one grouped ncclSend / ncclRecv), then *d_escaped = max over the ranks (ncclAllReduce of
 *                            one int): every rank takes the same fallback decision.  Without a communicator: no-op.
 *   ebo_band_finish_device   d_image_own [n_windows][own rows][W] (CV_64F) = own + received halos.
 *   ebo_band_gather_device   (ebo_comm_init) only when one rank wants the whole image: every rank's d_image_own into
 *                            d_full [n_windows][image_h][W] on `root` (grouped send / recv of the owned rows only).
 * Bytes a rank sends per window: (rows of top + rows of bottom) x W x 4. */
typedef struct ebo_band
{
	int band_row0, own_row0, own_row1, band_row1; /* image rows: band = [band_row0, band_row1), own inside it */
	int recv_above, recv_below;                   /* rows of the own region the neighbours' halos cover */
} ebo_band;
int ebo_band_plan(int image_h, const int* row_bounds, int nranks, int rank, int halo, ebo_band* out);
int ebo_count_image_band_device(ebo_ctx* ctx, int n_windows, const int64_t* window_t_ref_us, const double* d_flows_grid,
								const ebo_band* band, uint32_t* d_top, uint32_t* d_own, uint32_t* d_bottom, int32_t* d_escaped);
int ebo_band_exchange_device(ebo_ctx* ctx, int n_windows, const ebo_band* band, const uint32_t* d_top, const uint32_t* d_bottom,
							 uint32_t* d_from_above, uint32_t* d_from_below, int32_t* d_escaped);
int ebo_band_finish_device(ebo_ctx* ctx, int n_windows, const ebo_band* band, const uint32_t* d_own, const uint32_t* d_from_above,
						   const uint32_t* d_from_below, double* d_image_own);
int ebo_band_gather_device(ebo_ctx* ctx, int n_windows, const int* row_bounds, const double* d_image_own, int root, double* d_full);

/* R2 in one call: set window, solve, final warped count image.
 * flows_out [P][2], image_out [image_h][image_w] (may be NULL). */
int ebo_compensate_events_contrast(ebo_ctx* ctx, const ebo_event* ev, size_t n,
								   const ebo_solver_opts* o, double* flows_out,
								   double* image_out, ebo_summary* summary);

/* R5/R6 batched over tracked feature patches.  Patch i owns events
 * ev[offsets[i]..offsets[i+1]) in deque order (front = newest), a cv::Rect2d
 * rects[i][4] = (x,y,w,h), and writes a [int(h)][int(w)] signed count image at
 * nabla + nabla_offsets[i].  For R6, traj[i][6] = (prelast x,y,t_us, last x,y,t_us)
 * and mid_time[i]; updated[i] tells whether R6's time test passed. */
int ebo_patch_integrate(ebo_ctx* ctx, const ebo_event* ev, const size_t* offsets,
						int n_patches, const double* rects, const size_t* nabla_offsets,
						double* nabla, int64_t* current_ts, int64_t* time_last_update);
int ebo_patch_integrate_mc(ebo_ctx* ctx, const ebo_event* ev, const size_t* offsets,
						   int n_patches, const double* rects, const double* traj,
						   const int64_t* mid_time, const size_t* nabla_offsets,
						   double* nabla, int32_t* updated);

/* DAVIS240C events.txt reader, Davis240cReader::getEventSample
 * (tools/dataset_reader/src/davis240c_reader.cpp:60-92): "<seconds> <x> <y> <0|1>" per
 * line -> out[0..*n), at most cap events.  Host only; EBO_ERR_RANGE on a malformed line or a
 * sign other than 0/1 (the reference throws there), events parsed before it are kept. */
int ebo_read_events_txt(const char* path, ebo_event* out, size_t cap, size_t* n);
/* The same in pieces, as Davis240cReader::getEvents reads a recording (davis240c_reader.cpp:186-212: EVENT_LENGTH =
 * 1 000 000 lines per call, the next call continues behind them): at most cap events from byte *offset of the file on;
 * *offset moves behind the last line taken (start with 0; *n == 0 with EBO_OK = the end of the file). */
int ebo_read_events_txt_at(const char* path, uint64_t* offset, ebo_event* out, size_t cap, size_t* n);
/* Both of them parse on the host's threads since round 5, as the reference's reader does (DatasetReader::readFile,
 * tools/dataset_reader/include/dataset_reader/dataset_reader.h:33-97: hardware_concurrency() threads over the mapped
 * file): the mapped byte range is cut at line breaks into one chunk per thread (EBO_HOST_THREADS, default: the
 * machine's hardware threads).  Events, *n, *offset and the error are those of a single-thread walk whatever the
 * thread count.  This form takes the thread count from the caller (0: the default; offset may be NULL) and reports
 * how many threads parsed (threads_used, may be NULL). */
int ebo_read_events_txt_threads(const char* path, uint64_t* offset, ebo_event* out, size_t cap, size_t* n, int threads,
								int* threads_used);
/* The same reader writing the compact 8-byte records ebo_set_windows8 takes (a third of the bytes; no 24-byte array
 * between the text and the device): *t_base = the time stamp of the first event of THIS call, every record's t_rel_us is
 * relative to it -- pass it as t_base[w] of every window cut from the call's events.  offset may be NULL (from the start);
 * threads 0 = the default.  EBO_ERR_RANGE also for an event a compact record cannot hold (a coordinate beyond +-16384, a
 * time further than 2^31 us from the base): it stays in front of that line, as for a malformed one. */
int ebo_read_events_txt8(const char* path, uint64_t* offset, ebo_event8* out, size_t cap, size_t* n, int64_t* t_base, int threads);

/* Contiguous shard [begin,end) of n_units for rank of world (multi-GPU, §8e). */
int ebo_shard_range(int n_units, int rank, int world, int* begin, int* end);

/* Multi-GPU exchange (one process per GPU, RCCL over xGMI) for callers without a framework:
 * rank 0 makes an id (ebo_comm_unique_id) and hands its 128 bytes to the other ranks by any
 * means; every rank calls ebo_comm_init on its context; ebo_allgather_device gathers
 * count_per_rank doubles from every rank into d_recv [nranks][count_per_rank] on every rank,
 * asynchronously on the context's stream -- the single all-gather of the solved flows (or of
 * the (r, J0, J1) triples) of SURVEY 8(e).  librccl.so is loaded on first use only. */
typedef struct ebo_comm_id
{
	char internal[128];
} ebo_comm_id;
int ebo_comm_unique_id(ebo_comm_id* id);
int ebo_comm_init(ebo_ctx* ctx, const ebo_comm_id* id, int rank, int nranks);
int ebo_allgather_device(ebo_ctx* ctx, const double* d_send, double* d_recv, size_t count_per_rank);
/* Rank and size of the context's communicator (0 and 1 without one). */
int ebo_comm_size(const ebo_ctx* ctx, int* rank, int* nranks);
/* Element-wise sum over the ranks of count doubles, asynchronously on the context's stream: into d_recv on
 * `root` (ncclReduce; d_recv may be NULL elsewhere) or, root < 0, on every rank (ncclAllReduce).  The one
 * reduce of SURVEY 8(e)'s "final full-frame count image": the per-rank partial images of
 * ebo_count_image_shard are integer-valued doubles, so the sum is exact in whatever order RCCL adds. */
int ebo_reduce_sum_device(ebo_ctx* ctx, const double* d_send, double* d_recv, size_t count, int root);
int ebo_comm_destroy(ebo_ctx* ctx);

/* One sample of a tracked feature's trajectory, the record tools::Evaluator::saveFeaturesTrajectory
 * writes as "feature_id timestamp x y" (tools/evaluator/src/evaluator.cpp:125-150): Patch::getTrackId()
 * and one common::Sample<Point2d> of Patch::getTrajectory(). */
typedef struct ebo_track_point
{
	int64_t id;   /* Patch::getTrackId()                 */
	int64_t t_us; /* pos.timestamp (microseconds)        */
	double x, y;  /* pos.value                           */
} ebo_track_point;

/* The exchange of BASELINE config 5 (independent sequences, one per GPU): every rank
 * contributes the n_local track points of ITS sequence (host memory, any count, 0 allowed) and
 * every rank receives all of them, rank 0's first, each rank's in the order given.  Two
 * collectives on the context's stream: an all-gather of the counts, then ONE ncclAllGather of
 * max-count-padded 32-byte records; the padding is dropped on the way back to the host.
 * all [cap] (host) receives *n_all records; counts [nranks = ebo_comm_size] (may be NULL) the per-rank
 * counts.  EBO_ERR_STATE without ebo_comm_init; EBO_ERR_ARG when cap is too small (*n_all and counts are
 * still set, nothing is written to all).  Synchronous.  Every rank of the communicator must call it, and a
 * rank whose cap is too small still takes part in BOTH collectives before it returns the error, so the
 * others never wait for it; size the buffer with ebo_allgather_track_counts (the counts collective alone,
 * also on every rank) rather than with a cap = 0 call. */
int ebo_allgather_tracks(ebo_ctx* ctx, const ebo_track_point* local, size_t n_local, ebo_track_point* all,
						 size_t cap, size_t* n_all, size_t* counts);
int ebo_allgather_track_counts(ebo_ctx* ctx, size_t n_local, size_t* n_all, size_t* counts);

/* trajectory.txt as saveFeaturesTrajectory writes it (evaluator.cpp:125-150): one line
 * "<id> <seconds> <x> <y>" per point, std::fixed with 8 decimals, seconds =
 * std::chrono::duration<double>(timestamp).  Host only.  ebo_read_tracks_txt parses that format
 * back (seconds -> microseconds rounded to nearest); EBO_ERR_RANGE on a malformed line (*n = the records
 * before it); EBO_ERR_ARG when the file holds more than cap records (*n = the number it holds, out = the
 * first cap of them): a list is never truncated silently. */
int ebo_write_tracks_txt(const char* path, const ebo_track_point* pts, size_t n);
int ebo_read_tracks_txt(const char* path, ebo_track_point* out, size_t cap, size_t* n);

/* Device-side timing of everything enqueued between begin and end on the
 * context's stream (hipEvent based). */
int ebo_timer_begin(ebo_ctx* ctx);
int ebo_timer_end(ebo_ctx* ctx, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* EBO_H */
