/*
 * picp_hip.h -- C ABI of libpicp_hip.so: the projective-ICP pose loop and the appearance
 * association of ValerioSpagnoli/Monocular-Visual-Odometry as hand-written HIP kernels for
 * gfx950 (MI355X).
 *
 * The reference is pure Python and has no FFI; its operator interface for this path is two
 * Python callables (src/ProjectiveICP.py:8-381, src/data_association.py:3-57).  Every entry
 * point below names the reference lines whose behaviour it reproduces.  The package's Python
 * classes of the same names bind these symbols through ctypes.
 *
 * Conventions
 *  - every function returns PICP_OK (0) or a negative PICP_ERR_* code and never throws;
 *    picp_last_error() returns a human-readable message for the calling thread's last failure;
 *  - "device" pointers are HIP device addresses (e.g. torch.Tensor.data_ptr()); the caller owns
 *    every buffer it passes; "host" pointers are ordinary process memory;
 *  - all floating point data is IEEE float64; matrices are row-major; poses are 4x4 with last
 *    row (0,0,0,1), as every pose the reference builds (src/geometry_utils.py:38-57);
 *  - one picp_ctx per device and stream; a ctx is not thread-safe; work is enqueued on the ctx's
 *    stream and, unless a function says it synchronises, returns before the GPU has finished;
 *  - correspondences are struct-of-arrays: u[M], v[M] (measured pixel), x[M], y[M], z[M] (map
 *    point in the frame poses are expressed in), so that consecutive lanes read consecutive
 *    8-byte words.
 */
#ifndef PICP_HIP_H
#define PICP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PICP_VERSION 100

#define PICP_OK 0
#define PICP_ERR_INVALID (-1)   /* bad argument (null pointer, negative size, pose last row != 0,0,0,1) */
#define PICP_ERR_HIP (-2)       /* a HIP runtime call failed; see picp_last_error() */
#define PICP_ERR_CAPACITY (-3)  /* an output buffer is too small; the needed size is reported */
#define PICP_ERR_NO_DEVICE (-4) /* no usable gfx950 device */
#define PICP_ERR_COMM (-5)      /* RCCL could not be loaded, or one of its calls failed */
#define PICP_ERR_RETRY (-6)     /* an enqueued loop gave up on a shared device: run it again with picp_run */

/* Doubles per row of the per-iteration history and of the device state block. */
#define PICP_HIST_STRIDE 80
#define PICP_STATE_DOUBLES 96
/* Doubles in the stage-1 / stage-2 reduction records (the all-reduce payloads when sharded). */
#define PICP_S1_DOUBLES 4
#define PICP_S2_DOUBLES 32

/* History row layout (doubles).  Rows 0..19 are the reference's five per-iteration lists
 * (src/ProjectiveICP.py:173, :233-237); the rest is recorded for parity checks. */
#define PICP_H_T 0            /* [16] pose w_T_c after this iteration's step            (:233) */
#define PICP_H_ERROR 16       /* mean chi of the final inlier set, +inf if none          (:234) */
#define PICP_H_NUM_INLIERS 17 /*                                                         (:235) */
#define PICP_H_KERNEL_THR 18  /* kernel threshold AFTER this iteration's update          (:236) */
#define PICP_H_DAMPING 19     /* damping AFTER this iteration's update                   (:237) */
#define PICP_H_N_INSIDE 20    /* pairs whose map point projects inside the image (len(reference_image_points), :200) */
#define PICP_H_DONE 21        /* computation_done                                        (:263, :273) */
#define PICP_H_N_STAGE1 22    /* survivors of chi <= kernel_threshold                    (:294) */
#define PICP_H_N_STAGE2 23    /* survivors of chi < mean                                 (:302) */
#define PICP_H_USED_STAGE2 24 /* 1 when the stage-2 set replaced the stage-1 set         (:307) */
#define PICP_H_MEAN_STAGE1 25 /* mean chi of the stage-1 set                             (:301) */
#define PICP_H_DX 26          /* [6] solution of the damped system                       (:270) */
#define PICP_H_B 32           /* [6] b before the solve                                  (:317) */
#define PICP_H_H 38           /* [36] H before damping, row-major                        (:316) */

/* Camera model: K and the image size used by Camera.project_point (src/Camera.py:82-121).
 * The reference parses K with int() (src/Camera.py:23); pass the values as doubles. */
typedef struct picp_camera {
    double K[9];
    double width;  /* u must satisfy 0 <= u < width   (src/Camera.py:117) */
    double height; /* v must satisfy 0 <= v < height  (src/Camera.py:118) */
} picp_camera;

/* Loop parameters = ProjectiveICP constructor arguments (src/ProjectiveICP.py:9-18). */
typedef struct picp_schedule {
    int32_t num_iterations;
    int32_t min_inliers;
    double kernel_threshold;
    double min_kernel_threshold;
    double max_kernel_threshold;
    double damping; /* "dumping_factor" in the reference */
    double min_damping;
    double max_damping;
    int32_t disable_stop; /* 1: ignore the stop test (:228) and always run num_iterations (benchmarks) */
    int32_t reserved;
} picp_schedule;

/* Result of a whole loop (src/ProjectiveICP.py:249-257). */
typedef struct picp_result {
    double T[16];           /* returned pose: best iteration's pose, or the start pose if invalid */
    int32_t valid;          /* 0 when the best iteration's error is > 5 (:249) */
    int32_t num_iterations; /* iterations actually run */
    int32_t best_index;     /* first argmin of the error list (:249, :254) */
    int32_t reserved;
    double best_error;
} picp_result;

typedef struct picp_ctx picp_ctx;

int picp_version(void);
const char* picp_last_error(void);

/* stream: a hipStream_t (NULL = the default stream).  Allocates the ctx's reduction workspace. */
int picp_create(picp_ctx** ctx, int device, void* stream);
int picp_destroy(picp_ctx* ctx);
int picp_set_stream(picp_ctx* ctx, void* stream);
int picp_synchronize(picp_ctx* ctx);

/* ---- the LM loop, one call (src/ProjectiveICP.py:154-257 on pre-associated pairs) ----------
 * Runs up to sched->num_iterations iterations of: in-image filter of every pair under the
 * current pose (the pose-dependent part of the association call at :182, src/data_association.py:44-46),
 * linearise (:276-353), damped solve + pose update (:260-273), schedules (:200-228).
 * hist: device, [hist_rows][PICP_HIST_STRIDE], hist_rows >= num_iterations.
 * state: device, PICP_STATE_DOUBLES doubles of scratch owned by the caller.
 * Synchronises the stream before returning; result is host memory.  (After the synchronisation a
 * sub-microsecond housekeeping kernel may be left enqueued on the stream; later work is ordered
 * behind it.) */
int picp_run(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y, const double* z,
             int64_t M, const double* w_T_c0_host, const picp_camera* cam, const picp_schedule* sched,
             double* state, double* hist, int32_t hist_rows, picp_result* result);

/* picp_run has three device paths with the same results (sums differ only in the order of additions):
 *  - up to picp_resident_capacity(ctx) correspondences the whole loop is ONE launch that reads the
 *    five arrays once and keeps them in registers and LDS, exchanging the two per-iteration sums of
 *    :301 and :313-317 between workgroups inside the kernel;
 *  - larger inputs (16-byte aligned arrays) run the whole loop in one launch as well, with the same
 *    in-kernel exchange: the first 5 120 correspondences per compute unit stay in LDS and registers
 *    for the whole launch, the rest of the arrays is streamed from memory twice per iteration;
 *  - otherwise, and as the fallback of the other two, every iteration is three launches (two scans
 *    and the solve step).
 * picp_set_resident_limit lowers the switch point between the first two (0 = always the per-iteration
 * launches); the capacity is a property of the device (CUs x threads x registers). */
int picp_set_resident_limit(picp_ctx* ctx, int64_t max_correspondences);
int64_t picp_resident_capacity(picp_ctx* ctx);
/* How many of M correspondences the second path keeps on chip for the whole launch (0 when M leaves less
 * than one full streaming step beside them): what its scans do not read from memory again. */
int64_t picp_streaming_held(picp_ctx* ctx, int64_t M);
/* The single launch waits for all of its workgroups at every exchange, with a bound on each wait.  If
 * the device is shared and part of the grid is not scheduled, the launch gives up and picp_run
 * repeats the loop with the streaming kernels; this returns how often that has happened on ctx. */
int picp_resident_aborts(picp_ctx* ctx);
/* The on-chip single launch has a lean form of stage 2 that is valid while every value it handles
 * (points, measurements, camera matrix, the poses of the loop) is finite and below 2^199 in magnitude;
 * it checks that itself, and a launch that meets anything else leaves at once and is run again in the
 * form that assumes nothing, as are the next PICP_FULL_FORM_CALLS on-chip loops of ctx (the two forms
 * add the same numbers in the same order: identical results; no counterpart in the reference).  This
 * returns how often that has happened on ctx. */
#define PICP_FULL_FORM_CALLS 8
int picp_immoderate_exits(picp_ctx* ctx);
/* Timing aid for benchmarks: runs the single-launch loop `reps` times from w_T_c0_host, each launch
 * bracketed by HIP events bound to that dispatch, and reports the mean and minimum kernel duration
 * in milliseconds.  Beyond picp_resident_capacity it times the single launch that streams the arrays
 * (16-byte aligned arrays).  Synchronises. */
int picp_profile_run(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                     const double* z, int64_t M, const double* w_T_c0_host, const picp_camera* cam,
                     const picp_schedule* sched, double* state, int32_t reps, double* mean_ms, double* min_ms);

/* picp_run in two halves, for callers that keep several independent loops in flight (the host then
 * prepares and enqueues the next loop while the current one runs, instead of waiting in between).
 * picp_run_enqueue takes picp_run's arguments plus a slot number 0 .. PICP_ASYNC_SLOTS-1 and only
 * enqueues; it accepts what picp_run executes as a single launch (non-empty input within
 * picp_resident_capacity, or larger 16-byte aligned arrays; at least one iteration) and refuses
 * everything else with PICP_ERR_INVALID.  picp_run_collect waits for the loop of that slot and fills
 * `result` as picp_run does; PICP_ERR_RETRY means the launch computed nothing -- it gave up at an
 * exchange (shared device), or it met values outside the lean form's range (picp_immoderate_exits) --
 * and the caller should run that loop again with picp_run.  state and hist must stay untouched until
 * the slot has been collected, and a slot must be collected before it is used again.  Loops on one
 * ctx execute in the order they were enqueued. */
#define PICP_ASYNC_SLOTS 4
int picp_run_enqueue(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                     const double* z, int64_t M, const double* w_T_c0_host, const picp_camera* cam,
                     const picp_schedule* sched, double* state, double* hist, int32_t hist_rows, int32_t slot);
int picp_run_collect(picp_ctx* ctx, int32_t slot, const double* w_T_c0_host, picp_result* result);

/* Same loop for B independent problems in one launch, one workgroup per problem, the whole loop
 * inside the kernel.  Problem p owns correspondences [offsets[p], offsets[p+1]).
 * offsets: device int64[B+1]; poses0: device [B][16]; results: device picp_result[B];
 * hist: device [B][hist_rows][PICP_HIST_STRIDE] or NULL (then only results are produced).
 * max_count: an upper bound on the size of every problem, or 0 if the caller has none (offsets is
 * device memory, so the library cannot look).  With a bound of at most 4 096 the problems are held
 * in registers and LDS for the whole loop; otherwise they are re-read from memory every iteration.
 * A problem that exceeds a non-zero bound is not run: its result has num_iterations = -1.
 * Does not synchronise. */
int picp_run_batched(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                     const double* z, const int64_t* offsets, int32_t B, const double* poses0,
                     const picp_camera* cam, const picp_schedule* sched, picp_result* results, double* hist,
                     int32_t hist_rows, int64_t max_count);

/* ---- the loop's stages, for callers that interleave their own exchange (sharded runs) ------
 * picp_state_init   = loop entry (:157-171): pose, inverse pose, thresholds, counters.
 * picp_stage1       = first scan: per pair in-image test and chi; sums {sum chi, count} of the
 *                     pairs with chi <= kernel_threshold (:289-297) and the in-image count.
 * picp_stage2       = second scan: mean filter (:301-310) and the H, b sums (:313-317) of both
 *                     candidate inlier sets.
 * picp_solve        = damped solve, pose update, schedules, history row (:269-271, :200-238).
 * Each stage leaves per-workgroup partial sums in the ctx.  picp_stage1_reduce /
 * picp_stage2_reduce fold them into one record in caller memory (device, PICP_S1_DOUBLES /
 * PICP_S2_DOUBLES doubles) that the caller may all-reduce(sum) across shards and hand to the
 * next stage through s1 / s2.  Passing NULL for s1 / s2 makes the next stage fold the ctx's
 * partial sums itself (single-GPU path, no extra launch).  None of these synchronise; every
 * stage is a no-op once the state's stop flag is set. */
int picp_state_init(picp_ctx* ctx, double* state, const double* w_T_c0_host, const picp_schedule* sched);
int picp_stage1(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                const double* z, int64_t M, const double* state, const picp_camera* cam);
int picp_stage1_reduce(picp_ctx* ctx, double* s1);
int picp_stage2(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                const double* z, int64_t M, const double* state, const picp_camera* cam, const double* s1);
int picp_stage2_reduce(picp_ctx* ctx, double* s2);
/* Direction in which picp_stage2 walks the arrays: 0 = as picp_stage1 does, 1 = against it (stage 2
 * then starts with what stage 1 read last, and the next stage 1 with what stage 2 read last, so
 * arrays up to about twice the 256 MB memory-side cache are partly served from it), -1 = chosen by
 * size (the default).  The sums are the same up to the order of additions.  No counterpart in the
 * reference, whose loop visits the pairs in index order (src/ProjectiveICP.py:286). */
int picp_set_stage2_order(picp_ctx* ctx, int32_t order);
int picp_solve(picp_ctx* ctx, double* state, const picp_schedule* sched, const double* s1, const double* s2,
               double* hist, int32_t hist_rows);
/* Timing aid for benchmarks: launches the stage-2 scan `reps` times, each dispatch bracketed by HIP
 * events bound to that dispatch (hipExtLaunchKernelGGL), and reports the mean and minimum kernel
 * duration in milliseconds.  Needs a preceding picp_stage1 on the same ctx.  Synchronises. */
int picp_profile_stage2(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                        const double* z, int64_t M, const double* state, const picp_camera* cam, int32_t reps,
                        double* mean_ms, double* min_ms);
/* ---- sharded runs: the all-reduces on the ctx's stream ------------------------------------
 * One process per GPU, every rank holding a contiguous slice of the correspondence arrays and a
 * replica of the loop state.  The reference has no counterpart (it is one process); what is
 * exchanged are the sums of :301 and :313-317.  picp_comm_* wrap an RCCL communicator owned by the
 * ctx (RCCL is loaded at run time; without it these return PICP_ERR_COMM and everything else keeps
 * working): rank 0 obtains an id with picp_comm_unique_id, the caller hands its PICP_COMM_ID_BYTES
 * bytes to every rank by whatever means it has (torch.distributed's store, MPI, a file), and all
 * ranks call picp_comm_init together (it blocks until every rank has joined).
 * picp_comm_allreduce sums `count` doubles in place over the ranks, enqueued on the ctx's stream.
 * picp_sharded_rounds enqueues `rounds` rounds of
 *   stage 1 -> fold into s1 -> all-reduce(s1) -> stage 2 -> fold into s2 -> all-reduce(s2) -> solve
 * on this rank's slice (s1, s2: device, PICP_S1_DOUBLES / PICP_S2_DOUBLES doubles); with no
 * communicator the all-reduces are skipped.  Every rank must enqueue the same number of
 * rounds: the replicated state is identical on all ranks, so a stop flag read back on one is the
 * same on all (rounds after the stop are no-ops apart from the exchange: the scans no longer rewrite
 * s1 / s2, whose contents are undefined from then on -- every further all-reduce sums the stale
 * records once more).  Does not synchronise. */
#define PICP_COMM_ID_BYTES 128
int picp_comm_unique_id(void* id_host, int64_t bytes);
int picp_comm_init(picp_ctx* ctx, const void* id_host, int32_t rank, int32_t world);
int picp_comm_destroy(picp_ctx* ctx);
int picp_comm_world(picp_ctx* ctx); /* ranks of the ctx's communicator, 0 without one */
int picp_comm_allreduce(picp_ctx* ctx, double* buf, int64_t count);
int picp_sharded_rounds(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                        const double* z, int64_t M, double* state, const picp_camera* cam,
                        const picp_schedule* sched, double* s1, double* s2, double* hist, int32_t hist_rows,
                        int32_t rounds);

/* ---- sharded runs with the exchange INSIDE the kernel -----------------------------------------
 * The single-launch loop on every rank's slice, the two sums of :301 and :313-317 exchanged between the
 * GPUs by the kernels themselves: after the workgroups of a GPU have combined their sums, one workgroup
 * stores the GPU's record into a slot of every peer's exchange memory (device memory of the peer, mapped
 * through HIP IPC; the stores travel over xGMI), and every workgroup adds the `world` records in rank
 * order.  No launch and no library collective per iteration; all ranks compute bit-identical states.
 * The reference has no counterpart (it is one process).
 * Set-up, collective over the ranks (one process per GPU, at most PICP_PEER_MAX_RANKS):
 *   picp_peer_create   allocates this rank's exchange memory and writes its IPC handle
 *                      (PICP_PEER_HANDLE_BYTES bytes, host) for the caller to hand to every rank;
 *   picp_peer_connect  maps the exchange memory of all ranks; handles: host, [world][PICP_PEER_HANDLE_BYTES]
 *                      in rank order (this rank's own entry is ignored).  All ranks must have returned
 *                      from picp_peer_create before any rank calls it, and from picp_peer_connect before
 *                      any rank runs a loop;
 *   picp_peer_destroy  unmaps and frees (all ranks must have finished their loops).
 * picp_run_sharded runs the loop on this rank's slice (M may be 0) and synchronises; every rank must
 * call it with the same pose, camera and schedule.  Results as picp_run, identical on all ranks.
 * A slice that fits the register files of the GPU (picp_resident_capacity) is held on chip for the
 * whole loop, a larger or empty one is streamed from memory twice per iteration (all but a part of it
 * that the kernel keeps on chip, as picp_run does at such sizes); the two forms run
 * the same exchanges, so the choice is each rank's own (picp_set_resident_limit(ctx, 0): always stream).
 * Every wait inside the kernel is bounded (about a second; a launch's first exchange waits sixteen times as
 * long, because the ranks' hosts reach their launches at different times): if a peer's record does not
 * arrive (a rank died or was never launched) the launch gives up, the call returns PICP_ERR_RETRY and the exchange is unusable
 * until it has been destroyed and set up again; callers then fall back to picp_sharded_rounds.
 * The status is this rank's own: a rank can give up at the last exchange after publishing its record
 * while a slower rank still completes.  Ranks must therefore agree on the outcome (a logical AND over
 * their statuses, through whatever process group they have) BEFORE any of them acts on it, and fall
 * back together (the package's sharded.run_peer does this).
 * picp_peer_world: ranks of a usable exchange, 0 if there is none. */
#define PICP_PEER_MAX_RANKS 8
#define PICP_PEER_HANDLE_BYTES 64
int picp_peer_create(picp_ctx* ctx, int32_t rank, int32_t world, void* handle_host);
int picp_peer_connect(picp_ctx* ctx, const void* handles_host);
int picp_peer_destroy(picp_ctx* ctx);
int picp_peer_world(picp_ctx* ctx);
/* Checks to make BEFORE picp_peer_connect, so that a pair of GPUs without peer access ends in a message that
 * names the pair (and in the callers' fallback) rather than in a failed mapping or a fault:
 *   picp_device_bus_id    the PCI bus id of the ctx's GPU as a string (host, PICP_BUS_ID_BYTES bytes), for the
 *                         caller to hand to every rank next to the IPC handle (device ordinals are per process);
 *   picp_peer_can_access  *can = 1 when the ctx's GPU can map memory of the GPU with that bus id (or is that GPU),
 *                         0 when it cannot, -1 when this process does not see that GPU and so cannot tell.
 * picp_peer_memory_kind: how picp_peer_create got the exchange memory -- uncached device memory, or, where that
 * is refused, fine-grained device memory (PICP_PEER_MEMORY_*; 0 without an exchange).  No counterpart in the reference. */
#define PICP_BUS_ID_BYTES 64
#define PICP_PEER_MEMORY_NONE 0
#define PICP_PEER_MEMORY_UNCACHED 1
#define PICP_PEER_MEMORY_FINEGRAINED 2
int picp_device_bus_id(picp_ctx* ctx, char* bus_id_host, int64_t bytes);
int picp_peer_can_access(picp_ctx* ctx, const char* peer_bus_id_host, int32_t* can);
int picp_peer_memory_kind(picp_ctx* ctx);
/* 1 when librccl can be loaded by this process (picp_comm_init would not fail for that reason), else 0. */
int picp_comm_available(void);
int picp_run_sharded(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                     const double* z, int64_t M, const double* w_T_c0_host, const picp_camera* cam,
                     const picp_schedule* sched, double* state, double* hist, int32_t hist_rows,
                     picp_result* result);
/* Timing aid for benchmarks: with timing on, picp_run_sharded brackets its launch with HIP events bound
 * to that dispatch, and picp_last_launch_ms reports the duration of the last such launch. */
int picp_set_launch_timing(picp_ctx* ctx, int32_t on);
int picp_last_launch_ms(picp_ctx* ctx, double* ms);
/* Caps the number of workgroups of the single-launch loops (0: one per compute unit, the default).
 * For rehearsing a sharded run with several ranks on ONE GPU: the kernels of all ranks must be
 * resident together, so each may use only its share of the compute units. */
int picp_set_grid_limit(picp_ctx* ctx, int32_t max_workgroups);

/* Reads the history back (synchronises) and applies the best-iteration rule (:249-257). */
int picp_finish(picp_ctx* ctx, const double* state, const double* hist, int32_t hist_rows,
                const double* w_T_c0_host, picp_result* result);

/* ---- single calls mirroring the reference's private methods, for tests ---------------------
 * picp_linearize = __linearize (src/ProjectiveICP.py:276-323).  Synchronises.
 * Host outputs: H[36], b[6], num_inliers, error, n_inside. */
int picp_linearize(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                   const double* z, int64_t M, const double* w_T_c0_host, const picp_camera* cam,
                   double kernel_threshold, int32_t min_inliers, double* H, double* b, int64_t* num_inliers,
                   double* error, int64_t* n_inside);

/* picp_linearize for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference; it
 * computes what picp_linearize computes, with the same kernels, but every intermediate (the state
 * block, the per-workgroup records of both scans, the folded records) lives in a device array that is
 * part of the library's code object (322 KiB on every GPU the library runs on, held until the process
 * ends), and the results go to caller-owned DEVICE memory instead of host variables:
 *   out[0..35] H row-major, out[36..41] b, out[42] num_inliers, out[43] error, out[44] n_inside,
 *   out[45..47] zero                                         (PICP_LINEARIZE_OUT_DOUBLES doubles).
 * The call only enqueues kernels on the ctx's stream: no hipMalloc / hipFree, no host-device copy and
 * no synchronisation (w_T_c0_host and cam are read before it returns).  Inputs of more than
 * PICP_NOALLOC_MAX_CORRESPONDENCES correspondences are refused with PICP_ERR_CAPACITY.
 * All calls on one GPU share that GPU's array, whatever their ctx or stream: calls that overlap in
 * time on one GPU race; order them on one stream or by events.  Nothing in the array needs to be
 * cleared between calls (every call rewrites what it reads). */
#define PICP_LINEARIZE_OUT_DOUBLES 48
#define PICP_NOALLOC_MAX_CORRESPONDENCES (((int64_t)1) << 32)
int picp_linearize_noalloc(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                           const double* z, int64_t M, const double* w_T_c0_host, const picp_camera* cam,
                           double kernel_threshold, int32_t min_inliers, double* out);

/* ---- appearance association (src/data_association.py:3-57) ---------------------------------
 * All pairs (i, j) whose D-component descriptors are equal component by component under float
 * equality (-0.0 == 0.0, NaN never equal: Python list equality at :31), in i-major, j-minor
 * order (:29-30).  app1: device [N1][D], app2: device [N2][D]; idx1/idx2: device int64[capacity].
 * *count (host) receives the number of pairs; if it exceeds capacity nothing past capacity is
 * written and PICP_ERR_CAPACITY is returned.  Synchronises. */
int picp_associate(picp_ctx* ctx, const double* app1, int64_t N1, const double* app2, int64_t N2, int32_t D,
                   int64_t* idx1, int64_t* idx2, int64_t capacity, int64_t* count);
/* picp_associate has two methods with identical output.  mode 1: the all-pairs scan, the reference's
 * double loop (:29-30) with a 32-bit screening test per pair.  mode 2: a hash join -- set 2 goes into
 * an open-addressing table keyed by a hash of the descriptor values, each query row runs the full
 * comparison only against its bucket, matches are emitted in increasing set-2 row.  mode 0 (default):
 * the join for large inputs (N1 * N2 >= 4e9), the scan otherwise.  The join gives up on inputs with
 * long runs of equal rows (more than 32 equal set-2 rows for one query row), and does not take
 * descriptors of more than 23 components (a workgroup's rows are staged in LDS); the scan is run
 * instead; picp_association_fallbacks counts how often. */
int picp_set_association_mode(picp_ctx* ctx, int32_t mode);
int picp_association_fallbacks(picp_ctx* ctx);

/* Camera.project_point for N points (src/Camera.py:82-121): inside[N] (uint8, device) and the
 * projected pixel uv_u[N], uv_v[N] (device; written for every point, meaningful where inside).
 * c_T_w_host: the camera state matrix (src/Camera.py:76).  Does not synchronise. */
int picp_project(picp_ctx* ctx, const double* x, const double* y, const double* z, int64_t N,
                 const double* c_T_w_host, const picp_camera* cam, uint8_t* inside, double* uv_u, double* uv_v);

/* Builds the SoA correspondence arrays from pair indices: u,v from pos1[N1][2] rows idx1[k],
 * x,y,z from pos2[N2][3] rows idx2[k] (the np.array(matches[...]) step at src/ProjectiveICP.py:183-184).
 * Does not synchronise. */
int picp_gather_pairs(picp_ctx* ctx, const int64_t* idx1, const int64_t* idx2, int64_t n, const double* pos1,
                      const double* pos2, double* u, double* v, double* x, double* y, double* z);

/* 4x4 inverse as the loop computes it for camera.set_c_T_w (:232), host in / host out, runs on
 * the GPU so tests can pin the device routine.  Synchronises. */
int picp_invert_pose(picp_ctx* ctx, const double* T_host, double* Tinv_host);
/* v2T (src/geometry_utils.py:38-45) evaluated by the device routine.  Synchronises. */
int picp_v2T(picp_ctx* ctx, const double* v6_host, double* T_host);

#ifdef __cplusplus
}
#endif
#endif /* PICP_HIP_H */
