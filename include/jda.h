/*
 * jda.h -- C ABI of the MI355X-native JDA sliding-window detector (libjda.so).
 *
 * Section 1 is the drop-in boundary: the six entry points of the reference
 * header (reference c/jda.h:31-68) with identical names, argument order,
 * argument meaning and ownership rules, so a program written against the
 * reference C library links against this one unchanged.
 *
 * Section 2 is additive: batch / device-resident entry points, a second
 * numeric dialect (the fp64 `src/jda` path), per-window trace output used by
 * the parity tests, and an error channel.  Nothing in section 2 changes the
 * behaviour of section 1.
 *
 * Plain C types only: pointers, ints, floats.  No C++/HIP/torch types cross
 * this boundary (a HIP stream is passed as an opaque void*).
 */
#ifndef JDA_AMD_JDA_H_
#define JDA_AMD_JDA_H_

#include <stddef.h>
#include <stdint.h>

#if defined(_MSC_VER)
#  if defined(JDA_EXPORTS)
#    define JDA_API __declspec(dllexport)
#  else
#    define JDA_API __declspec(dllimport)
#  endif
#else
#  define JDA_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* 1. Reference-compatible surface                                          */
/* ------------------------------------------------------------------------ */

/* Detection list, returned BY VALUE (reference c/jda.h:18-24).
 * bboxes is n triples (x, y, size); shapes is n rows of 2*landmark_n floats
 * in absolute pixel coordinates (x1, y1, x2, y2, ...); scores is n floats.
 * The three arrays are malloc()ed by the library and owned by the caller
 * until jdaResultRelease(). */
typedef struct {
  int n;
  int landmark_n;
  int *bboxes;
  float *shapes;
  float *scores;
} jdaResult;

/* replaces reference c/jda.c:486-561 (c/jda.h:31).  Loads a model whose real
 * fields are 8-byte doubles (the trainer's format, src/jda/cascador.cpp:79).
 * Returns NULL if the file cannot be opened.  Unlike the reference, the
 * cascade dimensions are taken from the file header at run time, and a file
 * whose size does not match its header is refused (NULL). */
JDA_API void *jdaCascadorCreateDouble(const char *model);

/* replaces reference c/jda.c:563-638 (c/jda.h:32).  Same for the 4-byte float
 * layout that jdaCascadorSerializeTo() writes. */
JDA_API void *jdaCascadorCreateFloat(const char *model);

/* replaces reference c/jda.c:644-716 (c/jda.h:41).  Writes the float layout,
 * including the reference's header convention (stage index T+1, cart -1).
 * Silently returns if the file cannot be created, like the reference. */
JDA_API void jdaCascadorSerializeTo(void *cascador, const char *model);

/* replaces reference c/jda.c:718-720 (c/jda.h:47). NULL is accepted.
 * Batches that were submitted (jdaDetectBatchSubmit*) and never waited for are drained.  Like the reference's, this
 * must not race with a call on the same cascador from another thread: the reference frees what jdaDetect reads; here
 * such a call is given up to ten seconds to return before the cascador goes (tests/test_reentrant.py). */
JDA_API void jdaCascadorRelease(void *cascador);

/* replaces reference c/jda.c:443-480 (c/jda.h:62-63).
 *   data      borrowed, width*height contiguous 8-bit gray, row stride = width
 *   scale     growth factor of the window size between pyramid levels
 *   step      accepted and ignored, exactly like the reference, which
 *             overrides it with 10 % of the window size (c/jda.c:333)
 *   min_size  raised to 24 if smaller (c/jda.c:459)
 *   max_size  <= 0 means min(width, height) (c/jda.c:460)
 *   th        final score cut (c/jda.c:414)
 * Output order: scan order (level, y, x) of the windows that survive NMS.
 * Re-entrant like the reference (no globals, no locks in c/jda.c:443-480): any number of threads may call it -- and
 * every other detect entry below -- on ONE cascador at the same time; each call takes a lane (stream + workspace)
 * from the cascador's pool, the model and the scan plans are shared read-only.
 * Runs the cascade on the GPU (HIP device selected with jdaSetDevice, default
 * the current device).  There is no CPU fallback: if no HIP device is usable
 * the call returns an empty result, sets jdaGetLastError() and prints the
 * reason on stderr. */
JDA_API jdaResult jdaDetect(void *cascador, unsigned char *data, int width, int height,
                            float scale, float step, int min_size, int max_size, float th);

/* replaces reference c/jda.c:722-727 (c/jda.h:68). */
JDA_API void jdaResultRelease(jdaResult result);

/* ------------------------------------------------------------------------ */
/* 2. Additive extensions                                                   */
/* ------------------------------------------------------------------------ */

/* Numeric dialects of the same cascade.
 * JDA_DIALECT_C   : fp32, truncating coordinates, growing window with 10 %
 *                   step -- reference c/jda.c (what jdaDetect runs).
 * JDA_DIALECT_CPP : fp64, round() coordinates, fixed pixel step, no final
 *                   threshold, score-ordered NMS -- reference
 *                   src/jda/cascador.cpp:166-211,310-477 (method 1). */
enum { JDA_DIALECT_C = 0, JDA_DIALECT_CPP = 1 };

/* Thread-local message of the last failed call on this thread ("" if none).
 * No C++ exception leaves the library: an allocation failure inside any entry (std::bad_alloc from the host side's
 * containers, the tables of a large model) is caught at the boundary and turned into the entry's error value -- NULL,
 * -1 or an empty jdaResult, the reference's own answer to a failed malloc (c/jda.c:487-493) -- with the reason here.
 * A successful call leaves it empty.  (When the persistent stage-0 scan kernel gives up waiting inside a launch -- a
 * watchdog; never observed on hardware -- or covers fewer windows than the plan holds, the pass is run again with the
 * closed-tile scan kernel: the call succeeds with correct results, says so on stderr and counts it in
 * jdaStats::scan_fallbacks; it is not an error and is not reported here.) */
JDA_API const char *jdaGetLastError(void);

/* Opens a model of either layout; the layout is inferred from the file size
 * implied by the header (the file has no discriminator). */
JDA_API void *jdaCascadorCreate(const char *model);

typedef struct {
  int T;            /* stages                                   */
  int K;            /* carts per stage                          */
  int landmark_n;   /* landmarks L; a shape has 2L coordinates  */
  int tree_depth;   /* D; a cart has 2^(D-1)-1 split nodes      */
  int multi_scale;  /* 1 if any split node reads the half/quarter image */
  int source_real_bytes; /* 8 or 4: layout of the file it came from */
} jdaModelInfo;

JDA_API int jdaCascadorInfo(void *cascador, jdaModelInfo *info);

/* Pin a cascador to a HIP device ordinal (default: device current at first
 * use).  Must be called before the first detect on this cascador. */
JDA_API int jdaSetDevice(void *cascador, int device);

/* Tuning options of a cascador.  They start from the JDA_* environment variables (read once, when the cascador
 * is created) and can be changed here while NO call is running and no submitted batch is pending
 * on this cascador (refused otherwise); a change drops the cached scan plans.  None of them changes results.  Documented keys:
 *   "handoff"       carts of stage 0 the scan kernel evaluates before the finishing kernel takes over (128)
 *   "lanes"         sub-batches of one synchronous call that run side by side on their own streams (2)
 *   "dense"         whole-stage tile kernel for models that reject little: 0 off, 1 auto, 2 always (1)
 *   "workspace_mb"  device workspace budget of a call in MiB; larger batches run in several passes (24576)
 *   "ws_bound"      size the survivor queues of a pass from the fractions earlier passes left in them ("ws_factor_pct" = 400 % of
 *                   them, at least "ws_min_entries" = 65536) instead of for every window; a pass that outgrows them is rerun (1)
 *   "plan_cache"    scan plans (one per frame size and call parameters) kept per cascador (64)
 *   "predict"       size the finishing launches from the previous pass instead of a host round trip (1)
 *   "scan_p"        the persistent form of the stage-0 scan for large uniform batches: 0 off, 1 where it suits, 2 wherever it fits (1)
 *   "device_post"   per-frame sort, NMS and relocation of batches of 16 frames or more on the device instead of on the host (1)
 *   "hwq_place"     the cascador's streams are placed on the device's hardware queues: the HIP runtime deals a process's streams
 *                   to four queues in creation order, so whether two lanes share one (and run one after the other) depends on
 *                   the streams the host program created before; the library probes which of its streams share a queue and
 *                   hands them out by queue.  0: wherever the runtime puts them (1).  Read-only, what the pool found:
 *                   "hwq_queues", "hwq_streams", "hwq_probes", "hwq_max_mains" (most lanes whose main streams share a queue)
 * (the other keys of DESIGN.md section 8 are accepted as well; they are experiment switches).
 * Returns 0, or -1 for an unknown key / a running call or pending batch.  jdaGetOption returns the value (-1: unknown key). */
JDA_API int jdaSetOption(void *cascador, const char *key, long long value);
JDA_API long long jdaGetOption(void *cascador, const char *key);

/* Window enumeration of reference c/jda.c:320-339 without running anything:
 * number of candidate windows and pyramid levels for one frame. */
JDA_API int jdaCountWindows(int width, int height, float scale, int min_size, int max_size,
                            long long *n_windows, int *n_levels);

/* Work counters of one detect call: the DetectionStatisic of reference
 * include/jda/cascador.hpp:14-25, plus what the roofline accounting needs. */
typedef struct {
  long long patch_n;          /* candidate windows scanned                    */
  long long face_patch_n;     /* windows that passed every cart (+ final th)  */
  long long nonface_patch_n;  /* patch_n - face_patch_n                       */
  long long cart_gothrough_n; /* reject lengths (Validate's n, cascador.cpp:187) */
                              /* summed over the NON-face windows only, like  */
                              /* the reference (cascador.cpp:359-364)         */
  long long stage_done_n[16]; /* windows that completed stage t (shape update) */
  double average_cart_n;      /* cart_gothrough_n / nonface_patch_n           */
  double gpu_ms;              /* device time of the call (HIP events; the events -- five marker packets per */
                              /* pass -- are only recorded when statistics are asked for)                   */
  double scan_ms;             /* device time of the stage-0 scan launches (first start to last end; */
                              /* two sub-batches scan side by side on big batches)                  */
  double host_ms;             /* host post-processing (sort, NMS, relocation) */
  long long scan_cart_n;      /* part of cart_gothrough_n done by the stage-0 scan kernel */
  long long scan_patch_n;     /* windows the stage-0 scan kernel covered      */
  int scan_launches;          /* launches of the stage-0 scan kernel (one per tiled level) */
  long long handoff_n;        /* windows handed from the scan to the finishing kernel (incl. untiled levels) */
  long long cart_total_n;     /* carts evaluated over ALL windows (roofline accounting) */
  double call_ms;             /* wall clock of the whole C call                 */
  int dense_passes;           /* passes that ran in dense mode (whole stages per window tile, k_stage) */
  double scan_lds_ms;         /* HIP-event span of the LDS-tiled k_scan launches alone; only meaningful when the launches of
                                 a call run back to back on one stream (JDA_LANES=1 JDA_SIDE_STREAM=0), 0 otherwise */
  long long scan_lds_cart_n;  /* carts evaluated by the LDS-tiled k_scan launches (scan_cart_n minus the global-pixel levels) */
  int scan_fallbacks;         /* passes of this call that were run a second time with the closed-tile scan kernel because the
                                 persistent one tripped a watchdog or covered too few windows (results are correct; see stderr) */
  int ws_regrows;             /* passes of this call that were run a second time because a queue sized from earlier passes'
                                 survivor fractions was too small (option "ws_bound"; results are correct; see stderr) */
} jdaStats;

typedef struct {
  int dialect;          /* JDA_DIALECT_*                                      */
  int nms;              /* 1: apply NMS (default), 0: return every survivor   */
  float nms_overlap;    /* IoU threshold, 0.3 in reference c/jda.c:238        */
  int cpp_step;         /* dialect CPP: pixel step (config fddb.step)         */
  void *hip_stream;     /* hipStream_t to enqueue on, NULL = library's own    */
  jdaStats *stats;      /* optional out                                       */
} jdaDetectOptions;

JDA_API void jdaDetectOptionsInit(jdaDetectOptions *opt);

/* Batch of n equally sized frames in HOST memory (frames[i] is width*height
 * bytes).  out must point at n jdaResult slots; each is released separately
 * with jdaResultRelease.  Per frame the result is identical to n separate
 * jdaDetect calls.  Returns 0 on success. */
JDA_API int jdaDetectBatch(void *cascador, const unsigned char *const *frames, int n,
                           int width, int height, float scale, float step,
                           int min_size, int max_size, float th,
                           const jdaDetectOptions *opt, jdaResult *out);

/* Same, frames already resident in device memory: frame i starts at
 * d_frames + i*frame_stride (frame_stride >= width*height).  This is the
 * entry the throughput benchmark times. */
JDA_API int jdaDetectBatchDevice(void *cascador, const unsigned char *d_frames,
                                 size_t frame_stride, int n, int width, int height,
                                 float scale, float step, int min_size, int max_size,
                                 float th, const jdaDetectOptions *opt, jdaResult *out);

/* Ragged batch: n images of DIFFERENT sizes in one job -- the reference's FDDB loop, one Detect per image
 * (src/test.cpp:100-170), or any loop of jdaDetect calls over a list of images.  images[i] is widths[i]*heights[i]
 * bytes in HOST memory, rows back to back (the `data` of jdaDetect); out must point at n jdaResult slots.  Per
 * image the result is identical to jdaDetect(cascador, images[i], widths[i], heights[i], scale, step, min_size,
 * max_size, th).  The images are staged on the device with one common row pitch, the pyramid levels (the same
 * window-size series for every image, c/jda.c:331-333) share tile shapes and stage-0 tables, and the job runs as a
 * few large passes (chunks of `ragged_chunk_windows` candidate windows, three in flight) instead of n latency-bound
 * ones.  Models with multi-scale split nodes and cascades that reject almost nothing run image by image inside.
 * Images too small for a window give an empty result, like jdaDetect.  Returns 0 on success. */
JDA_API int jdaDetectBatchRagged(void *cascador, const unsigned char *const *images, const int *widths,
                                 const int *heights, int n, float scale, float step, int min_size, int max_size,
                                 float th, const jdaDetectOptions *opt, jdaResult *out);

/* Same, images already resident in device memory: image i starts at d_base + offsets[i], rows back to back. */
JDA_API int jdaDetectBatchRaggedDevice(void *cascador, const unsigned char *d_base, const size_t *offsets,
                                       const int *widths, const int *heights, int n, float scale, float step,
                                       int min_size, int max_size, float th, const jdaDetectOptions *opt,
                                       jdaResult *out);

/* The same two jobs with the results as ONE matrix instead of n jdaResults: a row per detection,
 *   [frame_offset + image index, x, y, size, score, shape (2L absolute coordinates)]   (5 + 2L floats),
 * images in order, detections of an image in jdaDetect's order -- exactly what jdaResultsPack makes of the n results,
 * and the form the multi-GPU gather ships (include/jda_dist.h).  A job of hundreds of images returns a few hundred
 * KB of rows; three allocations per image cost more host time than that (r06: 50 us of a 1.2-ms job).  *rows is
 * malloc'd by the library (also when *n_rows is 0) and released with jdaRowsRelease.  Returns 0 on success. */
JDA_API int jdaDetectBatchRaggedRows(void *cascador, const unsigned char *const *images, const int *widths,
                                     const int *heights, int n, float scale, float step, int min_size, int max_size,
                                     float th, const jdaDetectOptions *opt, int frame_offset, float **rows, int *n_rows);
JDA_API int jdaDetectBatchRaggedDeviceRows(void *cascador, const unsigned char *d_base, const size_t *offsets,
                                           const int *widths, const int *heights, int n, float scale, float step,
                                           int min_size, int max_size, float th, const jdaDetectOptions *opt,
                                           int frame_offset, float **rows, int *n_rows);
JDA_API void jdaRowsRelease(float *rows);

/* Up to three batches in flight on one cascador, driven by ONE host thread (streams of batches, e.g. video):
 * Submit queues a batch of device-resident frames on a lane of its own (stream + workspace) and returns at once with
 * a ticket (0..2; -1 on error: every ticket in use, multi-scale model); Wait collects that batch, post-processes it
 * and fills out[0..n) exactly like jdaDetectBatchDevice.  When earlier passes on this frame size have shown how many
 * windows survive the scan, the WHOLE batch (scan, finishing launches, copies of the results) is queued by Submit and
 * Wait is a single host wait.  Submitting batch i+1 before waiting for batch i keeps the GPU busy with batch i+1
 * while the host parts of batch i (D2H, sort, NMS, result assembly) run (one batch ahead is enough for frames that
 * are already on the device; frames coming from the host want two ahead, see jdaDetectBatchSubmitHost).
 * The frames must stay valid until Wait returns.  The other entry points keep working while tickets are pending
 * (they run on other lanes).  Wait takes the stats pointer (gpu_ms = that batch's own device span, call_ms = submit
 * to the end of wait); nothing is written through opt->stats by Submit, but a non-NULL opt->stats tells it to bracket
 * the pass with timing events -- without it Wait reports the counters and gpu_ms = scan_ms = 0. */
JDA_API int jdaDetectBatchSubmit(void *cascador, const unsigned char *d_frames, size_t frame_stride, int n,
                                 int width, int height, float scale, float step, int min_size, int max_size,
                                 float th, const jdaDetectOptions *opt);
JDA_API int jdaDetectBatchWait(void *cascador, int ticket, jdaStats *stats, jdaResult *out);

/* Submit for frames in HOST memory (frames[i] is width*height bytes): the batch is copied to a staging buffer
 * of its ticket (uploads of at least `h2d_min_bytes` go batch after batch through one upload stream of the cascador,
 * smaller ones on the ticket's own stream), then scanned like jdaDetectBatchSubmit.  The copy and the scan launches
 * are issued by a helper thread AFTER this call has returned, so in EVERY case -- pageable or pinned frames --
 * the frame BYTES must stay valid and unchanged until jdaDetectBatchWait for this ticket has returned (do not reuse
 * a capture buffer before that).  The frames[] pointer array itself is copied by this call and may go at once.
 * A ticket lives for copy + kernels + host work while the copy alone takes about half of that: keep TWO
 * batches submitted ahead of the one being waited for and the PCIe link never idles. */
JDA_API int jdaDetectBatchSubmitHost(void *cascador, const unsigned char *const *frames, int n,
                                     int width, int height, float scale, float step, int min_size, int max_size,
                                     float th, const jdaDetectOptions *opt);

/* Per-window trace of the cascade (parity instrumentation; HOST output
 * arrays of n*windows_per_frame entries in scan order, any may be NULL):
 *   carts_n    number of carts evaluated, counted like `n` in the reference's
 *              Validate (index of the rejecting cart + 1, or T*K)
 *   score      score when the walk stopped (before the final threshold)
 *   path_hash  FNV-1a over the leaf index of every evaluated cart
 *   shapes     2L floats, window-normalised shape after the last completed
 *              stage (mean shape if none completed)
 * Frames are host memory. Dialect C only. */
JDA_API int jdaTraceBatch(void *cascador, const unsigned char *const *frames, int n,
                          int width, int height, float scale, int min_size, int max_size,
                          int *carts_n, float *score, unsigned int *path_hash, float *shapes);

/* The image pair of reference c/jda.c:450-457 (half = 1/sqrt(2), quarter =
 * 1/2) built by the device resize kernel; exposed for the parity tests.
 * half/quarter are HOST buffers of hw*hh and qw*qh bytes. */
JDA_API int jdaBuildPyramid(void *cascador, const unsigned char *data, int width, int height,
                            unsigned char *half, int *hw, int *hh,
                            unsigned char *quarter, int *qw, int *qh);

/* Dialect CPP results carry doubles (reference Detect fills
 * vector<Rect>, vector<double>, vector<Mat_<double>>). */
typedef struct {
  int n;
  int landmark_n;
  int *rects;      /* n * (x, y, w, h)            */
  double *shapes;  /* n * 2L absolute coordinates */
  double *scores;  /* n                           */
} jdaResultD;

JDA_API void jdaResultDRelease(jdaResultD result);

/* Dialect CPP batch detect on host frames: reference JoinCascador::Detect with
 * fddb.method = 1 (src/jda/cascador.cpp:431-477): minimum_size, pixel step,
 * scale factor, NMS overlap, nms on/off come from the arguments instead of
 * the Config singleton.
 * A trainer snapshot (a double file whose header says the model is still in training: stage s < T, cart c) runs the way the
 * reference's Validate runs it -- stages [0, s) in full, then carts [0, c] of stage s without that stage's regression
 * (src/jda/cascador.cpp:177-209) -- in every dialect-CPP entry; jdaStats::cart_total_n counts the padding carts a face walks,
 * nothing else sees them.  With jdaSetSimilarityTransform(1) the stage in training walks with the parameter the stage before
 * it computed (Validate does not recompute stp_mc for it, cascador.cpp:178-200).  Refused (-1, jdaGetLastError): a training
 * status the reference's loader asserts against (cascador.cpp:138-141).  jdaDetect and the other dialect-C entries ignore
 * the header, like c/jda.c:499-505. */
JDA_API int jdaDetectBatchCpp(void *cascador, const unsigned char *const *frames, int n,
                              int width, int height, int minimum_size, int step,
                              double factor, double overlap, int nms,
                              jdaStats *stats, jdaResultD *out);

/* The same with the frames already resident in device memory (frame i at d_frames + i*frame_stride): the entry the
 * dialect-CPP throughput figures of bench.py are timed on. */
JDA_API int jdaDetectBatchCppDevice(void *cascador, const unsigned char *d_frames, size_t frame_stride, int n,
                                    int width, int height, int minimum_size, int step,
                                    double factor, double overlap, int nms,
                                    jdaStats *stats, jdaResultD *out);

/* Dialect CPP ragged batch: n images of DIFFERENT sizes as one job -- literally the loop of the reference's `jda fddb`
 * command, one joincascador.Detect(gray, ...) per image with fddb.method = 1 (src/test.cpp:100-170, line 142;
 * src/jda/cascador.cpp:310-376,431-477).  images[i] is widths[i]*heights[i] bytes in HOST memory, rows back to back;
 * out must point at n jdaResultD slots.  Per image the result is identical to jdaDetectBatchCpp on that image alone.
 * The window sizes minimum_size, int(win*factor), ... (cascador.cpp:314,369) are one series for every image, an image
 * uses the prefix that fits both its sides (cascador.cpp:333): levels, tile shapes and stage-0 tables are shared and
 * the job runs as a few large passes, like jdaDetectBatchRagged.  Models with multi-scale split nodes run image by
 * image inside.  PARITY UNPINNED like every dialect-CPP entry.  Returns 0 on success. */
JDA_API int jdaDetectBatchCppRagged(void *cascador, const unsigned char *const *images, const int *widths,
                                    const int *heights, int n, int minimum_size, int step,
                                    double factor, double overlap, int nms,
                                    jdaStats *stats, jdaResultD *out);

/* Same, images already resident in device memory: image i starts at d_base + offsets[i], rows back to back. */
JDA_API int jdaDetectBatchCppRaggedDevice(void *cascador, const unsigned char *d_base, const size_t *offsets,
                                          const int *widths, const int *heights, int n, int minimum_size, int step,
                                          double factor, double overlap, int nms,
                                          jdaStats *stats, jdaResultD *out);

/* The two dialect-CPP ragged jobs with the results as ONE matrix of rows of (6 + 2*landmark_n) doubles,
 * [frame_offset + image index, x, y, w, h, score, shape...], images in order, an image's rows in Detect's order (after the
 * score-ordered NMS, cascador.cpp:444-474) -- exactly what jdaResultsDPack makes of the n jdaResultDs, without building
 * them: the FDDB-sized job keeps 32 k candidates = 15 MB of rows, and going through 2,845 results and a pack cost 4 ms
 * of a 38-ms job.  *rows is malloc'd by the library (also when *n_rows is 0); release it with jdaRowsDRelease. */
JDA_API int jdaDetectBatchCppRaggedRows(void *cascador, const unsigned char *const *images, const int *widths,
                                        const int *heights, int n, int minimum_size, int step, double factor,
                                        double overlap, int nms, jdaStats *stats, int frame_offset,
                                        double **rows, int *n_rows);
JDA_API int jdaDetectBatchCppRaggedDeviceRows(void *cascador, const unsigned char *d_base, const size_t *offsets,
                                              const int *widths, const int *heights, int n, int minimum_size, int step,
                                              double factor, double overlap, int nms, jdaStats *stats, int frame_offset,
                                              double **rows, int *n_rows);
JDA_API void jdaRowsDRelease(double *rows);

/* Flattens n dialect-CPP results into rows of (6 + 2*landmark_n) doubles: [frame_offset + i, x, y, w, h, score, shape...]
 * (the rows reference src/test.cpp:153-163 prints per image, plus the landmarks) -- what is gathered across GPUs for the
 * dialect-CPP FDDB job.  rows may be NULL to query the row count.  Returns the number of rows, or -1 if capacity_rows is
 * too small. */
JDA_API int jdaResultsDPack(const jdaResultD *results, int n, int frame_offset, double *rows, int capacity_rows);

/* Releases n dialect-CPP results at once (same as n jdaResultDRelease calls). */
JDA_API void jdaResultsDRelease(jdaResultD *results, int n);

/* Dialect CPP only: the similarity-transform mode of Validate (reference
 * src/jda/data.cpp:64-126, config key face.similarity_transform, common.cpp:214; off in
 * the shipped config).  Per stage, sR = Calc(shape, mean_shape) rotates/scales the node
 * offsets and the regressed delta shape.  PARITY UNPINNED twice over: dialect CPP itself,
 * and two OpenCV details it leans on (cv::norm's accumulation order, `Mat_ /= double` as a
 * multiply by the reciprocal), both restated identically in the oracle and the kernel. */
JDA_API int jdaSetSimilarityTransform(void *cascador, int on);

/* Dialect CPP, detect method 0 -- the true image pyramid of reference
 * src/jda/cascador.cpp:216-308 (detectMultiScale + detectSingleScale): a fixed
 * origin_size x origin_size window (config image_size.origin_size, 48 in the shipped
 * config) slides with a pixel step over an image that is shrunk by 1/factor per level
 * ON THE DEVICE with a restatement of cv::resize(INTER_LINEAR); rects are scaled back
 * with truncating int *= double.  scale==0 models only (multi-scale models: jdaDetectBatchCppPyramidMS).  PARITY UNPINNED: cv::resize
 * itself cannot be compared here (no OpenCV), only its restatement in the oracle. */
JDA_API int jdaDetectBatchCppPyramid(void *cascador, const unsigned char *const *frames, int n,
                                     int width, int height, int origin_size, int step,
                                     double factor, double overlap, int nms,
                                     jdaStats *stats, jdaResultD *out);

/* The same for models with multi-scale split nodes: detectSingleScale resizes EVERY window's ROI to the config's
 * three sizes (image_size.origin_size / half_size / quarter_size -- 48 / 36 / 24 in the shipped config,
 * src/jda/cascador.cpp:243-245, common.cpp:129-131); a split node of scale 1 / 2 reads the window's own half / quarter
 * patch with coordinates scaled by that patch's side (data.cpp:21-51).  The patches are built on the device by the
 * cv::resize restatement, one per window and scale.  Serves scale==0 models too (the sizes are then unused).
 * PARITY UNPINNED like the entry above. */
JDA_API int jdaDetectBatchCppPyramidMS(void *cascador, const unsigned char *const *frames, int n,
                                       int width, int height, int origin_size, int half_size, int quarter_size,
                                       int step, double factor, double overlap, int nms,
                                       jdaStats *stats, jdaResultD *out);

/* The cv::resize(INTER_LINEAR, 8-bit gray) restatement by itself (device kernel), for tests. */
JDA_API int jdaResizeCv(void *cascador, const unsigned char *data, int width, int height,
                        unsigned char *out, int out_width, int out_height);

/* Host-only helpers (no GPU needed): the two NMS variants and the model-stream
 * size, exported so that they can be unit-tested and reused.
 * jdaNmsC   : reference c/jda.c:237-316; bboxes are (x,y,size) triples; keep[i]
 *             is set to 1/0; returns the number kept (scan order is preserved).
 * jdaNmsCpp : reference src/jda/cascador.cpp:387-429; rects are (x,y,w,h);
 *             picked[] receives indices in descending-score order; returns count. */
JDA_API int jdaNmsC(const int *bboxes, const float *scores, int n, float overlap, unsigned char *keep);
JDA_API int jdaNmsCpp(const int *rects, const double *scores, int n, double overlap, int *picked);
JDA_API long long jdaModelStreamBytes(int T, int K, int landmark_n, int tree_depth, int real_bytes);

/* The tile plan k_scan would use for a dialect-C call (no GPU needed; tests and tools): per pyramid level
 * 10 ints {win, step, nx, ny, mode, tw, th, pitch, tiles_x, tiles_y}; mode 1/3 = windows share an LDS pixel
 * tile of tw x th windows, 2 = pixels through L1/L2, 0 = not scanned.  Returns the number of levels. */
JDA_API int jdaDebugPlanTiles(void *cascador, int width, int height, float scale, int min_size, int max_size,
                              int *out, int cap_levels);

/* Flattens n per-frame results into rows of (5 + 2*landmark_n) floats:
 * [frame_offset + i, x, y, size, score, shape...] -- the (bbox, score, landmarks)
 * tuple that is gathered across GPUs.  rows may be NULL to query the row count.
 * Returns the number of rows, or -1 if capacity_rows is too small. */
JDA_API int jdaResultsPack(const jdaResult *results, int n, int frame_offset,
                           float *rows, int capacity_rows);

/* Releases n results at once (same as n jdaResultRelease calls). */
JDA_API void jdaResultsRelease(jdaResult *results, int n);

/* Per-window trace of the dialect-CPP cascade, like jdaTraceBatch but with the
 * fp64 state of reference Validate (src/jda/cascador.cpp:166-211): carts_n is
 * Validate's `n`. */
JDA_API int jdaTraceBatchCpp(void *cascador, const unsigned char *const *frames, int n,
                             int width, int height, int minimum_size, int step, double factor,
                             int *carts_n, double *score, unsigned int *path_hash, double *shapes);

#ifdef __cplusplus
}
#endif

#endif /* JDA_AMD_JDA_H_ */
