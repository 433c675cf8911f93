/*
 * fs_hip.h -- C ABI of libfs_hip.so: MI355X (gfx950) implementation of the
 * smooth-surface detection + CRF pixel-assignment stage ("find_surfaces").
 *
 * Every entry point names the reference interface it replaces (paths relative to
 * the reference repository).  All functions return 0 on success and a negative
 * value on failure; fs_last_error() then holds a message.  A context is bound to
 * one GPU and one stream and is not thread safe; use one context per GPU/thread.
 * Host-pointer entry points copy in and out synchronously.  "_dev" entry points
 * take device pointers, enqueue on the context's stream and return after the
 * single host synchronisation the label count requires.
 *
 * Array layouts are the reference's: row-major [n][H][W][C] with the channel
 * last, depth uint16 (0 = invalid), rgb uint8, patches/surfaces int64,
 * depth_edges int32, everything else float32.
 */
#ifndef FS_HIP_H
#define FS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fs_ctx fs_ctx;

/* Creates a context for frames of height x width on GPU `device`, able to process
 * up to max_batch frames per call.  Builds what get_GPU_models() builds once
 * (find_surfaces.py:26-32): the viewing-angle tangent tables of get_angle_arrays()
 * (image_processing_models_GPU.py:137-149, angles from config.py:6-7), and loads
 * nothing from disk -- similarity parameters come through fs_set_gauss_params(). */
int fs_create(fs_ctx **out, int device, int height, int width, int max_batch);
void fs_destroy(fs_ctx *ctx);
const char *fs_last_error(const fs_ctx *ctx);   /* ctx may be NULL: last creation error */

/* Viewing angles in radians (config.py:6-7); defaults are 62 and 48.6 degrees. */
int fs_set_view_angles(fs_ctx *ctx, double view_x, double view_y);

/* The 13 values load_Gauss_parameters() returns (find_surfaces.py:221-226), in its
 * order: w_1,w_2,w_3,bias,theta_1_1,theta_2_1,theta_3_1,theta_1_2,theta_2_2,
 * theta_3_2,theta_2_3,theta_3_3,weight -- the argument list of conv_crf_Gauss()
 * (find_surfaces.py:291).  Thetas are squared inside, as the graph does (:339-341). */
int fs_set_gauss_params(fs_ctx *ctx, const float params[13]);

/* Switches the CRF to the logistic-regression pixel similarity of conv_crf_LR (find_surfaces.py:371-434,
 * find_surfaces(mode="LR")): p = coef[0..3] (weights of the depth, angle, colour and position distances,
 * clf.coef_), bias (clf.intercept_ + 4, load_pixel_similarity_parameters, :216-218), weight and
 * auxiliary_weight (LR_CRF_parameters, :22).  fs_set_gauss_params() switches back. */
int fs_set_lr_params(fs_ctx *ctx, const float params[7]);

/* CRF window radius, the kernel_size argument of conv_crf_Gauss / find_surfaces
 * (find_surfaces.py:23, 291, 497); default 7, at most 15. */
int fs_set_kernel_size(fs_ctx *ctx, int kernel_size);

/* Raises the number of frames one call may process (the context's buffers grow; a smaller value changes nothing).  The
 * handle stays valid, unlike destroying the context and creating a larger one.  No counterpart in the reference. */
int fs_set_max_batch(fs_ctx *ctx, int max_batch);

/* Use the caller's HIP stream (a hipStream_t passed as void*) instead of the context's own, so that the
 * "_dev" entry points are ordered with the caller's other work on that stream.  A NULL handle is HIP's legacy
 * default stream (what torch.cuda.current_stream().cuda_stream is unless the caller made a stream current).
 * fs_use_own_stream() goes back to the context's own stream, which is created hipStreamNonBlocking: work on it
 * is NOT ordered with the default stream, so a caller that stays on the own stream must synchronise on
 * fs_stream() (or wait on an event recorded there) before it touches the outputs of a "_dev" call or
 * overwrites its inputs.  Both calls drain the stream they leave.  No counterpart in the reference (TensorFlow
 * owns its streams). */
int fs_set_stream(fs_ctx *ctx, void *hip_stream);
int fs_use_own_stream(fs_ctx *ctx);
void *fs_stream(fs_ctx *ctx);

/* ---- stage operators: the four models of get_GPU_models() ------------------ */

/* smoothing_model(depth, grid): gaussian_filter_with_depth_factor_model_GPU()
 * (image_processing_models_GPU.py:110-135) with its defaults size=3, sigma=0.
 * depth, out: float32 [n][H][W].  The grid input of the reference has no effect
 * at sigma_2 = 0 and is not taken. */
int fs_smooth_depth(fs_ctx *ctx, int n, const float *depth, float *out);

/* normals_model_GPU() (image_processing_models_GPU.py:151-182): smoothed depth
 * float32 [n][H][W] -> angles [n][H][W][2], vectors [n][H][W][3],
 * points [n][H][W][3].  raw_angles (may be NULL) receives the unfiltered angles. */
int fs_normals(fs_ctx *ctx, int n, const float *smoothed, float *angles, float *vectors, float *points,
               float *raw_angles);

/* find_surfaces_model_GPU() (find_surfaces.py:229-288) with its default
 * arguments: smoothed depth -> patches int64 [n][H][W], depth edges int32.
 * curvature_scores / curvature_mask (may be NULL) expose the intermediate of
 * :232-262 (float32 / uint8 [n][H][W]) for testing. */
int fs_patches(fs_ctx *ctx, int n, const float *smoothed, int64_t *patches, int32_t *edges,
               float *curvature_scores, uint8_t *curvature_mask);

/* Connected components alone: Components() (image_processing_models_GPU.py:74-76,
 * tfa.image.connected_components) on a uint8 mask [n][H][W]; components with
 * <= min_size_exclusive pixels are dropped before numbering (pass 0 to keep all). */
int fs_connected_components(fs_ctx *ctx, int n, const uint8_t *mask, int min_size_exclusive, int64_t *labels,
                            int32_t *num_labels);

/* Connected components without any allocation.  No counterpart in the reference: it exists for callers that
 * cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  Same result as fs_connected_components, but every pointer is device memory (d_mask uint8
 * [n][H][W], d_labels int64 [n][H][W], d_num_labels int32 [n] or NULL), the work is enqueued on the context's
 * stream without synchronising, and the work space is a static device array of the library (about 14 MiB per
 * GPU, held until the process ends) instead of the context's buffers: the call uses no hipMalloc / hipFree.
 * Limit: n * height * width <= fs_noalloc_max_pixels() (1280 * 960); larger calls fail with a message.
 * All calls on one GPU share that array.  The entry point orders them: a call's stream waits for the previous call on
 * the same GPU (an event created with the first context; waiting and recording allocate nothing), so contexts and
 * threads that share a GPU take turns.  On a stream under capture that ordering is left out, and calls that overlap on
 * one GPU race. */
int fs_connected_components_noalloc(fs_ctx *ctx, int n, const uint8_t *d_mask, int min_size_exclusive,
                                    int64_t *d_labels, int32_t *d_num_labels);
size_t fs_noalloc_max_pixels(void);

/* conv_crf_Gauss(...).predict([unary, Q, f1, f2, f3, val], batch_size=1)
 * (find_surfaces.py:291-367, 517-518): one mean-field update on B tiles.
 *   unary [B][h][w][S], Q [B][h+2r][w+2r][S], f1 [B][h+2r][w+2r][3],
 *   f2 [..][6], f3 [..][5], val [S] (must be the one-hot of the last label, the
 *   only value the reference ever passes, find_surfaces.py:451-452);
 *   out [B][h][w][S]; potentials (may be NULL) [B][h][w][S] receives
 *   unary + weight * compatibility, the softmax argument. */
int fs_crf_step(fs_ctx *ctx, int B, int h, int w, int S, const float *unary, const float *Q, const float *f1,
                const float *f2, const float *f3, const float *val, float *out, float *potentials);

/* conv_crf_LR(...).predict([unary, Q, f1, f2, f3, f4, val], batch_size=1) (find_surfaces.py:371-434):
 * f1 [B][h+2r][w+2r][1] depth, f2 [..][2] angles, f3 [..][3] colour, f4 [..][2] (row, col); the rest as
 * fs_crf_step.  Needs fs_set_lr_params(). */
int fs_crf_step_lr(fs_ctx *ctx, int B, int h, int w, int S, const float *unary, const float *Q, const float *f1,
                   const float *f2, const float *f3, const float *f4, const float *val, float *out, float *potentials);

/* ---- fused path: find_surfaces() for a batch of frames ---------------------
 * find_surfaces(models, image_data, mode, kernel_size) (find_surfaces.py:497-
 * 532) for n frames, mode "Gauss" or "LR" according to the parameters set last: depth uint16 [n][H][W], rgb uint8 [n][H][W][3] ->
 * the arrays of the returned dict / of the out/{index} .npy files (:537-545):
 * angles, vectors, points_3d, patches, depth_edges, surfaces ("Q.npy"),
 * num_surfaces[n] = max(patches)+1.  Any output pointer may be NULL.
 * Host pointers; returns when every output has arrived.  More than 16 frames go through in chunks of 16 whose
 * copies run beside the kernels of the neighbouring chunks (page-locked arrays make them DMA transfers:
 * fs_host_alloc); such a call sends patches and surfaces as 16-bit and depth_edges as 8-bit images over the link and
 * widens them into the caller's int64 / int32 arrays on a few worker threads of its own (the link bounds this entry
 * point, and the label images are 6.2 of a 640x480 frame's 15.8 MB as they are, 1.5 MB narrow).  A call of up to 16
 * frames sends each group of outputs on its way as soon as the kernels that complete it are enqueued, and copies
 * inputs that arrive in pageable memory into page-locked memory first. */
int fs_find_surfaces_batch(fs_ctx *ctx, int n, const uint16_t *depth, const uint8_t *rgb, float *angles,
                           float *vectors, float *points, int64_t *patches, int32_t *edges, int64_t *surfaces,
                           int32_t *num_surfaces);

/* The same stage for one frame with the mean-field updates run as diagnostic builds of the production kernels (same
 * arithmetic, instruction for instruction, plus stores): the three calls of conv_crf_Gauss(...).predict at
 * find_surfaces.py:517-529 leave their softmax argument unary + weight * compatibility (:361) in
 * potentials [3][S][H][W] and their result (:363, before the Q[depth == 0] assignments of :520-530) in
 * probabilities [3][S][H][W], label planes, S = num_surfaces + 1, written at pixels with a depth and zero elsewhere.
 * max_channels: the S the caller's arrays were sized for; a frame with more fails (num_surfaces is set all the same, so
 * the caller can size the arrays and call again).  Host pointers, any output may be NULL; Gaussian similarity with the
 * window constants of the trained parameters only.  This is how the tests hold the production kernels' floating-point
 * results against a CPU evaluation; it is several times slower than fs_find_surfaces_batch. */
int fs_find_surfaces_trace(fs_ctx *ctx, const uint16_t *depth, const uint8_t *rgb, int max_channels, float *potentials,
                           float *probabilities, int64_t *patches, int64_t *surfaces, int32_t *num_surfaces);

/* Page-locked host memory for the arrays passed to fs_find_surfaces_batch: with it the uploads and downloads
 * are DMA transfers that overlap the kernels of neighbouring chunks (several times the throughput of
 * pageable arrays).  No counterpart in the reference.  fs_host_alloc returns NULL on failure. */
void *fs_host_alloc(size_t bytes);
void fs_host_free(void *p);

/* Same with all image arrays resident in device memory (num_surfaces stays a host
 * array).  Work is enqueued on the context's stream; the call returns after the
 * last kernel has been enqueued (it waits once internally for the label counts: with several frames in
 * hipStreamSynchronize before the updates are launched, with one frame by polling a page-locked word while
 * the updates, launched ahead of the counts, already run: fs_get_speculation_counts). */
int fs_find_surfaces_batch_dev(fs_ctx *ctx, int n, const uint16_t *d_depth, const uint8_t *d_rgb, float *d_angles,
                               float *d_vectors, float *d_points, int64_t *d_patches, int32_t *d_edges,
                               int64_t *d_surfaces, int32_t *num_surfaces);

/* Waits until everything the context has enqueued on its stream is done (the calls that take device pointers return
 * before their kernels have run).  No counterpart in the reference, whose calls are synchronous. */
int fs_synchronize(fs_ctx *ctx);

/* post_processing.apply_CRF (post_processing.py:4-19): one mean-field pass with
 * `labels` (int64 [n][H][W], e.g. final_surfaces) in the role of the patches, then
 * argmax; depth uint16, rgb uint8, angles float32 [n][H][W][2] as saved by
 * find_surfaces.  Host pointers.  out int64 [n][H][W].  Labels must lie in [0, 510] (the frame's channel count is
 * their maximum + 2); they are checked on the device, and a call that finds one outside fails without touching out. */
int fs_apply_crf(fs_ctx *ctx, int n, const int64_t *labels, const uint16_t *depth, const uint8_t *rgb,
                 const float *angles, int64_t *out);

/* ---- per-surface statistics ---------------------------------------------------
 * The pixel-level part of process_surfaces.extract_information_from_surface_data_and_preprocess_surfaces
 * (process_surfaces.py:672-687) for one frame, host pointers, in the reference's order:
 *   extract_individual_points_from_surfaces (:96-108)  -> patch_point_offsets int32 [S+1],
 *       patch_points float32 [H*W][2] = (x, y) grouped by patch label, raster order inside a label
 *   relabel_surfaces (:236-242), determine_pixel_counts_and_average_position_for_patches (:111-129)
 *       -> avg_pos float64 [S][2] = (x, y), counts float64 [S]
 *   determine_surface_patch_centroids (:81-93) -> centroids float32 [S][3], centroid_indices int32 [S][2] = (y, x);
 *       a label without patch pixels gets -1 entries at its own position
 *   remove_disconnected_components (:8-28) -> surfaces_out int64 [H][W]
 *   determine_histograms_and_average_normals (:245-297) -> hist_color float64 [256][256][S], hist_angle float64
 *       [40][40][S], hist_texture float64 [S][texture_dim], avg_angles float64 [S][2] (the input of
 *       utils.angles_to_normals); lab float64 [H][W][3] and texture float32 [H/4][W/4][texture_dim] (may be
 *       NULL) are inputs, as in the reference
 *   determine_neighboring_surfaces_and_border_centers (:31-68) -> neighbors float64 [S][S] (pair counts after the
 *       "< 4 -> 1" rule; the neighbour list of :70-77 is neighbors[i][j] > 3 for j >= 1), border_centers float64 [S][S][2]
 *   norm (:686) float32 [H][W], calculate_depth_extend (:332-346) -> depth_extend float32 [S][4][2]
 * S = num_surfaces; every label of surfaces and patches must lie in [0, S).  Output pointers may be NULL. */
int fs_surface_stats(fs_ctx *ctx, int num_surfaces, const int64_t *surfaces, const int64_t *patches,
                     const int32_t *depth_edges, const float *points_3d, const float *angles, const double *lab,
                     const float *texture, int texture_dim, int64_t *surfaces_out, int32_t *patch_point_offsets,
                     float *patch_points, double *avg_pos, double *counts, float *centroids, int32_t *centroid_indices,
                     double *hist_color, double *hist_angle, double *hist_texture, double *avg_angles, double *neighbors,
                     double *border_centers, float *norm, float *depth_extend);

/* The same with the frame's colours as rgb (uint8 [H][W][3], data["rgb"]) instead of lab: utils.rgb_to_Lab (utils.py:7-8,
 * process_surfaces.py:678-679) runs on the device, and lab_out (float64 [H][W][3], may be NULL) receives data["lab"]. */
int fs_surface_stats_rgb(fs_ctx *ctx, int num_surfaces, const int64_t *surfaces, const int64_t *patches,
                         const int32_t *depth_edges, const float *points_3d, const float *angles, const uint8_t *rgb,
                         const float *texture, int texture_dim, double *lab_out, int64_t *surfaces_out,
                         int32_t *patch_point_offsets, float *patch_points, double *avg_pos, double *counts, float *centroids,
                         int32_t *centroid_indices, double *hist_color, double *hist_angle, double *hist_texture,
                         double *avg_angles, double *neighbors, double *border_centers, float *norm, float *depth_extend);

/* The same stage with every array resident in device memory, for callers that keep the outputs of
 * fs_find_surfaces_batch_dev where they are: the arguments and layouts are those of fs_surface_stats, frame by
 * frame (process_surfaces.py:672-687 is a per-frame function), but every pointer is a device pointer.  Output
 * pointers may be NULL.  Work is enqueued on the context's stream; the call synchronises once to read the
 * range-check flag (labels, colour bins, angle bins) and returns its error like fs_surface_stats. */
int fs_surface_stats_dev(fs_ctx *ctx, int num_surfaces, const int64_t *d_surfaces, const int64_t *d_patches,
                         const int32_t *d_depth_edges, const float *d_points_3d, const float *d_angles, const double *d_lab,
                         const float *d_texture, int texture_dim, int64_t *d_surfaces_out, int32_t *d_patch_point_offsets,
                         float *d_patch_points, double *d_avg_pos, double *d_counts, float *d_centroids,
                         int32_t *d_centroid_indices, double *d_hist_color, double *d_hist_angle, double *d_hist_texture,
                         double *d_avg_angles, double *d_neighbors, double *d_border_centers, float *d_norm,
                         float *d_depth_extend);

/* The statistics stage for a batch of n frames resident in device memory, e.g. the outputs of
 * fs_find_surfaces_batch_dev where they lie: the loop a caller of process_surfaces.py:672-687 would run over its frames,
 * in one pass of batched kernels.  Input arrays hold the frames back to back ([n][H][W]...); num_surfaces[n] is a
 * host array.  The colours come either as d_lab (float64 [n][H][W][3], data["lab"]) or as d_rgb (uint8 [n][H][W][3]):
 * then utils.rgb_to_Lab (utils.py:7-8: sRGB -> CIE L*a*b*, D65) runs on the device and d_lab_out (may be NULL) receives
 * the converted image.  Every output array holds the frames' parts back to back, frame i's part laid out as in
 * fs_surface_stats with its own S = num_surfaces[i] (so the part of frame i starts at the sum of the sizes of frames
 * 0 .. i-1: e.g. hist_color at 65536 * sum(S[0..i)) doubles, patch_point_offsets at sum(S[0..i)) + i ints).  Output
 * pointers may be NULL.  Work is enqueued on the context's stream; the call synchronises once at the end to read the
 * range-check flag. */
int fs_surface_stats_batch_dev(fs_ctx *ctx, int n, const int32_t *num_surfaces, const int64_t *d_surfaces,
                               const int64_t *d_patches, const int32_t *d_depth_edges, const float *d_points_3d,
                               const float *d_angles, const double *d_lab, const uint8_t *d_rgb, const float *d_texture,
                               int texture_dim, double *d_lab_out, int64_t *d_surfaces_out, int32_t *d_patch_point_offsets,
                               float *d_patch_points, double *d_avg_pos, double *d_counts, float *d_centroids,
                               int32_t *d_centroid_indices, double *d_hist_color, double *d_hist_angle,
                               double *d_hist_texture, double *d_avg_angles, double *d_neighbors, double *d_border_centers,
                               float *d_norm, float *d_depth_extend);

/* chi_squared_distances_model_2D(pool_size, strides) (image_processing_models_GPU.py:189-200) applied to
 * one histogram stack: hist float64 [A][B][S] (e.g. hist_color [256][256][S] with pool (10,10), strides
 * (4,4), or hist_angle [40][40][S] with (4,4), (1,1): assemble_objects_rules.py:13-14) is cast to float32,
 * average-pooled (VALID), and out float32 [S][S] receives sum over bins of divide_no_nan((a-b)^2, a+b).
 * chi_squared_distances_model_1D (:202-212) on hist [K][S] is the call with A = 1, B = K, pool and
 * strides 1. */
int fs_chi_squared_distances(fs_ctx *ctx, const double *hist, int A, int B, int S, int pool_h, int pool_w,
                             int stride_h, int stride_w, float *out);

/* The same on a histogram stack that is already in device memory (the hist_color / hist_angle / hist_texture
 * outputs of fs_surface_stats_dev); d_out float32 [S][S] is a device pointer.  The work is enqueued on the
 * context's stream and the call returns without synchronising. */
int fs_chi_squared_distances_dev(fs_ctx *ctx, const double *d_hist, int A, int B, int S, int pool_h, int pool_w,
                                 int stride_h, int stride_w, float *d_out);

/* The pixel loop of evaluate_models.extract_IoU_information (evaluate_models.py:28-35): overlap float64
 * [num_detections][num_annotations] = number of pixels with (detected == o, annotation == a); the row and
 * column sums are the two count vectors of :33-34.  int64 [H][W] inputs, host pointers. */
int fs_overlap_counts(fs_ctx *ctx, const int64_t *annotation, const int64_t *detected, int num_annotations,
                      int num_detections, double *overlap);

/* calculate_nearest_points_calc (process_surfaces.py:638-646): `groups` point sets (points float32 [P][2],
 * set g = rows point_offsets[g] .. point_offsets[g+1]) and as many target lists (targets float32 [T][2],
 * target_offsets); out int32 [T][2] receives, for every target, the point of its own set with the smallest
 * squared distance (float32 arithmetic, first minimum in list order), cast to int32 like the reference's
 * result array. */
int fs_nearest_points(fs_ctx *ctx, int groups, const float *points, const int32_t *point_offsets, const float *targets,
                      const int32_t *target_offsets, int32_t *out);

/* Testing aid, no counterpart in the reference: the work buffers named by `letters` are overwritten in front of every
 * later fs_find_surfaces_batch(_dev) / statistics call of this context (q a b s r: float buffers, as NaN; t: the CRF's
 * tile records, as -1; p c n k u o m v e l w: integer work buffers, as 1), so that a kernel that uses a value the call has
 * not written shows in the results.  NULL or "" switches it off.  A new context starts with the value of the
 * environment variable FS_DEBUG_NAN, if set. */
int fs_set_debug_fill(fs_ctx *ctx, const char *letters);

/* ---- instrumentation -------------------------------------------------------
 * group_mask bit g set: HIP events are recorded on the context's stream around
 * every launch of kernel group g; fs_get_timing() synchronises and returns the
 * accumulated device time (ms) and launch count of a group, then resets it.
 * Groups: 0 stencils, 1 connected components, 2 crf similarity, 3 crf init,
 * 4 crf iteration (one launch = one mean-field update of one frame). */
int fs_set_timing(fs_ctx *ctx, int group_mask);
/* Work the CRF update actually executed in the last fs_find_surfaces_batch(_dev) / fs_apply_crf call of n frames
 * (no more than 16 for the host-pointer call, which works in chunks): listed (may be NULL) int32
 * [n][tiles_y][tiles_x] receives the number of patch labels each tile lists (its update carries those labels plus
 * label 0, the "unassigned" label and one representative of all others instead of the frame's S channels);
 * the tile grid and tile size come back through the other pointers.  Host pointers; synchronises. */
int fs_get_tile_labels(fs_ctx *ctx, int n, int32_t *listed, int *tiles_x, int *tiles_y, int *tile_w, int *tile_h);
int fs_get_timing(fs_ctx *ctx, int group, double *total_ms, int64_t *launches);
/* One-frame calls of fs_find_surfaces_batch(_dev) enqueue the three mean-field updates before the frame's label
 * count has reached the host (sized for what earlier frames needed, at least 64 label channels and 16 channels per
 * tile) and repeat them when the frame turns out not to fit; results are the same either way.  held / repeated (may
 * be NULL) receive how often each happened since the context was created.  No counterpart in the reference
 * (find_surfaces.py:514-517 computes num_surfaces on the host between the two model calls). */
int fs_get_speculation_counts(fs_ctx *ctx, int64_t *held, int64_t *repeated);

#ifdef __cplusplus
}
#endif
#endif /* FS_HIP_H */
