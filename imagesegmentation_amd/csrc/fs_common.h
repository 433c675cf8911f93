// Shared declarations of the fs_hip library (host side + launch prototypes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

struct fs_ctx;

namespace fs {

// what entry points outside fs_api.hip need of a context
struct CtxView { int device, H, W; hipStream_t stream; };
CtxView ctx_view(const fs_ctx *ctx);
int ctx_fail(fs_ctx *ctx, const std::string &message);      // records the message, returns -1
// work space of fs_connected_components_noalloc (fs_noalloc.hip): address of this GPU's copy, looked up once
hipError_t noalloc_scratch_init(int device, int32_t **out);

// Grow-only device buffer owned by a context.  A buffer that has to grow takes a quarter more than asked (requests
// that creep up, e.g. with the number of surfaces of a frame, then cost a few reallocations instead of one each:
// hipFree + hipMalloc is tens of milliseconds); the first allocation is exact.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        size_t want = bytes;
        if (p) {
            want = bytes + bytes / 4;
            hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e;
        }
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess && want != bytes) { (void)hipGetLastError(); want = bytes; e = hipMalloc(&p, want); }
        if (e == hipSuccess) cap = want;
        // A new buffer starts as zeros, whatever the allocator hands out (memory that another context of the process
        // has freed keeps its contents): a value that a kernel multiplies by a weight of exactly zero -- the
        // probabilities of a neighbour without depth, say -- must not be the NaN that old integer data reads as.
        // (FS_DEBUG_NEW_BUFFERS_AS_ONES for testing: =1 fills with 32-bit words of value 1 instead, in range as an index, a
        // label or a count wherever a kernel might take one from a word it has not written; =2 with all-ones bytes, which
        // read as -1, as NaN and as an address far outside every array: a kernel that takes an index from a word nobody
        // wrote faults, and one that uses such a float shows NaN in its results)
        static const int fill = [] {
            const char *v = getenv("FS_DEBUG_NEW_BUFFERS_AS_ONES");
            const int f = v && v[0] >= '1' && v[0] <= '2' ? v[0] - '0' : 0;
            if (f) fprintf(stderr, "fs_hip: FS_DEBUG_NEW_BUFFERS_AS_ONES=%d, new device buffers start as %s\n", f, f == 1 ? "32-bit ones" : "all-ones bytes");
            return f;
        }();
        if (e == hipSuccess)
            e = fill == 1 ? hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(p), 1, want / 4) : hipMemset(p, fill == 2 ? 0xff : 0, want);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);      // (done before a kernel on another stream writes there)
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Grow-only page-locked host buffer owned by a context (contents are lost when it grows).
struct HostBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes, unsigned flags = hipHostMallocDefault)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipHostFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipHostMalloc(&p, bytes, flags);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Parameters of the Gaussian pixel-similarity function and of the mean-field
// update (find_surfaces.py:221-226, 291, 339-341, 352-361), as the kernels use them.
struct CrfParams {
    float w1, w2, w3, bias;
    float t11, t21, t31;   // squared thetas of the position terms
    float t12, t22, t32;   // squared thetas of the depth term
    float t23, t33;        // squared thetas of colour / angle terms
    float weight;
    float evidence = 0.00005f;   // message added to the last ("unassigned") label: 0.00005 (find_surfaces.py:356-357, 451-452)
    // mode 1: logistic-regression similarity of conv_crf_LR (find_surfaces.py:371-434) instead of the Gaussian kernels;
    // then lr_coef = (depth, angle, colour, position) weights, bias = intercept + 4, evidence = auxiliary weight
    int mode = 0;
    float lr_coef[4] = {0.f, 0.f, 0.f, 0.f};
};

constexpr int kMaxLabels = 512;       // label channels S = num_surfaces + 1 of a frame (multiple of 32)
constexpr int kMaxTileChannels = 128; // channels the update kernels carry per tile in registers (listed labels + 3); tiles with more
                                      // take one pass per 128 channels (kCrfWideClass); also the limit of fs_crf_step
constexpr int kCrfWideClass = 1 << 10;   // crf_channel_class of such frames; launch_crf_iter_fused then wants wide_channels =
                                         // the most channels of a tile (descs[].msg must point at S label planes)
constexpr int kCrfMultiClass = 1 << 11;  // | quads: the kernels that carry 4 * quads channels per walk of the window (see crf_update_plan)
constexpr int kMaxRadius = 15;    // largest CRF window radius (kernel_size)

// ---- launchers implemented in the .hip files (all asynchronous on `st`) ----
// stencils (k_stencils.hip)
// rgb, featA (both or neither): also writes the (depth, r, g, b) half of the fused CRF update's feature records,
// colours times colour_scale = crf_colour_feature_scale(P) (what launch_crf_pack_features writes)
void launch_smooth_depth_u16(const uint16_t *depth, float *out, int n, int H, int W, hipStream_t st, const uint8_t *rgb = nullptr,
                             void *featA = nullptr, float colour_scale = 0.0f);
void launch_smooth_depth_f32(const float *depth, float *out, int n, int H, int W, hipStream_t st);
void launch_normals(const float *D, const float *tanx, const float *tany, float *points, float *normals,
                    float *raw_angles, int n, int H, int W, hipStream_t st);
// feat (may be null): float2 [n][H][W] that receives feat_scale * angles, the angle half of the CRF feature records
void launch_angle_filter(const float *raw_angles, const float *D, float *angles, void *feat, float feat_scale, int n, int H, int W,
                         hipStream_t st);
// todo: curvature_todo_ints(todo_frames, todo_cap) ints of work space for the pixels whose threshold test needs the
// correctly rounded evaluation (a second, small kernel handles them); may be null.  Its first todo_frames words
// are per-frame counters that must be zero when the kernels start: todo_zeroed says they are, otherwise a memset
// precedes the kernels.  Returns whether the counters were used (and are now non-zero until somebody clears them:
// launch_majority_edges does, given the pointer).
bool launch_curvature_mask(const float *D, const float *ax, const float *ay, float threshold, uint8_t *mask,
                           float *scores, int32_t *todo, int todo_cap, int todo_frames, bool todo_zeroed, int n, int H, int W,
                           hipStream_t st);
size_t curvature_todo_ints(int frames, int todo_cap);
// clear (may be null): n words zeroed on the way, one per frame
void launch_majority_edges(const uint8_t *curv, const float *D, const float *ax, const float *ay, float alpha,
                           uint8_t *mask, int32_t *edges, int32_t *clear, int n, int H, int W, hipStream_t st);
// connected components (k_cc.hip)
struct CcWork {
    int32_t *parent;     // [n][N] union-find forest, root = smallest flat index of the component
    int32_t *count;      // [n][N] pixels per root
    int32_t *newid;      // [n][N] final id per surviving root
    int32_t *blocksum;   // [n][nblk]
    int32_t *nlabels;    // [n] largest id per image
    int32_t *clear = nullptr;   // [n] words zeroed on the way (or null)
};
size_t cc_blocks_per_image(int H, int W);
void launch_connected_components(const uint8_t *mask, const CcWork &w, int min_size_exclusive, int64_t *labels64,
                                 int32_t *labels32, int n, int H, int W, hipStream_t st);
// components of a label image (equal non-zero values, 4-connectivity): parent[p] = flat index of the first
// pixel of p's component (-1 where the label is 0), count[root] = pixels of the component (count may be null: not wanted)
void launch_label_components(const int32_t *labels, int32_t *parent, int32_t *count, int n, int H, int W, hipStream_t st);
// per-surface statistics (k_stats.hip): one descriptor per frame of a batch; every pointer is device memory
struct StatsFrame {
    // inputs ([H][W] images of the frame)
    const int64_t *surfaces, *patches;
    const int32_t *edges;
    const float *points_3d, *angles;     // [H][W][3], [H][W][2]
    const double *lab;                   // [H][W][3]: the caller's, or lab_work when rgb is given
    const uint8_t *rgb;                  // [H][W][3] or null
    double *lab_work;                    // where the colour conversion writes (rgb != null), or null: only color_bin is wanted
    uint32_t *color_bin;                 // [H][W] colour-histogram bin (a << 8 | b) of every pixel, written by the colour conversion
                                         // when nobody asked for the Lab image itself (lab_work == null); else null
    const float *texture;                // [H/4][W/4][T] or null
    // label-indexed scratch (stats_carve_scratch)
    unsigned long long *acc, *best, *nb_sum;
    double *angle_sum, *tex_sum;
    uint32_t *hist_color, *hist_angle, *nb_count, *cnt, *tex_cnt, *extent, *blockcnt;
    int32_t *cpix;
    // outputs
    int64_t *surfaces_out;               // [H][W] relabelled, disconnected parts removed
    int32_t *offsets;                    // [S+1]
    float *patch_points;                 // [N][2] (x, y), grouped by patch label, raster order inside a label
    double *avg_pos, *counts;            // [S][2], [S]
    float *centroids;                    // [S][3]
    int32_t *centroid_idx;               // [S][2] (y, x)
    double *hist_color_out, *hist_angle_out, *hist_texture_out, *avg_angles;   // [256][256][S], [40][40][S], [S][T], [S][2]
    double *neighbors;                   // [S][S] symmetric pair counts after the "< 4 -> 1" rule
    double *border_centers;              // [S][S][2]
    float *norm;                         // [H][W]
    float *depth_extend;                 // [S][4][2]
    int S;                               // num_surfaces of the frame
    int pad_;
};
struct StatsBatchWork {
    int32_t *surf32, *pat32, *parent, *count;   // [n][H*W] each
    const double *lin;                          // [256] linearised sRGB channel values
    const float *lin32;                         // [256] the same, rounded to float32
    uint32_t *ext_all;                          // the batch's min / max slots
    size_t ext_words;
    int *err;                                   // 0 ok, 1 label out of range, 2 colour bin out of range, 3 angle bin out of range
};
struct StatsScratchSizes { size_t zero, ones, ext, plain; };
StatsScratchSizes stats_scratch_sizes(int S, int T, int nblk);
void stats_carve_scratch(StatsFrame &f, int T, int nblk, char *zero, char *ones, char *ext, char *plain);
int stats_blocks_per_frame(int H, int W);
// after_stage (may be empty) is called when the kernels that complete a group of outputs have been enqueued:
// 0 = lab_work; 1 = offsets, patch_points, avg_pos, counts, centroids, centroid_idx, surfaces_out; 2 = everything;
// and before the first kernel that reads a group of inputs: 10 = rgb, 11 = surfaces and patches, 12 = points_3d,
// 13 = edges, angles, lab (when it is an input), texture
using StatsStageHook = std::function<void(int)>;
hipError_t launch_surface_stats_batch(const StatsFrame *frames, const StatsFrame *frames_host, int n, const StatsBatchWork &w, int H, int W,
                                      int T, hipStream_t st, const StatsStageHook &after_stage = StatsStageHook());
// pairwise reductions (k_pairs.hip); device pointers.  pooled: S * PA * PB floats of scratch, out [S][S]
void launch_chi_squared(const double *hist, float *pooled, float *out, int A, int B, int S, int ph, int pw, int sh, int sw,
                        hipStream_t st);
// overlap[nd][na] (zeroed by the caller) += 1 per pixel; *err = 1 if a label is out of range
void launch_overlap_counts(const int64_t *annotation, const int64_t *detected, unsigned *overlap, int *err, int N, int nd, int na,
                           hipStream_t st);
// nearest point of group target_group[t] to target t (device pointers); out [T][2], -1 for an empty group
void launch_nearest_points(const float *points, const int32_t *point_offsets, const float *targets, const int32_t *target_group,
                           int T, int32_t *out, hipStream_t st);
// CRF (k_crf.hip)
struct CrfFrameDesc {            // one frame of a batched launch (device-resident array of these)
    const float *sim0;           // similarity plane 0 of the frame (slack lies in front of it)
    const float *Qin;            // label planes [S][Hi*Wi]
    float *Qout;                 // label planes [S][oh*ow] or null
    float *potentials;           // label planes [S][oh*ow] or null
    float *trace_q;              // fused update, diagnostic kernels only: all S planes of the new probabilities, or null
    float *msg;                  // wide updates only: label planes [S][Hi*Wi] for the raw messages between the passes, or null
    const float *unary;          // explicit unary planes [S][oh*ow] or null (then synthesised from patches)
    const int32_t *patches;      // [Hi*Wi] or null
    const uint16_t *depth_raw;   // [Hi*Wi] or null: invalid-depth override (and depth feature of the fused update)
    const float4 *featA;         // fused update: [Hi*Wi] (depth, r, g, b) as floats, or null
    const float2 *featB;         // fused update: [Hi*Wi] (angle0, angle1), or null
    int64_t *surfaces;           // [oh*ow] or null: argmax output
    const int32_t *tile_recs;    // per-tile active-label records or null (every label carried)
    // Fused update only.  A launch may be enqueued before the frame's label count has reached the host (one frame per
    // call: the read-back would leave the GPU idle for its round trip).  `spec` then points at the device words
    // (largest patch label, most listed labels of a tile) that the preceding kernels write; the update derives S from
    // them and leaves the frame alone unless S <= spec_cap (what the Q planes hold) and every tile's channels fit the
    // launched kernel variant.  The host evaluates the same test once the words arrive and, if it fails, runs the
    // update again with what it then knows.  Null: S below is the label count.
    // publish (with spec, first update only; else null): page-locked word through which the two device words reach the
    // host, which polls it instead of putting a copy command into the stream (the updates would queue behind it, about
    // 10 us per call): the launch's first block stores largest label | listed labels << 32 | sequence number << 48, the
    // sequence number being the incremented device word *publish_seq (so the host can tell this call's word from the last).
    const int32_t *spec;
    unsigned long long *publish;
    unsigned *publish_seq;
    int S;                       // label channels
    int spec_cap;
};
struct CrfGeom {                 // common to all frames of a launch
    int Hi, Wi;                  // extent of Q / similarity / feature arrays
    int oy, ox, oh, ow;          // output window inside the image
    int r;                       // window radius
};
size_t crf_half_offsets(int r);
size_t crf_sim_slack(int W, int r);      // floats of zeroed slack in front of and behind a frame's planes
size_t crf_tile_rec_ints();              // ints per per-tile record
void crf_tile_grid(int H, int W, int *tiles_x, int *tiles_y);
int crf_channel_class(int channels);     // kernel variant able to carry `channels` channels per tile
void launch_crf_sim(const uint16_t *depth, const uint8_t *rgb, const float *angles, const CrfParams &p,
                    const float *pos_terms, float *simH, size_t sim_stride, int nframes, int H, int W, int r,
                    hipStream_t st);
void launch_crf_sim_explicit(const float *f1, const float *f2, const float *f3, const CrfParams &p, float *simH,
                             int Hi, int Wi, int r, hipStream_t st);
void launch_crf_init(const CrfFrameDesc *descs, int nframes, int H, int W, hipStream_t st);   // writes descs[].Qin
void launch_rows_to_planes(const float *rows, float *planes, size_t P, int S, hipStream_t st);
void launch_planes_to_rows(const float *planes, float *rows, size_t P, int S, hipStream_t st);
// patch labels present within `grow` pixels of every tile_w x tile_h tile, for n frames; max_active[n] must be zeroed
// clear (may be null): clear_words words zeroed on the way (the tile queues of the updates that follow)
void launch_crf_tile_active(const int32_t *patches, int n, int H, int W, int tile_w, int tile_h, int grow, int32_t *recs,
                            int32_t *max_active, hipStream_t st, unsigned *clear = nullptr, int clear_words = 0);
// int64 label images [n][N] -> int32; stats[n] (zeroed) receives every frame's largest label, stats[n + frame] = 1 if a label
// of the frame lies outside [0, max_label] (device pointers)
void launch_crf_labels_to_int32(const int64_t *labels, int32_t *out, int32_t *stats, int n, int N, int max_label, hipStream_t st);
// number of listed labels of `total` consecutive tile records -> out[total] (device pointers)
void launch_crf_tile_counts(const int32_t *recs, int32_t *out, int total, hipStream_t st);
// mode bits for launch_crf_iter
enum { CRF_LAST = 1, CRF_FIRST = 2 };
// one mean-field update of nframes frames whose per-tile channel count fits `channel_class`
hipError_t launch_crf_iter(const CrfFrameDesc *descs, int nframes, int channel_class, const CrfGeom &g, float weight,
                           float evidence, int mode, hipStream_t st);
// conv_crf_LR.predict for one tile on explicit pixel-major tensors (device pointers)
void launch_crf_lr_explicit(const float *unary, const float *Q, const float *f1, const float *f2, const float *f3, const float *f4,
                            const float coef[4], float bias, float weight, float aux, float *out, float *potentials, int h, int w,
                            int r, int S, hipStream_t st);
// same update with the similarity evaluated in the kernel from descs[].featA / featB (no similarity planes);
// wtab = per-offset weights [(2r+1)^2][4] = (w1', w2', w3', 0)
int crf_fused_tile_height(int r);        // 16 or 8 rows (tile width 32), 0 if the window is too large for the fused kernel
void crf_fused_tile_grid(int H, int W, int r, int *tiles_x, int *tiles_y);
// angles == nullptr: featB already holds crf_angle_feature_scale(P) * angles (written by the angle filter)
void launch_crf_pack_features(const uint16_t *depth, const uint8_t *rgb, const float *angles, void *featA, void *featB,
                              size_t total, const CrfParams &P, hipStream_t st);
float crf_angle_feature_scale(const CrfParams &P);
float crf_colour_feature_scale(const CrfParams &P);
// CRF_FIRST: Q is synthesised from descs[].patches instead of read from Qin.  Q planes are compact: only the
// planes of a tile's listed labels and of its representative are written (see k_crf.hip).
// tile_queue: kCrfQueueBytes of zeroed device memory that no other launch in flight uses (the caller keeps
// kCrfQueueSlots of them and clears them all with one memset; launches
// that share it must be ordered on one stream).
constexpr size_t kCrfQueueStride = 128, kCrfQueueBytes = 8 * kCrfQueueStride;
constexpr int kCrfQueueSlots = 64;
// trace: the diagnostic build of the kernel, which also fills descs[].potentials and descs[].trace_q (Gaussian similarity
// with uniform window constants only)
hipError_t launch_crf_iter_fused(const CrfFrameDesc *descs, int nframes, int channel_class, const CrfGeom &g,
                                 const CrfParams &P, const float *wtab, int mode, unsigned *tile_queue, hipStream_t st,
                                 bool trace = false, int wide_channels = 0, int a_min = 0, int a_max = 1 << 30, int32_t *tile_order = nullptr);
// tile_order: device scratch of crf_tile_order_ints(tiles of the launch) ints for the launches that sort their tiles
// (kCrfMultiClass), reusable by the next launch on the same stream
size_t crf_tile_order_ints(size_t tiles);
// a_min, a_max: the launch works on the tiles that list that many labels and skips the others (see FusedLaunch)
// One update of frames of class channel_class (crf_channel_class of their fullest tile) as a sequence of launches, each
// with the kernel variant that is fastest for the tiles it takes: out[i] = (class to launch, a_min, a_max); every launch
// needs a tile queue of its own.  Returns the number of launches (1: no split).
struct CrfSplit { int channel_class, a_min, a_max; };
constexpr int kCrfMaxSplits = 4;
// max_listed: most labels any tile of the frames lists (a launch for tiles beyond a variant's reach is left out when there are none)
int crf_update_plan(const CrfParams &P, int r, int channel_class, int max_listed, bool trace, CrfSplit (&out)[kCrfMaxSplits]);

}  // namespace fs
