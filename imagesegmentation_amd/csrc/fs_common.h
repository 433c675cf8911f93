// Shared declarations of the fs_hip library (host side + launch prototypes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace fs {

// Grow-only device buffer owned by a context.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
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
};

constexpr int kMaxLabels = 128;   // label channels S = num_surfaces + 1 supported by the CRF kernels
constexpr int kMaxRadius = 15;    // largest CRF window radius (kernel_size)

// ---- launchers implemented in the .hip files (all asynchronous on `st`) ----
// stencils (k_stencils.hip)
void launch_smooth_depth_u16(const uint16_t *depth, float *out, int n, int H, int W, hipStream_t st);
void launch_smooth_depth_f32(const float *depth, float *out, int n, int H, int W, hipStream_t st);
void launch_normals(const float *D, const float *tanx, const float *tany, float *points, float *normals,
                    float *raw_angles, int n, int H, int W, hipStream_t st);
void launch_angle_filter(const float *raw_angles, const float *D, float *angles, int n, int H, int W, hipStream_t st);
void launch_curvature_mask(const float *D, const float *ax, const float *ay, float threshold, uint8_t *mask,
                           float *scores, int n, int H, int W, hipStream_t st);
void launch_majority_edges(const uint8_t *curv, const float *D, const float *ax, const float *ay, float alpha,
                           uint8_t *mask, int32_t *edges, int n, int H, int W, hipStream_t st);
// connected components (k_cc.hip)
struct CcWork {
    int32_t *parent;     // [n][N] union-find forest, root = smallest flat index of the component
    int32_t *count;      // [n][N] pixels per root
    int32_t *newid;      // [n][N] final id per surviving root
    int32_t *blocksum;   // [n][nblk]
    int32_t *nlabels;    // [n] largest id per image
};
size_t cc_blocks_per_image(int H, int W);
void launch_connected_components(const uint8_t *mask, const CcWork &w, int min_size_exclusive, int64_t *labels64,
                                 int32_t *labels32, int n, int H, int W, hipStream_t st);
// CRF (k_crf.hip)
struct CrfFrame {            // one image (or one padded tile treated as an image)
    int Hi, Wi;              // extent of Q / feature arrays
    int oy, ox, oh, ow;      // output window inside the image
    int S;                   // label channels
    int r;                   // window radius
};
size_t crf_half_offsets(int r);
size_t crf_sim_slack(int W, int r);      // floats of zeroed slack before and after the similarity planes
int crf_tile_height(int r);              // rows per tile of the per-tile active-label records
size_t crf_tile_rec_ints();              // ints per record
void crf_tile_grid(int H, int W, int r, int *tiles_x, int *tiles_y);
void launch_crf_sim(const uint16_t *depth, const uint8_t *rgb, const float *angles, const CrfParams &p,
                    const float *pos_terms, float *simH, int H, int W, int r, hipStream_t st);
void launch_crf_sim_explicit(const float *f1, const float *f2, const float *f3, const CrfParams &p, float *simH,
                             int Hi, int Wi, int r, hipStream_t st);
void launch_crf_init(const int32_t *patches, float *Q, int H, int W, int S, hipStream_t st);
void launch_rows_to_planes(const float *rows, float *planes, size_t P, int S, hipStream_t st);
void launch_planes_to_rows(const float *planes, float *rows, size_t P, int S, hipStream_t st);
// patch labels present within `grow` pixels of every tile, for n frames; max_active[n] must be zeroed
void launch_crf_tile_active(const int32_t *patches, int n, int H, int W, int r, int grow, int32_t *recs,
                            int32_t *max_active, hipStream_t st);
// mode bits for launch_crf_iter
enum { CRF_LAST = 1 };
// Q / potentials / unary are label planes.  tile_recs == nullptr: every label is carried (operator-level API);
// otherwise max_channels = largest per-tile channel count (max active + 3).
hipError_t launch_crf_iter(const float *simH, const float *Qin, float *Qout, float *potentials,
                           const float *unary_explicit, const int32_t *patches, const uint16_t *depth_raw,
                           int64_t *surfaces, const int32_t *tile_recs, int max_channels, const CrfFrame &f, float weight,
                           int mode, hipStream_t st);

}  // namespace fs
