// LDS-tiled stencil kernels of the surface-detection front end (gfx950).
//
// Every kernel takes a batch of frames (blockIdx.z = frame), stages a tile plus
// halo in LDS with zero fill outside the frame (the reference's SAME padding)
// and evaluates one pixel per lane.  Arithmetic is fp32 with one rounding per
// operation in the order the reference graph applies them; the file is compiled
// with -ffp-contract=off and hipcc's default correctly-rounded fp32 divide/sqrt,
// so the discrete outputs that depend on float comparisons (curvature mask,
// depth edges) are bit-identical to an unfused CPU evaluation.
#include <algorithm>

#include "fs_common.h"

namespace fs {

namespace {

constexpr int TW = 32;   // tile width  (lanes 0..31 of each half wave run along x: 128-byte rows)
constexpr int TH = 8;    // tile height
constexpr int NT = TW * TH;

__device__ __forceinline__ float div_no_nan(float a, float b) { return b == 0.0f ? 0.0f : a / b; }

// Cooperative tile load with zero fill: tile[(TH+2R)][(TW+2R)] <- img[y0-R.., x0-R..]
template <int R, typename T>
__device__ __forceinline__ void load_tile(float (*tile)[TW + 2 * R + 1], const T *img, int H, int W, int x0, int y0)
{
    constexpr int LW = TW + 2 * R, LH = TH + 2 * R;
    for (int i = threadIdx.x; i < LW * LH; i += NT) {
        int ly = i / LW, lx = i - ly * LW;
        int y = y0 + ly - R, x = x0 + lx - R;
        float v = 0.0f;
        if (y >= 0 && y < H && x >= 0 && x < W) v = (float)img[(size_t)y * W + x];
        tile[ly][lx] = v;
    }
}

// ---------------------------------------------------------------------------
// Depth smoothing: mean of the valid (>= 0.001) depths of the 7x7 window, 0 at
// invalid centres.  Follows gaussian_filter_with_depth_factor_model_GPU
// (image_processing_models_GPU.py:110-135) with sigma_1 = sigma_2 = 0.
// ---------------------------------------------------------------------------
constexpr int STH = 32;   // tile height of the 16-bit smoothing: 256 lanes x 4 rows (the halo is 1.4x the tile instead of 2.1x)

// rgb / feat (16-bit depths only, may be null): the kernel also writes the (depth, r, g, b) half of the CRF's feature
// records, feat[p] = (depth, cscale r, cscale g, cscale b), for the pixels it smooths (launch_crf_pack_features' values).
template <typename T>
__global__ __launch_bounds__(NT) void smooth_depth_kernel(const T *__restrict__ depth, float *__restrict__ out, int H, int W,
                                                          const uint8_t *__restrict__ rgb, float4 *__restrict__ feat, float cscale)
{
    constexpr int R = 3;
    constexpr int LTH = sizeof(T) == 2 ? STH : TH;          // rows of the tile
    __shared__ float tile[LTH + 2 * R][TW + 2 * R + 1];
    const size_t fo = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * LTH;
    for (int i = threadIdx.x; i < (LTH + 2 * R) * (TW + 2 * R); i += NT) {
        const int ly = i / (TW + 2 * R), lx = i - ly * (TW + 2 * R);
        const int y = y0 + ly - R, x = x0 + lx - R;
        tile[ly][lx] = (y >= 0 && y < H && x >= 0 && x < W) ? (float)depth[fo + (size_t)y * W + x] : 0.0f;
    }
    __syncthreads();
    const int tx = threadIdx.x % TW, x = x0 + tx;
    if (sizeof(T) == 2) {
        // 16-bit depths: the sums are integers below 2^24, exact in fp32 in any order, so the window is
        // summed along rows first (once per row and column of the tile) and then along columns
        __shared__ float hsum[LTH + 2 * R][TW], hcnt[LTH + 2 * R][TW];
        for (int i = threadIdx.x; i < (LTH + 2 * R) * TW; i += NT) {
            const int ly = i / TW, lx = i - ly * TW;
            float s = 0.0f, c = 0.0f;
#pragma unroll
            for (int wx = 0; wx <= 2 * R; ++wx) {
                const float d = tile[ly][lx + wx];
                s = s + d;                                  // an invalid depth is 0
                c = c + (d < 0.001f ? 0.0f : 1.0f);
            }
            hsum[ly][lx] = s;
            hcnt[ly][lx] = c;
        }
        __syncthreads();
        if (x >= W) return;
        for (int ty = threadIdx.x / TW; ty < LTH; ty += NT / TW) {
            const int y = y0 + ty;
            if (y >= H) break;
            float sums = 0.0f, val = 0.0f;
#pragma unroll
            for (int wy = 0; wy <= 2 * R; ++wy) {
                sums = sums + hcnt[ty + wy][tx];
                val = val + hsum[ty + wy][tx];
            }
            const float c = tile[ty + R][tx + R];
            const size_t q = fo + (size_t)y * W + x;
            out[q] = div_no_nan(val, sums) * (c < 0.001f ? 0.0f : 1.0f);
            if (feat) feat[q] = make_float4(c, cscale * (float)rgb[3 * q], cscale * (float)rgb[3 * q + 1], cscale * (float)rgb[3 * q + 2]);
        }
    } else {
        const int ty = threadIdx.x / TW, y = y0 + ty;
        if (x >= W || y >= H) return;
        float sums = 0.0f, val = 0.0f;
#pragma unroll
        for (int wy = 0; wy <= 2 * R; ++wy)
#pragma unroll
            for (int wx = 0; wx <= 2 * R; ++wx) {
                float d = tile[ty + wy][tx + wx];
                float ind = d < 0.001f ? 0.0f : 1.0f;
                sums = sums + ind;
                val = val + ind * d;
            }
        const float c = tile[ty + R][tx + R];
        out[fo + (size_t)y * W + x] = div_no_nan(val, sums) * (c < 0.001f ? 0.0f : 1.0f);
    }
}

// ---------------------------------------------------------------------------
// Back-projection, normals and raw angles: normals_model_GPU
// (image_processing_models_GPU.py:151-178) + angle_calculator (:63-72).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float signf(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

__global__ __launch_bounds__(NT) void normals_kernel(const float *__restrict__ D, const float *__restrict__ tanx,
                                                     const float *__restrict__ tany, float *__restrict__ points,
                                                     float *__restrict__ normals, float *__restrict__ raw_angles,
                                                     int H, int W)
{
    constexpr int R = 1;
    __shared__ float tile[TH + 2 * R][TW + 2 * R + 1];
    const size_t fo = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    load_tile<R>(tile, D + fo, H, W, x0, y0);
    __syncthreads();
    const int tx = threadIdx.x % TW, ty = threadIdx.x / TW;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= W || y >= H) return;
    const float dc = tile[ty + 1][tx + 1];
    const float txc = tanx[x], tyc = tany[y];
    // neighbours' points; a neighbour outside the frame is the zero vector (tf.pad, :163-166)
    const float du = tile[ty][tx + 1], dd = tile[ty + 2][tx + 1], dl = tile[ty + 1][tx], dr = tile[ty + 1][tx + 2];
    const float tyu = y > 0 ? tany[y - 1] : 0.0f, tyd = y < H - 1 ? tany[y + 1] : 0.0f;
    const float txl = x > 0 ? tanx[x - 1] : 0.0f, txr = x < W - 1 ? tanx[x + 1] : 0.0f;
    const size_t i = fo + (size_t)y * W + x;
    if (du == 0.0f && dd == 0.0f && dl == 0.0f && dr == 0.0f) {
        // no depth around the pixel (a quarter of a sensor frame, in blobs): both differences are the zero
        // vector, and the formulas below give a zero normal (divide_no_nan) and zero angles (acos(0) * sign(0))
        points[3 * i + 0] = txc * dc;
        points[3 * i + 1] = tyc * dc;
        points[3 * i + 2] = dc;
        normals[3 * i + 0] = 0.0f;
        normals[3 * i + 1] = 0.0f;
        normals[3 * i + 2] = 0.0f;
        raw_angles[2 * i + 0] = 0.0f;
        raw_angles[2 * i + 1] = 0.0f;
        return;
    }
    float a[3], b[3], v[3];
    a[0] = txc * du - txc * dd;   // sub_1 = P[y-1] - P[y+1]
    a[1] = tyu * du - tyd * dd;
    a[2] = du - dd;
    b[0] = txr * dr - txl * dl;   // sub_2 = P[x+1] - P[x-1]
    b[1] = tyc * dr - tyc * dl;
    b[2] = dr - dl;
    v[0] = a[1] * b[2] - a[2] * b[1];
    v[1] = a[2] * b[0] - a[0] * b[2];
    v[2] = a[0] * b[1] - a[1] * b[0];
    const float cond = -(v[2] > 0.0f ? 1.0f : 0.0f);
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = 2.0f * (cond * v[i]) + v[i];
    const float nrm = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    float n[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) n[i] = div_no_nan(v[i], nrm);
    const float v1[3] = {n[0] * 0.0f, n[1] * 1.0f, n[2] * 1.0f};
    const float v2[3] = {n[0] * 1.0f, n[1] * 0.0f, n[2] * 1.0f};
    const float n1 = sqrtf((v1[0] * v1[0] + v1[1] * v1[1]) + v1[2] * v1[2]);
    const float n2 = sqrtf((v2[0] * v2[0] + v2[1] * v2[1]) + v2[2] * v2[2]);
    const float a1 = acosf(fminf(fmaxf(div_no_nan(-v1[2], n1), -1.0f), 1.0f)) * signf(v1[1]);
    const float a2 = acosf(fminf(fmaxf(div_no_nan(-v2[2], n2), -1.0f), 1.0f)) * (-signf(v2[0]));
    points[3 * i + 0] = txc * dc;
    points[3 * i + 1] = tyc * dc;
    points[3 * i + 2] = dc;
    normals[3 * i + 0] = n[0];
    normals[3 * i + 1] = n[1];
    normals[3 * i + 2] = n[2];
    raw_angles[2 * i + 0] = a1;
    raw_angles[2 * i + 1] = a2;
}

// ---------------------------------------------------------------------------
// Depth-gated 7x7 mean of the two angle channels:
// uniform_filter_with_depth_cutoff + Depth_Cutoff
// (image_processing_models_GPU.py:102-108, 86-88).
// ---------------------------------------------------------------------------
constexpr int ATH = 16;   // tile height of the angle filter: 256 lanes x 2 vertically adjacent pixels

// one tap of one pixel's window: product test, band flag, the three running sums
#define FS_ANGLE_TAP(Q, LO, HI, BAND, SUMS, V0, V1)        \
    {                                                      \
        const float m_ = (Q) > (HI) ? 0.0f : 1.0f;         \
        BAND |= (Q) >= (LO) && (Q) <= (HI);                \
        SUMS = SUMS + m_;                                  \
        V0 = V0 + m_ * a0;                                 \
        V1 = V1 + m_ * a1;                                 \
    }

// feat (may be null) receives the filtered angles times feat_scale: the angle half of the feature records the CRF
// update stages, written here while the values are in registers instead of by a pass of its own.
__global__ __launch_bounds__(NT) void angle_filter_kernel(const float *__restrict__ raw, const float *__restrict__ D,
                                                          float *__restrict__ angles, float2 *__restrict__ feat, float feat_scale,
                                                          int H, int W)
{
    constexpr int R = 3;
    constexpr int LW = TW + 2 * R, LH = ATH + 2 * R;
    __shared__ float td[LH][LW + 1];
    __shared__ float ta0[LH][LW + 1];
    __shared__ float ta1[LH][LW + 1];
    const size_t fo = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * ATH;
    for (int i = threadIdx.x; i < LW * LH; i += NT) {
        int ly = i / LW, lx = i - ly * LW;
        int y = y0 + ly - R, x = x0 + lx - R;
        float d = 0.0f, a0 = 0.0f, a1 = 0.0f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            size_t j = fo + (size_t)y * W + x;
            d = D[j];
            float2 a = reinterpret_cast<const float2 *>(raw)[j];
            a0 = a.x;
            a1 = a.y;
        }
        td[ly][lx] = d;
        ta0[ly][lx] = a0;
        ta1[ly][lx] = a1;
    }
    __syncthreads();
    // A lane owns the pixels (ty, tx) and (ty + 1, tx): their windows share six of seven rows, so the eight
    // rows are read once and each tap feeds the pixel(s) whose window holds it.  Every pixel still adds its
    // 49 taps in window order (rows top to bottom, left to right), which fixes the rounding of its sums.
    const int tx = threadIdx.x % TW, ty = (threadIdx.x / TW) * 2;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= W || y >= H) return;
    // Weight of a tap: m = |divide_no_nan(dj - dc, dc)| > 0.3 ? 0 : 1.  The window is summed with a product
    // test, q = |fl(dj - dc)| against 0.3f |dc|, instead of the division; a pixel that met the band around the
    // threshold is summed again with the division.  Why the band suffices: both forms start from the same
    // rounded difference q.  The reference rounds q / |dc| once (relative error <= 2^-24 = 6.0e-8) before it
    // compares with 0.3f, so it gates exactly when q / |dc| exceeds 0.3f by more than half an ulp of 0.3f
    // (5.0e-8 relative): its decision flips somewhere in q in [0.3f |dc|, 0.3f |dc| (1 + 5.0e-8)].  The bounds
    // lo / hi carry two roundings of their own (the product with |dc| and the product with the band factor,
    // 1.2e-7 together) and the factors 1 -+ 4e-7 round to 1 - 4.17e-7 and 1 + 3.58e-7, so
    // hi >= 0.3f |dc| (1 + 2.4e-7) lies above every q at which the reference can still pass a tap and
    // lo <= 0.3f |dc| (1 - 3.0e-7) below every q at which it can gate one: outside [lo, hi] the product test
    // and the division agree, with a margin of more than four times the largest error.
    // tests/test_gpu_predicates.py puts taps within a few ulp of both sides of the threshold.
    // dc == 0: divide_no_nan gives 0 and every weight is 1 (thresholds +inf).
    const float dcA = td[ty + R][tx + R], dcB = td[ty + 1 + R][tx + R];
    const float loA = dcA == 0.0f ? INFINITY : 0.3f * fabsf(dcA) * (1.0f - 4e-7f);
    const float hiA = dcA == 0.0f ? INFINITY : 0.3f * fabsf(dcA) * (1.0f + 4e-7f);
    const float loB = dcB == 0.0f ? INFINITY : 0.3f * fabsf(dcB) * (1.0f - 4e-7f);
    const float hiB = dcB == 0.0f ? INFINITY : 0.3f * fabsf(dcB) * (1.0f + 4e-7f);
    float sumsA = 0.0f, v0A = 0.0f, v1A = 0.0f, sumsB = 0.0f, v0B = 0.0f, v1B = 0.0f;
    bool bandA = false, bandB = false;
    // one region row: the taps go to the upper pixel (A), the lower one (B) or both.  The rows stay a loop
    // (fully unrolled, the compiler reads all 168 values first and the registers cost the occupancy).
    auto row = [&](int wy, bool toA, bool toB) {
#pragma unroll
        for (int wx = 0; wx <= 2 * R; ++wx) {
            const float dj = td[ty + wy][tx + wx], a0 = ta0[ty + wy][tx + wx], a1 = ta1[ty + wy][tx + wx];
            if (toA) {
                const float q = fabsf(dj - dcA);
                FS_ANGLE_TAP(q, loA, hiA, bandA, sumsA, v0A, v1A)
            }
            if (toB) {
                const float q = fabsf(dj - dcB);
                FS_ANGLE_TAP(q, loB, hiB, bandB, sumsB, v0B, v1B)
            }
        }
    };
    row(0, true, false);
#pragma unroll 1
    for (int wy = 1; wy <= 2 * R; ++wy) row(wy, true, true);
    row(2 * R + 1, false, true);
    auto exact = [&](int oy, float dc, float &sums, float &v0, float &v1) {
        sums = 0.0f; v0 = 0.0f; v1 = 0.0f;
        for (int wy = 0; wy <= 2 * R; ++wy)
            for (int wx = 0; wx <= 2 * R; ++wx) {
                const float dj = td[ty + oy + wy][tx + wx];
                const float m = fabsf(div_no_nan(dj - dc, dc)) > 0.3f ? 0.0f : 1.0f;
                sums = sums + m;
                v0 = v0 + m * ta0[ty + oy + wy][tx + wx];
                v1 = v1 + m * ta1[ty + oy + wy][tx + wx];
            }
    };
    if (bandA) exact(0, dcA, sumsA, v0A, v1A);
    if (bandB) exact(1, dcB, sumsB, v0B, v1B);
    const size_t o = fo + (size_t)y * W + x;
    float2 *out = reinterpret_cast<float2 *>(angles) + o;
    const float2 rA = make_float2(div_no_nan(v0A, sumsA), div_no_nan(v1A, sumsA));
    const float2 rB = make_float2(div_no_nan(v0B, sumsB), div_no_nan(v1B, sumsB));
    out[0] = rA;
    if (y + 1 < H) out[W] = rB;
    if (feat) {
        feat[o] = make_float2(feat_scale * rA.x, feat_scale * rA.y);
        if (y + 1 < H) feat[o + W] = make_float2(feat_scale * rB.x, feat_scale * rB.y);
    }
}
#undef FS_ANGLE_TAP

// ---------------------------------------------------------------------------
// Curvature score + threshold (find_surfaces.py:232-262, depth = 4).
// ax[c] = |tanx[c-1] - tanx[c+1]|, ay[r] = |tany[r-1] - tany[r+1]| are
// precomputed on the host with the same fp32 operations.
// ---------------------------------------------------------------------------
// depth(oy, ox) = smoothed depth at the centre's row + oy, column + ox
// One direction (d0, d1) of the reference's evaluation, one fp32 operation per TF op: curs[k-1] = the running sum
// `cur` after step k (find_surfaces.py:246-259); the total adds the 32 values direction by direction.
template <typename Depth>
__device__ __forceinline__ void curvature_direction_exact(Depth depth, int d0, int d1, float dc, float dfx, float dfy, float (&curs)[4])
{
    float cur = 0.0f, prevd = 0.0f, px = 0.0f, py = 0.0f, pz = dc;
#pragma unroll
    for (int k = 1; k <= 4; ++k) {
        const float nx = dfx * (float)k, ny = dfy * (float)k;
        const float nz = depth(-k * d0, -k * d1);
        const float e0 = nx - px, e1 = ny - py, e2 = nz - pz;
        const float a1 = (e0 * e0 + e1 * e1) + e2 * e2;
        const float g0 = nx - 0.0f, g1 = ny - 0.0f, g2 = nz - dc;
        const float a2 = (g0 * g0 + g1 * g1) + g2 * g2;
        const float dif1 = sqrtf(a1), dif2 = sqrtf(a2);
        const float score = div_no_nan(dif1 + prevd, dif2) - 1.0f;
        cur = cur + score;
        curs[k - 1] = cur;
        prevd = dif2;
        px = nx; py = ny; pz = nz;
    }
}

template <bool FAST, typename Depth>
__device__ __forceinline__ float curvature_total(Depth depth, float dc, float dfx, float dfy)
{
    constexpr int R = 4;
    float total = 0.0f;
    const int d0s[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    const int d1s[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    if (FAST) {
        // the lateral parts of both distances do not depend on the direction: consecutive points are
        // (dfx, dfy) apart, point k is k (dfx, dfy) from the centre
        const float q = fmaf(dfx, dfx, dfy * dfy);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            float cur = 0.0f, prevd = 0.0f, pz = dc;
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                const float nz = depth(-k * d0s[d], -k * d1s[d]);
                const float e2 = nz - pz, g2 = nz - dc;
                const float a1 = fmaf(e2, e2, q), a2 = fmaf(g2, g2, (float)(k * k) * q);
                // one reciprocal square root gives both |p_k - p_0| and its reciprocal (callers keep
                // centres without depth, whose a2 can be 0, away from this path)
                const float rs = __builtin_amdgcn_rsqf(a2);
                const float dif1 = __builtin_amdgcn_sqrtf(a1), dif2 = a2 * rs;
                cur = cur + (fmaf(dif1 + prevd, rs, -1.0f));
                total = total + cur;
                prevd = dif2;
                pz = nz;
            }
        }
        return total;
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        float curs[R];
        curvature_direction_exact(depth, d0s[d], d1s[d], dc, dfx, dfy, curs);
#pragma unroll
        for (int k = 0; k < R; ++k) total = total + curs[k];
    }
    return total;
}

__global__ __launch_bounds__(NT) void curvature_mask_kernel(const float *__restrict__ D, const float *__restrict__ ax,
                                                            const float *__restrict__ ay, float threshold,
                                                            uint8_t *__restrict__ mask, float *__restrict__ scores,
                                                            int32_t *__restrict__ todo, int32_t *__restrict__ todo_list, int todo_cap,
                                                            int H, int W)
{
    constexpr int R = 4;
    __shared__ float tile[TH + 2 * R][TW + 2 * R + 1];
    const size_t fo = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    load_tile<R>(tile, D + fo, H, W, x0, y0);
    __syncthreads();
    const int tx = threadIdx.x % TW, ty = threadIdx.x / TW;
    const int c = x0 + tx, r = y0 + ty;
    if (c >= W || r >= H) return;
    const size_t i = fo + (size_t)r * W + c;
    if (r < R || r >= H - R || c < R || c >= W - R) {   // zero border of tf.pad (:262)
        mask[i] = 0;
        if (scores) scores[i] = 0.0f;
        return;
    }
    const float dc = tile[ty + R][tx + R];
    const float dfx = (ax[c] * dc) * 0.5f;
    const float dfy = (ay[r] * dc) * 0.5f;
    // Only the comparison with the threshold leaves this kernel (unless the caller asks for the
    // scores), so the score is first evaluated with the hardware's approximate sqrt and reciprocal
    // square root (1 ulp = 1.2e-7 each) and fused multiply-adds, and with the lateral parts of the
    // distances shared between the directions.
    // Error of the approximate total against the reference's correctly rounded one: the total is the sum of 80
    // weighted ratios minus 80 (8 directions x 4 steps, step k counted 5 - k times), every ratio
    // (dif1 + dif2_prev) / dif2 >= 1 by the triangle inequality.  One approximate ratio differs from the real
    // value by at most 4.8e-7 relative (dif1: half the 6e-8 of its radicand + 1.2e-7 of v_sqrt; dif2_prev: 6e-8
    // + 1.5e-7 of its v_rsq + 6e-8 of the product; the sum 6e-8; the v_rsq of the divisor 1.5e-7; the final fma
    // rounds relative to the small score), the reference's by at most 3.0e-7 (three roundings in each radicand,
    // two correctly rounded roots, one sum, one division), so the two totals differ by at most
    // 7.8e-7 x (80 + total).  A wrong decision needs the totals on opposite sides of the threshold 0.0735 with the
    // approximate one more than the band away from it, i.e. a difference above 1e-4, which the bound allows only
    // for totals above 45, where both lie far above the threshold.  Inside the band of 1e-4 (about one pixel in a
    // thousand) the reference's evaluation is repeated.  The additions of the 64 terms themselves round at
    // 2^-24 of a total of ~0.07: 3e-7 in all.
    // tests/test_gpu_predicates.py tunes surfaces until their totals sit within 2e-4 of the threshold on both sides.
    if (!scores && !(dc > 0.001f)) {                          // no depth: the mask is 0 whatever the score
        mask[i] = 0;
        return;
    }
    // A wave with a single undecided lane would spend ten times the approximate evaluation on it, and about
    // one wave in twenty has one: undecided pixels go to a per-frame list (todo[frame] entries of todo_list[frame][todo_cap],
    // pixel indices) that curvature_exact_kernel works through with full waves; only when the list is full (or
    // absent) the exact evaluation runs here.
    auto at = [&](int oy, int ox) { return tile[ty + R + oy][tx + R + ox]; };
    float total;
    bool decided = false;
    if (!scores) {
        total = curvature_total<true>(at, dc, dfx, dfy);
        decided = fabsf(total - threshold) > 1e-4f;           // false for NaN: then the exact evaluation decides
        if (!decided && todo_cap > 0) {
            const int slot = atomicAdd(todo + blockIdx.z, 1);
            if (slot < todo_cap) {
                todo_list[(size_t)blockIdx.z * todo_cap + slot] = r * W + c;
                return;
            }
        }
    }
    if (!decided) total = curvature_total<false>(at, dc, dfx, dfy);
    if (scores) scores[i] = total;
    const uint8_t pix = total < threshold ? 1 : 0;
    mask[i] = dc > 0.001f ? pix : 0;
}

// The reference's correctly rounded evaluation for the pixels curvature_mask_kernel left undecided, depths read from
// global memory (the entries are scattered over the frame).  Eight lanes per list entry, one per direction: the
// evaluation is a chain of correctly rounded roots and divisions, and with one lane per entry a frame's few hundred
// entries kept a handful of waves busy for the length of eight such chains.  The first lane of an entry then adds the
// 32 running sums in the reference's order (direction by direction), so the total is the same number.
constexpr int kCurvExactBlocks = 64;      // blocks per frame; they stride over the list
__global__ __launch_bounds__(256) void curvature_exact_kernel(const float *__restrict__ D, const float *__restrict__ ax,
                                                              const float *__restrict__ ay, float threshold,
                                                              uint8_t *__restrict__ mask, const int32_t *__restrict__ todo,
                                                              const int32_t *__restrict__ todo_list, int todo_cap, int H, int W)
{
    const int32_t *list = todo_list + (size_t)blockIdx.y * todo_cap;
    const int count = min(todo[blockIdx.y], todo_cap);
    const float *F = D + (size_t)blockIdx.y * H * W;
    const int d = threadIdx.x & 7, lane = threadIdx.x & 63;
    const int d0 = d == 0 || d == 1 || d == 7 ? 1 : (d >= 3 && d <= 5 ? -1 : 0);     // [1,0],[1,1],[0,1],[-1,1],[-1,0],[-1,-1],[0,-1],[1,-1]
    const int d1 = d >= 1 && d <= 3 ? 1 : (d >= 5 ? -1 : 0);
    for (int t = blockIdx.x * 32 + (threadIdx.x >> 3); t < count; t += gridDim.x * 32) {    // the same t in all eight lanes of an entry
        const int p = list[t], r = p / W, c = p - r * W;
        const float dc = F[p];
        const float dfx = (ax[c] * dc) * 0.5f;
        const float dfy = (ay[r] * dc) * 0.5f;
        auto at = [&](int oy, int ox) { return F[p + oy * W + ox]; };   // the centre is at least 4 pixels inside the frame
        float curs[4];
        curvature_direction_exact(at, d0, d1, dc, dfx, dfy, curs);
        float total = 0.0f;
#pragma unroll
        for (int dd = 0; dd < 8; ++dd)
#pragma unroll
            for (int k = 0; k < 4; ++k) total = total + __shfl(curs[k], (lane & ~7) + dd);
        if (d == 0) mask[(size_t)blockIdx.y * H * W + p] = (total < threshold && dc > 0.001f) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------
// 5x5 (>11) then 3x3 (>5) majority filters (find_surfaces.py:265-266), depth
// edges (:269-278) and their product (:279), fused: the 5x5 result is kept in
// LDS for the tile plus a one-pixel ring (zero outside the frame, as the second
// extract_patches pads it).
// ---------------------------------------------------------------------------
constexpr int MTH = 32;   // tile height of the majority / edge kernel: 256 lanes x 4 rows (fewer, fatter blocks:
                          // the kernel is a chain of load and barrier latencies, not bandwidth)

__global__ __launch_bounds__(NT) void majority_edges_kernel(const uint8_t *__restrict__ curv, const float *__restrict__ D,
                                                            const float *__restrict__ ax, const float *__restrict__ ay,
                                                            float alpha, uint8_t *__restrict__ mask,
                                                            int32_t *__restrict__ edges, int32_t *__restrict__ clear, int H, int W)
{
    constexpr int R = 3;
    // the curvature pass's list counter of this frame has been consumed by now: zero it for the next call
    if (clear && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) clear[blockIdx.z] = 0;
    __shared__ float tm[MTH + 2 * R][TW + 2 * R + 1];
    __shared__ float td[MTH + 2][TW + 2 + 1];
    __shared__ int m5[MTH + 2][TW + 2 + 1];
    const size_t fo = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * MTH;
    for (int i = threadIdx.x; i < (MTH + 2 * R) * (TW + 2 * R); i += NT) {
        const int ly = i / (TW + 2 * R), lx = i - ly * (TW + 2 * R);
        const int y = y0 + ly - R, x = x0 + lx - R;
        tm[ly][lx] = (y >= 0 && y < H && x >= 0 && x < W) ? (float)curv[fo + (size_t)y * W + x] : 0.0f;
    }
    for (int i = threadIdx.x; i < (MTH + 2) * (TW + 2); i += NT) {
        const int ly = i / (TW + 2), lx = i - ly * (TW + 2);
        const int y = y0 + ly - 1, x = x0 + lx - 1;
        td[ly][lx] = (y >= 0 && y < H && x >= 0 && x < W) ? D[fo + (size_t)y * W + x] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (MTH + 2) * (TW + 2); i += NT) {
        int ly = i / (TW + 2), lx = i - ly * (TW + 2);
        int y = y0 + ly - 1, x = x0 + lx - 1;
        int v = 0;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            float s = 0.0f;   // small integers: exact in fp32
#pragma unroll
            for (int wy = 0; wy < 5; ++wy)
#pragma unroll
                for (int wx = 0; wx < 5; ++wx) s += tm[ly + wy][lx + wx];
            v = s > 11.0f ? 1 : 0;
        }
        m5[ly][lx] = v;
    }
    __syncthreads();
    const int tx = threadIdx.x % TW, c = x0 + tx;
    if (c >= W) return;
    for (int ty = threadIdx.x / TW; ty < MTH; ty += NT / TW) {
        const int r = y0 + ty;
        if (r >= H) break;
        int s3 = 0;
#pragma unroll
        for (int wy = 0; wy < 3; ++wy)
#pragma unroll
            for (int wx = 0; wx < 3; ++wx) s3 += m5[ty + wy][tx + wx];
        const int pix = s3 > 5 ? 1 : 0;
        int e = 0;
        if (r >= 1 && r < H - 1 && c >= 1 && c < W - 1) {
            const float dc = td[ty + 1][tx + 1];
            const float fx = ((ax[c] * dc) * 0.5f) * alpha;
            const float fy = ((ay[r] * dc) * 0.5f) * alpha;
            e = 1;
            if (fabsf(td[ty + 1][tx + 2] - dc) > fx) e = 0;
            if (fabsf(td[ty + 1][tx] - dc) > fx) e = 0;
            if (fabsf(td[ty][tx + 1] - dc) > fy) e = 0;
            if (fabsf(td[ty + 2][tx + 1] - dc) > fy) e = 0;
        }
        const size_t i = fo + (size_t)r * W + c;
        edges[i] = e;
        mask[i] = (uint8_t)(e * pix);
    }
}

inline dim3 tile_grid(int n, int H, int W) { return dim3((W + TW - 1) / TW, (H + TH - 1) / TH, n); }

}  // namespace

void launch_smooth_depth_u16(const uint16_t *depth, float *out, int n, int H, int W, hipStream_t st, const uint8_t *rgb, void *featA,
                             float colour_scale)
{
    hipLaunchKernelGGL(smooth_depth_kernel<uint16_t>, dim3((W + TW - 1) / TW, (H + STH - 1) / STH, n), dim3(NT), 0, st, depth, out, H, W,
                       rgb, rgb ? reinterpret_cast<float4 *>(featA) : nullptr, colour_scale);
}
void launch_smooth_depth_f32(const float *depth, float *out, int n, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(smooth_depth_kernel<float>, tile_grid(n, H, W), dim3(NT), 0, st, depth, out, H, W, nullptr, nullptr, 0.0f);
}
void launch_normals(const float *D, const float *tanx, const float *tany, float *points, float *normals,
                    float *raw_angles, int n, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(normals_kernel, tile_grid(n, H, W), dim3(NT), 0, st, D, tanx, tany, points, normals, raw_angles, H, W);
}
void launch_angle_filter(const float *raw_angles, const float *D, float *angles, void *feat, float feat_scale, int n, int H, int W,
                         hipStream_t st)
{
    hipLaunchKernelGGL(angle_filter_kernel, dim3((W + TW - 1) / TW, (H + ATH - 1) / ATH, n), dim3(NT), 0, st, raw_angles, D, angles,
                       reinterpret_cast<float2 *>(feat), feat_scale, H, W);
}
bool launch_curvature_mask(const float *D, const float *ax, const float *ay, float threshold, uint8_t *mask,
                           float *scores, int32_t *todo, int todo_cap, int todo_frames, bool todo_zeroed, int n, int H, int W, hipStream_t st)
{
    if (scores || !todo || todo_cap < 256 || n > todo_frames) {
        hipLaunchKernelGGL(curvature_mask_kernel, tile_grid(n, H, W), dim3(NT), 0, st, D, ax, ay, threshold, mask, scores, nullptr, nullptr,
                           0, H, W);
        return false;
    }
    int32_t *lists = todo + todo_frames;                // counters of all frames first, at a fixed place, then the lists
    if (!todo_zeroed) (void)hipMemsetAsync(todo, 0, (size_t)todo_frames * sizeof(int32_t), st);
    hipLaunchKernelGGL(curvature_mask_kernel, tile_grid(n, H, W), dim3(NT), 0, st, D, ax, ay, threshold, mask, scores, todo, lists, todo_cap,
                       H, W);
    hipLaunchKernelGGL(curvature_exact_kernel, dim3(std::min(kCurvExactBlocks, (todo_cap + 31) / 32), n), dim3(256), 0, st, D, ax, ay, threshold, mask, todo, lists,
                       todo_cap, H, W);
    return true;
}
size_t curvature_todo_ints(int frames, int todo_cap) { return (size_t)frames * (1 + (size_t)todo_cap); }
void launch_majority_edges(const uint8_t *curv, const float *D, const float *ax, const float *ay, float alpha,
                           uint8_t *mask, int32_t *edges, int32_t *clear, int n, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(majority_edges_kernel, dim3((W + TW - 1) / TW, (H + MTH - 1) / MTH, n), dim3(NT), 0, st, curv, D, ax, ay, alpha, mask, edges,
                       clear, H, W);
}

}  // namespace fs
