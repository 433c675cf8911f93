// Pairwise reductions that follow the per-surface statistics (gfx950):
//   chi-squared distances between the pooled histograms of all pairs of surfaces
//       (chi_squared_distances_model_2D / _1D, image_processing_models_GPU.py:189-212)
//   label-pair overlap counts of two label images
//       (the pixel loop of evaluate_models.extract_IoU_information, evaluate_models.py:28-35)
#include "fs_common.h"

namespace fs {

namespace {

// AveragePooling2D (VALID) of hist[A][B][S] (float64 in, cast to float32 as Keras does) into pooled[S][PA][PB]
__global__ __launch_bounds__(256) void chi_pool_kernel(const double *__restrict__ hist, float *__restrict__ pooled, int A, int B, int S,
                                                       int ph, int pw, int sh, int sw, int PA, int PB)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)S * PA * PB;
    if (i >= total) return;
    const int s = (int)(i % S);                       // s fastest: neighbouring threads read neighbouring doubles
    const int px = (int)((i / S) % PB), py = (int)(i / ((size_t)S * PB));
    float acc = 0.0f;
    for (int y = 0; y < ph; ++y)
        for (int x = 0; x < pw; ++x) acc = acc + (float)hist[((size_t)(py * sh + y) * B + (px * sw + x)) * S + s];
    pooled[((size_t)s * PA + py) * PB + px] = acc / (float)(ph * pw);
}

// distance[i][j] = sum over bins of divide_no_nan((a - b)^2, a + b); one block per pair i <= j
__global__ __launch_bounds__(256) void chi_pair_kernel(const float *__restrict__ pooled, float *__restrict__ out, int S, int K)
{
    __shared__ float part[256];
    const int i = blockIdx.x, j = blockIdx.y;
    if (j < i) return;
    const float *a = pooled + (size_t)i * K, *b = pooled + (size_t)j * K;
    float acc = 0.0f;
    if (i != j)
        for (int k = threadIdx.x; k < K; k += 256) {
            const float x = a[k], y = b[k], d = x - y, sum = x + y;
            acc = acc + (sum == 0.0f ? 0.0f : (d * d) / sum);
        }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) part[threadIdx.x] = part[threadIdx.x] + part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[(size_t)i * S + j] = part[0];
        out[(size_t)j * S + i] = part[0];
    }
}

// overlap[o][a] += 1 for every pixel; per-block privatisation in LDS when the matrix fits
__global__ __launch_bounds__(256) void overlap_kernel(const int64_t *__restrict__ annotation, const int64_t *__restrict__ detected,
                                                      unsigned *__restrict__ overlap, int *__restrict__ err, int N, int nd, int na,
                                                      int use_lds)
{
    extern __shared__ unsigned sh[];
    const int cells = nd * na;
    if (use_lds) {
        for (int i = threadIdx.x; i < cells; i += 256) sh[i] = 0u;
        __syncthreads();
    }
    for (int k = 0; k < 4; ++k) {
        const int p = blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (p >= N) continue;
        const int64_t a = annotation[p], o = detected[p];
        if (a < 0 || a >= na || o < 0 || o >= nd) { *err = 1; continue; }
        if (use_lds) atomicAdd(&sh[(int)o * na + (int)a], 1u);
        else atomicAdd(overlap + (size_t)o * na + a, 1u);
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < cells; i += 256)
            if (sh[i]) atomicAdd(overlap + i, sh[i]);
    }
}

// calculate_nearest_points_calc (process_surfaces.py:638-646): for every target of group g the point of
// group g with the smallest squared distance (float32 arithmetic, first minimum).  One block per target.
__global__ __launch_bounds__(256) void nearest_points_kernel(const float2 *__restrict__ points, const int32_t *__restrict__ point_offsets,
                                                             const float2 *__restrict__ targets, const int32_t *__restrict__ target_group,
                                                             int32_t *__restrict__ out)
{
    __shared__ float sd[256];
    __shared__ int si[256];
    const int t = blockIdx.x, g = target_group[t];
    const int a = point_offsets[g], b = point_offsets[g + 1];
    const float2 tg = targets[t];
    float best = INFINITY;
    int besti = 0x7fffffff;
    for (int k = a + (int)threadIdx.x; k < b; k += 256) {      // ascending per thread: the first minimum stays
        const float2 p = points[k];
        const float dx = p.x - tg.x, dy = p.y - tg.y;
        const float d = dx * dx + dy * dy;
        if (d < best) { best = d; besti = k; }
    }
    sd[threadIdx.x] = best;
    si[threadIdx.x] = besti;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float d2 = sd[threadIdx.x + off];
            const int i2 = si[threadIdx.x + off];
            if (d2 < sd[threadIdx.x] || (d2 == sd[threadIdx.x] && i2 < si[threadIdx.x])) { sd[threadIdx.x] = d2; si[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int k = si[0];
        out[2 * t] = k == 0x7fffffff ? -1 : (int32_t)points[k].x;
        out[2 * t + 1] = k == 0x7fffffff ? -1 : (int32_t)points[k].y;
    }
}

}  // namespace

void launch_chi_squared(const double *hist, float *pooled, float *out, int A, int B, int S, int ph, int pw, int sh, int sw,
                        hipStream_t st)
{
    const int PA = (A - ph) / sh + 1, PB = (B - pw) / sw + 1;
    const size_t total = (size_t)S * PA * PB;
    hipLaunchKernelGGL(chi_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, hist, pooled, A, B, S, ph, pw, sh, sw,
                       PA, PB);
    hipLaunchKernelGGL(chi_pair_kernel, dim3(S, S), dim3(256), 0, st, pooled, out, S, PA * PB);
}

void launch_overlap_counts(const int64_t *annotation, const int64_t *detected, unsigned *overlap, int *err, int N, int nd, int na,
                           hipStream_t st)
{
    const int use_lds = (size_t)nd * na <= 8192;
    hipLaunchKernelGGL(overlap_kernel, dim3((N + 1023) / 1024), dim3(256), use_lds ? (size_t)nd * na * 4 : 0, st, annotation, detected,
                       overlap, err, N, nd, na, use_lds);
}

void launch_nearest_points(const float *points, const int32_t *point_offsets, const float *targets, const int32_t *target_group,
                           int T, int32_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(nearest_points_kernel, dim3(T), dim3(256), 0, st, reinterpret_cast<const float2 *>(points), point_offsets,
                       reinterpret_cast<const float2 *>(targets), target_group, out);
}

}  // namespace fs
