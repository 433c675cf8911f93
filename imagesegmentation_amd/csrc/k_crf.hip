// Mean-field CRF pixel assignment with the trained Gaussian pixel-similarity
// function (gfx950).  Replaces conv_crf_Gauss (find_surfaces.py:291-367), the
// host glue around it (get_inputs / assemble_outputs / the per-pixel unary loop,
// find_surfaces.py:437-495) and the argmax + fix-ups of find_surfaces.py:520-532.
//
// Structure
//   crf_sim     the similarity sim(p, q) = w1 s1 + w2 s2 + w3 s3 + bias is
//               symmetric in (p, q) and does not change between the three
//               mean-field iterations, so it is evaluated once per frame for the
//               "forward" half of the window (dy > 0, or dy == 0 and dx > 0) and
//               kept as planes simH[offset][pixel]; the backward half is read
//               from the neighbour's forward entry.  Features are staged in LDS.
//   crf_init    initial probabilities from the patch labels.
//   crf_iter    one mean-field update: message sum over the (2r+1)^2-1 window
//               with the label distributions of the tile + halo staged in LDS,
//               Potts compatibility, softmax, the invalid-depth overrides and,
//               on the last iteration, the argmax.  One pixel per lane, all label
//               channels of the pixel in registers.
// Pixels outside the frame contribute nothing (their Q is 0), which is what the
// reference's zero padding of Q does; the reference's 4x4 tiling is not
// reproduced because a halo of r makes it numerically invisible.
#include "fs_common.h"

namespace fs {

namespace {

__device__ __forceinline__ float div_no_nan(float a, float b) { return b == 0.0f ? 0.0f : a / b; }

// index of a forward offset in the half-window enumeration
__host__ __device__ __forceinline__ int half_index(int dy, int dx, int r)
{
    return dy == 0 ? dx - 1 : r + (dy - 1) * (2 * r + 1) + (dx + r);
}

// ---------------------------------------------------------------------------
// similarity planes, fused-path features: raw depth (u16), raw RGB (u8), smoothed
// angles (f32 x2); position features are the pixel coordinates, so their terms
// depend on the offset only and come from pos_terms[offset][3].
// ---------------------------------------------------------------------------
constexpr int STW = 32, STH = 8, SNT = STW * STH;

__global__ __launch_bounds__(SNT) void crf_sim_kernel(const uint16_t *__restrict__ depth, const uint8_t *__restrict__ rgb,
                                                      const float *__restrict__ angles, CrfParams P,
                                                      const float *__restrict__ pos_terms, float *__restrict__ simH,
                                                      int H, int W, int r)
{
    extern __shared__ float lds[];
    const int LW = STW + 2 * r, LH = STH + r;
    const int plane = LW * LH;
    float *fd = lds, *fr = fd + plane, *fg = fr + plane, *fb = fg + plane, *fa0 = fb + plane, *fa1 = fa0 + plane;
    const int x0 = blockIdx.x * STW, y0 = blockIdx.y * STH;
    for (int i = threadIdx.x; i < plane; i += SNT) {
        const int ly = i / LW, lx = i - ly * LW;
        const int y = y0 + ly, x = x0 + lx - r;
        float d = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, a0 = 0.f, a1 = 0.f;
        if (y < H && x >= 0 && x < W) {
            const size_t j = (size_t)y * W + x;
            d = (float)depth[j];
            c0 = (float)rgb[3 * j]; c1 = (float)rgb[3 * j + 1]; c2 = (float)rgb[3 * j + 2];
            const float2 a = reinterpret_cast<const float2 *>(angles)[j];
            a0 = a.x; a1 = a.y;
        }
        fd[i] = d; fr[i] = c0; fg[i] = c1; fb[i] = c2; fa0[i] = a0; fa1[i] = a1;
    }
    __syncthreads();
    const int tx = threadIdx.x % STW, ty = threadIdx.x / STW;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= W || y >= H) return;
    const size_t N = (size_t)H * W, p = (size_t)y * W + x;
    const int ci = ty * LW + tx + r;
    const float dc = fd[ci], rc = fr[ci], gc = fg[ci], bc = fb[ci], a0c = fa0[ci], a1c = fa1[ci];
    const float nd = 6.0f, na = 2.0f, nc = (float)(1.0 / 50.0);   // similarity_normalizers, find_surfaces.py:20
    for (int dy = 0; dy <= r; ++dy) {
        for (int dx = (dy == 0 ? 1 : -r); dx <= r; ++dx) {
            const int oi = half_index(dy, dx, r);
            float sim = 0.0f;
            if (y + dy < H && x + dx >= 0 && x + dx < W) {
                const int ji = ci + dy * LW + dx;
                const float dj = fd[ji];
                const float d2 = div_no_nan(dj - dc, (dc + dj) / 2.0f) * nd;
                const float dd = (d2 * d2);
                const float e3 = ((fr[ji] - rc) / 1.0f) * nc, e4 = ((fg[ji] - gc) / 1.0f) * nc, e5 = ((fb[ji] - bc) / 1.0f) * nc;
                const float g3 = ((fa0[ji] - a0c) / 1.0f) * na, g4 = ((fa1[ji] - a1c) / 1.0f) * na;
                const float *pt = pos_terms + 3 * oi;
                const float s1 = expf(-(pt[0] + dd * P.t12));
                const float s2 = expf(-((((pt[1] + dd * P.t22) + (e3 * e3) * P.t23) + (e4 * e4) * P.t23) + (e5 * e5) * P.t23));
                const float s3 = expf(-(((pt[2] + dd * P.t32) + (g3 * g3) * P.t33) + (g4 * g4) * P.t33));
                sim = ((P.w1 * s1 + P.w2 * s2) + P.w3 * s3) + P.bias;
            }
            simH[(size_t)oi * N + p] = sim;
        }
    }
}

// ---------------------------------------------------------------------------
// similarity planes from explicit feature tensors (operator-level API): f1
// [Hi,Wi,3] = (y,x,depth), f2 [Hi,Wi,6], f3 [Hi,Wi,5], each with its own copies of
// the position and depth channels exactly as conv_crf_Gauss receives them.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crf_sim_explicit_kernel(const float *__restrict__ f1, const float *__restrict__ f2,
                                                               const float *__restrict__ f3, CrfParams P,
                                                               float *__restrict__ simH, int Hi, int Wi, int r)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= Wi || y >= Hi) return;
    const size_t N = (size_t)Hi * Wi, p = (size_t)y * Wi + x;
    const float nd = 6.0f, na = 2.0f, nc = (float)(1.0 / 50.0), np_ = (float)(1.0 / 40.0);
    const float *c1 = f1 + 3 * p, *c2 = f2 + 6 * p, *c3 = f3 + 5 * p;
    for (int dy = 0; dy <= r; ++dy)
        for (int dx = (dy == 0 ? 1 : -r); dx <= r; ++dx) {
            const int oi = half_index(dy, dx, r);
            float sim = 0.0f;
            if (y + dy < Hi && x + dx >= 0 && x + dx < Wi) {
                const size_t j = p + (size_t)dy * Wi + dx;
                const float *j1 = f1 + 3 * j, *j2 = f2 + 6 * j, *j3 = f3 + 5 * j;
                const float d0 = ((j1[0] - c1[0]) / 1.0f) * np_, d1 = ((j1[1] - c1[1]) / 1.0f) * np_;
                const float d2 = div_no_nan(j1[2] - c1[2], (c1[2] + j1[2]) / 2.0f) * nd;
                const float s1 = expf(-(((d0 * d0) * P.t11 + (d1 * d1) * P.t11) + (d2 * d2) * P.t12));
                const float e0 = ((j2[0] - c2[0]) / 1.0f) * np_, e1 = ((j2[1] - c2[1]) / 1.0f) * np_;
                const float e2 = div_no_nan(j2[2] - c2[2], (c2[2] + j2[2]) / 2.0f) * nd;
                const float e3 = ((j2[3] - c2[3]) / 1.0f) * nc, e4 = ((j2[4] - c2[4]) / 1.0f) * nc, e5 = ((j2[5] - c2[5]) / 1.0f) * nc;
                const float s2 = expf(-((((((e0 * e0) * P.t21 + (e1 * e1) * P.t21) + (e2 * e2) * P.t22) + (e3 * e3) * P.t23) +
                                         (e4 * e4) * P.t23) + (e5 * e5) * P.t23));
                const float g0 = ((j3[0] - c3[0]) / 1.0f) * np_, g1 = ((j3[1] - c3[1]) / 1.0f) * np_;
                const float g2 = div_no_nan(j3[2] - c3[2], (c3[2] + j3[2]) / 2.0f) * nd;
                const float g3 = ((j3[3] - c3[3]) / 1.0f) * na, g4 = ((j3[4] - c3[4]) / 1.0f) * na;
                const float s3 = expf(-(((((g0 * g0) * P.t31 + (g1 * g1) * P.t31) + (g2 * g2) * P.t32) + (g3 * g3) * P.t33) +
                                        (g4 * g4) * P.t33));
                sim = ((P.w1 * s1 + P.w2 * s2) + P.w3 * s3) + P.bias;
            }
            simH[(size_t)oi * N + p] = sim;
        }
}

// ---------------------------------------------------------------------------
// initial probabilities, get_unary_potentials_and_initial_probabilities
// (find_surfaces.py:456-477): double arithmetic, cast to float like Keras does.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crf_init_kernel(const int32_t *__restrict__ patches, float *__restrict__ Q, int N, int S)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    const int ns = S - 1;
    const int lab = patches[p];
    float *q = Q + (size_t)p * S;
    if (lab == 0) {
        const float pv = (float)(1.0 / (double)ns);
        q[0] = 0.0f;
        for (int l = 1; l < S; ++l) q[l] = pv;
    } else {
        const float other = (float)(1.0 / (double)(ns - 1) * 0.1);
        for (int l = 0; l < S; ++l) q[l] = other;
        q[lab] = (float)0.9;
        q[0] = 0.0f;
    }
}

// ---------------------------------------------------------------------------
// one mean-field update
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ int padded_labels(int S)
{
    int sp = (S + 3) & ~3;          // float4 granules ...
    if (((sp >> 2) & 1) == 0) sp += 4;   // ... an odd number of them per pixel spreads lanes over the LDS banks
    return sp;
}

template <int LC, int TWc, int THc>
__global__ __launch_bounds__(TWc *THc) void crf_iter_kernel(const float *__restrict__ simH, const float *__restrict__ Qin,
                                                            float *__restrict__ Qout, float *__restrict__ potentials,
                                                            const float *__restrict__ unary_explicit,
                                                            const int32_t *__restrict__ patches,
                                                            const uint16_t *__restrict__ depth_raw,
                                                            int64_t *__restrict__ surfaces, CrfFrame f, float weight, int mode)
{
    constexpr int NTc = TWc * THc;
    extern __shared__ float4 lds4[];
    float *qt = reinterpret_cast<float *>(lds4);
    const int r = f.r, S = f.S, SP = padded_labels(S);
    const int LW = TWc + 2 * r, LH = THc + 2 * r;
    const int wx0 = blockIdx.x * TWc, wy0 = blockIdx.y * THc;   // tile origin in the output window
    const int ix0 = f.ox + wx0, iy0 = f.oy + wy0;               // ... and in the image
    // stage Q of tile + halo: [LH][LW][SP], zero outside the image and in the label padding
    for (int i = threadIdx.x; i < LH * LW * SP; i += NTc) {
        const int px = i / SP, l = i - px * SP;
        const int ly = px / LW, lx = px - ly * LW;
        const int gy = iy0 - r + ly, gx = ix0 - r + lx;
        float v = 0.0f;
        if (l < S && gy >= 0 && gy < f.Hi && gx >= 0 && gx < f.Wi) v = Qin[((size_t)gy * f.Wi + gx) * S + l];
        qt[i] = v;
    }
    __syncthreads();
    const int tx = threadIdx.x % TWc, ty = threadIdx.x / TWc;
    const int wx = wx0 + tx, wy = wy0 + ty;
    if (wx >= f.ow || wy >= f.oh) return;
    const int gx = ix0 + tx, gy = iy0 + ty;
    const size_t N = (size_t)f.Hi * f.Wi, p = (size_t)gy * f.Wi + gx;

    float acc[LC];
#pragma unroll
    for (int l = 0; l < LC; ++l) acc[l] = 0.0f;

    const float *qc = qt + ((ty + r) * LW + (tx + r)) * SP;
    for (int dy = 0; dy <= r; ++dy) {
        const bool fy = gy + dy < f.Hi, by = gy - dy >= 0;
        for (int dx = (dy == 0 ? 1 : -r); dx <= r; ++dx) {
            const int oi = half_index(dy, dx, r);
            const bool fin = fy && gx + dx >= 0 && gx + dx < f.Wi;
            const bool bin = by && gx - dx >= 0 && gx - dx < f.Wi;
            const int doff = dy * f.Wi + dx;
            const float sf = fin ? simH[(size_t)oi * N + p] : 0.0f;
            const float sb = bin ? simH[(size_t)oi * N + p - doff] : 0.0f;
            const float4 *qf = reinterpret_cast<const float4 *>(qc + (dy * LW + dx) * SP);
            const float4 *qb = reinterpret_cast<const float4 *>(qc - (dy * LW + dx) * SP);
#pragma unroll
            for (int l4 = 0; l4 < LC / 4; ++l4) {
                if (l4 * 4 < SP) {
                    const float4 a = qf[l4], b = qb[l4];
                    acc[4 * l4 + 0] = fmaf(sf, a.x, acc[4 * l4 + 0]);
                    acc[4 * l4 + 1] = fmaf(sf, a.y, acc[4 * l4 + 1]);
                    acc[4 * l4 + 2] = fmaf(sf, a.z, acc[4 * l4 + 2]);
                    acc[4 * l4 + 3] = fmaf(sf, a.w, acc[4 * l4 + 3]);
                    acc[4 * l4 + 0] = fmaf(sb, b.x, acc[4 * l4 + 0]);
                    acc[4 * l4 + 1] = fmaf(sb, b.y, acc[4 * l4 + 1]);
                    acc[4 * l4 + 2] = fmaf(sb, b.z, acc[4 * l4 + 2]);
                    acc[4 * l4 + 3] = fmaf(sb, b.w, acc[4 * l4 + 3]);
                }
            }
        }
    }

    // message of the constant "unassigned" evidence: 0.00005 * onehot(S-1)
    // (find_surfaces.py:356-357, 451-452)
    // compensated total so that compatibility[l] = total - msg[l] keeps full relative accuracy
    float tot = 0.0f, comp_lo = 0.0f;
#pragma unroll
    for (int l = 0; l < LC; ++l) {
        if (l < S) {
            if (l == S - 1) acc[l] = acc[l] + 0.00005f;
            const float t = tot + acc[l];
            const float bp = t - tot;
            comp_lo += (tot - (t - bp)) + (acc[l] - bp);
            tot = t;
        }
    }
    const size_t wi = (size_t)wy * f.ow + wx;   // index in the output window
    const int lab = patches ? patches[p] : 0;
    float mx = -INFINITY;
#pragma unroll
    for (int l = 0; l < LC; ++l) {
        if (l < S) {
            const float comp = (tot - acc[l]) + comp_lo;   // Potts: sum of the other labels' messages
            float un;
            if (unary_explicit) un = unary_explicit[wi * S + l];
            else un = l == 0 ? 10.0f : ((lab != 0 && l == lab) ? 0.0f : 1.0f);
            const float add = un + weight * comp;
            if (potentials) potentials[wi * S + l] = add;
            acc[l] = -add;
            mx = fmaxf(mx, acc[l]);
        }
    }
    float den = 0.0f;
#pragma unroll
    for (int l = 0; l < LC; ++l) {
        if (l < S) {
            acc[l] = expf(acc[l] - mx);
            den = den + acc[l];
        }
    }
    const bool last = (mode & CRF_LAST) != 0;
    const bool invalid = depth_raw && depth_raw[p] == 0;   // Q[depth == 0] = prob | 0, find_surfaces.py:520-530
    const float pv = (float)(1.0 / (double)(S - 1));
    int best = 0;
    float bestv = -1.0f;
#pragma unroll
    for (int l = 0; l < LC; ++l) {
        if (l < S) {
            float q = acc[l] / den;
            if (invalid) q = last ? 0.0f : (l == 0 ? 0.0f : pv);
            if (q > bestv) { bestv = q; best = l; }
            if (Qout) Qout[wi * S + l] = q;
        }
    }
    if (surfaces) surfaces[wi] = (best == S - 1) ? 0 : best;   // find_surfaces.py:531-532
}

template <int LC, int TWc, int THc>
hipError_t launch_iter_t(const float *simH, const float *Qin, float *Qout, float *potentials, const float *unary_explicit,
                         const int32_t *patches, const uint16_t *depth_raw, int64_t *surfaces, const CrfFrame &f,
                         float weight, int mode, size_t lds_bytes, hipStream_t st)
{
    static bool attr_set = false;
    auto kern = crf_iter_kernel<LC, TWc, THc>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((f.ow + TWc - 1) / TWc, (f.oh + THc - 1) / THc);
    hipLaunchKernelGGL(kern, grid, dim3(TWc * THc), lds_bytes, st, simH, Qin, Qout, potentials, unary_explicit, patches,
                       depth_raw, surfaces, f, weight, mode);
    return hipGetLastError();
}

template <int TWc, int THc>
hipError_t launch_iter_lc(int LC, const float *simH, const float *Qin, float *Qout, float *potentials,
                          const float *unary_explicit, const int32_t *patches, const uint16_t *depth_raw, int64_t *surfaces,
                          const CrfFrame &f, float weight, int mode, size_t lds_bytes, hipStream_t st)
{
#define FS_CASE(L)                                                                                                   \
    case L:                                                                                                          \
        return launch_iter_t<L, TWc, THc>(simH, Qin, Qout, potentials, unary_explicit, patches, depth_raw, surfaces, \
                                          f, weight, mode, lds_bytes, st);
    switch (LC) {
        FS_CASE(8)
        FS_CASE(16)
        FS_CASE(24)
        FS_CASE(32)
        FS_CASE(48)
        FS_CASE(64)
        FS_CASE(96)
        FS_CASE(128)
    }
#undef FS_CASE
    return hipErrorInvalidValue;
}

}  // namespace

size_t crf_half_offsets(int r) { return (size_t)r + (size_t)r * (2 * r + 1); }

void launch_crf_sim(const uint16_t *depth, const uint8_t *rgb, const float *angles, const CrfParams &p,
                    const float *pos_terms, float *simH, int H, int W, int r, hipStream_t st)
{
    const size_t lds = (size_t)(STW + 2 * r) * (STH + r) * 6 * sizeof(float);
    dim3 grid((W + STW - 1) / STW, (H + STH - 1) / STH);
    hipLaunchKernelGGL(crf_sim_kernel, grid, dim3(SNT), lds, st, depth, rgb, angles, p, pos_terms, simH, H, W, r);
}

void launch_crf_sim_explicit(const float *f1, const float *f2, const float *f3, const CrfParams &p, float *simH, int Hi,
                             int Wi, int r, hipStream_t st)
{
    dim3 grid((Wi + 31) / 32, (Hi + 7) / 8);
    hipLaunchKernelGGL(crf_sim_explicit_kernel, grid, dim3(256), 0, st, f1, f2, f3, p, simH, Hi, Wi, r);
}

void launch_crf_init(const int32_t *patches, float *Q, int H, int W, int S, hipStream_t st)
{
    const int N = H * W;
    hipLaunchKernelGGL(crf_init_kernel, dim3((N + 255) / 256), dim3(256), 0, st, patches, Q, N, S);
}

hipError_t launch_crf_iter(const float *simH, const float *Qin, float *Qout, float *potentials,
                           const float *unary_explicit, const int32_t *patches, const uint16_t *depth_raw,
                           int64_t *surfaces, const CrfFrame &f, float weight, int mode, hipStream_t st)
{
    if (f.S < 2 || f.S > kMaxLabels || f.r < 1 || f.r > kMaxRadius) return hipErrorInvalidValue;
    static const int lcs[] = {8, 16, 24, 32, 48, 64, 96, 128};
    int LC = 128;
    for (int c : lcs)
        if (c >= f.S) { LC = c; break; }
    const int SP = padded_labels(f.S);
    const size_t budget = 160 * 1024;
    auto bytes = [&](int tw, int th) { return (size_t)(tw + 2 * f.r) * (th + 2 * f.r) * SP * sizeof(float); };
    // largest tile whose Q stage fits twice into a CU's LDS, else the largest that fits once
    if (bytes(16, 8) * 2 <= budget || (bytes(16, 8) <= budget && bytes(8, 8) * 2 > budget))
        return launch_iter_lc<16, 8>(LC, simH, Qin, Qout, potentials, unary_explicit, patches, depth_raw, surfaces, f, weight, mode, bytes(16, 8), st);
    if (bytes(8, 8) <= budget)
        return launch_iter_lc<8, 8>(LC, simH, Qin, Qout, potentials, unary_explicit, patches, depth_raw, surfaces, f, weight, mode, bytes(8, 8), st);
    return hipErrorInvalidValue;   // label count x window too large for the LDS-staged kernel
}

}  // namespace fs
