// Mean-field CRF pixel assignment with the trained Gaussian pixel-similarity
// function (gfx950).  Replaces conv_crf_Gauss (find_surfaces.py:291-367), the
// host glue around it (get_inputs / assemble_outputs / the per-pixel unary loop,
// find_surfaces.py:437-495) and the argmax + fix-ups of find_surfaces.py:520-532.
//
// Two update kernels share the pieces below:
//   crf_iter_fused_kernel   the production path (fs_find_surfaces_batch, fs_apply_crf): features of the
//                           frame, similarity evaluated in the kernel; described at its definition.
//   crf_iter_kernel         the operator-level entry (fs_crf_step: conv_crf_Gauss.predict on arbitrary
//                           feature tensors): similarity precomputed into planes simH.
//
// Data layout
//   Q          label planes Q[l][pixel] (the reference is pixel-major): a lane per
//              pixel then reads and writes 256 contiguous bytes per wave for every
//              label it touches.
//   simH       (crf_iter_kernel only) sim(p, q) = w1 s1 + w2 s2 + w3 s3 + bias is symmetric in
//              (p, q) and identical in all mean-field iterations, so it is evaluated once per
//              frame for the "forward" half of the window (dy > 0, or dy == 0 and
//              dx > 0) and kept as planes simH[offset][pixel]; the backward half is
//              the neighbour's forward entry.  The buffer carries r*W + r zeroed
//              floats of slack at both ends so that shifted reads never leave it.
//
// Label compaction (the step that makes the update cheap)
//   A label l whose patch has no pixel within r * iterations of a tile has, at
//   every pixel the tile's update reads or writes, the same unary (1) and the same
//   probability as every other such label: they are exchangeable.  crf_tile_active
//   lists, per tile, the patch labels present in the tile grown by that distance;
//   the update then carries only {label 0, the listed labels, one representative
//   of all other labels with its multiplicity, the "unassigned" label S-1} --
//   typically 4-8 channels instead of S = 10..65 -- and expands the result back to
//   S planes.  In exact arithmetic this is the same update; in fp32 it differs
//   from a dense evaluation by summation order only.
//
// crf_iter: one block = one 64 x 16 tile of one frame (blockIdx.z = frame of the
// launch), 16/PY waves, each lane PY vertically adjacent pixels of one column.
// Q of the tile + halo is staged in LDS as float4 channel quads, 8 channels per
// pass, so consecutive lanes read consecutive 16-byte slots (no bank conflicts).
// The window is walked by forward offset; at each offset a lane uses plane[p] for
// the neighbour p+o and plane[p-o] for the neighbour p-o.  The second read is the
// first read of the lane that owns p-o at the same step, so within a tile both
// uses of a similarity value reach L2/HBM once.
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "fs_common.h"
// Timing experiments only (never defined in the product build): -DFS_SKIPMAIN=1 drops the window walk of the
// fused update, -DFS_SKIPQ=1 the staging of Q, -DFS_SKIPFIN=1 the epilogue.  Builds with parts switched off,
// run under the SQ counters (tools/pmc_variant.sh), tell how the vector instructions split over the phases.
#ifndef FS_SKIPQ
#define FS_SKIPQ 0
#endif
#ifndef FS_SKIPFIN
#define FS_SKIPFIN 0
#endif
#ifndef FS_SKIPMAIN
#define FS_SKIPMAIN 0
#endif

namespace fs {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));      // packed pair: arithmetic lowers to v_pk_*_f32
typedef __attribute__((address_space(1))) float gfloat;   // pointer known to be global: saddr + 32-bit offset loads

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
// exp() is evaluated as exp2(x * log2 e) on the hardware unit: its absolute error
// on a value in (0, 1] stays below 3e-8, the size of one fp32 rounding.
// blockIdx.z = frame; sim planes of frame z start at simH + z * sim_stride.
// ---------------------------------------------------------------------------
constexpr int STW = 64, STH = 4, SNT = STW * STH;

__global__ __launch_bounds__(SNT) void crf_sim_kernel(const uint16_t *__restrict__ depth, const uint8_t *__restrict__ rgb,
                                                      const float *__restrict__ angles, CrfParams P,
                                                      const float *__restrict__ pos_terms, float *__restrict__ simH,
                                                      size_t sim_stride, int H, int W, int r)
{
    extern __shared__ float lds[];
    const size_t N = (size_t)H * W;
    depth += blockIdx.z * N;
    rgb += blockIdx.z * N * 3;
    angles += blockIdx.z * N * 2;
    simH += blockIdx.z * sim_stride;
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
    const size_t p = (size_t)y * W + x;
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
                const float e3 = (fr[ji] - rc) * nc, e4 = (fg[ji] - gc) * nc, e5 = (fb[ji] - bc) * nc;
                const float g3 = (fa0[ji] - a0c) * na, g4 = (fa1[ji] - a1c) * na;
                const float *pt = pos_terms + 3 * oi;
                const float s1 = __expf(-(pt[0] + dd * P.t12));
                const float s2 = __expf(-((((pt[1] + dd * P.t22) + (e3 * e3) * P.t23) + (e4 * e4) * P.t23) + (e5 * e5) * P.t23));
                const float s3 = __expf(-(((pt[2] + dd * P.t32) + (g3 * g3) * P.t33) + (g4 * g4) * P.t33));
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
// conv_crf_LR.predict on explicit feature tensors (operator-level API, find_surfaces.py:371-434):
// one thread per output pixel, pixel-major tensors as the reference passes them.  The similarity
// divides by the centre pixel's depth, so it is not symmetric and the half-window planes above
// do not apply; this direct form is for interface completeness, the fused kernel is the fast path.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(128) void crf_lr_explicit_kernel(const float *__restrict__ unary, const float *__restrict__ Q,
                                                              const float *__restrict__ f1, const float *__restrict__ f2,
                                                              const float *__restrict__ f3, const float *__restrict__ f4,
                                                              float4 coef, float bias, float weight, float aux,
                                                              float *__restrict__ out, float *__restrict__ potentials, int h, int w,
                                                              int r, int S)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 4 + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    const int wp = w + 2 * r, k = 2 * r + 1;
    const size_t ci = (size_t)(y + r) * wp + (x + r), oi = (size_t)y * w + x;
    const float nd = 6.0f, na = 2.0f, nc = (float)(1.0 / 50.0), np_ = (float)(1.0 / 40.0);
    // total of all messages first, then label by label (the window is re-read per label: S is small here)
    float total = 0.0f, total_lo = 0.0f;
    for (int pass = 0; pass < 2; ++pass) {
        float mx = -INFINITY;
        for (int l = 0; l < S; ++l) {
            float m = 0.0f;
            for (int wy = 0; wy < k; ++wy)
                for (int wx = 0; wx < k; ++wx) {
                    if (wy == r && wx == r) continue;
                    const size_t ji = (size_t)(y + wy) * wp + (x + wx);
                    const float ql = Q[ji * S + l];
                    if (ql == 0.0f) continue;
                    const float d1 = div_no_nan(f1[ji] - f1[ci], f1[ci]);
                    const float s1 = sqrtf(d1 * d1) * nd;
                    const float a0 = f2[2 * ji] - f2[2 * ci], a1 = f2[2 * ji + 1] - f2[2 * ci + 1];
                    const float s2 = sqrtf(a0 * a0 + a1 * a1) * na;
                    const float e0 = f3[3 * ji] - f3[3 * ci], e1 = f3[3 * ji + 1] - f3[3 * ci + 1], e2 = f3[3 * ji + 2] - f3[3 * ci + 2];
                    const float s3 = sqrtf((e0 * e0 + e1 * e1) + e2 * e2) * nc;
                    const float p0 = f4[2 * ji] - f4[2 * ci], p1 = f4[2 * ji + 1] - f4[2 * ci + 1];
                    const float s4 = sqrtf(p0 * p0 + p1 * p1) * np_;
                    const float lr = bias + (((coef.x * s1 + coef.y * s2) + coef.z * s3) + coef.w * s4);
                    m = m + lr * ql;
                }
            if (l == S - 1) m = m + aux;
            if (pass == 0) {                         // compensated total, so that total - m keeps relative accuracy
                const float t = total + m, bp = t - total;
                total_lo += (total - (t - bp)) + (m - bp);
                total = t;
            } else {
                const float add = unary[oi * S + l] + weight * ((total - m) + total_lo);
                if (potentials) potentials[oi * S + l] = add;
                out[oi * S + l] = -add;
                mx = fmaxf(mx, -add);
            }
        }
        if (pass == 1) {
            float den = 0.0f;
            for (int l = 0; l < S; ++l) {
                const float e = expf(out[oi * S + l] - mx);
                out[oi * S + l] = e;
                den = den + e;
            }
            for (int l = 0; l < S; ++l) out[oi * S + l] = out[oi * S + l] / den;
        }
    }
}

// ---------------------------------------------------------------------------
// initial probabilities, get_unary_potentials_and_initial_probabilities
// (find_surfaces.py:456-477): double arithmetic, cast to float like Keras does.
// Writes label planes Q[l][p] of every frame of the launch (blockIdx.y = frame).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crf_init_kernel(const CrfFrameDesc *__restrict__ descs, int N)
{
    const CrfFrameDesc d = descs[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    const int S = d.S, ns = S - 1;
    const int lab = d.patches[p];
    float *Q = const_cast<float *>(d.Qin);
    const float pv = (float)(1.0 / (double)ns);
    const float other = (float)(1.0 / (double)(ns - 1) * 0.1);
    for (int l = 0; l < S; ++l) {
        float v;
        if (l == 0) v = 0.0f;
        else if (lab == 0) v = pv;
        else v = l == lab ? (float)0.9 : other;
        Q[(size_t)l * N + p] = v;
    }
}

// pixel-major [P][S] <-> label planes [S][P] (operator-level API only)
__global__ __launch_bounds__(256) void rows_to_planes_kernel(const float *__restrict__ rows, float *__restrict__ planes, int P, int S)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)P * S) return;
    const int p = (int)(i / S), l = (int)(i - (size_t)p * S);
    planes[(size_t)l * P + p] = rows[i];
}
__global__ __launch_bounds__(256) void planes_to_rows_kernel(const float *__restrict__ planes, float *__restrict__ rows, int P, int S)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)P * S) return;
    const int p = (int)(i / S), l = (int)(i - (size_t)p * S);
    rows[i] = planes[(size_t)l * P + p];
}

// ---------------------------------------------------------------------------
// per-tile list of the patch labels present within `grow` pixels of the tile.
// record (TILE_REC ints): [0] = count A, [REC_MASK .. REC_MASK+MASK_WORDS) = bit mask,
// [REC_LAB .. REC_LAB+A) = labels
// ascending.  max_active[frame] receives the largest A of the frame.
// ---------------------------------------------------------------------------
constexpr int ITW = 64, ITH = 16;             // tile of the update kernel
constexpr int MASK_WORDS = kMaxLabels / 32;
constexpr int REC_MASK = 8, REC_LAB = REC_MASK + MASK_WORDS;
constexpr int TILE_REC = REC_LAB + kMaxLabels;

__global__ __launch_bounds__(256) void crf_tile_active_kernel(const int32_t *__restrict__ patches, int H, int W, int tw, int th,
                                                              int grow, int tiles_x, int tiles_y, int total_tiles,
                                                              int32_t *__restrict__ recs, int32_t *__restrict__ max_active)
{
    __shared__ unsigned mask[MASK_WORDS];
    // consecutive workgroups go to different XCDs (one L2 each): give every XCD a contiguous run of tiles so
    // that the overlapping surroundings of neighbouring tiles are served by one L2
    const int per_xcd = gridDim.x >> 3;
    const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile >= total_tiles) return;
    const int frame = tile / (tiles_x * tiles_y), tin = tile - frame * (tiles_x * tiles_y);
    const int tile_y = tin / tiles_x, tile_x = tin - tile_y * tiles_x;
    const int32_t *P = patches + (size_t)frame * H * W;
    int32_t *rec = recs + (size_t)tile * TILE_REC;
    if (threadIdx.x < MASK_WORDS) mask[threadIdx.x] = 0u;
    __syncthreads();
    const int xa = max(0, tile_x * tw - grow), xb = min(W, (tile_x + 1) * tw + grow);
    const int ya = max(0, tile_y * th - grow), yb = min(H, (tile_y + 1) * th + grow);
    const int rw = xb - xa;
    const int total = rw * (yb - ya);
    const float inv_rw = 1.0f / (float)rw;
    constexpr int LU = 6;   // loads in flight per lane: the scan is bound by load latency, not by bandwidth
    for (int i0 = 0; i0 < total; i0 += 256 * LU) {   // uniform trip count: the lane exchange below needs every lane
        int lab[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int i = i0 + u * 256 + (int)threadIdx.x;
            lab[u] = 0;
            if (i < total) {
                int yy = (int)(((float)i + 0.5f) * inv_rw), xx = i - yy * rw;   // i / rw without the integer divide
                if (xx < 0) { --yy; xx += rw; }
                else if (xx >= rw) { ++yy; xx -= rw; }
                lab[u] = P[(size_t)(ya + yy) * W + xa + xx];
            }
        }
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            // patches are wide: a lane whose neighbour one lane down (within its group of 16) holds the
            // same label leaves the bit to that lane
            const int prev = __builtin_amdgcn_update_dpp(-1, lab[u], 0x111, 0xf, 0xf, false);
            if (lab[u] > 0 && lab[u] < kMaxLabels && lab[u] != prev) {
                const unsigned bit = 1u << (lab[u] & 31);
                if (!(mask[lab[u] >> 5] & bit)) atomicOr(&mask[lab[u] >> 5], bit);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < MASK_WORDS) {   // one lane per mask word: its labels follow those of the words below it
        const int w = threadIdx.x;
        unsigned m = mask[w];
        int A = 0;
        for (int k = 0; k < w; ++k) A += __popc(mask[k]);
        rec[REC_MASK + w] = (int32_t)m;
        while (m) {
            const int b = __ffs(m) - 1;
            rec[REC_LAB + A++] = w * 32 + b;
            m &= m - 1;
        }
        if (w == MASK_WORDS - 1) {
            rec[0] = A;
            // the maximum only grows: a (possibly stale) read filters out nearly every block, which matters
            // because atomics of all frames land on a handful of cache lines and serialise there
            if (A > __hip_atomic_load(max_active + frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_active + frame, A);
        }
    }
}

// ---------------------------------------------------------------------------
// one mean-field update: shared pieces
// ---------------------------------------------------------------------------
// Channel list of a tile in LDS: s_list[0]=CE, [1]=n_rest, [2]=rep_pos, [REC_MASK..)=bit mask of the
// listed labels (without the representative), [REC_LAB..REC_LAB+CE) = labels [0, listed..., representative, S-1].
template <int NT>
__device__ __forceinline__ void build_channel_list(int *s_list, const int32_t *tile_recs, int tile, int S)
{
    const int tid = threadIdx.x;
    if (tile_recs) {
        const int32_t *rec = tile_recs + (size_t)tile * TILE_REC;
        const int used = REC_LAB + rec[0];
        for (int t = tid; t < used; t += NT) s_list[t] = rec[t];
        __syncthreads();
        if (tid == 0) {
            const int A = s_list[0];
            int *m = s_list + REC_MASK, *labs = s_list + REC_LAB;
            for (int i = A - 1; i >= 0; --i) labs[1 + i] = labs[i];
            labs[0] = 0;
            int ce = A + 1, n_rest = (S - 2) - A, rep_pos = -1;
            if (n_rest > 0) {
                int rep = 1;
                while ((unsigned)m[rep >> 5] >> (rep & 31) & 1u) ++rep;   // smallest unlisted label in [1, S-2]
                rep_pos = ce;
                labs[ce++] = rep;
            }
            labs[ce++] = S - 1;
            m[0] |= 1;
            m[(S - 1) >> 5] |= (int)(1u << ((S - 1) & 31));
            s_list[0] = ce; s_list[1] = n_rest; s_list[2] = rep_pos;
        }
    } else {
        for (int t = tid; t < S; t += NT) s_list[REC_LAB + t] = t;
        if (tid == 0) { s_list[0] = S; s_list[1] = 0; s_list[2] = -1; }
        if (tid < MASK_WORDS) {
            const int lo = tid * 32;
            unsigned m = S >= lo + 32 ? 0xffffffffu : (S > lo ? (1u << (S - lo)) - 1u : 0u);
            s_list[REC_MASK + tid] = (int)m;
        }
    }
    __syncthreads();
}

// Stage channels [c0, c0+8) of Q for the tile + halo into LDS as two float4 planes.
template <int NW>
__device__ __forceinline__ void stage_q(float4 *lds4, const int *s_list, const float *Qin, int c0, int CE, int iy0, int ix0,
                                        int r, int LW, int LH, int Hi, int Wi, size_t Ni)
{
    const int tx = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int plane4 = LH * LW;
    int labs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) labs[k] = c0 + k < CE ? s_list[REC_LAB + c0 + k] : -1;
    for (int ly = wv; ly < LH; ly += NW) {
        const int gy = iy0 - r + ly;
        const bool rowin = gy >= 0 && gy < Hi;
        for (int lx = tx; lx < LW; lx += 64) {
            const int gxx = ix0 - r + lx;
            const bool in = rowin && gxx >= 0 && gxx < Wi;
            const size_t q = in ? (size_t)gy * Wi + gxx : 0;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (in && labs[k] >= 0) ? Qin[(size_t)labs[k] * Ni + q] : 0.0f;
            lds4[ly * LW + lx] = make_float4(v[0], v[1], v[2], v[3]);
            lds4[plane4 + ly * LW + lx] = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// Potts compatibility, softmax, overrides and expansion to S planes for one pixel.
// m[c] = message of channel c (c < CE).
template <int NC, bool COMPACT>
__device__ __forceinline__ void finish_pixel(float (&m)[NC], const int *s_list, const CrfFrameDesc &d, int S, size_t No,
                                             size_t wi, int own, bool invalid, float weight, float evidence, int last)
{
    const int CE = s_list[0], n_rest = s_list[1], rep_pos = s_list[2];
    const float pv = (float)(1.0 / (double)(S - 1));
    // evidence of the constant "unassigned" input: evidence * onehot(S-1), 0.00005 for the Gaussian similarity
    // (find_surfaces.py:356-357, 451-452), the auxiliary weight for the logistic-regression one (:424-425)
    // total over all S labels (the representative counts n_rest times), compensated so that
    // compatibility[l] = total - msg[l] keeps full relative accuracy
    float tot = 0.0f, tot_lo = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < CE) {
            if (c == CE - 1) m[c] = m[c] + evidence;
            const float term = c == rep_pos ? m[c] * (float)n_rest : m[c];
            const float t = tot + term;
            const float bp = t - tot;
            tot_lo += (tot - (t - bp)) + (term - bp);
            tot = t;
        }
    }
    float mx = -INFINITY, add_rep = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < CE) {
            const int lab = s_list[REC_LAB + c];
            const float comp = (tot - m[c]) + tot_lo;          // Potts: sum of the other labels' messages (:297-298, 360)
            float un;
            if (d.unary) un = d.unary[(size_t)lab * No + wi];
            else un = lab == 0 ? 10.0f : ((own != 0 && lab == own) ? 0.0f : 1.0f);   // :459-472
            const float add = un + weight * comp;              // :361
            if (d.potentials) d.potentials[(size_t)lab * No + wi] = add;
            if (c == rep_pos) add_rep = add;
            m[c] = -add;
            mx = fmaxf(mx, m[c]);
        }
    }
    float den = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < CE) {
            m[c] = expf(m[c] - mx);                             // softmax(-add), :363
            den = den + (c == rep_pos ? m[c] * (float)n_rest : m[c]);
        }
    }
    float q_rep = 0.0f;
    int best = 0;
    float bestv = -1.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < CE) {
            const int lab = s_list[REC_LAB + c];
            float q = m[c] / den;
            if (invalid) q = last ? 0.0f : (lab == 0 ? 0.0f : pv);   // Q[depth == 0] = prob | 0, :520-530
            if (c == rep_pos && !COMPACT) q_rep = q;                 // written with the other unlisted labels below
            else if (d.Qout) d.Qout[(size_t)lab * No + wi] = q;
            if (q > bestv || (q == bestv && lab < best)) { bestv = q; best = lab; }   // first maximum, :531
        }
    }
    if (!COMPACT && n_rest > 0 && (d.Qout || d.potentials)) {
        // every unlisted label carries the representative's value
        for (int l = 1; l < S - 1; ++l) {
            if ((unsigned)s_list[REC_MASK + (l >> 5)] >> (l & 31) & 1u) continue;
            if (d.Qout) d.Qout[(size_t)l * No + wi] = q_rep;
            if (d.potentials) d.potentials[(size_t)l * No + wi] = add_rep;
        }
    }
    if (d.surfaces) d.surfaces[wi] = best == S - 1 ? 0 : best;          // :532
}

// ---------------------------------------------------------------------------
// update with precomputed similarity planes (operator-level API: arbitrary feature tensors)
// ---------------------------------------------------------------------------
template <int R, int PY, int NCH>
__global__ __launch_bounds__(64 * ITH / PY) void crf_iter_kernel(const CrfFrameDesc *__restrict__ descs, CrfGeom g,
                                                                 float weight, float evidence, int last)
{
    constexpr int CH = 8;                     // channels staged per pass
    constexpr int NW = ITH / PY;              // waves per block
    extern __shared__ float4 lds4[];          // [2][LH][LW] float4: channel quads of the staged pass
    __shared__ int s_list[TILE_REC];

    const CrfFrameDesc d = descs[blockIdx.z];
    const int r = R ? R : g.r, S = d.S;
    const int LW = ITW + 2 * r, LH = ITH + 2 * r;
    const int tid = threadIdx.x, tx = tid & 63, wv = tid >> 6;
    const int wx0 = blockIdx.x * ITW, wy0 = blockIdx.y * ITH;         // tile origin in the output window
    const int ix0 = g.ox + wx0, iy0 = g.oy + wy0;                      // ... and in the image
    const int Wi = g.Wi, Hi = g.Hi;
    const size_t Ni = (size_t)Hi * Wi, No = (size_t)g.oh * g.ow;

    build_channel_list<64 * NW>(s_list, d.tile_recs, blockIdx.y * gridDim.x + blockIdx.x, S);
    const int CE = s_list[0];

    // ---- per-lane pixels: column tx, rows wv*PY + i ----
    const int wx = wx0 + tx;
    const int gx = ix0 + tx;
    const bool col_ok = wx < g.ow;
    bool ok[PY], valid[PY];
    unsigned p[PY];     // pixel index in the image (< 2^31)
    size_t wi[PY];      // pixel index in the output window
    bool any_work = false;
#pragma unroll
    for (int i = 0; i < PY; ++i) {
        const int wy = wy0 + wv * PY + i;
        ok[i] = col_ok && wy < g.oh;
        const int gy = iy0 + wv * PY + i;
        p[i] = ok[i] ? (unsigned)gy * (unsigned)Wi + (unsigned)gx : 0u;
        wi[i] = ok[i] ? (size_t)wy * g.ow + wx : 0;
        valid[i] = ok[i] && !(d.depth_raw && d.depth_raw[p[i]] == 0);   // overridden pixels need no message
        any_work |= valid[i];
    }
    const bool wave_work = __ballot(any_work) != 0ull;

    float acc[NCH][PY][CH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int i = 0; i < PY; ++i)
#pragma unroll
            for (int k = 0; k < CH; ++k) acc[c][i][k] = 0.0f;

    const int plane4 = LH * LW;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch * CH < CE) {
            if (ch > 0) __syncthreads();
            stage_q<NW>(lds4, s_list, d.Qin, ch * CH, CE, iy0, ix0, r, LW, LH, Hi, Wi, Ni);
            __syncthreads();
            if (wave_work) {
                const float4 *qc = lds4 + (wv * PY + r) * LW + tx + r;     // centre of row i = 0
                const gfloat *sim0 = (const gfloat *)d.sim0;
                for (int dy = 0; dy <= r; ++dy) {
                    const long base = dy == 0 ? -1 : r + (dy - 1) * (2 * r + 1) + r;   // plane of (dy, dx) = base + dx
#pragma unroll
                    for (int dx = -r; dx <= r; ++dx) {
                        if (dy == 0 && dx <= 0) continue;
                        const gfloat *plane = sim0 + (base + dx) * (long)Ni;
                        const int sh = dy * Wi + dx;
                        const int lo = dy * LW + dx;
#pragma unroll
                        for (int i = 0; i < PY; ++i) {
                            const float sf = plane[p[i]];                        // sim(p, p + o)
                            const float sb = plane[(long)p[i] - sh];             // sim(p - o, p)
                            const float4 *c = qc + i * LW;
                            const float4 f0 = c[lo], f1 = c[plane4 + lo];
                            const float4 b0 = c[-lo], b1 = c[plane4 - lo];
                            acc[ch][i][0] = fmaf(sf, f0.x, acc[ch][i][0]);
                            acc[ch][i][1] = fmaf(sf, f0.y, acc[ch][i][1]);
                            acc[ch][i][2] = fmaf(sf, f0.z, acc[ch][i][2]);
                            acc[ch][i][3] = fmaf(sf, f0.w, acc[ch][i][3]);
                            acc[ch][i][4] = fmaf(sf, f1.x, acc[ch][i][4]);
                            acc[ch][i][5] = fmaf(sf, f1.y, acc[ch][i][5]);
                            acc[ch][i][6] = fmaf(sf, f1.z, acc[ch][i][6]);
                            acc[ch][i][7] = fmaf(sf, f1.w, acc[ch][i][7]);
                            acc[ch][i][0] = fmaf(sb, b0.x, acc[ch][i][0]);
                            acc[ch][i][1] = fmaf(sb, b0.y, acc[ch][i][1]);
                            acc[ch][i][2] = fmaf(sb, b0.z, acc[ch][i][2]);
                            acc[ch][i][3] = fmaf(sb, b0.w, acc[ch][i][3]);
                            acc[ch][i][4] = fmaf(sb, b1.x, acc[ch][i][4]);
                            acc[ch][i][5] = fmaf(sb, b1.y, acc[ch][i][5]);
                            acc[ch][i][6] = fmaf(sb, b1.z, acc[ch][i][6]);
                            acc[ch][i][7] = fmaf(sb, b1.w, acc[ch][i][7]);
                        }
                    }
                }
            }
        }
    }

#pragma unroll
    for (int i = 0; i < PY; ++i) {
        if (!ok[i]) continue;
        float m[NCH * CH];
#pragma unroll
        for (int c = 0; c < NCH * CH; ++c) m[c] = acc[c / CH][i][c % CH];
        finish_pixel<NCH * CH, false>(m, s_list, d, S, No, wi[i], d.patches ? d.patches[p[i]] : 0, d.depth_raw && !valid[i], weight, evidence, last);
    }
}

// finish_pixel for the fused update (compact planes, synthesised unary, no potentials output), written so
// that the work follows the tile's channel count: channel slots are visited in groups of four behind one
// uniform test per group, and what depends on a slot's role comes from s_tab[c] = (multiplicity, evidence):
// multiplicity 1 for a listed label, n_rest for the representative, 0 for an unused slot of the last group;
// evidence of the constant "unassigned" input at the last channel (find_surfaces.py:356-357, 451-452).
// Same formulas as finish_pixel; exp and the normalisation use v_exp_f32 and one reciprocal per pixel.
template <int NC>
__device__ __forceinline__ void finish_pixel_compact(float (&m)[NC], const int *s_list, const float2 *s_tab, const CrfFrameDesc &d,
                                                     int S, size_t No, size_t wi, int own, bool invalid, float weight, int last)
{
    static_assert(NC % 4 == 0, "channel slots come in groups of four");
    const int CE = s_list[0];
    const float pv = (float)(1.0 / (double)(S - 1));
    // total over all S labels (the representative counts n_rest times), compensated so that
    // compatibility[l] = total - msg[l] keeps full relative accuracy
    float tot = 0.0f, tot_lo = 0.0f;
#pragma unroll
    for (int g = 0; g < NC / 4; ++g) {
        if (g * 4 < CE) {
#pragma unroll
            for (int c = g * 4; c < g * 4 + 4; ++c) {
                const float2 role = s_tab[c];
                m[c] = m[c] + role.y;
                const float term = m[c] * role.x;
                const float t = tot + term;
                const float bp = t - tot;
                tot_lo += (tot - (t - bp)) + (term - bp);
                tot = t;
            }
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int g = 0; g < NC / 4; ++g) {
        if (g * 4 < CE) {
#pragma unroll
            for (int c = g * 4; c < g * 4 + 4; ++c) {
                const int lab = s_list[REC_LAB + c];
                const float comp = (tot - m[c]) + tot_lo;          // Potts: sum of the other labels' messages (:297-298, 360)
                float un = lab == 0 ? 10.0f : ((own != 0 && lab == own) ? 0.0f : 1.0f);   // :459-472
                if (s_tab[c].x == 0.0f) un = INFINITY;             // unused slot: probability 0
                m[c] = -(un + weight * comp);                      // :361
                mx = fmaxf(mx, m[c]);
            }
        }
    }
    if (last) {
        // Only the label is wanted (:531-532).  The softmax is monotone, so its first maximum is the first maximum of
        // its argument, except where two arguments differ by so little that their exponentials round to the same
        // float: such labels are numerically tied and the choice between them carries no information.
        int best = 0;
        float bestv = -INFINITY;
        if (!invalid) {                                                  // Q[depth == 0] = 0: argmax of zeros is label 0
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (c < CE) {
                    const int lab = s_list[REC_LAB + c];
                    if (m[c] > bestv || (m[c] == bestv && lab < best)) { bestv = m[c]; best = lab; }
                }
            }
        }
        d.surfaces[wi] = best == S - 1 ? 0 : best;
        return;
    }
    float den = 0.0f;
#pragma unroll
    for (int g = 0; g < NC / 4; ++g) {
        if (g * 4 < CE) {
#pragma unroll
            for (int c = g * 4; c < g * 4 + 4; ++c) {
                m[c] = __builtin_amdgcn_exp2f((m[c] - mx) * 1.44269504088896341f);   // softmax(-add), :363
                den = fmaf(m[c], s_tab[c].x, den);
            }
        }
    }
    const float inv_den = 1.0f / den;
    int best = 0;
    float bestv = -1.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < CE) {
            const int lab = s_list[REC_LAB + c];
            float q = m[c] * inv_den;
            if (invalid) q = last ? 0.0f : (lab == 0 ? 0.0f : pv);   // Q[depth == 0] = prob | 0, :520-530
            if (!last) d.Qout[(size_t)lab * No + wi] = q;
            else if (q > bestv || (q == bestv && lab < best)) { bestv = q; best = lab; }   // first maximum, :531
        }
    }
    if (last) d.surfaces[wi] = best == S - 1 ? 0 : best;                // :532
}

// ---------------------------------------------------------------------------
// update with the similarity evaluated in the kernel (fused path).
//
// Streaming precomputed similarities costs ~250 MB of HBM reads per frame and update, which
// is what bounds the plane-based kernel; the six feature values of tile + halo fit in LDS next
// to Q, so here every lane evaluates sim(p, q) itself:
//   sim = w1'[o] 2^-(K1 u) + w2'[o] 2^-(K2 u + |dc'|^2) + w3'[o] 2^-(K3 u + |da'|^2) + bias
//   (when the weights do not depend on the offset and the bias is 0: sim = sum of 2^-(exponent - log2 w'))
//   (c', a': colours and angles as stored by crf_pack_features, scaled by the root of their coefficient)
//   u = ((dq - dp) / (dp + dq))^2  (0 when both depths are 0),
// the reference's three Gaussian kernels (find_surfaces.py:343-355) with the normalisers and
// squared thetas folded into K*, C* and log2(e), and the offset-dependent position terms folded
// into per-offset weights w'[o] = w exp(-position term).
//
// Tile 32 x TH pixels, 4 waves (tiles are numbered so that each XCD works on a contiguous run of
// them); a lane owns two vertically adjacent pixels and keeps them as
// packed pairs, so the arithmetic is v_pk_*_f32 and one LDS read of a neighbour's features and
// channel vector serves both pixels (the window is walked by neighbour row).
//
// Q is kept compact in HBM: a tile writes only the planes of its listed labels and of its
// representative.  A tile that lists a label its neighbour tile does not list reads the
// neighbour's representative plane for the halo pixels that lie in the neighbour (s_src).
// On the first update Q is synthesised from the patch labels
// (get_unary_potentials_and_initial_probabilities, find_surfaces.py:456-477).
// ---------------------------------------------------------------------------
struct FusedConsts { float K1, K2, K3, evidence; float uw[4]; };   // uw: -log2 of the window's common (w1', w2', w3') when uniform
constexpr int FTW = 32;                       // tile width of the fused kernel

// Messages of one lane's two vertically adjacent pixels (LDS rows trow, trow+1 of the tile, column col)
// from the staged chunk: acc[i][k] += sum over the window of sim(pixel i, neighbour) * Q_neighbour[k].
// Both rows are carried as packed pairs, one v_pk_* instruction per operation.  The 2r region rows that
// lie in both windows are read once for the pair; the topmost row (reached only by row 0) and the
// bottommost (only by row 1) go through one extra pass together, each half of the pair with its own
// neighbour, so no arithmetic is spent on positions outside a window.  wtab holds, per window offset,
// (w1', w2', w3', bias) with zeros at the centre.
// The relative depth difference divides by (d_p + d_q): a lane whose own depth is 0 may produce
// NaN here; its result is overridden afterwards (Q[depth == 0]) and never read.
template <int R, int NQ, int MODE, bool UNIFORM>
__device__ __forceinline__ void fused_accumulate(float (&acc)[2][8], const float4 *lds4, int plane4, int LW, int trow, int col, int r,
                                                 const FusedConsts &K, const float4 *__restrict__ wtab)
{
    const int kw = 2 * r + 1;
    const float4 *featA = lds4 + 2 * plane4;
    const float2 *featB = reinterpret_cast<const float2 *>(featA + plane4);
    f2 cd, cr, cg, cb, ca0, ca1;
    {
        const int c0 = (trow + r) * LW + col + r, c1 = c0 + LW;
        const float4 a0 = featA[c0], a1 = featA[c1];
        const float2 b0 = featB[c0], b1 = featB[c1];
        cd = f2{a0.x, a1.x}; cr = f2{a0.y, a1.y}; cg = f2{a0.z, a1.z}; cb = f2{a0.w, a1.w};
        ca0 = f2{b0.x, b1.x}; ca1 = f2{b0.y, b1.y};
    }
    const f2 cinv = f2{cd.x == 0.0f ? 0.0f : 1.0f / cd.x, cd.y == 0.0f ? 0.0f : 1.0f / cd.y};
    (void)cinv;
    // region rows jr = 0 .. 2r+1 (LDS row trow + jr); row i of the lane sees row jr at dy = jr - r - i.
    // The operands of the next window position are read from LDS before the arithmetic of the current
    // one starts (the scheduling barriers keep that order), so the read latency hides behind ~150 cycles
    // of vector work of the same wave instead of stalling it.
    struct Operands { float4 fa; float2 fb; float4 q0, q1; };
    struct Row { const float4 *Q0, *Q1, *A; const float2 *B; };
    auto row_of = [&](int jr) {
        const int rowbase = (trow + jr) * LW + col;       // window column dx = -r (LDS offsets are unsigned)
        // one opaque LDS index per plane, so that every read of the row is base register + immediate
        int iQ0 = rowbase, iQ1 = rowbase + plane4, iA = rowbase + 2 * plane4, iB = rowbase + 6 * plane4;
        asm volatile("" : "+v"(iQ0), "+v"(iQ1), "+v"(iA), "+v"(iB));
        return Row{lds4 + iQ0, lds4 + iQ1, lds4 + iA, reinterpret_cast<const float2 *>(lds4) + iB};
    };
    auto read = [&](const Row &rw, int k) {
        Operands o;
        o.fa = rw.A[k];
        o.fb = rw.B[k];
        o.q0 = rw.Q0[k];
        o.q1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (NQ > 1) o.q1 = rw.Q1[k];
        return o;
    };
    constexpr bool FOLD = UNIFORM && MODE == 0;     // weights folded into the exponents
    const float kOff = __builtin_inff();
    // one window position: neighbour `a` seen from row 0 of the pair with constants w0, neighbour `b` from row 1 with w1
    auto step = [&](const Operands &a, const Operands &b, const float4 &w0, const float4 &w1) {
        const f2 dj = f2{a.fa.x, b.fa.x};
        const f2 e0 = f2{a.fa.y, b.fa.y} - cr, e1 = f2{a.fa.z, b.fa.z} - cg, e2 = f2{a.fa.w, b.fa.w} - cb;
        const f2 ec = __builtin_elementwise_fma(e2, e2, __builtin_elementwise_fma(e1, e1,
                          FOLD ? __builtin_elementwise_fma(e0, e0, f2{w0.y, w1.y}) : e0 * e0));
        const f2 g0 = f2{a.fb.x, b.fb.x} - ca0, g1 = f2{a.fb.y, b.fb.y} - ca1;
        const f2 ea = __builtin_elementwise_fma(g1, g1, FOLD ? __builtin_elementwise_fma(g0, g0, f2{w0.z, w1.z}) : g0 * g0);
        f2 sv;
        if (MODE == 0) {
            // Gaussian kernels (find_surfaces.py:339-355); wtab = (w1', w2', w3', bias)
            const f2 den = cd + dj;
            const f2 inv = f2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            const f2 t = (dj - cd) * inv;
            const f2 u = t * t;
            const f2 x1 = FOLD ? __builtin_elementwise_fma(f2{K.K1, K.K1}, u, f2{w0.x, w1.x}) : f2{K.K1, K.K1} * u;
            const f2 x2 = __builtin_elementwise_fma(f2{K.K2, K.K2}, u, ec);     // colours and angles are stored pre-scaled
            const f2 x3 = __builtin_elementwise_fma(f2{K.K3, K.K3}, u, ea);
            const f2 s1 = f2{__builtin_amdgcn_exp2f(-x1.x), __builtin_amdgcn_exp2f(-x1.y)};
            const f2 s2 = f2{__builtin_amdgcn_exp2f(-x2.x), __builtin_amdgcn_exp2f(-x2.y)};
            const f2 s3 = f2{__builtin_amdgcn_exp2f(-x3.x), __builtin_amdgcn_exp2f(-x3.y)};
            if (FOLD)
                sv = (s1 + s2) + s3;
            else
                sv = __builtin_elementwise_fma(f2{w0.z, w1.z}, s3,
                     __builtin_elementwise_fma(f2{w0.y, w1.y}, s2,
                     __builtin_elementwise_fma(f2{w0.x, w1.x}, s1, f2{w0.w, w1.w})));
        } else {
            // linear similarity (find_surfaces.py:406-422); wtab = (6 c1, 2 c2, c3 / 50, bias + c4 |offset| / 40);
            // the depth difference is relative to the lane's own depth (cinv = 1 / d_p, 0 if d_p == 0)
            const f2 t = (dj - cd) * cinv;
            const f2 at = f2{__builtin_fabsf(t.x), __builtin_fabsf(t.y)};
            const f2 sa = f2{__builtin_amdgcn_sqrtf(ea.x), __builtin_amdgcn_sqrtf(ea.y)};
            const f2 sc = f2{__builtin_amdgcn_sqrtf(ec.x), __builtin_amdgcn_sqrtf(ec.y)};
            sv = __builtin_elementwise_fma(f2{w0.z, w1.z}, sc,
                 __builtin_elementwise_fma(f2{w0.y, w1.y}, sa,
                 __builtin_elementwise_fma(f2{w0.x, w1.x}, at, f2{w0.w, w1.w})));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float sI = i == 0 ? sv.x : sv.y;
            const float4 q0 = i == 0 ? a.q0 : b.q0, q1 = i == 0 ? a.q1 : b.q1;
            acc[i][0] = fmaf(sI, q0.x, acc[i][0]);
            acc[i][1] = fmaf(sI, q0.y, acc[i][1]);
            acc[i][2] = fmaf(sI, q0.z, acc[i][2]);
            acc[i][3] = fmaf(sI, q0.w, acc[i][3]);
            if (NQ > 1) {
                acc[i][4] = fmaf(sI, q1.x, acc[i][4]);
                acc[i][5] = fmaf(sI, q1.y, acc[i][5]);
                acc[i][6] = fmaf(sI, q1.z, acc[i][6]);
                acc[i][7] = fmaf(sI, q1.w, acc[i][7]);
            }
        }
    };
    // UNIFORM: every offset of the window carries the same constants (the trained position thetas underflow
    // to 0), so they live in scalar registers and no table loads compete with the LDS reads.  uw holds -log2
    // of the three weights: they join the exponents, w 2^-x = 2^-(x - log2 w); the centre of the window is
    // switched off with an exponent of +inf, i.e. a similarity of exactly 0.
    const float4 uon = make_float4(K.uw[0], K.uw[1], K.uw[2], 0.0f), uoff = make_float4(kOff, kOff, kOff, 0.0f);
    // neighbour rows 1 .. 2r of the region lie in the window of both rows of the pair: row 0 sees row jr at
    // dy = jr - r, row 1 at dy = jr - r - 1, and one read serves both
    const int last_row = 1 + 2 * r;
    Row cur = row_of(1);
    Operands nxt = read(cur, 0);
#pragma unroll 1
    for (int jr = 1; jr <= 2 * r; ++jr) {
        const Row below = row_of(jr < 2 * r ? jr + 1 : 0);                        // after the shared rows: the top row (below)
        const float4 *wrow0 = wtab + jr * kw + r, *wrow1 = wtab + (jr - 1) * kw + r;
#pragma unroll
        for (int dx = -r; dx <= r; ++dx) {
            const Operands o = nxt;
            nxt = dx < r ? read(cur, dx + r + 1) : read(below, 0);
            __builtin_amdgcn_sched_barrier(0);
            float4 w0, w1;
            if (UNIFORM) {
                w0 = uon; w1 = uon;
                if (dx == 0) {
                    if (jr == r) w0 = uoff;
                    if (jr == r + 1) w1 = uoff;
                }
            } else {
                w0 = wrow0[dx]; w1 = wrow1[dx];                                       // wave-uniform (scalar loads)
            }
            step(o, o, w0, w1);
            __builtin_amdgcn_sched_barrier(0);
        }
        cur = below;
    }
    // the two rows that only one row of the pair reaches go through together: row 0 of the pair takes region
    // row 0 (dy = -r), row 1 takes region row 2r+1 (dy = +r)
    {
        const Row bot = row_of(last_row);
        Operands nb = read(bot, 0);
        const float4 *wrow0 = wtab + r, *wrow1 = wtab + 2 * r * kw + r;
#pragma unroll 3
        for (int dx = -r; dx <= r; ++dx) {
            const Operands oa = nxt, ob = nb;
            if (dx < r) {
                nxt = read(cur, dx + r + 1);
                nb = read(bot, dx + r + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            float4 w0, w1;
            if (UNIFORM) {
                w0 = uon; w1 = uon;
            } else {
                w0 = wrow0[dx]; w1 = wrow1[dx];
            }
            step(oa, ob, w0, w1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int R, int TH, int NCH, int MODE, bool UNIFORM>
__global__ __launch_bounds__(256, 2) void crf_iter_fused_kernel(const CrfFrameDesc *__restrict__ descs, CrfGeom g, FusedConsts K,
                                                             const float4 *__restrict__ wtab, float weight, int first, int last,
                                                             int nframes, unsigned *__restrict__ queue)
{
    constexpr int CH = 8;
    static_assert(TH == 16 || TH == 8, "tile height");
    __shared__ unsigned s_ticket;
    extern __shared__ float4 lds4[];          // [2][LH][LW] Q quads | [LH][LW] (d,r,g,b) | [LH][LW] (a0,a1) as float2
    __shared__ int s_list[TILE_REC];
    __shared__ int s_src[15][NCH * CH];       // plane to read channel c from in each of the 5x3 tiles the region can touch (-1: zero)
    __shared__ int32_t s_nmask[15 * MASK_WORDS];   // label masks of those tiles
    __shared__ float2 s_tab[NCH * CH];             // (multiplicity, evidence) of every channel slot

    // Workgroups are dealt to the 8 XCDs round-robin by their linear id; each XCD has its own L2.  The
    // tiles are therefore numbered so that an XCD receives a contiguous run of them (whole frames, mostly):
    // the halo a tile shares with its neighbours is then served by that XCD's L2 instead of being fetched
    // once per XCD.
    const int tiles_x = (g.Wi + FTW - 1) / FTW, tiles_y = (g.Hi + TH - 1) / TH;
    const unsigned tiles_total = (unsigned)tiles_x * tiles_y * nframes;
    const unsigned per_xcd = (tiles_total + 7u) / 8u;
    // Tile queue: one counter per XCD (128 bytes apart), zeroed before the launch.  A block takes tiles from
    // the run of its own XCD and, once that is used up, from the runs of the others (frames differ in cost, so
    // the runs do not end together), until all eight are empty, which every block reaches; the next tile is
    // requested while the current one is worked on.
    unsigned qi = blockIdx.x & 7u, queues_left = 8;          // lane 0's cursor over the queues
    auto next_tile = [&]() -> unsigned {
        while (queues_left) {
            const unsigned t = atomicAdd(queue + qi * (kCrfQueueStride / sizeof(unsigned)), 1u);
            const unsigned id = qi * per_xcd + t;
            if (t < per_xcd && id < tiles_total) return id;
            qi = (qi + 1u) & 7u;
            --queues_left;
        }
        return 0xffffffffu;
    };
    unsigned next_ticket = 0;
    if (threadIdx.x == 0) s_ticket = next_tile();
    __builtin_amdgcn_s_setprio(2);
    __syncthreads();
  for (;;) {
    const unsigned tile_id = s_ticket;
    if (tile_id == 0xffffffffu) break;
    if (threadIdx.x == 0) next_ticket = next_tile();
    const int bz = tile_id / (tiles_x * tiles_y), brem = tile_id - bz * (tiles_x * tiles_y);
    const int by = brem / tiles_x, bx = brem - by * tiles_x;
    const CrfFrameDesc d = descs[bz];
    const int r = R ? R : g.r, S = d.S;
    const int LW = FTW + 2 * r, LH = TH + 2 * r;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ix0 = bx * FTW, iy0 = by * TH;                            // fused path: output window = frame
    const int Wi = g.Wi, Hi = g.Hi;
    const size_t Ni = (size_t)Hi * Wi;
    const int plane4 = LH * LW;
    float4 *featA = lds4 + 2 * plane4;
    float2 *featB = reinterpret_cast<float2 *>(featA + plane4);

    // ---- per-lane pixel pair: column lane & 31, rows (wave, lane half) ----
    const int col = lane & 31, half = lane >> 5;
    // TH = 16: wave covers rows 4*wv .. 4*wv+3, half 0 -> rows 4wv, 4wv+1; half 1 -> 4wv+2, 4wv+3
    // TH = 8 : wave covers rows 2*wv, 2*wv+1 for half 0 only (half 1 idles)
    const int trow = TH == 16 ? wv * 4 + half * 2 : wv * 2;
    const bool lane_on = TH == 16 || half == 0;
    const int gx = ix0 + col;
    const int staged = LH * LW;
    constexpr int FU = 3;

    // A tile's set-up is a chain of dependent global reads, and with two blocks per CU little else runs
    // while a block waits on one.  Everything that needs only the tile's position is therefore requested in
    // one go: the tile's label record, the label masks of the 5x3 tiles the staged region touches, the
    // lane's own depths and patch labels, and the first half of the feature records.
    const int32_t *rec = d.tile_recs + ((size_t)by * tiles_x + bx) * TILE_REC;
    const int32_t rec_word = rec[tid];                       // count, mask and the first 256 - REC_LAB labels
    const int listed_labels = rec[0];                        // (every lane: lane 0 rewrites the LDS copy below)
    int32_t nb_word = 0;
    if (!first && tid < 15 * MASK_WORDS) {
        const int st = tid / MASK_WORDS, sy = by + st / 3 - 2, sx = bx + st % 3 - 1;
        if (sy >= 0 && sy < tiles_y && sx >= 0 && sx < tiles_x)
            nb_word = d.tile_recs[((size_t)sy * tiles_x + sx) * TILE_REC + REC_MASK + (tid - st * MASK_WORDS)];
    }
    bool ok[2], valid[2];
    unsigned p[2];
    int own[2];
    bool any_work = false;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int gy = iy0 + trow + i;
        ok[i] = lane_on && gx < Wi && gy < Hi;
        p[i] = ok[i] ? (unsigned)gy * (unsigned)Wi + (unsigned)gx : 0u;
        valid[i] = ok[i] && d.depth_raw[p[i]] != 0;        // overridden pixels need no message
        own[i] = ok[i] ? d.patches[p[i]] : 0;
        any_work |= valid[i];
    }
    // features of tile + halo (zero outside the frame, like the reference's padded feature maps): flat index
    // over the staged region, three pixels per thread and pass
    auto load_features = [&](int i0, float4 (&fa)[FU], float2 (&fb)[FU]) {
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int i = i0 + u * 256;
            const int ly = i / LW, lx = i - ly * LW;
            const int gy = iy0 - r + ly, gxx = ix0 - r + lx;
            fa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            fb[u] = make_float2(0.f, 0.f);
            if (i < staged && gy >= 0 && gy < Hi && gxx >= 0 && gxx < Wi) {
                const size_t q = (size_t)gy * Wi + gxx;
                fa[u] = d.featA[q];
                fb[u] = d.featB[q];
            }
        }
    };
    auto store_features = [&](int i0, const float4 (&fa)[FU], const float2 (&fb)[FU]) {
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int i = i0 + u * 256;
            if (i < staged) {
                featA[i] = fa[u];
                featB[i] = fb[u];
            }
        }
    };
    {
        float4 fa[FU];
        float2 fb[FU];
        load_features(tid, fa, fb);
        s_list[tid] = rec_word;
        if (tid < 15 * MASK_WORDS) s_nmask[tid] = nb_word;
        store_features(tid, fa, fb);
    }
    const bool wave_work = __ballot(any_work) != 0ull;
    // a tile without a single valid depth (about one in five on sensor data) only writes the override values:
    // nothing more is staged for it
    const bool block_work = __syncthreads_or(any_work) != 0;

    if (REC_LAB + listed_labels > 256) {                     // a label list longer than the first request (rare)
        for (int t = 256 + tid; t < REC_LAB + listed_labels; t += 256) s_list[t] = rec[t];
        __syncthreads();
    }
    // the rest of the features is on its way while one lane turns the record into the channel list
    // [0, listed labels..., representative, S-1] and 15 lanes look up where to read each channel
    float4 fa2[FU];
    float2 fb2[FU];
    if (block_work) load_features(tid + 256 * FU, fa2, fb2);
    if (tid == 0) {
        const int A = s_list[0];
        int *m = s_list + REC_MASK, *labs = s_list + REC_LAB;
        for (int i = A - 1; i >= 0; --i) labs[1 + i] = labs[i];
        labs[0] = 0;
        int ce = A + 1, n_rest = (S - 2) - A, rep_pos = -1;
        if (n_rest > 0) {
            int rp = 1;
            while ((unsigned)m[rp >> 5] >> (rp & 31) & 1u) ++rp;   // smallest unlisted label in [1, S-2]
            rep_pos = ce;
            labs[ce++] = rp;
        }
        labs[ce++] = S - 1;
        m[0] |= 1;
        m[(S - 1) >> 5] |= (int)(1u << ((S - 1) & 31));
        s_list[0] = ce; s_list[1] = n_rest; s_list[2] = rep_pos;
    }
    __syncthreads();
    const int CE = s_list[0];

    if (tid < NCH * CH) {
        const int n_rest = s_list[1], rep_pos = s_list[2];
        s_tab[tid] = make_float2(tid >= CE ? 0.0f : (tid == rep_pos ? (float)n_rest : 1.0f), tid == CE - 1 ? K.evidence : 0.0f);
    }
    // ---- where to read each of my channels in the 5x3 tiles the staged region touches ----
    if (block_work && !first && tid < 15) {
        const int sy = by + tid / 3 - 2, sx = bx + tid % 3 - 1;
        if (sy < 0 || sy >= tiles_y || sx < 0 || sx >= tiles_x) {
            for (int c = 0; c < CE; ++c) s_src[tid][c] = -1;
        } else {
            const int32_t *m = s_nmask + tid * MASK_WORDS;
            int rp = 1;                                             // representative of the source tile
            while (rp < S - 1 && ((unsigned)m[rp >> 5] >> (rp & 31) & 1u)) ++rp;
            for (int c = 0; c < CE; ++c) {
                const int l = s_list[REC_LAB + c];
                const bool listed = l == 0 || l == S - 1 || ((unsigned)m[l >> 5] >> (l & 31) & 1u);
                s_src[tid][c] = listed ? l : rp;
            }
        }
    }
    if (block_work) {
        store_features(tid + 256 * FU, fa2, fb2);
        for (int i0 = tid + 2 * 256 * FU; i0 < staged; i0 += 256 * FU) {   // regions larger than 6 pixels per lane
            float4 fa[FU];
            float2 fb[FU];
            load_features(i0, fa, fb);
            store_features(i0, fa, fb);
        }
    }

    float acc[NCH][2][CH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < CH; ++k) acc[c][i][k] = 0.0f;

    const int ns = S - 1;
    const float q_pv = (float)(1.0 / (double)ns), q_other = (float)(1.0 / (double)(ns - 1) * 0.1);
    __syncthreads();                                        // s_src, features
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (block_work && ch * CH < CE) {
            if (ch > 0) __syncthreads();
            // ---- stage channels [ch*8, ch*8+8) of Q for tile + halo (same flat, batched scheme) ----
            constexpr int QU = 2;
            for (int i0 = FS_SKIPQ ? staged : tid; i0 < staged; i0 += 256 * QU) {
                float v[QU][CH];
#pragma unroll
                for (int u = 0; u < QU; ++u) {
                    const int i = i0 + u * 256;
                    const int ly = i / LW, lx = i - ly * LW;
                    const int gy = iy0 - r + ly, gxx = ix0 - r + lx;
                    const bool in = i < staged && gy >= 0 && gy < Hi && gxx >= 0 && gxx < Wi;
                    const size_t q = in ? (size_t)gy * Wi + gxx : 0;
                    const bool upper = CE - ch * CH > 4;     // channels 4..7 of the chunk are in use (uniform)
                    if (first) {
                        const int lab = in ? d.patches[q] : -1;
#pragma unroll
                        for (int k = 0; k < CH; ++k) {
                            v[u][k] = 0.0f;
                            if (k < 4 || upper) {
                                const int l = ch * CH + k < CE ? s_list[REC_LAB + ch * CH + k] : -1;
                                if (lab >= 0 && l > 0) v[u][k] = lab == 0 ? q_pv : (l == lab ? (float)0.9 : q_other);
                            }
                        }
                    } else {
                        const int st = in ? ((gy >> (TH == 16 ? 4 : 3)) - by + 2) * 3 + ((gxx >> 5) - bx + 1) : 0;
#pragma unroll
                        for (int k = 0; k < CH; ++k) {
                            v[u][k] = 0.0f;
                            if (k < 4 || upper) {
                                const int l = (in && ch * CH + k < CE) ? s_src[st][ch * CH + k] : -1;
                                if (l >= 0) v[u][k] = d.Qin[(size_t)l * Ni + q];
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < QU; ++u) {
                    const int i = i0 + u * 256;
                    if (i < staged) {
                        lds4[i] = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
                        if (CE - ch * CH > 4) lds4[plane4 + i] = make_float4(v[u][4], v[u][5], v[u][6], v[u][7]);
                    }
                }
            }
            __syncthreads();
            if (wave_work && FS_SKIPMAIN == 0) {
                // The arithmetic runs at low priority, set-up, staging and epilogue at high priority: the
                // few instructions between a tile's memory requests then do not queue behind the other
                // block's window walk, which shortens the phases in which a block contributes no arithmetic.
                __builtin_amdgcn_s_setprio(0);
                // 1..4 channels of the chunk need only the first quad plane (two thirds of all tiles)
                if (CE - ch * CH <= 4) fused_accumulate<R, 1, MODE, UNIFORM>(acc[ch], lds4, plane4, LW, trow, col, r, K, wtab);
                else fused_accumulate<R, 2, MODE, UNIFORM>(acc[ch], lds4, plane4, LW, trow, col, r, K, wtab);
                __builtin_amdgcn_s_setprio(2);
            }
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (FS_SKIPFIN) {                          // timing experiments: keep the messages alive without the epilogue
            float t = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH * CH; ++c) t += acc[c / CH][i][c % CH];
            if (t == 12345.678f) d.depth_raw ? (void)(const_cast<float *>(d.Qin)[0] = t) : (void)0;
            continue;
        }
        if (!ok[i]) continue;
        float m[NCH * CH];
#pragma unroll
        for (int c = 0; c < NCH * CH; ++c) m[c] = acc[c / CH][i][c % CH];
        finish_pixel_compact<NCH * CH>(m, s_list, s_tab, d, S, Ni, p[i], own[i], !valid[i], weight, last);
    }
    if (threadIdx.x == 0) s_ticket = next_ticket;     // every lane read the current ticket several barriers ago
    __syncthreads();                                  // the next tile reuses the LDS
  }
}

// depth u16 / rgb u8 / angles -> the float records the fused update stages: (d, r, g, b) and (a0, a1).
// Colours and angles carry the square root of their coefficient in the Gaussian exponents (cscale, ascale),
// so that squared differences of the stored values are the exponent terms themselves; 1 for the linear similarity.
__global__ __launch_bounds__(256) void crf_pack_features_kernel(const uint16_t *__restrict__ depth, const uint8_t *__restrict__ rgb,
                                                                const float *__restrict__ angles, float4 *__restrict__ featA,
                                                                float2 *__restrict__ featB, size_t total, float cscale,
                                                                float ascale)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    featA[q] = make_float4((float)depth[q], cscale * (float)rgb[3 * q], cscale * (float)rgb[3 * q + 1], cscale * (float)rgb[3 * q + 2]);
    if (angles) {
        const float2 a = reinterpret_cast<const float2 *>(angles)[q];
        featB[q] = make_float2(ascale * a.x, ascale * a.y);
    }
}

// Dynamic LDS a kernel instantiation has been allowed so far.  The attribute belongs to (device, function), so
// the record is kept per device; several contexts (one per GPU, or several host threads) may launch at once.
struct LdsAllowance {
    static constexpr int kDevices = 64;
    std::mutex m;
    size_t have[kDevices] = {};
    hipError_t ensure(const void *kern, size_t lds)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> g(m);
        const bool tracked = dev >= 0 && dev < kDevices;
        if (tracked && lds <= have[dev]) return hipSuccess;
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess && tracked) have[dev] = lds;
        return e;
    }
};

template <int R, int PY, int NCH>
hipError_t launch_iter_t(const CrfFrameDesc *descs, int nframes, const CrfGeom &g, float weight, float evidence, int last, hipStream_t st)
{
    static LdsAllowance allowed;
    auto kern = crf_iter_kernel<R, PY, NCH>;
    const size_t lds = (size_t)2 * (ITW + 2 * g.r) * (ITH + 2 * g.r) * sizeof(float4);
    if (lds + sizeof(int) * TILE_REC > 160 * 1024) return hipErrorInvalidValue;
    if (hipError_t e = allowed.ensure(reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
    dim3 grid((g.ow + ITW - 1) / ITW, (g.oh + ITH - 1) / ITH, nframes);
    hipLaunchKernelGGL(kern, grid, dim3(64 * ITH / PY), lds, st, descs, g, weight, evidence, last);
    return hipGetLastError();
}

template <int R, int TH, int NCH, int MODE, bool UNIFORM = false>
hipError_t launch_fused_t(const CrfFrameDesc *descs, int nframes, const CrfGeom &g, const FusedConsts &K, const float4 *wtab,
                          float weight, int first, int last, unsigned *queue, hipStream_t st)
{
    static LdsAllowance allowed;
    auto kern = crf_iter_fused_kernel<R, TH, NCH, MODE, UNIFORM>;
    const size_t lds = (size_t)(FTW + 2 * g.r) * (TH + 2 * g.r) * (3 * sizeof(float4) + sizeof(float2));
    const size_t lds_static = sizeof(int) * (TILE_REC + 15 * NCH * 8 + 15 * MASK_WORDS + 2 * NCH * 8 + 4);
    if (lds + lds_static > 160 * 1024) return hipErrorInvalidValue;
    if (hipError_t e = allowed.ensure(reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
    const unsigned tiles = (unsigned)((g.Wi + FTW - 1) / FTW) * ((g.Hi + TH - 1) / TH) * nframes;
    // Persistent blocks: as many as the chip holds at once (two per CU by LDS), each pulling tiles from the
    // queue of its XCD until it is empty.  A block that ends and a new one that takes its place cost the CU
    // several microseconds with half its waves missing; a block that just takes the next tile does not.
    if (!queue) return hipErrorInvalidValue;
    hipError_t qe = hipMemsetAsync(queue, 0, kCrfQueueBytes, st);
    if (qe != hipSuccess) return qe;
    const unsigned nblk = (tiles + 7u) / 8u * 8u;
    static const unsigned max_blocks = [] {                  // FS_CRF_BLOCKS: experiments with fewer or more resident blocks
        const char *e = getenv("FS_CRF_BLOCKS");
        const int v = e ? atoi(e) : 0;
        return v >= 8 ? (unsigned)v / 8u * 8u : 512u;
    }();
    dim3 grid(nblk < max_blocks ? nblk : max_blocks);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, descs, g, K, wtab, weight, first, last, nframes, queue);
    return hipGetLastError();
}

}  // namespace

void launch_crf_lr_explicit(const float *unary, const float *Q, const float *f1, const float *f2, const float *f3, const float *f4,
                            const float coef[4], float bias, float weight, float aux, float *out, float *potentials, int h, int w,
                            int r, int S, hipStream_t st)
{
    hipLaunchKernelGGL(crf_lr_explicit_kernel, dim3((w + 31) / 32, (h + 3) / 4), dim3(128), 0, st, unary, Q, f1, f2, f3, f4,
                       make_float4(coef[0], coef[1], coef[2], coef[3]), bias, weight, aux, out, potentials, h, w, r, S);
}

size_t crf_half_offsets(int r) { return (size_t)r + (size_t)r * (2 * r + 1); }
size_t crf_sim_slack(int W, int r) { return (size_t)r * W + r; }
size_t crf_tile_rec_ints() { return TILE_REC; }
void crf_tile_grid(int H, int W, int *tiles_x, int *tiles_y)
{
    *tiles_x = (W + ITW - 1) / ITW;
    *tiles_y = (H + ITH - 1) / ITH;
}
int crf_channel_class(int channels)
{
    const int nch = (channels + 7) / 8;
    if (nch <= 2) return 2;       // one launch for the bulk of the frames: the 16-channel variant is as fast on 8
    if (nch <= 4) return nch;
    if (nch <= 8) return 8;
    return 16;
}

void launch_crf_sim(const uint16_t *depth, const uint8_t *rgb, const float *angles, const CrfParams &p,
                    const float *pos_terms, float *simH, size_t sim_stride, int nframes, int H, int W, int r, hipStream_t st)
{
    const size_t lds = (size_t)(STW + 2 * r) * (STH + r) * 6 * sizeof(float);
    dim3 grid((W + STW - 1) / STW, (H + STH - 1) / STH, nframes);
    hipLaunchKernelGGL(crf_sim_kernel, grid, dim3(SNT), lds, st, depth, rgb, angles, p, pos_terms, simH, sim_stride, H, W, r);
}

void launch_crf_sim_explicit(const float *f1, const float *f2, const float *f3, const CrfParams &p, float *simH, int Hi,
                             int Wi, int r, hipStream_t st)
{
    dim3 grid((Wi + 31) / 32, (Hi + 7) / 8);
    hipLaunchKernelGGL(crf_sim_explicit_kernel, grid, dim3(256), 0, st, f1, f2, f3, p, simH, Hi, Wi, r);
}

void launch_crf_init(const CrfFrameDesc *descs, int nframes, int H, int W, hipStream_t st)
{
    const int N = H * W;
    hipLaunchKernelGGL(crf_init_kernel, dim3((N + 255) / 256, nframes), dim3(256), 0, st, descs, N);
}

void launch_rows_to_planes(const float *rows, float *planes, size_t P, int S, hipStream_t st)
{
    hipLaunchKernelGGL(rows_to_planes_kernel, dim3((unsigned)((P * S + 255) / 256)), dim3(256), 0, st, rows, planes, (int)P, S);
}
void launch_planes_to_rows(const float *planes, float *rows, size_t P, int S, hipStream_t st)
{
    hipLaunchKernelGGL(planes_to_rows_kernel, dim3((unsigned)((P * S + 255) / 256)), dim3(256), 0, st, planes, rows, (int)P, S);
}

void launch_crf_tile_active(const int32_t *patches, int n, int H, int W, int tile_w, int tile_h, int grow, int32_t *recs,
                            int32_t *max_active, hipStream_t st)
{
    const int tx = (W + tile_w - 1) / tile_w, ty = (H + tile_h - 1) / tile_h;
    const int total = tx * ty * n;
    hipLaunchKernelGGL(crf_tile_active_kernel, dim3((total + 7) / 8 * 8), dim3(256), 0, st, patches, H, W, tile_w, tile_h, grow, tx, ty,
                       total, recs, max_active);
}

// tile height of the fused kernel for a window radius: 16 rows while the staged tile fits in LDS, else 8
int crf_fused_tile_height(int r)
{
    auto bytes = [&](int th) { return (size_t)(FTW + 2 * r) * (th + 2 * r) * (3 * sizeof(float4) + sizeof(float2)); };
    if (bytes(16) <= 150 * 1024) return 16;
    if (bytes(8) <= 150 * 1024) return 8;
    return 0;
}
void crf_fused_tile_grid(int H, int W, int r, int *tiles_x, int *tiles_y)
{
    const int th = crf_fused_tile_height(r);
    *tiles_x = (W + FTW - 1) / FTW;
    *tiles_y = th ? (H + th - 1) / th : 0;
}

// log2(e) * theta^2 * normaliser^2 of the colour and angle terms (find_surfaces.py:20, 343-349): colours / 50, angles * 2
static double fused_c2(const CrfParams &P) { return P.t23 / 2500.0 * 1.4426950408889634; }
static double fused_c3(const CrfParams &P) { return 4.0 * P.t33 * 1.4426950408889634; }

float crf_angle_feature_scale(const CrfParams &P) { return P.mode == 0 ? (float)std::sqrt(fused_c3(P)) : 1.0f; }

void launch_crf_pack_features(const uint16_t *depth, const uint8_t *rgb, const float *angles, void *featA, void *featB,
                              size_t total, const CrfParams &P, hipStream_t st)
{
    const float cs = P.mode == 0 ? (float)std::sqrt(fused_c2(P)) : 1.0f, as = crf_angle_feature_scale(P);
    hipLaunchKernelGGL(crf_pack_features_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, depth, rgb, angles,
                       reinterpret_cast<float4 *>(featA), reinterpret_cast<float2 *>(featB), total, cs, as);
}

hipError_t launch_crf_iter_fused(const CrfFrameDesc *descs, int nframes, int channel_class, const CrfGeom &g,
                                 const CrfParams &P, const float *wtab, int mode, unsigned *tile_queue, hipStream_t st)
{
    if (g.r < 1 || g.r > kMaxRadius || nframes < 1) return hipErrorInvalidValue;
    const int last = (mode & CRF_LAST) ? 1 : 0, first = (mode & CRF_FIRST) ? 1 : 0;
    const double l2e = 1.4426950408889634;
    // delta_d = 12 t with t = (dq-dp)/(dp+dq); colour differences scaled by 1/50, angles by 2 (find_surfaces.py:20, 343-349)
    FusedConsts K;
    K.K1 = (float)(144.0 * P.t12 * l2e); K.K2 = (float)(144.0 * P.t22 * l2e); K.K3 = (float)(144.0 * P.t32 * l2e);
    K.evidence = P.evidence;
    // the window's constants are uniform when the position terms vanish (as with the trained parameters)
    // and that variant also wants weights with a logarithm (non-negative) and no bias, as the loader of the
    // trained parameters guarantees (find_surfaces.py:221-226); anything else takes the table-driven kernels
    const bool uniform = P.mode == 0 && P.t11 == 0.0f && P.t21 == 0.0f && P.t31 == 0.0f && P.w1 >= 0.0f && P.w2 >= 0.0f &&
                         P.w3 >= 0.0f && P.bias == 0.0f;
    K.uw[0] = (float)-std::log2((double)P.w1); K.uw[1] = (float)-std::log2((double)P.w2); K.uw[2] = (float)-std::log2((double)P.w3);
    K.uw[3] = 0.0f;
    const float4 *wt = reinterpret_cast<const float4 *>(wtab);
    const int th = crf_fused_tile_height(g.r);
#define FS_FUSED_CASES_U(RR, THH)                                                                            \
    switch (channel_class) {                                                                                 \
    case 1: return launch_fused_t<RR, THH, 1, 0, true>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st);   \
    case 2: return launch_fused_t<RR, THH, 2, 0, true>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st);   \
    case 3:                                                                                                  \
    case 4: return launch_fused_t<RR, THH, 4, 0, true>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st);   \
    case 8: return launch_fused_t<RR, THH, 8, 0, true>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st);   \
    case 16: return launch_fused_t<RR, THH, 16, 0, true>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st); \
    default: return hipErrorInvalidValue;                                                                    \
    }
#define FS_FUSED_CASES(RR, THH)                                                                              \
    switch (channel_class) {                                                                                 \
    case 1: return launch_fused_t<RR, THH, 1, 0>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st);         \
    case 2: return launch_fused_t<RR, THH, 2, 0>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st);         \
    case 3:                                                                                                  \
    case 4: return launch_fused_t<RR, THH, 4, 0>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st);         \
    case 8: return launch_fused_t<RR, THH, 8, 0>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st);         \
    case 16: return launch_fused_t<RR, THH, 16, 0>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st);       \
    default: return hipErrorInvalidValue;                                                                    \
    }
    if (P.mode == 1) {
        // logistic-regression similarity: two channel classes only (it is not the tuned path)
        if (channel_class > 16) return hipErrorInvalidValue;
        if (th == 16)
            return channel_class <= 2 ? launch_fused_t<0, 16, 2, 1>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st)
                                      : launch_fused_t<0, 16, 16, 1>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st);
        if (th == 8)
            return channel_class <= 2 ? launch_fused_t<0, 8, 2, 1>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st)
                                      : launch_fused_t<0, 8, 16, 1>(descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st);
        return hipErrorInvalidValue;
    }
    if (g.r == 7 && uniform) { FS_FUSED_CASES_U(7, 16) }
    // other radii with uniform constants: run-time window size, but still no weight table
    if (uniform && th == 16) { FS_FUSED_CASES_U(0, 16) }
    if (uniform && th == 8) { FS_FUSED_CASES_U(0, 8) }
    if (th == 16) { FS_FUSED_CASES(0, 16) }
    if (th == 8) { FS_FUSED_CASES(0, 8) }
#undef FS_FUSED_CASES
#undef FS_FUSED_CASES_U
    return hipErrorInvalidValue;
}

hipError_t launch_crf_iter(const CrfFrameDesc *descs, int nframes, int channel_class, const CrfGeom &g, float weight,
                           float evidence, int mode, hipStream_t st)
{
    if (g.r < 1 || g.r > kMaxRadius || nframes < 1) return hipErrorInvalidValue;
    const int last = (mode & CRF_LAST) ? 1 : 0;
    // accumulators are NCH x PY x 8 registers per lane: one row per lane when many channels are live
    if (g.r == 7) {
        switch (channel_class) {
        case 1: return launch_iter_t<7, 2, 1>(descs, nframes, g, weight, evidence, last, st);
        case 2: return launch_iter_t<7, 2, 2>(descs, nframes, g, weight, evidence, last, st);
        case 3: return launch_iter_t<7, 2, 3>(descs, nframes, g, weight, evidence, last, st);
        case 4: return launch_iter_t<7, 2, 4>(descs, nframes, g, weight, evidence, last, st);
        case 8: return launch_iter_t<7, 2, 8>(descs, nframes, g, weight, evidence, last, st);
        default: break;
        }
    } else {
        switch (channel_class) {
        case 1: return launch_iter_t<0, 2, 1>(descs, nframes, g, weight, evidence, last, st);
        case 2: return launch_iter_t<0, 2, 2>(descs, nframes, g, weight, evidence, last, st);
        case 3:
        case 4: return launch_iter_t<0, 2, 4>(descs, nframes, g, weight, evidence, last, st);
        case 8: return launch_iter_t<0, 2, 8>(descs, nframes, g, weight, evidence, last, st);
        default: break;
        }
    }
    if (channel_class == 16) return launch_iter_t<0, 1, 16>(descs, nframes, g, weight, evidence, last, st);
    return hipErrorInvalidValue;
}

}  // namespace fs
