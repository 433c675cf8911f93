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
#include "k_crf_common.h"

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

__global__ __launch_bounds__(256) void crf_tile_active_kernel(const int32_t *__restrict__ patches, int H, int W, int tw, int th,
                                                              int grow, int tiles_x, int tiles_y, int total_tiles,
                                                              int32_t *__restrict__ recs, int32_t *__restrict__ max_active,
                                                              unsigned *__restrict__ clear, int clear_words)
{
    __shared__ unsigned mask[MASK_WORDS];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < clear_words; i += gridDim.x * 256) clear[i] = 0u;
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
            // the slots that fill up the last group of four get a label no pixel carries: what the record held behind its
            // labels (another tile's, another call's) must not pass for a pixel's own label where the groups are walked
            for (int pad = ce; pad < ((ce + 3) & ~3); ++pad) labs[pad] = -2;
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

// Second half of a wide update (tiles with more channels than the update kernels keep in registers): the passes of
// crf_iter_roll_kernel<..., WIDE> have left the raw messages of every channel slot of every tile in d.msg; this kernel
// walks a tile's channel list per pixel and does what finish_pair_compact does on registers, in the same order: total
// over all labels (compensated), Potts compatibility, softmax, Q[depth == 0] override, compact planes or the argmax.
// One block per 32 x 16 tile, two pixels per lane one after the other.
__global__ __launch_bounds__(256) void crf_wide_finish_kernel(const CrfFrameDesc *__restrict__ descs, CrfGeom g, float weight,
                                                             float evidence, int last)
{
    __shared__ int s_list[TILE_REC + 8];
    const CrfFrameDesc d = descs[blockIdx.z];
    const int S = d.S, Wi = g.Wi, Hi = g.Hi;
    const unsigned Ni = (unsigned)Hi * (unsigned)Wi;
    const int tiles_x = (Wi + FTW - 1) / FTW;
    build_channel_list<256>(s_list, d.tile_recs, blockIdx.y * tiles_x + blockIdx.x, S);
    const int CE = s_list[0], n_rest = s_list[1], rep_pos = s_list[2];
    const float pv = (float)(1.0 / (double)(S - 1));
    for (int i = threadIdx.x; i < FTW * 16; i += 256) {
        const int gx = blockIdx.x * FTW + (i & (FTW - 1)), gy = blockIdx.y * 16 + i / FTW;
        if (gx >= Wi || gy >= Hi) continue;
        const unsigned p = (unsigned)gy * (unsigned)Wi + (unsigned)gx;
        const bool valid = global_ptr(d.depth_raw)[p] != 0;
        const int own = global_ptr(d.patches)[p];
        const int key = own != 0 ? own : -1;
        const gfloat *M = global_ptr(d.msg);
        auto message = [&](int c) { return M[(unsigned)s_list[REC_LAB + c] * Ni + p] + (c == CE - 1 ? evidence : 0.0f); };
        auto mult = [&](int c) { return c == rep_pos ? (float)n_rest : 1.0f; };
        float tot = 0.0f, tot_lo = 0.0f, mx = -INFINITY, den = 0.0f;
        if (valid) {
            for (int c = 0; c < CE; ++c) {
                const float term = message(c) * mult(c);
                const float t = tot + term, bp = t - tot;
                tot_lo += (tot - (t - bp)) + (term - bp);
                tot = t;
            }
        }
        auto argument = [&](int c) {
            const int lab = s_list[REC_LAB + c];
            const float comp = (tot - message(c)) + tot_lo;                           // find_surfaces.py:297-298, 360
            const float un = lab == key ? 0.0f : (c == 0 ? 10.0f : 1.0f);             // :459-472
            return -(un + weight * comp);                                             // :361
        };
        if (valid) {
            for (int c = 0; c < CE; ++c) mx = fmaxf(mx, argument(c));
            for (int c = 0; c < CE; ++c) den = fmaf(__builtin_amdgcn_exp2f((argument(c) - mx) * 1.44269504088896341f), mult(c), den);
        }
        if (last) {
            int best = 0;
            float bestv = -1.0f;
            if (valid) {
                for (int c = 0; c < CE; ++c) {
                    const int lab = s_list[REC_LAB + c];
                    const float v = __builtin_amdgcn_exp2f((argument(c) - mx) * 1.44269504088896341f) / den;
                    if (v > bestv || (v == bestv && lab < best)) { bestv = v; best = lab; }    // first maximum, :531
                }
            }
            global_ptr(d.surfaces)[p] = best == S - 1 ? 0 : best;                     // :532
        } else {
            const float inv_den = 1.0f / den;
            for (int c = 0; c < CE; ++c) {
                const int lab = s_list[REC_LAB + c];
                float q = lab == 0 ? 0.0f : pv;                                       // Q[depth == 0] = prob, :520-525
                if (valid) q = __builtin_amdgcn_exp2f((argument(c) - mx) * 1.44269504088896341f) * inv_den;
                global_ptr(d.Qout)[(unsigned)lab * Ni + p] = q;
            }
        }
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

// int64 label images -> int32 working copies; stats[frame] = largest label of the frame, stats[nframes + frame] = 1 if
// a label lies outside [0, max_label] (such pixels become 0).  stats must be zeroed.
__global__ __launch_bounds__(256) void crf_labels_to_int32_kernel(const int64_t *__restrict__ labels, int32_t *__restrict__ out,
                                                                  int32_t *__restrict__ stats, int N, int nframes, int max_label)
{
    __shared__ int s_max, s_bad;
    if (threadIdx.x == 0) { s_max = 0; s_bad = 0; }
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x, frame = blockIdx.y;
    int v = 0;
    bool bad = false;
    if (p < N) {
        const int64_t l = labels[(size_t)frame * N + p];
        bad = l < 0 || l > max_label;
        v = bad ? 0 : (int)l;
        out[(size_t)frame * N + p] = v;
    }
    // one LDS atomic per wave, one global atomic per block
    int m = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(&s_max, m);
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) s_bad = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_max > 0) atomicMax(stats + frame, s_max);
        if (s_bad) stats[nframes + frame] = 1;
    }
}

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
    if (nch <= 16) return 16;
    return kCrfWideClass;         // more channels around a tile than a lane's registers hold: passes through memory
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
                            int32_t *max_active, hipStream_t st, unsigned *clear, int clear_words)
{
    const int tx = (W + tile_w - 1) / tile_w, ty = (H + tile_h - 1) / tile_h;
    const int total = tx * ty * n;
    hipLaunchKernelGGL(crf_tile_active_kernel, dim3((total + 7) / 8 * 8), dim3(256), 0, st, patches, H, W, tile_w, tile_h, grow, tx, ty,
                       total, recs, max_active, clear, clear ? clear_words : 0);
}

// word 0 (number of listed labels) of every tile record, gathered into a dense array
__global__ __launch_bounds__(256) void crf_tile_counts_kernel(const int32_t *__restrict__ recs, int32_t *__restrict__ out, int total)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < total) out[t] = recs[(size_t)t * TILE_REC];
}
void launch_crf_tile_counts(const int32_t *recs, int32_t *out, int total, hipStream_t st)
{
    hipLaunchKernelGGL(crf_tile_counts_kernel, dim3((total + 255) / 256), dim3(256), 0, st, recs, out, total);
}

// tile height of the fused kernels (tile width 32); 0 if the window is too large for them
int crf_fused_tile_height(int r) { return r >= 1 && r <= kMaxRadius ? 16 : 0; }
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
float crf_colour_feature_scale(const CrfParams &P) { return P.mode == 0 ? (float)std::sqrt(fused_c2(P)) : 1.0f; }

void launch_crf_labels_to_int32(const int64_t *labels, int32_t *out, int32_t *stats, int n, int N, int max_label, hipStream_t st)
{
    hipLaunchKernelGGL(crf_labels_to_int32_kernel, dim3((N + 255) / 256, n), dim3(256), 0, st, labels, out, stats, N, n, max_label);
}

void launch_crf_pack_features(const uint16_t *depth, const uint8_t *rgb, const float *angles, void *featA, void *featB,
                              size_t total, const CrfParams &P, hipStream_t st)
{
    const float cs = crf_colour_feature_scale(P), as = crf_angle_feature_scale(P);
    hipLaunchKernelGGL(crf_pack_features_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, depth, rgb, angles,
                       reinterpret_cast<float4 *>(featA), reinterpret_cast<float2 *>(featB), total, cs, as);
}

// The tiles of a launch that list a_min .. a_max labels, sorted by the number of walks of the window they take (walk_channels
// channels per walk): order[0 .. 3] counts of the tiles with four (or more), three, two walks and one walk, then the four
// lists, `total` entries apart.  Within a list the order is that of the atomic counter, which no result depends on.
__global__ __launch_bounds__(256) void crf_tile_order_kernel(const CrfFrameDesc *__restrict__ descs, int tiles_per_frame, int total, int a_min,
                                                             int a_max, int walk_channels, int32_t *__restrict__ order)
{
    const int t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    int bucket = -1;
    if (t < total) {
        const int bz = t / tiles_per_frame, rem = t - bz * tiles_per_frame;
        const int A = descs[bz].tile_recs[(size_t)rem * TILE_REC];
        if (A >= a_min && A <= a_max) {
            const int walks = (A + 3 + walk_channels - 1) / walk_channels;
            bucket = walks >= 4 ? 0 : 4 - walks;
        }
    }
    // one atomic per wave and list: its tiles take consecutive places
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const unsigned long long m = __ballot(bucket == b);
        if (!m) continue;
        const int leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(order + b, __popcll(m));
        base = __shfl(base, leader);
        if (bucket == b) order[4 + (size_t)b * total + base + __popcll(m & ((1ull << lane) - 1ull))] = t;
    }
}
size_t crf_tile_order_ints(size_t tiles) { return 4 + 4 * tiles; }

// the window's constants are uniform when the position terms vanish (as with the trained parameters)
// and that variant also wants weights with a logarithm (non-negative) and no bias, as the loader of the
// trained parameters guarantees (find_surfaces.py:221-226); anything else takes the table-driven kernels
static bool crf_uniform(const CrfParams &P)
{
    return P.mode == 0 && P.t11 == 0.0f && P.t21 == 0.0f && P.t31 == 0.0f && P.w1 >= 0.0f && P.w2 >= 0.0f && P.w3 >= 0.0f && P.bias == 0.0f;
}

// A tile that lists A labels carries at most A + 3 channels (label 0, the representative of the unlisted labels, the
// "unassigned" label).  Frames whose fullest tile needs a variant with more than 16 channels: their other tiles (most of
// them, as a rule) go through the variant that is fastest for them, in a launch of its own; every launch walks all tiles
// of its frames and leaves alone those that list too many or too few labels for it.
//   r = 7, 10, 15 with the trained parameters: tiles of 9 .. 64 channels (48 at r = 15) take the kernels that evaluate the
//          similarity once per window position for 16 (12) channels (k_crf_roll_m.hip), the rest the 8-channels-per-walk
//          kernels.  1280 x 960 frames with 420 surfaces close together: 449 -> 293 us per update at r = 7, 677 -> 467 at
//          r = 10, 1 683 -> 1 176 at r = 15.
//   r = 7, other parameter sets: the static-tile kernel's 32-channel variant spills, so tiles of up to 16 channels go
//          through the 16-channel variant.
int crf_update_plan(const CrfParams &P, int r, int cls, int max_listed, bool trace, CrfSplit (&out)[kCrfMaxSplits])
{
    constexpr int all = 1 << 30;
    out[0] = CrfSplit{cls, 0, all};
    if (trace || cls <= 2 || cls == kCrfWideClass) return 1;
    if (r == 7 && !crf_uniform(P)) {
        out[0] = CrfSplit{cls, 14, all};
        out[1] = CrfSplit{2, 0, 13};
        return 2;
    }
    if ((r == 7 || r == 10 || r == 15) && crf_uniform(P)) {
        int n = 0;
        const int quads = r == 15 ? 3 : 4, most = 4 * 4 * quads - 3;               // four walks of 4 * quads channels
        if (max_listed > most) out[n++] = CrfSplit{cls, most + 1, all};
        out[n++] = CrfSplit{kCrfMultiClass | quads, 6, most};
        out[n++] = CrfSplit{2, 0, 5};
        return n;
    }
    return 1;
}

hipError_t launch_crf_iter_fused(const CrfFrameDesc *descs, int nframes, int channel_class, const CrfGeom &g,
                                 const CrfParams &P, const float *wtab, int mode, unsigned *tile_queue, hipStream_t st, bool trace,
                                 int wide_channels, int a_min, int a_max, int32_t *tile_order)
{
    if (g.r < 1 || g.r > kMaxRadius || nframes < 1) return hipErrorInvalidValue;
    const int last = (mode & CRF_LAST) ? 1 : 0, first = (mode & CRF_FIRST) ? 1 : 0;
    const double l2e = 1.4426950408889634;
    // delta_d = 12 t with t = (dq-dp)/(dp+dq); colour differences scaled by 1/50, angles by 2 (find_surfaces.py:20, 343-349)
    FusedConsts K;
    K.K1 = (float)(144.0 * P.t12 * l2e); K.K2 = (float)(144.0 * P.t22 * l2e); K.K3 = (float)(144.0 * P.t32 * l2e);
    K.evidence = P.evidence;
    const bool uniform = crf_uniform(P);
    K.uw[0] = (float)-std::log2((double)P.w1); K.uw[1] = (float)-std::log2((double)P.w2); K.uw[2] = (float)-std::log2((double)P.w3);
    K.uw[3] = 0.0f;
    const float4 *wt = reinterpret_cast<const float4 *>(wtab);
    FusedLaunch L{descs, nframes, g, K, wt, P.weight, first, last, tile_queue, st, trace};
    L.a_min = a_min;
    L.a_max = a_max;
    if (channel_class == kCrfWideClass) {
        // one pass per window of kMaxTileChannels channel slots, messages to descs[].msg, then the finishing kernel
        if (trace || wide_channels < 1) return hipErrorInvalidValue;
        const int variant = P.mode == 1 ? 2 : (uniform ? 0 : 1);
        for (int c0 = 0; c0 < wide_channels; c0 += kMaxTileChannels) {
            if (c0 > 0) {                       // (the caller hands over a zeroed tile queue: good for the first pass)
                if (hipError_t e = hipMemsetAsync(tile_queue, 0, kCrfQueueBytes, st); e != hipSuccess) return e;
            }
            L.c0 = c0;
            if (hipError_t e = launch_roll_wide(variant, L); e != hipSuccess) return e;
        }
        int tx, ty;
        crf_fused_tile_grid(g.Hi, g.Wi, g.r, &tx, &ty);
        hipLaunchKernelGGL(crf_wide_finish_kernel, dim3(tx, ty, nframes), dim3(256), 0, st, descs, g, P.weight, P.evidence, last);
        return hipGetLastError();
    }
    if (trace) return uniform ? launch_fused_trace(channel_class, L) : hipErrorInvalidValue;
    if (channel_class & kCrfMultiClass) {
        if (!uniform) return hipErrorInvalidValue;
        const int quads = channel_class & 0xff;
        if (tile_order) {
            int tx, ty;
            crf_fused_tile_grid(g.Hi, g.Wi, g.r, &tx, &ty);
            const int total = tx * ty * nframes;
            if (hipError_t e = hipMemsetAsync(tile_order, 0, 4 * sizeof(int32_t), st); e != hipSuccess) return e;
            hipLaunchKernelGGL(crf_tile_order_kernel, dim3((total + 255) / 256), dim3(256), 0, st, descs, tx * ty, total, a_min, a_max, 4 * quads, tile_order);
            L.order = tile_order;
        }
        return launch_roll_multi(quads, L);
    }
    if (P.mode == 1) return launch_roll_table(2, channel_class, L);
    if (!uniform) return launch_roll_table(1, channel_class, L);
    // r = 7 keeps the tile with its whole halo in LDS (no step barriers, no half-idle first and last window row); a
    // rolling r = 7 instance at three blocks per CU measured the same 44.7 us per 640x480 update, so the older kernel
    // stays.  Every other radius rolls the window through a 16-row ring.
    if (g.r == 7) return launch_fused_r7_uniform(channel_class, L);
    return launch_roll_uniform(channel_class, L);
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
