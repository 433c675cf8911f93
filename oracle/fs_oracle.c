/*
 * fs_oracle.c -- CPU restatement of the reference's find_surfaces path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may link, import or
 * call this file: it is the checker used by tests/, by __graft_entry__.smoke()
 * and by the cpu_baseline leg of bench.py.
 *
 * PARITY UNPINNED.  The reference implementation (Python on TensorFlow 2.6 +
 * tensorflow-addons 0.14 + numba) cannot be imported in this environment and
 * ships no tests, golden vectors or saved outputs for this path.  This file
 * restates the algorithm from the reference's source text, one fp32 operation
 * per TensorFlow op, in the order written there (compile with
 * -ffp-contract=off so no operation is fused).  Three things are assumptions
 * rather than facts, because the kernels that would settle them live in
 * TensorFlow / TensorFlow-addons and are not in /root/reference:
 *   (1) reductions over short axes (3-, 6-, 5-, 49-, 226-term reduce_sum) are
 *       evaluated left to right;
 *   (2) tan/acos/exp are the host libm float functions;
 *   (3) tfa.image.connected_components is 4-connected, labels components
 *       1..n in row-major order of their first pixel and keeps 0 as
 *       background (its documented behaviour; cross-checked in the tests
 *       against scipy.ndimage.label, which documents the same convention).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference repository root).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FSO_API __attribute__((visibility("default")))

/* tf.math.divide_no_nan: 0 where the denominator is 0. */
static inline float div_no_nan(float a, float b) { return b == 0.0f ? 0.0f : a / b; }
static inline float clipf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
static inline float signf(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

/* ------------------------------------------------------------------------- *
 * get_angle_arrays, image_processing_models_GPU.py:137-149.
 * tanx[c] = tan(x_val * (c / (W/2 - 0.5) - 1)),  tany[r] uses the reversed
 * range H-1 .. 0.  viewing angles are config.py:6-7 (passed in radians).
 * ------------------------------------------------------------------------- */
FSO_API void fso_angle_tables(int H, int W, double view_x, double view_y, float *tanx, float *tany)
{
    const float x_val = (float)(view_x / 2.0);
    const float y_val = (float)(view_y / 2.0);
    const float dx = (float)(W / 2.0 - 0.5);
    const float dy = (float)(H / 2.0 - 0.5);
    for (int c = 0; c < W; ++c) {
        float v = ((float)c / dx) - 1.0f;
        tanx[c] = tanf(x_val * v);
    }
    for (int r = 0; r < H; ++r) {
        float v = ((float)(H - 1 - r) / dy) - 1.0f;
        tany[r] = tanf(y_val * v);
    }
}

/* ------------------------------------------------------------------------- *
 * gaussian_filter_with_depth_factor_model_GPU(size=3, sigma_1=0, sigma_2=0),
 * image_processing_models_GPU.py:110-135.  With both sigmas 0 the Gaussian
 * weights are exp(-0-0)=1, so the filter is the mean of the valid
 * (depth >= 0.001) pixels of the (2*size+1)^2 SAME-zero-padded window, times
 * the validity of the centre.
 * ------------------------------------------------------------------------- */
FSO_API void fso_smooth_depth(const float *depth, int H, int W, int size, float *out)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float sums = 0.0f, val = 0.0f;
            for (int wy = -size; wy <= size; ++wy)
                for (int wx = -size; wx <= size; ++wx) {
                    int yy = y + wy, xx = x + wx;
                    float d = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? depth[yy * W + xx] : 0.0f;
                    float ind = d < 0.001f ? 0.0f : 1.0f;
                    sums = sums + ind;
                    val = val + ind * d;
                }
            float c = depth[y * W + x];
            out[y * W + x] = div_no_nan(val, sums) * (c < 0.001f ? 0.0f : 1.0f);
        }
}

/* angle_calculator, image_processing_models_GPU.py:63-72 */
static void angle_calculator(const float n[3], float a[2])
{
    float v1[3] = {n[0] * 0.0f, n[1] * 1.0f, n[2] * 1.0f};
    float v2[3] = {n[0] * 1.0f, n[1] * 0.0f, n[2] * 1.0f};
    float n1 = sqrtf((v1[0] * v1[0] + v1[1] * v1[1]) + v1[2] * v1[2]);
    float n2 = sqrtf((v2[0] * v2[0] + v2[1] * v2[1]) + v2[2] * v2[2]);
    a[0] = acosf(clipf(div_no_nan(-v1[2], n1), -1.0f, 1.0f)) * signf(v1[1]);
    a[1] = acosf(clipf(div_no_nan(-v2[2], n2), -1.0f, 1.0f)) * (-signf(v2[0]));
}

/* ------------------------------------------------------------------------- *
 * normals_model_GPU, image_processing_models_GPU.py:151-178 (up to the raw
 * angles): back-projection, central differences with zero outside the image,
 * cross product, sign flip where z > 0, normalisation, two angles.
 * points/normals are [H,W,3], raw_angles is [H,W,2].
 * ------------------------------------------------------------------------- */
FSO_API void fso_normals(const float *D, int H, int W, const float *tanx, const float *tany,
                         float *points, float *normals, float *raw_angles)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float d = D[y * W + x];
            float *p = points + 3 * (y * W + x);
            p[0] = tanx[x] * d;
            p[1] = tany[y] * d;
            p[2] = d;
        }
    static const float zero3[3] = {0.0f, 0.0f, 0.0f};
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float *up = y > 0 ? points + 3 * ((y - 1) * W + x) : zero3;
            const float *dn = y < H - 1 ? points + 3 * ((y + 1) * W + x) : zero3;
            const float *lf = x > 0 ? points + 3 * (y * W + x - 1) : zero3;
            const float *rt = x < W - 1 ? points + 3 * (y * W + x + 1) : zero3;
            float a[3], b[3], v[3];
            for (int i = 0; i < 3; ++i) {
                a[i] = up[i] - dn[i]; /* sub_1, :163-168 */
                b[i] = rt[i] - lf[i]; /* sub_2, :165-169 */
            }
            v[0] = a[1] * b[2] - a[2] * b[1];
            v[1] = a[2] * b[0] - a[0] * b[2];
            v[2] = a[0] * b[1] - a[1] * b[0];
            float cond = -(v[2] > 0.0f ? 1.0f : 0.0f); /* :172 */
            for (int i = 0; i < 3; ++i)
                v[i] = 2.0f * (cond * v[i]) + v[i];    /* :173 */
            float nrm = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
            float *n = normals + 3 * (y * W + x);
            for (int i = 0; i < 3; ++i)
                n[i] = div_no_nan(v[i], nrm);
            angle_calculator(n, raw_angles + 2 * (y * W + x));
        }
}

/* ------------------------------------------------------------------------- *
 * uniform_filter_with_depth_cutoff + Depth_Cutoff,
 * image_processing_models_GPU.py:102-108, 86-88.  SAME zero padding for both
 * the angle and the depth windows; weight 0 where |(Dj-Dc)/Dc| > 0.3
 * (divide_no_nan, so a zero centre keeps all (2*pool+1)^2 weights at 1).
 * ------------------------------------------------------------------------- */
FSO_API void fso_angle_filter(const float *raw_angles, const float *D, int H, int W, int pool, float *angles)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float dc = D[y * W + x];
            float sums = 0.0f, v0 = 0.0f, v1 = 0.0f;
            for (int wy = -pool; wy <= pool; ++wy)
                for (int wx = -pool; wx <= pool; ++wx) {
                    int yy = y + wy, xx = x + wx;
                    int in = (yy >= 0 && yy < H && xx >= 0 && xx < W);
                    float dj = in ? D[yy * W + xx] : 0.0f;
                    float a0 = in ? raw_angles[2 * (yy * W + xx)] : 0.0f;
                    float a1 = in ? raw_angles[2 * (yy * W + xx) + 1] : 0.0f;
                    float m = fabsf(div_no_nan(dj - dc, dc)) > 0.3f ? 0.0f : 1.0f;
                    sums = sums + m;
                    v0 = v0 + m * a0;
                    v1 = v1 + m * a1;
                }
            angles[2 * (y * W + x)] = div_no_nan(v0, sums);
            angles[2 * (y * W + x) + 1] = div_no_nan(v1, sums);
        }
}

/* ------------------------------------------------------------------------- *
 * Curvature score and threshold, find_surfaces.py:232-262.
 * scores (optional) and mask are full [H,W] frames; the `depth`-pixel border
 * is zero (tf.pad at :262).
 * ------------------------------------------------------------------------- */
FSO_API void fso_curvature_mask(const float *D, int H, int W, const float *tanx, const float *tany,
                                int depth, float threshold, int32_t *mask, float *scores)
{
    static const int dirs[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
    memset(mask, 0, sizeof(int32_t) * (size_t)H * W);
    if (scores) memset(scores, 0, sizeof(float) * (size_t)H * W);
    for (int r = depth; r < H - depth; ++r)
        for (int c = depth; c < W - depth; ++c) {
            float dc = D[r * W + c];
            float dfx = (fabsf(tanx[c - 1] - tanx[c + 1]) * dc) * 0.5f; /* :237 */
            float dfy = (fabsf(tany[r - 1] - tany[r + 1]) * dc) * 0.5f; /* :238 */
            float total = 0.0f;
            for (int d = 0; d < 8; ++d) {
                float cur = 0.0f, prevd = 0.0f;
                float px = 0.0f, py = 0.0f, pz = dc;
                for (int k = 1; k <= depth; ++k) {
                    float nx = dfx * (float)k, ny = dfy * (float)k;
                    float nz = D[(r - k * dirs[d][0]) * W + (c - k * dirs[d][1])]; /* :252 */
                    float e0 = nx - px, e1 = ny - py, e2 = nz - pz;
                    float dif1 = sqrtf((e0 * e0 + e1 * e1) + e2 * e2);
                    float g0 = nx - 0.0f, g1 = ny - 0.0f, g2 = nz - dc;
                    float dif2 = sqrtf((g0 * g0 + g1 * g1) + g2 * g2);
                    float score = div_no_nan(dif1 + prevd, dif2) - 1.0f;
                    cur = cur + score;
                    total = total + cur;
                    prevd = dif2;
                    px = nx; py = ny; pz = nz;
                }
            }
            if (scores) scores[r * W + c] = total;
            int32_t pix = total < threshold ? 1 : 0;
            mask[r * W + c] = dc > 0.001f ? pix : 0;
        }
}

/* Majority filter: SAME-padded (2s+1)^2 box sum > n, find_surfaces.py:265-266 */
FSO_API void fso_majority(const int32_t *in, int H, int W, int s, int n, int32_t *out)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int sum = 0;
            for (int wy = -s; wy <= s; ++wy)
                for (int wx = -s; wx <= s; ++wx) {
                    int yy = y + wy, xx = x + wx;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) sum += in[yy * W + xx];
                }
            out[y * W + x] = sum > n ? 1 : 0;
        }
}

/* Depth edges, find_surfaces.py:269-278: 1 where all four neighbours are
 * within alpha * lateral pixel size of the centre, 0 on the 1-pixel border. */
FSO_API void fso_depth_edges(const float *D, int H, int W, const float *tanx, const float *tany,
                             float alpha, int32_t *edges)
{
    memset(edges, 0, sizeof(int32_t) * (size_t)H * W);
    for (int r = 1; r < H - 1; ++r)
        for (int c = 1; c < W - 1; ++c) {
            float dc = D[r * W + c];
            float fx = ((fabsf(tanx[c - 1] - tanx[c + 1]) * dc) * 0.5f) * alpha;
            float fy = ((fabsf(tany[r - 1] - tany[r + 1]) * dc) * 0.5f) * alpha;
            int e = 1;
            if (fabsf(D[r * W + c + 1] - dc) > fx) e = 0;
            if (fabsf(D[r * W + c - 1] - dc) > fx) e = 0;
            if (fabsf(D[(r - 1) * W + c] - dc) > fy) e = 0;
            if (fabsf(D[(r + 1) * W + c] - dc) > fy) e = 0;
            edges[r * W + c] = e;
        }
}

/* ------------------------------------------------------------------------- *
 * Components, image_processing_models_GPU.py:74-76
 * (tfa.image.connected_components, TensorFlow-addons 0.14.0, not vendored):
 * 4-connectivity between equal non-zero values, ids 1..n in row-major order
 * of each component's first pixel, zeros stay 0, int64 output.
 * Classic two-pass union-find.  Returns n.
 * ------------------------------------------------------------------------- */
static int uf_find(int *parent, int a)
{
    while (parent[a] != a) {
        parent[a] = parent[parent[a]];
        a = parent[a];
    }
    return a;
}

FSO_API int fso_connected_components(const int32_t *img, int H, int W, int64_t *out)
{
    int N = H * W;
    int *parent = (int *)malloc(sizeof(int) * (size_t)N);
    for (int i = 0; i < N; ++i) parent[i] = i;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int i = y * W + x;
            if (img[i] == 0) continue;
            if (x > 0 && img[i - 1] == img[i]) {
                int a = uf_find(parent, i), b = uf_find(parent, i - 1);
                if (a != b) parent[a > b ? a : b] = a > b ? b : a;
            }
            if (y > 0 && img[i - W] == img[i]) {
                int a = uf_find(parent, i), b = uf_find(parent, i - W);
                if (a != b) parent[a > b ? a : b] = a > b ? b : a;
            }
        }
    int *ids = (int *)calloc((size_t)N, sizeof(int));
    int n = 0;
    for (int i = 0; i < N; ++i) {
        if (img[i] == 0) { out[i] = 0; continue; }
        int r = uf_find(parent, i); /* smallest index of the component */
        if (ids[r] == 0) ids[r] = ++n;
        out[i] = ids[r];
    }
    free(ids);
    free(parent);
    return n;
}

/* ------------------------------------------------------------------------- *
 * find_surfaces_model_GPU, find_surfaces.py:229-288 (whole graph).
 * D is the *smoothed* depth.  Returns max(patches).
 * ------------------------------------------------------------------------- */
FSO_API int fso_patches(const float *D, int H, int W, const float *tanx, const float *tany,
                        int depth, float threshold, float alpha, int s1, int s2, int n1, int n2,
                        int component_threshold, int64_t *patches, int32_t *edges)
{
    size_t N = (size_t)H * W;
    int32_t *m0 = (int32_t *)malloc(sizeof(int32_t) * N);
    int32_t *m1 = (int32_t *)malloc(sizeof(int32_t) * N);
    int64_t *comp = (int64_t *)malloc(sizeof(int64_t) * N);
    fso_curvature_mask(D, H, W, tanx, tany, depth, threshold, m0, NULL);
    fso_majority(m0, H, W, s1, n1, m1);
    fso_majority(m1, H, W, s2, n2, m0);
    fso_depth_edges(D, H, W, tanx, tany, alpha, edges);
    for (size_t i = 0; i < N; ++i) m0[i] = edges[i] * m0[i];            /* :279 */
    int n = fso_connected_components(m0, H, W, comp);                     /* :281 */
    int64_t *cnt = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));     /* :282 bincount */
    for (size_t i = 0; i < N; ++i) cnt[comp[i]]++;
    for (size_t i = 0; i < N; ++i)                                        /* :284 */
        m0[i] = m0[i] * (cnt[comp[i]] > component_threshold ? 1 : 0);
    n = fso_connected_components(m0, H, W, patches);                      /* :285 */
    free(cnt); free(comp); free(m1); free(m0);
    return n;
}

/* ------------------------------------------------------------------------- *
 * get_unary_potentials_and_initial_probabilities, find_surfaces.py:456-477.
 * The reference builds float64 arrays that Keras casts to float32 on entry
 * to predict(); values are computed here in double and cast the same way.
 * S = num_surfaces + 1 channels.  prob is [S].
 * ------------------------------------------------------------------------- */
FSO_API void fso_unary_q0(const int64_t *patches, int H, int W, int num_surfaces,
                          float *unary, float *q0, float *prob)
{
    int S = num_surfaces + 1;
    double pv = 1.0 / (double)num_surfaces;
    for (int l = 0; l < S; ++l) prob[l] = (float)(l == 0 ? 0.0 : pv);
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        float *u = unary + i * S, *q = q0 + i * S;
        int64_t p = patches[i];
        if (p == 0) {
            for (int l = 0; l < S; ++l) { u[l] = l == 0 ? 10.0f : 1.0f; q[l] = prob[l]; }
        } else {
            double other = 1.0 / (double)(num_surfaces - 1) * 0.1;       /* :473 */
            for (int l = 0; l < S; ++l) { u[l] = 1.0f; q[l] = (float)other; }
            u[p] = 0.0f; u[0] = 10.0f;
            q[p] = (float)0.9; q[0] = 0.0f;
        }
    }
}

/* ------------------------------------------------------------------------- *
 * conv_crf_Gauss forward pass, find_surfaces.py:291-367: one mean-field
 * update on a batch of B tiles.
 *   unary [B,h,w,S]; Q [B,h+2r,w+2r,S]; f1 [B,h+2r,w+2r,3] = (y,x,depth);
 *   f2 [..,6] = (y,x,depth,c0,c1,c2); f3 [..,5] = (y,x,depth,a0,a1);
 *   val [S] (row of the reference's [1,1,1,1,S] input; the same for every
 *   tile); params = the 13 values returned by load_Gauss_parameters
 *   (find_surfaces.py:221-226): w1,w2,w3,bias,t11,t21,t31,t12,t22,t32,t23,
 *   t33,weight.
 *   out [B,h,w,S]; potentials (optional, may be NULL) [B,h,w,S] receives
 *   unary + weight * compatibility (the argument of the softmax, :361).
 * ------------------------------------------------------------------------- */
FSO_API void fso_crf_step(const float *unary, const float *Q, const float *f1, const float *f2,
                          const float *f3, const float *val, const float *params, int r, int h, int w,
                          int S, int B, float *out, float *potentials)
{
    const int k = 2 * r + 1, hp = h + 2 * r, wp = w + 2 * r;
    const float w1 = params[0], w2 = params[1], w3 = params[2], bias = params[3];
    const float t11 = params[4] * params[4], t21 = params[5] * params[5], t31 = params[6] * params[6];
    const float t12 = params[7] * params[7], t22 = params[8] * params[8], t32 = params[9] * params[9];
    const float t23 = params[10] * params[10], t33 = params[11] * params[11];
    const float weight = params[12];
    /* similarity_normalizers = [6, 2, 1/50, 1/40], find_surfaces.py:20 */
    const float nd = 6.0f, na = 2.0f, nc = (float)(1.0 / 50.0), np_ = (float)(1.0 / 40.0);
#pragma omp parallel
    {
        float *msg = (float *)malloc(sizeof(float) * (size_t)S);
        float *add = (float *)malloc(sizeof(float) * (size_t)S);
#pragma omp for collapse(2) schedule(dynamic, 4)
        for (int b = 0; b < B; ++b)
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const size_t base = (size_t)b * hp * wp;
                    const size_t ci = base + (size_t)(y + r) * wp + (x + r);
                    const float *c1 = f1 + ci * 3, *c2 = f2 + ci * 6, *c3 = f3 + ci * 5;
                    for (int l = 0; l < S; ++l) msg[l] = 0.0f;
                    for (int wy = 0; wy < k; ++wy)
                        for (int wx = 0; wx < k; ++wx) {
                            const size_t ji = base + (size_t)(y + wy) * wp + (x + wx);
                            const float *j1 = f1 + ji * 3, *j2 = f2 + ji * 6, *j3 = f3 + ji * 5;
                            /* :343-344; exponent mask is 1 on the depth channel only, pow(x,0)=1 */
                            float d0 = div_no_nan(j1[0] - c1[0], (1.0f + 1.0f) / 2.0f) * np_;
                            float d1 = div_no_nan(j1[1] - c1[1], (1.0f + 1.0f) / 2.0f) * np_;
                            float d2 = div_no_nan(j1[2] - c1[2], (c1[2] + j1[2]) / 2.0f) * nd;
                            float s1 = expf(-(((d0 * d0) * t11 + (d1 * d1) * t11) + (d2 * d2) * t12));
                            /* :346-347 */
                            float e0 = div_no_nan(j2[0] - c2[0], (1.0f + 1.0f) / 2.0f) * np_;
                            float e1 = div_no_nan(j2[1] - c2[1], (1.0f + 1.0f) / 2.0f) * np_;
                            float e2 = div_no_nan(j2[2] - c2[2], (c2[2] + j2[2]) / 2.0f) * nd;
                            float e3 = div_no_nan(j2[3] - c2[3], (1.0f + 1.0f) / 2.0f) * nc;
                            float e4 = div_no_nan(j2[4] - c2[4], (1.0f + 1.0f) / 2.0f) * nc;
                            float e5 = div_no_nan(j2[5] - c2[5], (1.0f + 1.0f) / 2.0f) * nc;
                            float s2 = expf(-((((((e0 * e0) * t21 + (e1 * e1) * t21) + (e2 * e2) * t22) +
                                                (e3 * e3) * t23) + (e4 * e4) * t23) + (e5 * e5) * t23));
                            /* :349-350 */
                            float g0 = div_no_nan(j3[0] - c3[0], (1.0f + 1.0f) / 2.0f) * np_;
                            float g1 = div_no_nan(j3[1] - c3[1], (1.0f + 1.0f) / 2.0f) * np_;
                            float g2 = div_no_nan(j3[2] - c3[2], (c3[2] + j3[2]) / 2.0f) * nd;
                            float g3 = div_no_nan(j3[3] - c3[3], (1.0f + 1.0f) / 2.0f) * na;
                            float g4 = div_no_nan(j3[4] - c3[4], (1.0f + 1.0f) / 2.0f) * na;
                            float s3 = expf(-(((((g0 * g0) * t31 + (g1 * g1) * t31) + (g2 * g2) * t32) +
                                               (g3 * g3) * t33) + (g4 * g4) * t33));
                            /* :352-355 */
                            float sim = ((w1 * s1 + w2 * s2) + w3 * s3) + bias;
                            /* :313 centre masked, :358 */
                            float qm = (wy == r && wx == r) ? 0.0f : 1.0f;
                            const float *q = Q + ji * S;
                            for (int l = 0; l < S; ++l) msg[l] = msg[l] + sim * (q[l] * qm);
                        }
                    for (int l = 0; l < S; ++l) msg[l] = msg[l] + 0.00005f * val[l]; /* :356-358 */
                    const size_t oi = ((size_t)b * h + y) * w + x;
                    float mx = -INFINITY;
                    for (int l = 0; l < S; ++l) {
                        float comp = 0.0f; /* :297-298, :360 Potts matrix 1 - I */
                        for (int m = 0; m < S; ++m) comp = comp + (m == l ? 0.0f : 1.0f) * msg[m];
                        add[l] = unary[oi * S + l] + weight * comp; /* :361 */
                        if (potentials) potentials[oi * S + l] = add[l];
                        mx = fmaxf(mx, -add[l]);
                    }
                    float den = 0.0f; /* :363 softmax(-add) */
                    for (int l = 0; l < S; ++l) { add[l] = expf(-add[l] - mx); den = den + add[l]; }
                    for (int l = 0; l < S; ++l) out[oi * S + l] = add[l] / den;
                }
        free(msg);
        free(add);
    }
}

/* ------------------------------------------------------------------------- *
 * Whole-frame equivalent of get_inputs -> predict -> assemble_outputs
 * (find_surfaces.py:437-454, 487-495, 517-519).  The 4x4 tiling with a
 * kernel_size halo cut from the zero-padded frame is numerically invisible,
 * so the frame is processed as one zero-padded tile.  depth_raw, rgb and
 * angles are the feature sources of extract_features (:479-484, :500-501).
 * ------------------------------------------------------------------------- */
FSO_API void fso_crf_frame(const float *unary, const float *Qin, const uint16_t *depth_raw,
                           const uint8_t *rgb, const float *angles, const float *params, int r, int H,
                           int W, int S, float *out, float *potentials)
{
    const int hp = H + 2 * r, wp = W + 2 * r;
    size_t NP = (size_t)hp * wp;
    float *Qp = (float *)calloc(NP * S, sizeof(float));
    float *f1 = (float *)calloc(NP * 3, sizeof(float));
    float *f2 = (float *)calloc(NP * 6, sizeof(float));
    float *f3 = (float *)calloc(NP * 5, sizeof(float));
    float *val = (float *)calloc((size_t)S, sizeof(float));
    val[S - 1] = 1.0f; /* :451-452 */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)y * W + x, j = (size_t)(y + r) * wp + (x + r);
            memcpy(Qp + j * S, Qin + i * S, sizeof(float) * (size_t)S);
            float d = (float)depth_raw[i];
            f1[j * 3 + 0] = (float)y; f1[j * 3 + 1] = (float)x; f1[j * 3 + 2] = d;
            f2[j * 6 + 0] = (float)y; f2[j * 6 + 1] = (float)x; f2[j * 6 + 2] = d;
            f2[j * 6 + 3] = (float)rgb[i * 3]; f2[j * 6 + 4] = (float)rgb[i * 3 + 1]; f2[j * 6 + 5] = (float)rgb[i * 3 + 2];
            f3[j * 5 + 0] = (float)y; f3[j * 5 + 1] = (float)x; f3[j * 5 + 2] = d;
            f3[j * 5 + 3] = angles[i * 2]; f3[j * 5 + 4] = angles[i * 2 + 1];
        }
    fso_crf_step(unary, Qp, f1, f2, f3, val, params, r, H, W, S, 1, out, potentials);
    free(val); free(f3); free(f2); free(f1); free(Qp);
}

/* ------------------------------------------------------------------------- *
 * find_surfaces, find_surfaces.py:497-532 for one frame (mode "Gauss").
 * Outputs follow the returned dict: angles [H,W,2], vectors [H,W,3],
 * points [H,W,3], patches int64 [H,W], edges int32 [H,W], surfaces int64
 * [H,W]; returns num_surfaces = max(patches)+1.  q_final (optional) receives
 * the last Q [H,W,num_surfaces+1] before the depth==0 reset of :530.
 * ------------------------------------------------------------------------- */
FSO_API int fso_find_surfaces(const uint16_t *depth_raw, const uint8_t *rgb, int H, int W,
                              double view_x, double view_y, const float *params, int r, int iterations,
                              float *angles, float *vectors, float *points, int64_t *patches,
                              int32_t *edges, int64_t *surfaces, float *smoothed_out, float *q_final)
{
    size_t N = (size_t)H * W;
    float *tanx = (float *)malloc(sizeof(float) * (size_t)W), *tany = (float *)malloc(sizeof(float) * (size_t)H);
    float *depth_f = (float *)malloc(sizeof(float) * N), *smoothed = (float *)malloc(sizeof(float) * N);
    float *raw_angles = (float *)malloc(sizeof(float) * N * 2);
    fso_angle_tables(H, W, view_x, view_y, tanx, tany);
    for (size_t i = 0; i < N; ++i) depth_f[i] = (float)depth_raw[i];
    fso_smooth_depth(depth_f, H, W, 3, smoothed);                                   /* :505 */
    if (smoothed_out) memcpy(smoothed_out, smoothed, sizeof(float) * N);
    fso_normals(smoothed, H, W, tanx, tany, points, vectors, raw_angles);           /* :507 */
    fso_angle_filter(raw_angles, smoothed, H, W, 3, angles);
    float thr = (float)(0.007003343675404672 * 10.5);                               /* :229 */
    int maxp = fso_patches(smoothed, H, W, tanx, tany, 4, thr, 3.0f, 2, 1, 11, 5, 20, patches, edges); /* :513 */
    int num_surfaces = maxp + 1;                                                    /* :514 */
    int S = num_surfaces + 1;
    float *unary = (float *)malloc(sizeof(float) * N * S), *Q = (float *)malloc(sizeof(float) * N * S);
    float *Q2 = (float *)malloc(sizeof(float) * N * S), *prob = (float *)malloc(sizeof(float) * (size_t)S);
    fso_unary_q0(patches, H, W, num_surfaces, unary, Q, prob);                      /* :515 */
    for (int it = 0; it < iterations; ++it) {                                       /* :517-530 */
        fso_crf_frame(unary, Q, depth_raw, rgb, angles, params, r, H, W, S, Q2, NULL);
        if (it == iterations - 1 && q_final) memcpy(q_final, Q2, sizeof(float) * N * S);
        for (size_t i = 0; i < N; ++i)
            if (depth_raw[i] == 0)
                for (int l = 0; l < S; ++l) Q2[i * S + l] = (it == iterations - 1) ? 0.0f : prob[l];
        float *t = Q; Q = Q2; Q2 = t;
    }
    for (size_t i = 0; i < N; ++i) {                                                /* :531-532 */
        int best = 0;
        for (int l = 1; l < S; ++l)
            if (Q[i * S + l] > Q[i * S + best]) best = l;
        surfaces[i] = best == num_surfaces ? 0 : best;
    }
    free(prob); free(Q2); free(Q); free(unary);
    free(raw_angles); free(smoothed); free(depth_f); free(tany); free(tanx);
    return num_surfaces;
}
