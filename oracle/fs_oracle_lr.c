/*
 * fs_oracle_lr.c -- CPU restatement of the mean-field update with the logistic-regression
 * pixel similarity, conv_crf_LR (find_surfaces.py:371-434), and of its whole-frame use in
 * find_surfaces(mode="LR") (find_surfaces.py:508-509, 517-530).
 *
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (see fs_oracle.c): one fp32 operation per TF op,
 * sums in the order reduce_sum would produce (channels left to right, window row-major,
 * the auxiliary term last).
 *
 * params[7] = coef[0..3] (depth, angle, colour, position), bias (= intercept + 4,
 * find_surfaces.py:218), weight, auxiliary_weight (LR_CRF_parameters, :22).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FSO_API __attribute__((visibility("default")))

static inline float div_no_nan(float a, float b) { return b == 0.0f ? 0.0f : a / b; }

/* inputs as conv_crf_LR takes them: unary [B,h,w,S], Q [B,h+2r,w+2r,S], f1 [..,1] depth,
 * f2 [..,2] angles, f3 [..,3] colour, f4 [..,2] (row, col), val [S] */
FSO_API void fso_crf_step_lr(const float *unary, const float *Q, const float *f1, const float *f2, const float *f3,
                             const float *f4, const float *val, const float *params, int r, int h, int w, int S, int B,
                             float *out, float *potentials)
{
    const int k = 2 * r + 1, hp = h + 2 * r, wp = w + 2 * r;
    const float c1 = params[0], c2 = params[1], c3 = params[2], c4 = params[3], bias = params[4];
    const float weight = params[5], aux = params[6];
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
                    for (int l = 0; l < S; ++l) msg[l] = 0.0f;
                    for (int wy = 0; wy < k; ++wy)
                        for (int wx = 0; wx < k; ++wx) {
                            const size_t ji = base + (size_t)(y + wy) * wp + (x + wx);
                            /* :406-416 */
                            const float d1 = div_no_nan(f1[ji] - f1[ci], f1[ci]);
                            const float s1 = sqrtf(d1 * d1) * nd;
                            const float a0 = f2[2 * ji] - f2[2 * ci], a1 = f2[2 * ji + 1] - f2[2 * ci + 1];
                            const float s2 = sqrtf(a0 * a0 + a1 * a1) * na;
                            const float e0 = f3[3 * ji] - f3[3 * ci], e1 = f3[3 * ji + 1] - f3[3 * ci + 1], e2 = f3[3 * ji + 2] - f3[3 * ci + 2];
                            const float s3 = sqrtf((e0 * e0 + e1 * e1) + e2 * e2) * nc;
                            const float p0 = f4[2 * ji] - f4[2 * ci], p1 = f4[2 * ji + 1] - f4[2 * ci + 1];
                            const float s4 = sqrtf(p0 * p0 + p1 * p1) * np_;
                            /* :421-422 */
                            const float mult = ((c1 * s1 + c2 * s2) + c3 * s3) + c4 * s4;
                            const float lr = bias + mult;
                            const float qm = (wy == r && wx == r) ? 0.0f : 1.0f;     /* :380-382, 390 */
                            const float *q = Q + ji * S;
                            for (int l = 0; l < S; ++l) msg[l] = msg[l] + lr * (q[l] * qm);
                        }
                    for (int l = 0; l < S; ++l) msg[l] = msg[l] + aux * val[l];      /* :424-426 */
                    const size_t oi = ((size_t)b * h + y) * w + x;
                    float mx = -INFINITY;
                    for (int l = 0; l < S; ++l) {
                        float comp = 0.0f;                                            /* :377-378, 428 */
                        for (int m = 0; m < S; ++m) comp = comp + (m == l ? 0.0f : 1.0f) * msg[m];
                        add[l] = unary[oi * S + l] + weight * comp;                   /* :429 */
                        if (potentials) potentials[oi * S + l] = add[l];
                        mx = fmaxf(mx, -add[l]);
                    }
                    float den = 0.0f;                                                 /* :431 */
                    for (int l = 0; l < S; ++l) { add[l] = expf(-add[l] - mx); den = den + add[l]; }
                    for (int l = 0; l < S; ++l) out[oi * S + l] = add[l] / den;
                }
        free(msg);
        free(add);
    }
}

/* whole-frame update with the LR-mode features of find_surfaces.py:508-509:
 * [raw depth, smoothed angles, raw rgb, (row, col)] zero-padded by r */
FSO_API void fso_crf_frame_lr(const float *unary, const float *Qin, const uint16_t *depth_raw, const uint8_t *rgb,
                              const float *angles, const float *params, int r, int H, int W, int S, float *out,
                              float *potentials)
{
    const int hp = H + 2 * r, wp = W + 2 * r;
    const size_t NP = (size_t)hp * wp;
    float *Qp = (float *)calloc(NP * S, sizeof(float));
    float *f1 = (float *)calloc(NP, sizeof(float)), *f2 = (float *)calloc(NP * 2, sizeof(float));
    float *f3 = (float *)calloc(NP * 3, sizeof(float)), *f4 = (float *)calloc(NP * 2, sizeof(float));
    float *val = (float *)calloc((size_t)S, sizeof(float));
    val[S - 1] = 1.0f;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x, j = (size_t)(y + r) * wp + (x + r);
            memcpy(Qp + j * S, Qin + i * S, sizeof(float) * (size_t)S);
            f1[j] = (float)depth_raw[i];
            f2[2 * j] = angles[2 * i]; f2[2 * j + 1] = angles[2 * i + 1];
            f3[3 * j] = (float)rgb[3 * i]; f3[3 * j + 1] = (float)rgb[3 * i + 1]; f3[3 * j + 2] = (float)rgb[3 * i + 2];
            f4[2 * j] = (float)y; f4[2 * j + 1] = (float)x;
        }
    fso_crf_step_lr(unary, Qp, f1, f2, f3, f4, val, params, r, H, W, S, 1, out, potentials);
    free(val); free(f4); free(f3); free(f2); free(f1); free(Qp);
}
