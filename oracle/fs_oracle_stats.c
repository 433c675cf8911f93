/*
 * fs_oracle_stats.c -- CPU restatement of the per-surface statistics stage of the
 * reference (process_surfaces.py:8-129, 236-297, 332-346, called in the order of
 * extract_information_from_surface_data_and_preprocess_surfaces, :672-689).
 *
 * TEST INFRASTRUCTURE ONLY (see fs_oracle.c).  PARITY UNPINNED: the reference's
 * numba loops cannot be run here (numba is not installed) and the reference ships no
 * outputs of this stage.  Assumptions that cannot be verified here:
 *   - inside numba-compiled code, `float32_array / float64_scalar` promotes to float64
 *     (numba types by dtype, not by value), so the angle-histogram bin index is computed
 *     in double (process_surfaces.py:251, 273);
 *   - Python's round() on a float rounds half to even (:257, 265).
 * Loops run in the reference's raster order, so integer results and the visiting order
 * of first-minimum searches are the reference's.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FSO_API __attribute__((visibility("default")))

/* extract_individual_points_from_surfaces, process_surfaces.py:96-108: per label the
 * [x, y] of its pixels in raster order.  offsets[S+1], points[N][2] (label-major). */
FSO_API void fso_extract_points(const int64_t *patches, int H, int W, int S, int32_t *offsets, float *points)
{
    const int N = H * W;
    memset(offsets, 0, sizeof(int32_t) * (S + 1));
    for (int i = 0; i < N; ++i) offsets[patches[i] + 1]++;
    for (int l = 0; l < S; ++l) offsets[l + 1] += offsets[l];
    int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * S);
    memcpy(cur, offsets, sizeof(int32_t) * S);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int k = cur[patches[y * W + x]]++;
            points[2 * k] = (float)x;
            points[2 * k + 1] = (float)y;
        }
    free(cur);
}

/* relabel_surfaces, :236-242 (in place) */
FSO_API void fso_relabel(int64_t *surfaces, const int64_t *patches, int N)
{
    for (int i = 0; i < N; ++i)
        if (patches[i] != 0) surfaces[i] = patches[i];
}

/* determine_pixel_counts_and_average_position_for_patches, :111-129: interior pixels only;
 * avg_pos[S][2] = (mean x, mean y), counts[S] */
FSO_API void fso_counts_avg_pos(const int64_t *surfaces, int H, int W, int S, double *avg_pos, double *counts)
{
    int64_t *sx = (int64_t *)calloc(S, sizeof(int64_t)), *sy = (int64_t *)calloc(S, sizeof(int64_t));
    for (int l = 0; l < S; ++l) counts[l] = 0.0;
    for (int y = 1; y < H - 1; ++y)
        for (int x = 1; x < W - 1; ++x) {
            const int64_t s = surfaces[y * W + x];
            counts[s] += 1.0;
            sx[s] += x;
            sy[s] += y;
        }
    for (int l = 0; l < S; ++l) {
        avg_pos[2 * l] = avg_pos[2 * l + 1] = 0.0;
        if (counts[l] > 0) {
            avg_pos[2 * l] = (double)sx[l] / counts[l];
            avg_pos[2 * l + 1] = (double)sy[l] / counts[l];
        }
    }
    free(sx);
    free(sy);
}

/* determine_surface_patch_centroids, :81-93: per label the first patch point (raster order)
 * with the smallest squared distance to avg_pos; centroid = points_3d at it.  A label
 * without patch points gets index (-1,-1) and centroid (-1,-1,-1) at its own position
 * (the reference appends nothing to the index list there, shifting later labels; labels
 * produced by the patch stage always have points, so that case does not occur on this path). */
FSO_API void fso_centroids(const float *points, const int32_t *offsets, const double *avg_pos, const float *points_3d, int S, int W,
                   float *centroids, int32_t *centroid_idx)
{
    for (int l = 0; l < S; ++l) {
        const int a = offsets[l], b = offsets[l + 1];
        if (a == b) {
            centroid_idx[2 * l] = centroid_idx[2 * l + 1] = -1;
            centroids[3 * l] = centroids[3 * l + 1] = centroids[3 * l + 2] = -1.0f;
            continue;
        }
        int best = a;
        double bd = INFINITY;
        for (int k = a; k < b; ++k) {
            const double dx = (double)points[2 * k] - avg_pos[2 * l], dy = (double)points[2 * k + 1] - avg_pos[2 * l + 1];
            const double d = dx * dx + dy * dy;
            if (d < bd) { bd = d; best = k; }
        }
        const int x = (int)points[2 * best], y = (int)points[2 * best + 1];
        centroid_idx[2 * l] = y;
        centroid_idx[2 * l + 1] = x;
        memcpy(centroids + 3 * l, points_3d + 3 * ((size_t)y * W + x), 3 * sizeof(float));
    }
}

/* remove_disconnected_components, :8-28: 4-connected flood fill of `surfaces == i` from label
 * i's centroid pixel (the start pixel is kept whatever its label); everything unreached -> 0. */
FSO_API void fso_remove_disconnected(int64_t *surfaces, int H, int W, const int32_t *centroid_idx, int S)
{
    const int N = H * W;
    uint8_t *conn = (uint8_t *)calloc(N, 1);
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (4 * (size_t)N + 4));
    for (int i = 0; i < S; ++i) {
        if (centroid_idx[2 * i] == -1) continue;
        size_t top = 0;
        stack[top++] = centroid_idx[2 * i] * W + centroid_idx[2 * i + 1];
        while (top > 0) {
            const int p = stack[--top];
            const int y = p / W, x = p - y * W;
            conn[p] = 1;
            const int ny[4] = {y - 1, y + 1, y, y}, nx[4] = {x, x, x - 1, x + 1};
            for (int k = 0; k < 4; ++k) {
                if (ny[k] < 0 || nx[k] < 0 || ny[k] >= H || nx[k] >= W) continue;
                const int q = ny[k] * W + nx[k];
                if (conn[q] || surfaces[q] != i) continue;
                conn[q] = 1;          /* marking at push time visits the same set as marking at pop time */
                stack[top++] = q;
            }
        }
    }
    for (int p = 0; p < N; ++p)
        if (!conn[p]) surfaces[p] = 0;
    free(conn);
    free(stack);
}

/* determine_histograms_and_average_normals, :245-297.  lab f64 [H][W][3], angles f32 [H][W][2],
 * texture f32 [H/4][W/4][T] (may be NULL: histogram stays 0).  Outputs: hist_color f64
 * [256][256][S], hist_angle f64 [40][40][S], hist_texture f64 [S][T], avg_angles f64 [S][2]. */
FSO_API void fso_histograms(const double *lab, const float *angles, const float *texture, int T, const int64_t *surfaces,
                    const int64_t *patches, int H, int W, int S, double *hist_color, double *hist_angle, double *hist_texture,
                    double *avg_angles)
{
    memset(hist_color, 0, sizeof(double) * 256 * 256 * (size_t)S);
    memset(hist_angle, 0, sizeof(double) * 40 * 40 * (size_t)S);
    memset(hist_texture, 0, sizeof(double) * (size_t)S * T);
    memset(avg_angles, 0, sizeof(double) * 2 * (size_t)S);
    double *counts = (double *)calloc(S, sizeof(double)), *patch_counter = (double *)calloc(S, sizeof(double));
    const int TW = W / 4;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t p = (size_t)y * W + x;
            int64_t s = surfaces[p];
            if (s < 1) continue;
            const long a = lrint(lab[3 * p + 1] + 127.0), b = lrint(lab[3 * p + 2] + 127.0);   /* round half to even */
            hist_color[((size_t)a * 256 + b) * S + s] += 1.0;
            s = patches[p];
            if (s < 1) continue;
            const double n1 = ((double)angles[2 * p] / (2 * M_PI) + 0.5) * 39, n2 = ((double)angles[2 * p + 1] / (2 * M_PI) + 0.5) * 39;
            hist_angle[((size_t)(int)n1 * 40 + (int)n2) * S + s] += 1.0;
            avg_angles[2 * s] += (double)angles[2 * p];
            avg_angles[2 * s + 1] += (double)angles[2 * p + 1];
            patch_counter[s] += 1.0;
            if (y % 4 == 0 && x % 4 == 0) {
                if (texture) {
                    const float *t = texture + ((size_t)(y / 4) * TW + x / 4) * T;
                    for (int k = 0; k < T; ++k) hist_texture[(size_t)s * T + k] += (double)t[k];
                }
                counts[s] += 1.0;
            }
        }
    for (int s = 0; s < S; ++s) {
        double norm = 0.0;
        for (size_t ab = 0; ab < 256 * 256; ++ab) norm += hist_color[ab * S + s];
        norm /= 10;
        if (norm == 0) norm = 1;
        for (size_t ab = 0; ab < 256 * 256; ++ab) hist_color[ab * S + s] /= norm;
        if (patch_counter[s] == 0) patch_counter[s] = 1;
        avg_angles[2 * s] /= patch_counter[s];
        avg_angles[2 * s + 1] /= patch_counter[s];
        const double an = patch_counter[s] / 10;
        for (int ab = 0; ab < 40 * 40; ++ab) hist_angle[(size_t)ab * S + s] /= an;
        if (counts[s] == 0) counts[s] = 1;
        for (int k = 0; k < T; ++k) hist_texture[(size_t)s * T + k] = hist_texture[(size_t)s * T + k] / counts[s] * 10;
    }
    free(counts);
    free(patch_counter);
}

/* determine_neighboring_surfaces_and_border_centers, :31-68 (the matrices; the neighbour lists
 * of :70-77 are `neighbors[i][j] > 3`, j >= 1).  neighbors f64 [S][S], border f64 [S][S][2]. */
FSO_API void fso_neighbors(const int64_t *surfaces, const int32_t *depth_edges, int H, int W, int S, double *neighbors, double *border)
{
    memset(neighbors, 0, sizeof(double) * (size_t)S * S);
    memset(border, 0, sizeof(double) * 2 * (size_t)S * S);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t p = (size_t)y * W + x;
            if (depth_edges[p] == 0 || surfaces[p] == 0) continue;
            const int64_t i = surfaces[p] * depth_edges[p];
            if (y < H - 1) {
                const int64_t o = surfaces[p + W] * depth_edges[p + W];
                neighbors[i * S + o] += 1;
                neighbors[o * S + i] += 1;
                border[2 * (i * S + o)] += 2 * x + 1;  border[2 * (i * S + o) + 1] += 2 * y;
                border[2 * (o * S + i)] += 2 * x + 1;  border[2 * (o * S + i) + 1] += 2 * y;
            }
            if (x < W - 1) {
                const int64_t o = surfaces[p + 1] * depth_edges[p + 1];
                neighbors[i * S + o] += 1;
                neighbors[o * S + i] += 1;
                border[2 * (i * S + o)] += 2 * x;  border[2 * (i * S + o) + 1] += 2 * y + 1;
                border[2 * (o * S + i)] += 2 * x;  border[2 * (o * S + i) + 1] += 2 * y + 1;
            }
        }
    for (int j = 0; j < S; ++j) {
        neighbors[j] = 0; neighbors[(size_t)j * S] = 0;
        border[2 * j] = border[2 * j + 1] = 0;
        border[2 * (size_t)j * S] = border[2 * (size_t)j * S + 1] = 0;
    }
    for (int i = 0; i < S; ++i) {
        neighbors[(size_t)i * S + i] = 0;
        border[2 * ((size_t)i * S + i)] = border[2 * ((size_t)i * S + i) + 1] = 0;
        for (int j = 0; j < S; ++j)
            if (neighbors[(size_t)i * S + j] < 4) {
                neighbors[(size_t)i * S + j] = 1;
                neighbors[(size_t)j * S + i] = 1;
            }
    }
    for (size_t k = 0; k < (size_t)S * S; ++k) {
        border[2 * k] /= 2 * neighbors[k];
        border[2 * k + 1] /= 2 * neighbors[k];
    }
}

/* norm = np.linalg.norm(points_3d, axis=-1) (:686, float32) and calculate_depth_extend, :332-346:
 * values f32 [S][4][2] = min / max of (norm, x, y, z) over the pixels of a label with norm != 0 */
FSO_API void fso_depth_extend(const int64_t *surfaces, const float *points_3d, int H, int W, int S, float *norm, float *values)
{
    for (int s = 0; s < S; ++s)
        for (int c = 0; c < 4; ++c) {
            values[(s * 4 + c) * 2] = INFINITY;
            values[(s * 4 + c) * 2 + 1] = -INFINITY;
        }
    for (size_t p = 0; p < (size_t)H * W; ++p) {
        const float x = points_3d[3 * p], y = points_3d[3 * p + 1], z = points_3d[3 * p + 2];
        const float n = sqrtf((x * x + y * y) + z * z);
        norm[p] = n;
        if (n == 0) continue;
        const float v[4] = {n, x, y, z};
        float *dst = values + surfaces[p] * 8;
        for (int c = 0; c < 4; ++c) {
            dst[2 * c] = fminf(dst[2 * c], v[c]);
            dst[2 * c + 1] = fmaxf(dst[2 * c + 1], v[c]);
        }
    }
}
