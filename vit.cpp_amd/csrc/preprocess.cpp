// preprocess.cpp -- host resize + normalise, replaces vit_image_preprocess
// (/root/reference/vit.cpp:289-305; bicubic 204-287, bilinear 130-196, clip 198-201).
//
// Behaviour kept from the reference: direct (aspect-ignoring) resize to img_size^2 with
// source scale nx/img_size, no half-pixel offset and no antialiasing for bicubic;
// 4x4 neighbourhood with edge clamp; cubic coefficients evaluated in double then narrowed
// to float, polynomial in float; result rounded, clamped to [0,255], narrowed to u8 and
// normalised with the ImageNet mean/std (vit.cpp:233-234).  Rows are independent, so the
// work is split over host threads (the reference runs it on one thread).
#include <math.h>
#include <stdint.h>

#include <thread>
#include <vector>

#include "model_file.h"

namespace {

const float kMean[3] = {123.675f, 116.280f, 103.530f};
const float kStd[3] = {58.395f, 57.120f, 57.375f};

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline float to_u8_norm(float v, int k) {
    const uint8_t q = (uint8_t)fminf(fmaxf(roundf(v), 0.0f), 255.0f);
    return ((float)q - kMean[k]) / kStd[k];
}
// the reference's cubic through 4 samples p0..p3 at fractional offset t (vit.cpp:260-268)
inline float cubic(float p0, float p1, float p2, float p3, float t) {
    const float d0 = p0 - p1, d2 = p2 - p1, d3 = p3 - p1;
    const float a1 = (float)(-1.0 / 3 * d0 + d2 - 1.0 / 6 * d3);
    const float a2 = (float)(1.0 / 2 * d0 + 1.0 / 2 * d2);
    const float a3 = (float)(-1.0 / 6 * d0 - 1.0 / 2 * d2 + 1.0 / 6 * d3);
    return p1 + a1 * t + a2 * t * t + a3 * t * t * t;
}

void bicubic_rows(const uint8_t *src, int nx, int ny, int S, float *dst, int row0, int row1) {
    const float tx = (float)nx / (float)S, ty = (float)ny / (float)S;
    for (int i = row0; i < row1; ++i) {
        const int y = (int)(ty * i);
        const float dy = ty * i - y;
        int ys[4];
        for (int jj = 0; jj < 4; ++jj) ys[jj] = clampi(y - 1 + jj, 0, ny - 1);
        for (int j = 0; j < S; ++j) {
            const int x = (int)(tx * j);
            const float dx = tx * j - x;
            const int x0 = clampi(x - 1, 0, nx - 1), x1 = clampi(x, 0, nx - 1), x2 = clampi(x + 1, 0, nx - 1), x3 = clampi(x + 2, 0, nx - 1);
            for (int k = 0; k < 3; ++k) {
                float C[4];
                for (int jj = 0; jj < 4; ++jj) {
                    const uint8_t *r = src + (size_t)ys[jj] * nx * 3 + k;
                    C[jj] = cubic(r[x0 * 3], r[x1 * 3], r[x2 * 3], r[x3 * 3], dx);
                }
                dst[((size_t)i * S + j) * 3 + k] = to_u8_norm(cubic(C[0], C[1], C[2], C[3], dy), k);
            }
        }
    }
}

void bilinear_rows(const uint8_t *src, int nx, int ny, int S, float *dst, int row0, int row1) {
    const float xs = nx / (float)S, ys = ny / (float)S;
    for (int y = row0; y < row1; ++y) {
        const float sy = (y + 0.5f) * ys - 0.5f;
        const int y0 = sy < 0.0f ? 0 : (int)floorf(sy), y1 = y0 + 1 < ny - 1 ? y0 + 1 : ny - 1;
        const float dy = sy - y0;
        for (int x = 0; x < S; ++x) {
            const float sx = (x + 0.5f) * xs - 0.5f;
            const int x0 = sx < 0.0f ? 0 : (int)floorf(sx), x1 = x0 + 1 < nx - 1 ? x0 + 1 : nx - 1;
            const float dx = sx - x0;
            for (int c = 0; c < 3; ++c) {
                const float v00 = src[3 * ((size_t)y0 * nx + x0) + c], v01 = src[3 * ((size_t)y0 * nx + x1) + c];
                const float v10 = src[3 * ((size_t)y1 * nx + x0) + c], v11 = src[3 * ((size_t)y1 * nx + x1) + c];
                const float v0 = v00 * (1.0f - dx) + v01 * dx, v1 = v10 * (1.0f - dx) + v11 * dx;
                dst[3 * ((size_t)y * S + x) + c] = to_u8_norm(v0 * (1.0f - dy) + v1 * dy, c);
            }
        }
    }
}

}  // namespace

extern "C" int vitx_preprocess_u8(const uint8_t *hwc, int nx, int ny, int img_size, int interp, float *out) {
    if (!hwc || !out || nx <= 0 || ny <= 0 || img_size <= 0) { vitx::set_error("vitx_preprocess_u8: invalid argument"); return VITX_ERR_ARG; }
    if (interp != VITX_BICUBIC && interp != VITX_BILINEAR) {   // vit.cpp:300-304 returns false for any other mode
        vitx::set_error("vitx_preprocess_u8: interpolation mode %d is not supported", interp); return VITX_ERR_ARG;
    }
    const int S = img_size;
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 16) nt = 16;
    if ((unsigned)S < nt * 8) nt = 1;
    auto run = [&](int r0, int r1) { if (interp == VITX_BICUBIC) bicubic_rows(hwc, nx, ny, S, out, r0, r1); else bilinear_rows(hwc, nx, ny, S, out, r0, r1); };
    if (nt == 1) { run(0, S); return VITX_OK; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(run, (int)((long)S * t / nt), (int)((long)S * (t + 1) / nt));
    for (auto &t : th) t.join();
    return VITX_OK;
}

// ---- ViTSTR (extensions/vitstr.cpp) ------------------------------------------------------------------------------------
// vit_image_preprocess of the scene-text extension (vitstr.cpp:135-201): RGB -> grey with PIL's weights evaluated in double and
// TRUNCATED to u8 (:128-132), direct resize by (nx / S, ny / S) with a 2x2 linear blend anchored at the truncated source
// coordinate (clamped so that the +1 neighbour exists), then (v / 255 - 0.5) * 2.  The "padding" loops of the reference run from
// res.nx == S and are no-ops.  Output: ONE channel, [S][S] f32.
extern "C" int vitx_preprocess_vitstr_u8(const uint8_t *hwc, int nx, int ny, int img_size, float *out) {
    if (!hwc || !out || nx <= 0 || ny <= 0 || img_size <= 0) { vitx::set_error("vitx_preprocess_vitstr_u8: invalid argument"); return VITX_ERR_ARG; }
    // the reference reads pixel (x + 1, y + 1) unconditionally: a 1-pixel-wide or 1-pixel-high image is outside its domain
    if (nx < 2 || ny < 2) { vitx::set_error("vitx_preprocess_vitstr_u8: image must be at least 2 x 2 (got %d x %d)", nx, ny); return VITX_ERR_ARG; }
    const int S = img_size;
    std::vector<uint8_t> grey((size_t)nx * ny);
    for (size_t i = 0; i < (size_t)nx * ny; ++i) grey[i] = (uint8_t)(0.299 * hwc[3 * i] + 0.587 * hwc[3 * i + 1] + 0.114 * hwc[3 * i + 2]);
    const float xs = (float)nx / S, ys = (float)ny / S;
    for (int y = 0; y < S; ++y) {
        for (int x = 0; x < S; ++x) {
            const float gx = x * xs, gy = y * ys;
            const int gxi = (int)gx, gyi = (int)gy;
            const float u = gx - gxi, v = gy - gyi;
            const int px0 = clampi(gxi, 0, nx - 2), py0 = clampi(gyi, 0, ny - 2), px1 = px0 + 1, py1 = py0 + 1;
            float val = (1 - u) * (1 - v) * grey[(size_t)py0 * nx + px0] + u * (1 - v) * grey[(size_t)py0 * nx + px1] +
                        (1 - u) * v * grey[(size_t)py1 * nx + px0] + u * v * grey[(size_t)py1 * nx + px1];
            val = (val / 255.0f - 0.5f) * 2.0f;
            out[(size_t)y * S + x] = val;
        }
    }
    return VITX_OK;
}

// The greedy decode of the extension's vit_predict (vitstr.cpp:1025-1051): positions 1 .. seq_len - 1 (position 0 is the [GO] slot),
// arg-max class per position (first maximum wins, strict '>'), stop at class 1 = "[s]"; the score is the product of the maxima of
// the characters emitted.  ids receives the class of every emitted character.
extern "C" int vitx_vitstr_decode(const float *probs, int seq_len, int num_classes, int32_t *ids, int *n_ids, double *score) {
    if (!probs || !ids || !n_ids || seq_len <= 0 || num_classes <= 0) { vitx::set_error("vitx_vitstr_decode: invalid argument"); return VITX_ERR_ARG; }
    double conf = 1.0; int n = 0;
    for (int col = 1; col < seq_len; ++col) {
        const float *p = probs + (size_t)col * num_classes;
        int best = 0; float best_v = p[0];
        for (int row = 1; row < num_classes; ++row) if (p[row] > best_v) { best_v = p[row]; best = row; }
        if (best == 1) break;
        conf *= best_v;
        ids[n++] = best;
    }
    *n_ids = n;
    if (score) *score = conf;
    return VITX_OK;
}
