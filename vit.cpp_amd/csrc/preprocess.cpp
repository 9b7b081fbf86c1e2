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
