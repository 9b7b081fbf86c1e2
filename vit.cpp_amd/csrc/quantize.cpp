// quantize.cpp -- native re-encoder of a legacy-ggml ViT file to q4_0 / q4_1 / q5_0 / q5_1 / q8_0.
//
// Replaces the reference's offline `quantize` tool (/root/reference/quantize.cpp:34-353) for the hot path's
// input format.  Same rules: the header is copied with ftype := target type (:113), labels are
// re-emitted in key order (:137-146), every 2-D tensor whose name ends in "weight" (:207-223) is read as f16/f32,
// widened to f32 and re-encoded in blocks of 32 (:271-303); everything else is copied byte for byte (:247-252).
// The block encoders restate ggml's quantize_row_q*_reference:
//   q4_0: d = (value of largest |x|) / -8, q = min(15, (int8)(x/d + 8.5)); 16 low nibbles then 16 high nibbles
//   q4_1: d = (max-min)/15, m = min,       q = min(15, (int8)((x-m)/d + 0.5))
//   q5_0: d = (value of largest |x|) / -16, q = min(31, (int8)(x/d + 16.5)); 5th bits packed in a u32
//   q5_1: d = (max-min)/31, m = min,       q = (uint8)((x-m)/d + 0.5)
//   q8_0: d = max|x| / 127,                q = roundf(x/d)
// Host only; the GPU engine consumes the result through vitx_model_load like any other file.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "model_file.h"

namespace vitx {

static inline void put16(uint8_t *p, float v) { const uint16_t h = f32_to_f16_bits(v); memcpy(p, &h, 2); }

static void encode_block(int type, const float *x, uint8_t *out) {
    constexpr int QK = 32;
    switch (type) {
    case T_Q4_0: case T_Q5_0: {
        float amax = 0.0f, mx = 0.0f;
        for (int j = 0; j < QK; ++j) { const float a = fabsf(x[j]); if (a > amax) { amax = a; mx = x[j]; } }
        const float d = mx / (type == T_Q4_0 ? -8.0f : -16.0f), id = d ? 1.0f / d : 0.0f;
        put16(out, d);
        if (type == T_Q4_0) {
            for (int j = 0; j < 16; ++j) {
                const uint8_t a = (uint8_t)std::min(15, (int)(int8_t)(x[j] * id + 8.5f)), b = (uint8_t)std::min(15, (int)(int8_t)(x[j + 16] * id + 8.5f));
                out[2 + j] = (uint8_t)(a | (b << 4));
            }
        } else {
            uint32_t qh = 0;
            for (int j = 0; j < 16; ++j) {
                const uint8_t a = (uint8_t)std::min(31, (int)(int8_t)(x[j] * id + 16.5f)), b = (uint8_t)std::min(31, (int)(int8_t)(x[j + 16] * id + 16.5f));
                out[6 + j] = (uint8_t)((a & 0x0F) | ((b & 0x0F) << 4));
                qh |= (uint32_t)((a & 0x10) >> 4) << j;
                qh |= (uint32_t)((b & 0x10) >> 4) << (j + 16);
            }
            memcpy(out + 2, &qh, 4);
        }
        break;
    }
    case T_Q4_1: case T_Q5_1: {
        float mn = x[0], mx = x[0];
        for (int j = 1; j < QK; ++j) { mn = std::min(mn, x[j]); mx = std::max(mx, x[j]); }
        const float d = (mx - mn) / (type == T_Q4_1 ? 15.0f : 31.0f), id = d ? 1.0f / d : 0.0f;
        put16(out, d); put16(out + 2, mn);
        if (type == T_Q4_1) {
            for (int j = 0; j < 16; ++j) {
                const uint8_t a = (uint8_t)std::min(15, (int)(int8_t)((x[j] - mn) * id + 0.5f)), b = (uint8_t)std::min(15, (int)(int8_t)((x[j + 16] - mn) * id + 0.5f));
                out[4 + j] = (uint8_t)(a | (b << 4));
            }
        } else {
            uint32_t qh = 0;
            for (int j = 0; j < 16; ++j) {
                const uint8_t a = (uint8_t)((x[j] - mn) * id + 0.5f), b = (uint8_t)((x[j + 16] - mn) * id + 0.5f);
                out[8 + j] = (uint8_t)((a & 0x0F) | ((b & 0x0F) << 4));
                qh |= (uint32_t)((a & 0x10) >> 4) << j;
                qh |= (uint32_t)((b & 0x10) >> 4) << (j + 16);
            }
            memcpy(out + 4, &qh, 4);
        }
        break;
    }
    case T_Q8_0: {
        float amax = 0.0f;
        for (int j = 0; j < QK; ++j) amax = std::max(amax, fabsf(x[j]));
        const float d = amax / 127.0f, id = d ? 1.0f / d : 0.0f;
        put16(out, d);
        for (int j = 0; j < QK; ++j) out[2 + j] = (uint8_t)(int8_t)roundf(x[j] * id);
        break;
    }
    }
}

static bool ends_with(const std::string &s, const char *suffix) {
    const size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

}  // namespace vitx

using namespace vitx;

extern "C" int vitx_quantize_file(const char *path_in, const char *path_out, int ftype) {
    if (!path_in || !path_out) { set_error("vitx_quantize_file: NULL path"); return VITX_ERR_ARG; }
    if (ftype != T_Q4_0 && ftype != T_Q4_1 && ftype != T_Q5_0 && ftype != T_Q5_1 && ftype != T_Q8_0) {
        set_error("vitx_quantize_file: unsupported target type %d (2 q4_0, 3 q4_1, 6 q5_0, 7 q5_1, 8 q8_0)", ftype);   // quantize.cpp:296-300
        return VITX_ERR_ARG;
    }
    if (strcmp(path_in, path_out) == 0) { set_error("vitx_quantize_file: input and output are the same file '%s'", path_in); return VITX_ERR_ARG; }
    vitx_model *m = nullptr;
    int rc = vitx_model_load(path_in, &m);
    if (rc != VITX_OK) return rc;
    // validate every tensor BEFORE any byte is written, then write to a temporary file that replaces path_out only on success:
    // a failure never leaves a truncated model behind
    for (const HostTensor &t : m->tensors) {
        if (!(t.n_dims == 2 && ends_with(t.name, "weight"))) continue;
        if (t.type != T_F32 && t.type != T_F16) { set_error("vitx_quantize_file: tensor '%s' is already quantised (type %d)", t.name.c_str(), t.type); vitx_model_free(m); return VITX_ERR_FORMAT; }
        if (t.ne[0] % 32) { set_error("vitx_quantize_file: row length %lld of '%s' is not a multiple of 32", (long long)t.ne[0], t.name.c_str()); vitx_model_free(m); return VITX_ERR_FORMAT; }
    }
    const std::string tmp = std::string(path_out) + ".tmp" + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) { set_error("vitx_quantize_file: failed to open '%s' for writing", tmp.c_str()); vitx_model_free(m); return VITX_ERR_IO; }
    auto w32 = [&](int32_t v) { return fwrite(&v, 4, 1, f) == 1; };
    bool ok = w32(0x67676d6c) && w32(m->hp.hidden_size) && w32(m->hp.num_hidden_layers) && w32(m->hp.num_attention_heads) && w32(m->hp.num_classes) &&
              w32(m->hp.patch_size) && w32(m->hp.img_size) && w32(ftype) && w32((int32_t)m->id2label.size());
    for (const auto &kv : m->id2label) ok = ok && w32(kv.first) && w32((int32_t)kv.second.size()) && (kv.second.empty() || fwrite(kv.second.data(), kv.second.size(), 1, f) == 1);
    std::vector<float> f32;
    std::vector<uint8_t> blocks;
    for (const HostTensor &t : m->tensors) {
        const bool quantize = t.n_dims == 2 && ends_with(t.name, "weight");
        int32_t out_type = t.type;
        if (quantize) {
            if (t.type != T_F32 && t.type != T_F16) { set_error("vitx_quantize_file: tensor '%s' is already quantised (type %d)", t.name.c_str(), t.type); rc = VITX_ERR_FORMAT; break; }
            if (t.ne[0] % 32) { set_error("vitx_quantize_file: row length %lld of '%s' is not a multiple of 32", (long long)t.ne[0], t.name.c_str()); rc = VITX_ERR_FORMAT; break; }
            out_type = ftype;
        }
        ok = ok && w32(t.n_dims) && w32((int32_t)t.name.size()) && w32(out_type);
        for (int i = 0; i < t.n_dims; ++i) ok = ok && w32((int32_t)t.ne[i]);
        ok = ok && fwrite(t.name.data(), t.name.size(), 1, f) == 1;
        if (quantize) {
            const int64_t n = t.nelements(), nb = n / 32;
            const int bb = type_block_bytes(ftype);
            f32.resize(n); blocks.resize((size_t)nb * bb);
            t.decode_f32(f32.data());
            for (int64_t b = 0; b < nb; ++b) encode_block(ftype, f32.data() + b * 32, blocks.data() + (size_t)b * bb);
            ok = ok && fwrite(blocks.data(), blocks.size(), 1, f) == 1;
        } else {
            ok = ok && (t.raw.empty() || fwrite(t.raw.data(), t.raw.size(), 1, f) == 1);
        }
    }
    if (fclose(f) != 0) ok = false;
    vitx_model_free(m);
    if (rc != VITX_OK) { (void)remove(tmp.c_str()); return rc; }
    if (!ok) { (void)remove(tmp.c_str()); set_error("vitx_quantize_file: short write to '%s'", tmp.c_str()); return VITX_ERR_IO; }
    if (rename(tmp.c_str(), path_out) != 0) { (void)remove(tmp.c_str()); set_error("vitx_quantize_file: cannot move the result to '%s'", path_out); return VITX_ERR_IO; }
    return VITX_OK;
}
