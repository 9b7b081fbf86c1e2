// model_file.cpp -- legacy-ggml ".gguf" reader (host only, no GPU, no ggml).
//
// Same acceptance rules as the reference loader (/root/reference/vit.cpp:308-712):
// magic (320-328), 7 int32 hparams (335-341, ftype %= 1000 at 354), id2label (356-371),
// then named tensors until EOF (590-695): unknown name -> error (618-622), element-count
// and shape must match what the hparams imply (627-641), byte size must match the type
// (680-685), and every expected tensor must appear exactly once (697-701).  One deliberate
// widening: patch_embed.proj.weight is accepted as f32 as well as f16 (the reference only
// takes f16 there, vit.cpp:515, so "--ftype 0" files it cannot load are loadable here).
#include "model_file.h"

#include <stdarg.h>
#include <stdio.h>
#include <atomic>
#include <string.h>

#include <algorithm>
#include <memory>
#include <new>

namespace vitx {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
const char *last_error() { return g_err; }

int type_block_bytes(int t) {
    switch (t) { case T_F32: return 4; case T_F16: return 2; case T_Q4_0: return 18; case T_Q4_1: return 20;
                 case T_Q5_0: return 22; case T_Q5_1: return 24; case T_Q8_0: return 34; default: return 0; }
}
int type_block_elems(int t) { return (t == T_F32 || t == T_F16) ? 1 : 32; }

float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { int e = -1; do { ++e; man <<= 1; } while (!(man & 0x400u)); bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13); }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}
uint16_t f32_to_f16_bits(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));   // inf / nan
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                    // rounds to inf
    if (x < 0x33000001u) return (uint16_t)sign;                                                 // rounds to zero
    if (x < 0x38800000u) {                                                                      // subnormal half
        const int shift = 126 - (int)(x >> 23);                  // 14..24
        const uint32_t man = (x & 0x7fffffu) | 0x800000u;
        uint32_t r = man >> shift; const uint32_t rem = man & ((1u << shift) - 1), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (r & 1))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((x - 0x38000000u) >> 13); const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;
    return (uint16_t)(sign | r);
}
uint16_t f32_to_bf16_bits(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u);
    x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}

static inline float rd16(const uint8_t *p) { uint16_t h; memcpy(&h, p, 2); return f16_bits_to_f32(h); }

void HostTensor::decode_f32(float *out) const {
    const int64_t n = nelements();
    if (type == T_F32) { memcpy(out, raw.data(), (size_t)n * 4); return; }
    if (type == T_F16) { const uint8_t *p = raw.data(); for (int64_t i = 0; i < n; ++i) out[i] = rd16(p + 2 * i); return; }
    const int bb = type_block_bytes(type);
    const int64_t nb = n / 32;
    for (int64_t b = 0; b < nb; ++b) {
        const uint8_t *p = raw.data() + (size_t)b * bb; float *o = out + b * 32;
        switch (type) {
        case T_Q4_0: { const float d = rd16(p); const uint8_t *qs = p + 2;
            for (int j = 0; j < 16; ++j) { o[j] = ((qs[j] & 0x0F) - 8) * d; o[j + 16] = ((qs[j] >> 4) - 8) * d; } } break;
        case T_Q4_1: { const float d = rd16(p), m = rd16(p + 2); const uint8_t *qs = p + 4;
            for (int j = 0; j < 16; ++j) { o[j] = (qs[j] & 0x0F) * d + m; o[j + 16] = (qs[j] >> 4) * d + m; } } break;
        case T_Q5_0: { const float d = rd16(p); uint32_t qh; memcpy(&qh, p + 2, 4); const uint8_t *qs = p + 6;
            for (int j = 0; j < 16; ++j) { const uint8_t h0 = ((qh >> j) << 4) & 0x10, h1 = (qh >> (j + 12)) & 0x10;
                o[j] = (((qs[j] & 0x0F) | h0) - 16) * d; o[j + 16] = (((qs[j] >> 4) | h1) - 16) * d; } } break;
        case T_Q5_1: { const float d = rd16(p), m = rd16(p + 2); uint32_t qh; memcpy(&qh, p + 4, 4); const uint8_t *qs = p + 8;
            for (int j = 0; j < 16; ++j) { const uint8_t h0 = ((qh >> j) << 4) & 0x10, h1 = (qh >> (j + 12)) & 0x10;
                o[j] = ((qs[j] & 0x0F) | h0) * d + m; o[j + 16] = ((qs[j] >> 4) | h1) * d + m; } } break;
        case T_Q8_0: { const float d = rd16(p); const int8_t *qs = (const int8_t *)(p + 2);
            for (int j = 0; j < 32; ++j) o[j] = qs[j] * d; } break;
        }
    }
}

// expected tensors and their ggml shapes (vit.cpp:506-581); type classes: 0 = must be f32,
// 1 = "wtype" (f32/f16/q*), 2 = patch kernel (f16, f32 also accepted here)
struct Expect { int64_t ne[4]; int cls; };
static std::map<std::string, Expect> expected_tensors(const vitx_hparams &hp) {
    std::map<std::string, Expect> e;
    const int64_t D = hp.hidden_size, C = hp.num_classes, P = hp.patch_size, g = hp.img_size / hp.patch_size, N = g * g + 1;
    e["pos_embed"] = {{D, N, 1, 1}, 0};
    e["cls_token"] = {{D, 1, 1, 1}, 0};
    e["patch_embed.proj.weight"] = {{P, P, 3, D}, 2};
    e["patch_embed.proj.bias"] = {{1, 1, D, 1}, 0};
    for (int i = 0; i < hp.num_hidden_layers; ++i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        e[p + "norm1.weight"] = {{D, 1, 1, 1}, 0}; e[p + "norm1.bias"] = {{D, 1, 1, 1}, 0};
        e[p + "attn.qkv.weight"] = {{D, 3 * D, 1, 1}, 1}; e[p + "attn.qkv.bias"] = {{3 * D, 1, 1, 1}, 0};
        e[p + "attn.proj.weight"] = {{D, D, 1, 1}, 1}; e[p + "attn.proj.bias"] = {{D, 1, 1, 1}, 0};
        e[p + "norm2.weight"] = {{D, 1, 1, 1}, 0}; e[p + "norm2.bias"] = {{D, 1, 1, 1}, 0};
        e[p + "mlp.fc1.weight"] = {{D, 4 * D, 1, 1}, 1}; e[p + "mlp.fc1.bias"] = {{4 * D, 1, 1, 1}, 0};
        e[p + "mlp.fc2.weight"] = {{4 * D, D, 1, 1}, 1}; e[p + "mlp.fc2.bias"] = {{D, 1, 1, 1}, 0};
    }
    e["norm.weight"] = {{D, 1, 1, 1}, 0}; e["norm.bias"] = {{D, 1, 1, 1}, 0};
    e["head.weight"] = {{D, C, 1, 1}, 1}; e["head.bias"] = {{C, 1, 1, 1}, 0};
    return e;
}

struct FileCloser { void operator()(FILE *f) const { if (f) fclose(f); } };

static int load_impl(const char *path, vitx_model &m) {
    std::unique_ptr<FILE, FileCloser> fh(fopen(path, "rb"));
    FILE *f = fh.get();
    if (!f) { set_error("vitx_model_load: failed to open '%s'", path); return VITX_ERR_IO; }
    auto rd_i32 = [&](int32_t &v) { return fread(&v, 4, 1, f) == 1; };
    int32_t magic = 0;
    if (!rd_i32(magic) || (uint32_t)magic != 0x67676d6cu) { set_error("vitx_model_load: invalid model file '%s' (bad magic)", path); return VITX_ERR_FORMAT; }
    int32_t hv[7];
    for (int i = 0; i < 7; ++i) if (!rd_i32(hv[i])) { set_error("vitx_model_load: truncated header"); return VITX_ERR_IO; }
    vitx_hparams &hp = m.hp;
    hp.hidden_size = hv[0]; hp.num_hidden_layers = hv[1]; hp.num_attention_heads = hv[2]; hp.num_classes = hv[3];
    hp.patch_size = hv[4]; hp.img_size = hv[5]; hp.ftype = hv[6] % 1000 /* GGML_QNT_VERSION_FACTOR */; hp.eps = 1e-6f;
    if (hp.hidden_size <= 0 || hp.num_hidden_layers <= 0 || hp.num_attention_heads <= 0 || hp.num_classes <= 0 || hp.patch_size <= 0 ||
        hp.img_size <= 0 || hp.img_size % hp.patch_size || hp.hidden_size % hp.num_attention_heads || hp.num_hidden_layers > 4096) {
        set_error("vitx_model_load: implausible hparams in '%s'", path); return VITX_ERR_FORMAT;
    }
    if (!type_block_bytes(hp.ftype)) { set_error("vitx_model_load: invalid model file '%s' (bad ftype value %d)", path, hp.ftype); return VITX_ERR_FORMAT; }
    int32_t nl = 0;
    if (!rd_i32(nl) || nl < 0 || nl > (1 << 24)) { set_error("vitx_model_load: bad label count"); return VITX_ERR_FORMAT; }
    for (int i = 0; i < nl; ++i) {
        int32_t key, len;
        if (!rd_i32(key) || !rd_i32(len) || len < 0 || len > (1 << 20)) { set_error("vitx_model_load: bad id2label entry"); return VITX_ERR_FORMAT; }
        std::string v((size_t)len, '\0');
        if (len && fread(&v[0], 1, (size_t)len, f) != (size_t)len) { set_error("vitx_model_load: truncated id2label"); return VITX_ERR_IO; }
        m.id2label[key] = v;
    }
    const auto expect = expected_tensors(hp);
    for (;;) {
        int32_t n_dims, name_len, ttype;
        if (!rd_i32(n_dims)) break;                                  // clean EOF (vit.cpp:600-603)
        if (!rd_i32(name_len) || !rd_i32(ttype)) { set_error("vitx_model_load: truncated tensor header"); return VITX_ERR_IO; }
        if (n_dims < 1 || n_dims > 4 || name_len <= 0 || name_len > 255) { set_error("vitx_model_load: bad tensor header"); return VITX_ERR_FORMAT; }
        vitx::HostTensor t; t.n_dims = n_dims; t.type = ttype;
        for (int i = 0; i < n_dims; ++i) { int32_t v; if (!rd_i32(v) || v <= 0) { set_error("vitx_model_load: bad dims"); return VITX_ERR_FORMAT; } t.ne[i] = v; }
        t.name.resize((size_t)name_len);
        if (fread(&t.name[0], 1, (size_t)name_len, f) != (size_t)name_len) { set_error("vitx_model_load: truncated name"); return VITX_ERR_IO; }
        auto it = expect.find(t.name);
        if (it == expect.end()) { set_error("vitx_model_load: unknown tensor '%s' in model file", t.name.c_str()); return VITX_ERR_FORMAT; }
        if (m.index.count(t.name)) { set_error("vitx_model_load: duplicate tensor '%s'", t.name.c_str()); return VITX_ERR_FORMAT; }
        Expect ex = it->second;
        // ViTSTR files (extensions/vitstr.cpp/vitstr.cpp:482) carry a ONE-channel patch kernel [P, P, 1, D]: same format otherwise
        if (t.name == "patch_embed.proj.weight" && t.ne[2] == 1 && t.ne[0] == ex.ne[0] && t.ne[1] == ex.ne[1] && t.ne[3] == ex.ne[3]) { ex.ne[2] = 1; m.in_chans = 1; }
        const int64_t want = ex.ne[0] * ex.ne[1] * ex.ne[2] * ex.ne[3];
        if (t.nelements() != want) { set_error("vitx_model_load: tensor '%s' has wrong size in model file: got %lld, expected %lld", t.name.c_str(), (long long)t.nelements(), (long long)want); return VITX_ERR_FORMAT; }
        if (t.ne[0] != ex.ne[0] || t.ne[1] != ex.ne[1] || t.ne[2] != ex.ne[2] || t.ne[3] != ex.ne[3]) {
            set_error("vitx_model_load: tensor '%s' has wrong shape in model file: got [%lld, %lld, %lld, %lld], expected [%lld, %lld, %lld, %lld]", t.name.c_str(),
                      (long long)t.ne[0], (long long)t.ne[1], (long long)t.ne[2], (long long)t.ne[3], (long long)ex.ne[0], (long long)ex.ne[1], (long long)ex.ne[2], (long long)ex.ne[3]);
            return VITX_ERR_FORMAT;
        }
        const int bb = type_block_bytes(ttype), be = type_block_elems(ttype);
        if (!bb) { set_error("vitx_model_load: unknown ftype %d in model file", ttype); return VITX_ERR_FORMAT; }
        if (be > 1 && t.ne[0] % 64) { set_error("vitx_model_load: quantised tensor '%s' needs ne[0] %% 64 == 0", t.name.c_str()); return VITX_ERR_FORMAT; }
        // the declared type of each slot decides the expected byte size (vit.cpp:680-685)
        const bool type_ok = (ex.cls == 0) ? (ttype == T_F32) : (ex.cls == 2) ? (ttype == T_F16 || ttype == T_F32) : (ttype == hp.ftype);
        if (!type_ok) { set_error("vitx_model_load: tensor '%s' has wrong size in model file (type %d not allowed for this slot, file ftype %d)", t.name.c_str(), ttype, hp.ftype); return VITX_ERR_FORMAT; }
        const size_t nbytes = (size_t)(t.nelements() / be) * bb;
        t.raw.resize(nbytes);
        if (fread(t.raw.data(), 1, nbytes, f) != nbytes) { set_error("vitx_model_load: tensor '%s' is truncated", t.name.c_str()); return VITX_ERR_IO; }
        m.index[t.name] = (int)m.tensors.size();
        m.tensors.push_back(std::move(t));
    }
    if (m.tensors.size() != expect.size()) {
        set_error("vitx_model_load: model file has %d tensors, but %d tensors were expected", (int)m.tensors.size(), (int)expect.size());
        return VITX_ERR_FORMAT;
    }
    return VITX_OK;
}

}  // namespace vitx

extern "C" {

const char *vitx_status_str(int s) {
    switch (s) {
    case VITX_OK: return "ok"; case VITX_ERR_IO: return "io error"; case VITX_ERR_FORMAT: return "bad model file";
    case VITX_ERR_ARG: return "invalid argument"; case VITX_ERR_HIP: return "HIP error"; case VITX_ERR_UNSUPPORTED: return "unsupported model shape";
    case VITX_ERR_NOMEM: return "out of memory"; default: return "unknown status";
    }
}
const char *vitx_last_error(void) { return vitx::last_error(); }

int vitx_model_load(const char *path, vitx_model **out) {
    if (!path || !out) { vitx::set_error("vitx_model_load: NULL argument"); return VITX_ERR_ARG; }
    *out = nullptr;
    std::unique_ptr<vitx_model> m(new (std::nothrow) vitx_model());
    if (!m) return VITX_ERR_NOMEM;
    const int rc = vitx::load_impl(path, *m);
    if (rc != VITX_OK) return rc;
    static std::atomic<uint64_t> next_uid{1};
    m->uid = next_uid.fetch_add(1);
    *out = m.release();
    return VITX_OK;
}
void vitx_model_free(vitx_model *m) { delete m; }
uint64_t vitx_model_uid(const vitx_model *m) { return m ? m->uid : 0; }
int vitx_model_hparams(const vitx_model *m, vitx_hparams *out) {
    if (!m || !out) return VITX_ERR_ARG;
    *out = m->hp; return VITX_OK;
}
int vitx_model_num_labels(const vitx_model *m) { return m ? (int)m->id2label.size() : 0; }
int vitx_model_in_channels(const vitx_model *m) { return m ? m->in_chans : 0; }
int vitx_model_seq_len(const vitx_model *m) { return (m && m->in_chans == 1) ? VITX_VITSTR_SEQ_LEN : 0; }
const char *vitx_model_label(const vitx_model *m, int id) {
    if (!m) return nullptr;
    auto it = m->id2label.find(id);
    return it == m->id2label.end() ? nullptr : it->second.c_str();
}
int vitx_model_num_tensors(const vitx_model *m) { return m ? (int)m->tensors.size() : 0; }
int vitx_model_tensor_info(const vitx_model *m, int i, const char **name, int32_t *type, int64_t ne[4], size_t *nbytes) {
    if (!m || i < 0 || i >= (int)m->tensors.size()) return VITX_ERR_ARG;
    const vitx::HostTensor &t = m->tensors[i];
    if (name) *name = t.name.c_str();
    if (type) *type = t.type;
    if (ne) for (int k = 0; k < 4; ++k) ne[k] = t.ne[k];
    if (nbytes) *nbytes = t.raw.size();
    return VITX_OK;
}
int vitx_model_tensor_f32(const vitx_model *m, int i, float *out, size_t n) {
    if (!m || !out || i < 0 || i >= (int)m->tensors.size() || n != (size_t)m->tensors[i].nelements()) return VITX_ERR_ARG;
    m->tensors[i].decode_f32(out); return VITX_OK;
}

int vitx_topk(const float *probs, int C, int k, int32_t *idx, float *p) {
    if (!probs || !idx || C <= 0 || k <= 0) return VITX_ERR_ARG;
    if (k > C) k = C;
    // descending by probability (vit.cpp:1053-1057); ties broken by lower class id for determinism
    std::vector<int32_t> order((size_t)C);
    for (int i = 0; i < C; ++i) order[i] = i;
    std::partial_sort(order.begin(), order.begin() + k, order.end(), [&](int32_t a, int32_t b) { return probs[a] > probs[b] || (probs[a] == probs[b] && a < b); });
    for (int i = 0; i < k; ++i) { idx[i] = order[i]; if (p) p[i] = probs[order[i]]; }
    return VITX_OK;
}

}  // extern "C"
