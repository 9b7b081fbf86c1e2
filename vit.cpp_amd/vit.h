// vit.h -- C++ drop-in mirror of the reference's public API for the forward path.
//
// Same entry-point names, argument meaning and error behaviour as
// /root/reference/vit.h:115-124, re-declared so the reference's callers (main.cpp:57-98, tests/benchmark.cpp:57-122) build
// against libvitx.so once their ggml lines are removed (main.cpp:82-91 creates state.ctx / state.prediction by hand and :110
// frees model.ctx -- ggml objects this engine does not have; examples/vit_main.cpp is main.cpp without them):
//   * vit_model / vit_state keep their names but hold opaque engine handles instead of
//     ggml_tensor* / ggml_context* (the reference's fields are ggml internals);
//   * vit_predict still fills `predictions` with all (prob, class) pairs sorted
//     descending and prints the top-k lines to stdout (vit.cpp:1043-1067);
//   * vit_predict_batch is NEW (the reference has no batched call): n images per launch.
// Everything below is a thin wrapper over the C ABI in include/vitx.h.
#pragma once

#include <algorithm>
#include <cstdint>
#include <map>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../include/vitx.h"

struct vit_hparams {                      // vit.h:20-37
    int32_t hidden_size = 768;
    int32_t num_hidden_layers = 12;
    int32_t num_attention_heads = 12;
    int32_t num_classes = 1000;
    int32_t patch_size = 8;
    int32_t img_size = 224;
    int32_t ftype = 1;
    float eps = 1e-6f;
    std::string interpolation = "bicubic";
    std::map<int, std::string> id2label;

    int32_t n_enc_head_dim() const { return hidden_size / num_attention_heads; }   // vit.cpp:30-33
    int32_t n_img_size() const { return img_size; }                                // vit.cpp:35-38
    int32_t n_patch_size() const { return patch_size; }                            // vit.cpp:40-43
    int32_t n_img_embd() const { return img_size / patch_size; }                   // vit.cpp:45-48
};

struct vit_model {                        // vit.h:82-89 (tensor handles replaced by the engine's)
    vit_hparams hparams;
    vitx_model *handle = nullptr;         // parsed weight file (host)
    vit_model() = default;
    vit_model(const vit_model &) = delete;
    vit_model &operator=(const vit_model &) = delete;
    ~vit_model();
};

struct vit_state {                        // vit.h:72-80: per-caller mutable scratch
    vitx_ctx *ctx = nullptr;              // created lazily by vit_predict on `device`
    uint64_t ctx_model_uid = 0;           // vitx_model_uid of the parsed file `ctx` holds the weights of: a state reused with another (or a
                                          // reloaded) vit_model gets a fresh context instead of silently running the old weights.  An id,
                                          // not the pointer: `vit_model m; vit_model_load(f, m);` in a loop usually re-allocates the same address
    int device = 0;
    int max_batch = 1;                    // capacity of ctx; grown on demand by vit_predict_batch
    int dtype = VITX_F16;                 // MFMA operand type (VITX_F16 reproduces the reference's rounding)
    std::vector<float> prediction;        // class probabilities of the last call ([n][num_classes])
    vit_state() = default;
    vit_state(const vit_state &) = delete;
    vit_state &operator=(const vit_state &) = delete;
    ~vit_state();
};

struct image_u8 { int nx; int ny; std::vector<uint8_t> data; };     // vit.h:91-96
struct image_f32 { int nx; int ny; std::vector<float> data; };      // vit.h:98-103

struct vit_params {                       // vit.h:105-113
    int32_t seed = -1;
    int32_t n_threads = std::min(4, (int32_t)std::thread::hardware_concurrency());   // unused: the GPU does the work
    int32_t topk = 5;
    std::string model = "../ggml-model-f16.gguf";
    std::string fname_inp = "../assets/tench.jpg";
    float eps = 1e-6f;                    // parsed but unused by the forward, as in the reference (vit.cpp:984-987 vs 808)
};

bool load_image_from_file(const std::string &fname, image_u8 &img);                                    // vit.h:118 (stbi_load replaced by csrc/image_decode.cpp)
bool vit_model_load(const std::string &fname, vit_model &model);                                       // vit.h:120
bool vit_image_preprocess(const image_u8 &img, image_f32 &res, const vit_hparams &params);            // vit.h:119
int vit_predict(const vit_model &model, vit_state &state, const image_f32 img1, const vit_params &params,
                std::vector<std::pair<float, int>> &predictions);                                      // vit.h:122
int vit_predict_batch(const vit_model &model, vit_state &state, const image_f32 *imgs, int n, const vit_params &params,
                      std::vector<std::vector<std::pair<float, int>>> &predictions, bool print = false);
// ---- the ViTSTR scene-text extension (extensions/vitstr.cpp).  It is a separate program in the reference that re-uses the names
// vit_image_preprocess / vit_predict with different bodies (vitstr.h:115-119); here both programs live in one library, so the
// extension's two functions carry a vitstr_ prefix.  vit_model_load is shared: a file whose patch kernel has ONE input channel is
// a ViTSTR model (vitstr.cpp:482).
bool vitstr_image_preprocess(const image_u8 &img, image_f32 &res, const vit_hparams &params);         // vitstr.cpp:135-201: grey, [img_size][img_size]
// vitstr.cpp:970-1061: forward + greedy decode; prints the text and "score : x.xx" framed by dashed lines exactly like the
// extension, and also returns them.  state.prediction = [25][num_classes] probabilities.
int vitstr_predict(const vit_model &model, vit_state &state, const image_f32 img1, const vit_params &params, std::string &text, double &score);
void print_usage(int argc, char **argv, const vit_params &params);                                     // vit.h:123
bool vit_params_parse(int argc, char **argv, vit_params &params);                                      // vit.h:124
