// vit_api.cpp -- the reference's C++ entry points on top of the C ABI (see ../vit.h).
#include "../vit.h"

#include <stdio.h>
#include <stdlib.h>

vit_model::~vit_model() { vitx_model_free(handle); }
vit_state::~vit_state() { vitx_ctx_free(ctx); }

// vit.cpp:109-127 -- false + message on stderr when the file cannot be read or decoded
bool load_image_from_file(const std::string &fname, image_u8 &img) {
    uint8_t *data = nullptr; int nx = 0, ny = 0;
    if (vitx_image_load(fname.c_str(), &data, &nx, &ny) != VITX_OK) {
        fprintf(stderr, "%s: failed to load '%s'\n", __func__, fname.c_str());
        return false;
    }
    img.nx = nx; img.ny = ny;
    img.data.assign(data, data + (size_t)nx * ny * 3);
    vitx_image_free(data);
    return true;
}

// vit.cpp:308-712 -- false + message on stderr for any error
bool vit_model_load(const std::string &fname, vit_model &model) {
    printf("%s: loading model from '%s' - please wait\n", __func__, fname.c_str());
    vitx_model *h = nullptr;
    const int rc = vitx_model_load(fname.c_str(), &h);
    if (rc != VITX_OK) { fprintf(stderr, "%s: %s\n", __func__, vitx_last_error()); return false; }
    vitx_model_free(model.handle);
    model.handle = h;
    vitx_hparams hp;
    vitx_model_hparams(h, &hp);
    vit_hparams &o = model.hparams;
    o.hidden_size = hp.hidden_size; o.num_hidden_layers = hp.num_hidden_layers; o.num_attention_heads = hp.num_attention_heads;
    o.num_classes = hp.num_classes; o.patch_size = hp.patch_size; o.img_size = hp.img_size; o.ftype = hp.ftype; o.eps = hp.eps;
    o.id2label.clear();
    for (int i = 0; i < hp.num_classes; ++i) { const char *l = vitx_model_label(h, i); if (l) o.id2label[i] = l; }
    printf("%s: hidden_size            = %d\n", __func__, o.hidden_size);
    printf("%s: num_hidden_layers      = %d\n", __func__, o.num_hidden_layers);
    printf("%s: num_attention_heads    = %d\n", __func__, o.num_attention_heads);
    printf("%s: patch_size             = %d\n", __func__, o.patch_size);
    printf("%s: img_size               = %d\n", __func__, o.img_size);
    printf("%s: num_classes            = %d\n", __func__, o.num_classes);
    printf("%s: ftype                  = %d\n", __func__, o.ftype);
    return true;
}

// vit.cpp:289-305 -- false only for an unknown interpolation mode
bool vit_image_preprocess(const image_u8 &img, image_f32 &res, const vit_hparams &params) {
    int interp;
    if (params.interpolation == "bilinear") interp = VITX_BILINEAR;
    else if (params.interpolation == "bicubic") interp = VITX_BICUBIC;
    else { printf("Interpolation mode '%s' is not supported; returning 'false'...", params.interpolation.c_str()); return false; }
    const int S = params.n_img_size();
    res.nx = S; res.ny = S;
    res.data.resize((size_t)3 * S * S);
    return vitx_preprocess_u8(img.data.data(), img.nx, img.ny, S, interp, res.data.data()) == VITX_OK;
}

// The reference's vit_state carries no weights and can be reused with any model; ours caches a context that does, so the cache
// is keyed on the unique id of the parsed model it was built from (vit_model_load frees and replaces that handle on every call).
static int ensure_ctx(const vit_model &model, vit_state &state, int n) {
    if (state.ctx && state.ctx_model_uid == vitx_model_uid(model.handle) && vitx_ctx_max_batch(state.ctx) >= n) return VITX_OK;
    vitx_ctx_free(state.ctx); state.ctx = nullptr; state.ctx_model_uid = 0;
    state.max_batch = std::max(state.max_batch, n);
    const int rc = vitx_ctx_create(model.handle, state.device, state.max_batch, state.dtype, &state.ctx);
    if (rc == VITX_OK) state.ctx_model_uid = vitx_model_uid(model.handle);
    return rc;
}

int vit_predict_batch(const vit_model &model, vit_state &state, const image_f32 *imgs, int n, const vit_params &params,
                      std::vector<std::vector<std::pair<float, int>>> &predictions, bool print) {
    predictions.clear();
    if (!model.handle || !imgs || n <= 0) { fprintf(stderr, "%s: invalid argument\n", __func__); return 1; }
    if (vitx_model_seq_len(model.handle) > 0) { fprintf(stderr, "%s: this is a ViTSTR model (one-channel patch kernel): use vitstr_predict\n", __func__); return 1; }
    const int S = model.hparams.img_size, C = model.hparams.num_classes;
    for (int i = 0; i < n; ++i)
        if (imgs[i].nx != S || imgs[i].ny != S || imgs[i].data.size() != (size_t)3 * S * S) {      // GGML_ASSERT at vit.cpp:757
            fprintf(stderr, "%s: image %d is %dx%d, model expects %dx%d\n", __func__, i, imgs[i].nx, imgs[i].ny, S, S);
            abort();
        }
    if (ensure_ctx(model, state, n) != VITX_OK) { fprintf(stderr, "%s: failed to encode image: %s\n", __func__, vitx_last_error()); return 1; }
    std::vector<float> batch;
    const float *src = imgs[0].data.data();
    if (n > 1) {
        batch.resize((size_t)n * 3 * S * S);
        for (int i = 0; i < n; ++i) std::copy(imgs[i].data.begin(), imgs[i].data.end(), batch.begin() + (size_t)i * 3 * S * S);
        src = batch.data();
    }
    state.prediction.resize((size_t)n * C);
    if (vitx_forward(state.ctx, src, n, state.prediction.data(), nullptr) != VITX_OK) {
        fprintf(stderr, "%s: failed to encode image: %s\n", __func__, vitx_last_error());
        return 1;
    }
    predictions.resize(n);
    for (int b = 0; b < n; ++b) {
        auto &p = predictions[b];
        p.reserve(C);
        for (int i = 0; i < C; ++i) p.push_back(std::make_pair(state.prediction[(size_t)b * C + i], i));      // vit.cpp:1047-1050
        std::sort(p.begin(), p.end(), [](const std::pair<float, int> &a, const std::pair<float, int> &b2) { return a.first > b2.first; });   // vit.cpp:1053-1057
        if (print) {
            fprintf(stderr, "\n");
            for (int i = 0; i < params.topk && i < (int)p.size(); ++i)                                        // vit.cpp:1062-1067
                printf(" > %s : %.2f\n", model.hparams.id2label.at(p[i].second).c_str(), p[i].first);
        }
    }
    return 0;
}

// extensions/vitstr.cpp/vitstr.cpp:135-201
bool vitstr_image_preprocess(const image_u8 &img, image_f32 &res, const vit_hparams &params) {
    const int S = params.n_img_size();
    res.nx = S; res.ny = S;
    res.data.resize((size_t)S * S);
    return vitx_preprocess_vitstr_u8(img.data.data(), img.nx, img.ny, S, res.data.data()) == VITX_OK;
}

// extensions/vitstr.cpp/vitstr.cpp:970-1061 -- 0 ok / 1 failure
int vitstr_predict(const vit_model &model, vit_state &state, const image_f32 img1, const vit_params &params, std::string &text, double &score) {
    (void)params;
    text.clear(); score = 0.0;
    if (!model.handle) { fprintf(stderr, "%s: invalid argument\n", __func__); return 1; }
    const int R = vitx_model_seq_len(model.handle), S = model.hparams.img_size, C = model.hparams.num_classes;
    if (R <= 0) { fprintf(stderr, "%s: not a ViTSTR model (the patch kernel has 3 input channels): use vit_predict\n", __func__); return 1; }
    if (img1.nx != S || img1.ny != S || img1.data.size() != (size_t)S * S) { fprintf(stderr, "%s: image is %dx%d, model expects %dx%d grey\n", __func__, img1.nx, img1.ny, S, S); abort(); }
    if (ensure_ctx(model, state, 1) != VITX_OK) { fprintf(stderr, "%s: failed to encode image: %s\n", __func__, vitx_last_error()); return 1; }
    state.prediction.resize((size_t)R * C);
    if (vitx_forward(state.ctx, img1.data.data(), 1, state.prediction.data(), nullptr) != VITX_OK) {
        fprintf(stderr, "%s: failed to encode image: %s\n", __func__, vitx_last_error());
        return 1;
    }
    std::vector<int32_t> ids((size_t)R);
    int n_ids = 0;
    if (vitx_vitstr_decode(state.prediction.data(), R, C, ids.data(), &n_ids, &score) != VITX_OK) return 1;
    printf("------------------ \n");                                              // vitstr.cpp:1024
    for (int i = 0; i < n_ids; ++i) { const std::string &ch = model.hparams.id2label.at(ids[i]); printf("%s", ch.c_str()); text += ch; }   // :1050
    printf("\n");
    printf("score : %.2f \n", score);
    printf("------------------ \n");
    return 0;
}

// vit.cpp:1004-1075 -- 0 ok / 1 failure; prints the top-k lines like the reference does
int vit_predict(const vit_model &model, vit_state &state, const image_f32 img1, const vit_params &params,
                std::vector<std::pair<float, int>> &predictions) {
    std::vector<std::vector<std::pair<float, int>>> all;
    const int rc = vit_predict_batch(model, state, &img1, 1, params, all, /*print=*/true);
    predictions.clear();
    if (rc == 0) predictions = std::move(all[0]);
    return rc;
}

void print_usage(int argc, char **argv, const vit_params &params) {                                      // vit.cpp:943-956
    (void)argc;
    fprintf(stderr, "usage: %s [options]\n", argv[0]);
    fprintf(stderr, "\n");
    fprintf(stderr, "options:\n");
    fprintf(stderr, "  -h, --help              show this help message and exit\n");
    fprintf(stderr, "  -m FNAME, --model       model path (default: %s)\n", params.model.c_str());
    fprintf(stderr, "  -i FNAME, --inp         input file (default: %s)\n", params.fname_inp.c_str());
    fprintf(stderr, "  -t N, --threads         number of threads to use during computation (default: %d)\n", params.n_threads);
    fprintf(stderr, "  -k N, --topk            top k classes to print (default: %d)\n", params.topk);
    fprintf(stderr, "  -s SEED, --seed         RNG seed (default: -1)\n");
    fprintf(stderr, "  -e FLOAT, --epsilon     epsilon constant in Layer Norm layers (default: %f)\n", params.eps);
    fprintf(stderr, "\n");
}

bool vit_params_parse(int argc, char **argv, vit_params &params) {                                       // vit.cpp:958-1002
    for (int i = 1; i < argc; i++) {
        const std::string arg = argv[i];
        const bool has_val = i + 1 < argc;
        if ((arg == "-s" || arg == "--seed") && has_val) params.seed = std::stoi(argv[++i]);
        else if ((arg == "-t" || arg == "--threads") && has_val) params.n_threads = std::stoi(argv[++i]);
        else if ((arg == "-m" || arg == "--model") && has_val) params.model = argv[++i];
        else if ((arg == "-i" || arg == "--inp") && has_val) params.fname_inp = argv[++i];
        else if ((arg == "-k" || arg == "--topk") && has_val) params.topk = std::stoi(argv[++i]);
        else if ((arg == "-e" || arg == "--epsilon") && has_val) params.eps = std::stof(argv[++i]);
        else if (arg == "-h" || arg == "--help") { print_usage(argc, argv, params); exit(0); }
        else { fprintf(stderr, "error: unknown argument: %s\n", arg.c_str()); print_usage(argc, argv, params); exit(0); }
    }
    return true;
}
