// vit_main.cpp -- the reference's command-line flow (/root/reference/main.cpp:25-112) written against the drop-in C++
// header vit.cpp_amd/vit.h: vit_params_parse -> vit_model_load -> load image -> vit_image_preprocess -> vit_predict,
// same stdout / stderr lines.  Differences a maintainer would see when switching main.cpp over:
//   * no ggml_init / ggml_free: vit_state owns an engine context that vit_predict creates on first use;
//   * load_image_from_file (stb_image in the reference, inside the absent ggml tree) is libvitx.so's own JPEG / PNG / PPM decoder.
// Build:  g++ -std=c++17 -O2 examples/vit_main.cpp -Ivit.cpp_amd -Lvit.cpp_amd -lvitx -Wl,-rpath,$PWD/vit.cpp_amd -o vit
#include <chrono>
#include <cstdio>
#include <cstring>

#include "vit.h"

int main(int argc, char **argv) {
    const auto t_main_start = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    vit_params params;
    image_u8 img0;
    image_f32 img1;
    vit_model model;
    vit_state state;
    std::vector<std::pair<float, int>> predictions;

    if (vit_params_parse(argc, argv, params) == false) return 1;
    if (params.seed < 0) params.seed = (int32_t)time(NULL);
    fprintf(stderr, "%s: seed = %d\n", __func__, params.seed);
    fprintf(stderr, "%s: n_threads = %d / %d\n", __func__, params.n_threads, (int32_t)std::thread::hardware_concurrency());

    const auto t_load_start = std::chrono::steady_clock::now();
    if (!vit_model_load(params.model.c_str(), model)) {                 // main.cpp:57-61
        fprintf(stderr, "%s: failed to load model from '%s'\n", __func__, params.model.c_str());
        return 1;
    }
    const double t_load_ms = ms_since(t_load_start);

    if (!load_image_from_file(params.fname_inp.c_str(), img0)) {        // main.cpp:67-73
        fprintf(stderr, "%s: failed to load image from '%s'\n", __func__, params.fname_inp.c_str());
        return 1;
    }
    fprintf(stderr, "%s: loaded image '%s' (%d x %d)\n", __func__, params.fname_inp.c_str(), img0.nx, img0.ny);

    if (vit_image_preprocess(img0, img1, model.hparams)) fprintf(stderr, "processed, out dims : (%d x %d)\n", img1.nx, img1.ny);

    if (vit_predict(model, state, img1, params, predictions) != 0) return 1;     // prints the top-k lines (vit.cpp:1062-1067)

    const double t_total_ms = ms_since(t_main_start);
    fprintf(stderr, "\n\n");
    fprintf(stderr, "%s:    model load time = %8.2f ms\n", __func__, t_load_ms);
    fprintf(stderr, "%s:    processing time = %8.2f ms\n", __func__, t_total_ms - t_load_ms);
    fprintf(stderr, "%s:    total time      = %8.2f ms\n", __func__, t_total_ms);
    return 0;
}
