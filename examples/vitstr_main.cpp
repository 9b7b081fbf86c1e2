// vitstr_main.cpp -- the command-line flow of the reference's ViTSTR extension (/root/reference/extensions/vitstr.cpp/main.cpp:26-110)
// written against the drop-in C++ header vit.cpp_amd/vit.h: vit_params_parse -> vit_model_load -> load image -> grey preprocess ->
// forward + greedy decode.  The extension re-uses the names vit_image_preprocess / vit_predict with different bodies; in this
// library they are vitstr_image_preprocess / vitstr_predict (vit.h), everything else is shared with examples/vit_main.cpp.
// Build:  g++ -std=c++17 -O2 examples/vitstr_main.cpp -Ivit.cpp_amd -Lvit.cpp_amd -lvitx -Wl,-rpath,$PWD/vit.cpp_amd -o vitstr
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

    if (vit_params_parse(argc, argv, params) == false) return 1;
    if (params.seed < 0) params.seed = (int32_t)time(NULL);
    fprintf(stderr, "%s: seed = %d\n", __func__, params.seed);
    fprintf(stderr, "%s: n_threads = %d / %d\n", __func__, params.n_threads, (int32_t)std::thread::hardware_concurrency());

    const auto t_load_start = std::chrono::steady_clock::now();
    if (!vit_model_load(params.model.c_str(), model)) {                 // main.cpp:55-59
        fprintf(stderr, "%s: failed to load model from '%s'\n", __func__, params.model.c_str());
        return 1;
    }
    const double t_load_ms = ms_since(t_load_start);

    if (!load_image_from_file(params.fname_inp.c_str(), img0)) {        // main.cpp:65-69
        fprintf(stderr, "%s: failed to load image from '%s'\n", __func__, params.fname_inp.c_str());
        return 1;
    }
    fprintf(stderr, "%s: loaded image '%s' (%d x %d)\n", __func__, params.fname_inp.c_str(), img0.nx, img0.ny);

    if (vitstr_image_preprocess(img0, img1, model.hparams)) fprintf(stderr, "processed, out dims : (%d x %d)\n", img1.nx, img1.ny);   // main.cpp:73-76

    std::string text; double score = 0.0;
    if (vitstr_predict(model, state, img1, params, text, score) != 0) return 1;      // prints the decoded text and its score (vitstr.cpp:1024-1054)

    const double t_total_ms = ms_since(t_main_start);
    fprintf(stderr, "\n\n");
    fprintf(stderr, "%s:    model load time = %8.2f ms\n", __func__, t_load_ms);
    fprintf(stderr, "%s:    processing time = %8.2f ms\n", __func__, t_total_ms - t_load_ms);
    fprintf(stderr, "%s:    total time      = %8.2f ms\n", __func__, t_total_ms);
    return 0;
}
