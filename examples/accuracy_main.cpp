// accuracy_main.cpp -- the reference's top-1 accuracy harness (/root/reference/tests/benchmark.cpp:34-150) written against the
// drop-in C++ header vit.cpp_amd/vit.h:
//     accuracy <model_path> <dataset_dir> <num_images_per_class> [output_file]
// <dataset_dir> holds one sub-directory per class (named after the class), <dataset_dir>/../classnames.json is the JSON array of
// class names in class-index order (benchmark.cpp:52-54); every *.JPEG of every class directory is decoded, preprocessed and
// classified, "file,class,predicted class" goes to the output file (default predictions.txt) and "Top-1 Accuracy: x%" to stdout.
// What differs from the reference, by design:
//   * no ggml_init / state.prediction set-up (benchmark.cpp:69-81): vit_state owns an engine context;
//   * images are classified in BATCHES (vit_predict_batch; the reference calls vit_predict once per image): the engine's throughput
//     comes from batching, results per image are identical to one-by-one calls (images are independent in every kernel);
//   * <num_images_per_class> is honoured (the reference parses it and never uses it: images_processed is never incremented);
//     0 or a negative number = every image, which is what the reference does for any value;
//   * no nlohmann/json (not in the tree): classnames.json is a flat array of strings and is read by a 20-line scanner.
// Build:  g++ -std=c++17 -O2 examples/accuracy_main.cpp -Ivit.cpp_amd -Lvit.cpp_amd -lvitx -Wl,-rpath,$PWD/vit.cpp_amd -o accuracy
#include <algorithm>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "vit.h"

namespace fs = std::filesystem;

// ["tench", "goldfish", ...] -> vector (handles \" \\ \/ \n \t \uXXXX for the BMP; that is all ImageNet's names need)
static std::vector<std::string> read_class_names(const std::string &filename) {
    std::ifstream file(filename);
    if (!file.is_open()) { std::cerr << "Cannot open file: " << filename << std::endl; return {}; }
    std::stringstream ss; ss << file.rdbuf();
    const std::string s = ss.str();
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size() && s[i] != '[') ++i;
    for (++i; i < s.size();) {
        while (i < s.size() && s[i] != '"' && s[i] != ']') ++i;
        if (i >= s.size() || s[i] == ']') break;
        std::string cur;
        for (++i; i < s.size() && s[i] != '"'; ++i) {
            if (s[i] != '\\' || i + 1 >= s.size()) { cur += s[i]; continue; }
            const char e = s[++i];
            if (e == 'n') cur += '\n';
            else if (e == 't') cur += '\t';
            else if (e == 'u' && i + 4 < s.size()) {
                const unsigned cp = (unsigned)std::stoul(s.substr(i + 1, 4), nullptr, 16); i += 4;
                if (cp < 0x80) cur += (char)cp;
                else if (cp < 0x800) { cur += (char)(0xC0 | (cp >> 6)); cur += (char)(0x80 | (cp & 0x3F)); }
                else { cur += (char)(0xE0 | (cp >> 12)); cur += (char)(0x80 | ((cp >> 6) & 0x3F)); cur += (char)(0x80 | (cp & 0x3F)); }
            } else cur += e;       // \" \\ \/
        }
        out.push_back(cur);
        ++i;
    }
    return out;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        std::cerr << "usage: " << argv[0] << " <model_path> <dataset_dir> <num_images_per_class> [output_file]" << std::endl;
        return 1;
    }
    const std::string model_path = argv[1], dataset_dir = argv[2];
    const int num_images_per_class = std::stoi(argv[3]);
    const std::string output_file = (argc == 5) ? argv[4] : "predictions.txt";
    const int batch = 64;

    vit_state state;
    vit_params params;
    vit_model model;

    const fs::path classnames_path = fs::path(dataset_dir).parent_path() / "classnames.json";
    const std::vector<std::string> CLASS_NAMES = read_class_names(classnames_path.string());

    if (!vit_model_load(model_path, model)) { std::cerr << "Failed to load model from " << model_path << std::endl; return 1; }
    std::ofstream out_file(output_file);
    if (!out_file) { std::cerr << "Failed to open output file: " << output_file << std::endl; return 1; }

    int total_images = 0, correct_predictions = 0;
    std::vector<image_f32> pending;
    std::vector<std::pair<std::string, std::string>> pending_meta;      // (file name, true class)
    auto flush = [&]() -> bool {
        if (pending.empty()) return true;
        std::vector<std::vector<std::pair<float, int>>> preds;
        if (vit_predict_batch(model, state, pending.data(), (int)pending.size(), params, preds, false) != 0) {
            std::cerr << "Inference failed for a batch of " << pending.size() << " images" << std::endl;
            pending.clear(); pending_meta.clear();
            return false;
        }
        for (size_t k = 0; k < preds.size(); ++k) {
            const int top = preds[k].front().second;
            const std::string predicted = top >= 0 && (size_t)top < CLASS_NAMES.size() ? CLASS_NAMES[top] : std::to_string(top);
            if (pending_meta[k].second == predicted) ++correct_predictions;
            ++total_images;
            out_file << pending_meta[k].first << "," << pending_meta[k].second << "," << predicted << std::endl;
        }
        pending.clear(); pending_meta.clear();
        return true;
    };

    std::vector<fs::path> class_dirs;
    for (const auto &e : fs::directory_iterator(dataset_dir)) if (e.is_directory()) class_dirs.push_back(e.path());
    std::sort(class_dirs.begin(), class_dirs.end());        // directory_iterator's order is unspecified: a stable output file
    for (const fs::path &cd : class_dirs) {
        const std::string class_name = cd.filename().string();
        std::vector<fs::path> files;
        for (const auto &ie : fs::directory_iterator(cd)) if (ie.path().extension() == ".JPEG") files.push_back(ie.path());
        std::sort(files.begin(), files.end());
        int images_processed = 0;
        for (const fs::path &ip : files) {
            if (num_images_per_class > 0 && images_processed >= num_images_per_class) break;
            image_u8 img;
            if (!load_image_from_file(ip.string(), img)) { std::cerr << "Failed to load image from " << ip.string() << std::endl; continue; }
            image_f32 processed;
            if (!vit_image_preprocess(img, processed, model.hparams)) { std::cerr << "Error in preprocessing image " << ip.string() << std::endl; continue; }
            pending.push_back(std::move(processed));
            pending_meta.emplace_back(ip.filename().string(), class_name);
            ++images_processed;
            if ((int)pending.size() == batch) flush();
        }
    }
    flush();

    const double accuracy = total_images ? static_cast<double>(correct_predictions) / total_images : 0.0;
    std::cout << "Top-1 Accuracy: " << accuracy * 100.0 << "%" << std::endl;
    out_file.close();
    return 0;
}
