// model_file.h -- host-side parse of the reference's legacy-ggml ".gguf" weight file.
// Replaces the file half of vit_model_load (/root/reference/vit.cpp:308-712).
#pragma once
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

#include "../../include/vitx.h"

namespace vitx {

enum { T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q4_1 = 3, T_Q5_0 = 6, T_Q5_1 = 7, T_Q8_0 = 8 };
int type_block_bytes(int t);   // 0 for unknown
int type_block_elems(int t);

struct HostTensor {
    std::string name;
    int32_t type = 0;
    int32_t n_dims = 0;
    int64_t ne[4] = {1, 1, 1, 1};
    std::vector<uint8_t> raw;
    int64_t nelements() const { return ne[0] * ne[1] * ne[2] * ne[3]; }
    // exact decode to f32 (f16 widening / block dequantisation, ggml dequantize_row_*)
    void decode_f32(float *out) const;
};

}  // namespace vitx

struct vitx_model {
    vitx_hparams hp;
    uint64_t uid = 0;                                 // unique per successful vitx_model_load in this process (never reused: an address can be)
    int in_chans = 3;                                 // 1 = ViTSTR file (grey input, sequence head), from the patch kernel's shape
    std::map<int, std::string> id2label;
    std::vector<vitx::HostTensor> tensors;            // file order
    std::map<std::string, int> index;                 // name -> position
    const vitx::HostTensor *find(const std::string &n) const {
        auto it = index.find(n);
        return it == index.end() ? nullptr : &tensors[it->second];
    }
};

namespace vitx {
void set_error(const char *fmt, ...);
float f16_bits_to_f32(uint16_t h);
uint16_t f32_to_f16_bits(float f);     // round-to-nearest-even
uint16_t f32_to_bf16_bits(float f);    // round-to-nearest-even
}  // namespace vitx
