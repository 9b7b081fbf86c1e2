// quant.hip -- block-quantised weights on the device (gfx950).
//
// The reference keeps the 2-D "*weight" tensors of a quantised file in ggml block form through compute (ftype -> wtype,
// /root/reference/vit.cpp:384-414; tensors allocated in that type, :645-678; the offline tool that writes them,
// quantize.cpp:271-303).  Here the blocks are what lives in HBM (4.5 / 5 / 5.5 / 6 / 8.5 bits per weight); this file expands
// them on the device, just in time:
//   * launch_dequant: one launch expands the (up to four) matrices of an encoder layer into an operand-type scratch that the
//     wide-tile GEMMs then stream like any other weight matrix.  The scratch is per sub-batch stream and re-used by every
//     layer (ViT-B: 14 MB, resident in the 256 MB Infinity Cache), so HBM only ever holds and serves the blocks.
//   * the small-batch GEMM expands q4_0 blocks inside its own LDS-fill path instead (gemm_nt_kernel<.., Q4 = true> in kernels.hip).
// Values are exactly HostTensor::decode_f32 (model_file.cpp; ggml's dequantize_row_*), rounded ONCE to the operand type with
// round-to-nearest-even -- bit-identical to expanding on the host at upload, which the tests assert.
#include <stdint.h>

#include "device_common.h"
#include "kernels.h"

namespace vitx {

namespace {

struct DequantArgs { DequantJob job[4]; int first[5]; int njobs; };   // first[j] = global id of job j's first (padded) block

template <int QT> struct QInfo;
template <> struct QInfo<QT_Q4_0> { [[maybe_unused]] static constexpr int BB = 16; };   // nibble plane only: the scales are a separate plane (dequant_q4_0_kernel expands this type)
template <> struct QInfo<QT_Q4_1> { static constexpr int BB = 20; };
template <> struct QInfo<QT_Q5_0> { static constexpr int BB = 22; };
template <> struct QInfo<QT_Q5_1> { static constexpr int BB = 24; };
template <> struct QInfo<QT_Q8_0> { static constexpr int BB = 34; };

__device__ __forceinline__ float h2f(uint16_t bits) { return (float)__builtin_bit_cast(_Float16, bits); }

// One thread per block of 32 weights.  Block bytes are read as 16-bit words (every block type is 2-byte aligned in the file
// layout; q4_0's nibble plane is 16-byte aligned and read as one dwordx4).
template <typename T, int QT>
__global__ __launch_bounds__(256) void dequant_kernel(DequantArgs a) {
    constexpr int BB = QInfo<QT>::BB;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= a.first[a.njobs]) return;
    int j = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) if (k < a.njobs && gid >= a.first[k]) j = k;
    const DequantJob &job = a.job[j];
    const int b = gid - a.first[j];
    const int row = b / job.nbk, kb = b - row * job.nbk;
    float o[32];
    if (row >= job.N) {
#pragma unroll
        for (int i = 0; i < 32; ++i) o[i] = 0.0f;
    } else {
        const size_t blk = (size_t)row * job.nbk + kb;
        uint16_t w[BB / 2];
        if constexpr (QT == QT_Q4_0) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 q = *(const u32x4 *)((const unsigned char *)job.src + blk * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) { w[2 * i] = (uint16_t)(q[i] & 0xffffu); w[2 * i + 1] = (uint16_t)(q[i] >> 16); }
        } else {
            const uint16_t *p = (const uint16_t *)((const unsigned char *)job.src + blk * BB);
#pragma unroll
            for (int i = 0; i < BB / 2; ++i) w[i] = p[i];
        }
        auto byte_at = [&](int i) -> unsigned { return (w[i >> 1] >> ((i & 1) * 8)) & 0xffu; };      // byte i of the block
        if constexpr (QT == QT_Q4_0) {
            const float d = h2f(((const uint16_t *)job.scales)[blk]);
#pragma unroll
            for (int i = 0; i < 16; ++i) { const unsigned q = byte_at(i); o[i] = (float)((int)(q & 15u) - 8) * d; o[i + 16] = (float)((int)(q >> 4) - 8) * d; }
        } else if constexpr (QT == QT_Q4_1) {
            const float d = h2f(w[0]), m = h2f(w[1]);
#pragma unroll
            for (int i = 0; i < 16; ++i) { const unsigned q = byte_at(4 + i); o[i] = (float)(q & 15u) * d + m; o[i + 16] = (float)(q >> 4) * d + m; }
        } else if constexpr (QT == QT_Q5_0) {
            const float d = h2f(w[0]);
            const unsigned qh = (unsigned)w[1] | ((unsigned)w[2] << 16);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned q = byte_at(6 + i), h0 = ((qh >> i) << 4) & 0x10u, h1 = (qh >> (i + 12)) & 0x10u;
                o[i] = (float)((int)((q & 15u) | h0) - 16) * d; o[i + 16] = (float)((int)((q >> 4) | h1) - 16) * d;
            }
        } else if constexpr (QT == QT_Q5_1) {
            const float d = h2f(w[0]), m = h2f(w[1]);
            const unsigned qh = (unsigned)w[2] | ((unsigned)w[3] << 16);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned q = byte_at(8 + i), h0 = ((qh >> i) << 4) & 0x10u, h1 = (qh >> (i + 12)) & 0x10u;
                o[i] = (float)((q & 15u) | h0) * d + m; o[i + 16] = (float)((q >> 4) | h1) * d + m;
            }
        } else {   // QT_Q8_0
            const float d = h2f(w[0]);
#pragma unroll
            for (int i = 0; i < 32; ++i) o[i] = (float)(int)(signed char)byte_at(2 + i) * d;
        }
    }
    typename Elem<T>::v8 *dst = (typename Elem<T>::v8 *)((T *)job.dst + ((size_t)row * job.nbk + kb) * 32);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        typename Elem<T>::v8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (T)o[s * 8 + e];
        dst[s] = v;
    }
}

// q4_0 (BASELINE config 5: one launch per layer and sub-batch stream in front of the qkv GEMM) with FOUR lanes per block: lane quarter q
// expands elements 8 q .. 8 q + 7 -- bytes 8 (q & 1) .. + 7 of the nibble plane, low nibbles for q < 2, high nibbles above (ggml's
// dequantize_row_q4_0 order) -- and writes ONE 16-byte piece, so a wave's store instruction covers 1 KiB of consecutive bytes instead of
// 64 pieces at a 64-byte stride.  Same arithmetic per element, (float)(nibble - 8) * d rounded once: the same bits as dequant_kernel.
template <typename T>
__global__ __launch_bounds__(256) void dequant_q4_0_kernel(DequantArgs a) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int blk_id = gid >> 2, q = gid & 3;
    if (blk_id >= a.first[a.njobs]) return;
    int j = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) if (k < a.njobs && blk_id >= a.first[k]) j = k;
    const DequantJob &job = a.job[j];
    const int b = blk_id - a.first[j];
    const int row = b / job.nbk, kb = b - row * job.nbk;
    typename Elem<T>::v8 v;
    if (row >= job.N) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (T)0.0f;
    } else {
        const size_t blk = (size_t)row * job.nbk + kb;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 w = *(const u32x2 *)((const unsigned char *)job.src + blk * 16 + (q & 1) * 8);
        const float d = h2f(((const uint16_t *)job.scales)[blk]);
        const int sh = (q >> 1) * 4;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned byte = (w[e >> 2] >> ((e & 3) * 8)) & 0xffu;
            v[e] = (T)((float)((int)((byte >> sh) & 15u) - 8) * d);
        }
    }
    *(typename Elem<T>::v8 *)((T *)job.dst + ((size_t)row * job.nbk + kb) * 32 + q * 8) = v;
}

template <typename T, int QT>
hipError_t launch_dequant_inst(const DequantArgs &a, hipStream_t stream) {
    const int total = a.first[a.njobs];
    if (total <= 0) return hipSuccess;
    if constexpr (QT == QT_Q4_0) hipLaunchKernelGGL((dequant_q4_0_kernel<T>), dim3((int)(((long)total * 4 + 255) / 256)), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((dequant_kernel<T, QT>), dim3((total + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_dequant_t(int qtype, const DequantArgs &a, hipStream_t stream) {
    switch (qtype) {
    case QT_Q4_0: return launch_dequant_inst<T, QT_Q4_0>(a, stream);
    case QT_Q4_1: return launch_dequant_inst<T, QT_Q4_1>(a, stream);
    case QT_Q5_0: return launch_dequant_inst<T, QT_Q5_0>(a, stream);
    case QT_Q5_1: return launch_dequant_inst<T, QT_Q5_1>(a, stream);
    case QT_Q8_0: return launch_dequant_inst<T, QT_Q8_0>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

hipError_t launch_dequant(int dtype, int qtype, const DequantJob *jobs, int njobs, hipStream_t stream) {
    if (njobs < 1 || njobs > 4) return hipErrorInvalidValue;
    DequantArgs a{};
    a.njobs = njobs;
    long total = 0;
    for (int j = 0; j < njobs; ++j) {
        if (!jobs[j].src || !jobs[j].dst || jobs[j].nbk <= 0 || jobs[j].n_pad < jobs[j].N || (qtype == QT_Q4_0 && !jobs[j].scales)) return hipErrorInvalidValue;
        a.job[j] = jobs[j]; a.first[j] = (int)total;
        total += (long)jobs[j].n_pad * jobs[j].nbk;
        if (total > 0x1fffffffL) return hipErrorInvalidValue;      // (x 4 lanes per block in the q4_0 kernel's thread index)
    }
    for (int j = njobs; j <= 4; ++j) a.first[j] = (int)total;
    return dtype == DT_F16 ? launch_dequant_t<_Float16>(qtype, a, stream) : launch_dequant_t<__bf16>(qtype, a, stream);
}

}  // namespace vitx
