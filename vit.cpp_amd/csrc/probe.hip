// probe.hip -- measurement probes exported through the C ABI (nothing here is on the forward path).
//
// vitx_probe_mfma: what the matrix pipe of THIS device sustains under its power management, independent of any GEMM structure:
// every CU runs 8 waves (2 per SIMD, like the GEMM kernels) of back-to-back v_mfma_f32_16x16x32 -- the instruction the GEMM kernels
// use -- on register operands: no LDS, no memory, no barriers.  On MI355X the result depends on the operand VALUES: zero-filled
// operands run at the nominal clock (~2480 TFLOP/s bf16), uniform random operands make the package hit its 1400 W cap and the
// clock drops to ~1.98 GHz (~2030 TFLOP/s bf16; the 32x32x16 form costs 11 % more energy per flop: ~1.76 GHz, ~1815 TFLOP/s --
// tools/mfma_ceiling.hip measures both, profiles/r02f/mfma_ceiling_both_shapes.txt).  bench.py reports the random-operand number
// next to the nominal peak so the roofline fraction can be read against what the silicon can actually deliver on non-trivial data.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vitx.h"
#include "device_common.h"
#include "model_file.h"

namespace vitx {
namespace {

__device__ __forceinline__ uint32_t probe_hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ float probe_unit(uint32_t s) { return (float)(probe_hash(s) >> 8) * (1.0f / 8388608.0f) - 1.0f; }

struct ProbeOut { unsigned long long cycles, realtime; float sink; int pad; };

template <typename T>
__global__ __launch_bounds__(512, 2) void mfma_probe_kernel(ProbeOut *out, int iters, int fill) {
    typedef typename Elem<T>::v8 v8;
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    v8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (T)(fill == 0 ? 0.0f : (fill == 1 ? 0.5f : probe_unit(tid * 64 + i * 8 + e)));
            b[i][e] = (T)(fill == 0 ? 0.0f : (fill == 1 ? 0.5f : probe_unit(tid * 64 + 32 + i * 8 + e) * 0.05f));
        }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = Elem<T>::mfma16(a[(i + k) & 3], b[i & 3], acc[i]);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (threadIdx.x == 0) out[blockIdx.x] = ProbeOut{c1 - c0, r1 - r0, s, 0};
}

}  // namespace
}  // namespace vitx

using namespace vitx;

extern "C" int vitx_probe_mfma(int device, int dtype, int fill, double target_ms, double *tflops, double *clock_mhz) {
    if ((dtype != VITX_F16 && dtype != VITX_BF16) || fill < 0 || fill > 2 || target_ms <= 0 || !tflops) { set_error("vitx_probe_mfma: invalid argument"); return VITX_ERR_ARG; }
    if (hipSetDevice(device) != hipSuccess) { set_error("vitx_probe_mfma: no such HIP device %d", device); return VITX_ERR_HIP; }
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device);
    ProbeOut *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(ProbeOut) * n_cu) != hipSuccess) { set_error("vitx_probe_mfma: hipMalloc failed"); return VITX_ERR_HIP; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int iters = 2000; float ms = 0.0f;
    for (int pass = 0; pass < 2; ++pass) {          // pass 0 calibrates the iteration count to the requested duration
        (void)hipEventRecord(e0, 0);
        if (dtype == VITX_F16) hipLaunchKernelGGL((mfma_probe_kernel<_Float16>), dim3(n_cu), dim3(512), 0, 0, d, iters, fill);
        else hipLaunchKernelGGL((mfma_probe_kernel<__bf16>), dim3(n_cu), dim3(512), 0, 0, d, iters, fill);
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.0f) {
            (void)hipFree(d); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            set_error("vitx_probe_mfma: kernel failed: %s", hipGetErrorString(hipGetLastError())); return VITX_ERR_HIP;
        }
        if (pass == 0) { const double want = (double)iters * target_ms / ms; iters = want < 100 ? 100 : (want > 2e9 ? 2000000000 : (int)want); }
    }
    ProbeOut *h = new ProbeOut[n_cu];
    (void)hipMemcpy(h, d, sizeof(ProbeOut) * n_cu, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < n_cu; ++i) { cyc += (double)h[i].cycles; rt += (double)h[i].realtime; }
    delete[] h;
    *tflops = 2.0 * 16 * 16 * 32 * (double)iters * 64 * 8 * n_cu / (ms * 1e-3) / 1e12;      // 64 MFMAs per iteration per wave, 8 waves per CU
    if (clock_mhz) *clock_mhz = rt > 0 ? cyc / (rt / 100.0) : 0.0;                            // s_memrealtime ticks at 100 MHz
    (void)hipFree(d); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return VITX_OK;
}
