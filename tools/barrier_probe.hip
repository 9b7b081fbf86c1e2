// Micro-probe: what does one s_barrier per 16-MFMA burst cost on gfx950?   hipcc --offload-arch=gfx950 -O3 tools/barrier_probe.hip -o /tmp/barrier_probe
// Each wave owns 8 accumulator tiles (32x32 f32) and runs `slots` bursts of 16 independent MFMAs, optionally followed by a
// workgroup barrier, with NW waves per workgroup and one workgroup per CU.  Reports ns per burst-slot and the MFMA rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(float *out, int slots, int nmfma) {
    f16v acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
    for (int s = 0; s < slots; ++s) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        if (MODE == 1) asm volatile("s_barrier" ::: "memory");
        if (MODE == 2) { asm volatile("s_setprio 0\n\ts_barrier\n\ts_setprio 3" ::: "memory"); }
        if (MODE == 3) { if ((s & 1) == 1) asm volatile("s_barrier" ::: "memory"); }
        if (MODE == 4) { asm volatile("s_nop 0" ::: "memory"); }
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][15];
    if (t == 123.456f) out[0] = t;
}

template <int MODE>
static void run(const char *name, int nw, float *d) {
    const int slots = 4096, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, nw * 64>>>(d, 64, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<blocks, nw * 64>>>(d, slots, 16);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns_slot = ms * 1e6 / slots;
    const double tf = 2.0 * 32 * 32 * 16 * 16.0 * nw * blocks * slots / (ms * 1e-3) / 1e12;
    printf("%-28s waves/WG %d: %7.1f ns per slot (16 MFMA/wave)  %7.1f TF/s\n", name, nw, ns_slot, tf);
}

int main() {
    float *d; hipMalloc(&d, 1024);
    for (int nw : {4, 8}) {
        run<0>("no barrier", nw, d);
        run<1>("s_barrier per slot", nw, d);
        run<2>("s_barrier + setprio", nw, d);
        run<3>("s_barrier every 2nd slot", nw, d);
    }
    return 0;
}
