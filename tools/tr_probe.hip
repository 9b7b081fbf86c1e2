// Probe of ds_read_b64_tr_b16 on gfx950: which LDS element lands in which (lane, slot).
// LDS holds u16 element e at byte 2e; lane L passes byte address base(L) = L * 8 (its own 4 contiguous elements).
// Prints, for every lane, the 4 element indices it received.   hipcc --offload-arch=gfx950 -o tr_probe.bin tr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short *out, int stride_elems) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    // lane -> address: row = lane / 4 (stride `stride_elems`), 4-element segment = lane % 4
    const int elem = (lane >> 2) * stride_elems + (lane & 3) * 4;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lds + elem));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short *d, h[256];
    hipMalloc(&d, 512);
    for (int stride : {16, 64}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("stride %d elements: lane L reads elements [(L/4)*stride + (L%%4)*4, +4)\n", stride);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 1) ? "\n" : "    ");
    }
    return 0;
}
