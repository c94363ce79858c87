// Ceiling probe for v_mfma_f32_32x32x2_f32: pure MFMA loops, no memory traffic.
//   mode 0: 8 independent accumulators, back to back        mode 1: + s_barrier every 16 MFMAs
//   mode 2: 2 accumulators alternating (the per-step pattern of the Winograd kernel)   mode 3: mode 2 + barrier every 16
// usage: mfma_probe <waves_per_simd 1|2|3> ; prints executed TFLOP/s per mode.   Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a, float b) {
    extern __shared__ float lds[];
    f32x16 acc[8];
    for (int f = 0; f < 8; ++f)
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (MODE == 1 || MODE == 3) __syncthreads();
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (MODE >= 2) {
                    acc[g * 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g * 2], 0, 0, 0);
                    acc[g * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g * 2 + 1], 0, 0, 0);
                } else {
                    acc[(2 * s) & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[(2 * s) & 7], 0, 0, 0);
                    acc[(2 * s + 1) & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[(2 * s + 1) & 7], 0, 0, 0);
                }
            }
        }
    }
    float s = 0.f;
    for (int f = 0; f < 8; ++f)
        for (int r = 0; r < 16; ++r) s += acc[f][r];
    if (s == 12345.f) out[threadIdx.x] = s + lds[0];
}

template <int MODE>
static double run(int wps, int iters, float* out) {
    // one 256-thread workgroup = 1 wave per SIMD; wps workgroups per CU via the LDS allocation (160 KiB / wps)
    size_t lds = wps == 1 ? 100 * 1024 : wps == 2 ? 78 * 1024 : 50 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int blocks = 256 * wps * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), lds, 0, out, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), lds, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 /*waves*/ * iters * 64.0 * 4096.0;
    return flops / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 2;
    int iters = argc > 2 ? atoi(argv[2]) : 400;
    float* out;
    hipMalloc(&out, 4096);
    printf("waves/SIMD %d: mode0 %.1f  mode1 %.1f  mode2 %.1f  mode3 %.1f TFLOP/s (executed)\n", wps, run<0>(wps, iters, out), run<1>(wps, iters, out),
           run<2>(wps, iters, out), run<3>(wps, iters, out));
    return 0;
}
