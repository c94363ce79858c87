// Issue-interference probe: 16 MFMAs per step (2 alternating accumulators, like the Winograd kernel) plus, per step,
// NV independent VALU fmas, ND ds_read_b128 (+ consumed by one VALU each) and optionally a barrier.
// usage: mfma_probe2 <wgs_per_cu>   Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int ND, int BAR>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, float a, float b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x16 acc[8];
    for (int f = 0; f < 8; ++f)
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const float* lp = lds + threadIdx.x * 4;
    f32x4 dsum = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (BAR) __syncthreads();
            f32x4 d[ND > 0 ? ND : 1];
#pragma unroll
            for (int k = 0; k < ND; ++k) d[k] = *reinterpret_cast<const f32x4*>(lp + ((k + g * ND + it) & 15) * 1024);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                acc[g * 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g * 2], 0, 0, 0);
                acc[g * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g * 2 + 1], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < (NV + 7) / 8; ++k)
                    if (s * ((NV + 7) / 8) + k < NV) v[(s + k) & 7] = fmaf(v[(s + k) & 7], b, a);
            }
#pragma unroll
            for (int k = 0; k < ND; ++k) dsum += d[k];
            if (NV > 0 || ND > 0) {      // spread the side work between the MFMAs (VALU 0x2, MFMA 0x8, DS read 0x100)
                __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (NV + ND * 4 + 7) / 8, 0);
                }
            }
        }
    }
    float s = dsum.x + dsum.y + dsum.z + dsum.w;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int f = 0; f < 8; ++f)
        for (int r = 0; r < 16; ++r) s += acc[f][r];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int NV, int ND, int BAR>
static void run(int wps, int iters, float* out) {
    size_t lds = wps == 1 ? 100 * 1024 : 78 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<NV, ND, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int blocks = 256 * wps * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NV, ND, BAR>), dim3(blocks), dim3(256), lds, 0, out, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NV, ND, BAR>), dim3(blocks), dim3(256), lds, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 64.0 * 4096.0;
    printf("  per 16 MFMAs: %2d VALU, %2d ds_read_b128, barrier %d -> %.1f TFLOP/s\n", NV, ND, BAR, flops / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 2;
    int iters = 400;
    float* out;
    hipMalloc(&out, 4096);
    printf("workgroups (4 waves) per CU: %d\n", wps);
    run<0, 0, 0>(wps, iters, out);
    run<8, 0, 0>(wps, iters, out);
    run<16, 0, 0>(wps, iters, out);
    run<32, 0, 0>(wps, iters, out);
    run<64, 0, 0>(wps, iters, out);
    run<0, 4, 0>(wps, iters, out);
    run<0, 8, 0>(wps, iters, out);
    run<0, 16, 0>(wps, iters, out);
    run<0, 8, 1>(wps, iters, out);
    run<16, 8, 1>(wps, iters, out);
    run<32, 16, 1>(wps, iters, out);
    return 0;
}
