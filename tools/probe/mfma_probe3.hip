// VMEM-cost probe: 16 MFMAs per step (two alternating accumulators) plus NL vector-memory instructions per step, each a
// 64-lane x 16 B read of an L2-resident buffer (MODE 0: global_load_dwordx4 into registers, consumed one step later;
// MODE 1: global_load_lds_dwordx4 into LDS, drained by s_waitcnt vmcnt(0) + barrier at the next step like the earlier LDS-DMA Winograd forms).
// usage: mfma_probe3   Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int NL, int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, const float* __restrict__ src, int iters, float a, float b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x16 acc[8];
    for (int f = 0; f < 8; ++f)
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* p = src + (blockIdx.x & 63) * 16384 + threadIdx.x * 4;       // 64 KiB windows of a 4 MiB buffer
    f32x4 v[NL > 0 ? NL : 1];
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < NL; ++k) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < NL; ++k) sum += v[k];
            }
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const float* s = p + ((it * 4 + g) & 3) * 4096 + k * 1024;
                if (MODE == 0) v[k] = *reinterpret_cast<const f32x4*>(s);
                else __builtin_amdgcn_global_load_lds(s, (lds_void*)(lds + ((g & 1) * 4 + wave) * 1024 + k * 256), 16, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                acc[g * 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g * 2], 0, 0, 0);
                acc[g * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g * 2 + 1], 0, 0, 0);
            }
        }
    }
    float s = sum.x + sum.y + sum.z + sum.w + lds[threadIdx.x];
    for (int f = 0; f < 8; ++f)
        for (int r = 0; r < 16; ++r) s += acc[f][r];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int NL, int MODE>
static void run(float* out, const float* src) {
    const int iters = 400;
    size_t lds = 78 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<NL, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int blocks = 256 * 2 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NL, MODE>), dim3(blocks), dim3(256), lds, 0, out, src, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NL, MODE>), dim3(blocks), dim3(256), lds, 0, out, src, iters, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 64.0 * 4096.0;
    printf("  %d x %s per 16 MFMAs -> %.1f TFLOP/s\n", NL, MODE ? "global_load_lds (vmcnt(0)+barrier per step)" : "global_load_dwordx4 (register, used next step)",
           flops / (ms * 1e-3) / 1e12);
}

int main() {
    float *out, *src;
    hipMalloc(&out, 4096);
    hipMalloc(&src, 8 << 20);
    hipMemset(src, 0, 8 << 20);
    run<0, 0>(out, src);
    run<1, 0>(out, src);
    run<2, 0>(out, src);
    run<4, 0>(out, src);
    run<8, 0>(out, src);
    run<0, 1>(out, src);
    run<1, 1>(out, src);
    run<2, 1>(out, src);
    run<4, 1>(out, src);
    return 0;
}
