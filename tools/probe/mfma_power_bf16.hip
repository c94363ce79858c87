// Sustained bare v_mfma_f32_32x32x16_bf16 loop on per-lane random operands (no memory traffic in the loop): the rate and the clock a bf16-split
// path (tools/split_bf16_numerics.py) would start from.   mfma_power_bf16 <seconds> <waves_per_simd 1|2>
// Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void burn(const float* in, float* out, int iters) {
    extern __shared__ float lds[];
    f32x16 acc[8];
    for (int f = 0; f < 8; ++f)
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    bf16x8 a[8], b[8];
    for (int k = 0; k < 8; ++k) {
        f32x4v va = *reinterpret_cast<const f32x4v*>(in + ((threadIdx.x * 32 + k * 4) & 4095));
        f32x4v vb = *reinterpret_cast<const f32x4v*>(in + ((threadIdx.x * 32 + k * 4 + 2048) & 4095));
        a[k] = __builtin_bit_cast(bf16x8, va);          // random bit patterns of the float table: random bf16 pairs
        b[k] = __builtin_bit_cast(bf16x8, vb);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int f = 0; f < 8; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + f) & 7], b[(s * 3 + f) & 7], acc[f], 0, 0, 0);
    }
    float s = 0.f;
    for (int f = 0; f < 8; ++f)
        for (int r = 0; r < 16; ++r) s += acc[f][r];
    if (s == 12345.f) out[threadIdx.x] = s + lds[0];
}
int main(int argc, char** argv) {
    double secs = argc > 1 ? atof(argv[1]) : 4.0;
    int wps = argc > 2 ? atoi(argv[2]) : 2;
    float h[4096], *in, *out;
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMalloc(&in, sizeof(h)); hipMalloc(&out, 4096); hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    size_t lds = wps == 1 ? 100 * 1024 : 78 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(burn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * wps * 4, iters = 20000;
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), lds, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // one 32x32x16 MFMA = 32768 FLOP
        printf("%.1f TFLOP/s executed (%.1f ms) = %.1f fp32-equivalent TFLOP/s at 6 products, %.1f at 3\n", (double)blocks * 4 * iters * 64.0 * 32768.0 / (ms * 1e-3) / 1e12, ms,
               (double)blocks * 4 * iters * 64.0 * 32768.0 / (ms * 1e-3) / 1e12 / 6.0, (double)blocks * 4 * iters * 64.0 * 32768.0 / (ms * 1e-3) / 1e12 / 3.0); fflush(stdout);
    }
    return 0;
}
