// checks the packed-fp32 asm helpers of conv_wino6.hip against plain arithmetic.  hipcc --offload-arch=gfx950 -O3 pk_probe.hip -o pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma4(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_fma_f32 %0, %1, 4.0, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_fnma4(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_fma_f32 %0, %1, -4.0, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_fma2(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_fnma2(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_fma_f32 %0, %1, -2.0, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_fnma5(f32x2 a, f32x2 b, f32x2 five) {
    f32x2 d; asm("v_pk_fma_f32 %0, %1, %3, %2 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "s"(five)); return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__global__ void k(const float* in, float* out) {
    f32x2 a = {in[threadIdx.x * 4], in[threadIdx.x * 4 + 1]}, b = {in[threadIdx.x * 4 + 2], in[threadIdx.x * 4 + 3]};
    const f32x2 five = {5.0f, 5.0f};
    f32x2 r[7] = {pk_fma4(a, b), pk_fnma4(a, b), pk_fma2(a, b), pk_fnma2(a, b), pk_fnma5(a, b, five), pk_add(a, b), pk_sub(a, b)};
    for (int i = 0; i < 7; ++i) { out[(threadIdx.x * 7 + i) * 2] = r[i].x; out[(threadIdx.x * 7 + i) * 2 + 1] = r[i].y; }
}
int main() {
    float h[256], *d, *o, ho[64 * 14];
    for (int i = 0; i < 256; ++i) h[i] = 0.37f * i - 11.f;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o); hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    const char* names[7] = {"4a+b", "b-4a", "2a+b", "b-2a", "b-5a", "a+b", "a-b"};
    int bad = 0;
    for (int t = 0; t < 64; ++t) for (int i = 0; i < 7; ++i) for (int c = 0; c < 2; ++c) {
        float a = h[t * 4 + c], b = h[t * 4 + 2 + c];
        float want = i == 0 ? fmaf(4, a, b) : i == 1 ? fmaf(-4, a, b) : i == 2 ? fmaf(2, a, b) : i == 3 ? fmaf(-2, a, b) : i == 4 ? fmaf(-5, a, b) : i == 5 ? a + b : a - b;
        if (ho[(t * 7 + i) * 2 + c] != want) { if (bad < 8) printf("MISMATCH %s lane %d half %d: got %g want %g (a %g b %g)\n", names[i], t, c, ho[(t * 7 + i) * 2 + c], want, a, b); ++bad; }
    }
    printf("%d mismatches\n", bad);
    return bad != 0;
}
