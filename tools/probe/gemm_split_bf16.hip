// PROBE (not part of the package): an fp32-accurate 1x1-conv GEMM built from bf16-split products.
//   Y[M][N] = X[M][K] * W[K][N], X / Y fp32 (NHWC pixels x channels, as the model's 1x1 aggregation convs), every fp32 operand split into three
//   bf16 pieces (hi + mid + lo, exact to 2^-24), the six products of weight >= 2^-16 issued as v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//   X is split ON THE FLY while it is staged into LDS (the activations stay fp32 in HBM); W is split once on the host.
// Workgroup = 256 pixels x 128 couts, 4 waves as 2 x 2, wave tile 128 x 64 (8 accumulators), K step 16, two LDS stages.
// Prints the rate in fp32-equivalent TFLOP/s (2*M*N*K / time) next to the max error against float64 on sampled outputs, for the OSA aggregation
// shapes.  Build: hipcc --offload-arch=gfx950 -O3 gemm_split_bf16.hip -o gemm_split_bf16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {          // round-to-nearest-even pair -> {lo 16: a, hi 16: b}
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// split 4 fp32 values into 3 x 4 bf16 (as 2 dwords per piece)
__device__ __forceinline__ void split4(const f32x4 x, u32x2& hi, u32x2& mid, u32x2& lo) {
    hi.x = pk_bf16(x.x, x.y); hi.y = pk_bf16(x.z, x.w);
    f32x4 r;
    r.x = x.x - __builtin_bit_cast(float, hi.x << 16); r.y = x.y - __builtin_bit_cast(float, hi.x & 0xffff0000u);
    r.z = x.z - __builtin_bit_cast(float, hi.y << 16); r.w = x.w - __builtin_bit_cast(float, hi.y & 0xffff0000u);
    mid.x = pk_bf16(r.x, r.y); mid.y = pk_bf16(r.z, r.w);
    f32x4 q;
    q.x = r.x - __builtin_bit_cast(float, mid.x << 16); q.y = r.y - __builtin_bit_cast(float, mid.x & 0xffff0000u);
    q.z = r.z - __builtin_bit_cast(float, mid.y << 16); q.w = r.w - __builtin_bit_cast(float, mid.y & 0xffff0000u);
    lo.x = pk_bf16(q.x, q.y); lo.y = pk_bf16(q.z, q.w);
}

#ifndef GS_ABL
#define GS_ABL 0      // timing ablations (wrong results): 1 no split arithmetic, 2 no X loads in the loop, 4 no W loads in the loop, 8 no staging stores
#endif
constexpr int BM = 256, BN = 128, KS = 16;
constexpr int STAGE_BYTES = (BM / 32) * 3 * 64 * 16;       // [row block 8][piece 3][lane 64][8 bf16] = 24 KiB

// Wp: [K/16][N/32][piece 3][lane 64][8 bf16]
__global__ __launch_bounds__(256, 2) void gemm_split(const float* __restrict__ X, const u32x4* __restrict__ Wp, float* __restrict__ Y, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const long row0 = (long)blockIdx.x * BM;
    const int cb0 = blockIdx.y * (BN / 32) + 2 * wn;       // first 32-cout block of this wave
    const int nsteps = K / KS, ncb = N / 32;
    // staging item of this thread: quad q of rows (tid / 4) + 64 j, j = 0..3
    const int q = tid & 3, r_base = tid >> 2;
    const float* xp = X + (row0 + r_base) * K + q * 4;
    const long xrow = 64L * K;
    // LDS byte offset of the item's 8-byte half slot: row r -> (rb = r / 32, li = r % 32), lane = (q >> 1) * 32 + li
    int st_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r_base + 64 * j;
        st_off[j] = (((r >> 5) * 3) * 64 + (q >> 1) * 32 + (r & 31)) * 16 + (q & 1) * 8;
    }
    f32x4 xr[4];
    auto load_x = [&](int s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xr[j] = *reinterpret_cast<const f32x4*>(xp + j * xrow + (long)s * KS);
    };
    auto stage = [&](unsigned char* buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x2 h, m, l;
#if GS_ABL & 1
            h = u32x2{__builtin_bit_cast(unsigned, xr[j].x), __builtin_bit_cast(unsigned, xr[j].y)}; m = u32x2{__builtin_bit_cast(unsigned, xr[j].z), __builtin_bit_cast(unsigned, xr[j].w)}; l = h;
#else
            split4(xr[j], h, m, l);
#endif
#if GS_ABL & 8
            asm volatile("" :: "v"(h), "v"(m), "v"(l));
            continue;
#endif
            *reinterpret_cast<u32x2*>(buf + st_off[j]) = h;
            *reinterpret_cast<u32x2*>(buf + st_off[j] + 64 * 16) = m;
            *reinterpret_cast<u32x2*>(buf + st_off[j] + 2 * 64 * 16) = l;
        }
    };
    u32x4 wb[2][3];
    const u32x4* wbase = Wp + ((long)cb0 * 3) * 64 + lane;
    const long wstep = (long)ncb * 3 * 64;
    auto load_w = [&](int s) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int p = 0; p < 3; ++p) wb[cb][p] = wbase[(long)s * wstep + (cb * 3 + p) * 64];
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    load_x(0);
    load_w(0);
    stage(smem);
    if (nsteps > 1) load_x(1);
    for (int s = 0; s < nsteps; ++s) {
        unsigned char* cur = smem + (s & 1) * STAGE_BYTES;
        unsigned char* nxt = smem + ((s + 1) & 1) * STAGE_BYTES;
        __syncthreads();
        const u32x4* ap = reinterpret_cast<const u32x4*>(cur) + (wm * 4 * 3) * 64 + lane;
        u32x4 wcur[2][3];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int p = 0; p < 3; ++p) wcur[cb][p] = wb[cb][p];
#if !(GS_ABL & 4)
        if (s + 1 < nsteps) load_w(s + 1);
#endif
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const u32x4 ah = ap[(rb * 3 + 0) * 64], am = ap[(rb * 3 + 1) * 64], al = ap[(rb * 3 + 2) * 64];
            if (rb == 1 && s + 1 < nsteps) {          // split the next step's activations between the MFMA groups
                stage(nxt);
#if !(GS_ABL & 2)
                if (s + 2 < nsteps) load_x(s + 2);
#endif
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Am = __builtin_bit_cast(bf16x8, am), Al = __builtin_bit_cast(bf16x8, al);
                const bf16x8 Bh = __builtin_bit_cast(bf16x8, wcur[cb][0]), Bm = __builtin_bit_cast(bf16x8, wcur[cb][1]), Bl = __builtin_bit_cast(bf16x8, wcur[cb][2]);
                f32x16 c = acc[rb][cb];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, c, 0, 0, 0);      // small terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, c, 0, 0, 0);
                acc[rb][cb] = c;
            }
        }
    }
    // epilogue: accumulator register r of lane (hh, li) is row (r & 3) + 8 (r >> 2) + 4 hh, column li of the block
    const int hh = lane >> 5, li = lane & 31;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = row0 + (wm * 4 + rb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (row < M) Y[row * N + (cb0 + cb) * 32 + li] = acc[rb][cb][r];
            }
}

static uint16_t bf16_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    struct Shape { const char* name; int M, K, N; } shapes[] = {{"OSA2 768->256 @8x200x320", 8 * 200 * 320, 768, 256}, {"OSA3 1056->512 @8x100x160", 8 * 100 * 160, 1056, 512},
                                                                  {"OSA4 1472->768 @8x50x80", 8 * 50 * 80, 1472, 768}, {"OSA5 1888->1024 @8x25x40", 8 * 25 * 40, 1888, 1024}};
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    for (const Shape& sh : shapes) {
        const int M = sh.M, K = sh.K, N = sh.N;
        if (M % BM || N % BN || K % KS) { printf("%s: skipped (tile divisibility)\n", sh.name); continue; }
        std::vector<float> hx((size_t)M * K), hw((size_t)K * N);
        srand(7);
        for (auto& v : hx) v = (float)rand() / RAND_MAX * 2.f - 0.5f;                       // post-ReLU-like, mostly positive
        for (auto& v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 2.f * sqrtf(3.f / K);
        // host split of W into the packed layout [K/16][N/32][piece][lane = (hh, li)][8 bf16]: k = 16 s + 8 hh + e, n = 32 cb + li
        std::vector<uint16_t> hp((size_t)K * N * 3);
        for (int s = 0; s < K / 16; ++s)
            for (int cb = 0; cb < N / 32; ++cb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int k = 16 * s + 8 * (lane >> 5) + e, n = 32 * cb + (lane & 31);
                        float x = hw[(size_t)k * N + n];
                        for (int p = 0; p < 3; ++p) {
                            const uint16_t h = bf16_rne(x);
                            hp[((((size_t)s * (N / 32) + cb) * 3 + p) * 64 + lane) * 8 + e] = h;
                            x -= bf16_to_f(h);
                        }
                    }
        float *dx, *dy; u32x4* dw;
        hipMalloc(&dx, hx.size() * 4); hipMalloc(&dy, (size_t)M * N * 4); hipMalloc(&dw, hp.size() * 2);
        hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
        const dim3 grid(M / BM, N / BN);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_split, grid, dim3(256), 2 * STAGE_BYTES, 0, dx, dw, dy, M, N, K);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int it = 20;
        hipEventRecord(e0);
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(gemm_split, grid, dim3(256), 2 * STAGE_BYTES, 0, dx, dw, dy, M, N, K);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
        std::vector<float> hy((size_t)M * N);
        hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0.0, maxref = 0.0, err32 = 0.0;
        for (int t = 0; t < 2000; ++t) {
            const long row = (long)(rand() % M); const int n = rand() % N;
            double ref = 0.0; float f32 = 0.f;
            for (int k = 0; k < K; ++k) { ref += (double)hx[row * K + k] * (double)hw[(size_t)k * N + n]; f32 = fmaf(hx[row * K + k], hw[(size_t)k * N + n], f32); }
            maxerr = fmax(maxerr, fabs((double)hy[row * N + n] - ref)); err32 = fmax(err32, fabs((double)f32 - ref)); maxref = fmax(maxref, fabs(ref));
        }
        printf("%-28s %.3f ms  %.1f fp32-equivalent TFLOP/s   max|err| vs float64 %.2e (a sequential fp32 fma chain: %.2e; max|ref| %.2f)\n", sh.name, ms,
               2.0 * M * N * K / (ms * 1e-3) / 1e12, maxerr, err32, maxref);
        fflush(stdout);
        hipFree(dx); hipFree(dy); hipFree(dw);
    }
    return 0;
}
