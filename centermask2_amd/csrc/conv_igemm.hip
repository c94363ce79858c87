// Convolution (3x3 s1/s2, 1x1, and FC as 1x1 over "pixels" = rows) as an implicit GEMM on the CDNA4 matrix pipe.
//
//   GEMM view:  M = output pixels, N = output channels, K = taps * Cin, all fp32.
//   Instruction: v_mfma_f32_32x32x2_f32 — exact fp32 (bitwise a k-ordered fmaf chain), 64 FLOP/clk/SIMD, the same
//   peak as the fp32 VALU but one VGPR per operand, so the tile is fed from LDS with ds_read_b128 instead of
//   per-FMA register traffic.  A 32x32 accumulator's column sits on the lane, so Cout is mapped to the lane
//   (128-byte contiguous NHWC stores) and pixels to the accumulator rows.
//
//   Tiling (block = 4 waves stacked along M):
//     3x3:  spatial tile of (8*WM) x 16 output pixels; its input halo tile is staged ONCE per 16-channel K chunk
//           in LDS and the 9 taps read it at shifted addresses (9x fewer global->LDS bytes than per-tap im2col).
//     1x1:  128*WM consecutive pixels of the flattened (N*H*W) axis.
//     Per K chunk and tap the block stages a [32*WN couts][16 ci] weight slab (pre-packed, contiguous in HBM).
//   K order inside a 16-chunk is permuted so that MFMA k-step s of lane half h uses channel 8h+s: every lane then
//   reads 8 contiguous floats per operand row (2 x ds_read_b128) for 8 MFMAs.  Rows are padded to 20 floats
//   (80 B), which spreads the 16-lane ds_read_b128 groups over all 16-byte LDS slots.
//   Pipeline: global loads for step s+1 (weights) and chunk c+1 (halo) are issued before the barrier of step s
//   and written to the other LDS buffer after its MFMAs (register-staged double buffering, one barrier per step).
//
// Reference call sites replaced: aten::conv2d + FrozenBN + ReLU vovnet.py:205-236; d2 FPN convs (vovnet.py:547-554);
// fpn.py:27-35; fcos.py:169-200; sam.py:58-83; maskiou_head.py:81-93; nn.Linear maskiou_head.py:89-91.
#include "cmk_common.hpp"

namespace cmk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PST = 20;  // LDS row pitch in floats: 16 channels + 4 pad

struct ConvArgs {
    const float* x; const float* w; const float* scale; const float* shift; const float* res; float* y;
    int N, H, W, Ho, Wo, Cin, Cout;
    int x_cs, x_co, y_cs, y_co, res_cs, res_co, res_mode, Hr, Wr;
    int relu_upto, in_relu;
    int tiles_h, tiles_w, cout_pad;
    long total_pix;  // N*Ho*Wo
};

template <int TAPS, int STRIDE, int WM, int WN>
struct Geo {
    static constexpr int SUBT = 4 * WM;  // 32-pixel sub-tiles per block
    static constexpr int BM = 32 * SUBT;
    static constexpr int BN = 32 * WN;
    static constexpr int TH = (TAPS == 9) ? 2 * SUBT : 1;
    static constexpr int TW = (TAPS == 9) ? 16 : BM;
    static constexpr int HH = (TAPS == 9) ? (TH - 1) * STRIDE + 3 : 1;
    static constexpr int HWD = (TAPS == 9) ? (TW - 1) * STRIDE + 3 : BM;
    static constexpr int APIX = HH * HWD;
    static constexpr int A_BYTES = APIX * PST * 4;
    static constexpr int B_BYTES = BN * PST * 4;
    static constexpr bool ADB = (2 * A_BYTES + 2 * B_BYTES) <= 80 * 1024;  // double-buffer the halo if 2 blocks/CU still fit
    static constexpr int LDS_BYTES = (ADB ? 2 : 1) * A_BYTES + 2 * B_BYTES;
    static constexpr int A_ITERS = (APIX * 4 + 255) / 256;
    static constexpr int B_ITERS = (BN * 4 + 255) / 256;
};

template <int TAPS, int STRIDE, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs p) {
    using G = Geo<TAPS, STRIDE, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + (G::ADB ? 2 : 1) * G::APIX * PST;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hh = lane >> 5;  // k half
    const int li = lane & 31;

    int n = 0, oh0 = 0, ow0 = 0;
    long pix0 = 0;
    if (TAPS == 9) {
        int tile = blockIdx.x;
        int tw = tile % p.tiles_w;
        int t2 = tile / p.tiles_w;
        int th = t2 % p.tiles_h;
        n = t2 / p.tiles_h;
        oh0 = th * G::TH;
        ow0 = tw * G::TW;
    } else {
        pix0 = (long)blockIdx.x * G::BM;
    }
    const int co0 = blockIdx.y * G::BN;
    const int nchunks = p.Cin >> 4;
    const int total_steps = nchunks * TAPS;

    // ---- per-thread staging descriptors ------------------------------------------------------------------
    const float* xin = p.x + (TAPS == 9 ? (long)n * p.H * p.W * p.x_cs : 0L) + p.x_co;
    long a_goff[G::A_ITERS];
    int a_loff[G::A_ITERS];
#pragma unroll
    for (int it = 0; it < G::A_ITERS; ++it) {
        int idx = it * 256 + tid;
        int pix = idx >> 2, q = idx & 3;
        a_loff[it] = pix * PST + q * 4;
        a_goff[it] = -1;
        if (idx < G::APIX * 4) {
            if (TAPS == 9) {
                int hr = pix / G::HWD, hc = pix - hr * G::HWD;
                int ih = oh0 * STRIDE - 1 + hr, iw = ow0 * STRIDE - 1 + hc;
                if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) a_goff[it] = ((long)ih * p.W + iw) * p.x_cs + q * 4;
            } else {
                long P = pix0 + pix;
                if (P < p.total_pix) a_goff[it] = P * p.x_cs + q * 4;
            }
        } else {
            a_loff[it] = -1;
        }
    }
    f32x4 a_stage[G::A_ITERS];
    f32x4 b_stage[G::B_ITERS];

    auto load_A = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < G::A_ITERS; ++it) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (a_goff[it] >= 0) v = *reinterpret_cast<const f32x4*>(xin + a_goff[it] + chunk * 16);
            a_stage[it] = v;
        }
    };
    auto store_A = [&](int buf) {
        float* dst = sA + buf * (G::APIX * PST);
#pragma unroll
        for (int it = 0; it < G::A_ITERS; ++it) {
            if (a_loff[it] >= 0) {
                f32x4 v = a_stage[it];
                if (p.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<f32x4*>(dst + a_loff[it]) = v;
            }
        }
    };
    auto load_B = [&](int step) {
        int chunk = step / TAPS, tap = step - chunk * TAPS;
        const float* wsrc = p.w + ((long)(tap * nchunks + chunk) * p.cout_pad + co0) * 16;
#pragma unroll
        for (int it = 0; it < G::B_ITERS; ++it) {
            int idx = it * 256 + tid;
            if (idx < G::BN * 4) b_stage[it] = *reinterpret_cast<const f32x4*>(wsrc + idx * 4);
        }
    };
    auto store_B = [&](int buf) {
        float* dst = sB + buf * (G::BN * PST);
#pragma unroll
        for (int it = 0; it < G::B_ITERS; ++it) {
            int idx = it * 256 + tid;
            if (idx < G::BN * 4) *reinterpret_cast<f32x4*>(dst + (idx >> 2) * PST + (idx & 3) * 4) = b_stage[it];
        }
    };

    // ---- MFMA operand addresses -----------------------------------------------------------------------------
    int a_off[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        int u = wave * WM + m;
        if (TAPS == 9) {
            int r = li >> 4, cc = li & 15;
            a_off[m] = (((u * 2 + r) * STRIDE) * G::HWD + cc * STRIDE) * PST + hh * 8;
        } else {
            a_off[m] = (u * 32 + li) * PST + hh * 8;
        }
    }
    const int b_off = li * PST + hh * 8;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int nn = 0; nn < WN; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;

    // ---- prologue ---------------------------------------------------------------------------------------------
    load_A(0);
    load_B(0);
    store_A(0);
    store_B(0);

    int step = 0;
    for (int c = 0; c < nchunks; ++c) {
        const bool has_next_chunk = (c + 1 < nchunks);
        if (has_next_chunk) load_A(c + 1);
        const float* Abase = sA + (G::ADB ? (c & 1) : 0) * (G::APIX * PST);
#pragma unroll 1
        for (int t = 0; t < TAPS; ++t, ++step) {
            const bool has_next = (step + 1 < total_steps);
            if (has_next) load_B(step + 1);
            __syncthreads();  // staged data of this step visible; every wave is done with step-1

            int tapoff = 0;
            if (TAPS == 9) {
                int kh = t / 3, kw = t - kh * 3;
                tapoff = (kh * G::HWD + kw) * PST;
            }
            const float* A = Abase + tapoff;
            const float* B = sB + (step & 1) * (G::BN * PST) + b_off;
            f32x4 a0[WM], a1[WM];
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                a0[m] = *reinterpret_cast<const f32x4*>(A + a_off[m]);
                a1[m] = *reinterpret_cast<const f32x4*>(A + a_off[m] + 4);
            }
#pragma unroll
            for (int nn = 0; nn < WN; ++nn) {
                f32x4 b0 = *reinterpret_cast<const f32x4*>(B + nn * 32 * PST);
                f32x4 b1 = *reinterpret_cast<const f32x4*>(B + nn * 32 * PST + 4);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int m = 0; m < WM; ++m)
                        acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m][s], b0[s], acc[m][nn], 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int m = 0; m < WM; ++m)
                        acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m][s], b1[s], acc[m][nn], 0, 0, 0);
            }
            if (has_next) store_B((step + 1) & 1);
            if (t == TAPS - 1 && has_next_chunk) {
                if (!G::ADB) __syncthreads();  // single halo buffer: everyone must be done reading it
                store_A(G::ADB ? ((c + 1) & 1) : 0);
            }
        }
    }

    // ---- epilogue: scale/shift (+residual) (+ReLU), NHWC store -------------------------------------------------
#pragma unroll
    for (int nn = 0; nn < WN; ++nn) {
        const int co = co0 + nn * 32 + li;
        const bool cvalid = co < p.Cout;
        const float sc = cvalid ? p.scale[co] : 0.f;
        const float sh = cvalid ? p.shift[co] : 0.f;
        const bool do_relu = co < p.relu_upto;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const int u = wave * WM + m;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
                long opix;
                bool pvalid;
                int oh = 0, ow = 0;
                if (TAPS == 9) {
                    oh = oh0 + u * 2 + (row >> 4);
                    ow = ow0 + (row & 15);
                    pvalid = (oh < p.Ho) && (ow < p.Wo);
                    opix = ((long)n * p.Ho + oh) * p.Wo + ow;
                } else {
                    opix = pix0 + u * 32 + row;
                    pvalid = opix < p.total_pix;
                }
                if (cvalid && pvalid) {
                    float v = acc[m][nn][r] * sc + sh;
                    if (p.res_mode == 1) {
                        v += p.res[opix * p.res_cs + p.res_co + co];
                    } else if (p.res_mode == 2) {
                        if (TAPS != 9) {  // recover (n, oh, ow) from the flattened pixel index
                            long hw = (long)p.Ho * p.Wo;
                            n = (int)(opix / hw);
                            int rem = (int)(opix - (long)n * hw);
                            oh = rem / p.Wo;
                            ow = rem - oh * p.Wo;
                        }
                        long rp = ((long)n * p.Hr + (oh >> 1)) * p.Wr + (ow >> 1);
                        v += p.res[rp * p.res_cs + p.res_co + co];
                    }
                    if (do_relu) v = fmaxf(v, 0.f);
                    p.y[opix * p.y_cs + p.y_co + co] = v;
                }
            }
        }
    }
}

template <int TAPS, int STRIDE, int WM, int WN>
static int launch(const ConvArgs& a, int grid_x, int grid_y, hipStream_t st) {
    using G = Geo<TAPS, STRIDE, WM, WN>;
    static bool attr_set = false;
    auto kern = conv_igemm_kernel<TAPS, STRIDE, WM, WN>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           G::LDS_BYTES);
        if (e != hipSuccess) return fail(CMK_ELAUNCH, "conv: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid_x, grid_y), dim3(256), G::LDS_BYTES, st, a);
    return check_launch("conv_igemm");
}

template <int TAPS, int STRIDE>
static int dispatch_wn(const ConvArgs& a0, int cout32, hipStream_t st) {
    ConvArgs a = a0;
    // WN<=4: WM=2 (except stride 2, whose halo only fits with WM=1); 5..7: WM=1; >=8: 128-wide N tiles.
    constexpr int WMs = (STRIDE == 2) ? 1 : 2;
    int wn = cout32 <= 7 ? cout32 : 4;
    int grid_y = cout32 <= 7 ? 1 : cdiv(cout32, 4);
    a.cout_pad = grid_y * wn * 32;
    int wm = (wn <= 4) ? WMs : 1;
    int grid_x;
    if (TAPS == 9) {
        int th = 8 * wm;
        a.tiles_h = cdiv(a.Ho, th);
        a.tiles_w = cdiv(a.Wo, 16);
        grid_x = a.N * a.tiles_h * a.tiles_w;
    } else {
        grid_x = (int)((a.total_pix + 128 * wm - 1) / (128 * wm));
    }
    switch (wn) {
        case 1: return launch<TAPS, STRIDE, WMs, 1>(a, grid_x, grid_y, st);
        case 2: return launch<TAPS, STRIDE, WMs, 2>(a, grid_x, grid_y, st);
        case 3: return launch<TAPS, STRIDE, WMs, 3>(a, grid_x, grid_y, st);
        case 4: return launch<TAPS, STRIDE, WMs, 4>(a, grid_x, grid_y, st);
        case 5: return launch<TAPS, STRIDE, 1, 5>(a, grid_x, grid_y, st);
        case 6: return launch<TAPS, STRIDE, 1, 6>(a, grid_x, grid_y, st);
        case 7: return launch<TAPS, STRIDE, 1, 7>(a, grid_x, grid_y, st);
    }
    return fail(CMK_EINVAL, "conv: bad WN%s", "");
}

}  // namespace cmk

extern "C" int cmk_conv_cout_pad(int Cout) {
    int c32 = (Cout + 31) / 32;
    return c32 <= 7 ? c32 * 32 : ((c32 + 3) / 4) * 128;
}

extern "C" int64_t cmk_conv_packed_floats(int Cout, int Cin, int ksize) {
    int64_t taps = (int64_t)ksize * ksize;
    int64_t nch = (Cin + 15) / 16;
    return taps * nch * cmk_conv_cout_pad(Cout) * 16;
}

extern "C" int cmk_conv2d_nhwc(const cmk_conv_desc* d, void* stream) {
    using namespace cmk;
    if (!d || !d->x || !d->w || !d->y || !d->scale || !d->shift) return fail(CMK_EINVAL, "conv: null pointer%s", "");
    if (d->ksize != 1 && d->ksize != 3) return fail(CMK_EINVAL, "conv: ksize must be 1 or 3%s", "");
    if (d->stride != 1 && d->stride != 2) return fail(CMK_EINVAL, "conv: stride must be 1 or 2%s", "");
    if (d->ksize == 1 && d->stride != 1) return fail(CMK_EINVAL, "conv: 1x1 stride 2 unsupported%s", "");
    if (d->Cin <= 0 || (d->Cin & 15)) return fail(CMK_EINVAL, "conv: Cin (%s%ld) must be a positive multiple of 16", "", d->Cin);
    if (d->Cout <= 0 || d->N <= 0 || d->H <= 0 || d->W <= 0) return fail(CMK_EINVAL, "conv: empty shape%s", "");
    if ((d->x_cs & 3) || (d->x_co & 3)) return fail(CMK_EINVAL, "conv: input view must be 16-byte aligned per pixel%s", "");
    if (((uintptr_t)d->x & 15) || ((uintptr_t)d->w & 15)) return fail(CMK_EINVAL, "conv: x/w must be 16-byte aligned%s", "");
    if (d->x_co + d->Cin > d->x_cs || d->y_co + d->Cout > d->y_cs) return fail(CMK_EINVAL, "conv: channel view out of range%s", "");
    if (d->res_mode < 0 || d->res_mode > 2 || (d->res_mode && !d->res)) return fail(CMK_EINVAL, "conv: bad residual%s", "");
    ConvArgs a;
    a.x = d->x; a.w = d->w; a.scale = d->scale; a.shift = d->shift; a.res = d->res; a.y = d->y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
    a.Ho = d->stride == 1 ? d->H : (d->H - 1) / 2 + 1;  // k3 p1 s2: floor((H+2-3)/2)+1
    a.Wo = d->stride == 1 ? d->W : (d->W - 1) / 2 + 1;
    a.x_cs = d->x_cs; a.x_co = d->x_co; a.y_cs = d->y_cs; a.y_co = d->y_co;
    a.res_cs = d->res_cs; a.res_co = d->res_co; a.res_mode = d->res_mode; a.Hr = d->Hr; a.Wr = d->Wr;
    if (a.res_mode == 2 && (a.Hr * 2 < a.Ho || a.Wr * 2 < a.Wo)) return fail(CMK_EINVAL, "conv: upsampled residual too small%s", "");
    a.relu_upto = d->relu_upto; a.in_relu = d->in_relu;
    a.tiles_h = a.tiles_w = 0; a.cout_pad = 0;
    a.total_pix = (long)a.N * a.Ho * a.Wo;
    const int cout32 = (d->Cout + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (d->ksize == 1) return dispatch_wn<1, 1>(a, cout32, st);
    if (d->stride == 1) return dispatch_wn<9, 1>(a, cout32, st);
    return dispatch_wn<9, 2>(a, cout32, st);
}
