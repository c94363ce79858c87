// HBM-bound NHWC kernels of the backbone and the FCOS towers: stem conv on the raw NCHW image, ceil-mode max pool,
// eSE (global average pool -> fc -> hsigmoid -> channel scale [+ identity]) and GroupNorm(32)+ReLU.
// All of them move 16 bytes per lane (float4) with channels innermost, so a wave reads/writes whole 256 B-1 KiB runs.
#include "cmk_common.hpp"

namespace cmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------
// stem_1 (vovnet.py:409): conv3x3 s2 p1 on (N,3,H,W) NCHW -> (N,Ho,Wo,Cout) NHWC, y = relu(acc*scale + shift).
// 4 threads per output pixel, 16 output channels each (Cout = 64); 27 taps broadcast from LDS.
// ---------------------------------------------------------------------------------------------------------------
template <int NB>   // NB = Cout / 32
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       float* __restrict__ y, int N, int H, int W, int Ho, int Wo, long tiles) {
    // GEMM on the matrix pipe: M = output pixels (32 per wave tile), K = 27 taps padded to 28, N = Cout.
    // A[pixel][k] is gathered straight from the NCHW image (lane = pixel, lane>>5 = k parity), the 14 x NB weight
    // operands B[k][cout] stay in registers for the whole kernel, the accumulators come out with cout on the lane so
    // every store instruction writes two full 128-byte NHWC runs.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, hh = lane >> 5;
    constexpr int Cout = NB * 32;
    float b[14][NB], sc[NB], sh[NB];
    int dlt[14], khs[14], kws[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int k = 2 * s + hh;                 // tap index (kh*3+kw)*3+ci, 27 = padding
        const int ci = k % 3, t = k / 3, kw = t % 3, kh = t / 3;
        khs[s] = kh - 1; kws[s] = kw - 1;
        dlt[s] = (ci * H + kh - 1) * W + kw - 1;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[s][nb] = k < 27 ? w[k * Cout + nb * 32 + li] : 0.f;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { sc[nb] = scale[nb * 32 + li]; sh[nb] = shift[nb * 32 + li]; }
    const long total = (long)N * Ho * Wo;
    const long hw = (long)Ho * Wo;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < tiles; tile += (long)gridDim.x * 4) {
        const long pix = tile * 32 + li;
        const bool pv = pix < total;
        const int n = (int)(pix / hw);
        const int rem = (int)(pix - (long)n * hw);
        const int oh = rem / Wo, ow = rem - oh * Wo;
        const int ih0 = oh * 2, iw0 = ow * 2;
        const float* xb = x + ((long)n * 3 * H + ih0) * W + iw0;
        float av[14];
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int ih = ih0 + khs[s], iw = iw0 + kws[s];
            const bool ok = pv && (2 * s + hh < 27) && ih >= 0 && ih < H && iw >= 0 && iw < W;
            av[s] = ok ? xb[dlt[s]] : 0.f;
        }
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 14; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b[s][nb], acc[nb], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long op = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (op < total) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) y[op * Cout + nb * 32 + li] = fmaxf(acc[nb][r] * sc[nb] + sh[nb], 0.f);
            }
        }
    }
}

// depth-wise 3x3 (vovnet.py:110-119 'dw_conv3x3': groups = channels, pad 1, no bias, no activation; the point-wise 1x1 +
// FrozenBN + ReLU that follows is an ordinary conv launch).  HBM-bound: one thread = 4 channels x T adjacent output columns,
// so the 3 x (T*S+2) input quads are each loaded once per thread and the 9 weight quads live in registers.
template <int S, int T>
__global__ __launch_bounds__(256) void dwconv3_kernel(const float* __restrict__ x, int x_cs, int x_co, const float* __restrict__ w,
                                                     float* __restrict__ y, int y_cs, int y_co, int N, int H, int W, int Ho, int Wo,
                                                     int C4) {
    const int WT = (Wo + T - 1) / T;
    const long total = (long)N * Ho * WT * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        long r = i / C4;
        const int wt = (int)(r % WT);
        r /= WT;
        const int oh = (int)(r % Ho);
        const int n = (int)(r / Ho);
        f32x4 wq[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wq[t] = *reinterpret_cast<const f32x4*>(w + (long)t * C4 * 4 + c4 * 4);
        f32x4 acc[T];
#pragma unroll
        for (int j = 0; j < T; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int iw0 = wt * T * S - 1;
        constexpr int COLS = (T - 1) * S + 3;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * S - 1 + kh;
            if (ih < 0 || ih >= H) continue;
            const float* row = x + ((long)n * H + ih) * W * x_cs + x_co + c4 * 4;
#pragma unroll
            for (int cc = 0; cc < COLS; ++cc) {
                const int iw = iw0 + cc;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (iw >= 0 && iw < W) v = *reinterpret_cast<const f32x4*>(row + (long)iw * x_cs);
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    const int kw = cc - j * S;       // compile-time after unrolling
                    if (kw >= 0 && kw < 3) {
                        const f32x4 wv = wq[kh * 3 + kw];
                        acc[j].x = fmaf(v.x, wv.x, acc[j].x); acc[j].y = fmaf(v.y, wv.y, acc[j].y);
                        acc[j].z = fmaf(v.z, wv.z, acc[j].z); acc[j].w = fmaf(v.w, wv.w, acc[j].w);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int ow = wt * T + j;
            if (ow < Wo) *reinterpret_cast<f32x4*>(y + (((long)n * Ho + oh) * Wo + ow) * y_cs + y_co + c4 * 4) = acc[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// MaxPool2d(3, 2, ceil_mode=True), no padding (vovnet.py:349-350).  One lane = one output pixel x 4 channels.
// ---------------------------------------------------------------------------------------------------------------
// `gate` (optional, N*C, all >= 0): the eSE channel gate of the producer block applied after the max — max(x*g) == g*max(x)
// for g >= 0, so stage 2's eSE scale pass (whose output nobody else reads) folds into the pool.
__global__ __launch_bounds__(256) void maxpool3_kernel(const float* __restrict__ x, int x_cs, int x_co, float* __restrict__ y,
                                                      int y_cs, int y_co, int N, int H, int W, int Ho, int Wo, int C4,
                                                      const float* __restrict__ gate) {
    long total = (long)N * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4 = (int)(i % C4);
        long p = i / C4;
        int ow = (int)(p % Wo);
        int oh = (int)((p / Wo) % Ho);
        int n = (int)(p / ((long)Wo * Ho));
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                int ih = oh * 2 + kh, iw = ow * 2 + kw;
                if (ih < H && iw < W) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long)n * H + ih) * W + iw) * x_cs + x_co + c4 * 4);
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            }
        if (gate) {
            f32x4 g = *reinterpret_cast<const f32x4*>(gate + ((long)n * C4 + c4) * 4);
            m.x *= g.x; m.y *= g.y; m.z *= g.z; m.w *= g.w;
        }
        *reinterpret_cast<f32x4*>(y + p * y_cs + y_co + c4 * 4) = m;
    }
}

// MaxPool2d(kernel 1, stride 2) = every second pixel (d2's LastLevelMaxPool, the top block of build_vovnet_fpn_backbone vovnet.py:504-524)
__global__ __launch_bounds__(256) void subsample2_kernel(const float* __restrict__ x, int x_cs, int x_co, float* __restrict__ y, int y_cs, int y_co,
                                                        int N, int H, int W, int Ho, int Wo, int C4) {
    long total = (long)N * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4 = (int)(i % C4);
        long p = i / C4;
        int ow = (int)(p % Wo);
        int oh = (int)((p / Wo) % Ho);
        int n = (int)(p / ((long)Wo * Ho));
        *reinterpret_cast<f32x4*>(y + p * y_cs + y_co + c4 * 4) =
            *reinterpret_cast<const f32x4*>(x + (((long)n * H + 2 * oh) * W + 2 * ow) * x_cs + x_co + c4 * 4);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// eSE (vovnet.py:247-260).  Stage 1: per (image, pixel chunk) channel sums (deterministic: fixed order, no atomics).
// Stage 2: mean -> fc (C x C mat-vec, one wave per output, wave64 shuffle reduction) -> relu6(v+3)/6.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ese_partial_kernel(const float* __restrict__ x, int x_cs, int x_co, float* __restrict__ ws,
                                                         int HW, int C, int chunks) {
    extern __shared__ float red[];  // [ppl][C]
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int G = C >> 2;
    const int ppl = G >= 256 ? 1 : 256 / G;  // pixel lanes that share a channel group
    const int per = cdiv(HW, chunks);
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    const float* xn = x + (long)n * HW * x_cs + x_co;
    for (int gi = threadIdx.x; gi < G * ppl; gi += 256) {
        int g = gi % G, pl = gi / G;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int p = p0 + pl; p < p1; p += ppl) {
            f32x4 v = *reinterpret_cast<const f32x4*>(xn + (long)p * x_cs + g * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<f32x4*>(red + pl * C + g * 4) = s;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int pl = 0; pl < ppl; ++pl) s += red[pl * C + c];
        ws[((long)n * chunks + chunk) * C + c] = s;
    }
}

__global__ __launch_bounds__(256) void ese_fc_kernel(const float* __restrict__ ws, const float* __restrict__ fc_w,
                                                    const float* __restrict__ fc_b, float* __restrict__ gate, int HW, int C,
                                                    int chunks) {
    // 16 outputs per workgroup.  Stage 1 reduces the per-chunk partial sums to the channel means: the chunk range is split over
    // `parts` thread groups (16 B per thread per chunk, independent loads), folded through LDS.  Stage 2: one wave per output.
    // (float64 from the partial sums on, as in ese_fc_pooled_kernel)
    extern __shared__ double smd[];  // [parts][C] partials, then mean in row 0
    const int n = blockIdx.y;
    const int G = C >> 2;
    const int parts = G >= 256 ? 1 : 256 / G;
    const float* wsn = ws + (long)n * chunks * C;
    for (int gi = threadIdx.x; gi < G * parts; gi += 256) {
        const int g = gi % G, part = gi / G;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 8
        for (int k = part; k < chunks; k += parts) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wsn + (long)k * C + g * 4);
            s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
        }
        double* o = smd + part * C + g * 4;
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double s = 0.0;
        for (int part = 0; part < parts; ++part) s += smd[part * C + c];
        smd[c] = s / (double)HW;   // row 0 is only read by its own thread above
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = blockIdx.x * 16 + wave; o < min(C, (int)(blockIdx.x + 1) * 16); o += 4) {
        const float* wr = fc_w + (long)o * C;
        double s = 0.0;
        for (int c = lane * 4; c < C; c += 256) {
            f32x4 wv = *reinterpret_cast<const f32x4*>(wr + c);
            s += (double)wv.x * smd[c] + (double)wv.y * smd[c + 1] + (double)wv.z * smd[c + 2] + (double)wv.w * smd[c + 3];
        }
        s = wave_sum(s);
        if (lane == 0) {
            float v = (float)(s + (double)fc_b[o] + 3.0);
            gate[(long)n * C + o] = fminf(fmaxf(v, 0.f), 6.f) / 6.0f;
        }
    }
}

// The gate from the per-block sums the aggregation conv left behind (cmk_conv_desc.pool_ws): record 2g = rows of block g in the image of
// its first pixel, 2g+1 = rows of block g in the next image.  Stage 1 (channel means) differs from ese_fc_kernel, stage 2 is the same.
__global__ __launch_bounds__(256) void ese_fc_pooled_kernel(const float* __restrict__ rec, int rows, const float* __restrict__ fc_w,
                                                           const float* __restrict__ fc_b, float* __restrict__ gate, int HW, int C) {
    // The records are fp32 sums of at most `rows` values each; everything downstream of them — the sum over the records, the mean and
    // the C x C fully-connected layer — is accumulated in float64 in a fixed order (a few thousand adds per image: free), so the gate
    // carries the rounding of the records and of one final conversion only (the detection order of near-tied scores
    // depends on summation-order noise upstream; this removes this kernel's share of it).
    extern __shared__ double smd[];  // [parts][C] partials, then the mean in row 0
    const int n = blockIdx.y;
    const int G = C >> 2;
    const int parts = G >= 256 ? 1 : 256 / G;
    const long p0 = (long)n * HW;
    const long g0 = p0 / rows, g1 = (p0 + HW - 1) / rows;
    const long first = (g0 * rows == p0) ? 2 * g0 : 2 * g0 + 1;          // the record of block g0 that holds this image's rows
    for (int gi = threadIdx.x; gi < G * parts; gi += 256) {
        const int g = gi % G, part = gi / G;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (part == 0) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rec + first * C + g * 4);
            s0 = v.x; s1 = v.y; s2 = v.z; s3 = v.w;
        }
#pragma unroll 8
        for (long k = g0 + 1 + part; k <= g1; k += parts) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rec + 2 * k * C + g * 4);
            s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
        }
        double* o = smd + part * C + g * 4;
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double s = 0.0;
        for (int part = 0; part < parts; ++part) s += smd[part * C + c];
        smd[c] = s / (double)HW;   // row 0 is only read by its own thread above
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = blockIdx.x * 16 + wave; o < min(C, (int)(blockIdx.x + 1) * 16); o += 4) {
        const float* wr = fc_w + (long)o * C;
        double s = 0.0;
        for (int c = lane * 4; c < C; c += 256) {
            f32x4 wv = *reinterpret_cast<const f32x4*>(wr + c);
            s += (double)wv.x * smd[c] + (double)wv.y * smd[c + 1] + (double)wv.z * smd[c + 2] + (double)wv.w * smd[c + 3];
        }
        s = wave_sum(s);
        if (lane == 0) {
            float v = (float)(s + (double)fc_b[o] + 3.0);
            gate[(long)n * C + o] = fminf(fmaxf(v, 0.f), 6.f) / 6.0f;
        }
    }
}

__global__ __launch_bounds__(256) void ese_scale_kernel(const float* __restrict__ x, int x_cs, int x_co, const float* __restrict__ gate,
                                                       const float* __restrict__ idn, int id_cs, int id_co, float* __restrict__ y,
                                                       int y_cs, int y_co, int N, int HW, int C4) {
    long total = (long)N * HW * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4 = (int)(i % C4);
        long p = i / C4;
        int n = (int)(p / HW);
        f32x4 v = *reinterpret_cast<const f32x4*>(x + p * x_cs + x_co + c4 * 4);
        f32x4 g = *reinterpret_cast<const f32x4*>(gate + (long)n * C4 * 4 + c4 * 4);
        f32x4 o = {v.x * g.x, v.y * g.y, v.z * g.z, v.w * g.w};
        if (idn) {
            f32x4 d = *reinterpret_cast<const f32x4*>(idn + p * id_cs + id_co + c4 * 4);
            o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w;
        }
        *reinterpret_cast<f32x4*>(y + p * y_cs + y_co + c4 * 4) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm(32, C) + ReLU in place on a dense (N,HW,C) tensor (fcos.py:182-186).
// Stage 1: per (image, pixel chunk) fp64 sum / sum-of-squares per group (fixed order).  Stage 2: normalise.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ ws, int HW, int C, int groups,
                                                      int chunks) {
    __shared__ double rs[256], rss[256];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int G = C >> 2;                    // float4 groups per pixel (<= 256 required)
    const int ppl = 256 / G;
    const int per = cdiv(HW, chunks);
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    const float* xn = x + (long)n * HW * C;
    double s = 0.0, ss = 0.0;
    const int g = threadIdx.x % G, pl = threadIdx.x / G;
    if (pl < ppl) {
        for (int p = p0 + pl; p < p1; p += ppl) {
            f32x4 v = *reinterpret_cast<const f32x4*>(xn + (long)p * C + g * 4);
            s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    }
    rs[threadIdx.x] = s;
    rss[threadIdx.x] = ss;
    __syncthreads();
    if (threadIdx.x < groups) {
        const int f4pg = (C / groups) >> 2;  // float4 groups per GN group
        double a = 0.0, b = 0.0;
        for (int l = 0; l < ppl; ++l)
            for (int k = 0; k < f4pg; ++k) {
                a += rs[l * G + threadIdx.x * f4pg + k];
                b += rss[l * G + threadIdx.x * f4pg + k];
            }
        double* o = ws + (((long)n * groups + threadIdx.x) * chunks + chunk) * 2;
        o[0] = a;
        o[1] = b;
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const double* __restrict__ ws, int HW, int C, int groups, int chunks, float eps,
                                                      int blocks_per_image, float lo) {
    // lo: 0 = GroupNorm + ReLU, -inf = GroupNorm alone (max(v, -inf) = v)
    __shared__ float s_mean[64], s_rstd[64];
    const int n = blockIdx.y;
    if (threadIdx.x < groups) {
        double a = 0.0, b = 0.0;
        const double* w = ws + ((long)n * groups + threadIdx.x) * chunks * 2;
        for (int k = 0; k < chunks; ++k) { a += w[2 * k]; b += w[2 * k + 1]; }
        double cnt = (double)HW * (C / groups);
        double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[threadIdx.x] = (float)mean;
        s_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int C4 = C >> 2;
    const int cpg = C / groups;
    float* xn = x + (long)n * HW * C;
    long total = (long)HW * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)blocks_per_image * 256) {
        int c4 = (int)(i % C4);
        int c = c4 * 4;
        int grp = c / cpg;
        float mean = s_mean[grp], rstd = s_rstd[grp];
        f32x4 v = *reinterpret_cast<f32x4*>(xn + i * 4);
        f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
        f32x4 be = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o;
        o.x = fmaxf((v.x - mean) * rstd * ga.x + be.x, lo);
        o.y = fmaxf((v.y - mean) * rstd * ga.y + be.y, lo);
        o.z = fmaxf((v.z - mean) * rstd * ga.z + be.z, lo);
        o.w = fmaxf((v.w - mean) * rstd * ga.w + be.w, lo);
        *reinterpret_cast<f32x4*>(xn + i * 4) = o;
    }
}

// finalize: per (image, channel) scale/shift of GroupNorm for a consumer that applies it while staging its input
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ ws, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ out_scale,
                                                         float* __restrict__ out_shift, int HW, int C, int groups, int chunks, float eps) {
    __shared__ float s_mean[64], s_rstd[64];
    const int n = blockIdx.x;
    if (threadIdx.x < groups) {
        double a = 0.0, b = 0.0;
        const double* w = ws + ((long)n * groups + threadIdx.x) * chunks * 2;
        for (int k = 0; k < chunks; ++k) { a += w[2 * k]; b += w[2 * k + 1]; }
        double cnt = (double)HW * (C / groups);
        double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[threadIdx.x] = (float)mean;
        s_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cpg = C / groups;
    for (int c = threadIdx.x; c < C; c += 256) {
        float sc = s_rstd[c / cpg] * gamma[c];
        out_scale[(long)n * C + c] = sc;
        out_shift[(long)n * C + c] = beta[c] - s_mean[c / cpg] * sc;
    }
}

// All FPN levels of one tower conv in one launch each (the per-level kernels are launch-latency sized).
constexpr int GN_MAXL = 5;
struct GnLevels {
    const float* x[GN_MAXL];
    float* out_scale[GN_MAXL];
    float* out_shift[GN_MAXL];
    int HW[GN_MAXL];
    int nlev;
};

__global__ __launch_bounds__(256) void gn_stats_multi_kernel(const GnLevels L, double* __restrict__ ws, int N, int C, int groups, int chunks) {
    __shared__ double rs[256], rss[256];
    const int chunk = blockIdx.x, n = blockIdx.y, l = blockIdx.z;
    const int HW = L.HW[l];
    const int G = C >> 2;
    const int ppl = 256 / G;
    const int per = cdiv(HW, chunks);
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    const float* xn = L.x[l] + (long)n * HW * C;
    double s = 0.0, ss = 0.0;
    const int g = threadIdx.x % G, pl = threadIdx.x / G;
    if (pl < ppl) {
        for (int p = p0 + pl; p < p1; p += ppl) {
            f32x4 v = *reinterpret_cast<const f32x4*>(xn + (long)p * C + g * 4);
            s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    }
    rs[threadIdx.x] = s;
    rss[threadIdx.x] = ss;
    __syncthreads();
    if (threadIdx.x < groups) {
        const int f4pg = (C / groups) >> 2;
        double a = 0.0, b = 0.0;
        for (int k = 0; k < ppl; ++k)
            for (int q = 0; q < f4pg; ++q) {
                a += rs[k * G + threadIdx.x * f4pg + q];
                b += rss[k * G + threadIdx.x * f4pg + q];
            }
        double* o = ws + ((((long)l * N + n) * groups + threadIdx.x) * chunks + chunk) * 2;
        o[0] = a;
        o[1] = b;
    }
}

__global__ __launch_bounds__(256) void gn_finalize_multi_kernel(const GnLevels L, const double* __restrict__ ws, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int N, int C, int groups, int chunks, float eps) {
    __shared__ float s_mean[64], s_rstd[64];
    const int n = blockIdx.x, l = blockIdx.y;
    if (threadIdx.x < groups) {
        double a = 0.0, b = 0.0;
        const double* w = ws + (((long)l * N + n) * groups + threadIdx.x) * chunks * 2;
        for (int k = 0; k < chunks; ++k) { a += w[2 * k]; b += w[2 * k + 1]; }
        double cnt = (double)L.HW[l] * (C / groups);
        double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[threadIdx.x] = (float)mean;
        s_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cpg = C / groups;
    for (int c = threadIdx.x; c < C; c += 256) {
        float sc = s_rstd[c / cpg] * gamma[c];
        L.out_scale[l][(long)n * C + c] = sc;
        L.out_shift[l][(long)n * C + c] = beta[c] - s_mean[c / cpg] * sc;
    }
}

static inline int stream_grid(long work_items) {
    long b = (work_items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace cmk

using namespace cmk;

extern "C" int cmk_stem_conv_nchw3(const float* x, const float* w, const float* scale, const float* shift, float* y, int N, int H,
                                   int W, int Cout, void* stream) {
    if (!x || !w || !scale || !shift || !y) return fail(CMK_EINVAL, "stem: null pointer%s", "");
    if (Cout != 32 && Cout != 64 && Cout != 128) return fail(CMK_EINVAL, "stem: Cout (%s%ld) must be 32, 64 or 128", "", (long)Cout);
    int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    long total = (long)N * Ho * Wo;
    if ((long)3 * H * W >= (1L << 31)) return fail(CMK_EINVAL, "stem: image too large%s", "");
    long tiles = (total + 31) / 32;
    unsigned grid = (unsigned)std::min<long>((tiles + 3) / 4, 256 * 16);
    if (Cout == 32)
        hipLaunchKernelGGL(stem_conv_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, y, N, H, W, Ho, Wo, tiles);
    else if (Cout == 64)
        hipLaunchKernelGGL(stem_conv_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, y, N, H, W, Ho, Wo, tiles);
    else
        hipLaunchKernelGGL(stem_conv_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, y, N, H, W, Ho, Wo, tiles);
    return check_launch("stem_conv");
}

extern "C" int cmk_dwconv3x3_nhwc(const float* x, int x_cs, int x_co, const float* w, float* y, int y_cs, int y_co, int N, int H, int W,
                                  int C, int stride, void* stream) {
    if (!x || !w || !y) return fail(CMK_EINVAL, "dwconv: null pointer%s", "");
    if ((C & 3) || (x_cs & 3) || (x_co & 3) || (y_cs & 3) || (y_co & 3) || C < 4) return fail(CMK_EINVAL, "dwconv: channels must be multiples of 4%s", "");
    if (stride != 1 && stride != 2) return fail(CMK_EINVAL, "dwconv: stride must be 1 or 2%s", "");
    if (N < 1 || H < 1 || W < 1) return fail(CMK_EINVAL, "dwconv: empty input%s", "");
    int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    if (stride == 1) {
        long total = (long)N * Ho * ((Wo + 3) / 4) * (C >> 2);
        hipLaunchKernelGGL((dwconv3_kernel<1, 4>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, w, y, y_cs,
                           y_co, N, H, W, Ho, Wo, C >> 2);
    } else {
        long total = (long)N * Ho * ((Wo + 1) / 2) * (C >> 2);
        hipLaunchKernelGGL((dwconv3_kernel<2, 2>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, w, y, y_cs,
                           y_co, N, H, W, Ho, Wo, C >> 2);
    }
    return check_launch("dwconv3");
}

extern "C" int cmk_maxpool3x3s2_ceil_nhwc(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co, int N, int H, int W, int C,
                                          const float* gate, void* stream) {
    if (!x || !y) return fail(CMK_EINVAL, "maxpool: null pointer%s", "");
    if ((C & 3) || (x_cs & 3) || (x_co & 3) || (y_cs & 3) || (y_co & 3)) return fail(CMK_EINVAL, "maxpool: channels must be multiples of 4%s", "");
    if (H < 3 || W < 3) return fail(CMK_EINVAL, "maxpool: input smaller than the window%s", "");
    int Ho = (H - 3 + 1) / 2 + 1, Wo = (W - 3 + 1) / 2 + 1;  // ceil((H-3)/2)+1
    if ((Ho - 1) * 2 >= H) --Ho;
    if ((Wo - 1) * 2 >= W) --Wo;
    long total = (long)N * Ho * Wo * (C >> 2);
    hipLaunchKernelGGL(maxpool3_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, y, y_cs, y_co, N, H, W,
                       Ho, Wo, C >> 2, gate);
    return check_launch("maxpool3");
}

extern "C" int cmk_maxpool1x1s2_nhwc(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co, int N, int H, int W, int C, void* stream) {
    if (!x || !y) return fail(CMK_EINVAL, "maxpool1x1s2: null pointer%s", "");
    if ((C & 3) || (x_cs & 3) || (x_co & 3) || (y_cs & 3) || (y_co & 3) || N < 1 || H < 1 || W < 1)
        return fail(CMK_EINVAL, "maxpool1x1s2: channels must be multiples of 4, N, H, W >= 1%s", "");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    long total = (long)N * Ho * Wo * (C >> 2);
    hipLaunchKernelGGL(subsample2_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, y, y_cs, y_co, N, H, W, Ho, Wo,
                       C >> 2);
    return check_launch("maxpool1x1s2");
}

extern "C" int cmk_ese_gate(const float* x, int x_cs, int x_co, const float* fc_w, const float* fc_b, float* gate, float* ws,
                            int ws_chunks, int N, int HW, int C, void* stream) {
    if (!x || !fc_w || !fc_b || !gate || !ws) return fail(CMK_EINVAL, "ese_gate: null pointer%s", "");
    if ((C & 3) || (x_cs & 3) || (x_co & 3) || ws_chunks < 1) return fail(CMK_EINVAL, "ese_gate: bad shape%s", "");
    int G = C >> 2;
    int ppl = G >= 256 ? 1 : 256 / G;
    size_t lds1 = (size_t)ppl * C * sizeof(float);
    hipLaunchKernelGGL(ese_partial_kernel, dim3(ws_chunks, N), dim3(256), lds1, (hipStream_t)stream, x, x_cs, x_co, ws, HW, C, ws_chunks);
    int rc = check_launch("ese_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(ese_fc_kernel, dim3(cdiv(C, 16), N), dim3(256), 2 * lds1 /* doubles */, (hipStream_t)stream, ws, fc_w, fc_b,
                       gate, HW, C, ws_chunks);
    return check_launch("ese_fc");
}

extern "C" int cmk_ese_gate_pooled(const float* pool_ws, int rows, const float* fc_w, const float* fc_b, float* gate, int N, int HW, int C, void* stream) {
    if (!pool_ws || !fc_w || !fc_b || !gate) return fail(CMK_EINVAL, "ese_gate_pooled: null pointer%s", "");
    if ((C & 3) || rows < 1 || HW < rows || N < 1) return fail(CMK_EINVAL, "ese_gate_pooled: bad shape (C %% 4, H*W >= rows)%s", "");
    const int G = C >> 2;
    const int parts = G >= 256 ? 1 : 256 / G;
    hipLaunchKernelGGL(ese_fc_pooled_kernel, dim3(cdiv(C, 16), N), dim3(256), (size_t)parts * C * sizeof(double), (hipStream_t)stream, pool_ws, rows,
                       fc_w, fc_b, gate, HW, C);
    return check_launch("ese_fc_pooled");
}

extern "C" int cmk_ese_scale(const float* x, int x_cs, int x_co, const float* gate, const float* identity, int id_cs, int id_co, float* y,
                             int y_cs, int y_co, int N, int HW, int C, void* stream) {
    if (!x || !gate || !y) return fail(CMK_EINVAL, "ese_scale: null pointer%s", "");
    if ((C & 3) || (x_cs & 3) || (x_co & 3) || (y_cs & 3) || (y_co & 3) || (id_cs & 3) || (id_co & 3))
        return fail(CMK_EINVAL, "ese_scale: channels must be multiples of 4%s", "");
    long total = (long)N * HW * (C >> 2);
    hipLaunchKernelGGL(ese_scale_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, gate, identity, id_cs,
                       id_co, y, y_cs, y_co, N, HW, C >> 2);
    return check_launch("ese_scale");
}

// nearest-neighbour 2x upsampling of `coarse` added to `y` in place (d2 FPN's top-down path, ctor at vovnet.py:547-554, when a norm sits between the
// lateral conv and the sum, so that the sum cannot ride in the conv's epilogue)
__global__ __launch_bounds__(256) void upsample2x_add_kernel(float* __restrict__ y, const float* __restrict__ c, int N, int H, int W, int Hc, int Wc, int C4) {
    long total = (long)N * H * W * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4 = (int)(i % C4);
        long p = i / C4;
        int w = (int)(p % W);
        int h = (int)((p / W) % H);
        int n = (int)(p / ((long)W * H));
        const f32x4 a = *reinterpret_cast<const f32x4*>(y + i * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(c + ((((long)n * Hc + (h >> 1)) * Wc + (w >> 1)) * C4 + c4) * 4);
        *reinterpret_cast<f32x4*>(y + i * 4) = f32x4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
    }
}

extern "C" int cmk_upsample2x_add_nhwc(float* y, const float* coarse, int N, int H, int W, int Hc, int Wc, int C, void* stream) {
    if (!y || !coarse) return fail(CMK_EINVAL, "upsample2x_add: null pointer%s", "");
    if ((C & 3) || N < 1 || H < 1 || W < 1 || Hc * 2 < H || Wc * 2 < W) return fail(CMK_EINVAL, "upsample2x_add: C %% 4 == 0 and a coarse map of at least half the size%s", "");
    long total = (long)N * H * W * (C >> 2);
    hipLaunchKernelGGL(upsample2x_add_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, y, coarse, N, H, W, Hc, Wc, C >> 2);
    return check_launch("upsample2x_add");
}

static int groupnorm_inplace(float* x, const float* gamma, const float* beta, double* ws, int ws_chunks, int N, int HW, int C, int groups, float eps,
                             float lo, void* stream);

extern "C" int cmk_groupnorm_relu_nhwc(float* x, const float* gamma, const float* beta, double* ws, int ws_chunks, int N, int HW, int C,
                                       int groups, float eps, void* stream) {
    return groupnorm_inplace(x, gamma, beta, ws, ws_chunks, N, HW, C, groups, eps, 0.f, stream);
}

extern "C" int cmk_groupnorm_nhwc(float* x, const float* gamma, const float* beta, double* ws, int ws_chunks, int N, int HW, int C,
                                  int groups, float eps, void* stream) {
    return groupnorm_inplace(x, gamma, beta, ws, ws_chunks, N, HW, C, groups, eps, -INFINITY, stream);
}

static int groupnorm_inplace(float* x, const float* gamma, const float* beta, double* ws, int ws_chunks, int N, int HW, int C, int groups, float eps,
                             float lo, void* stream) {
    if (!x || !gamma || !beta || !ws) return fail(CMK_EINVAL, "groupnorm: null pointer%s", "");
    if ((C & 3) || C > 1024 || groups < 1 || groups > 64 || C % groups || ((C / groups) & 3) || 256 % (C >> 2) || ws_chunks < 1)
        return fail(CMK_EINVAL, "groupnorm: unsupported C/groups%s", "");
    hipLaunchKernelGGL(gn_stats_kernel, dim3(ws_chunks, N), dim3(256), 0, (hipStream_t)stream, x, ws, HW, C, groups, ws_chunks);
    int rc = check_launch("gn_stats");
    if (rc) return rc;
    long total = (long)HW * (C >> 2);
    int bpi = stream_grid(total);
    if (bpi > 1024) bpi = 1024;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(bpi, N), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, ws, HW, C, groups, ws_chunks, eps,
                       bpi, lo);
    return check_launch("gn_apply");
}

extern "C" int cmk_groupnorm_affine(const float* x, const float* gamma, const float* beta, double* ws, int ws_chunks, int N, int HW, int C,
                                    int groups, float eps, float* out_scale, float* out_shift, void* stream) {
    if (!x || !gamma || !beta || !ws || !out_scale || !out_shift) return fail(CMK_EINVAL, "groupnorm_affine: null pointer%s", "");
    if ((C & 3) || C > 1024 || groups < 1 || groups > 64 || C % groups || ((C / groups) & 3) || 256 % (C >> 2) || ws_chunks < 1)
        return fail(CMK_EINVAL, "groupnorm_affine: unsupported C/groups%s", "");
    hipLaunchKernelGGL(gn_stats_kernel, dim3(ws_chunks, N), dim3(256), 0, (hipStream_t)stream, x, ws, HW, C, groups, ws_chunks);
    int rc = check_launch("gn_stats");
    if (rc) return rc;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, ws, gamma, beta, out_scale, out_shift, HW, C, groups,
                       ws_chunks, eps);
    return check_launch("gn_finalize");
}

struct GnTileLevels {
    float* out_scale[GN_MAXL];
    float* out_shift[GN_MAXL];
    int HW[GN_MAXL], tile_begin[GN_MAXL], tiles[GN_MAXL];
};
// statistics written by the conv epilogue (cmk_conv_desc.gn_ws): records ((tile*2 + parity)*groups + group) x {sum, sumsq}
__global__ __launch_bounds__(256) void gn_finalize_tiles_kernel(const GnTileLevels L, const double* __restrict__ ws, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int N, int C, int groups, float eps) {
    __shared__ double rs[256], rss[256];
    __shared__ float s_mean[64], s_rstd[64];
    const int n = blockIdx.x, l = blockIdx.y;
    const int parts = 256 / groups;                       // thread = (part, group): consecutive threads read consecutive groups
    const int g = threadIdx.x % groups, part = threadIdx.x / groups;
    double a = 0.0, b = 0.0;
    if (part < parts) {
        const long r0 = (long)L.tile_begin[l] + (long)n * L.tiles[l];      // "tiles" here = records per image
        const int nrec = L.tiles[l];
        double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, a3 = 0.0, b3 = 0.0;      // four independent chains: the loads overlap
        int r = part;
        for (; r + 3 * parts < nrec; r += 4 * parts) {
            const double* w = ws + ((r0 + r) * groups + g) * 2;
            const long st = (long)parts * groups * 2;
            a += w[0]; b += w[1];
            a1 += w[st]; b1 += w[st + 1];
            a2 += w[2 * st]; b2 += w[2 * st + 1];
            a3 += w[3 * st]; b3 += w[3 * st + 1];
        }
        for (; r < nrec; r += parts) {
            const double* w = ws + ((r0 + r) * groups + g) * 2;
            a += w[0];
            b += w[1];
        }
        a += a1 + a2 + a3;
        b += b1 + b2 + b3;
    }
    rs[threadIdx.x] = a;
    rss[threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < groups) {
        a = 0.0; b = 0.0;
        for (int k = 0; k < parts; ++k) { a += rs[k * groups + threadIdx.x]; b += rss[k * groups + threadIdx.x]; }
        const double cnt = (double)L.HW[l] * (C / groups);
        const double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[threadIdx.x] = (float)mean;
        s_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cpg = C / groups;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float sc = s_rstd[c / cpg] * gamma[c];
        L.out_scale[l][(long)n * C + c] = sc;
        L.out_shift[l][(long)n * C + c] = beta[c] - s_mean[c / cpg] * sc;
    }
}

extern "C" int cmk_groupnorm_affine_tiles(const double* ws, const int* Hs, const int* Ws, const int* recs, int nlev, const float* gamma, const float* beta, int N,
                                          int C, int groups, float eps, float* const* out_scale, float* const* out_shift, void* stream) {
    if (!ws || !Hs || !Ws || !recs || !gamma || !beta || !out_scale || !out_shift) return fail(CMK_EINVAL, "groupnorm_affine_tiles: null pointer%s", "");
    if (nlev < 1 || nlev > GN_MAXL || N < 1) return fail(CMK_EINVAL, "groupnorm_affine_tiles: 1..5 levels%s", "");
    if (groups < 1 || groups > 64 || C % groups || C > 4096) return fail(CMK_EINVAL, "groupnorm_affine_tiles: unsupported C/groups%s", "");
    GnTileLevels L;
    int begin = 0;
    for (int l = 0; l < GN_MAXL; ++l) {
        const bool ok = l < nlev;
        if (ok && (!out_scale[l] || !out_shift[l] || Hs[l] < 1 || Ws[l] < 1 || recs[l] < 1)) return fail(CMK_EINVAL, "groupnorm_affine_tiles: bad level%s", "");
        L.out_scale[l] = ok ? out_scale[l] : nullptr; L.out_shift[l] = ok ? out_shift[l] : nullptr;
        L.HW[l] = ok ? Hs[l] * Ws[l] : 1;
        L.tiles[l] = ok ? recs[l] : 0;                                     // cmk_conv_gn_records of the producing conv
        L.tile_begin[l] = begin;
        begin += N * L.tiles[l];
    }
    hipLaunchKernelGGL(gn_finalize_tiles_kernel, dim3(N, nlev), dim3(256), 0, (hipStream_t)stream, L, ws, gamma, beta, N, C, groups, eps);
    return check_launch("gn_finalize_tiles");
}

extern "C" int cmk_groupnorm_affine_multi(const float* const* xs, const int* HWs, int nlev, const float* gamma, const float* beta, double* ws,
                                          int ws_chunks, int N, int C, int groups, float eps, float* const* out_scale, float* const* out_shift,
                                          void* stream) {
    if (!xs || !HWs || !gamma || !beta || !ws || !out_scale || !out_shift) return fail(CMK_EINVAL, "groupnorm_affine_multi: null pointer%s", "");
    if (nlev < 1 || nlev > GN_MAXL) return fail(CMK_EINVAL, "groupnorm_affine_multi: 1..5 levels%s", "");
    if ((C & 3) || C > 1024 || groups < 1 || groups > 64 || C % groups || ((C / groups) & 3) || 256 % (C >> 2) || ws_chunks < 1)
        return fail(CMK_EINVAL, "groupnorm_affine_multi: unsupported C/groups%s", "");
    GnLevels L;
    L.nlev = nlev;
    for (int l = 0; l < GN_MAXL; ++l) {
        bool ok = l < nlev;
        if (ok && (!xs[l] || !out_scale[l] || !out_shift[l] || HWs[l] < 1)) return fail(CMK_EINVAL, "groupnorm_affine_multi: bad level%s", "");
        L.x[l] = ok ? xs[l] : nullptr; L.out_scale[l] = ok ? out_scale[l] : nullptr; L.out_shift[l] = ok ? out_shift[l] : nullptr;
        L.HW[l] = ok ? HWs[l] : 1;
    }
    hipLaunchKernelGGL(gn_stats_multi_kernel, dim3(ws_chunks, N, nlev), dim3(256), 0, (hipStream_t)stream, L, ws, N, C, groups, ws_chunks);
    int rc = check_launch("gn_stats_multi");
    if (rc) return rc;
    hipLaunchKernelGGL(gn_finalize_multi_kernel, dim3(N, nlev), dim3(256), 0, (hipStream_t)stream, L, ws, gamma, beta, N, C, groups, ws_chunks, eps);
    return check_launch("gn_finalize_multi");
}
