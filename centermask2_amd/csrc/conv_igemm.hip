// Convolution (3x3 s1/s2, 1x1, and FC as 1x1 over "pixels" = rows) as an implicit GEMM on the CDNA4 matrix pipe.
//
//   GEMM view:  M = output pixels, N = output channels, K = taps * Cin, all fp32.
//   Instruction: v_mfma_f32_32x32x2_f32 — exact fp32 (bitwise a k-ordered fmaf chain), 64 FLOP/clk/SIMD, the same
//   peak as the fp32 VALU but one VGPR per operand, so the tile is fed from LDS with ds_read_b128 instead of
//   per-FMA register traffic.  A 32x32 accumulator's column sits on the lane, so Cout is mapped to the lane
//   (128-byte contiguous NHWC stores) and pixels to the accumulator rows.
//
//   Tiling (block = 4 waves stacked along M, wave tile = WM x WN accumulators of 32 pixels x 32 couts):
//     3x3:  spatial tile of (4*WM*SR) x SC output pixels, a wave sub-tile being SR x SC = 2x16 or 1x32 pixels; the
//           input halo tile is staged ONCE per 16-channel K chunk in LDS and the 9 taps read it at shifted addresses
//           (9x fewer global->LDS bytes than per-tap im2col).
//     1x1:  128*WM consecutive pixels of the flattened (N*H*W) axis.
//     Per K chunk and tap the block stages a [32*WN couts][16 ci] weight slab (pre-packed, contiguous in HBM).
//   K order inside a 16-chunk is permuted so that MFMA k-step s of lane half h uses channel 8h+s: every lane then
//   reads 8 contiguous floats per operand row (2 x ds_read_b128) for 8 MFMAs.  Rows are padded to 20 floats
//   (80 B), which spreads the 16-lane ds_read_b128 groups over all 16-byte LDS slots.
//   Pipeline: global loads for step s+1 (weights) and chunk c+1 (halo) are issued before the barrier of step s
//   and written to the other LDS buffer after its MFMAs (register-staged double buffering, one barrier per step).
//   Staging loads are branch-free (clamped address, zero selected at the LDS write) so the compiler keeps them in
//   flight across the MFMA block.
//   Scheduling: the matrix pipe is the bound, so what matters is that every CU holds the same number of equally long
//   workgroups.  The host picks (WM, sub-tile shape) per layer from a small menu with a residency/round cost model
//   (choose_variant), and layers that share weights across FPN levels (FCOS towers/predictors) run as ONE launch
//   whose block index walks the tiles of all levels.
//
// Reference call sites replaced: aten::conv2d + FrozenBN + ReLU vovnet.py:205-236; d2 FPN convs (vovnet.py:547-554);
// fpn.py:27-35; fcos.py:169-200; sam.py:58-83; maskiou_head.py:81-93; nn.Linear maskiou_head.py:89-91.
#include <stdlib.h>

#include "conv_args.hpp"

namespace cmk {

template <int TAPS, int STRIDE, int WM, int WN, int SC>
struct Geo {
    static constexpr int SR = 32 / SC;   // sub-tile rows
    static constexpr int SUBT = 4 * WM;  // 32-pixel sub-tiles per block
    static constexpr int BM = 32 * SUBT;
    static constexpr int BN = 32 * WN;
    static constexpr int TH = (TAPS == 9) ? SR * SUBT : 1;
    static constexpr int TW = (TAPS == 9) ? SC : BM;
    static constexpr int HH = (TAPS == 9) ? (TH - 1) * STRIDE + 3 : 1;
    static constexpr int HWD = (TAPS == 9) ? (TW - 1) * STRIDE + 3 : BM;
    static constexpr int APIX = HH * HWD;
    static constexpr int A_BYTES = APIX * PST * 4;
    static constexpr int B_BYTES = BN * PST * 4;
    static constexpr int OCC = occ_of(WM, WN, STRIDE);
    static constexpr bool ADB = (2 * A_BYTES + 2 * B_BYTES) * OCC <= LDS_CU;   // double-buffer the halo only if residency is kept
    static constexpr int LDS_BYTES = (ADB ? 2 : 1) * A_BYTES + 2 * B_BYTES;
    static constexpr int RESIDENT = (LDS_BYTES * OCC <= LDS_CU) ? OCC : (LDS_BYTES * 2 <= LDS_CU ? 2 : 1);
    static constexpr int A_ITERS = (APIX * 4 + 255) / 256;
    static constexpr int B_ITERS = (BN * 4 + 255) / 256;
};

// GA ("gather") form: a 3x3 conv run as a flattened-pixel GEMM (TAPS == 1 geometry) whose K walks 9 taps x Cin/16 chunks; each
// thread gathers its A rows per tap straight from the image (zero outside).  No halo reuse, so it only pays on maps too small
// to fill the spatial tiles (7x7 RoI maps, P6/P7).
template <int TAPS, int STRIDE, int WM, int WN, int SC, bool GA = false>
__global__ __launch_bounds__(256, (occ_of(WM, WN, STRIDE))) void conv_igemm_kernel(const ConvArgs a) {
    using G = Geo<TAPS, STRIDE, WM, WN, SC>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + (G::ADB ? 2 : 1) * G::APIX * PST;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hh = lane >> 5;  // k half
    const int li = lane & 31;

    // ---- which problem (FPN level) and which tile ------------------------------------------------------------
    // XCD-aware order (see conv_wino4r_kernel): the grid_y workgroups of one input tile go to one XCD, back to back
    const int xq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int bx = (xq / a.grid_y) * 8 + xcd, by = xq % a.grid_y;
    if (bx >= a.total_tiles) return;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAXP; ++i)
        if (i < a.nprob && bx >= a.p[i].tile_begin) pi = i;
    const ConvProblem& P = a.p[pi];
    const int H = P.H, W = P.W, Ho = P.Ho, Wo = P.Wo;
    const long total_pix = P.total_pix;
    const int tile = bx - P.tile_begin;
    int n = 0, oh0 = 0, ow0 = 0;
    long pix0 = 0;
    if (TAPS == 9) {
        int tw = tile % P.tiles_w;
        int t2 = tile / P.tiles_w;
        int th = t2 % P.tiles_h;
        n = t2 / P.tiles_h;
        oh0 = th * G::TH;
        ow0 = tw * G::TW;
    } else {
        pix0 = (long)tile * G::BM;
    }
    const int co0 = by * G::BN;
    const int cin_chunks = a.Cin >> 4;
    const int nchunks = GA ? 9 * cin_chunks : cin_chunks;
    const int c_lo = (int)((long)blockIdx.y * nchunks / a.ksplit), c_hi = (int)((long)(blockIdx.y + 1) * nchunks / a.ksplit);  // even bounds (host)
    const int total_steps = c_hi * TAPS;

    // ---- per-thread staging descriptors ------------------------------------------------------------------------
    const float* xin = P.x + (TAPS == 9 ? (long)n * H * W * a.x_cs : 0L) + a.x_co;
    long a_goff[G::A_ITERS];   // clamped to a valid address; a_ok tells whether the value is used
    int ga_ih0[GA ? G::A_ITERS : 1], ga_iw0[GA ? G::A_ITERS : 1];   // GA: top-left input coordinate of the row's 3x3 window
    unsigned a_ok = 0;
#pragma unroll
    for (int it = 0; it < G::A_ITERS; ++it) {
        int idx = it * 256 + tid;
        int pix = idx >> 2, q = idx & 3;
        long off = 0;
        if (idx < G::APIX * 4) {
            if (GA) {
                long Pp = pix0 + pix;
                const bool pv = Pp < total_pix;
                if (!pv) Pp = 0;
                const long hw = (long)Ho * Wo;
                const int n_ = (int)(Pp / hw);
                const int rem = (int)(Pp - (long)n_ * hw);
                const int oh = rem / Wo, ow = rem - oh * Wo;
                ga_ih0[it] = pv ? oh * a.ga_stride - 1 : -4;      // -4: every tap lands outside
                ga_iw0[it] = ow * a.ga_stride - 1;
                off = (long)n_ * H * W * a.x_cs + q * 4;
            } else if (TAPS == 9) {
                int hr = pix / G::HWD, hc = pix - hr * G::HWD;
                int ih = oh0 * STRIDE - 1 + hr, iw = ow0 * STRIDE - 1 + hc;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) { off = ((long)ih * W + iw) * a.x_cs + q * 4; a_ok |= 1u << it; }
            } else {
                long Pp = pix0 + pix;
                if (Pp < total_pix) { off = Pp * a.x_cs + q * 4; a_ok |= 1u << it; }
            }
        }
        a_goff[it] = off;
    }
    f32x4 a_stage[G::A_ITERS];
    f32x4 b_stage[G::B_ITERS];
    // fused input affine (GroupNorm apply + ReLU of the producer): the channel quad of a thread is fixed (idx & 3 == tid & 3)
    const bool has_aff = P.in_scale != nullptr;
    int aff_n = n;                                   // 1x1 (flattened pixels): image index of this tile's first pixel; tiles never
    if (TAPS != 9 && has_aff) aff_n = (int)(pix0 / ((long)Ho * Wo));   // straddle images when the affine is used (checked on the host)
    const float* aff_s = has_aff ? P.in_scale + (long)aff_n * a.Cin + (tid & 3) * 4 : nullptr;
    const float* aff_b = has_aff ? P.in_shift + (long)aff_n * a.Cin + (tid & 3) * 4 : nullptr;
    f32x4 in_sc = {1.f, 1.f, 1.f, 1.f}, in_sh = {0.f, 0.f, 0.f, 0.f};

    auto load_A = [&](int chunk) {
        if (GA) {
            const int tap = chunk / cin_chunks, ch = chunk - tap * cin_chunks;
            const int kh = tap / 3, kw = tap - kh * 3;
            a_ok = 0;
#pragma unroll
            for (int it = 0; it < G::A_ITERS; ++it) {
                const int ih = ga_ih0[it] + kh, iw = ga_iw0[it] + kw;
                const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
                const long off = ok ? a_goff[it] + ((long)ih * W + iw) * a.x_cs + ch * 16 : 0L;
                a_stage[it] = *reinterpret_cast<const f32x4*>(xin + off);
                a_ok |= (ok ? 1u : 0u) << it;
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < G::A_ITERS; ++it) a_stage[it] = *reinterpret_cast<const f32x4*>(xin + a_goff[it] + chunk * 16);
        if (has_aff) {
            in_sc = *reinterpret_cast<const f32x4*>(aff_s + chunk * 16);
            in_sh = *reinterpret_cast<const f32x4*>(aff_b + chunk * 16);
        }
    };
    auto store_A = [&](int buf) {
        float* dst = sA + buf * (G::APIX * PST);
#pragma unroll
        for (int it = 0; it < G::A_ITERS; ++it) {
            int idx = it * 256 + tid;
            if ((it + 1) * 256 <= G::APIX * 4 || idx < G::APIX * 4) {
                f32x4 v = a_stage[it];
                const bool ok = (a_ok >> it) & 1u;
                // rare, wave-uniform options: real branches (the empty asm keeps hipcc from if-converting them into selects that
                // every conv would execute — each VALU instruction here costs the matrix pipe ~4 cycles, tools/probe/mfma_probe2)
                if (has_aff) {
                    asm volatile("" ::: "memory");
                    v.x = fmaxf(v.x * in_sc.x + in_sh.x, 0.f); v.y = fmaxf(v.y * in_sc.y + in_sh.y, 0.f);
                    v.z = fmaxf(v.z * in_sc.z + in_sh.z, 0.f); v.w = fmaxf(v.w * in_sc.w + in_sh.w, 0.f);
                }
                if (a.in_relu) {
                    asm volatile("" ::: "memory");
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                v.x = ok ? v.x : 0.f;
                v.y = ok ? v.y : 0.f;
                v.z = ok ? v.z : 0.f;
                v.w = ok ? v.w : 0.f;
                *reinterpret_cast<f32x4*>(dst + (idx >> 2) * PST + (idx & 3) * 4) = v;
            }
        }
    };
    auto load_B = [&](int step) {
        int chunk = step / TAPS, tap = step - chunk * TAPS;
        const float* wsrc = a.w + ((long)(tap * nchunks + chunk) * a.cout_pad + co0) * 16;
#pragma unroll
        for (int it = 0; it < G::B_ITERS; ++it) {
            int idx = it * 256 + tid;
            if ((it + 1) * 256 > G::BN * 4) idx = min(idx, G::BN * 4 - 1);   // ragged last iteration: clamp, do not branch
            b_stage[it] = *reinterpret_cast<const f32x4*>(wsrc + idx * 4);
        }
    };
    auto store_B = [&](int buf) {
        float* dst = sB + buf * (G::BN * PST);
#pragma unroll
        for (int it = 0; it < G::B_ITERS; ++it) {
            int idx = it * 256 + tid;
            if ((it + 1) * 256 <= G::BN * 4 || idx < G::BN * 4)
                *reinterpret_cast<f32x4*>(dst + (idx >> 2) * PST + (idx & 3) * 4) = b_stage[it];
        }
    };

    // ---- MFMA operand addresses -----------------------------------------------------------------------------------
    int a_off[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        int u = wave * WM + m;
        if (TAPS == 9) {
            int r = li / SC, cc = li % SC;
            a_off[m] = (((u * G::SR + r) * STRIDE) * G::HWD + cc * STRIDE) * PST + hh * 8;
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

    // ---- prologue ---------------------------------------------------------------------------------------------------
    load_A(c_lo);
    load_B(c_lo * TAPS);
    store_A(0);          // c_lo is even, so buffer parity (c & 1, step & 1) starts at 0
    store_B(0);

    int step = c_lo * TAPS;
    for (int c = c_lo; c < c_hi; ++c) {
        const bool has_next_chunk = (c + 1 < c_hi);
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

    // ---- split-K: raw partial sums to the workspace, the reduce kernel applies the epilogue ----------------------------
    if (a.ksplit > 1) {
        float* wz = a.ws + (long)blockIdx.y * total_pix * a.cout_pad + co0 + li;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const int u = wave * WM + m;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
                long opix;
                bool pvalid;
                if (TAPS == 9) {
                    const int oh = oh0 + u * G::SR + row / SC, ow = ow0 + row % SC;
                    pvalid = (oh < Ho) && (ow < Wo);
                    opix = ((long)n * Ho + oh) * Wo + ow;
                } else {
                    opix = pix0 + u * 32 + row;
                    pvalid = opix < total_pix;
                }
                if (pvalid) {
#pragma unroll
                    for (int nn = 0; nn < WN; ++nn) wz[opix * a.cout_pad + nn * 32] = acc[m][nn][r];
                }
            }
        }
        return;
    }

    // ---- epilogue: scale/shift (+residual) (+ReLU), NHWC store -------------------------------------------------------
#pragma unroll
    for (int nn = 0; nn < WN; ++nn) {
        const int co = co0 + nn * 32 + li;
        const bool cvalid = co < a.Cout;
        const float sc = cvalid ? P.scale[co] : 0.f;
        const float sh = cvalid ? P.shift[co] : 0.f;
        const bool do_relu = co < a.relu_upto;
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
                    oh = oh0 + u * G::SR + row / SC;
                    ow = ow0 + row % SC;
                    pvalid = (oh < Ho) && (ow < Wo);
                    opix = ((long)n * Ho + oh) * Wo + ow;
                } else {
                    opix = pix0 + u * 32 + row;
                    pvalid = opix < total_pix;
                }
                if (cvalid && pvalid) {
                    float v = acc[m][nn][r] * sc + sh;
                    if (a.res_mode == 1) {
                        v += a.res[opix * a.res_cs + a.res_co + co];
                    } else if (a.res_mode == 2) {
                        int nn_ = n;
                        if (TAPS != 9) {  // recover (n, oh, ow) from the flattened pixel index
                            long hw = (long)Ho * Wo;
                            nn_ = (int)(opix / hw);
                            int rem = (int)(opix - (long)nn_ * hw);
                            oh = rem / Wo;
                            ow = rem - oh * Wo;
                        }
                        long rp = ((long)nn_ * a.Hr + (oh >> 1)) * a.Wr + (ow >> 1);
                        v += a.res[rp * a.res_cs + a.res_co + co];
                    }
                    if (do_relu) v = fmaxf(v, 0.f);
                    P.y[opix * a.y_cs + a.y_co + co] = v;
                }
            }
        }
    }
}


// split-K second pass: y = epilogue(sum_z ws[z]); one thread = 4 couts of one pixel
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int ksplit, long total_pix, int cout_pad,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, int Cout,
                                                           int relu_upto, const float* __restrict__ res, int res_cs, int res_co,
                                                           float* __restrict__ y, int y_cs, int y_co) {
    const int c4n = cout_pad >> 2;
    const long total = total_pix * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pix = i / c4n;
        const int co = (int)(i - pix * c4n) * 4;
        if (co >= Cout) continue;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < ksplit; ++z) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ws + ((long)z * total_pix + pix) * cout_pad + co);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = co + j;
            if (c < Cout) {
                float v = s[j] * scale[c] + shift[c];
                if (res) v += res[pix * res_cs + res_co + c];
                if (c < relu_upto) v = fmaxf(v, 0.f);
                y[pix * y_cs + y_co + c] = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) form for 3x3 stride-1 convs, fused in one kernel (input transform, 16 frequency GEMMs on the
// matrix pipe, output transform all on chip): 2.25x fewer MFMA flops than the direct form, still plain fp32 arithmetic
// (transform matrices hold only 0, +-1, +-1/2; results differ from the direct kernel by fp32 rounding only).
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        per 4x4 input patch d -> 2x2 outputs, summed over input channels
//   workgroup = 8x16 output pixels (32 tiles of 2x2) x 64 output channels, 4 waves = 2 frequency halves (fh) x 2 cout halves (ng);
//   a wave keeps 8 accumulators of 32 tiles x 32 couts (128 VGPRs), so two workgroups (78 KiB of LDS each) live on a CU and one
//   workgroup's staging / transform / barrier phases hide under the other's MFMAs;
//   per 16-channel chunk: halo (10x18 px) -> LDS, every thread transforms (tile, channel quad) patches into the 16 frequency planes
//   V[f][tile][ci] (16-byte chunks XOR-swizzled, conflict-free ds_read_b128 without padding);
//   a lane of a 32x32 accumulator holds every frequency of its (tile, cout) entries for its half, so the output transform is
//   per-lane register arithmetic; the two frequency halves swap partial sums through LDS once at the end.
// Weights: what bounded the earlier LDS-DMA forms was a latency chain, not throughput — a weight piece could only be requested one
// step ahead (two 16 KiB LDS buffers were all that fit) and an L2 round trip under load is about as long as a step, so every step
// waited for it (tools/probe/trace_wino.py, mfma_probe3).  Here each lane loads its own U operand pieces from global memory
// (L2-resident, shared by all workgroups) into registers TWO steps ahead (layout R = one contiguous KiB per wave load,
// cmk_conv_desc.w_wino); the LDS that a weight buffer would take holds a second V buffer, so the input transform of chunk c+1
// overlaps the MFMAs of chunk c with two barriers per chunk, none of which waits for memory.
// ---------------------------------------------------------------------------------------------------------------
constexpr int S_HALO = 10 * 18;
constexpr int S_SH = S_HALO * PST;                 // floats
constexpr int S_SV = 16 * 32 * 16;
constexpr int S_H_ITERS = (S_HALO * 4 + 255) / 256;
constexpr int R_LDS_BYTES = (S_SH + 2 * S_SV) * 4;
// AFF: the producer's GroupNorm+ReLU is applied while the halo is staged (FCOS tower convs 2-4 and the predictors).  Without it the
// staging registers of the affine are free and the weights are fetched three steps ahead instead of two.
template <bool AFF>
__global__ __launch_bounds__(256, 2) void conv_wino4r_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sH = smem;
    float* sV = smem + S_SH;                 // two V buffers (chunk parity)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, li = lane & 31;
    const int fh = wave >> 1, ng = wave & 1;

    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so the grid_y workgroups that
    // share an input tile are given to the SAME XCD, back to back: the tile's halo is fetched into one L2 once.
    // (b % 8 only says which workgroups share an XCD; nothing here depends on it for correctness.)
    const int xq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int bx = (xq / a.grid_y) * 8 + xcd, by = xq % a.grid_y;
    if (bx >= a.total_tiles) return;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAXP; ++i)
        if (i < a.nprob && bx >= a.p[i].tile_begin) pi = i;
    const ConvProblem& P = a.p[pi];
    const int H = P.H, W = P.W;
    const int tile = bx - P.tile_begin;
    const int tw = tile % P.tiles_w;
    const int t2 = tile / P.tiles_w;
    const int th = t2 % P.tiles_h;
    const int n = t2 / P.tiles_h;
    const int oh0 = th * 8, ow0 = tw * 16;
    const int co0 = by * 64;
    const int nchunks = a.Cin >> 4;

    const float* xin = P.x + (long)n * H * W * a.x_cs + a.x_co;
    long g_off[S_H_ITERS];
    unsigned ok = 0;
#pragma unroll
    for (int it = 0; it < S_H_ITERS; ++it) {
        int idx = it * 256 + tid;
        int pix = idx >> 2, q = idx & 3;
        long off = 0;
        if (idx < S_HALO * 4) {
            int hr = pix / 18, hc = pix - hr * 18;
            int ih = oh0 - 1 + hr, iw = ow0 - 1 + hc;
            if (ih >= 0 && ih < H && iw >= 0 && iw < W) { off = ((long)ih * W + iw) * a.x_cs + q * 4; ok |= 1u << it; }
        }
        g_off[it] = off;
    }
    f32x4 h_stage[S_H_ITERS];
    constexpr bool has_aff = AFF;                    // fused GroupNorm apply + ReLU of the producer
    const float* aff_s = has_aff ? P.in_scale + (long)n * a.Cin + (tid & 3) * 4 : nullptr;
    const float* aff_b = has_aff ? P.in_shift + (long)n * a.Cin + (tid & 3) * 4 : nullptr;
    f32x4 in_sc = {1.f, 1.f, 1.f, 1.f}, in_sh = {0.f, 0.f, 0.f, 0.f};
    auto load_H = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < S_H_ITERS; ++it) h_stage[it] = *reinterpret_cast<const f32x4*>(xin + g_off[it] + chunk * 16);
        if (has_aff) {
            in_sc = *reinterpret_cast<const f32x4*>(aff_s + chunk * 16);
            in_sh = *reinterpret_cast<const f32x4*>(aff_b + chunk * 16);
        }
    };
    auto store_H = [&]() {
#pragma unroll
        for (int it = 0; it < S_H_ITERS; ++it) {
            int idx = it * 256 + tid;
            if ((it + 1) * 256 <= S_HALO * 4 || idx < S_HALO * 4) {
                f32x4 v = h_stage[it];
                const bool k = (ok >> it) & 1u;
                if (has_aff) {
                    v.x = fmaxf(v.x * in_sc.x + in_sh.x, 0.f); v.y = fmaxf(v.y * in_sc.y + in_sh.y, 0.f);
                    v.z = fmaxf(v.z * in_sc.z + in_sh.z, 0.f); v.w = fmaxf(v.w * in_sc.w + in_sh.w, 0.f);
                }
                v.x = k ? v.x : 0.f; v.y = k ? v.y : 0.f; v.z = k ? v.z : 0.f; v.w = k ? v.w : 0.f;
                *reinterpret_cast<f32x4*>(sH + (idx >> 2) * PST + (idx & 3) * 4) = v;
            }
        }
    };
    // Weights: same packed U image as the LDS-DMA form ([chunk][ntile][4 steps][4 freq][64 co][16 ci], 16-byte chunks XOR-swizzled),
    // but every lane fetches its own two 16-byte operand pieces per frequency straight into registers (L2-resident, shared by
    // all workgroups): no LDS space, no LDS reads and no DMA drain in front of the barriers.
    const long u_chunk_stride = (long)a.grid_y * 16 * (64 * 16);
    const int t_half = tid >> 7, t_tile = (tid >> 2) & 31, t_q = tid & 3;
    const int t_ty = t_tile >> 3, t_tx = t_tile & 7;
    const int v_chunk = (t_q ^ ((t_tile >> 2) & 3)) * 4;
    // Input transform of one frequency row pair: part 0 -> rows {0, 2} (frequencies 0-3 / 8-11, used by steps 0-1),
    // part 1 -> rows {1, 3} (frequencies 4-7 / 12-15, used by steps 2-3).  Thread half h2 owns rows {2*h2, 2*h2+1}.
    // Row ii of a thread half is p + sg*q of two patch rows (B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]):
    //   half 0: row 0 = d0 - d2, row 1 = d1 + d2;   half 1: row 2 = d2 - d1, row 3 = d1 - d3.
    // The half is wave-uniform, so (p row, q row, sign) are scalars: 8 loads + 16 FMAs per part, no select of two variants.
    const int uhalf = __builtin_amdgcn_readfirstlane(t_half);
    const int prow0 = uhalf ? 2 : 0, qrow0 = uhalf ? 1 : 2;      // part 0 (ii = 0): sign -1 for both halves
    const int prow1 = 1, qrow1 = uhalf ? 3 : 2;                  // part 1 (ii = 1)
    const float sg1 = uhalf ? -1.0f : 1.0f;
    auto transform_part = [&](int ii, int buf) {
        const float* src = sH + ((2 * t_ty) * 18 + 2 * t_tx) * PST + t_q * 4;
        const float* ps = src + (ii == 0 ? prow0 : prow1) * 18 * PST;
        const float* qs = src + (ii == 0 ? qrow0 : qrow1) * 18 * PST;
        const float sg = ii == 0 ? -1.0f : sg1;
        f32x4 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 pv = *reinterpret_cast<const f32x4*>(ps + j * PST);
            f32x4 qv = *reinterpret_cast<const f32x4*>(qs + j * PST);
            x[j] = pv + sg * qv;
        }
        f32x4 v0 = x[0] - x[2], v1 = x[1] + x[2], v2 = x[2] - x[1], v3 = x[1] - x[3];
        float* dst = sV + buf * S_SV + (((2 * t_half + ii) * 4) * 32 + t_tile) * 16 + v_chunk;
        *reinterpret_cast<f32x4*>(dst + 0 * 32 * 16) = v0;
        *reinterpret_cast<f32x4*>(dst + 1 * 32 * 16) = v1;
        *reinterpret_cast<f32x4*>(dst + 2 * 32 * 16) = v2;
        *reinterpret_cast<f32x4*>(dst + 3 * 32 * 16) = v3;
    };

    // epilogue scale/shift are fetched here, long before they are needed: loaded in the epilogue (under the cout mask) the compiler's
    // waitcnt bookkeeping could not prove them landed at the joins of the masked store blocks and put `s_waitcnt vmcnt(0)` in front
    // of every one of a lane's 32 global stores, i.e. each store waited for the previous one to retire
    const int co = co0 + ng * 32 + li;
    const bool cvalid = co < a.Cout;
    const float sc = P.scale[min(co, a.Cout - 1)];
    const float sh = P.shift[min(co, a.Cout - 1)];

    f32x16 acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    const int sw = (li >> 2) & 3;
    const int c0 = ((2 * hh) ^ sw) * 4, c1 = ((2 * hh + 1) ^ sw) * 4;
    const float* Abase = sV + ((fh * 8) * 32 + li) * 16;
    // register-weight layout R: [chunk][ntile][step 4][fh 2][ng 2][fl 2][piece 2][lane 64][4 floats] — every operand load of a wave is
    // one contiguous KiB (lane = 32*hh + li holds channels 8*hh + 4*piece .. +3 of cout ng*32 + li)
    const float* u_lane = a.w + (long)by * 16 * (64 * 16) + (fh * 2 + ng) * 1024 + lane * 4;
    f32x4 bq[4][2][2];                       // [register buffer: step % 4][fl][operand piece]
    auto load_B = [&](int step, int buf) {   // step = chunk*4 + group: frequencies {2g, 2g+1} of this wave's half
        const float* s = u_lane + (step >> 2) * u_chunk_stride + (step & 3) * (4 * 64 * 16);
#pragma unroll
        for (int fl = 0; fl < 2; ++fl) {
            bq[buf][fl][0] = *reinterpret_cast<const f32x4*>(s + (fl * 2 + 0) * 256);
            bq[buf][fl][1] = *reinterpret_cast<const f32x4*>(s + (fl * 2 + 1) * 256);
        }
    };

    // Schedule: TWO barriers per chunk, 32 MFMAs per wave between them; no barrier waits for a weight load.
    //   steps 0-1 of chunk c: MFMAs on V(c) (buffer c&1) + the input transform of chunk c+1 from the halo into the other V buffer
    //   barrier (everyone is done reading the halo)
    //   steps 2-3: MFMAs + the halo of chunk c+2 registers -> LDS (its global loads were issued in step 0)
    //   barrier (V(c+1) and the new halo are visible; V(c) may be overwritten)
    const int total_steps = nchunks * 4;
    // prologue: the halos of chunks 0 AND 1 and the first weight pieces are requested together, so only one memory round trip is
    // exposed before the first MFMA (chunk 1's halo waits in registers until chunk 0's has been transformed)
    load_H(0);
    f32x4 h_next[S_H_ITERS], sc_next = in_sc, sh_next = in_sh;
    {
        const int c1 = min(1, nchunks - 1);
#pragma unroll
        for (int it = 0; it < S_H_ITERS; ++it) h_next[it] = *reinterpret_cast<const f32x4*>(xin + g_off[it] + c1 * 16);
        if (has_aff) {
            sc_next = *reinterpret_cast<const f32x4*>(aff_s + c1 * 16);
            sh_next = *reinterpret_cast<const f32x4*>(aff_b + c1 * 16);
        }
    }
    // weight operands are fetched PF steps ahead into four register buffers (index = step % 4 = g; a buffer is live for PF steps);
    // the first PF pieces are requested here, with the halos
    constexpr int PF = AFF ? 2 : 3;
#pragma unroll
    for (int t = 0; t < PF; ++t) load_B(min(t, total_steps - 1), t);
    store_H();
    __syncthreads();
    transform_part(0, 0);
    transform_part(1, 0);
#pragma unroll
    for (int it = 0; it < S_H_ITERS; ++it) h_stage[it] = h_next[it];
    in_sc = sc_next; in_sh = sh_next;
    __syncthreads();
    store_H();
    for (int c = 0; c < nchunks; ++c) {
        const int cnn = min(c + 2, nchunks - 1);
        const float* A = Abase + (c & 1) * S_SV;
        const int nbuf = (c + 1) & 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int step = c * 4 + g;
            const int sb = g, sb2 = (g + PF) & 3;
            if (g == 0 || g == 2) __syncthreads();
            if (g == 0) load_H(cnn);
            load_B(min(step + PF, total_steps - 1), sb2);
            f32x4 a0[2], a1[2];
#pragma unroll
            for (int fl = 0; fl < 2; ++fl) {
                const int al = g * 2 + fl;
                a0[fl] = *reinterpret_cast<const f32x4*>(A + al * 32 * 16 + c0);
                a1[fl] = *reinterpret_cast<const f32x4*>(A + al * 32 * 16 + c1);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int fl = 0; fl < 2; ++fl)
                    acc[g * 2 + fl] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[fl][s], bq[sb][fl][0][s], acc[g * 2 + fl], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int fl = 0; fl < 2; ++fl)
                    acc[g * 2 + fl] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[fl][s], bq[sb][fl][1][s], acc[g * 2 + fl], 0, 0, 0);
            if (g == 0) transform_part(0, nbuf);
            if (g == 1) transform_part(1, nbuf);
            if (g == 2) store_H();
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            if (g == 0 || g == 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            } else if (g == 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
        }
    }

    __syncthreads();
    float2* ex = reinterpret_cast<float2*>(sV);     // [wave 4][r 16][64 lanes] x (dx 0,1) = 32 KiB
    float keep[16][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float s0[2], s1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float m0 = acc[i * 4 + 0][r], m1 = acc[i * 4 + 1][r], m2 = acc[i * 4 + 2][r], m3 = acc[i * 4 + 3][r];
            s0[i] = m0 + m1 + m2;
            s1[i] = m1 - m2 - m3;
        }
        float2 send;
        if (fh == 0) {
            keep[r][0] = s0[0] + s0[1]; keep[r][1] = s1[0] + s1[1];
            send = make_float2(s0[1], s1[1]);
        } else {
            keep[r][0] = -s0[0] - s0[1]; keep[r][1] = -s1[0] - s1[1];
            send = make_float2(s0[0], s1[0]);
        }
        ex[(wave * 16 + r) * 64 + lane] = send;
    }
    __syncthreads();
    const int partner = wave ^ 2;
    const bool do_relu = co < a.relu_upto;
    // accumulator row r of lane half hh is tile (ty, tx) = (r >> 2, (r & 3) + 4*hh): the row offset of a store is uniform per r,
    // only the 8*hh column shift and the channel are per lane -> one lane base pointer, scalar offsets
    const int ow_l = ow0 + 8 * hh;
    float* ybase = P.y + (((long)n * H + oh0 + fh) * W + ow_l) * a.y_cs + a.y_co + co;
    float gs = 0.f, gss = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ty = r >> 2, txr = r & 3;
        const float2 other = ex[(partner * 16 + r) * 64 + lane];
        const bool row_ok = cvalid && (oh0 + 2 * ty + fh < H);
        float* yp = ybase + ((long)(2 * ty) * W + 2 * txr) * a.y_cs;
        float v0 = (keep[r][0] + other.x) * sc + sh, v1 = (keep[r][1] + other.y) * sc + sh;
        if (do_relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
        const bool ok0 = row_ok && ow_l + 2 * txr < W, ok1 = row_ok && ow_l + 2 * txr + 1 < W;
        if (ok0) { yp[0] = v0; gs += v0; gss = fmaf(v0, v0, gss); }
        if (ok1) { yp[a.y_cs] = v1; gs += v1; gss = fmaf(v1, v1, gss); }
    }
    // fused GroupNorm statistics of the NEXT layer's normalisation (fcos.py:182-186): fold the lane's 32 outputs over the
    // channels of its group (adjacent lanes) and the two column halves, one {sum, sumsq} record per (tile, row parity, group)
    if (a.gn_ws) {
        for (int o = 1; o < a.gn_cpg; o <<= 1) { gs += __shfl_xor(gs, o); gss += __shfl_xor(gss, o); }
        gs += __shfl_xor(gs, 32);
        gss += __shfl_xor(gss, 32);
        if (cvalid && hh == 0 && (li & (a.gn_cpg - 1)) == 0) {
            double* o = a.gn_ws + (((long)bx * 2 + fh) * a.gn_groups + co / a.gn_cpg) * 2;
            o[0] = (double)gs;
            o[1] = (double)gss;
        }
    }
}

static int launch_wino(ConvArgs& a, hipStream_t st) {
    static DeviceOnce once;
    int rc0 = once.run([]() {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino4r_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino4r_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_BYTES);
        return e == hipSuccess ? CMK_OK : fail(CMK_ELAUNCH, "conv_wino: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    });
    if (rc0) return rc0;
    int blocks = 0;
    for (int i = 0; i < a.nprob; ++i) {
        ConvProblem& p = a.p[i];
        p.tile_begin = blocks;
        p.tiles_h = cdiv(p.Ho, 8);
        p.tiles_w = cdiv(p.Wo, 16);
        blocks += p.N * p.tiles_h * p.tiles_w;
    }
    a.grid_y = cdiv(a.Cout, 64);
    a.total_tiles = blocks;
    const dim3 grid(((blocks + 7) / 8) * 8 * a.grid_y);
    if (a.p[0].in_scale)          // all problems of a launch agree on this (validated)
        hipLaunchKernelGGL(conv_wino4r_kernel<true>, grid, dim3(256), R_LDS_BYTES, st, a);
    else
        hipLaunchKernelGGL(conv_wino4r_kernel<false>, grid, dim3(256), R_LDS_BYTES, st, a);
    return check_launch("conv_wino");
}

// ---------------------------------------------------------------------------------------------------------------
// host side: variant menu + cost model
// ---------------------------------------------------------------------------------------------------------------
struct Variant { int wm, sc, wn; };

template <int TAPS, int STRIDE, int WM, int WN, int SC, bool GA = false>
static int launch(ConvArgs& a, int grid_y, hipStream_t st) {
    using G = Geo<TAPS, STRIDE, WM, WN, SC>;
    static DeviceOnce once;      // one per template instantiation
    auto kern = conv_igemm_kernel<TAPS, STRIDE, WM, WN, SC, GA>;
    int rc0 = once.run([kern]() {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        return e == hipSuccess ? CMK_OK : fail(CMK_ELAUNCH, "conv: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    });
    if (rc0) return rc0;
    int blocks = 0;
    for (int i = 0; i < a.nprob; ++i) {
        ConvProblem& p = a.p[i];
        p.tile_begin = blocks;
        if (TAPS == 9) {
            p.tiles_h = cdiv(p.Ho, G::TH);
            p.tiles_w = cdiv(p.Wo, G::TW);
            blocks += p.N * p.tiles_h * p.tiles_w;
        } else {
            p.tiles_h = p.tiles_w = 0;
            blocks += (int)((p.total_pix + G::BM - 1) / G::BM);
        }
    }
    a.grid_y = grid_y;
    a.total_tiles = blocks;
    if (a.ksplit < 1) a.ksplit = 1;
    const int vchunks = (GA ? 9 : 1) * (a.Cin >> 4);
    if (a.ksplit > 1 && (a.nprob != 1 || a.res_mode == 2 || !a.ws || vchunks % (2 * a.ksplit)))
        return fail(CMK_EINVAL, "conv: split-K needs one problem, a workspace, no upsampled residual and K chunks %% (2*splitk) == 0%s", "");
    hipLaunchKernelGGL(kern, dim3(((blocks + 7) / 8) * 8 * grid_y, a.ksplit), dim3(256), G::LDS_BYTES, st, a);
    int rc = check_launch("conv_igemm");
    if (rc || a.ksplit == 1) return rc;
    const ConvProblem& p = a.p[0];
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long>((p.total_pix * (a.cout_pad >> 2) + 255) / 256, 256L * 32)), dim3(256), 0, st, a.ws, a.ksplit, p.total_pix,
                       a.cout_pad, p.scale, p.shift, a.Cout, a.relu_upto, a.res_mode == 1 ? a.res : nullptr, a.res_cs, a.res_co, p.y, a.y_cs,
                       a.y_co);
    return check_launch("splitk_reduce");
}

// cost of running `blocks` equal workgroups with `resident` per CU on 256 CUs: full rounds keep every CU at `resident`
// workgroups; the last round spreads round-robin.  eff(j) = matrix-pipe utilisation with j workgroups on a CU.
static double round_cost(long blocks, int resident, double wg_cost) {
    static const double eff[4] = {1.0, 0.70, 0.90, 0.95};
    const long slots = 256L * resident;
    long full = blocks / slots, rem = blocks % slots;
    double c = (double)full * resident * wg_cost / eff[resident];
    if (rem) {
        int j = (int)((rem + 255) / 256);
        c += (double)j * wg_cost / eff[j];
    }
    return c;
}

static bool variant_ok(int taps, int stride, int cout32, int wm, int sc, int wn) {
    if (wn < 1 || wn > 7 || (wm != 1 && wm != 2) || (sc != 16 && sc != 32)) return false;
    if (cout32 <= 7 ? (cout32 % wn != 0) : (wn != 1 && wn != 2 && wn != 4)) return false;   // packed cout_pad must be a multiple of 32*wn
    if (wm == 2 && (wn > 4 || stride == 2)) return false;
    if (taps == 1 && sc != 32) return false;
    if (stride == 2 && sc != 16) return false;
    return true;
}

// Default choice when the caller gives no tuned variant: minimise modelled time over the menu.
static Variant choose_variant(const ConvArgs& a, int taps, int stride, int cout32) {
    Variant best{1, taps == 1 ? 32 : 16, cout32 <= 7 ? cout32 : 4};
    double best_cost = 1e300;
    const int cout_pad32 = cout32 <= 7 ? cout32 : cdiv(cout32, 4) * 4;
    for (int wn = 7; wn >= 1; --wn)
        for (int wm = 2; wm >= 1; --wm)
            for (int sc = 16; sc <= 32; sc += 16) {
                if (!variant_ok(taps, stride, cout32, wm, sc, wn)) continue;
                const int sr = 32 / sc, bm = 128 * wm;
                const int th = taps == 9 ? sr * 4 * wm : 1, tw = taps == 9 ? sc : bm;
                long blocks = 0;
                for (int i = 0; i < a.nprob; ++i)
                    blocks += taps == 9 ? (long)a.p[i].N * cdiv(a.p[i].Ho, th) * cdiv(a.p[i].Wo, tw) : (a.p[i].total_pix + bm - 1) / bm;
                blocks *= cout_pad32 / wn;
                const int apix = taps == 9 ? ((th - 1) * stride + 3) * ((tw - 1) * stride + 3) : bm;
                const int abytes = apix * PST * 4, bbytes = 32 * wn * PST * 4;
                const int occ = occ_of(wm, wn, stride);
                const bool adb = (2 * abytes + 2 * bbytes) * occ <= LDS_CU;
                const int lds = (adb ? 2 : 1) * abytes + 2 * bbytes;
                const int resident = lds * occ <= LDS_CU ? occ : (lds * 2 <= LDS_CU ? 2 : 1);
                // per-workgroup time ~ MFMA cycles per step + a fixed per-step overhead (barrier, LDS fill, address math)
                const double wg_cost = 512.0 * wm * wn + 260.0;
                double cost = round_cost(blocks, resident, wg_cost);
                if (cost < best_cost) { best_cost = cost; best = Variant{wm, sc, wn}; }
            }
    return best;
}

template <int TAPS, int STRIDE, int WN>
static int dispatch_variant(ConvArgs& a, int grid_y, Variant v, hipStream_t st) {
    if constexpr (TAPS == 1) {
        if constexpr (WN <= 4) { if (v.wm == 2) return launch<1, 1, 2, WN, 32>(a, grid_y, st); }
        return launch<1, 1, 1, WN, 32>(a, grid_y, st);
    } else if constexpr (STRIDE == 2) {
        return launch<9, 2, 1, WN, 16>(a, grid_y, st);
    } else {
        if constexpr (WN <= 4) {
            if (v.wm == 2) return v.sc == 16 ? launch<9, 1, 2, WN, 16>(a, grid_y, st) : launch<9, 1, 2, WN, 32>(a, grid_y, st);
        }
        return v.sc == 16 ? launch<9, 1, 1, WN, 16>(a, grid_y, st) : launch<9, 1, 1, WN, 32>(a, grid_y, st);
    }
}

template <int TAPS, int STRIDE>
static int dispatch_wn(ConvArgs& a, int cout32, Variant v, hipStream_t st) {
    const int cout_pad32 = cout32 <= 7 ? cout32 : cdiv(cout32, 4) * 4;
    a.cout_pad = cout_pad32 * 32;
    const int grid_y = cout_pad32 / v.wn;
    switch (v.wn) {
        case 1: return dispatch_variant<TAPS, STRIDE, 1>(a, grid_y, v, st);
        case 2: return dispatch_variant<TAPS, STRIDE, 2>(a, grid_y, v, st);
        case 3: return dispatch_variant<TAPS, STRIDE, 3>(a, grid_y, v, st);
        case 4: return dispatch_variant<TAPS, STRIDE, 4>(a, grid_y, v, st);
        case 5: return dispatch_variant<TAPS, STRIDE, 5>(a, grid_y, v, st);
        case 6: return dispatch_variant<TAPS, STRIDE, 6>(a, grid_y, v, st);
        case 7: return dispatch_variant<TAPS, STRIDE, 7>(a, grid_y, v, st);
    }
    return fail(CMK_EINVAL, "conv: bad WN%s", "");
}

static int validate(const cmk_conv_desc* d) {
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
    if ((d->in_scale == nullptr) != (d->in_shift == nullptr)) return fail(CMK_EINVAL, "conv: in_scale and in_shift come together%s", "");
    if (d->in_scale && d->ksize == 1 && ((long)d->H * d->W) % 256) return fail(CMK_EINVAL, "conv: input affine on a 1x1 conv needs H*W %% 256 == 0%s", "");
    if (d->pool_ws && d->ksize != 1) return fail(CMK_EINVAL, "conv: pooled sums are for 1x1 convs%s", "");
    return CMK_OK;
}

static void fill_problem(ConvProblem& p, const cmk_conv_desc* d) {
    p.x = d->x; p.y = d->y; p.scale = d->scale; p.shift = d->shift;
    p.in_scale = d->in_scale; p.in_shift = d->in_shift;
    p.w = d->w_wino6;
    p.N = d->N; p.H = d->H; p.W = d->W;
    p.Ho = d->stride == 1 ? d->H : (d->H - 1) / 2 + 1;  // k3 p1 s2: floor((H+2-3)/2)+1
    p.Wo = d->stride == 1 ? d->W : (d->W - 1) / 2 + 1;
    p.tiles_h = p.tiles_w = p.tile_begin = 0;
    p.total_pix = (long)p.N * p.Ho * p.Wo;
}

static int setup_gn(ConvArgs& a, const cmk_conv_desc* d) {
    const int cpg = d->gn_groups > 0 ? d->Cout / d->gn_groups : 0;
    if (d->relu_upto != 0 || d->gn_groups < 1 || d->Cout % d->gn_groups || cpg > 32 || (cpg & (cpg - 1)))
        return fail(CMK_EINVAL, "conv: fused GroupNorm statistics need relu_upto == 0 and a power-of-two group width <= 32%s", "");
    a.gn_ws = d->gn_ws; a.gn_cpg = cpg; a.gn_groups = d->gn_groups;
    return CMK_OK;
}

// The tile height (4 | 2) with which descriptor d runs on the pointwise GEMM kernel (conv_pw.hip), 0 if it does not: tune_wm 8 as given, or
// the untuned default — 1x1 convs with enough pixels and output channels to fill the chip (measured 1.12-1.2x conv_igemm on every concat /
// lateral / deconv shape of the model, tools/bench_pw.py), 256-pixel workgroups from 2 rounds on.
static int pointwise_mt(const cmk_conv_desc* d, int n) {
    const int cout32 = (d->Cout + 31) / 32;
    const long total_pix = (long)d->N * d->H * d->W;
    if (d->ksize != 1 || n != 1 || cout32 <= 7 || (d->Cin & 31) || d->in_scale || d->in_relu || d->gn_ws || total_pix * d->x_cs * 4 >= (1L << 31))
        return 0;
    if (d->splitk > 1 && (d->tune_wm != 8 || !d->splitk_ws || d->res_mode == 2 || d->pool_ws || (d->Cin >> 4) % (2 * d->splitk))) return 0;
    if (d->res_mode == 2 && ((d->W & 1) || d->pool_ws || (long)d->N * d->Hr * d->Wr * d->res_cs * 4 >= (1L << 31))) return 0;     // FPN top-down add: even widths
    if (d->tune_wm == 8) return (d->tune_wn == 4 || d->tune_wn == 2) ? d->tune_wn : 0;
    if (d->tune_wm == 10) return (d->w_split && d->tune_wn == 4 && d->res_mode != 1 && d->splitk <= 1) ? 4 : 0;      // the bf16-split form: the 256-pixel tile
    if (d->tune_wm == 12) return (d->w_splith && d->tune_wn == 4 && d->res_mode != 1 && d->splitk <= 1) ? 4 : 0;     // the fp16-split form
    if (d->tune_wm || d->tune_sc || d->tune_wn) return 0;
    const long ctiles = cdiv(cout32, 4);
    const long wg2 = ((total_pix + 127) / 128) * ctiles, wg4 = ((total_pix + 255) / 256) * ctiles;
    return wg2 >= 256 ? (wg4 >= 1024 ? 4 : 2) : 0;
}

// The same for the gather form of a 3x3 conv on that kernel (tune_wm 9, or the untuned default for stride-2 convs of at least 1024
// 256-pixel workgroups: stem_3).
static int gather_mt(const cmk_conv_desc* d, int n) {
    const int cout32 = (d->Cout + 31) / 32;
    const long in_pix = (long)d->N * d->H * d->W;
    const long out_pix = (long)d->N * (d->stride == 1 ? d->H : (d->H - 1) / 2 + 1) * (d->stride == 1 ? d->W : (d->W - 1) / 2 + 1);
    if (d->ksize != 3 || n != 1 || (cout32 != 4 && cout32 <= 7) || (d->Cin & 31) || d->in_scale || d->in_relu || d->res_mode == 2 ||
        d->gn_ws || d->pool_ws || in_pix * d->x_cs * 4 >= (1L << 31) || d->H >= 32768 || d->W >= 32768)
        return 0;
    if (d->splitk > 1 && (d->tune_wm != 9 || !d->splitk_ws || (9 * (d->Cin >> 4)) % (2 * d->splitk))) return 0;
    if (d->tune_wm == 9) return (d->tune_wn == 4 || d->tune_wn == 2) ? d->tune_wn : 0;
    if (d->tune_wm == 10) return (d->w_split && d->tune_wn == 4 && d->res_mode == 0 && d->splitk <= 1) ? 4 : 0;      // the bf16-split gather form
    if (d->tune_wm == 12) return (d->w_splith && d->tune_wn == 4 && d->res_mode == 0 && d->splitk <= 1) ? 4 : 0;     // the fp16-split gather form
    if (d->tune_wm || d->tune_sc || d->tune_wn || d->stride != 2) return 0;
    const long ctiles = cdiv(cout32, 4);
    const long wg4 = ((out_pix + 255) / 256) * ctiles;
    return wg4 >= 1024 ? 4 : 0;          // measured (tools/bench_ga.py): stem_3 1.28x conv_igemm; the 14 -> 7 maskiou conv and P6/P7 stay on its split-K gather form
}

// the pointwise kernel's launch, followed by the split-K reduction when it left partial sums
static int run_pointwise(ConvArgs& a, int mt, hipStream_t st) {
    int rc = launch_pw(a, mt, st);
    if (rc || a.ksplit <= 1) return rc;
    const ConvProblem& p = a.p[0];
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long>((p.total_pix * (a.cout_pad >> 2) + 255) / 256, 256L * 32)), dim3(256), 0, st, a.ws, a.ksplit, p.total_pix,
                       a.cout_pad, p.scale, p.shift, a.Cout, a.relu_upto, a.res_mode == 1 ? a.res : nullptr, a.res_cs, a.res_co, p.y, a.y_cs,
                       a.y_co);
    return check_launch("splitk_reduce");
}

static int run(const cmk_conv_desc* descs, int n, void* stream) {
    const cmk_conv_desc* d = &descs[0];
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.nprob = n;
    for (int i = 0; i < n; ++i) fill_problem(a.p[i], &descs[i]);
    a.w = d->w; a.res = d->res;
    a.Cin = d->Cin; a.Cout = d->Cout;
    a.x_cs = d->x_cs; a.x_co = d->x_co; a.y_cs = d->y_cs; a.y_co = d->y_co;
    a.res_cs = d->res_cs; a.res_co = d->res_co; a.res_mode = d->res_mode; a.Hr = d->Hr; a.Wr = d->Wr;
    if (a.res_mode == 2 && (a.Hr * 2 < a.p[0].Ho || a.Wr * 2 < a.p[0].Wo)) return fail(CMK_EINVAL, "conv: upsampled residual too small%s", "");
    a.relu_upto = d->relu_upto; a.in_relu = d->in_relu;
    const int cout32 = (d->Cout + 31) / 32;
    const int taps = d->ksize * d->ksize;
    hipStream_t st = (hipStream_t)stream;
    if (d->tune_wm == 5) {          // Winograd F(2x2,3x3): 3x3 stride 1, no residual / input ReLU
        if (d->ksize != 3 || d->stride != 1 || d->res_mode != 0 || d->in_relu || !d->w_wino)
            return fail(CMK_EINVAL, "conv: Winograd variant not available for this conv%s", "");
        if (d->splitk > 1) return fail(CMK_EINVAL, "conv: split-K is a direct-kernel feature%s", "");
        if (d->gn_ws) {
            int rc = setup_gn(a, d);
            if (rc) return rc;
        }
        a.w = d->w_wino;
        return launch_wino(a, st);
    }
    if (d->tune_wm == 6) {          // Winograd F(4x4,3x3): same conditions, its own packed weights
        if (d->ksize != 3 || d->stride != 1 || d->res_mode != 0 || d->in_relu || !d->w_wino6 || (d->Cin & 7))
            return fail(CMK_EINVAL, "conv: Winograd F(4x4,3x3) variant not available for this conv%s", "");
        if (d->gn_ws) {
            int rc = setup_gn(a, d);
            if (rc) return rc;
        }
        a.w = d->w_wino6;
        a.ws = d->splitk_ws;          // split-K slabs (instrumented W6_TRACE builds: a stamp buffer)
        a.ksplit = d->splitk > 1 ? d->splitk : 1;
        a.cout_pad = cmk_conv_cout_pad(d->Cout);
        if (a.ksplit > 1) {           // F(4x4) with split-K (32-cout form, map tiles): partial sums + the reduce kernel of the direct path
            if (d->tune_sc == 64 || d->tune_wn != 1 || n != 1) return fail(CMK_EINVAL, "conv: Winograd split-K needs tune_sc 16, tune_wn 1, one problem%s", "");
            int rc = launch_wino6(a, 0, st);
            if (rc) return rc;
            const ConvProblem& p = a.p[0];
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long>((p.total_pix * (a.cout_pad >> 2) + 255) / 256, 256L * 32)), dim3(256), 0, st, a.ws, a.ksplit,
                               p.total_pix, a.cout_pad, p.scale, p.shift, a.Cout, a.relu_upto, (const float*)nullptr, 0, 0, p.y, a.y_cs, a.y_co);
            return check_launch("splitk_reduce");
        }
        if (d->tune_wn != 1 && d->tune_wn != 2) return fail(CMK_EINVAL, "conv: tune_wm 6 takes tune_wn 1 (12x40 map tiles) or 2 (pairs of RoI maps up to 16x14)%s", "");
        if (d->tune_sc == 64) return launch_wino6s(a, d->tune_wn == 2 ? 1 : 0, st);      // 64 couts per workgroup, shared frequency image
        return launch_wino6(a, d->tune_wn == 2 ? 1 : 0, st);
    }
    a.ksplit = d->splitk > 1 ? d->splitk : 1;
    a.ws = d->splitk_ws;
    if (d->pool_ws && !pointwise_mt(d, n)) return fail(CMK_EINVAL, "conv: pooled sums are produced by the pointwise GEMM kernel only (cmk_conv_pool_rows)%s", "");
    a.pool_ws = d->pool_ws;
    if (d->tune_wm == 8) {                             // pointwise GEMM kernel (conv_pw.hip); tune_wn = accumulator rows per wave
        if (d->ksize != 1 || cout32 <= 7) return fail(CMK_EINVAL, "conv: pointwise variant needs a 1x1 conv with Cout > 224%s", "");
        a.cout_pad = cdiv(cout32, 4) * 128;
        a.gn_ws = d->gn_ws;
        if (a.ksplit > 1 && !pointwise_mt(d, n)) return fail(CMK_EINVAL, "conv: pointwise variant: split-K not available for this conv%s", "");
        return run_pointwise(a, d->tune_wn, st);
    }
    if (d->tune_wm == 10) {                            // opt-in: the pointwise GEMM from bf16-split products (fp32-accurate, cmk.h w_split)
        if (!(d->ksize == 1 ? pointwise_mt(d, n) : gather_mt(d, n)))
            return fail(CMK_EINVAL, "conv: the bf16-split variant needs w_split and a conv the pointwise GEMM kernel takes (1x1, or 3x3 in its gather form)%s", "");
        a.cout_pad = cdiv(cout32, 4) * 128;
        a.w = reinterpret_cast<const float*>(d->w_split);
        a.ksplit = 1;
        a.ga_stride = d->ksize == 3 ? d->stride : 0;
        return launch_pw_split(a, 1, st);
    }
    if (d->tune_wm == 12) {                            // opt-in: the same on two fp16 pieces per operand / three products (cmk.h w_splith)
        if (!(d->ksize == 1 ? pointwise_mt(d, n) : gather_mt(d, n)) || !(d->w_splith_scale > 0.f))
            return fail(CMK_EINVAL, "conv: the fp16-split variant needs w_splith, w_splith_scale and a conv the pointwise GEMM kernel takes (1x1, or 3x3 in its gather form)%s", "");
        a.cout_pad = cdiv(cout32, 4) * 128;
        a.w = reinterpret_cast<const float*>(d->w_splith);
        a.p[0].acc_scale = d->w_splith_scale;
        a.ksplit = 1;
        a.ga_stride = d->ksize == 3 ? d->stride : 0;
        return launch_pw_split(a, 2, st);
    }
    if (d->tune_wm == 11) {                            // opt-in: direct 3x3 conv on bf16-split products (conv_sp3.hip); tune_sc = pieces, tune_wn = geometry
        if (d->ksize != 3 || d->stride != 1 || !d->w_splith || d->splitk > 1 || d->res_mode != 0 || d->in_relu || d->pool_ws)
            return fail(CMK_EINVAL, "conv: the direct fp16-split variant needs w_splith and a plain 3x3 stride-1 conv%s", "");
        if (d->gn_ws) {
            int rc = setup_gn(a, d);
            if (rc) return rc;
        }
        for (int i = 0; i < n; ++i) {
            if (!descs[i].w_splith || !(descs[i].w_splith_scale > 0.f)) return fail(CMK_EINVAL, "conv: w_splith / w_splith_scale missing%s", "");
            a.p[i].w = reinterpret_cast<const float*>(descs[i].w_splith);
            a.p[i].acc_scale = descs[i].w_splith_scale;
        }
        a.cout_pad = cdiv(cout32, 4) * 128;
        a.ksplit = 1;
        return launch_sp3(a, d->tune_wn, d->tune_sc, st);
    }
    if (d->tune_wm == 9) {                             // gather form of a 3x3 conv on the pointwise GEMM kernel; tune_wn = accumulator rows per wave
        const int mt = gather_mt(d, n);
        if (!mt) return fail(CMK_EINVAL, "conv: pointwise gather variant not available for this conv%s", "");
        a.cout_pad = cdiv(cout32, 4) * 128;
        a.ga_stride = d->stride;
        return run_pointwise(a, mt, st);
    }
    if (d->tune_wm == 7) {                             // gather form: 3x3 (stride 1|2) as a flattened-pixel GEMM over 9x the K chunks
        if (d->ksize != 3 || n != 1 || d->res_mode == 2 || d->in_scale || (d->tune_wn != 1 && d->tune_wn != 2 && d->tune_wn != 4))
            return fail(CMK_EINVAL, "conv: gather variant not available for this conv%s", "");
        const int cout_pad32 = cout32 <= 7 ? cout32 : cdiv(cout32, 4) * 4;
        if (cout_pad32 % d->tune_wn) return fail(CMK_EINVAL, "conv: gather variant: Cout tiles %% WN != 0%s", "");
        a.cout_pad = cout_pad32 * 32;
        a.ga_stride = d->stride;
        const int gy = cout_pad32 / d->tune_wn;
        return d->tune_wn == 4 ? launch<1, 1, 1, 4, 32, true>(a, gy, st) : d->tune_wn == 2 ? launch<1, 1, 1, 2, 32, true>(a, gy, st)
                                                                                           : launch<1, 1, 1, 1, 32, true>(a, gy, st);
    }
    Variant v;
    if (d->tune_wm || d->tune_sc || d->tune_wn) {      // the caller measured and picked a variant
        if (d->gn_ws) return fail(CMK_EINVAL, "conv: fused GroupNorm statistics are only produced by the Winograd form (tune_wm 5)%s", "");
        v = Variant{d->tune_wm, d->tune_sc, d->tune_wn};
        if (!variant_ok(taps, d->stride, cout32, v.wm, v.sc, v.wn)) return fail(CMK_EINVAL, "conv: variant not available for this shape%s", "");
    } else {
        // untuned default: the 2-WG/CU Winograd form wins on every 3x3 stride-1 shape measured (tools/bench_wino.py), so take it
        // whenever the caller packed the transformed weights; otherwise the direct-kernel cost model decides
        // F(4x4,3x3) where its 12x40 tiles are reasonably full and there are enough of them (measured on the model's maps, tools/bench_wino6.py:
        // 1.1-1.5x the 2x2 form down to 25x40 maps, 0.4x on 14x14 RoI maps whose tiles are 20 % full)
        if (d->ksize == 3 && d->stride == 1 && d->res_mode == 0 && !d->in_relu && d->w_wino6 && d->Cin >= 32 && !(d->Cin & 7) && d->splitk <= 1) {
            double px = 0.0, covered = 0.0;
            long wgs = 0;
            for (int i = 0; i < n; ++i) {
                const long t = (long)descs[i].N * cdiv(descs[i].H, 12) * cdiv(descs[i].W, 40);
                px += (double)descs[i].N * descs[i].H * descs[i].W;
                covered += (double)t * 480.0;
                wgs += t * cdiv(d->Cout, 32);
            }
            // maps of at most 16 x 14 (the 14x14 RoI features): two whole maps per workgroup instead of 12x40 tiles that would be 20 % full
            if (n == 1 && d->H <= 16 && d->W <= 14 && !d->gn_ws && (long)cdiv(d->N, 2) * cdiv(d->Cout, 32) >= 256) {
                a.w = d->w_wino6;
                return launch_wino6(a, 1, st);
            }
            if (px >= 0.55 * covered && wgs >= 256) {
                if (d->gn_ws) {
                    int rc = setup_gn(a, d);
                    if (rc) return rc;
                }
                a.w = d->w_wino6;
                return launch_wino6(a, 0, st);
            }
        }
        if (d->ksize == 3 && d->stride == 1 && d->res_mode == 0 && !d->in_relu && d->w_wino && d->Cin >= 32 && d->splitk <= 1) {
            if (d->gn_ws) {
                int rc = setup_gn(a, d);
                if (rc) return rc;
            }
            a.w = d->w_wino;
            return launch_wino(a, st);
        }
        if (d->gn_ws) return fail(CMK_EINVAL, "conv: fused GroupNorm statistics are only produced by the Winograd form%s", "");
        // stride-2 3x3 on a map of at most 16x16 outputs (maskiou conv4 14->7, P6/P7): the spatial tiles would be mostly empty
        if (d->ksize == 3 && d->stride == 2 && n == 1 && d->res_mode != 2 && !d->in_scale && a.p[0].Ho <= 16 && a.p[0].Wo <= 16) {
            const int cout_pad32 = cout32 <= 7 ? cout32 : cdiv(cout32, 4) * 4;
            const int wn = (cout_pad32 % 4 == 0 && a.p[0].total_pix >= 8192) ? 4 : (cout_pad32 % 2 == 0 && a.p[0].total_pix >= 2048) ? 2 : 1;
            a.cout_pad = cout_pad32 * 32;
            a.ga_stride = 2;
            const int gy = cout_pad32 / wn;
            return wn == 4 ? launch<1, 1, 1, 4, 32, true>(a, gy, st) : wn == 2 ? launch<1, 1, 1, 2, 32, true>(a, gy, st)
                                                                               : launch<1, 1, 1, 1, 32, true>(a, gy, st);
        }
        if (const int mt = pointwise_mt(d, n)) {
            a.cout_pad = cdiv(cout32, 4) * 128;
            return launch_pw(a, mt, st);
        }
        if (const int mt = gather_mt(d, n)) {
            a.cout_pad = cdiv(cout32, 4) * 128;
            a.ga_stride = d->stride;
            return launch_pw(a, mt, st);
        }
        v = choose_variant(a, taps, d->stride, cout32);
    }
    if (d->ksize == 1) return dispatch_wn<1, 1>(a, cout32, v, st);
    if (d->stride == 1) return dispatch_wn<9, 1>(a, cout32, v, st);
    return dispatch_wn<9, 2>(a, cout32, v, st);
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

extern "C" int cmk_conv_gn_tiles(int H, int W) { return ((H + 7) / 8) * ((W + 15) / 16); }

// {sum, sumsq} records per image that a conv with fused GroupNorm statistics writes: tune_wm 5 -> 2 per 8x16 tile, 6 -> 4 per 12x40 tile,
// 110 + g (tune_wm 11, geometry g: conv_sp3.hip) -> 2 per 8x32 (g 0) | 1 per 4x32 (1) | 2 per 16x16 (2) | 1 per 8x16 (3) tile
extern "C" int cmk_conv_gn_records(int H, int W, int tune_wm) {
    if (tune_wm >= 110 && tune_wm <= 113) {
        const int g = tune_wm - 110;
        const int th = g == 0 ? 8 : g == 1 ? 4 : g == 2 ? 16 : 8, tw = g < 2 ? 32 : 16;
        return ((g & 1) ? 1 : 2) * ((H + th - 1) / th) * ((W + tw - 1) / tw);
    }
    return tune_wm == 6 ? 4 * ((H + 11) / 12) * ((W + 39) / 40) : 2 * ((H + 7) / 8) * ((W + 15) / 16);
}

extern "C" int64_t cmk_splith_packed_halves(int Cout, int Cin) {     // per tap: two fp16 pieces per weight (cmk.h w_splith)
    return (int64_t)((Cin + 15) / 16) * (((Cout + 127) / 128) * 4) * 2 * 64 * 8;
}

extern "C" int64_t cmk_split_packed_halves(int Cout, int Cin) {      // per tap of the conv: a 3x3 conv in the gather form holds nine of these, tap-major
    return (int64_t)((Cin + 15) / 16) * (((Cout + 127) / 128) * 4) * 3 * 64 * 8;
}

extern "C" int64_t cmk_wino_packed_floats(int Cout, int Cin) {
    return (int64_t)((Cin + 15) / 16) * ((Cout + 63) / 64) * 16 * 64 * 16;
}

extern "C" int cmk_conv_pool_rows(const cmk_conv_desc* d) {
    if (!d) return 0;
    const int mt = cmk::pointwise_mt(d, 1);
    return (mt && (long)d->H * d->W >= 32 * mt) ? 32 * mt : 0;
}

extern "C" int cmk_conv2d_nhwc(const cmk_conv_desc* d, void* stream) {
    int rc = cmk::validate(d);
    if (rc) return rc;
    return cmk::run(d, 1, stream);
}

extern "C" int cmk_conv2d_nhwc_multi(const cmk_conv_desc* descs, int n, void* stream) {
    using namespace cmk;
    if (!descs || n < 1 || n > MAXP) return fail(CMK_EINVAL, "conv_multi: need 1..%s%ld problems", "", MAXP);
    bool same_w = true;
    for (int i = 0; i < n; ++i) {
        int rc = validate(&descs[i]);
        if (rc) return rc;
        const cmk_conv_desc *a = &descs[0], *b = &descs[i];
        same_w = same_w && b->w == a->w && b->w_wino == a->w_wino && b->w_wino6 == a->w_wino6 && b->w_split == a->w_split && b->w_splith == a->w_splith;
        if (b->Cin != a->Cin || b->Cout != a->Cout || b->ksize != a->ksize || b->stride != a->stride ||
            b->relu_upto != a->relu_upto || b->in_relu != a->in_relu || b->x_cs != a->x_cs || b->x_co != a->x_co || b->y_cs != a->y_cs ||
            b->y_co != a->y_co || b->res_mode != 0 || b->tune_wm != a->tune_wm || b->tune_sc != a->tune_sc || b->tune_wn != a->tune_wn || (b->in_scale == nullptr) != (a->in_scale == nullptr) ||
            b->gn_ws != a->gn_ws || b->gn_groups != a->gn_groups || b->splitk > 1 || b->pool_ws)
            return fail(CMK_EINVAL, "conv_multi: problems must share channels/views/flags and carry no residual%s", "");
    }
    // problems with different weights (the cls and the bbox tower of the FCOS head, fcos.py:227-231, in one launch) and more than 5 problems:
    // the F(4x4) kernels only, which take the packed weights per problem
    if ((!same_w || n > 5) && (descs[0].tune_wm != 6 || descs[0].tune_wn != 1) && descs[0].tune_wm != 11)
        return fail(CMK_EINVAL, "conv_multi: different weights per problem / more than 5 problems need tune_wm 6, tune_wn 1 (the F(4x4) map kernels)%s", "");
    for (int i = 0; i < n; ++i)
        if (descs[0].tune_wm == 6 && !descs[i].w_wino6) return fail(CMK_EINVAL, "conv_multi: w_wino6 missing%s", "");
    return run(descs, n, stream);
}
