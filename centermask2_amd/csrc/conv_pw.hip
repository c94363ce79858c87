// Pointwise (1x1) convolution of the big aggregation layers as a plain fp32 GEMM on the matrix pipe, built the way the F(4x4) kernel
// taught (conv_wino6.hip): what limits these kernels on MI355X is the energy spent moving operands, not issue slots.
//
//   GEMM view: M = flattened pixels (N*H*W), N = output channels, K = Cin; v_mfma_f32_32x32x2_f32, pixels on the accumulator rows,
//   output channels on the lanes (128-byte NHWC stores).
//   Workgroup = 4 waves as 2 (pixels) x 2 (couts); wave tile = MT x 2 accumulators of 32 px x 32 couts (MT = 4: 128 px x 64 couts,
//   workgroup 256 px x 128 couts; MT = 2 for layers with too few pixels to fill the chip with the large tile); two workgroups per CU.
//   Operand traffic per MFMA: 1 activation register from LDS per 2 MFMAs (each ds_read_b128 feeds 4 k-steps x 2 cout tiles), one
//   weight register per MT MFMAs straight from L2/L1 into registers (no LDS slab, no barrier dependence, fetched two chunks ahead through
//   a wave-uniform scalar base) — 0.75 operand registers per MFMA where the 32 px x 128 couts wave tile of conv_igemm took 1.25, and half
//   its global->LDS staging bytes.  The activations are staged global -> registers -> LDS once per workgroup and 16-channel chunk with
//   raw buffer loads (rows past the last pixel come back as zeros from the hardware range check: no selects), double-buffered, one
//   barrier per chunk = per 64 (MT = 4) MFMAs of a wave, placed BETWEEN the two halves of a chunk with the LDS operand reads half a
//   chunk ahead of their MFMAs (see "pipeline" below).
//   The weights are the layout conv_igemm already packs ([chunk][cout_pad][16 channels]); K order and accumulation order are the
//   same as conv_igemm's, so the two kernels agree bit for bit.
//   Everything outside the MFMAs is priced in VALU instructions: fp32 MFMA and VALU do not co-execute, and a VALU instruction issued
//   next to the other workgroup's MFMA stream waits for a gap in it — the epilogue is 1.5 VALU per stored value (see there).
//   POOL variant: the per-cout sums of the stored values over the wave's rows (the eSE average pool, vovnet.py:255-256) leave with the
//   epilogue — a lane is a cout, so it is one packed add per two values and no pass over the map.
//
// Reference call sites replaced: the OSA concat 1x1 convs (vovnet.py:222-236), an FPN lateral, the mask head's deconv as a 1x1.
#include <math.h>

#include <type_traits>

#include "conv_args.hpp"

namespace cmk {

typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 pw_buffer_load(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

// POOL: also leave the per-block sums of the stored values in a.pool_ws (the eSE average pool of the aggregation conv, cmk.h)
// GA ("gather"): a 3x3 conv (stride a.ga_stride = 1 | 2, padding 1) as the same GEMM over flattened OUTPUT pixels whose K walks 9 taps x Cin/16
// chunks: the staged row of a thread is gathered per tap straight from the image (an offset outside the resource where the tap falls
// outside it: zeros).  No halo reuse — each input pixel is read 9/stride^2 times from L2 — so it is for the convs Winograd does not take:
// stride 2 with enough pixels to fill the chip (stem_3 vovnet.py:412: 1.28x conv_igemm; the small stride-2 convs stay on its split-K
// gather form).  Weights: conv_igemm's [tap][chunk][cout_pad][16]; same tap-major K order as its gather form (bit-identical to it).
// UPRES: the FPN top-down add — the residual is the nearest-neighbour 2x upsampling of a map of half the size (a.res, a.Hr x a.Wr); needs an
// even output width (then the pixel pairs the epilogue handles share one residual pixel).
// SPLITK: blockIdx.y owns an (even) range of the K chunks and leaves raw partial sums in a.ws[z][pixel][cout_pad]; the caller runs
// conv_igemm's reduce kernel (scale/shift/residual/ReLU there).  For the skinny GEMMs: MaskIoU fc1 (400 x 12544 x 1024), the 14 -> 7 conv.
// SPLIT (opt-in, cmk.h tune_wm 10; MT 4, no GA / UPRES / SPLITK): the same GEMM with fp32-ACCURATE products built from bf16 pieces on
// v_mfma_f32_32x32x16_bf16 (2.0 PFLOP/s against the fp32 instruction's 0.157): every fp32 operand is split into three bf16 values
// (x = hi + mid + lo exactly to 2^-24: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), round to nearest even) and the six products of
// weight >= 2^-16 — lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi, small terms first — are accumulated in fp32.  Measured error against float64:
// that of a sequential fp32 fma chain (tools/probe/gemm_split_bf16.hip, tools/split_bf16_numerics.py).  The activations stay fp32 in HBM
// and are split while they are staged (global -> registers -> split -> LDS as [32-row block][piece][lane][8 bf16], the MFMA's A operand
// layout: one ds_read_b128 per piece and block); the weights are split once on the host (a.w: [K/16][cout_pad/32][piece 3][lane 64][8 bf16]).
// The accumulator layout is the fp32 kernel's, so the epilogue (scale/shift/ReLU, stores, POOL sums) is shared.
// SPLIT 2 (opt-in, cmk.h tune_wm 12): the same GEMM on TWO FP16 pieces per operand — 22 bits of significand, three products m*h, h*m, h*h (what
// is dropped is 2^-22 of a product), half the MFMAs of the bf16 form for the same fp32-class error; fp16's exponent range is met by exact
// power-of-two scaling as in conv_sp3.hip (activations x 2^-4, their residual x 2^11 and the weight piece it meets x 2^-11 in registers; the
// weights x S_w on the host, a.p[0].acc_scale = 1 / S_w; the accumulator x 2^4 / S_w folded into the epilogue's per-channel scale).
// a.w: [K/16][cout_pad/32][piece 2][lane 64][8 fp16] (cmk.h w_splith).
template <int MT, bool POOL, bool GA, bool UPRES, bool SPLITK, int SPLIT = 0>
__global__ __launch_bounds__(256, 2) void conv_pw_kernel(const ConvArgs a) {
    static_assert(!SPLIT || (MT == 4 && !SPLITK && !(GA && (UPRES || POOL))), "the split forms: the 256 x 128 tile, no split-K");
    constexpr int BM = 64 * MT;         // pixels per workgroup
    constexpr int ABUF = BM * PST;      // floats per LDS buffer (rows of 16 channels, pitch 20)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, li = lane & 31;
    const int wm = wave & 1, wn = wave >> 1;

    // XCD-aware order: the cout tiles of one pixel tile run back to back on one XCD (they share the activations in its L2)
    const int xq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int bx = (xq / a.grid_y) * 8 + xcd, by = xq % a.grid_y;
    if (bx >= a.total_tiles) return;
    const ConvProblem& P = a.p[0];
    const long total_pix = P.total_pix;
    const long pix0 = (long)bx * BM;
    const int co0 = by * 128 + wn * 64;
    const int cin_chunks = a.Cin >> 4;
    const int nchunks_all = GA ? 9 * cin_chunks : cin_chunks;     // even (host)
    const int c_lo = SPLITK ? (int)((long)blockIdx.y * nchunks_all / a.ksplit) : 0;            // even bounds (host)
    const int nchunks = SPLITK ? (int)((long)(blockIdx.y + 1) * nchunks_all / a.ksplit) : nchunks_all;      // one past this slice's last chunk

    // ---- activations: global -> registers -> LDS ------------------------------------------------------------------------------------
    // thread = (row tid>>2 (+64 per iteration), channel quad tid&3); a row past the last pixel gets an offset outside the resource
    i32x4 rsrc;
    {
        const unsigned long long base = (unsigned long long)P.x;
        rsrc.x = __builtin_amdgcn_readfirstlane((int)(base & 0xffffffffull));
        rsrc.y = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffffull));
        rsrc.z = __builtin_amdgcn_readfirstlane((int)((GA ? (long)P.N * P.H * P.W : total_pix) * a.x_cs * 4));      // < 2^31 (host)
        rsrc.w = 0x00020000;
    }
    int a_voff[MT];
    int ga_base[GA ? MT : 1], ga_hw[GA ? MT : 1];       // GA: byte offset of the window's top-left sample (may be negative), (ih0 << 16) | (iw0 & 0xffff)
#pragma unroll
    for (int it = 0; it < MT; ++it) {
        const long px = pix0 + it * 64 + (tid >> 2);
        if (GA) {
            const bool pv = px < total_pix;
            const long pp = pv ? px : 0;
            const long hw = (long)P.Ho * P.Wo;
            const int n_ = (int)(pp / hw);
            const int rem = (int)(pp - (long)n_ * hw);
            const int oh = rem / P.Wo, ow = rem - oh * P.Wo;
            const int ih0 = pv ? oh * a.ga_stride - 1 : -4, iw0 = ow * a.ga_stride - 1;      // -4: every tap lands outside
            ga_base[it] = (((n_ * P.H + ih0) * P.W + iw0) * a.x_cs + a.x_co + (tid & 3) * 4) * 4;
            ga_hw[it] = (ih0 << 16) | (iw0 & 0xffff);
            a_voff[it] = (int)0x80000000;
        } else {
            a_voff[it] = px < total_pix ? (int)((px * a.x_cs + a.x_co + (tid & 3) * 4) * 4) : (int)0x80000000;
        }
    }
    const int a_dst = (tid >> 2) * PST + (tid & 3) * 4;
    f32x4 a_st[MT];
    int ga_tap = -1;                    // GA: the tap a_voff[] is set up for (the requests walk the chunks in order)
    auto load_A = [&](int chunk) {
        int soff = chunk * 64;
        if (GA) {
            const int tap = chunk / cin_chunks;             // wave-uniform
            soff = (chunk - tap * cin_chunks) * 64;
            if (tap != ga_tap) {
                ga_tap = tap;
                const int kh = tap / 3, kw = tap - kh * 3;
                const int delta = (kh * P.W + kw) * a.x_cs * 4;
#pragma unroll
                for (int it = 0; it < MT; ++it) {
                    const int ih = (ga_hw[it] >> 16) + kh, iw = (int)(short)(ga_hw[it] & 0xffff) + kw;
                    a_voff[it] = (ih >= 0 && ih < P.H && iw >= 0 && iw < P.W) ? ga_base[it] + delta : (int)0x80000000;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < MT; ++it) a_st[it] = pw_buffer_load(rsrc, a_voff[it], soff, 0);
    };
    auto store_A = [&](int buf) {
        float* dst = smem + buf * ABUF + a_dst;
#pragma unroll
        for (int it = 0; it < MT; ++it) *reinterpret_cast<f32x4*>(dst + it * 64 * PST) = a_st[it];
    };

#ifdef PW_TRACE
    // instrumented build (tools/ab/trace_pw.py): lane 0 of every wave of every 16th workgroup stamps the shader clock into LDS (a global
    // store inside the loop would make every barrier drain the loads in flight) and copies the stamps to a.ws at the end:
    // [0] start, [1] prologue done, [2+2c] barrier of chunk c reached, [3+2c] passed (c < 28), [58] loop done, [59] stores issued,
    // [60] XCC id, [61] HW id, [62]/[63] real time at the end/start
    const bool tracing = a.ws && (blockIdx.x % 16) == 0 && lane == 0;
    unsigned long long* trl = reinterpret_cast<unsigned long long*>(smem + 2 * ABUF) + wave * 64;
    unsigned long long* trc = reinterpret_cast<unsigned long long*>(a.ws) + ((blockIdx.x / 16) * 4 + wave) * 64;
#define PW_STAMP(slot) do { if (tracing) trl[slot] = __builtin_readcyclecounter(); } while (0)
    if (tracing) {
        trl[63] = __builtin_amdgcn_s_memrealtime();
        trl[61] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));
        trl[60] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((32 - 1) << 11));
    }
    PW_STAMP(0);
#else
#define PW_STAMP(slot) do { } while (0)
#endif
    // PW_ABL (timing ablations, results wrong): 1 no barrier, 2 no activation loads, 4 no weight loads, 8 no LDS writes
#ifndef PW_ABL
#define PW_ABL 0
#endif
#if PW_ABL & 1
#define PW_SYNC
#else
#define PW_SYNC __syncthreads()
#endif
#if PW_ABL & 2
#define PW_LOAD_A(c)
#else
#define PW_LOAD_A(c) load_A(c)
#endif
#if PW_ABL & 4
#define PW_LOAD_B(c, s)
#else
#define PW_LOAD_B(c, s) load_B(c, s)
#endif
#if PW_ABL & 8
#define PW_STORE_A(b)
#else
#define PW_STORE_A(b) store_A(b)
#endif
    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;
    if constexpr (SPLIT == 2 && !GA) {
        // 1x1 conv on two fp16 pieces, 32 channels per stage: a thread requests 16 bytes of a 128-byte line whose other seven sixteenths are requested by
        // its neighbours in the same instruction (the 16-channel form takes half of every line now and the other half one chunk later: measured, the
        // activation requests alone cost that form 44 % of its time, profiles/r03_conv_sp3.txt section 8), and a barrier spans 48 MFMAs of a wave
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        constexpr float SX = 0.0625f, RS = 2048.f;              // activation scale 2^-4, residual scale 2^11
        constexpr int NIT = BM / 32;                            // 8 requests per thread and stage: row tid / 8 + 32 it, channels 4 (tid & 7) .. + 3 of the 32
        constexpr int STAGE = (BM / 32) * 2 * 2 * 64 * 16;      // bytes per LDS stage: [row block][k16 half][piece][lane][8 fp16]
        unsigned char* sb = reinterpret_cast<unsigned char*>(smem);
        auto pk = [](float x, float y) { return __builtin_bit_cast(unsigned, f16x2{(_Float16)x, (_Float16)y}); };       // round to nearest even
        auto unpk = [](unsigned p_) { const f16x2 h_ = __builtin_bit_cast(f16x2, p_); return f32x2{(float)h_.x, (float)h_.y}; };
        const int oct = tid & 7;
        int x_voff[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const long px = pix0 + it * 32 + (tid >> 3);
            x_voff[it] = px < total_pix ? (int)((px * a.x_cs + a.x_co + oct * 4) * 4) : (int)0x80000000;
        }
        // row r = tid / 8 + 32 it is row (tid >> 3) of row block it; channels 4 oct .. : k16 half oct >> 2, lane half (oct & 3) >> 1, 8-byte half slot oct & 1
        const int st_off = (((oct >> 2) * 2) * 64 + ((oct & 3) >> 1) * 32 + (tid >> 3)) * 16 + (oct & 1) * 8;
        f32x4 xs[NIT];
        auto load_X = [&](int c32) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) xs[it] = pw_buffer_load(rsrc, x_voff[it], c32 * 128, 0);
        };
        auto stage = [&](int buf) {
            unsigned char* dst = sb + buf * STAGE + st_off;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const f32x4 x = xs[it] * SX;
                u32x2 h, m_;
                h.x = pk(x.x, x.y); h.y = pk(x.z, x.w);
                const f32x2 h01 = unpk(h.x), h23 = unpk(h.y);
                const f32x4 r1 = f32x4{x.x - h01.x, x.y - h01.y, x.z - h23.x, x.w - h23.y} * RS;      // exact: h holds the leading bits of x
                m_.x = pk(r1.x, r1.y); m_.y = pk(r1.z, r1.w);
                *reinterpret_cast<u32x2*>(dst + it * (4 * 64 * 16)) = h;
                *reinterpret_cast<u32x2*>(dst + it * (4 * 64 * 16) + 64 * 16) = m_;
            }
        };
        const u32x4* wsp = reinterpret_cast<const u32x4*>(a.w) + ((long)(co0 >> 5) * 2) * 64 + lane;      // wave-uniform base + lane
        const long wstep = (long)(a.cout_pad >> 5) * 2 * 64;                                               // per 16 channels
        u32x4 wb[2][2][2];                  // [k16 half][cout tile][piece]
        auto load_Bs = [&](int s_, int k16) {
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int p_ = 0; p_ < 2; ++p_) wb[s_][nn][p_] = wsp[k16 * wstep + (nn * 2 + p_) * 64];
        };
        const int n32 = a.Cin >> 5;
        load_X(0);
        load_Bs(0, 0);
        load_Bs(1, 1);
        stage(0);
        load_X(min(1, n32 - 1));
        for (int c = 0; c < n32; ++c) {
            __syncthreads();            // stage c & 1 is complete; everybody has read all of the other stage
            const u32x4* ap = reinterpret_cast<const u32x4*>(sb + (c & 1) * STAGE) + (wm * MT * 4) * 64 + lane;
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                f16x8 Bhs[2];
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) Bhs[nn] = __builtin_bit_cast(f16x8, wb[s_][nn][0]) * (_Float16)(1.f / RS);      // meets the activations' scaled residual
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const u32x4 ah = ap[(m * 4 + s_ * 2 + 0) * 64], am = ap[(m * 4 + s_ * 2 + 1) * 64];
                    if (s_ == 0 && m == 2) {        // the next stage's activations (in registers since the last stage) are split between the MFMA groups
                        stage((c + 1) & 1);
                        load_X(min(c + 2, n32 - 1));
                    }
                    const f16x8 Ah = __builtin_bit_cast(f16x8, ah), Am = __builtin_bit_cast(f16x8, am);
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn) {
                        f32x16 cacc = acc[m][nn];
                        cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Am, Bhs[nn], cacc, 0, 0, 0);
                        cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8, wb[s_][nn][1]), cacc, 0, 0, 0);
                        cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8, wb[s_][nn][0]), cacc, 0, 0, 0);
                        acc[m][nn] = cacc;
                    }
                }
                load_Bs(s_, min(2 * (c + 1) + s_, 2 * n32 - 1));     // this half of the next stage's weights: in flight during the other half's MFMAs
            }
        }
        __syncthreads();                // the epilogue may reuse the LDS
    } else if constexpr (SPLIT == 2) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        constexpr float SX = 0.0625f, RS = 2048.f;              // activation scale 2^-4, residual scale 2^11
        constexpr int STAGE = (BM / 32) * 2 * 64 * 16;          // bytes per LDS stage: [row block][piece][lane][8 fp16]
        unsigned char* sb = reinterpret_cast<unsigned char*>(smem);
        auto pk = [](float x, float y) { return __builtin_bit_cast(unsigned, f16x2{(_Float16)x, (_Float16)y}); };       // round to nearest even
        auto unpk = [](unsigned p_) { const f16x2 h_ = __builtin_bit_cast(f16x2, p_); return f32x2{(float)h_.x, (float)h_.y}; };
        int st_off[MT];
#pragma unroll
        for (int it = 0; it < MT; ++it) {
            const int r = (tid >> 2) + 64 * it;
            st_off[it] = (((r >> 5) * 2) * 64 + ((tid & 3) >> 1) * 32 + (r & 31)) * 16 + (tid & 1) * 8;
        }
        auto stage = [&](int buf) {
            unsigned char* dst = sb + buf * STAGE;
#pragma unroll
            for (int it = 0; it < MT; ++it) {
                const f32x4 x = a_st[it] * SX;
                u32x2 h, m_;
                h.x = pk(x.x, x.y); h.y = pk(x.z, x.w);
                const f32x2 h01 = unpk(h.x), h23 = unpk(h.y);
                const f32x4 r1 = f32x4{x.x - h01.x, x.y - h01.y, x.z - h23.x, x.w - h23.y} * RS;      // exact: h holds the leading bits of x
                m_.x = pk(r1.x, r1.y); m_.y = pk(r1.z, r1.w);
                *reinterpret_cast<u32x2*>(dst + st_off[it]) = h;
                *reinterpret_cast<u32x2*>(dst + st_off[it] + 64 * 16) = m_;
            }
        };
        const u32x4* wsp = reinterpret_cast<const u32x4*>(a.w) + ((long)(co0 >> 5) * 2) * 64 + lane;      // wave-uniform base + lane
        const long wstep = (long)(a.cout_pad >> 5) * 2 * 64;
        u32x4 wb[2][2];
        auto load_Bs = [&](int chunk) {
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int p_ = 0; p_ < 2; ++p_) wb[nn][p_] = wsp[chunk * wstep + (nn * 2 + p_) * 64];
        };
        load_A(0);
        load_Bs(0);
        stage(0);
        load_A(min(1, nchunks - 1));
        for (int c = 0; c < nchunks; ++c) {
            __syncthreads();            // stage c & 1 is complete; everybody has read all of the other stage
            const u32x4* ap = reinterpret_cast<const u32x4*>(sb + (c & 1) * STAGE) + (wm * MT * 2) * 64 + lane;
            f16x8 Bh[2], Bm[2], Bhs[2];
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) {
                Bh[nn] = __builtin_bit_cast(f16x8, wb[nn][0]);
                Bm[nn] = __builtin_bit_cast(f16x8, wb[nn][1]);
                Bhs[nn] = Bh[nn] * (_Float16)(1.f / RS);        // meets the activations' scaled residual
            }
            load_Bs(min(c + 1, nchunks - 1));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const u32x4 ah = ap[(m * 2 + 0) * 64], am = ap[(m * 2 + 1) * 64];
                if (m == 1) {           // the next chunk's activations (in registers since the last chunk) are split between the MFMA groups
                    stage((c + 1) & 1);
                    load_A(min(c + 2, nchunks - 1));
                }
                const f16x8 Ah = __builtin_bit_cast(f16x8, ah), Am = __builtin_bit_cast(f16x8, am);
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) {
                    f32x16 cacc = acc[m][nn];
                    cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Am, Bhs[nn], cacc, 0, 0, 0);
                    cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bm[nn], cacc, 0, 0, 0);
                    cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh[nn], cacc, 0, 0, 0);
                    acc[m][nn] = cacc;
                }
            }
        }
        __syncthreads();                // the epilogue may reuse the LDS
    } else if constexpr (SPLIT == 1) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        constexpr int STAGE = (BM / 32) * 3 * 64 * 16;          // bytes per LDS stage: [row block][piece][lane][8 bf16]
        unsigned char* sb = reinterpret_cast<unsigned char*>(smem);
        auto pk = [](float x, float y) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
        auto f_lo = [](unsigned p_) { return __builtin_bit_cast(float, p_ << 16); };
        auto f_hi = [](unsigned p_) { return __builtin_bit_cast(float, p_ & 0xffff0000u); };
        // this thread's 8-byte half slots: row r = tid / 4 + 64 it -> (block r / 32, li = r % 32); quad tid & 3 -> lane half (tid & 3) / 2, half slot (tid & 3) & 1
        int st_off[MT];
#pragma unroll
        for (int it = 0; it < MT; ++it) {
            const int r = (tid >> 2) + 64 * it;
            st_off[it] = (((r >> 5) * 3) * 64 + ((tid & 3) >> 1) * 32 + (r & 31)) * 16 + (tid & 1) * 8;
        }
        auto stage = [&](int buf) {
            unsigned char* dst = sb + buf * STAGE;
#pragma unroll
            for (int it = 0; it < MT; ++it) {
                const f32x4 x = a_st[it];
                u32x2 h, m_, l;
                h.x = pk(x.x, x.y); h.y = pk(x.z, x.w);
                const f32x4 r1 = {x.x - f_lo(h.x), x.y - f_hi(h.x), x.z - f_lo(h.y), x.w - f_hi(h.y)};
                m_.x = pk(r1.x, r1.y); m_.y = pk(r1.z, r1.w);
                const f32x4 r2 = {r1.x - f_lo(m_.x), r1.y - f_hi(m_.x), r1.z - f_lo(m_.y), r1.w - f_hi(m_.y)};
                *reinterpret_cast<u32x2*>(dst + st_off[it]) = h;
                *reinterpret_cast<u32x2*>(dst + st_off[it] + 64 * 16) = m_;
#if !(PW_ABL & 16)
                l.x = pk(r2.x, r2.y); l.y = pk(r2.z, r2.w);
                *reinterpret_cast<u32x2*>(dst + st_off[it] + 2 * 64 * 16) = l;
#endif
            }
        };
        const u32x4* wsp = reinterpret_cast<const u32x4*>(a.w) + ((long)(co0 >> 5) * 3) * 64 + lane;      // wave-uniform base + lane
        const long wstep = (long)(a.cout_pad >> 5) * 3 * 64;
        u32x4 wb[2][3];
        auto load_Bs = [&](int chunk) {
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int p_ = 0; p_ < ((PW_ABL & 16) ? 2 : 3); ++p_) wb[nn][p_] = wsp[chunk * wstep + (nn * 3 + p_) * 64];
        };
        // (GA: load_A walks the 9 taps x Cin / 16 chunks in order, re-pointing the rows per tap; the split weights are packed tap-major to match)
        load_A(0);
        load_Bs(0);
        stage(0);
        load_A(min(1, nchunks - 1));
        for (int c = 0; c < nchunks; ++c) {
            __syncthreads();            // stage c & 1 is complete; everybody has read all of the other stage
            const u32x4* ap = reinterpret_cast<const u32x4*>(sb + (c & 1) * STAGE) + (wm * MT * 3) * 64 + lane;
            u32x4 wc[2][3];
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int p_ = 0; p_ < 3; ++p_) wc[nn][p_] = wb[nn][p_];
            load_Bs(min(c + 1, nchunks - 1));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const u32x4 ah = ap[(m * 3 + 0) * 64], am = ap[(m * 3 + 1) * 64], al = (PW_ABL & 16) ? am : ap[(m * 3 + 2) * 64];
                if (m == 1) {           // the next chunk's activations (in registers since the last chunk) are split between the MFMA groups
                    stage((c + 1) & 1);
                    load_A(min(c + 2, nchunks - 1));
                }
                const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Am = __builtin_bit_cast(bf16x8, am), Al = __builtin_bit_cast(bf16x8, al);
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) {
                    const bf16x8 Bh = __builtin_bit_cast(bf16x8, wc[nn][0]), Bm = __builtin_bit_cast(bf16x8, wc[nn][1]), Bl = __builtin_bit_cast(bf16x8, wc[nn][2]);
                    f32x16 cacc = acc[m][nn];
#if !(PW_ABL & 16)
                    cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, cacc, 0, 0, 0);
                    cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, cacc, 0, 0, 0);
                    cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, cacc, 0, 0, 0);
#endif
                    cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, cacc, 0, 0, 0);
                    cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, cacc, 0, 0, 0);
                    cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, cacc, 0, 0, 0);
                    acc[m][nn] = cacc;
                }
            }
        }
        __syncthreads();                // the epilogue may reuse the LDS
    } else {
    // ---- weights: L2 -> registers; lane (li, hh) takes channels 8*hh .. 8*hh+7 of output channel (tile base + li) ---------------------
    const float* wbase[2];
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) wbase[nn] = a.w + (long)min(co0 + nn * 32, a.cout_pad - 32) * 16;     // wave-uniform
    const long w_chunk = (long)a.cout_pad * 16;
    const unsigned w_lane = li * 16 + hh * 8;
    f32x4 bq[2][2][2];                  // [register set][cout tile][k-steps 0..3 | 4..7]
    auto load_B = [&](int chunk, int set) {
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
            const float* src = wbase[nn] + chunk * w_chunk;
            bq[set][nn][0] = *reinterpret_cast<const f32x4*>(src + w_lane);
            bq[set][nn][1] = *reinterpret_cast<const f32x4*>(src + (w_lane + 4));
        }
    };

    // ---- MFMA side --------------------------------------------------------------------------------------------------------------------
    int a_off[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a_off[m] = ((wm * MT + m) * 32 + li) * PST + hh * 8;
    f32x4 avA[MT], avB[MT];             // operands of the first / second half (4 k-steps each) of a chunk
    auto rd = [&](f32x4 (&v)[MT], int buf, int h) {
        const float* A = smem + buf * ABUF + 4 * h;
#pragma unroll
        for (int m = 0; m < MT; ++m) v[m] = *reinterpret_cast<const f32x4*>(A + a_off[m]);
    };
    // 4 k-steps: channels 4h..4h+3 (lane half 0) and 8+4h..8+4h+3 (lane half 1) of the chunk — conv_igemm's order
    auto mh = [&](const f32x4 (&v)[MT], int set, int h) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn)
                    acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[m][s], bq[set][nn][h][s], acc[m][nn], 0, 0, 0);
    };

    // ---- pipeline ------------------------------------------------------------------------------------------------------------------
    // chunk c, LDS buffer u = c & 1, weight set u:
    //   read the second-half operands of chunk c; 32 MFMAs of the first half, among them: A(c+1) (in registers since chunk c-1) into
    //   buffer u^1, request A(c+2)
    //   barrier: buffer u^1 is complete, everybody has read all of buffer u
    //   read the first-half operands of chunk c+1; 32 MFMAs of the second half; request B(c+2) into set u
    // No MFMA waits for an LDS round trip (each read has 32 MFMAs to land), the activations have a whole chunk to arrive, the weights
    // more.  Measured with the trace build: reading all of a chunk's operands behind its barrier cost a wave alone on its SIMD 700 of
    // 4800 cycles per chunk.
    // The fences pin the order of the prologue's requests: the compiler's wait counts at the loop head are the minimum over the prologue's
    // and the loop's order, so a weight request scheduled late there makes every chunk wait for younger loads than it needs.
    load_A(c_lo);
    __builtin_amdgcn_sched_barrier(0);
    load_B(c_lo, 0);
    __builtin_amdgcn_sched_barrier(0);
    store_A(0);
    load_A(c_lo + 1);
    __builtin_amdgcn_sched_barrier(0);
    load_B(c_lo + 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    PW_STAMP(1);
    PW_SYNC;
    rd(avA, 0, 0);
    for (int c = c_lo; c < nchunks; c += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            rd(avB, u, 1);
            mh(avA, u, 0);
            PW_STORE_A(u ^ 1);
            PW_LOAD_A(min(c + u + 2, nchunks - 1));
            __builtin_amdgcn_sched_barrier(0);      // (the compiler otherwise moves the second half's MFMAs in front of the barrier)
#ifdef PW_TRACE
            if (c + u < 28) PW_STAMP(2 + 2 * (c + u));
#endif
            PW_SYNC;
#ifdef PW_TRACE
            if (c + u < 28) PW_STAMP(3 + 2 * (c + u));
#endif
            rd(avA, u ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mh(avB, u, 1);
            PW_LOAD_B(min(c + u + 2, nchunks - 1), u);
        }
    }

    }       // fp32-MFMA operand path
    PW_STAMP(58);
    // ---- epilogue: scale/shift (+same-size residual) (+ReLU), NHWC stores ------------------------------------------------------------------------
    // accumulator register r of lane half hh is pixel row (r & 3) + 8 * (r >> 2) + 4 * hh of the 32-pixel sub-tile; the lane is the cout
    const long wpix0 = pix0 + wm * (MT * 32);
    if constexpr (SPLITK) {
        float* wz = a.ws + ((long)blockIdx.y * total_pix + wpix0) * a.cout_pad + co0 + li;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (wpix0 + row < total_pix) {
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn)
                        if (co0 + nn * 32 < a.cout_pad) wz[(long)row * a.cout_pad + nn * 32] = acc[m][nn][r];
                }
            }
        return;
    }
    float* ybase = P.y + wpix0 * a.y_cs;                                   // wave-uniform; rows are added to the lane offset below
    const float* rbase = a.res_mode == 1 ? a.res + wpix0 * a.res_cs : nullptr;
    const bool interior = (UPRES || a.res_mode == 0) && pix0 + BM <= total_pix && co0 + 64 <= a.Cout;
    if (UPRES) {
        // byte offset of every row's residual pixel, once per workgroup, into the (now free) LDS: the epilogue then needs no address arithmetic
        // beyond one add per pixel pair
        if (tid < BM) {
            const long p = min(pix0 + tid, total_pix - 1);
            const long hw = (long)P.Ho * P.Wo;
            const int n_ = (int)(p / hw);
            const int rem = (int)(p - (long)n_ * hw);
            const int oh = rem / P.Wo, ow = rem - oh * P.Wo;
            reinterpret_cast<int*>(smem)[tid] = (int)(((((long)n_ * a.Hr + (oh >> 1)) * a.Wr + (ow >> 1)) * a.res_cs + a.res_co) * 4);
        }
        __syncthreads();
    }
    // POOL: this wave's 32*MT rows are block g of the flattened pixels; rows from `bnd` on belong to the next image
    const long blk_g = (long)bx * 2 + wm;
    int bnd = MT * 32;
    if (POOL) {
        const long hw = (long)P.Ho * P.Wo, first = blk_g * (MT * 32);
        bnd = (int)min((long)(MT * 32), (first / hw + 1) * hw - first);
    }
    auto epilogue = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 pool2 = {0.f, 0.f};             // sums of this lane's stored values: rows before / from `bnd` (INTERIOR without a boundary: .x and .y are two halves of one sum)
            float poolA = 0.f, poolB = 0.f;
            const int co = co0 + nn * 32 + li;
            const bool cvalid = INTERIOR || co < a.Cout;
            float sc = cvalid ? P.scale[co] : 0.f;
            if constexpr (SPLIT == 2) sc *= P.acc_scale * 16.f;         // 1 / (S_x * S_w)
            float sh = cvalid ? P.shift[co] : 0.f;
            const float lo = co < a.relu_upto ? 0.f : __builtin_nanf("");      // max(v, NaN) = v: lanes without the ReLU
            // the values are waited for once, here; the compiler cannot see through the asm and so does not put a full wait in front
            // of every store (conv_wino6.hip)
            asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(sc), "+v"(sh));
            if constexpr (INTERIOR) {
                // Every VALU instruction of a wave in its epilogue waits for a gap in the MFMA stream of the other workgroup on the SIMD
                // (and takes the slot from it): 1.5 per stored value — one packed fma per two rows, one max each; the address is a
                // wave-uniform row pointer walked by the scalar unit plus a fixed lane offset (global_store saddr form, written as
                // asm because the compiler renders the same C as 64-bit vector adds: 5 VALU per value, trace build: 37k-cycle epilogues).
                const f32x2 sc2 = {sc, sc}, sh2 = {sh, sh};
                const unsigned voff = (unsigned)(4 * hh * a.y_cs + a.y_co + co) * 4u;
                const unsigned long long yb = (unsigned long long)ybase;
                unsigned long long rowp = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(yb >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)yb);
                const unsigned long long row1 = (unsigned long long)a.y_cs * 4u, row5 = row1 * 5u;
                float rv[UPRES ? MT : 1][8];
                if constexpr (UPRES) {
                    // one residual value per pixel pair (rows r, r+1 with r even: the same coarse pixel), all 8*MT of them requested before the
                    // first store of this cout tile and waited for ONCE: a wait in front of each use would also wait for the stores in between
                    const int* tab = reinterpret_cast<const int*>(smem) + wm * (MT * 32) + 4 * hh;
                    const char* rb = reinterpret_cast<const char*>(a.res);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            rv[m][j] = *reinterpret_cast<const float*>(rb + (unsigned)(tab[m * 32 + ((2 * j) & 3) + 8 * (j >> 1)] + co * 4));
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        asm volatile("" : "+v"(rv[m][0]), "+v"(rv[m][1]), "+v"(rv[m][2]), "+v"(rv[m][3]), "+v"(rv[m][4]), "+v"(rv[m][5]), "+v"(rv[m][6]), "+v"(rv[m][7]));
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2 v = __builtin_elementwise_fma(f32x2{acc[m][nn][r], acc[m][nn][r + 1]}, sc2, sh2);
                        if constexpr (UPRES) v += f32x2{rv[m][r >> 1], rv[m][r >> 1]};
                        const float v0 = fmaxf(v.x, lo), v1 = fmaxf(v.y, lo);
                        if (POOL) pool2 += f32x2{v0, v1};
                        // ("+s": the pointer is walked between the stores, not computed 128 times up front and spilled)
                        asm volatile("global_store_dword %1, %2, %0" : "+s"(rowp) : "v"(voff), "v"(v0) : "memory");
                        rowp += row1;
                        asm volatile("global_store_dword %1, %2, %0" : "+s"(rowp) : "v"(voff), "v"(v1) : "memory");
                        rowp += (r & 3) == 2 ? row5 : row1;      // rows 0..3, 8..11, 16..19, 24..27 (+4 for lane half 1), next sub-tile at 32
                    }
                if (POOL) {
                    poolA = pool2.x + pool2.y;
                    if (bnd < MT * 32) {            // an image ends inside this block (wave-uniform, rare): split the sum by row
                        asm volatile("" ::: "memory");
                        poolA = 0.f;
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = fmaxf(acc[m][nn][r] * sc + sh, lo);
                                if (m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh < bnd) poolA += v; else poolB += v;
                            }
                    }
                }
            } else {
                unsigned off = (unsigned)(4 * hh * a.y_cs + a.y_co + co);          // walks the rows 0..3, 8..11, ... of each sub-tile: +1 +1 +1 +5
                unsigned roff = (unsigned)(4 * hh * a.res_cs + a.res_co + co);     // the same walk over the residual
                int rows_left = cvalid ? (int)min(total_pix - wpix0 - 4 * hh, 1L << 20) : 0;   // rows of this lane half that exist
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    // keep the walk a walk (else: 128 hoisted row offsets / predicates, spilled)
                    asm volatile("" : "+v"(off), "+v"(roff), "+v"(rows_left));
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[m][nn][r] * sc + sh;
                        const int step = (r & 3) == 3 ? 5 : 1;
                        if (m * 32 + (r & 3) + 8 * (r >> 2) < rows_left) {
                            if (a.res_mode == 1) v += rbase[roff];
                            if (UPRES)
                                v += *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.res) +
                                     (unsigned)(reinterpret_cast<const int*>(smem)[wm * (MT * 32) + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh] + co * 4));
                            v = fmaxf(v, lo);
                            ybase[off] = v;
                            if (POOL) { if (m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh < bnd) poolA += v; else poolB += v; }
                        }
                        off += step * a.y_cs;
                        roff += step * a.res_cs;
                    }
                }
            }
            if (POOL) {                                   // the two lane halves hold different rows of the same cout
                poolA += __shfl_xor(poolA, 32);
                poolB += __shfl_xor(poolB, 32);
                if (hh == 0 && co < a.Cout) {
                    a.pool_ws[(blk_g * 2) * a.Cout + co] = poolA;
                    a.pool_ws[(blk_g * 2 + 1) * a.Cout + co] = poolB;
                }
            }
        }
    };
    if (interior) epilogue(std::true_type{});
    else epilogue(std::false_type{});
#ifdef PW_TRACE
    PW_STAMP(59);
    if (tracing) {
        trl[62] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 64; ++i) trc[i] = trl[i];
    }
#endif
}

template <int MT, bool POOL, bool GA, bool UPRES = false, bool SPLITK = false, int SPLIT = 0>
static int launch_pw_mt(ConvArgs& a, hipStream_t st) {
    constexpr int BM = 64 * MT;
#ifdef PW_TRACE
#ifndef PW_LDS_EXTRA
#define PW_LDS_EXTRA 0
#endif
    constexpr int LDS_BYTES = 2 * BM * PST * 4 + 4 * 64 * 8 + PW_LDS_EXTRA;      // PW_LDS_EXTRA: experiments with one workgroup per CU
    static DeviceOnce once;
    int rc0 = once.run([]() {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pw_kernel<MT, POOL, GA, UPRES, SPLITK, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        return e == hipSuccess ? CMK_OK : fail(CMK_ELAUNCH, "conv_pw: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    });
    if (rc0) return rc0;
#else
    constexpr int LDS_BYTES = SPLIT == 2 && !GA ? 2 * (BM / 32) * 4 * 64 * 16 : SPLIT ? 2 * (BM / 32) * (SPLIT == 2 ? 2 : 3) * 64 * 16 : 2 * BM * PST * 4;       // 40 KB (MT 4) / 20 KB (MT 2) / 48 | 32 KB (split): under the 64 KB a kernel gets without an attribute
#endif
    ConvProblem& p = a.p[0];
    p.tile_begin = 0;
    p.tiles_h = p.tiles_w = 0;
    const long tiles = (p.total_pix + BM - 1) / BM;
    a.total_tiles = (int)tiles;
    a.grid_y = a.cout_pad / 128;
    hipLaunchKernelGGL((conv_pw_kernel<MT, POOL, GA, UPRES, SPLITK, SPLIT>), dim3((unsigned)(((tiles + 7) / 8) * 8 * a.grid_y), SPLITK ? a.ksplit : 1), dim3(256), LDS_BYTES, st, a);
    return check_launch("conv_pw");
}

// mt = 4 | 2.  The caller (conv_igemm.hip: run) has filled the problem, views, epilogue options and cout_pad; a.ga_stride = 1 | 2 asks for
// the gather form of a 3x3 conv (a.w then is conv_igemm's 9-tap packing), 0 for a 1x1 conv.
int launch_pw(ConvArgs& a, int mt, hipStream_t st) {
    const ConvProblem& p = a.p[0];
    if (a.nprob != 1 || p.in_scale || a.in_relu || a.gn_ws)
        return fail(CMK_EINVAL, "conv_pw: one problem, no input affine / input ReLU / GroupNorm statistics%s", "");
    if (a.ksplit > 1) {             // raw partial sums into a.ws; the caller reduces
        const int nch = (a.ga_stride ? 9 : 1) * (a.Cin >> 4);
        if (!a.ws || a.res_mode == 2 || a.pool_ws || nch % (2 * a.ksplit))
            return fail(CMK_EINVAL, "conv_pw: split-K needs a workspace, K chunks %% (2*splitk) == 0, no upsampled residual / pooled sums%s", "");
        if (a.ga_stride) return mt == 4 ? launch_pw_mt<4, false, true, false, true>(a, st) : mt == 2 ? launch_pw_mt<2, false, true, false, true>(a, st)
                                                                                                      : fail(CMK_EINVAL, "conv_pw: tile height must be 4 or 2%s", "");
        return mt == 4 ? launch_pw_mt<4, false, false, false, true>(a, st) : mt == 2 ? launch_pw_mt<2, false, false, false, true>(a, st)
                                                                                     : fail(CMK_EINVAL, "conv_pw: tile height must be 4 or 2%s", "");
    }
    if (a.res_mode == 2) {          // FPN top-down add
        if (a.ga_stride || a.pool_ws || (p.Wo & 1) || (long)p.N * a.Hr * a.Wr * a.res_cs * 4 >= (1L << 31))
            return fail(CMK_EINVAL, "conv_pw: the upsampled residual needs a 1x1 conv, an even output width, no pooled sums, a residual below 2 GiB%s", "");
    }
    if ((a.Cin & 31) || (a.cout_pad & 127)) return fail(CMK_EINVAL, "conv_pw: needs Cin %% 32 == 0 and Cout in 97..128 or > 224%s", "");
    const long in_pix = a.ga_stride ? (long)p.N * p.H * p.W : p.total_pix;
    if (in_pix * a.x_cs * 4 >= (1L << 31)) return fail(CMK_EINVAL, "conv_pw: input view of 2 GiB or more%s", "");
    if ((long)(64 * 4 + 8) * a.y_cs >= (1L << 30) || (long)(64 * 4 + 8) * a.res_cs >= (1L << 30)) return fail(CMK_EINVAL, "conv_pw: output row too wide%s", "");
    if (a.ga_stride) {
        if (a.pool_ws || p.H >= 32768 || p.W >= 32768) return fail(CMK_EINVAL, "conv_pw: gather form: no pooled sums, maps below 32768 x 32768%s", "");
        if (mt == 4) return launch_pw_mt<4, false, true>(a, st);
        if (mt == 2) return launch_pw_mt<2, false, true>(a, st);
        return fail(CMK_EINVAL, "conv_pw: tile height must be 4 or 2%s", "");
    }
    if (a.pool_ws && (long)p.Ho * p.Wo < 32 * mt) return fail(CMK_EINVAL, "conv_pw: pooled sums need H*W >= the block of %s%ld rows", "", 32 * mt);
    if (a.res_mode == 2) {
        if (mt == 4) return launch_pw_mt<4, false, false, true>(a, st);
        if (mt == 2) return launch_pw_mt<2, false, false, true>(a, st);
        return fail(CMK_EINVAL, "conv_pw: tile height must be 4 or 2%s", "");
    }
    if (mt == 4) return a.pool_ws ? launch_pw_mt<4, true, false>(a, st) : launch_pw_mt<4, false, false>(a, st);
    if (mt == 2) return a.pool_ws ? launch_pw_mt<2, true, false>(a, st) : launch_pw_mt<2, false, false>(a, st);
    return fail(CMK_EINVAL, "conv_pw: tile height must be 4 or 2%s", "");
}

// The split forms (cmk.h tune_wm 10: three bf16 pieces, six products; 12: two fp16 pieces, three products): a.w is the split packing; plain 1x1
// conv, optionally with the pooled sums / the upsampled residual, or a 3x3 conv in the gather form.
template <int SPLIT>
static int launch_pw_split_mode(ConvArgs& a, hipStream_t st) {
    const ConvProblem& p = a.p[0];
    if (a.nprob != 1 || p.in_scale || a.in_relu || a.gn_ws || a.ksplit > 1 || a.res_mode == 1)
        return fail(CMK_EINVAL, "conv_pw (split): one problem, no input affine / input ReLU / GroupNorm statistics / split-K / same-size residual%s", "");
    if ((a.Cin & 15) || (a.cout_pad & 127)) return fail(CMK_EINVAL, "conv_pw (split): needs Cin %% 16 == 0 and a cout padding of 128%s", "");
    const long in_pix = a.ga_stride ? (long)p.N * p.H * p.W : p.total_pix;
    if (in_pix * a.x_cs * 4 >= (1L << 31)) return fail(CMK_EINVAL, "conv_pw (split): input view of 2 GiB or more%s", "");
    if ((long)(64 * 4 + 8) * a.y_cs >= (1L << 30) || (long)(64 * 4 + 8) * a.res_cs >= (1L << 30)) return fail(CMK_EINVAL, "conv_pw (split): output row too wide%s", "");
    if (a.ga_stride) {              // 3x3 conv (stride 1 | 2) as the gather GEMM over 9 taps
        if (a.pool_ws || a.res_mode || p.H >= 32768 || p.W >= 32768) return fail(CMK_EINVAL, "conv_pw (split): gather form: no pooled sums / residual, maps below 32768 x 32768%s", "");
        return launch_pw_mt<4, false, true, false, false, SPLIT>(a, st);
    }
    if (a.res_mode == 2) {          // FPN top-down add in the epilogue
        if (a.pool_ws || (p.Wo & 1) || (long)p.N * a.Hr * a.Wr * a.res_cs * 4 >= (1L << 31))
            return fail(CMK_EINVAL, "conv_pw (split): the upsampled residual needs an even output width, no pooled sums, a residual below 2 GiB%s", "");
        return launch_pw_mt<4, false, false, true, false, SPLIT>(a, st);
    }
    if (a.pool_ws && (long)p.Ho * p.Wo < 128) return fail(CMK_EINVAL, "conv_pw (split): pooled sums need H*W >= 128%s", "");
    return a.pool_ws ? launch_pw_mt<4, true, false, false, false, SPLIT>(a, st) : launch_pw_mt<4, false, false, false, false, SPLIT>(a, st);
}

int launch_pw_split(ConvArgs& a, int mode, hipStream_t st) {
    if (mode == 2 && !(a.p[0].acc_scale > 0.f)) return fail(CMK_EINVAL, "conv_pw (split): w_splith_scale missing%s", "");
    return mode == 2 ? launch_pw_split_mode<2>(a, st) : launch_pw_split_mode<1>(a, st);
}

}  // namespace cmk
