// OPT-IN (cmk.h tune_wm 11): 3x3 stride-1 convolution as a DIRECT implicit GEMM on fp16-split products, v_mfma_f32_32x32x16_f16.
//
// Why: the fp32 matrix instruction (157 TFLOP/s) holds the F(4x4,3x3) Winograd kernels at 75-90 executed = 300-360 direct-equivalent
// TFLOP/s, for structural reasons.  The 16-bit instructions run at 2010.  Every fp32 operand is split into TWO fp16
// pieces — x = h + m, h = fp16(x), m = fp16(x - h): 11 + 11 = 22 bits of significand, the residual x - h is exact in fp32 — and the three
// products m*h, h*m, h*h are accumulated in fp32, small terms first (what is dropped, m*m, is 2^-22 of the product).  Measured: the
// representation error is a third of an fp32 GEMM's own accumulation error (profiles/r03_conv_sp3.txt section 3), i.e. the result carries the
// error of an fp32 accumulation.  (Two BF16 pieces, 16 bits, were built first and measured 5x the fp32 path's end-to-end error: reg 1.4e-3 against
// the 1e-3 bar; three bf16 pieces / six products are fp32-accurate but slower than Winograd on a 3x3 conv — same file.)
// fp16's narrow exponent is handled by scaling, all powers of two (exact):
//   activations  x' = x * 2^-4 (|x| up to 1e6 stays finite); the residual is stored as fp16((x' - h) * 2^11), so it keeps 11 bits down to
//                |x| = 2^-21 (unscaled it would be subnormal below |x| = 0.06), and the weight piece it meets is multiplied by 2^-11 in
//                registers (4 packed multiplies per tap and cout tile);
//   weights      w' = w * S_w with S_w the power of two that puts max |w'| in [2^14, 2^15) (host, per conv): both pieces of every weight
//                larger than 2^-17 of the largest are normal fp16;
//   the accumulator is multiplied by 2^4 / S_w in the epilogue (folded into the per-channel scale).
// No transform, so no Winograd error amplification either.
//
// What the gather form of conv_pw.hip (tune_wm 10 / 12 on a 3x3 conv) pays nine times — the activation load, the split, the LDS write — is
// paid once here: a workgroup stages the HALO of its pixel tile, 16 input channels at a time, split into pieces, as
//   LDS [stage 2][piece P][k half 2][halo row][pitch][8 fp16]
// and the nine taps read the MFMA's A operand (32 pixels = a 4 x 8 patch, lane (li, hh) = pixel li, channels 8hh..8hh+7 of the chunk)
// from it at a shifted address (an immediate offset of the ds_read).  The row pitch (40 | 24 sixteen-byte units) makes the four rows of a
// patch tile the 512-byte bank space.  The weights come straight from L2/L1 into registers (buffer loads: lane offset + scalar offset), two taps
// ahead on three register sets, one request per patch (cmk_conv_desc.w_splith: [tap][Cin/16][cout_pad/32][piece 2][lane][8 fp16]).
//
// Workgroup = 4 waves, wave tile = 4 patches (128 pixels) x NB cout tiles of 32, accumulators as in conv_pw (pixels on the rows, couts on the lanes):
//   GEO 0: waves 2 (pixels) x 2 (couts);  tile  8 rows x 32 columns x 128 couts      the large maps of 128-cout layers (200 x 320); NB = 1 (64 couts) for stem_2
//   GEO 1: waves 1 x 4;                   tile  4 rows x 32 columns x 256 couts      100 x 160 maps of 256-cout layers (25 x 5 tiles, no waste)
//   GEO 2: waves 2 x 2;                   tile 16 rows x 16 columns x 128 couts      RoI maps (14 x 14: one map per tile)
//   GEO 3: waves 1 x 4;                   tile  8 rows x 16 columns x 256 couts      50 x 80 maps of 256-cout layers
// Two workgroups per CU.  One barrier per 16-channel chunk = per 216 (P = 2) MFMAs of a wave.
// AFF: x' = relu(x * in_scale[n][c] + in_shift[n][c]) while staging (the fused GroupNorm + ReLU of the FCOS towers), padding stays zero.
//
// Reference call sites replaced (when opted in): conv3x3 of the OSA modules (vovnet.py:205-219), the FPN output convs, the FCOS towers
// (fcos.py:170-186), the mask / mask-IoU head convs (sam.py:53-66, maskiou_head.py:80-93).
#include <math.h>

#include <type_traits>

#include "conv_args.hpp"

namespace cmk {

typedef int sp3_i32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 sp3_buffer_load(sp3_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

template <int GEO> struct SP3G;
template <> struct SP3G<0> { static constexpr int WCOLS = 2, BR = 2, BC = 4; };
template <> struct SP3G<1> { static constexpr int WCOLS = 4, BR = 1, BC = 4; };
template <> struct SP3G<2> { static constexpr int WCOLS = 2, BR = 4, BC = 2; };
template <> struct SP3G<3> { static constexpr int WCOLS = 4, BR = 2, BC = 2; };

template <int GEO, int P>
struct SP3L {
    typedef SP3G<GEO> G;
    static constexpr int TH = 4 * G::BR, TW = 8 * G::BC;            // output pixels of a workgroup
    static constexpr int HR = TH + 2, HC = TW + 2;                  // its halo
    static constexpr int PITCH = G::BC == 4 ? 40 : 24;              // 16-byte units per halo row: = 8 | 24 (mod 32)
    static constexpr int PLANE = HR * PITCH * 16;                   // bytes of one (piece, k half) plane
    static constexpr int STAGE = P * 2 * PLANE;
    static constexpr int ITEMS = HR * HC * 4;                       // float4 loads per chunk
    static constexpr int IT = (ITEMS + 255) / 256;
    static constexpr int LDS_BYTES = 2 * STAGE;
#ifdef SP3_TRACE
    static constexpr int LDS_ALLOC = LDS_BYTES + 4 * 128 * 8;
#else
    static constexpr int LDS_ALLOC = LDS_BYTES;
#endif
};

#ifndef SP3_ABL
#define SP3_ABL 0      // timing ablations (results wrong): 1 no activation loads, 2 no split / LDS writes, 4 no weight loads, 8 no A reads, 16 no MFMAs, 32 no stores, 64 all weight requests to the same lines, 128 no 2^-11 multiply of the weight piece
#endif

template <int GEO, int NB, int P, bool AFF>
__global__ __launch_bounds__(256, 2) void conv_sp3_kernel(const ConvArgs a) {
    typedef SP3G<GEO> G;
    typedef SP3L<GEO, P> L;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    constexpr int WROWS = 4 / G::WCOLS;
    constexpr int IT = L::IT;
    static_assert((IT - 1) * 256 < L::ITEMS && IT <= 8, "only a thread's last staging item may fall outside the halo; the requests are spread over 8 slots");
    extern __shared__ __attribute__((aligned(16))) unsigned char sb[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, li = lane & 31;
    const int wr = wave % WROWS, wc = wave / WROWS;

    // XCD-aware order: the cout tiles of one pixel tile run back to back on one XCD (conv_pw.hip)
    const int xq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int bx = (xq / a.grid_y) * 8 + xcd, by = xq % a.grid_y;
    if (bx >= a.total_tiles) return;
    int pi = 0;
    for (int i = 1; i < a.nprob; ++i)
        if (bx >= a.p[i].tile_begin) pi = i;
    const ConvProblem& Pb = a.p[pi];
    const int H = Pb.H, W = Pb.W;
    int t = bx - Pb.tile_begin;
    const int per_img = Pb.tiles_h * Pb.tiles_w;
    const int n = t / per_img;
    t -= n * per_img;
    const int trow = t / Pb.tiles_w, tcol = t - trow * Pb.tiles_w;
    const int oy0 = trow * L::TH, ox0 = tcol * L::TW;
    const int nchunks = a.Cin >> 4;
    const int nblocks = a.cout_pad >> 5;
    const int cb0 = (by * G::WCOLS + wc) * NB;          // this wave's first cout tile of 32

    // ---- staging set-up: item idx = it * 256 + tid -> halo pixel idx >> 2, channel quad idx & 3 (= tid & 3) -----------------------------------
    sp3_i32x4 rsrc;
    {
        const unsigned long long base = (unsigned long long)Pb.x;
        rsrc.x = __builtin_amdgcn_readfirstlane((int)(base & 0xffffffffull));
        rsrc.y = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffffull));
        rsrc.z = __builtin_amdgcn_readfirstlane((int)((long)Pb.N * H * W * a.x_cs * 4));      // < 2^31 (host)
        rsrc.w = 0x00020000;
    }
    const int q = tid & 3;
    int st_voff[IT], st_dst[IT];        // byte offset into x (outside the resource: zeros), byte offset into a stage (-1: no item)
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = it * 256 + tid;
        const int px = idx >> 2;
        const int hr = px / L::HC, hc = px - hr * L::HC;
        const int ih = oy0 + hr - 1, iw = ox0 + hc - 1;
        const bool item = idx < L::ITEMS;
        const bool inside = item && ih >= 0 && ih < H && iw >= 0 && iw < W;
        st_voff[it] = inside ? (((n * H + ih) * W + iw) * a.x_cs + a.x_co + q * 4) * 4 : (int)0x80000000;
        st_dst[it] = item ? (q >> 1) * L::PLANE + (hr * L::PITCH + hc) * 16 + (q & 1) * 8 : -1;
    }
    f32x4 st[IT];
    auto load_X1 = [&](int chunk, int it) {
#if !(SP3_ABL & 1)
        st[it] = sp3_buffer_load(rsrc, st_voff[it], chunk * 64, 0);
#endif
    };
    auto load_X = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < IT; ++it) load_X1(chunk, it);
    };
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr float SX = 0.0625f, RS = 2048.f;       // activation scale 2^-4, residual scale 2^11
    auto pk = [](float x, float y) { return __builtin_bit_cast(unsigned, f16x2{(_Float16)x, (_Float16)y}); };       // round to nearest even
    auto unpk = [](unsigned p_) { const f16x2 h_ = __builtin_bit_cast(f16x2, p_); return f32x2{(float)h_.x, (float)h_.y}; };
    auto stage = [&](int buf, int chunk) {
#if !(SP3_ABL & 2)
        unsigned char* dst = sb + buf * L::STAGE;
        f32x4 isc = {SX, SX, SX, SX}, ish = {0.f, 0.f, 0.f, 0.f};
        if constexpr (AFF) {
            isc = *reinterpret_cast<const f32x4*>(Pb.in_scale + (long)n * a.Cin + chunk * 16 + q * 4) * SX;
            ish = *reinterpret_cast<const f32x4*>(Pb.in_shift + (long)n * a.Cin + chunk * 16 + q * 4) * SX;
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            f32x4 x = st[it];
            if constexpr (AFF) {
                const bool in = st_voff[it] >= 0;           // the padding is zero AFTER the normalisation
                x.x = in ? fmaxf(fmaf(x.x, isc.x, ish.x), 0.f) : 0.f;
                x.y = in ? fmaxf(fmaf(x.y, isc.y, ish.y), 0.f) : 0.f;
                x.z = in ? fmaxf(fmaf(x.z, isc.z, ish.z), 0.f) : 0.f;
                x.w = in ? fmaxf(fmaf(x.w, isc.w, ish.w), 0.f) : 0.f;
            } else {
                x *= SX;
            }
            u32x2 h, m_;
            h.x = pk(x.x, x.y); h.y = pk(x.z, x.w);
            const f32x2 h01 = unpk(h.x), h23 = unpk(h.y);
            const f32x4 r1 = f32x4{x.x - h01.x, x.y - h01.y, x.z - h23.x, x.w - h23.y} * RS;      // exact: h holds the leading bits of x
            m_.x = pk(r1.x, r1.y); m_.y = pk(r1.z, r1.w);
            if (it < IT - 1 || st_dst[it] >= 0) {
                *reinterpret_cast<u32x2*>(dst + st_dst[it]) = h;
                *reinterpret_cast<u32x2*>(dst + st_dst[it] + 2 * L::PLANE) = m_;
            }
        }
#endif
    };

    // ---- weights: wave-uniform base + lane ----------------------------------------------------------------------------------------------------
    // (problems of one launch may differ in weights: the two FCOS towers)  Buffer loads: the lane offset is the only vector operand, the
    // position of the (tap, chunk, cout tile, piece) KiB is a scalar offset — no 64-bit vector address arithmetic per request
    sp3_i32x4 wrsrc;
    {
        const unsigned long long base = (unsigned long long)Pb.w;
        wrsrc.x = __builtin_amdgcn_readfirstlane((int)(base & 0xffffffffull));
        wrsrc.y = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffffull));
        wrsrc.z = __builtin_amdgcn_readfirstlane(9 * nchunks * nblocks * P * 1024);       // < 2^31 (host)
        wrsrc.w = 0x00020000;
    }
    const int wlane = lane * 16;
    int wblk[NB];
#pragma unroll
    for (int nn = 0; nn < NB; ++nn) wblk[nn] = ((SP3_ABL & 64) ? nn : min(cb0 + nn, nblocks - 1)) * P * 64;       // (tiles past cout_pad: any valid block, never stored)
    static_assert(P == 2, "two fp16 pieces per operand");
    constexpr int BD = 2;                   // taps a weight request runs ahead (BD + 1 register sets; 9 taps % (BD + 1) == 0: no set is indexed at run time)
    constexpr int ST = 3, XT = 4;           // the tap after whose MFMAs the next chunk's halo is split and written / at which the one after is requested
    u32x4 wq[BD + 1][NB][P];
    auto load_B = [&](int set, int chunk, int tap, int part = -1) {      // part: one of the NB * P registers, -1: all
#if !(SP3_ABL & 4)
#if SP3_ABL & 64
        const long k = 0 * (tap + chunk);        // every request of the launch hits the same lines
#else
        const long k = ((long)tap * nchunks + chunk) * nblocks * P * 64;
#endif
#pragma unroll
        for (int j = 0; j < NB * P; ++j)
            if (part < 0 || part == j)
                wq[set][j / P][j % P] = __builtin_bit_cast(u32x4, sp3_buffer_load(wrsrc, wlane, (int)(k + wblk[j / P] + (j % P) * 64) * 16, 0));
#endif
    };

    // ---- A operand addresses: patch b = wr * 4 + m -> (b / BC, b % BC); lane -> pixel (li >> 3, li & 7) of the patch, k half hh ----------------
    int a_addr[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int b = wr * 4 + m;
        const int br = b / G::BC, bc = b % G::BC;
        a_addr[m] = hh * L::PLANE + ((br * 4 + (li >> 3)) * L::PITCH + bc * 8 + (li & 7)) * 16;
    }

#ifdef SP3_TRACE
    // instrumented build (tools/ab/trace_sp3.py): lane 0 of every wave of every 8th workgroup stamps the shader clock into LDS (a global store in
    // the loop would make every wait drain it) and copies the stamps to a.ws at the end: [0] start, [1] prologue done, chunk c < 6 at 2 + 11 c:
    // barrier reached, passed, taps 0..8 done; [120] loop done, [121] stores issued, [122] / [123] real time at the end / start, [124] HW id
    const bool tracing = a.ws && (blockIdx.x % 8) == 0 && lane == 0;
    unsigned long long* trl = reinterpret_cast<unsigned long long*>(sb + L::LDS_BYTES) + wave * 128;
    unsigned long long* trc = reinterpret_cast<unsigned long long*>(a.ws) + ((blockIdx.x / 8) * 4 + wave) * 128;
#define SP3_STAMP(slot) do { if (tracing) trl[slot] = __builtin_readcyclecounter(); } while (0)
    if (tracing) {
        for (int i = 0; i < 128; ++i) trl[i] = 0;
        trl[123] = __builtin_amdgcn_s_memrealtime();
        trl[124] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));
    }
    SP3_STAMP(0);
#else
#define SP3_STAMP(slot) do { } while (0)
#endif
    f32x16 acc[4][NB];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < NB; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;

    load_X(0);
#pragma unroll
    for (int t_ = 0; t_ < BD; ++t_) load_B(t_, 0, t_);
    stage(0, 0);
    if (nchunks > 1) load_X(1);
    // one 16-channel chunk = nine taps; the weights of tap t + BD are requested at tap t into register set (t + BD) % (BD + 1) — an L2 round
    // trip is about one tap of a wave's MFMAs, and the wait counter is in order: a halo request (HBM latency) issued between two weight requests
    // is waited for with the second one, BD taps later
    auto chunk_body = [&](int c) {
#ifdef SP3_TRACE
        if (c < 6) SP3_STAMP(2 + 11 * c);
#endif
        __syncthreads();                // stage c & 1 is complete; everybody has read all of the other stage
#ifdef SP3_TRACE
        if (c < 6) SP3_STAMP(3 + 11 * c);
#endif
        const unsigned char* As = sb + (c & 1) * L::STAGE;
        u32x4 av[2][P];                 // the A operands of patch m are read while patch m - 1 is multiplied
        auto read_A = [&](int slot, int tap, int m) {
            const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int p_ = 0; p_ < P; ++p_) {
#if SP3_ABL & 8
                av[slot][p_] = u32x4{(unsigned)tap, (unsigned)m, (unsigned)p_, 0u};
#else
                av[slot][p_] = *reinterpret_cast<const u32x4*>(As + a_addr[m] + (kh * L::PITCH + kw) * 16 + p_ * 2 * L::PLANE);
#endif
            }
        };
        read_A(0, 0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int set = tap % (BD + 1);
            f16x8 Bhs[NB];              // the main weight piece x 2^-11: meets the activations' scaled residual
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                // the requests are spread over the patches: four in a row find the texture unit's queue full and hold the wave's MFMAs back
                constexpr int PER = (NB * P + 3) / 4;
#pragma unroll
                for (int j = m * PER; j < (m + 1) * PER && j < NB * P; ++j) {
                    if (tap + BD < 9) load_B((tap + BD) % (BD + 1), c, tap + BD, j);
                    else load_B((tap + BD) % (BD + 1), min(c + 1, nchunks - 1), tap + BD - 9, j);
                }
                if ((tap == XT || tap == XT + 1) && c + 2 < nchunks) {
                    constexpr int XPER = (IT + 7) / 8;
#pragma unroll
                    for (int it = ((tap - XT) * 4 + m) * XPER; it < ((tap - XT) * 4 + m + 1) * XPER && it < IT; ++it) load_X1(c + 2, it);
                }
                if (m < 3) read_A((m + 1) & 1, tap, m + 1);
                else if (tap < 8) read_A((m + 1) & 1, tap + 1, 0);
                __builtin_amdgcn_sched_barrier(0);        // (the compiler otherwise sinks the requests to just before their use: no run-ahead left)
                const f16x8 Ah = __builtin_bit_cast(f16x8, av[m & 1][0]), Am = __builtin_bit_cast(f16x8, av[m & 1][1]);
#if !(SP3_ABL & 16)
                if (m == 0) {
#pragma unroll
                    for (int nn = 0; nn < NB; ++nn)
                        Bhs[nn] = (SP3_ABL & 128) ? __builtin_bit_cast(f16x8, wq[set][nn][0]) : __builtin_bit_cast(f16x8, wq[set][nn][0]) * (_Float16)(1.f / RS);
                }
#pragma unroll
                for (int nn = 0; nn < NB; ++nn) {
                    const f16x8 Bh = __builtin_bit_cast(f16x8, wq[set][nn][0]), Bm = __builtin_bit_cast(f16x8, wq[set][nn][1]);
                    f32x16 cacc = acc[m][nn];
                    cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Am, Bhs[nn], cacc, 0, 0, 0);
                    cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bm, cacc, 0, 0, 0);
                    cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, cacc, 0, 0, 0);
                    acc[m][nn] = cacc;
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tap == ST && c + 1 < nchunks) {       // the next chunk's halo: in registers since taps XT, XT + 1 of the last chunk
                stage((c + 1) & 1, c + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef SP3_TRACE
            if (c < 6) SP3_STAMP(4 + 11 * c + tap);
#endif
        }
    };
    SP3_STAMP(1);
    for (int c = 0; c < nchunks; ++c) chunk_body(c);
    SP3_STAMP(120);

    // ---- epilogue: scale / shift (+ReLU), NHWC stores: accumulator register r of lane half hh is pixel (r >> 2, (r & 3) + 4 hh) of the patch ---
#if !(SP3_ABL & 32)
    float* yimg = Pb.y + (long)n * H * W * a.y_cs + a.y_co;
    const bool interior = oy0 + L::TH <= H && ox0 + L::TW <= W && (cb0 + NB) * 32 <= a.Cout;
    // fused GroupNorm statistics of the NEXT layer's normalisation (fcos.py:182-186): one {sum, sumsq} record per (spatial tile, pixel row of
    // waves), every group of it written by the wave that owns those couts (cmk_conv_gn_records: WROWS records per tile)
    auto trace_out = [&]() {
#ifdef SP3_TRACE
        SP3_STAMP(121);
        if (tracing) {
            trl[122] = __builtin_amdgcn_s_memrealtime();
            for (int i = 0; i < 128; ++i) trc[i] = trl[i];
        }
#endif
    };
    const bool want_stats = a.gn_ws != nullptr;
    const float acc_scale = Pb.acc_scale * (1.f / SX);          // 1 / (S_x * S_w)
    auto put_stats = [&](int co, bool cvalid, float gs, float gss) {
        for (int o = 1; o < a.gn_cpg; o <<= 1) { gs += __shfl_xor(gs, o); gss += __shfl_xor(gss, o); }
        gs += __shfl_xor(gs, 32);
        gss += __shfl_xor(gss, 32);
        if (cvalid && hh == 0 && (li & (a.gn_cpg - 1)) == 0) {
            double* o = a.gn_ws + (((long)bx * WROWS + wr) * a.gn_groups + co / a.gn_cpg) * 2;
            o[0] = (double)gs;
            o[1] = (double)gss;
        }
    };
    if (interior) {
        // 1.5 VALU per stored value (one packed fma per two, one max each); the address is a wave-uniform pointer walked by the scalar unit plus a
        // fixed lane offset (global_store saddr form, written as asm: the compiler renders the same C as 64-bit vector adds per store — conv_pw.hip)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const unsigned long long yb = (unsigned long long)yimg;
        const unsigned long long ybs = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(yb >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)yb);
        const unsigned long long px_b = (unsigned long long)a.y_cs * 4u, rowskip_b = (unsigned long long)(W - 3) * a.y_cs * 4u;
#pragma unroll
        for (int nn = 0; nn < NB; ++nn) {
            const int co = (cb0 + nn) * 32 + li;
            float sc = Pb.scale[co] * acc_scale, sh = Pb.shift[co];
            const float lo = co < a.relu_upto ? 0.f : __builtin_nanf("");      // max(v, NaN) = v: lanes without the ReLU
            asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(sc), "+v"(sh));     // waited for once, here
            const f32x2 sc2 = {sc, sc}, sh2 = {sh, sh};
            const unsigned voff = (unsigned)(4 * hh * a.y_cs + co) * 4u;
            f32x2 gs2 = {0.f, 0.f}, gss2 = {0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int b = wr * 4 + m;
                unsigned long long rowp = ybs + ((unsigned long long)(oy0 + (b / G::BC) * 4) * W + (ox0 + (b % G::BC) * 8)) * px_b;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 v = __builtin_elementwise_fma(f32x2{acc[m][nn][r], acc[m][nn][r + 1]}, sc2, sh2);
                    const float v0 = fmaxf(v.x, lo), v1 = fmaxf(v.y, lo);
                    if (want_stats) {                     // (relu_upto == 0 with fused statistics: v0, v1 are v)
                        gs2 += v;
                        gss2 = __builtin_elementwise_fma(v, v, gss2);
                    }
                    asm volatile("global_store_dword %1, %2, %0" : "+s"(rowp) : "v"(voff), "v"(v0) : "memory");
                    rowp += px_b;
                    asm volatile("global_store_dword %1, %2, %0" : "+s"(rowp) : "v"(voff), "v"(v1) : "memory");
                    rowp += (r & 3) == 2 ? rowskip_b : px_b;
                }
            }
            if (want_stats) put_stats(co, true, gs2.x + gs2.y, gss2.x + gss2.y);
        }
        trace_out();
        return;
    }
    // border tiles: the same walk; a store is predicated on its row (wave-uniform) and its column / cout (per lane, four compares per patch)
    {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const unsigned long long yb = (unsigned long long)yimg;
        const unsigned long long ybs = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(yb >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)yb);
        const unsigned long long px_b = (unsigned long long)a.y_cs * 4u, rowskip_b = (unsigned long long)(W - 3) * a.y_cs * 4u;
#pragma unroll
        for (int nn = 0; nn < NB; ++nn) {
            const int co = (cb0 + nn) * 32 + li;
            const bool cvalid = co < a.Cout;
            float sc = cvalid ? Pb.scale[co] * acc_scale : 0.f, sh = cvalid ? Pb.shift[co] : 0.f;
            const float lo = co < a.relu_upto ? 0.f : __builtin_nanf("");
            asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(sc), "+v"(sh));
            const f32x2 sc2 = {sc, sc}, sh2 = {sh, sh};
            const unsigned voff = (unsigned)(4 * hh * a.y_cs + co) * 4u;
            float gs_b = 0.f, gss_b = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int b = wr * 4 + m;
                const int oy = oy0 + (b / G::BC) * 4, ox = ox0 + (b % G::BC) * 8;
                unsigned long long rowp = ybs + ((unsigned long long)oy * W + ox) * px_b;
                unsigned long long xm[4];          // lanes whose column (and cout) exists, per column of the lane half
                float fx[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = cvalid && ox + 4 * hh + j < W;
                    xm[j] = __ballot(ok);
                    fx[j] = ok ? 1.f : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 v = __builtin_elementwise_fma(f32x2{acc[m][nn][r], acc[m][nn][r + 1]}, sc2, sh2);
                    const float v0 = fmaxf(v.x, lo), v1 = fmaxf(v.y, lo);
                    const bool yok = oy + (r >> 2) < H;            // wave-uniform
                    if (want_stats && yok) {
                        const float t0 = v.x * fx[r & 3], t1 = v.y * fx[(r & 3) + 1];
                        gs_b += t0 + t1;
                        gss_b = fmaf(t0, v.x, fmaf(t1, v.y, gss_b));
                    }
                    unsigned long long keep;
                    if (yok)
                        asm volatile("s_and_saveexec_b64 %0, %4\n\tglobal_store_dword %2, %3, %1\n\ts_mov_b64 exec, %0"
                                     : "=&s"(keep) : "s"(rowp), "v"(voff), "v"(v0), "s"(xm[r & 3]) : "memory", "scc");
                    rowp += px_b;
                    if (yok)
                        asm volatile("s_and_saveexec_b64 %0, %4\n\tglobal_store_dword %2, %3, %1\n\ts_mov_b64 exec, %0"
                                     : "=&s"(keep) : "s"(rowp), "v"(voff), "v"(v1), "s"(xm[(r & 3) + 1]) : "memory", "scc");
                    rowp += (r & 3) == 2 ? rowskip_b : px_b;
                }
            }
            if (want_stats) put_stats(co, cvalid, gs_b, gss_b);
        }
    }
    trace_out();
#endif
}

template <int GEO, int NB, int P>
static int launch_sp3_geo(ConvArgs& a, hipStream_t st) {
    typedef SP3L<GEO, P> L;
    static DeviceOnce once;
    int rc = once.run([]() {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sp3_kernel<GEO, NB, P, false>), hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS_ALLOC);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sp3_kernel<GEO, NB, P, true>), hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS_ALLOC);
        return e == hipSuccess ? CMK_OK : fail(CMK_ELAUNCH, "conv_sp3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    });
    if (rc) return rc;
    int blocks = 0;
    for (int i = 0; i < a.nprob; ++i) {
        ConvProblem& p = a.p[i];
        p.tile_begin = blocks;
        p.tiles_h = cdiv(p.Ho, L::TH);
        p.tiles_w = cdiv(p.Wo, L::TW);
        blocks += p.N * p.tiles_h * p.tiles_w;
    }
    a.total_tiles = blocks;
    a.grid_y = cdiv(a.Cout, 32 * NB * SP3G<GEO>::WCOLS);
    const dim3 grid(((blocks + 7) / 8) * 8 * a.grid_y);
    if (a.p[0].in_scale)
        hipLaunchKernelGGL((conv_sp3_kernel<GEO, NB, P, true>), grid, dim3(256), L::LDS_ALLOC, st, a);
    else
        hipLaunchKernelGGL((conv_sp3_kernel<GEO, NB, P, false>), grid, dim3(256), L::LDS_ALLOC, st, a);
    return check_launch("conv_sp3");
}

// geo 0..3 (above), pieces = 2.  p[i].w = the fp16 split packing (tap-major), p[i].acc_scale = 1 / S_w, a.cout_pad = the packing's padded Cout (a multiple of 128).
int launch_sp3(ConvArgs& a, int geo, int pieces, hipStream_t st) {
    if ((a.Cin & 15) || a.cout_pad % 128 || a.cout_pad < a.Cout || a.ksplit > 1 || a.res_mode != 0 || a.in_relu ||
        9L * (a.Cin >> 4) * (a.cout_pad >> 5) * 2 * 1024 >= (1L << 31))
        return fail(CMK_EINVAL, "conv_sp3: needs Cin %% 16 == 0, no split-K / residual, split weights below 2 GiB%s", "");
    for (int i = 0; i < a.nprob; ++i) {
        const ConvProblem& p = a.p[i];
        if ((long)p.N * p.H * p.W * a.x_cs * 4 >= (1L << 31) || p.Ho != p.H || p.Wo != p.W || (!p.in_scale) != (!a.p[0].in_scale) || !p.w || !(p.acc_scale > 0.f))
            return fail(CMK_EINVAL, "conv_sp3: an input of 2 GiB or more, a strided conv, problems that differ in the input affine, or no split weights%s", "");
    }
    // tune_sc 2: two pieces per operand; 21: the same with ONE cout tile per wave (workgroups of 64 couts in geometries 0 / 2, of 128 in 1 / 3): less
    // cout padding (three 64-cout workgroups cover 160 or 192 couts where two 128-cout ones pad them to 256), 3-4x the workgroups on the small maps, and
    // 158 registers = three workgroups per CU; at the price of staging the halo once per 64 (128) couts.  Geometry 0 takes it by itself up to 64 couts.
    if (pieces != 2 && pieces != 21)
        return fail(CMK_EINVAL, "conv_sp3: tune_sc must be 2 (two pieces per operand) or 21 (the same, one cout tile per wave)%s", "");
    const bool nb1 = pieces == 21 || (geo == 0 && a.Cout <= 64);
    switch (geo) {
        case 0: return nb1 ? launch_sp3_geo<0, 1, 2>(a, st) : launch_sp3_geo<0, 2, 2>(a, st);
        case 1: return nb1 ? launch_sp3_geo<1, 1, 2>(a, st) : launch_sp3_geo<1, 2, 2>(a, st);
        case 2: return nb1 ? launch_sp3_geo<2, 1, 2>(a, st) : launch_sp3_geo<2, 2, 2>(a, st);
        case 3: return nb1 ? launch_sp3_geo<3, 1, 2>(a, st) : launch_sp3_geo<3, 2, 2>(a, st);
    }
    return fail(CMK_EINVAL, "conv_sp3: geometry 0..3%s", "");
}

}  // namespace cmk
