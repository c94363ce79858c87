// Winograd F(4x4, 3x3), "shared V" form: ONE workgroup of 8 waves per CU computes 64 output channels of a spatial tile from one
// frequency image of the input held in LDS.
//
// conv_wino6.hip (32 couts per workgroup, two workgroups per CU) redoes the whole input transform — halo fetch, column pass, row pass —
// for every 32-cout tile, and its row pass sits in the MFMA waves: 60 packed VALU instructions per 36 MFMAs.  fp32 MFMA and VALU do not
// co-execute on gfx950 (they share the FMA hardware), so those instructions, not the MFMAs, were what kept the matrix pipe at 60 %
// (profiles/r02_pmc_mfma.json), and the repeated halo fetch is where the 2.25-4.1x of algorithmic traffic came from
// (profiles/r02_pmc_traffic.json).  Here the transform is done ONCE per spatial tile and 8-channel chunk, by all
// 512 threads together, and its result feeds 64 couts:
//   pass 1  thread = (tile row, halo column, channel PAIR): 6 bounds-checked 8-byte loads one period ahead, 12 packed VALU, 6 ds_write_b64
//           into the column-transformed image W (the same conflict-free slot function as conv_wino6.hip);
//   pass 2  24 wave-tasks (grid row a, half, channel pair), 3 per wave: 5 ds_read_b64 of W, 6 packed VALU, 3 ds_write_b64 into the
//           frequency image V[36][quad][tile] — exactly the operand layout of the MFMA: a wave's A operand of (frequency, 4 channels)
//           is ONE ds_read_b128 of a contiguous KiB;
//   MFMA    wave = (cout tile ct = wave / 4, frequency group g = wave % 4) owns the same 9 frequencies as in conv_wino6.hip (grid row g
//           and half of row 4 or 5) for its 32 couts: 9 ds_read_b128 + 9 global_load_dwordx4 (weights, straight into registers, as
//           before) + 36 MFMAs per chunk and NO VALU instruction.
// Per chunk a wave issues 30 packed VALU instructions for 36 MFMAs (it was 88), the halo is fetched and transformed once per 64 couts,
// and one barrier per chunk suffices: everything inside a period is independent (pass 1 writes W(c+2), pass 2 turns W(c+1) into V(c+1),
// the MFMAs read V(c); two W and two V buffers, 128-140 KiB of LDS — one workgroup per CU, which the 160 KiB of gfx950 allow).
// The K order and the arithmetic of every transform are those of conv_wino6.hip: the results are bit-identical to it.
// A layer whose number of 32-cout tiles is odd leaves the second wave group of its last workgroup row without a tile: those waves take
// part in the transforms and barriers and skip the MFMAs.
//
// Reference call sites replaced: the 3x3 stride-1 convs, as conv_wino6.hip (vovnet.py:205-219, d2 FPN outputs, fcos.py:169-200,
// sam.py:58-70, maskiou_head.py:81-88).
#include <type_traits>

#include "wino6_common.hpp"

#ifndef W6S_FENCE
#define W6S_FENCE __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef W6S_PRIO
#define W6S_PRIO 1    // wave priority during the transform phase (0..3); measured 1.8 % faster than 0 on the model's map shapes, 2 and 3 the same
#endif
#ifndef W6S_ABL
#define W6S_ABL 0     // timing ablations (tools/ab; results are wrong with any set): 1 no pass 1, 2 no halo loads, 4 no weight loads, 8 no pass 2,
#endif                //   16 no MFMAs, 32 no barrier, 64 no V reads; finer: 128 no V stores (pass 2), 256 no W stores (pass 1), 512 no pass-2 VALU,
                      //   1024 no W sample reads, 2048 no pass-1 VALU

namespace cmk {

__device__ f32x2 w6s_buffer_load2(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");

template <int GEO> struct W6S {
    using G = W6G<GEO>;
    static constexpr int V_SLOTS = 36 * 64;                  // one V buffer in 16-byte slots: [36 frequencies][channel quad 2][tile 32]
    static constexpr int W_OFF = 2 * V_SLOTS;                // [V0 V1][W0 W1]
    static constexpr int SLOTS = 2 * V_SLOTS + 2 * G::WB;
    static constexpr int EX_BYTES = 2 * 64 * 1024;           // epilogue exchange, 64 KiB per cout tile (the layout of conv_wino6.hip, twice)
    static constexpr int LDS_BYTES = SLOTS * 16 > EX_BYTES ? SLOTS * 16 : EX_BYTES;
    // pass 1 items: (tile row, halo column, channel quad, channel pair)
    static constexpr int ITEMS = GEO == 0 ? 3 * G::HC * 4 : 4 * G::HC * 4;       // 504 of 512 threads | 240 of the 256 threads of an image
    __device__ static __forceinline__ void item_of(int tid, int& img, int& q, int& pair, int& t, int& col, bool& active) {
        const int i = GEO == 0 ? tid : (tid & 255);
        img = GEO == 0 ? 0 : (tid >> 8);
        active = i < ITEMS;
        const int j = min(i, ITEMS - 1);                      // the idle threads repeat the last item (same values, same slots)
        pair = j & 1; q = (j >> 1) & 1;
        const int cc = j >> 2;
        t = cc / G::HC; col = cc - t * G::HC;
    }
};
static_assert(W6S<0>::LDS_BYTES <= LDS_CU && W6S<1>::LDS_BYTES <= LDS_CU, "one workgroup per CU");
static_assert(W6S<0>::ITEMS <= 512 && W6S<1>::ITEMS <= 256, "one pass-1 item per thread");

template <bool AFF, int GEO>
__global__ __launch_bounds__(512, 2) void conv_wino6s_kernel(const ConvArgs a) {
    using G = W6G<GEO>;
    using S = W6S<GEO>;
    constexpr int AP = G::AP, WB = G::WB, K = G::CK, VS = S::V_SLOTS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* sV = reinterpret_cast<f32x4*>(smem);
    f32x4* sW = sV + S::W_OFF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, li = lane & 31;
    const int ct = wave >> 2, g = wave & 3;

    // XCD-aware order: the workgroup rows (pairs of cout tiles) of one spatial tile go to the same XCD, back to back
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
    int n, oh0, ow0;
    if (GEO == 0) {
        const int tw = tile % P.tiles_w;
        const int t2 = tile / P.tiles_w;
        const int th = t2 % P.tiles_h;
        n = t2 / P.tiles_h;
        oh0 = th * G::OH; ow0 = tw * G::OW;
    } else {                      // a pair of whole images
        n = tile * 2; oh0 = 0; ow0 = 0;
    }
    const int ntiles32 = (a.Cout + 31) >> 5;
    const int t32 = by * 2 + ct;
    const bool act = t32 < ntiles32;              // wave-uniform: this wave group has a cout tile
    const int co0 = t32 * 32;
    const int nchunks = a.Cin >> 3;

    // ---- pass 1 item of this thread ----------------------------------------------------------------------------------------------
    int p_img, p_q, p_pair, p_t, p_col;
    bool p_active;
    S::item_of(tid, p_img, p_q, p_pair, p_t, p_col, p_active);
    i32x4 rsrc;      // GEO 1: wave-uniform image (waves 0-3 the first image of the pair, waves 4-7 the second); an image past the batch reads zeros
    {
        const int img_n = n + (GEO == 1 ? (wave >> 2) : 0);
        const unsigned long long base = (unsigned long long)(P.x + (long)min(img_n, P.N - 1) * H * W * a.x_cs);
        rsrc.x = __builtin_amdgcn_readfirstlane((int)(base & 0xffffffffull));
        rsrc.y = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffffull));
        rsrc.z = __builtin_amdgcn_readfirstlane(img_n < P.N ? H * W * a.x_cs * 4 : 0);
        rsrc.w = 0x00020000;
    }
    const int row_bytes = W * a.x_cs * 4;
    const int ih0 = oh0 - 1 + 4 * p_t, iw = ow0 - 1 + p_col;
    // rows above/below the image are out of the resource's range by themselves; a column outside the image would alias the neighbouring
    // row, so it is pushed out of range
    const int voff0 = (iw >= 0 && iw < W) ? (ih0 * W + iw) * a.x_cs * 4 + (a.x_co + p_q * 4 + p_pair * 2) * 4 : (int)0x80000000;
    unsigned okm = 0;                                              // AFF only: relu(0*s + b) != 0, so padding needs the mask
    if (AFF) {
#pragma unroll
        for (int i = 0; i < 6; ++i) okm |= ((iw >= 0 && iw < W && ih0 + i >= 0 && ih0 + i < H) ? 1u : 0u) << i;
    }
    const unsigned aff_off = AFF ? (unsigned)(min(n + p_img, P.N - 1) * a.Cin + p_q * 4 + p_pair * 2) : 0u;
    const f32x2 five = {5.0f, 5.0f};
    f32x2 d[6];
    f32x2 in_sc = {1.f, 1.f}, in_sh = {0.f, 0.f};
    auto load_D = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = w6s_buffer_load2(rsrc, voff0 + i * row_bytes, chunk * 32, 0);
        if (AFF) {
            in_sc = *reinterpret_cast<const f32x2*>(P.in_scale + chunk * 8 + aff_off);
            in_sh = *reinterpret_cast<const f32x2*>(P.in_shift + chunk * 8 + aff_off);
        }
    };
    // f32x2 index of the item's W entry of grid row 0 inside a W buffer
    const int p_dst2 = w6_slot<GEO>(p_q, (GEO == 1 ? 4 * p_img : 0) + p_t, 0, p_col) * 2 + p_pair;
    f32x2* const wwr2 = reinterpret_cast<f32x2*>(sW) + p_dst2;
    auto pass1 = [&](f32x2* dst) {
        if (AFF) {
            // relu(x * s + b), and 0 for a sample outside the image: one packed fma per row, then one med3 per value —
            // med3(v, 0, +inf) = max(v, 0), med3(v, 0, 0) = 0 — the third operand made from the row's mask bit
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float kinf = ((okm >> i) & 1u) ? __builtin_inff() : 0.f;
                const f32x2 v = __builtin_elementwise_fma(d[i], in_sc, in_sh);
                d[i] = f32x2{__builtin_amdgcn_fmed3f(v.x, 0.f, kinf), __builtin_amdgcn_fmed3f(v.y, 0.f, kinf)};
            }
        }
        if (GEO == 1 && !p_active) return;
        f32x2 w0, w1, w2, w3, w4, w5;
#if W6S_ABL & 2048
        w0 = d[0]; w1 = d[1]; w2 = d[2]; w3 = d[3]; w4 = d[4]; w5 = d[5];
#else
        w6_half_first(d[0], d[1], d[2], d[3], d[4], five, w0, w1, w2);
        w6_half_second(d[1], d[2], d[3], d[4], d[5], five, w3, w4, w5);
#endif
#if W6S_ABL & 256
        asm volatile("" :: "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5));
#else
        dst[0 * 2 * AP] = w0; dst[1 * 2 * AP] = w1; dst[2 * 2 * AP] = w2;
        dst[3 * 2 * AP] = w3; dst[4 * 2 * AP] = w4; dst[5 * 2 * AP] = w5;
#endif
    };

    // ---- pass 2 tasks of this wave ------------------------------------------------------------------------------------------------
    // A task = (grid row a, half, channel pair h): lane (hh, li) = (channel quad, tile) reads the 5 W samples the half needs, forms 3
    // frequencies of 2 channels and writes them to V.  24 tasks, 3 per wave:
    //   X (both pairs h = 0, 1):  waves 0-3: (a = wave, first half);   waves 4-7: (a = 4 + (wave & 3) / 2, half = wave & 1)
    //   Y (pair h = wave / 4):    (a = wave & 3, second half)
    int m_img, m_t, m_tc;
    G::tile_of(min(li, G::TILES - 1), m_img, m_t, m_tc);      // GEO 0: rows 30, 31 carry no tile; they redo tile 29 into their own (unused) V rows
    const int aX = wave < 4 ? wave : 4 + (g >> 1), halfX = wave < 4 ? 0 : (g & 1);
    const int aY = g, hY = ct;
    const f32x2* const wl2 = reinterpret_cast<const f32x2*>(sW + w6_slot<GEO>(hh, (GEO == 1 ? 4 * m_img : 0) + m_t, 0, 4 * m_tc));
    // with A = row + (second half ? K : 0) both halves read A[0], A[K], A[2K], A[1] and one more sample, A[3K] or A[1 - K] (slots of 16 B)
    const f32x2* const xa = wl2 + (aX * AP + (halfX ? K : 0)) * 2;
    const f32x2* const x3 = wl2 + (aX * AP + (halfX ? 1 : 3 * K)) * 2;
    const f32x2* const ya = wl2 + (aY * AP + K) * 2 + hY;
    const f32x2* const y3 = wl2 + (aY * AP + 1) * 2 + hY;
    f32x2* const vX = reinterpret_cast<f32x2*>(sV + (aX * 6 + 3 * halfX) * 64 + lane);
    f32x2* const vY = reinterpret_cast<f32x2*>(sV + (aY * 6 + 3) * 64 + lane) + hY;
    struct X5 { f32x2 x0, x1, x2, x3, x4; };
    auto rd5 = [&](const f32x2* pa, const f32x2* p3) {
        X5 x;
#if W6S_ABL & 1024
        x.x0 = x.x1 = x.x2 = x.x3 = x.x4 = f32x2{1.f, 2.f};
        asm volatile("" : "+v"(x.x0), "+v"(x.x1), "+v"(x.x2), "+v"(x.x3), "+v"(x.x4));
#else
        x.x0 = pa[0]; x.x1 = pa[K * 2]; x.x2 = pa[2 * K * 2]; x.x3 = p3[0]; x.x4 = pa[1 * 2];
#endif
        return x;
    };
    auto taskX = [&](const X5& x, f32x2* dst) {        // which half is wave-uniform; only the transform sits in the branch
        f32x2 v0, v1, v2;
#if W6S_ABL & 512
        v0 = x.x0 ; v1 = x.x1; v2 = x.x2; asm volatile("" :: "v"(x.x3), "v"(x.x4));
#else
        if (halfX == 0) { asm volatile("" ::: "memory"); w6_half_first(x.x0, x.x1, x.x2, x.x3, x.x4, five, v0, v1, v2); }
        else            { asm volatile("" ::: "memory"); w6_half_second(x.x0, x.x1, x.x2, x.x3, x.x4, five, v0, v1, v2); }
#endif
#if W6S_ABL & 128
        asm volatile("" :: "v"(v0), "v"(v1), "v"(v2));
#else
        dst[0] = v0; dst[64 * 2] = v1; dst[2 * 64 * 2] = v2;
#endif
    };
    // X with both channel pairs at once: 5 ds_read_b128 (conflict-free by the TP rule, as conv_wino6.hip's b128 reads) and 3 ds_write_b128
    // (a contiguous KiB each) instead of 10 + 6 eight-byte accesses with their 2-way bank conflicts
    struct X5q { f32x4 x0, x1, x2, x3, x4; };
    auto rd5q = [&](const f32x4* pa, const f32x4* p3) {
        X5q x;
        x.x0 = pa[0]; x.x1 = pa[K]; x.x2 = pa[2 * K]; x.x3 = p3[0]; x.x4 = pa[1];
        return x;
    };
    auto taskXq = [&](const X5q& x, f32x4* dst) {
        f32x2 a0, a1, a2, b0, b1, b2;
        const f32x2 l0 = {x.x0.x, x.x0.y}, l1 = {x.x1.x, x.x1.y}, l2 = {x.x2.x, x.x2.y}, l3 = {x.x3.x, x.x3.y}, l4 = {x.x4.x, x.x4.y};
        const f32x2 h0 = {x.x0.z, x.x0.w}, h1 = {x.x1.z, x.x1.w}, h2 = {x.x2.z, x.x2.w}, h3 = {x.x3.z, x.x3.w}, h4 = {x.x4.z, x.x4.w};
        if (halfX == 0) {
            asm volatile("" ::: "memory");
            w6_half_first(l0, l1, l2, l3, l4, five, a0, a1, a2);
            w6_half_first(h0, h1, h2, h3, h4, five, b0, b1, b2);
        } else {
            asm volatile("" ::: "memory");
            w6_half_second(l0, l1, l2, l3, l4, five, a0, a1, a2);
            w6_half_second(h0, h1, h2, h3, h4, five, b0, b1, b2);
        }
        dst[0] = f32x4{a0.x, a0.y, b0.x, b0.y}; dst[64] = f32x4{a1.x, a1.y, b1.x, b1.y}; dst[2 * 64] = f32x4{a2.x, a2.y, b2.x, b2.y};
    };
    auto taskY = [&](const X5& x, f32x2* dst) {
        f32x2 v0, v1, v2;
#if W6S_ABL & 512
        v0 = x.x0 ; v1 = x.x1; v2 = x.x2; asm volatile("" :: "v"(x.x3), "v"(x.x4));
#else
        w6_half_second(x.x0, x.x1, x.x2, x.x3, x.x4, five, v0, v1, v2);
#endif
#if W6S_ABL & 128
        asm volatile("" :: "v"(v0), "v"(v1), "v"(v2));
#else
        dst[0] = v0; dst[64 * 2] = v1; dst[2 * 64 * 2] = v2;
#endif
    };

    // ---- MFMA side -------------------------------------------------------------------------------------------------------------
    // A operand row li = tile, lane half hh = channel quad; accumulator register r of lane half hh is tile (r & 3) + 8*(r >> 2) + 4*hh, column
    // li = output channel co0 + li.  Slot k < 6: frequency (g, k); k >= 6: (4 + g/2, 3*(g & 1) + k - 6) — the ownership of conv_wino6.hip.
    const f32x4* const vA = sV + (g * 6) * 64 + lane;
    const f32x4* const vB = sV + ((4 + (g >> 1)) * 6 + 3 * (g & 1)) * 64 + lane;
    // U image: [chunk][cout tile][g][9 slots][lane 64][4 floats] (ops.pack_wino6_weight)
    const float* u_wave = P.w + ((long)min(t32, ntiles32 - 1) * 4 + g) * (9 * 256);       // wave-uniform: scalar base, lane * 16 B as offset
    const long u_chunk = (long)ntiles32 * (36 * 256);
    const int u_lane_off = lane * 4;
    f32x4 ub[9];
    auto load_U1 = [&](int chunk, int k) {
#if W6S_ABL & 4
        if (chunk > 0) return;
#endif
        ub[k] = *reinterpret_cast<const f32x4*>(u_wave + chunk * u_chunk + (u_lane_off + k * 256));
    };

#ifdef W6S_TRACE
    // instrumented build (tools/ab/trace_wino6s.py): lane 0 of every wave of every 8th workgroup stamps the shader clock into a.ws
    unsigned long long* trc = (a.ws && (blockIdx.x % 8) == 0 && lane == 0) ? reinterpret_cast<unsigned long long*>(a.ws) + ((blockIdx.x / 8) * 8 + wave) * 64 : nullptr;
#define W6S_STAMP(slot) do { if (trc) trc[slot] = __builtin_readcyclecounter(); } while (0)
#define W6S_STAMP_P(cc, k) do { if ((cc) < 9) W6S_STAMP(3 + 4 * (cc) + (k)); } while (0)
    if (trc) trc[63] = __builtin_amdgcn_s_memrealtime();
#else
#define W6S_STAMP(slot) do { } while (0)
#define W6S_STAMP_P(cc, k) do { } while (0)
#endif
    W6S_STAMP(0);
    if (GEO == 1) {
        // halo columns 15..17 (image columns >= 14) are zero for every image this geometry accepts: their W slots are cleared once, in both
        // buffers, and pass 1 never touches them
        for (int i = tid; i < 2 * 2 * G::RG * 6 * 3; i += 512) {
            const int c3 = i % 3, rest = i / 3;
            const int a6 = rest % 6, r2 = rest / 6;
            const int r = r2 % G::RG, qb = r2 / G::RG;          // qb = buffer * 2 + quad
            sW[(qb >> 1) * WB + w6_slot<GEO>(qb & 1, r, a6, 15 + c3)] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- prologue ----------------------------------------------------------------------------------------------------------------
    // the halos of chunks 0, 1, 2 and the first weights are requested together: one memory round trip.  Then W(0); barrier; V(0) and W(1);
    // the loop's first barrier.
    {
        load_D(0);
        f32x2 d0[6], sc0 = in_sc, sh0 = in_sh;
#pragma unroll
        for (int i = 0; i < 6; ++i) d0[i] = d[i];
        load_D(min(1, nchunks - 1));
        f32x2 d1[6], sc1 = in_sc, sh1 = in_sh;
#pragma unroll
        for (int i = 0; i < 6; ++i) d1[i] = d[i];
        load_D(min(2, nchunks - 1));
        f32x2 d2[6], sc2 = in_sc, sh2 = in_sh;
#pragma unroll
        for (int i = 0; i < 6; ++i) d2[i] = d[i];
        if (act) {
#pragma unroll
            for (int k = 0; k < 9; ++k) load_U1(0, k);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = d0[i];
        in_sc = sc0; in_sh = sh0;
        pass1(wwr2);                                      // W(0) -> buffer 0
        __syncthreads();
        {
            X5 x = rd5(xa, x3);
            taskX(x, vX);
            x = rd5(xa + 1, x3 + 1);
            taskX(x, vX + 1);
            x = rd5(ya, y3);
            taskY(x, vY);                                 // V(0) -> buffer 0
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = d1[i];
        in_sc = sc1; in_sh = sh1;
        pass1(wwr2 + WB * 2);                             // W(1) -> buffer 1
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = d2[i];
        in_sc = sc2; in_sh = sh2;
    }

    f32x16 acc[9];
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    // ---- main loop: one period per 8-channel chunk, one barrier per period ------------------------------------------------------------
    //   in period c:  the MFMAs read V(c) [buffer c & 1];  pass 2 turns W(c+1) [buffer (c+1) & 1] into V(c+1);  pass 1 turns the halo of
    //   chunk c+2 (in registers since period c-1) into W(c+2) [buffer c & 1];  the halo of chunk c+3 and the weights of chunk c+1 are requested.
    //   Past the last chunk the transforms run on repeated data and nobody reads their output.
    auto mm4 = [&](const f32x4 va, const int k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[j], ub[k][j], acc[k], 0, 0, 0);
    };
    auto period = [&](const int c, auto parity, auto ACT) {
        constexpr int par = decltype(parity)::value;
#if W6S_ABL & 16
        constexpr bool A = false;
#else
        constexpr bool A = decltype(ACT)::value;
#endif
        constexpr int vcur = par * VS;                       // slots
        constexpr int vnxt2 = (1 - par) * VS * 2;            // f32x2 units
        constexpr int wrd2 = (1 - par) * WB * 2;
        constexpr int wwr = par * WB * 2;
        const int cn = min(c + 1, nchunks - 1);
#if !(W6S_ABL & 32)
        __syncthreads();
#endif
        f32x4 va0, va1, va2;
#if W6S_ABL & 64
#define W6S_VRD(dst, src) dst = f32x4{1.f, 2.f, 3.f, 4.f}
#else
#define W6S_VRD(dst, src) dst = src
#endif
        if (A) { W6S_VRD(va0, vA[vcur + 0 * 64]); W6S_VRD(va1, vA[vcur + 1 * 64]); W6S_VRD(va2, vA[vcur + 2 * 64]); }
#if W6S_ABL & 8
#define W6S_P2(stmt)
        X5 x;
#else
#define W6S_P2(stmt) stmt
        X5 x = rd5(xa + wrd2, x3 + wrd2);
#endif
        W6S_FENCE;
#if !(W6S_ABL & 1)
        pass1(wwr2 + wwr);
#endif
#if !(W6S_ABL & 2)
        load_D(min(c + 3, nchunks - 1));
#endif
        W6S_FENCE;
        if (A) { mm4(va0, 0); load_U1(cn, 0); W6S_VRD(va0, vA[vcur + 3 * 64]); }
        W6S_FENCE;
        W6S_P2(taskX(x, vX + vnxt2));
        W6S_P2(x = rd5(xa + wrd2 + 1, x3 + wrd2 + 1));
        W6S_FENCE;
        if (A) { mm4(va1, 1); load_U1(cn, 1); W6S_VRD(va1, vA[vcur + 4 * 64]); }
        W6S_FENCE;
        if (A) { mm4(va2, 2); load_U1(cn, 2); W6S_VRD(va2, vA[vcur + 5 * 64]); }
        W6S_FENCE;
        W6S_P2(taskX(x, vX + vnxt2 + 1));
        W6S_P2(x = rd5(ya + wrd2, y3 + wrd2));
        W6S_FENCE;
        if (A) { mm4(va0, 3); load_U1(cn, 3); W6S_VRD(va0, vB[vcur + 0 * 64]); }
        W6S_FENCE;
        if (A) { mm4(va1, 4); load_U1(cn, 4); W6S_VRD(va1, vB[vcur + 1 * 64]); }
        W6S_FENCE;
        W6S_P2(taskY(x, vY + vnxt2));
        W6S_FENCE;
        if (A) { mm4(va2, 5); load_U1(cn, 5); W6S_VRD(va2, vB[vcur + 2 * 64]); }
        W6S_FENCE;
        if (A) { mm4(va0, 6); load_U1(cn, 6); }
        W6S_FENCE;
        if (A) { mm4(va1, 7); load_U1(cn, 7); }
        W6S_FENCE;
        if (A) { mm4(va2, 8); load_U1(cn, 8); }
    };
    // Staggered schedule (W6S_SCHED 1, the default): the two waves of a SIMD (wave w and w + 4, i.e. the two cout tiles) would otherwise run
    // the same phases at the same time — both transforming, the matrix pipe idle; both issuing MFMAs, queueing.  Between two barriers waves
    // 0-3 do their transforms FIRST and then their MFMAs, waves 4-7 issue their MFMAs first and transform LAST: one wave of a SIMD always
    // has MFMAs to issue.  There is ONE copy of the code: the instruction stream is the cycle  T(c) M(c) T(c+1) M(c+1) ...  and only the
    // place of the barrier in it differs — waves 0-3:  B T(c) M(c) B ...,  waves 4-7:  B M(c) T(c) B ...  = the same loop body
    // [T(c); M(c + ct)] with the barrier after M (ct = 0) or between T and M (ct = 1), one extra M(0) in front for ct = 1.  (Two copies of
    // the transforms under complementary wave-uniform branches made the register allocator spill 246 registers.)  Three accumulators are
    // walked round-robin so that consecutive MFMAs never depend on each other.
    auto mm12 = [&](const f32x4 a0, const f32x4 a1, const f32x4 a2, const int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[k0 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], ub[k0 + 0][j], acc[k0 + 0], 0, 0, 0);
            acc[k0 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], ub[k0 + 1][j], acc[k0 + 1], 0, 0, 0);
            acc[k0 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j], ub[k0 + 2][j], acc[k0 + 2], 0, 0, 0);
        }
    };
    // T(c): pass 2 W(c+1) -> V(c+1), pass 1 halo(c+2) -> W(c+2), request halo(c+3);  par = c & 1
    auto T = [&](const int c, auto parity) {
        constexpr int par = decltype(parity)::value;
        constexpr int vnxt2 = (1 - par) * VS * 2;            // f32x2 units
        constexpr int wrd2 = (1 - par) * WB * 2;
        constexpr int wwr = par * WB * 2;
#if W6S_PRIO
        __builtin_amdgcn_s_setprio(W6S_PRIO);             // the transform phase competes with the SIMD partner's MFMA stream for issue slots
#endif
#if !(W6S_ABL & 8)
#ifdef W6S_X64
        X5 x0 = rd5(xa + wrd2, x3 + wrd2);
        X5 x1 = rd5(xa + wrd2 + 1, x3 + wrd2 + 1);
#else
        X5q xq = rd5q(reinterpret_cast<const f32x4*>(xa + wrd2), reinterpret_cast<const f32x4*>(x3 + wrd2));
#endif
        X5 xy = rd5(ya + wrd2, y3 + wrd2);
#endif
#if !(W6S_ABL & 1)
        pass1(wwr2 + wwr);
#endif
#if !(W6S_ABL & 2)
        load_D(min(c + 3, nchunks - 1));
#endif
#if !(W6S_ABL & 8)
#ifdef W6S_X64
        taskX(x0, vX + vnxt2);
        taskX(x1, vX + vnxt2 + 1);
#else
        taskXq(xq, reinterpret_cast<f32x4*>(vX + vnxt2));
#endif
        taskY(xy, vY + vnxt2);
#endif
#if W6S_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    // M(cw): the 36 MFMAs of chunk cw on V(cw) (pa, pb: the wave's row-A / row-B operand pointers into that V buffer), weights of chunk cw + 1 requested
    auto M = [&](const int cw, const f32x4* pa, const f32x4* pb) {
        const int cn = min(cw + 1, nchunks - 1);
        f32x4 va0, va1, va2, vb0, vb1, vb2;
        W6S_VRD(va0, pa[0 * 64]); W6S_VRD(va1, pa[1 * 64]); W6S_VRD(va2, pa[2 * 64]);
        W6S_VRD(vb0, pa[3 * 64]); W6S_VRD(vb1, pa[4 * 64]); W6S_VRD(vb2, pa[5 * 64]);
        W6S_FENCE;
        mm12(va0, va1, va2, 0);
        load_U1(cn, 0); load_U1(cn, 1); load_U1(cn, 2);
        W6S_VRD(va0, pb[0 * 64]); W6S_VRD(va1, pb[1 * 64]); W6S_VRD(va2, pb[2 * 64]);
        W6S_FENCE;
        mm12(vb0, vb1, vb2, 3);
        load_U1(cn, 3); load_U1(cn, 4); load_U1(cn, 5);
        W6S_FENCE;
        mm12(va0, va1, va2, 6);
        load_U1(cn, 6); load_U1(cn, 7); load_U1(cn, 8);
        W6S_FENCE;
    };
#if W6S_ABL & 16
    const bool do_m = false;
#else
    const bool do_m = act;
#endif
#ifndef W6S_SCHED
#define W6S_SCHED 1
#endif
#if W6S_SCHED == 1
    // Every path through the loop body issues the same memory operations in the same order — the M phase is unconditional inside the loop
    // (waves without a cout tile run their own copy of the loop, and the one M a ct = 1 wave must not run is peeled off with the last
    // trip): the compiler's s_waitcnt vmcnt(N) in front of pass 1 is the minimum over the paths, and with a skippable M between the halo
    // request and its use it was vmcnt(0) — every period then waited for the nine weight loads it had just issued (1400 cycles in the trace).
    auto loop = [&](auto DOM) {
        constexpr bool DM = decltype(DOM)::value;
        // operand pointers of the MFMA phase of an even / odd trip: a wave of ct = 1 works one chunk ahead of the loop counter
        const f32x4* const vA_e = vA + ct * VS;
        const f32x4* const vB_e = vB + ct * VS;
        const f32x4* const vA_o = vA + (1 - ct) * VS;
        const f32x4* const vB_o = vB + (1 - ct) * VS;
        W6S_STAMP(1);
        __syncthreads();                                  // V(0) and W(1) complete
        W6S_STAMP(2);
        if (DM && ct != 0) M(0, vA, vB);
        auto trip = [&](const int c, auto LAST) {
            constexpr bool last = decltype(LAST)::value;
            W6S_STAMP_P(c, 0);
            T(c, std::integral_constant<int, 0>{});
            W6S_STAMP_P(c, 1);
#if !(W6S_ABL & 32)
            if (ct != 0) __syncthreads();
#endif
            W6S_STAMP_P(c, 2);
            if (DM) M(c + ct, vA_e, vB_e);
            W6S_STAMP_P(c, 3);
#if !(W6S_ABL & 32)
            if (ct == 0) __syncthreads();
#endif
            W6S_STAMP_P(c + 1, 0);
            T(c + 1, std::integral_constant<int, 1>{});
            W6S_STAMP_P(c + 1, 1);
#if !(W6S_ABL & 32)
            if (ct != 0) __syncthreads();
#endif
            W6S_STAMP_P(c + 1, 2);
            if (DM) {
                if (!last) M(c + 1 + ct, vA_o, vB_o);
                else if (ct == 0) M(c + 1, vA_o, vB_o);
            }
            W6S_STAMP_P(c + 1, 3);
#if !(W6S_ABL & 32)
            if (ct == 0) __syncthreads();
#endif
        };
        int c = 0;
        for (; c + 2 < nchunks; c += 2) trip(c, std::false_type{});        // nchunks is even: Cin is a multiple of 16 (validate)
        trip(c, std::true_type{});
    };
    if (do_m) loop(std::true_type{});
    else loop(std::false_type{});
#else
    if (act) {
        for (int c = 0; c < nchunks; c += 2) {
            period(c, std::integral_constant<int, 0>{}, std::true_type{});
            period(c + 1, std::integral_constant<int, 1>{}, std::true_type{});
        }
    } else {
        for (int c = 0; c < nchunks; c += 2) {
            period(c, std::integral_constant<int, 0>{}, std::false_type{});
            period(c + 1, std::integral_constant<int, 1>{}, std::false_type{});
        }
    }
#endif

    // ---- epilogue (conv_wino6.hip's, per cout tile: wave group ct works in its own 64 KiB of the exchange area) ----------------------
    // A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].  Row pass in registers: per accumulator entry the wave's 6 + 3
    // frequencies become P[rowA][0..3] and the partial P[rowB][0..3] of its half.  Wave d of a group finishes the tiles of accumulator
    // registers 4d..4d+3: in round q the other waves of the group send it the 8 values of registers 4d+2q, 4d+2q+1.
    const int halfB = g & 1;
    const int co = co0 + li;
    const bool cvalid = act && co < a.Cout;
    float sc = P.scale[min(co, a.Cout - 1)];
    float sh = P.shift[min(co, a.Cout - 1)];
    W6S_STAMP(42);
    __syncthreads();
    W6S_STAMP(43);
    asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(sc), "+v"(sh) : : "memory");
    f32x2* ex2 = reinterpret_cast<f32x2*>(smem) + ct * (64 * 1024 / 8);        // exchange: [src wave][dst wave][value 0..7][lane] pairs = 64 KiB
    const float lo = co < a.relu_upto ? 0.f : __builtin_nanf("");      // max(v, NaN) = v: lanes without the ReLU
    const f32x2 sc2 = {sc, sc}, sh2 = {sh, sh};
    auto fma2 = [](f32x2 x, float k, f32x2 y) { return __builtin_elementwise_fma(x, f32x2{k, k}, y); };
    const bool want_stats = a.gn_ws != nullptr;
    f32x2 gs2 = {0.f, 0.f}, gss2 = {0.f, 0.f};
    float gs = 0.f, gss = 0.f;
    float* yimg = P.y + (long)n * H * W * a.y_cs + a.y_co + co;
    unsigned long long ybase_s;
    {
        const unsigned long long yb = (unsigned long long)(P.y + (long)n * H * W * a.y_cs);
        ybase_s = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(yb >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)yb);
    }
    const unsigned long long px_b = (unsigned long long)a.y_cs * 4u, rowskip_b = (unsigned long long)(W - 3) * a.y_cs * 4u;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        f32x2 own[8];
        if (act) {
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const int r0 = 4 * dd + 2 * q;
                f32x2 v[8];
                {
                    const f32x2 m0 = {acc[0][r0], acc[0][r0 + 1]}, m1 = {acc[1][r0], acc[1][r0 + 1]}, m2 = {acc[2][r0], acc[2][r0 + 1]},
                                m3 = {acc[3][r0], acc[3][r0 + 1]}, m4 = {acc[4][r0], acc[4][r0 + 1]}, m5 = {acc[5][r0], acc[5][r0 + 1]};
                    const f32x2 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                    v[0] = m0 + s1 + s2;
                    v[1] = fma2(d2, 2.0f, d1);
                    v[2] = fma2(s2, 4.0f, s1);
                    v[3] = fma2(d2, 8.0f, d1) + m5;
                    const f32x2 n0 = {acc[6][r0], acc[6][r0 + 1]}, n1 = {acc[7][r0], acc[7][r0 + 1]}, n2 = {acc[8][r0], acc[8][r0 + 1]};
                    if (halfB == 0) {      // b = 0, 1, 2
                        const f32x2 t1 = n1 + n2, e1 = n1 - n2;
                        v[4] = n0 + t1; v[5] = e1; v[6] = t1; v[7] = e1;
                    } else {               // b = 3, 4, 5
                        const f32x2 t2s = n0 + n1, e2 = n0 - n1;
                        v[4] = t2s; v[5] = e2 + e2; v[6] = t2s * 4.0f; v[7] = fma2(e2, 8.0f, n2);
                    }
                }
                if (dd == g) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) own[k] = v[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) ex2[(((g * 4 + dd) * 8 + k) << 6) + lane] = v[k];
                }
            }
        }
        __syncthreads();
        if (act) {
            // P[a][j]: rows 0..3 from waves 0..3 (values 0..3), row 4 = halves of waves 0, 1, row 5 = halves of waves 2, 3 (values 4..7)
            f32x2 Pm[6][4];
            {
                f32x2 part[4][4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    f32x2 v[8];
                    if (s == g) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = own[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = ex2[(((s * 4 + g) * 8 + k) << 6) + lane];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { Pm[s][j] = v[j]; part[s][j] = v[4 + j]; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { Pm[4][j] = part[0][j] + part[1][j]; Pm[5][j] = part[2][j] + part[3][j]; }
            }
            f32x2 yv[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 s1 = Pm[1][j] + Pm[2][j], d1 = Pm[1][j] - Pm[2][j], s2 = Pm[3][j] + Pm[4][j], d2 = Pm[3][j] - Pm[4][j];
                f32x2 y[4];
                y[0] = Pm[0][j] + s1 + s2;
                y[1] = fma2(d2, 2.0f, d1);
                y[2] = fma2(s2, 4.0f, s1);
                y[3] = fma2(d2, 8.0f, d1) + Pm[5][j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x2 t = __builtin_elementwise_fma(y[i], sc2, sh2);
                    t.x = fmaxf(t.x, lo);
                    t.y = fmaxf(t.y, lo);
                    yv[i][j] = t;
                }
            }
            // the pair's entries are accumulator registers 4*g + 2q, +1 of lane half hh: tiles m, m + 1
            bool full[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int m = 2 * q + rr + 8 * g + 4 * hh;
                int mimg, mt, mtc;
                G::tile_of(m, mimg, mt, mtc);
                const int oh = oh0 + 4 * mt, ow = ow0 + 4 * mtc;
                const bool tile_ok = cvalid && m < G::TILES && n + mimg < P.N;
                full[rr] = tile_ok && oh + 4 <= H && ow + 4 <= W;
                if (full[rr]) {                             // interior tile: 16 stores, no per-store predicate, no vector address arithmetic
                    const unsigned voff = (unsigned)((((mimg * H + oh) * W + ow) * a.y_cs + a.y_co + co) * 4);
                    unsigned long long sp = ybase_s;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float val = rr ? yv[i][j].y : yv[i][j].x;
                            // nt: the output is a stream (84 MB per launch at stage 2) that must not push the weights and the halo lines this launch re-reads
                        // out of L2; measured -2.2 % on the map shapes, +0.4 % end to end (profiles/r03_ablations.txt), sc0 / sc1 nothing
                        asm volatile("global_store_dword %1, %2, %0 nt" : "+s"(sp) : "v"(voff), "v"(val) : "memory");
                            sp += j == 3 ? rowskip_b : px_b;
                        }
                } else if (tile_ok) {
                    float* yp0 = yimg + (((long)mimg * H + oh) * W + ow) * a.y_cs;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (oh + i < H && ow + j < W) {
                                const float val = rr ? yv[i][j].y : yv[i][j].x;
                                yp0[((long)i * W + j) * a.y_cs] = val;
                                gs += val;
                                gss = fmaf(val, val, gss);
                            }
                }
            }
            if (want_stats) {                               // whole tiles: packed, masked by tile
                const f32x2 mask = {full[0] ? 1.f : 0.f, full[1] ? 1.f : 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2 t = yv[i][j] * mask;
                        gs2 += t;
                        gss2 = __builtin_elementwise_fma(t, yv[i][j], gss2);
                    }
            }
        }
        if (q == 0) __syncthreads();                        // the exchange buffer is reused by round 1
    }
    W6S_STAMP(49);
#ifdef W6S_TRACE
    if (trc) trc[62] = __builtin_amdgcn_s_memrealtime();
#endif
    gs += gs2.x + gs2.y;
    gss += gss2.x + gss2.y;
    // fused GroupNorm statistics of the NEXT layer's normalisation (fcos.py:182-186): one {sum, sumsq} record per
    // (spatial tile, wave g of the group, group of channels) — the record layout of conv_wino6.hip
    if (a.gn_ws && act) {
        for (int o = 1; o < a.gn_cpg; o <<= 1) { gs += __shfl_xor(gs, o); gss += __shfl_xor(gss, o); }
        gs += __shfl_xor(gs, 32);
        gss += __shfl_xor(gss, 32);
        if (cvalid && hh == 0 && (li & (a.gn_cpg - 1)) == 0) {
            double* o = a.gn_ws + (((long)bx * 4 + g) * a.gn_groups + co / a.gn_cpg) * 2;
            o[0] = (double)gs;
            o[1] = (double)gss;
        }
    }
}

template <int GEO>
static int launch_wino6s_geo(ConvArgs& a, hipStream_t st) {
    static DeviceOnce once;
    int rc = once.run([]() {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino6s_kernel<false, GEO>), hipFuncAttributeMaxDynamicSharedMemorySize, W6S<GEO>::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino6s_kernel<true, GEO>), hipFuncAttributeMaxDynamicSharedMemorySize, W6S<GEO>::LDS_BYTES);
        return e == hipSuccess ? CMK_OK : fail(CMK_ELAUNCH, "conv_wino6s: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    });
    if (rc) return rc;
    int blocks = 0;
    for (int i = 0; i < a.nprob; ++i) {
        ConvProblem& p = a.p[i];
        p.tile_begin = blocks;
        if (GEO == 0) {
            p.tiles_h = cdiv(p.Ho, W6G<0>::OH);
            p.tiles_w = cdiv(p.Wo, W6G<0>::OW);
            blocks += p.N * p.tiles_h * p.tiles_w;
        } else {
            p.tiles_h = p.tiles_w = 1;
            blocks += cdiv(p.N, 2);
        }
    }
    a.grid_y = cdiv(cdiv(a.Cout, 32), 2);
    a.total_tiles = blocks;
    if (a.ksplit > 1) return fail(CMK_EINVAL, "conv_wino6s: split-K is a feature of the 32-cout form (tune_sc 16)%s", "");
    const dim3 grid(((blocks + 7) / 8) * 8 * a.grid_y);
    if (a.p[0].in_scale)
        hipLaunchKernelGGL((conv_wino6s_kernel<true, GEO>), grid, dim3(512), W6S<GEO>::LDS_BYTES, st, a);
    else
        hipLaunchKernelGGL((conv_wino6s_kernel<false, GEO>), grid, dim3(512), W6S<GEO>::LDS_BYTES, st, a);
    return check_launch("conv_wino6s");
}

// geo 0: 12x40-pixel tiles of one image; geo 1: pairs of whole maps of at most 16 rows x 14 columns (one problem, no fused GN statistics)
int launch_wino6s(ConvArgs& a, int geo, hipStream_t st) {
    for (int i = 0; i < a.nprob; ++i)       // the epilogue's stores take a 32-bit byte offset inside the output image (GEO 1: inside a pair of images)
        if ((long)(geo == 0 ? 1 : 2) * a.p[i].H * a.p[i].W * a.y_cs * 4 >= (1L << 32))
            return fail(CMK_EINVAL, "conv_wino6s: an output image of 4 GiB or more%s", "");
    if (geo == 0) return launch_wino6s_geo<0>(a, st);
    if (a.nprob != 1 || a.p[0].H > 16 || a.p[0].W > 14 || a.gn_ws)
        return fail(CMK_EINVAL, "conv_wino6s: the RoI-pair geometry takes one problem of maps up to 16x14 and produces no GroupNorm statistics%s", "");
    return launch_wino6s_geo<1>(a, st);
}

}  // namespace cmk
