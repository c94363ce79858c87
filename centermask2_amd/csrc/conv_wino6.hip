// Winograd F(4x4, 3x3) for 3x3 stride-1 convs, fused in one kernel, plain fp32 on the CDNA4 matrix pipe.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A    per 6x6 input patch d -> 4x4 outputs, summed over input channels: 36 multiplies per
//   16 outputs = 2.25 per output against 9 (direct) and 4 (F(2x2,3x3), conv_wino4r_kernel) — 1.78x fewer MFMAs than the 2x2 form.
//   Interpolation points 0, +-1, +-2, inf (Lavin & Gray); every transform is exact-coefficient fp32 arithmetic, U = G g G^T is
//   computed on the host in fp64 and rounded once.  Measured error of this form on the full model (tools/wino_numerics.py, CPU
//   emulation of exactly this arithmetic): features/logits rms 1.6e-6 relative (F(2x2): 1.0e-6), labels and locations identical.
//
// Shape of the kernel (what differs from conv_wino4r_kernel, and why):
//   * workgroup = 3 x 10 tiles of 4x4 outputs (12 x 40 pixels: every map width of the model is a multiple of 40) x 32 output
//     channels, 4 waves, two workgroups per CU.  30 of the 32 MFMA rows carry a tile.  The 36 frequencies of the 6x6 grid are 36
//     accumulators of 32 tiles x 32 couts; a wave owns 9: grid row a = wave (6 frequencies) and half of row 4 or 5 (3 frequencies).
//     144 accumulator VGPRs per lane — the register budget shapes everything else.
//   * the input transform is split.  B^T d B = (column pass) then (row pass):
//       pass 1 (once per workgroup and 8-channel chunk): thread = (tile row t, halo column, channel quad) — 3 x 42 x 2 = 252 items —
//         loads the 6 input rows of its item straight from global memory into registers one period ahead (buffer loads: rows and
//         columns outside the image come back as 0 from the hardware range check, no masks), forms W[t][a][col] = sum_i B^T[a][i] d[4t+i][col]
//         in place and writes the W image to LDS (28 KiB per chunk, two buffers).  The column pass is shared by horizontally adjacent
//         tiles, and there is no raw-halo stage in LDS at all;
//       pass 2 (by the MFMA waves, in registers, right in front of the MFMAs): a lane reads the 5 W values of its tile that one half of
//         a frequency row needs (conflict-free image, see w6_slot) and forms 3 frequencies with 6 FMAs per channel.
//     So the 36-plane V image (74 KiB per 16 channels — it would not fit twice beside a second workgroup) never exists, LDS traffic per
//     MFMA is a fraction of the 2x2 kernel's, and ONE barrier per chunk (36 MFMAs per wave) is enough — it never waits for memory.
//   * VALU instructions are the currency: fp32 MFMA and VALU do not co-execute here and a period costs 2 waves x (36 MFMAs x 64 +
//     N_valu x ~8) cycles.  A wave issues 60 packed transform instructions + 4 others per period (hand-written v_pk_* blocks, row-B sample
//     addresses set up once, two periods per trip so that the W buffer is a compile-time offset); the fused-affine variant adds 48.
//   * weights never touch LDS: U is packed so that every operand load of a wave is one contiguous KiB ([chunk][cout tile][wave][9][lane][4])
//     and is fetched two steps (24 MFMAs) ahead into registers, as in the 2x2 kernel.
//   * epilogue: each wave reduces its frequencies along b in registers (6 -> 4 and 3 -> 4 partial values per entry), the four waves swap
//     those through LDS in two rounds, and wave w finishes the 8 tiles of accumulator registers 4w..4w+3: column pass, scale/shift/ReLU,
//     NHWC stores, GroupNorm statistics — on pairs of accumulator registers with packed fp32, interior tiles stored through a scalar-walked
//     base + one lane offset per tile (~460 VALU instructions per wave; the scalar form took ~1080 and 12.4 us beside a partner's MFMAs).
//
// Reference call sites replaced: the same 3x3 stride-1 convs as the 2x2 kernel (vovnet.py:205-219, d2 FPN outputs, fcos.py:169-200,
// sam.py:58-70, maskiou_head.py:81-88).
#include <type_traits>

#include "wino6_common.hpp"

#ifndef W6_TRACE_EVERY
#define W6_TRACE_EVERY 16
#endif
#ifndef W6_ABL
#define W6_ABL 0     // timing ablations (tools/ab): 1 no pass 1, 2 no halo loads, 4 no weight loads, 8 no barrier, 16 no W sample reads — results are wrong with any set
#endif
#ifndef W6_NOFENCE
#define W6_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define W6_FENCE
#endif

namespace cmk {

constexpr int W6_EX_FLOATS = 4 * 4 * 2 * 8 * 64;       // epilogue exchange: [src wave][dst wave][value][lane][register pair of the round] = 64 KiB
template <int GEO> constexpr int w6_lds_bytes() { return (2 * W6G<GEO>::WB * 16 > W6_EX_FLOATS * 4) ? 2 * W6G<GEO>::WB * 16 : W6_EX_FLOATS * 4; }
#ifdef W6_ONE_WG     // experiment: one workgroup per CU (how fast is a workgroup without a partner?)
template <int GEO> constexpr int w6_lds_alloc() { return 96 * 1024; }
#else
template <int GEO> constexpr int w6_lds_alloc() { return w6_lds_bytes<GEO>(); }
static_assert(2 * w6_lds_bytes<0>() <= LDS_CU && 2 * w6_lds_bytes<1>() <= LDS_CU, "two workgroups per CU");
#endif

// AFF: the producer's GroupNorm+ReLU is applied to the input in pass 1 (FCOS tower convs 2-4 and the predictors)
template <bool AFF, int GEO>
__global__ __launch_bounds__(256, 2) void conv_wino6_kernel(const ConvArgs a) {
    using G = W6G<GEO>;
    constexpr int W6_AP = G::AP, W6_WB = G::WB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* sW = reinterpret_cast<f32x4*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, li = lane & 31;

    // XCD-aware order, as in the other conv kernels: the grid_y cout tiles of one spatial tile go to the same XCD, back to back
    const int xq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    // (the opposite, weight-stationary mapping — XCD k takes the cout tiles == k mod 8 of every spatial tile so that its L2 holds 1/8 of
    // U — was measured 3-4 % slower on the 256 -> 256 layers: every XCD then reads the whole input)
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
    const int co0 = by * 32;
    // split-K (a.ksplit > 1, blockIdx.y): this workgroup owns the 8-channel chunks [c_lo, nchunks) of the conv's Cin / 8 and leaves raw partial
    // sums in a.ws (see the epilogue); for launches of about one round of workgroups whose life is one long chunk loop (the first conv of a
    // stage-4 / stage-5 OSA block: 512..1024 input channels on 50x80 / 25x40 maps) two or four workgroups share that loop
    const int ks = blockIdx.y;
    const int c_lo = (int)((long)ks * (a.Cin >> 3) / a.ksplit);
    const int nchunks = (int)((long)(ks + 1) * (a.Cin >> 3) / a.ksplit);       // END of this workgroup's chunk range (even bounds: host)

    // ---- pass 1 item of this thread ----------------------------------------------------------------------------------------------
    int p_img, p_q, p_t, p_col;
    bool p_active;
    G::item_of(tid, p_img, p_q, p_t, p_col, p_active);
    // buffer resource over the image this thread stages (GEO 1: wave-uniform, waves 0-1 the first image of the pair, waves 2-3 the second;
    // an image index past the batch gets an empty resource: every load returns 0): {base, num_records = bytes of the image, raw dword format}
    i32x4 rsrc;
    {
        const int img_n = n + (GEO == 1 ? (wave >> 1) : 0);
        const unsigned long long base = (unsigned long long)(P.x + (long)min(img_n, P.N - 1) * H * W * a.x_cs);
        rsrc.x = __builtin_amdgcn_readfirstlane((int)(base & 0xffffffffull));
        rsrc.y = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffffull));
        rsrc.z = __builtin_amdgcn_readfirstlane(img_n < P.N ? H * W * a.x_cs * 4 : 0);
        rsrc.w = 0x00020000;
    }
    const int row_bytes = W * a.x_cs * 4;
    const int ih0 = oh0 - 1 + 4 * p_t, iw = ow0 - 1 + p_col;
    // byte offset of input row 0 of the item; rows above/below the image are out of the resource's range by themselves, a column
    // outside the image would alias the neighbouring row, so it is pushed out of range
    const int voff0 = (iw >= 0 && iw < W) ? (ih0 * W + iw) * a.x_cs * 4 + (a.x_co + p_q * 4) * 4 : (int)0x80000000;
    unsigned okm = 0;                                              // AFF only: relu(0*s + b) != 0, so padding needs the mask
    if (AFF) {
#pragma unroll
        for (int i = 0; i < 6; ++i) okm |= ((iw >= 0 && iw < W && ih0 + i >= 0 && ih0 + i < H) ? 1u : 0u) << i;
    }
    // scale/shift of the fused input affine: wave-uniform base (per chunk) + one 32-bit lane offset (two 64-bit lane pointers cost the
    // variant the registers it does not have)
    const unsigned aff_off = AFF ? (unsigned)(min(n + p_img, P.N - 1) * a.Cin + p_q * 4) : 0u;
    const f32x2 five = {5.0f, 5.0f};                   // the one transform coefficient that is not an inline constant: an SGPR pair
    f32x4 d[6];
    f32x4 in_sc = {1.f, 1.f, 1.f, 1.f}, in_sh = {0.f, 0.f, 0.f, 0.f};
    auto load_D = [&](int chunk) {
#pragma unroll
#ifndef W6_HALO_AUX
#define W6_HALO_AUX 0          // cache policy of the halo loads (experiments: 1 sc0, 2 nt, 16 sc1)
#endif
        for (int i = 0; i < 6; ++i) d[i] = w6_buffer_load(rsrc, voff0 + i * row_bytes, chunk * 32, W6_HALO_AUX);
        if (AFF) {
            in_sc = *reinterpret_cast<const f32x4*>(P.in_scale + chunk * 8 + aff_off);
            in_sh = *reinterpret_cast<const f32x4*>(P.in_shift + chunk * 8 + aff_off);
        }
    };
    const int p_dst = w6_slot<GEO>(p_q, (GEO == 1 ? 4 * p_img : 0) + p_t, 0, p_col);
    auto pass1 = [&](f32x4* wbuf) {
        if (AFF) {
            // relu(x * s + b), and 0 for a sample outside the image (relu(0 * s + b) != 0): 2 packed fma per row, then ONE med3 per value —
            // med3(v, 0, +inf) = max(v, 0), med3(v, 0, 0) = 0 — with the row's third operand made from the mask bit
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float kinf = ((okm >> i) & 1u) ? __builtin_inff() : 0.f;
                const f32x4 v = __builtin_elementwise_fma(d[i], in_sc, in_sh);
                d[i] = f32x4{__builtin_amdgcn_fmed3f(v.x, 0.f, kinf), __builtin_amdgcn_fmed3f(v.y, 0.f, kinf),
                             __builtin_amdgcn_fmed3f(v.z, 0.f, kinf), __builtin_amdgcn_fmed3f(v.w, 0.f, kinf)};
            }
        }
        f32x2* dst = reinterpret_cast<f32x2*>(wbuf + p_dst);
        if (GEO == 1 && !p_active) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {                 // channel pairs: 12 packed instructions and 6 ds_write_b64 each
            f32x2 e[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) e[i] = h ? f32x2{d[i].z, d[i].w} : f32x2{d[i].x, d[i].y};
            f32x2 w0, w1, w2, w3, w4, w5;
            w6_half_first(e[0], e[1], e[2], e[3], e[4], five, w0, w1, w2);
            w6_half_second(e[1], e[2], e[3], e[4], e[5], five, w3, w4, w5);
            dst[0 * 2 * W6_AP + h] = w0; dst[1 * 2 * W6_AP + h] = w1; dst[2 * 2 * W6_AP + h] = w2;
            dst[3 * 2 * W6_AP + h] = w3; dst[4 * 2 * W6_AP + h] = w4; dst[5 * 2 * W6_AP + h] = w5;
        }
    };

    // ---- MFMA side -------------------------------------------------------------------------------------------------------------
    // A operand row li = tile m (GEO 0: m = 10*t + tc, rows 30, 31 carry no tile and their accumulator rows are never stored); lane half hh = channel quad; accumulator register r of lane half hh is tile m = (r & 3) + 8*(r >> 2) + 4*hh, column
    // li = output channel co0 + li
    const int rowA = wave, rowB = 4 + (wave >> 1), halfB = wave & 1;
    int m_img, m_t, m_tc;
    G::tile_of(min(li, G::TILES - 1), m_img, m_t, m_tc);     // GEO 0: rows 30, 31 carry no tile; they re-read tile 29's samples (never stored)
    const f32x4* wl = sW + w6_slot<GEO>(hh, (GEO == 1 ? 4 * m_img : 0) + m_t, 0, 4 * m_tc);
    const f32x4* wA = wl + rowA * W6_AP;
    const f32x4* wB = wl + rowB * W6_AP;
    const f32x4* wBa = wB + (halfB ? G::CK : 0);                // see rdB
    const f32x4* wB3 = wB + (halfB ? 1 : 3 * G::CK);
    // U image: [chunk][cout tile][wave][9 slots][lane 64][4 floats]; slot k < 6: frequency (rowA, k); k >= 6: (rowB, 3*halfB + k - 6);
    // lane = 32*hh + li holds channels 8*chunk + 4*hh .. +3 of output channel 32*tile + li
    const float* u_wave = P.w + ((long)by * 4 + wave) * (9 * 256);       // wave-uniform: the loads take it as a scalar base, lane * 16 B as offset
    const long u_chunk = (long)a.grid_y * (36 * 256);
    const int u_lane_off = lane * 4;
    f32x4 ub[3][3];
    auto load_U = [&](int step, int buf) {          // step = chunk*3 + s
#if W6_ABL & 4
        if (step > 2) return;
#endif
        const int c = step / 3, s = step - c * 3;
        const float* src = u_wave + c * u_chunk + s * (3 * 256);
#pragma unroll
        for (int k = 0; k < 3; ++k) ub[buf][k] = *reinterpret_cast<const f32x4*>(src + (u_lane_off + k * 256));
    };
    const int total_steps = nchunks * 3;                        // END of the step range

#ifdef W6_TRACE
    // instrumented build (tools/ab/trace_wino6.py): lane 0 of every wave of every 16th workgroup stamps the shader clock into a.ws
    unsigned long long* trc = (a.ws && (blockIdx.x % W6_TRACE_EVERY) == 0 && lane == 0) ? reinterpret_cast<unsigned long long*>(a.ws) + ((blockIdx.x / W6_TRACE_EVERY) * 4 + wave) * 64 : nullptr;
    int trn = 0;
#define W6_STAMP() do { if (trc) { trc[trn] = __builtin_readcyclecounter(); } ++trn; } while (0)
    if (trc) {
        trc[63] = __builtin_amdgcn_s_memrealtime();
        trc[61] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));       // HW_REG_HW_ID (wave slot, SIMD, CU, SH, SE)
        trc[60] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((32 - 1) << 11));      // HW_REG_XCC_ID
    }
#else
#define W6_STAMP() do { } while (0)
#endif
    if (GEO == 1) {
        // halo columns 15..17 (image columns >= 14) are zero for every image this geometry accepts: their W slots are cleared here, once,
        // in both buffers, and pass 1 never touches them
        for (int i = tid; i < 2 * 2 * G::RG * 6 * 3; i += 256) {
            const int c3 = i % 3, rest = i / 3;
            const int a6 = rest % 6, r2 = rest / 6;
            const int r = r2 % G::RG, qb = r2 / G::RG;          // qb = buffer * 2 + quad
            sW[(qb >> 1) * G::WB + w6_slot<GEO>(qb & 1, r, a6, 15 + c3)] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    W6_STAMP();                                       // 0: start
    // ---- prologue ----------------------------------------------------------------------------------------------------------------
    // the halos of chunks 0 and 1 and the first weights are requested together: one memory round trip before the first MFMA
    load_D(c_lo);
    f32x4 d_first[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) d_first[i] = d[i];
    f32x4 sc_first = in_sc, sh_first = in_sh;
    load_D(min(c_lo + 1, nchunks - 1));
    load_U(c_lo * 3, 0);
    load_U(min(c_lo * 3 + 1, total_steps - 1), 1);
    {
        f32x4 d_keep[6], sc_keep = in_sc, sh_keep = in_sh;
#pragma unroll
        for (int i = 0; i < 6; ++i) { d_keep[i] = d[i]; d[i] = d_first[i]; }
        in_sc = sc_first; in_sh = sh_first;
        pass1(sW);
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = d_keep[i];
        in_sc = sc_keep; in_sh = sh_keep;
    }

    f32x16 acc[9];
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    // ---- main loop: one period per 8-channel chunk ------------------------------------------------------------------------------
    //   barrier: W(c) complete, everybody is done with W(c-1)
    //   step 0 (row A, first half); pass 1 of chunk c+1 (in registers since the last period) into the other W buffer; request chunk c+2
    //   steps 1, 2 (row A second half, row B half)
    // The wave's 4 channels are transformed and consumed two at a time, which halves the live registers of pass 2.
    struct X5 { f32x2 x0, x1, x2, x3, x4; };
    // the five W samples one half of a frequency row needs, for channel pair h of the lane's quad: first half = columns 0..4 of the
    // tile's six, second half = columns 1..5 (slot offsets of w6_slot: +12 per column, column 4 -> +1, column 5 -> +13)
    auto rd = [&](const f32x4* wrow, bool second, int h) {
        const f32x2* w2 = reinterpret_cast<const f32x2*>(wrow) + h;
        X5 x;
        constexpr int K = G::CK;
#if W6_ABL & 16
        x.x0 = x.x1 = x.x2 = x.x3 = x.x4 = f32x2{1.f, 2.f};
        asm volatile("" : "+v"(x.x0), "+v"(x.x1), "+v"(x.x2), "+v"(x.x3), "+v"(x.x4) : "v"(w2));
        return x;
#endif
        if (!second) { x.x0 = w2[0 * 2]; x.x1 = w2[K * 2]; x.x2 = w2[2 * K * 2]; x.x3 = w2[3 * K * 2]; x.x4 = w2[1 * 2]; }
        else         { x.x0 = w2[K * 2]; x.x1 = w2[2 * K * 2]; x.x2 = w2[3 * K * 2]; x.x3 = w2[1 * 2]; x.x4 = w2[(K + 1) * 2]; }
        return x;
    };
    // row B: which half is wave-uniform.  With A = row + (second half ? K : 0) both halves read A[0], A[K], A[2K], A[1] and one more sample,
    // A[3K] or A[1-K]: two address registers set up once (a branch between the two offset sets cost 22 VALU instructions per period in
    // address arithmetic and moves, each of which takes a slot from the matrix pipe)
    auto rdB = [&](const f32x4* wa, const f32x4* w3, int h) {
        const f32x2* a2 = reinterpret_cast<const f32x2*>(wa) + h;
        X5 x;
        constexpr int K = G::CK;
#if W6_ABL & 16
        x.x0 = x.x1 = x.x2 = x.x3 = x.x4 = f32x2{1.f, 2.f};
        asm volatile("" : "+v"(x.x0), "+v"(x.x1), "+v"(x.x2), "+v"(x.x3), "+v"(x.x4) : "v"(a2), "v"(w3));
        return x;
#endif
        x.x0 = a2[0]; x.x1 = a2[K * 2]; x.x2 = a2[2 * K * 2]; x.x3 = reinterpret_cast<const f32x2*>(w3)[h]; x.x4 = a2[1 * 2];
        return x;
    };
    auto mm = [&](const f32x2 v0, const f32x2 v1, const f32x2 v2, int h, int sbuf, int abase) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc[abase + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[s], ub[sbuf][0][2 * h + s], acc[abase + 0], 0, 0, 0);
            acc[abase + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[s], ub[sbuf][1][2 * h + s], acc[abase + 1], 0, 0, 0);
            acc[abase + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2[s], ub[sbuf][2][2 * h + s], acc[abase + 2], 0, 0, 0);
        }
    };
    // one half-step = 6 MFMAs: transform the samples read during the previous half-step, issue the MFMAs
    auto half_step = [&](const X5& x, bool second, int h, int sbuf, int abase) {
        f32x2 v0, v1, v2;
        if (!second) w6_half_first(x.x0, x.x1, x.x2, x.x3, x.x4, five, v0, v1, v2);
        else w6_half_second(x.x0, x.x1, x.x2, x.x3, x.x4, five, v0, v1, v2);
        mm(v0, v1, v2, h, sbuf, abase);
    };
    auto half_stepB = [&](const X5& x, int h) {      // MFMAs stay outside the branch: on both sides of one the allocator keeps two
        f32x2 v0, v1, v2;                           // copies of the accumulators they touch
        if (halfB == 0) { asm volatile("" ::: "memory"); w6_half_first(x.x0, x.x1, x.x2, x.x3, x.x4, five, v0, v1, v2); }
        else            { asm volatile("" ::: "memory"); w6_half_second(x.x0, x.x1, x.x2, x.x3, x.x4, five, v0, v1, v2); }
        mm(v0, v1, v2, h, 2, 6);
    };
    W6_STAMP();                                       // 1: prologue done
    // two periods per trip: the W buffer of a period is then a compile-time offset of every LDS instruction (the parity as a register cost
    // vector adds per period)
    auto period = [&](const int c, auto parity) {
        constexpr int wcur = decltype(parity)::value * W6_WB;
        f32x4* wnext = sW + (1 - decltype(parity)::value) * W6_WB;
        const int step = c * 3;
#if !(W6_ABL & 8)
        __syncthreads();
#endif
#ifdef W6_TRACE
        if (c < 40) W6_STAMP();                       // 2 + c: period c entered
#endif
        // the LDS reads of half-step k+1 are issued in front of the MFMAs of half-step k: their latency hides under the 6 MFMAs
#ifdef W6_TRACE
#define W6_STAMP_AT(slot) do { if (trc && c == 8) trc[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define W6_STAMP_AT(slot) do { } while (0)
#endif
        W6_STAMP_AT(50);
        // AHEAD (variants without the fused input affine): the LDS reads of half-step k+1 are issued in front of the MFMAs of half-step k.
        // With the affine its scale/shift registers leave no room for the second sample set (6 spilled registers, reloaded every period):
        // there each half-step reads its own samples.
        constexpr bool AHEAD = !AFF;
        load_U(min(step + 2, total_steps - 1), 2);
        X5 xa = rd(wA + wcur, false, 0), xb;
        if (AHEAD) xb = rd(wA + wcur, false, 1);
        half_step(xa, false, 0, 0, 0);
        W6_STAMP_AT(51);                              // after reads + transform + 6 MFMAs issued
        if (AHEAD) xa = rd(wA + wcur, true, 0); else xb = rd(wA + wcur, false, 1);
        half_step(xb, false, 1, 0, 0);
        W6_STAMP_AT(52);                              // step 0 issued
        W6_FENCE;
#if !(W6_ABL & 1)
        pass1(wnext);
#endif
        W6_STAMP_AT(53);                              // pass 1 done (LDS writes issued)
#if !(W6_ABL & 2)
        load_D(min(c + 2, nchunks - 1));
#endif
        W6_STAMP_AT(54);                              // halo loads issued
        W6_FENCE;
        load_U(min(step + 3, total_steps - 1), 0);
        if (AHEAD) xb = rd(wA + wcur, true, 1); else xa = rd(wA + wcur, true, 0);
        half_step(xa, true, 0, 1, 3);
        if (AHEAD) xa = rdB(wBa + wcur, wB3 + wcur, 0); else xb = rd(wA + wcur, true, 1);
        half_step(xb, true, 1, 1, 3);
        W6_STAMP_AT(55);                              // step 1 issued
        W6_FENCE;
        load_U(min(step + 4, total_steps - 1), 1);
        if (AHEAD) xb = rdB(wBa + wcur, wB3 + wcur, 1); else xa = rdB(wBa + wcur, wB3 + wcur, 0);
        half_stepB(xa, 0);
        if (!AHEAD) xb = rdB(wBa + wcur, wB3 + wcur, 1);
        half_stepB(xb, 1);
        W6_STAMP_AT(56);                              // step 2 issued
    };
    for (int c = c_lo; c < nchunks; c += 2) {     // an even number of chunks: Cin is a multiple of 16 (validate), split-K bounds are even (host)
        period(c, std::integral_constant<int, 0>{});
        period(c + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------------------------
    // A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].  Row pass in registers: per accumulator entry the wave's 6 + 3
    // frequencies become P[rowA][0..3] and the partial P[rowB][0..3] of its half (the two halves add up).  Wave d finishes the tiles of
    // accumulator registers 4d..4d+3: in round q the other waves send it the 8 values of registers 4d+2q, 4d+2q+1.
    // scale/shift are requested here: the two exchange rounds cover their latency (loaded inside the store loop they would serialise it)
    const int co = co0 + li;
    const bool cvalid = co < a.Cout;
    float sc = P.scale[min(co, a.Cout - 1)];
    float sh = P.shift[min(co, a.Cout - 1)];
#ifdef W6_TRACE
    trn = 42;
#endif
    W6_STAMP();                                       // 42: loop done (own MFMAs issued)
    __syncthreads();
    W6_STAMP();                                       // 43: everybody done
    // The stores below sit in per-tile predicated blocks; the compiler's wait-count pass cannot prove across their joins that the two loads
    // above have landed and would put `s_waitcnt vmcnt(0)` in front of every store — which also waits for the previous STORE to retire
    // (measured: 340 ns per store, 26 us of a 76 us workgroup).  Wait once here and hand the values over through an asm the pass
    // cannot see through: from now on they are plain register values.
    asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(sc), "+v"(sh) : : "memory");
    // split-K: raw partial sums (no scale / shift / ReLU) go to this split's slab of a.ws, laid out [pixel][cout_pad]; the reduce kernel of
    // conv_igemm.hip sums the slabs in a fixed order and applies the epilogue
    const bool raw = a.ksplit > 1;
    float* const ybuf = raw ? a.ws + (long)ks * P.total_pix * a.cout_pad : P.y;
    const int ycs = raw ? a.cout_pad : a.y_cs, yco = raw ? 0 : a.y_co;
    if (raw) { sc = 1.f; sh = 0.f; }
    // Everything below works on PAIRS of accumulator registers (r, r+1 = two tiles of the lane) with packed-fp32 instructions, and the
    // stores of interior tiles take a wave-uniform (row, column) base from the scalar unit plus one lane offset per tile: a VALU instruction
    // issued here waits for a gap in the MFMA stream of the other workgroup on the SIMD and takes the slot from it (trace: this epilogue
    // ran 12.4 us next to a partner, 5.6 us alone), so the epilogue is priced in VALU instructions — 3x fewer than the scalar form.
    f32x2* ex2 = reinterpret_cast<f32x2*>(smem);        // exchange: [src wave][dst wave][value 0..7][lane] pairs = 64 KiB
    const float lo = (co < a.relu_upto && !raw) ? 0.f : __builtin_nanf("");      // max(v, NaN) = v: lanes without the ReLU
    const f32x2 sc2 = {sc, sc}, sh2 = {sh, sh};
    auto fma2 = [](f32x2 x, float k, f32x2 y) { return __builtin_elementwise_fma(x, f32x2{k, k}, y); };
    const bool want_stats = a.gn_ws != nullptr;
    f32x2 gs2 = {0.f, 0.f}, gss2 = {0.f, 0.f};
    float gs = 0.f, gss = 0.f;
    float* yimg = ybuf + (long)n * H * W * ycs + yco + co;
    // scalar side of the store addresses: image base (the pair's first image for GEO 1) and the byte strides of one pixel / one row
    unsigned long long ybase_s;
    {
        const unsigned long long yb = (unsigned long long)(ybuf + (long)n * H * W * ycs);
        ybase_s = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(yb >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)yb);
    }
    const unsigned long long px_b = (unsigned long long)ycs * 4u, rowskip_b = (unsigned long long)(W - 3) * ycs * 4u;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        f32x2 own[8];
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
            if (dd == wave) {
#pragma unroll
                for (int k = 0; k < 8; ++k) own[k] = v[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) ex2[(((wave * 4 + dd) * 8 + k) << 6) + lane] = v[k];
            }
        }
        W6_STAMP();                                   // 44 / 47: round written
        __syncthreads();
        W6_STAMP();                                   // 45 / 48: round visible
        // P[a][j]: rows 0..3 from waves 0..3 (values 0..3), row 4 = halves of waves 0, 1, row 5 = halves of waves 2, 3 (values 4..7)
        f32x2 Pm[6][4];
        {
            f32x2 part[4][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x2 v[8];
                if (s == wave) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = own[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = ex2[(((s * 4 + wave) * 8 + k) << 6) + lane];
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
        // the pair's entries are accumulator registers 4*wave + 2q, +1 of lane half hh: tiles m, m + 1
        bool full[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int m = 2 * q + rr + 8 * wave + 4 * hh;
            int mimg, mt, mtc;
            G::tile_of(m, mimg, mt, mtc);
            const int oh = oh0 + 4 * mt, ow = ow0 + 4 * mtc;
            const bool tile_ok = cvalid && m < G::TILES && n + mimg < P.N;
            full[rr] = tile_ok && oh + 4 <= H && ow + 4 <= W;
            if (full[rr]) {                             // interior tile: 16 stores, no per-store predicate, no vector address arithmetic
                const unsigned voff = (unsigned)((((mimg * H + oh) * W + ow) * ycs + yco + co) * 4);
                unsigned long long sp = ybase_s;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float val = rr ? yv[i][j].y : yv[i][j].x;
                        // ("+s": the pointer is walked between the stores, not computed 16 times up front)
                        // nt: the output is a stream (84 MB per launch at stage 2) that must not push the weights and the halo lines this launch re-reads
                        // out of L2; measured -2.2 % on the map shapes, +0.4 % end to end (profiles/r03_ablations.txt), sc0 / sc1 nothing
                        asm volatile("global_store_dword %1, %2, %0 nt" : "+s"(sp) : "v"(voff), "v"(val) : "memory");
                        sp += j == 3 ? rowskip_b : px_b;
                    }
            } else if (tile_ok) {
                float* yp0 = yimg + (((long)mimg * H + oh) * W + ow) * ycs;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (oh + i < H && ow + j < W) {
                            const float val = rr ? yv[i][j].y : yv[i][j].x;
                            yp0[((long)i * W + j) * ycs] = val;
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
        W6_STAMP();                                   // 46 / 49: round stored
        if (q == 0) __syncthreads();                        // the exchange buffer is reused by round 1
    }
    gs += gs2.x + gs2.y;
    gss += gss2.x + gss2.y;
    // fused GroupNorm statistics of the NEXT layer's normalisation (fcos.py:182-186): one {sum, sumsq} record per
    // (spatial tile, wave, group)
#ifdef W6_TRACE
    if (trc) trc[62] = __builtin_amdgcn_s_memrealtime();
#endif
    if (a.gn_ws) {
        for (int o = 1; o < a.gn_cpg; o <<= 1) { gs += __shfl_xor(gs, o); gss += __shfl_xor(gss, o); }
        gs += __shfl_xor(gs, 32);
        gss += __shfl_xor(gss, 32);
        if (cvalid && hh == 0 && (li & (a.gn_cpg - 1)) == 0) {
            double* o = a.gn_ws + (((long)bx * 4 + wave) * a.gn_groups + co / a.gn_cpg) * 2;
            o[0] = (double)gs;
            o[1] = (double)gss;
        }
    }
}

template <int GEO>
static int launch_wino6_geo(ConvArgs& a, hipStream_t st) {
    static DeviceOnce once;
    int rc = once.run([]() {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino6_kernel<false, GEO>), hipFuncAttributeMaxDynamicSharedMemorySize, w6_lds_alloc<GEO>());
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino6_kernel<true, GEO>), hipFuncAttributeMaxDynamicSharedMemorySize, w6_lds_alloc<GEO>());
        return e == hipSuccess ? CMK_OK : fail(CMK_ELAUNCH, "conv_wino6: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
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
    a.grid_y = cdiv(a.Cout, 32);
    a.total_tiles = blocks;
    if (a.ksplit < 1) a.ksplit = 1;
    const dim3 grid(((blocks + 7) / 8) * 8 * a.grid_y, a.ksplit);
    if (a.p[0].in_scale)
        hipLaunchKernelGGL((conv_wino6_kernel<true, GEO>), grid, dim3(256), w6_lds_alloc<GEO>(), st, a);
    else
        hipLaunchKernelGGL((conv_wino6_kernel<false, GEO>), grid, dim3(256), w6_lds_alloc<GEO>(), st, a);
    return check_launch("conv_wino6");
}

// geo 0: 12x40-pixel tiles of one image; geo 1: pairs of whole maps of at most 16 rows x 14 columns (one problem, no fused GN statistics)
int launch_wino6(ConvArgs& a, int geo, hipStream_t st) {
    for (int i = 0; i < a.nprob; ++i)       // the epilogue's stores take a 32-bit byte offset inside the output image (GEO 1: inside a pair of images)
        if ((long)(geo == 0 ? 1 : 2) * a.p[i].H * a.p[i].W * std::max(a.y_cs, a.cout_pad) * 4 >= (1L << 32))
            return fail(CMK_EINVAL, "conv_wino6: an output image of 4 GiB or more%s", "");
    if (a.ksplit > 1 && (a.nprob != 1 || a.gn_ws || !a.ws || ((a.Cin >> 3) % (2 * a.ksplit)) || a.cout_pad < cdiv(a.Cout, 32) * 32))
        return fail(CMK_EINVAL, "conv_wino6: split-K takes one problem, no GroupNorm statistics, a workspace and Cin / 8 chunks %% (2 * splitk) == 0%s", "");
    if (geo == 0) return launch_wino6_geo<0>(a, st);
    if (a.nprob != 1 || a.p[0].H > 16 || a.p[0].W > 14 || a.gn_ws)
        return fail(CMK_EINVAL, "conv_wino6: the RoI-pair geometry takes one problem of maps up to 16x14 and produces no GroupNorm statistics%s", "");
    return launch_wino6_geo<1>(a, st);
}

}  // namespace cmk

extern "C" int64_t cmk_wino6_packed_floats(int Cout, int Cin) {
    return (int64_t)((Cin + 7) / 8) * ((Cout + 31) / 32) * 36 * 256;
}
