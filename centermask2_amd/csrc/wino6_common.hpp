// Shared by the two Winograd F(4x4,3x3) kernels (conv_wino6.hip: 32 couts per workgroup, two workgroups per CU; conv_wino6s.hip: 64 couts
// per workgroup with the frequency image V shared through LDS): tile geometries, the conflict-free W-image slot function and the
// packed-fp32 half transforms.
#pragma once
#include "conv_args.hpp"

namespace cmk {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// bounds-checked 16-byte load: lanes whose byte offset lies outside [0, num_records) of the resource get 0
__device__ f32x4 w6_buffer_load(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

// Two tilings of the 32 MFMA rows (a workgroup's 32 tiles of 4x4 outputs):
//   GEO 0  maps: 3 x 10 tiles of ONE image (12 x 40 pixels: every map width of the model is a multiple of 40); rows 30, 31 carry no tile;
//   GEO 1  RoI maps (at most 16 rows x 14 columns, e.g. the 14x14 RoI features of the mask / mask-IoU heads): 4 x 4 tiles of each of
//          TWO consecutive images; halo columns 15..17 (image columns >= 14) are zero by construction, so their W slots are cleared
//          once and only 15 columns go through pass 1: 2 x (4 x 15 x 2) = 240 items, waves 0-1 image 0, waves 2-3 image 1.
// W image, in 16-byte slots: entry (channel quad q, row group r, grid row a, halo column col) lives at
//   q*QP + r*TP + a*AP + (col & 3)*CK + (col >> 2)          r = tile row (GEO 0) or 4*image + tile row (GEO 1)
// A lane of the MFMA side is tile m and reads col = 4*tc + j, i.e. slot = const + r*TP + tc (+1 for j >= 4).  A ds_read_b128 is served
// in groups of 16 lanes {0-3,12-15,20-27} / {4-11,16-19,28-31} per half wave; TP is chosen modulo 16 (10 for GEO 0, 4 for GEO 1) so that
// the tiles of either group fall on 16 distinct slots modulo 16, whatever a and j are: conflict-free without padding the rows.
template <int GEO> struct W6G;
template <> struct W6G<0> {
    static constexpr int OH = 12, OW = 40, HC = 42, CK = 12, AP = 48, TP = 6 * 48 + 10, RG = 3, QP = RG * TP, WB = 2 * QP;
    static constexpr int ITEMS = 3 * HC * 2;                // 252: (tile row, halo column, channel quad)
    static constexpr int TILES = 30;
    __device__ static __forceinline__ void tile_of(int m, int& img, int& t, int& tc) { img = 0; t = (m * 205) >> 11; tc = m - t * 10; }      // m / 10, m < 32
    __device__ static __forceinline__ void item_of(int tid, int& img, int& q, int& t, int& col, bool& active) {
        const int i = min(tid, ITEMS - 1);                 // threads 252..255 repeat the last item (same values, same slots)
        img = 0; q = i & 1; const int cc = i >> 1; t = cc / HC; col = cc - t * HC; active = true;
    }
};
template <> struct W6G<1> {
    static constexpr int OH = 16, OW = 14, HC = 15, CK = 5, AP = 20, TP = 6 * 20 + 12, RG = 8, QP = RG * TP, WB = 2 * QP;
    static constexpr int ITEMS = 256;
    static constexpr int TILES = 32;
    __device__ static __forceinline__ void tile_of(int m, int& img, int& t, int& tc) { img = m >> 4; t = (m >> 2) & 3; tc = m & 3; }
    __device__ static __forceinline__ void item_of(int tid, int& img, int& q, int& t, int& col, bool& active) {
        img = tid >> 7; const int i = tid & 127; active = i < 4 * HC * 2;
        const int j = min(i, 4 * HC * 2 - 1); q = j & 1; const int cc = j >> 1; t = cc / HC; col = cc - t * HC;
    }
};
static_assert(W6G<0>::TP % 16 == 10 && W6G<1>::TP % 16 == 4, "conflict-free W image");

template <int GEO>
__device__ __forceinline__ int w6_slot(int q, int r, int a, int col) {
    using G = W6G<GEO>;
    return q * G::QP + r * G::TP + a * G::AP + (col & 3) * G::CK + (col >> 2);
}

// Packed-fp32 arithmetic spelled out.  The transforms are the minimal sequences of v_pk_* instructions (6 per half transform of two
// channels); left to the compiler the same formulas came out as a mix of scalar FMAs, sign flips (v_xor) and register moves — 5.5 VALU
// instructions per MFMA instead of 1.5.  One asm block per half transform: the compiler cannot see what kind of instruction wrote the
// results, so it cannot keep its own distance rules between a VALU write and the MFMA / LDS store that reads it (built from single-
// instruction asm statements the kernel computed garbage as soon as the scheduler moved them); the block ends with the wait states itself.
//   first  (B^T rows 0-2 on x0..x4 = d0..d4):  v0 = 4x0 - 5x2 + x4,  v1 = (x4 - 4x2) + (x3 - 4x1),  v2 = (x4 - 4x2) - (x3 - 4x1)
//   second (B^T rows 3-5 on x0..x4 = d1..d5):  v0 = (x3 - x1) + 2(x2 - x0),  v1 = (x3 - x1) - 2(x2 - x0),  v2 = 4x0 - 5x2 + x4
__device__ __forceinline__ void w6_half_first(const f32x2 x0, const f32x2 x1, const f32x2 x2, const f32x2 x3, const f32x2 x4, const f32x2 five,
                                              f32x2& v0, f32x2& v1, f32x2& v2) {
    f32x2 p, q;
    asm volatile("v_pk_fma_f32 %0, %7, %10, %9 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"      // v0 = x4 - 5 x2
                 "v_pk_fma_f32 %3, %7, 4.0, %9 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"       // p  = x4 - 4 x2
                 "v_pk_fma_f32 %4, %6, 4.0, %8 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"       // q  = x3 - 4 x1
                 "v_pk_fma_f32 %0, %5, 4.0, %0 op_sel_hi:[1,0,1]\n\t"                                      // v0 += 4 x0
                 "v_pk_add_f32 %1, %3, %4\n\t"                                                             // v1 = p + q
                 "v_pk_add_f32 %2, %3, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"                                   // v2 = p - q
                 "s_nop 1"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(p), "=&v"(q)
                 : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "s"(five));
}
__device__ __forceinline__ void w6_half_second(const f32x2 x0, const f32x2 x1, const f32x2 x2, const f32x2 x3, const f32x2 x4, const f32x2 five,
                                               f32x2& v0, f32x2& v1, f32x2& v2) {
    f32x2 r, t;
    asm volatile("v_pk_add_f32 %3, %8, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"                                   // r  = x3 - x1
                 "v_pk_add_f32 %4, %7, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"                                   // t  = x2 - x0
                 "v_pk_fma_f32 %2, %7, %10, %9 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"      // v2 = x4 - 5 x2
                 "v_pk_fma_f32 %0, %4, 2.0, %3 op_sel_hi:[1,0,1]\n\t"                                      // v0 = r + 2t
                 "v_pk_fma_f32 %1, %4, 2.0, %3 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"       // v1 = r - 2t
                 "v_pk_fma_f32 %2, %5, 4.0, %2 op_sel_hi:[1,0,1]\n\t"                                      // v2 += 4 x0
                 "s_nop 1"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(r), "=&v"(t)
                 : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "s"(five));
}

}  // namespace cmk
