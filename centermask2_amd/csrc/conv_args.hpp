// Shared by the conv kernels of libcmk_hip.so: launch arguments (one struct passed by value) and vector types.
#pragma once
#include "cmk_common.hpp"

namespace cmk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PST = 20;      // LDS row pitch in floats: 16 channels + 4 pad
constexpr int MAXP = 10;     // problems per launch: the 5 FPN levels, twice where two convs with different weights share a launch (F(4x4) kernels)
constexpr int LDS_CU = 160 * 1024;

// workgroups per CU the register budget allows (accumulators: 16 VGPRs per 32x32 tile)
__host__ __device__ constexpr int occ_of(int wm, int wn, int stride) { return (stride == 1 && (wm == 1 ? wn <= 5 : wn <= 2)) ? 3 : 2; }

struct ConvProblem {
    const float* x; float* y; const float* scale; const float* shift;
    const float* in_scale; const float* in_shift;   // optional (N, Cin): x' = relu(x * in_scale + in_shift) while staging (fused GroupNorm+ReLU)
    const float* w;                                 // F(4x4) kernels: this problem's packed U (problems of one launch may differ in weights); conv_sp3: its fp16 split packing
    float acc_scale;                                // conv_sp3: 1 / S_w of this problem's weights (cmk.h w_splith_scale)
    int N, H, W, Ho, Wo;
    int tiles_h, tiles_w, tile_begin;
    long total_pix;  // N*Ho*Wo
};

struct ConvArgs {
    ConvProblem p[MAXP];
    int nprob;
    const float* w; const float* res;
    int Cin, Cout;
    int x_cs, x_co, y_cs, y_co, res_cs, res_co, res_mode, Hr, Wr;
    int relu_upto, in_relu;
    int cout_pad;
    int total_tiles;   // spatial tiles of all problems (XCD-aware kernels pad the grid to a multiple of 8 tiles)
    int ksplit;        // split-K: blockIdx.y owns an (even) range of the 16-channel chunks and writes raw partial sums to ws
    float* ws;         // [ksplit][total_pix][cout_pad]
    double* gn_ws;     // Winograd 2-WG form: per (spatial tile, row parity, group) partial {sum, sum of squares} of the outputs (fused GroupNorm statistics)
    int gn_cpg, gn_groups;
    int ga_stride;     // gather form (GA): stride of the 3x3 conv whose taps are walked as 9x more K chunks
    float* pool_ws;    // conv_pw only: per 32*MT-row block two records of Cout floats (cmk.h: cmk_conv_desc.pool_ws)
    int grid_y;   // N tiles; the N-tile index is the FASTEST block coordinate so the workgroups sharing an input tile run together (L2 reuse)
};

// per-device one-time kernel attribute set-up (hipFuncSetAttribute is per device; a process may drive several)
struct DeviceOnce {
    unsigned char done[64] = {0};
    template <typename F>
    int run(F&& f) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(CMK_ELAUNCH, "hipGetDevice failed%s", "");
        if (__atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) return CMK_OK;
        int rc = f();
        if (rc == CMK_OK) __atomic_store_n(&done[dev], 1, __ATOMIC_RELEASE);   // idempotent: a race only repeats the call
        return rc;
    }
};

// conv_wino6.hip: fused Winograd F(4x4,3x3)
int launch_wino6(ConvArgs& a, int geo, hipStream_t st);
// conv_wino6s.hip: the same, 64 couts per workgroup from one frequency image shared through LDS (one 8-wave workgroup per CU)
int launch_wino6s(ConvArgs& a, int geo, hipStream_t st);
// conv_pw.hip: 1x1 conv as a GEMM with the weights fetched straight into registers; mt = 4 | 2 accumulator rows per wave
int launch_pw(ConvArgs& a, int mt, hipStream_t st);
// conv_pw.hip, opt-in: the same GEMM from split products (fp32-accurate; mode 1: three bf16 pieces, a.w = cmk.h w_split; 2: two fp16 pieces, w_splith)
int launch_pw_split(ConvArgs& a, int mode, hipStream_t st);
// conv_sp3.hip, opt-in: 3x3 stride-1 conv as a direct implicit GEMM on fp16-split products (halo tile in LDS, 2 pieces per operand, geo 0..3)
int launch_sp3(ConvArgs& a, int geo, int pieces, hipStream_t st);

}  // namespace cmk
