// Internal (C++) interfaces between the C-ABI translation unit and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/aid_hip.h"

namespace aid {

struct GemmDesc {
    const void* a;
    const void* b;
    void*       c;
    const void* bias;
    const void* residual;                             // added after the rounding of scale * A B^T + bias (NULL = none)
    int32_t m, n, k;
    int32_t lda, ldb, ldc;
    int32_t batch;
    float   scale;                                    // C = scale * A B^T + bias
    int64_t stride_a, stride_b, stride_c;
    // LayerNorm folded into the projection (AidGemmProblem.ln_*): the weight operand holds W' = W * gamma, the epilogue
    // turns  x W'^T  into  LayerNorm(x) W^T = rstd (x W'^T - mean * colsum) + shift
    const float* ln_stats;                            // [activation rows, 2] = (mean, rstd); NULL = no LayerNorm
    const float* ln_colsum;                           // [weight rows]  sum_k W'[row, k]
    const float* ln_shift;                            // [weight rows]  sum_k beta[k] W[row, k]
    int32_t ln_side;                                  // 1: the activation is A (statistics by m), 2: it is B (by n)
    int32_t trans_rows;                               // > 0: C is written transposed per frame of `trans_rows` rows (AidGemmProblem)
    int64_t stride_stats;                             // activation rows per batch
};

struct GemmGroup {                        // passed by value as the kernel argument
    GemmDesc p[AID_GEMM_MAX_PROBLEMS];
    int32_t  tile_start[AID_GEMM_MAX_PROBLEMS + 1];   // prefix sums of block counts (filled by the launcher)
    int32_t  n_problems;
    int32_t  interleave;                              // 1: every XCD gets a chunk of each problem (unequal K loops)
};

// Side problems of a ping-pong launch: the few short problems of a group whose K loop differs from the main ones (the
// K = 2048 text-context projections next to the K = 1280 query projection of a cross-attention layer).  They run as
// 128 x 128 lock-step tiles in the FIRST blocks of the same launch, so the main problems keep the big-tile engine.
struct GemmSide {
    GemmDesc p[4];
    int32_t  tile_start[5];               // prefix sums of 128 x 128 tile counts
    int32_t  n;                           // number of side problems (0 = none)
    int32_t  tiles;                       // total side tiles
    int32_t  pad_tiles;                   // tiles rounded up to a multiple of 8 (keeps block -> XCD of the main tiles)
};

// Development / tuning knobs.  They are NOT read from the environment in the launch path: the table is filled once when
// the library is loaded (environment variables AID_<NAME>, e.g. AID_ATTN_NW=8) and can be changed at run time through
// aid_set_tuning() — same-process A/B runs flip a knob between launches.  -1 = unset (the launch heuristics decide).
enum Tune {
    TUNE_GEMM_VARIANT = 0,      // 7 = force the lock-step engine, 31 = force the ping-pong engine
    TUNE_GEMM_PP,               // ping-pong K loop: 0 = DMA issued between the MFMAs, 1 = in the read slot, 2 = 1 + s_setprio, 3 = 0 + s_setprio
    TUNE_GEMM_TRI,              // 0 = never use the twelve-wave 288 x 256 engine, 1 = force it where the shape allows
    TUNE_ATTN_NW,               // 4 / 8 waves per workgroup
    TUNE_ATTN_QB,               // 1 / 2 query blocks per wave (d = 40 PLAIN)
    TUNE_ATTN_PIPE,             // 0 / 1 software-pipelined loop
    TUNE_ATTN_RES,              // 0 / 1 resident key segments
    TUNE_ATTN_RES_CHUNKS,       // > 0: chunks per (frame, head) of the resident variant
    TUNE_ATTN_ORDER,            // 0 = plain XCD order for mixed launches
    TUNE_ATTN_V2,               // ping-pong d = 64 kernel: 0 never / 1 wherever supported; default: fused OUTER l >= 1024, everything else l >= 2048
    TUNE_CU_SHARE,              // n > 1: the caller runs n independent launch streams side by side (two passes on two streams): the GEMM
                                // engine choice plans with 1 / n of the CUs; a hint, results never depend on it
    TUNE_GEMM_RS,               // row-stationary engine for the short-K levels (aid_gemm_rs.hip): 0 = never, 1 = wherever the shape allows;
                                // default: wherever the shape allows AND the activation has enough row tiles to fill the device
    TUNE_ATTN_TX,               // text-key kernel (aid_attn_tx.hip: d = 64, <= 96 keys per segment, PLAIN / INNER / OUTER): 0 = never; default: wherever supported
    TUNE_ATTN_TX_TILES,         // > 0: 32-row tiles per wave of that kernel (sets the workgroups per (frame, head))
    TUNE_GEMM_LS,               // ring of the lock-step engine: 0 = 2 stages of 64 k, 2 workgroups / CU (rounds 1 - 5); 1 = 4 stages of 64 k,
                                // 1 workgroup / CU; default: 1 for launches of at most one workgroup per CU with >= 16 K tiles (aid_gemm.hip)
    TUNE_COUNT
};
int tune(int id);

// picks the tile shape, fills g.tile_start and launches
// cu_share > 1: the caller runs that many launch streams side by side (AidGemmProblem.cu_share); 0 / 1: the process-wide CU_SHARE knob decides
hipError_t gemm_group_launch(GemmGroup& g, int dtype, hipStream_t stream, const char** variant = nullptr, int cu_share = 0);

// row-stationary engine (aid_gemm_rs.hip): K = 320 / 640, one shared tall activation; `ncu` = CUs the launch may count on
bool       gemm_rs_supported(const GemmGroup& g, int ncu, bool ignore_size);
hipError_t gemm_rs_launch(const GemmGroup& g, int dtype, int ncu, hipStream_t stream);

// float32 storage path (aid_f32.hip)
hipError_t gemm_f32_launch(GemmGroup& g, hipStream_t stream);
hipError_t attn_f32_launch(const AidAttnArgs& a, hipStream_t stream);
hipError_t lerp_kv_f32_launch(const void* k, const void* vt, void* k2, void* vt2, const float* coef, int n_frames, int begin,
                              int end, int64_t k_fs, int64_t vt_fs, hipStream_t stream);

// attention core; returns hipSuccess / error, writes the variant name for profiling
// skip_single: the frames with ONE key segment are left to the ping-pong kernel (attn_pp_launch on the same stream)
hipError_t attn_launch(const AidAttnArgs& a, hipStream_t stream, const char** variant, bool skip_single = false);
// d = 64 ping-pong kernel for the single-segment frames of a call (aid_attn_pp.hip); frames with more segments exit at once
bool       attn_pp_supported(const AidAttnArgs& a);
hipError_t attn_pp_launch(const AidAttnArgs& a, hipStream_t stream, bool multi);
// d = 64 kernel for TEXT keys (<= 96 keys per segment resident in LDS; aid_attn_tx.hip): whole PLAIN / INNER / OUTER calls
bool       attn_tx_supported(const AidAttnArgs& a);
hipError_t attn_tx_launch(const AidAttnArgs& a, hipStream_t stream);
bool       attn_head_dim_supported(int d);
hipError_t lerp_kv_launch(const void* k, const void* vt, void* k2, void* vt2, const float* coef, int n_frames, int begin,
                          int end, int64_t k_fs, int64_t vt_fs, int dtype, hipStream_t stream);
const char* attn_variant_name(const AidAttnArgs& a);   // thread-local buffer
hipError_t layernorm_launch(const void* x, const void* gamma, const void* beta, void* y, int64_t rows, int c, float eps,
                            int dtype, hipStream_t stream);
hipError_t ln_stats_launch(const void* x, float* stats, int64_t rows, int c, float eps, int dtype, hipStream_t stream);
hipError_t ln_fold_launch(const void* w, const void* gamma, const void* beta, void* w_folded, float* colsum, float* shift,
                          int rows, int c, int dtype, hipStream_t stream);
bool       layernorm_width_supported(int c);

}  // namespace aid
