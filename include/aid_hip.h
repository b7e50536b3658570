/*
 * aid_hip.h — C ABI of libaid_hip.so: the MI355X (gfx950 / CDNA4) implementation of the
 * interpolated-attention hot path of QY-H00/attention-interpolation-diffusion.
 *
 * Plain pointers and sizes only: no torch / HIP types appear in any signature, so the
 * library can be bound from ctypes (what this repository's Python processors do), cffi,
 * pybind11 or any other FFI.  All pointers are DEVICE pointers unless stated otherwise;
 * `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).
 * Every entry point only enqueues work on `stream`: no allocation, no host
 * synchronisation, no hidden copies.  Return value: AID_OK (0) or a negative AID_ERR_*
 * code; aid_strerror() gives the text.  No C++ exception crosses this boundary.
 *
 * Reference interfaces replaced (file:line in /root/reference):
 *   aid_processor_fwd      <- InterpolatedAttnProcessor family __call__ bodies:
 *                             OuterInterpolatedAttnProcessor   interpolation.py:573-679
 *                             InnerInterpolatedAttnProcessor   interpolation.py:707-804
 *                             de-activated fallback (AttnProcessor2_0)  interpolation.py:581-584
 *                             IP-Adapter variants (image branch = the ip_* fields):
 *                             OuterInterpolatedIPAttnProcessor  interpolation.py:240-387
 *                             InnerInterpolatedIPAttnProcessor  interpolation.py:417-545
 *                             ScaleControlIPAttnProcessor       interpolation.py:76-211
 *                             de-activated IP fallback (diffusers IPAdapterAttnProcessor2_0, called at
 *                             interpolation.py:248-251 / 425-428; semantics SURVEY.md App. A)
 *   aid_gemm_nt            <- attn.to_q / to_k / to_v / to_out[0]   interpolation.py:613,623-624,666
 *   aid_attn_fwd           <- end-point select / replicate / concat / get_attention_scores /
 *                             bmm / batch_to_head_dim / outer|inner lerp
 *                             interpolation.py:626-664 (outer), 760-790 (inner)
 */
#ifndef AID_HIP_H
#define AID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AID_ABI_VERSION 8

/* element types of activations / weights (accumulation is always fp32) */
#define AID_DTYPE_F16  0
#define AID_DTYPE_BF16 1
/* ABI v7: float32 tensors in and out, float32 arithmetic (v_mfma_f32_32x32x2_f32, the f32 vector rate) — the reference's own default
 * for SD1.x (gradio_src/app.py:62, 414) and its CPU path.  aid_gemm_nt, aid_attn_fwd, aid_lerp_kv and aid_processor_fwd take it
 * (probabilities are then not rounded before the PV product, like the reference's float32 get_attention_scores); the LayerNorm
 * entry points and the ln_* options of aid_processor_fwd are 16-bit only (AID_ERR_DTYPE); `residual` works with every dtype. */
#define AID_DTYPE_F32  2

/* image branch of the IP-Adapter processors (AidProcessorArgs.ip_mode) */
#define AID_IP_NONE       0
#define AID_IP_SAME       1
#define AID_IP_PLAIN      2

/* attention modes */
#define AID_MODE_PLAIN 0   /* softmax(Q K_i^T) V_i                       (AID de-activated)          */
#define AID_MODE_INNER 1   /* keys/values lerped between the end-point frames before attention     */
#define AID_MODE_OUTER 2   /* attention against each end-point frame, outputs lerped               */

/* error codes */
#define AID_OK             0
#define AID_ERR_ARG       -1   /* NULL pointer / negative size / inconsistent field              */
#define AID_ERR_DTYPE     -2   /* dtype not one of AID_DTYPE_*                                    */
#define AID_ERR_SHAPE     -3   /* unsupported head dim / alignment (see each function)            */
#define AID_ERR_WORKSPACE -4   /* workspace too small                                             */
#define AID_ERR_LAUNCH    -5   /* hipLaunchKernel failed (hipGetLastError text via aid_strerror)  */
#define AID_ERR_NO_DEVICE -6   /* no gfx950 device visible                                        */

/* ---------------------------------------------------------------------------------------
 * Grouped "NT" GEMM:   C[b] = A[b] * B[b]^T (+ bias)        for b in [0, batch)
 *   A [m, k] row-major (lda), B [n, k] row-major (ldb)  -> both operands K-contiguous,
 *   C [m, n] row-major (ldc).  bias (optional) is indexed by n and has the operand dtype.
 *   Up to AID_GEMM_MAX_PROBLEMS independent problems run in ONE launch (q, k and V^T
 *   projections of an attention layer).
 *   `residual` (optional, laid out like C incl. stride_c) is added AFTER the result was rounded to the
 *   operand dtype: C = round(round(scale * A B^T + bias) + residual) — bit for bit the reference's separate
 *   `hidden_states = attn_output + hidden_states` on the out-projection.
 *   Requirements: k % 8 == 0, lda % 8 == 0, ldb % 8 == 0, ldc % 4 == 0, ldc >= round_up(n, 4),
 *   all base pointers 16-byte aligned.  Columns [n, round_up(n,4)) of C are written with zeros.
 * ------------------------------------------------------------------------------------- */
#define AID_GEMM_MAX_PROBLEMS 6

typedef struct AidGemmProblem {
    const void* a;
    const void* b;
    void*       c;
    const void* bias;          /* NULL = none */
    int32_t     m, n, k;
    int32_t     lda, ldb, ldc; /* in elements */
    int32_t     batch;         /* >= 1 */
    float       scale;         /* C = scale * (A B^T) + bias; 0 means 1 (zero-initialised structs)  */
    int64_t     stride_a, stride_b, stride_c;   /* per-batch strides in elements (0 = shared) */
    const void* residual;      /* NULL = none; [m, ldc] per batch like C, operand dtype               */
    /* LayerNorm folded into the projection (ln_stats == NULL: none).  The ACTIVATION operand is un-normalised x, the     */
    /* WEIGHT operand is W' = W * gamma (aid_ln_fold), and the epilogue computes, before `scale` and `bias`,              */
    /*     LayerNorm(x) W^T = rstd * (x W'^T - mean * ln_colsum) + ln_shift                                               */
    /* ln_side 1: A is the activation (statistics row m, weight constants column n); 2: B is (statistics by n, constants  */
    /* by m — the V^T = Wv x^T problem).  ln_stats: fp32 [activation rows, 2] = (mean, rstd) from aid_ln_stats, batch b   */
    /* starts at row b * stride_stats; ln_colsum / ln_shift: fp32 [weight rows] from aid_ln_fold.                          */
    const float* ln_stats;
    const float* ln_colsum;
    const float* ln_shift;
    int32_t      ln_side;
    /* trans_rows > 0: C is written TRANSPOSED per frame of `trans_rows` rows: element (row m, column j) goes to          */
    /*     c + (m / trans_rows) * stride_c + j * ldc + m % trans_rows                                                     */
    /* i.e. V^T[frame][channel][key] from the flat value projection  E Wv^T  (a = E [frames * keys, k], b = Wv [n, k]),    */
    /* the layout the attention core reads.  Requirements: batch == 1, m % trans_rows == 0, trans_rows % 8 == 0,          */
    /* ldc >= trans_rows, ldc % 8 == 0, stride_c % 8 == 0, no bias, no residual, ln_side 1 if LayerNorm is folded.         */
    /* (Same numbers as the batched form  V^T[f] = Wv E_f^T; the library picks whichever its tile engines run faster.)    */
    int32_t      trans_rows;
    int64_t      stride_stats;
    /* ABI v7.  cu_share (read from problems[0]): n > 1 says that n independent launch streams of the CALLER run side by side (the   */
    /* conditional and the unconditional UNet call of a step on two streams), so this launch should plan with 1 / n of the CUs —    */
    /* a per-call hint (two host threads may pass different values); results never depend on it.  0 / 1: the whole device.          */
    int32_t      cu_share;
    int32_t      reserved0;
} AidGemmProblem;

int aid_gemm_nt(const AidGemmProblem* problems /* host */, int n_problems, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * LayerNorm over the last dimension: y[r, :] = (x[r, :] - mean_r) * rsqrt(var_r + eps) * gamma + beta
 * — the norm1 / norm2 in front of the attention call in diffusers' BasicTransformerBlock (the step
 * before the path).  fp32 statistics, one rounding; gamma / beta may be NULL.
 * Requirements: c % 8 == 0, 8 <= c <= 2048, x / y / gamma / beta 16-byte aligned, rows contiguous.
 * ------------------------------------------------------------------------------------- */
int aid_layernorm(const void* x, const void* gamma, const void* beta, void* y, int64_t rows, int32_t c,
                  float eps, int32_t dtype, void* stream);

/* The same LayerNorm FOLDED into the projections that consume it, so that LayerNorm(x) is never written or re-read:
 *   aid_ln_stats   stats[r] = (mean_r, rsqrt(var_r + eps)), fp32 [rows, 2] — one read of x, same arithmetic as aid_layernorm
 *   aid_ln_fold    once per (weight, gamma, beta): w_folded = w * gamma (storage dtype), colsum[n] = sum_k w_folded[n, k],
 *                  shift[n] = sum_k beta[k] w[n, k]   (fp32 [rows]); w is [rows, c] = torch Linear.weight
 * and AidGemmProblem.ln_* applies  rstd (x W'^T - mean colsum) + shift  in the GEMM epilogue.  Requirements as aid_layernorm. */
int aid_ln_stats(const void* x, float* stats, int64_t rows, int32_t c, float eps, int32_t dtype, void* stream);
int aid_ln_fold(const void* w, const void* gamma, const void* beta, void* w_folded, float* colsum, float* shift,
                int32_t rows, int32_t c, int32_t dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Interpolated attention core on projected tensors.
 *   q   [n_frames, s, heads*d]        row stride ldq, frame stride q_fs   (elements)
 *   k   [n_kv,     l, heads*d]        row stride ldk, frame stride k_fs
 *   vt  [n_kv, heads*d, ldvt]         V TRANSPOSED (channel-major): vt[f][c][key]; ldvt >= l,
 *                                     ldvt % 8 == 0, frame stride vt_fs
 *   out [n_frames, s, heads*d]        row stride ldo, frame stride o_fs
 * Frame i attends with its own keys  K[kv_of(i)]  (kv_of = kv_map[i] if kv_map else i) and with
 * the end-point frames K[begin], K[end]:
 *   PLAIN : O_i = A(Q_i, K_i, V_i)
 *   INNER : Kc = (1-c_i) K[begin] + c_i K[end]  (same for V);
 *           O_i = fused ? A(Q_i, [K_i ; Kc], [V_i ; Vc]) : A(Q_i, Kc, Vc)
 *           Kc / Vc^T of the interior frames (0 < c_i < 1) are read from k2 / vt2, laid out like k / vt
 *           with one row per FRAME (frame strides k_fs / vt_fs); fill them with aid_lerp_kv() first.
 *   OUTER : O_i = (1-c_i) A(Q_i, [K_i;] K[begin], ..) + c_i A(Q_i, [K_i;] K[end], ..)
 * with A(Q,K,V) = softmax(Q K^T * softmax_scale) V per head, c_i = coef[i] (fp32, device).
 * A NEGATIVE coefficient marks frame i as PLAIN inside an INNER / OUTER launch: the unconditional half of a
 * classifier-free-guidance batch [cond frames ; uncond frames] then rides in the same call (begin / end must
 * index the cond half).  n_plain = number of such frames (host-side accounting only, may be 0).
 * Finally   out_i = (accumulate ? out_i : 0) + out_scale * (frame_scale ? frame_scale[i] : 1) * O_i.
 * Supported head dims d: 40, 64, 80, 160 (SD1.5 / SDXL).  No sequence-length restriction.
 * ------------------------------------------------------------------------------------- */
typedef struct AidAttnArgs {
    const void*    q;
    const void*    k;
    const void*    vt;
    void*          out;
    const float*   coef;         /* device [n_frames]; may be NULL for PLAIN                 */
    const float*   frame_scale;  /* device [n_frames] or NULL                                */
    const int32_t* kv_map;       /* device [n_frames] or NULL (identity)                     */
    const void*    k2;           /* INNER only: interpolated keys    [n_frames, l, heads*d]  */
    const void*    vt2;          /* INNER only: interpolated values^T [n_frames, heads*d, ldvt] */
    int32_t n_frames, n_kv;
    int32_t s, l, heads, d;
    int32_t ldq, ldk, ldvt, ldo;
    int64_t q_fs, k_fs, vt_fs, o_fs;
    int32_t mode;                /* AID_MODE_*                                               */
    int32_t fused;               /* 0/1: prepend the frame's own keys/values                 */
    int32_t begin, end;          /* end-point frame indices into k / vt                      */
    int32_t accumulate;          /* 0/1                                                      */
    int32_t dtype;               /* AID_DTYPE_*                                              */
    float   softmax_scale;       /* d^-0.5 for diffusers Attention                           */
    float   out_scale;
    int32_t n_plain;             /* frames with a negative coefficient (profiling accounting) */
    int32_t q_prescaled;         /* 1: q already holds q * softmax_scale * log2(e) (fold it into the     */
                                 /* q-projection via AidGemmProblem.scale); 0: the kernel scales Q itself */
    int32_t seg_executed;        /* profiling accounting: (frame, key segment) passes this launch really runs —  */
                                 /* fused END-POINT frames and coefficient-0/1 sides run fewer than the          */
                                 /* algorithmic count; 0 = unknown (reported equal to the algorithmic count)     */
    int32_t reserved0;
    /* ABI v8 — additive score bias (NULL: none): diffusers' attention_mask after Attention.prepare_attention_mask, which every  */
    /* reference processor hands to get_attention_scores (interpolation.py:115-117, 604-606, 651, 738-739, 787):              */
    /*     scores = softmax_scale * Q K^T + bias        (baddbmm(attention_mask, q, k^T, beta = 1, alpha = scale))              */
    /* bias element (frame f, head h, query row r, key j) at  bias + f * bias_fs + h * bias_hs + r * bias_rs + j  (elements of    */
    /* the operand dtype; bias_hs = 0: one mask for all heads, bias_rs = 0: one row for all queries — the [B * H, 1, L] tensor   */
    /* of prepare_attention_mask is bias_fs = H * L, bias_hs = L, bias_rs = 0).  It covers ONE key segment of l keys: a call     */
    /* whose frames attend to [own ; end-point] keys (fused) is refused (AID_ERR_ARG) — the reference fails there too, at the    */
    /* broadcast of an l-wide mask against 2 l scores.  INNER / OUTER apply it to the begin, end and interpolated segments       */
    /* alike.  Strides are multiples of 1 element; the base pointer needs the operand dtype's alignment only.  v7's kv_padded    */
    /* (a layout promise no kernel depended on) is gone.                                                                         */
    const void* bias;
    int64_t bias_fs;
    int32_t bias_hs, bias_rs;
} AidAttnArgs;

int aid_attn_fwd(const AidAttnArgs* args /* host */, void* stream);

/* Interpolated keys / values for AID_MODE_INNER (reference interpolation.py:772-775):
 *   k2[i] = (1 - coef[i]) * k[begin] + coef[i] * k[end]     (and the same for vt -> vt2)
 * for every frame i with 0 < coef[i] < 1 (other frames are left untouched: the attention kernel reads
 * the end-point frames themselves).  k_fs / vt_fs are the frame strides in elements (multiples of 8)
 * of all four tensors.  One streaming launch. */
int aid_lerp_kv(const void* k, const void* vt, void* k2, void* vt2, const float* coef /* device */,
                int32_t n_frames, int32_t begin, int32_t end, int64_t k_fs, int64_t vt_fs, int32_t dtype,
                void* stream);

/* ---------------------------------------------------------------------------------------
 * One whole processor call (what diffusers' Attention.forward hands to the AID processor):
 *   x   [n_frames, s, c]   hidden states          ctx [n_frames, l, cc] or NULL (self-attn: ctx = x)
 *   wq [c, c]  wk [c, cc]  wv [c, cc]  wo [c, c]  bo [c]      (torch Linear.weight layout [out, in])
 *   y   [n_frames, s, c]   = to_out( AID-attention( to_q(x), to_k(ctx), to_v(ctx) ) )
 * Launches: [1 LayerNorm or 1 row-statistics pass,] 1 grouped GEMM (q, k, V^T [, K_ip, V_ip^T]), [INNER: 1 streaming K/V lerp,] 1 attention
 * kernel [+ 1 for the image branch, accumulating], 1 GEMM (out-proj + bias [+ residual]).
 * Optional block-level fusion: with ln_eps > 0 the call computes on LayerNorm(x) (the block's
 * norm1 / norm2; self-attention keys / values use the normalised x too, a cross-attention ctx is left alone), and
 * with `residual` it returns  residual + to_out(...)  — together  h + attn(norm(h))  in one call.
 * `workspace` must hold aid_processor_workspace_bytes() bytes (16-byte aligned); it is
 * scratch, owned by the caller, and may be reused by the next call on the same stream.
 * ------------------------------------------------------------------------------------- */
typedef struct AidProcessorArgs {
    const void*  x;
    const void*  ctx;            /* NULL => self-attention                                   */
    const void*  wq;
    const void*  wk;
    const void*  wv;
    const void*  wo;
    const void*  bo;             /* NULL = no bias                                           */
    void*        y;
    const float* coef;           /* device [n_frames] (already rounded to the compute dtype, */
                                 /* interpolation.py:663); NULL for PLAIN                    */
    void*        workspace;
    size_t       workspace_bytes;
    int32_t n_frames, s, l, c, cc, heads;
    int32_t mode, fused;
    int32_t begin, end;          /* end-point frames (reference: 0 and n_frames-1)           */
    int32_t dtype;
    int32_t n_ctx;               /* number of ctx frames (n_frames, or fewer with ctx_map)   */
    const int32_t* ctx_map;      /* device [n_frames] frame -> ctx row, or NULL (identity)   */
    int32_t n_plain;             /* frames whose coef is negative (PLAIN riders), accounting  */
    float   ln_eps;              /* > 0: LayerNorm(x) over c first (gamma / beta below)       */
    const void* ln_gamma;        /* [c] or NULL                                               */
    const void* ln_beta;         /* [c] or NULL                                               */
    const void* residual;        /* [n_frames, s, c] or NULL: added to y (may alias x)        */
    /* ---- IP-Adapter image branch (ip == NULL: none).  Image tokens: n_ip rows of [t_ip, cc] (cc = the text   */
    /* context width), row r at ip + r * ip_stride elements (a strided view such as ip_hidden_states[0][::3]    */
    /* needs no copy).  K_ip = to_k_ip(ip), V_ip = to_v_ip(ip) are projected in the same grouped launch as      */
    /* q / k / V^T; frame i uses image row ip_map[i] (NULL = identity, then n_ip == number of AID frames).      */
    /*   AID_IP_SAME       O += ip_scale * [the text mode's interpolation scheme on the image keys]             */
    /*                     (OUTER only: interpolation.py:329-367, outputs are lerped AFTER the sum)             */
    /*   AID_IP_PLAIN      O += ip_scale * A(Q_i, K_ip[row i], V_ip[row i])      (interpolation.py:525-530;     */
    /*                     de-activated IPAdapterAttnProcessor2_0 with the [9,1,T,Cc] fold as t_ip = 3 T)       */
    /* ip_frame_scale (device [n_frames] or NULL) multiplies ip_scale per frame: ScaleControlIPAttnProcessor's     */
    /* `coef * ip_hidden_states` (interpolation.py:146-150, 196) is AID_IP_PLAIN with ip_frame_scale = coef.       */
    const void*    ip;
    const void*    wk_ip;        /* [c, cc] */
    const void*    wv_ip;        /* [c, cc] */
    const int32_t* ip_map;       /* device [n_frames] or NULL */
    const float*   ip_frame_scale; /* device [n_frames] or NULL */
    int64_t ip_stride;           /* elements between consecutive image rows (>= t_ip * cc, multiple of 8) */
    int32_t n_ip, t_ip;
    int32_t ip_mode;             /* AID_IP_*                                                  */
    float   ip_scale;
    int32_t ip_begin, ip_end;    /* AID_IP_SAME: image rows of the end-point frames           */
    int32_t seg_executed;        /* accounting, see AidAttnArgs.seg_executed (text launch only) */
    int32_t reserved2;           /* (v6 / v7: kv_cached_lt, a tile-padded layout of the cached keys that no kernel needed; removed in v8) */
    /* ---- folded LayerNorm (ln_eps > 0 and ln_wq != NULL): the call runs aid_ln_stats on x instead of aid_layernorm and  */
    /* projects x with the folded weights (aid_ln_fold of wq / wk / wv with ln_gamma / ln_beta; wk / wv only for           */
    /* self-attention, a cross-attention ctx is not normalised).  ln_const: fp32 [6, c] = colsum_q, shift_q, colsum_k,     */
    /* shift_k, colsum_v, shift_v.  Needs c % 64 == 0 (the projections then run on the k % 64 == 0 engines).               */
    const void*  ln_wq;
    const void*  ln_wk;
    const void*  ln_wv;
    const float* ln_const;
    /* ---- step-invariant keys / values of a cross-attention layer (ABI v5; both NULL: project them in this call).  The text  */
    /* context does not change over the denoising loop (reference pipeline_interpolated_sd.py:1859-1867 hands the same       */
    /* prompt_embeds to every step), so K = to_k(ctx) [n_ctx, l, c] and V^T = Wv ctx^T [n_ctx, c, round_up(l, 8)] can be      */
    /* projected ONCE per (layer, context) with aid_gemm_nt and handed to every later call: the grouped launch then holds     */
    /* the query projection alone.  Only valid with ctx != NULL; the caller owns the buffers and their validity.             */
    const void*  k_cached;
    const void*  vt_cached;
    int32_t      cu_share;       /* ABI v7: see AidGemmProblem.cu_share — handed to every GEMM launch of the call            */
    int32_t      reserved1;
    /* ABI v8: additive score bias of the text attention (attention_mask), see AidAttnArgs.bias; not with `fused`, not with `ip` */
    const void*  attn_bias;
    int64_t      attn_bias_fs;
    int32_t      attn_bias_hs, attn_bias_rs;
} AidProcessorArgs;

size_t aid_processor_workspace_bytes(const AidProcessorArgs* args /* host */);
int    aid_processor_fwd(const AidProcessorArgs* args /* host */, void* stream);

/* ---------------------------------------------------------------------------------------
 * Live kernel timing (bench.py's roofline leg): between aid_profile_begin() and
 * aid_profile_end() every kernel launched through this library is bracketed by a pair of
 * HIP events recorded on the launch stream.  aid_profile_end() synchronises those events and
 * returns one entry per launch: elapsed milliseconds plus the ALGORITHMIC work of the launch
 * (flops: 2*m*n*k per GEMM problem, 4*s*l*c per (frame, key segment) of attention with the
 * segment count of SURVEY.md §8d: plain 1, pure inner 1, fused inner 2, pure outer 2, fused
 * outer 3; bytes: operands read once + result written once).  Not for use inside stream capture.
 * ------------------------------------------------------------------------------------- */
typedef struct AidProfileEntry {
    char   kernel[64];     /* kernel that ran: "aid_attn<f16,d40,inner,nw4>", "aid_gemm_nt_pp_kernel<bf16>", ... */
    double ms;
    double flops;          /* algorithmic                                                      */
    double bytes;
    double flops_executed; /* what the launch really computes: attention with the executed segment count      */
                           /* (AidAttnArgs.seg_executed), GEMM = algorithmic                                  */
} AidProfileEntry;

int aid_profile_begin(void);
/* returns the number of entries written (<= max_entries) or a negative error code */
int aid_profile_end(AidProfileEntry* entries /* host */, int max_entries);

/* misc */
int         aid_abi_version(void);
const char* aid_strerror(int code);
/* name of the kernel variant the last aid_attn_fwd call on this thread launched (for profiling) */
const char* aid_last_attn_variant(void);
/* same for the last aid_gemm_nt launch: "lockstep128", "pingpong256" (+ "+tail128") or "edge" */
const char* aid_last_gemm_variant(void);
/* Thread safety (ABI v7): every entry point may be called from several host threads at once, each on its own stream with its own
 * workspace.  The tuning table is a set of independent atomic integers (a knob flipped by one thread is seen by the launches of all),
 * the profiling list is guarded by a mutex and collects the launches of every thread between begin and end, the error / variant
 * strings are thread-local. */
/* Development / tuning knobs (kernel-variant choices the launch heuristics normally make: "GEMM_VARIANT", "GEMM_PP",
 * "ATTN_NW", "ATTN_RES", ... — the list is in csrc/aid_kernels.hpp).  The table is filled ONCE when the library is
 * loaded, from environment variables AID_<NAME>; this call changes an entry at run time (value < 0 = back to the
 * heuristic).  Nothing on the launch path reads the environment.  Every value a knob accepts selects among kernels that compute the
 * same result; a value outside a knob's range is refused (AID_ERR_ARG) here and ignored in the environment — no setting can make the
 * library skip work.  Returns AID_ERR_ARG for an unknown name.
 * "CU_SHARE" = n (1 .. 8) is the process-wide default of the per-call hint AidGemmProblem.cu_share / AidProcessorArgs.cu_share (n
 * independent launch streams run side by side; the GEMM engine choice plans with 1 / n of the CUs) for callers that cannot pass it per
 * call; a per-call value > 0 wins.  Results do not depend on it. */
int aid_set_tuning(const char* name, int value);
/* current value of a knob (-1 = the launch heuristics decide) */
int aid_get_tuning(const char* name, int* value);
/* device properties of the current device: returns AID_OK and fills what is non-NULL */
int aid_device_info(int* n_cu, int* clock_khz, char* arch /* >= 32 bytes */);
/* ABI v7.  *id = the id of the stream capture `stream` is part of (hipStreamGetCaptureInfo), 0 when it is not capturing.  Callers that
 * own scratch memory use it to give every capture its own (the Python layer never allocates a workspace inside a capture). */
int aid_stream_capture_id(void* stream, unsigned long long* id);

#ifdef __cplusplus
}
#endif
#endif /* AID_HIP_H */
