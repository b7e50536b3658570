// extern "C" boundary of libaid_hip.so (see include/aid_hip.h).  Argument checking, workspace
// carving and launch sequencing only — every byte of arithmetic happens in the gfx950 kernels of
// aid_gemm.hip / aid_attn.hip.  No allocation, no synchronisation, no exceptions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "aid_kernels.hpp"

namespace {

thread_local char g_err[256] = "";
thread_local const char* g_variant = "";
thread_local const char* g_gemm_variant = "";

int fail_hip(hipError_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return AID_ERR_LAUNCH;
}

// ---- live timing (aid_profile_begin / aid_profile_end) -------------------------------------------
struct ProfRec {
    hipEvent_t t0, t1;
    char name[64];
    double flops, bytes, flops_exec;
};
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;                      // guards g_prof: launches of several host threads may be profiled at once
std::vector<ProfRec> g_prof;
unsigned g_prof_gen = 0;                   // bumped by aid_profile_begin (under g_prof_mu): a ProfScope that straddles a restart of the
                                           // recorder must not write into the NEW list's record that happens to sit at its index

struct ProfScope {          // records an event pair around one launch when profiling is on
    hipStream_t stream;
    bool on;
    size_t idx = 0;         // this launch's record (other threads may append while the launch is being issued)
    unsigned gen = 0;       // recorder generation the record belongs to
    ProfScope(hipStream_t s, const char* name, double flops, double bytes, double flops_exec = -1.0)
        : stream(s), on(g_prof_on.load(std::memory_order_relaxed)) {
        if (!on) return;
        ProfRec r;
        (void)hipEventCreate(&r.t0);
        (void)hipEventCreate(&r.t1);
        snprintf(r.name, sizeof(r.name), "%s", name);
        r.flops = flops;
        r.bytes = bytes;
        r.flops_exec = flops_exec < 0.0 ? flops : flops_exec;
        (void)hipEventRecord(r.t0, stream);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        idx = g_prof.size();
        gen = g_prof_gen;
        g_prof.push_back(r);
    }
    void rename(const char* name) {      // the kernel symbol is known only after the launcher picked an engine
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (gen == g_prof_gen && idx < g_prof.size()) snprintf(g_prof[idx].name, sizeof(g_prof[idx].name), "%s", name);
    }
    ~ProfScope() {
        if (!on) return;
        hipEvent_t t1 = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_prof_mu);
            if (gen == g_prof_gen && idx < g_prof.size()) t1 = g_prof[idx].t1;
        }
        if (t1) (void)hipEventRecord(t1, stream);
    }
};

// ---- tuning knobs (aid_kernels.hpp: enum Tune) -----------------------------------------------------
const char* const g_tune_names[aid::TUNE_COUNT] = {"GEMM_VARIANT", "GEMM_PP", "GEMM_TRI", "ATTN_NW", "ATTN_QB", "ATTN_PIPE",
                                                   "ATTN_RES", "ATTN_RES_CHUNKS", "ATTN_ORDER", "ATTN_V2", "CU_SHARE", "GEMM_RS", "ATTN_TX", "ATTN_TX_TILES", "GEMM_LS"};
// Largest value a knob accepts.  Every accepted value selects between kernels / launch shapes that compute THE SAME RESULT (the parity
// suite runs under each of them); values beyond the range are refused by aid_set_tuning and ignored in the environment.  The timing
// ablations (kernels that skip work, "results are garbage") exist only in development builds (-DAID_ABLATIONS, tools/dev/Makefile ->
// tools/dev/libaid_abl.so) and are addressed through the same table there.
#ifdef AID_ABLATIONS
const int g_tune_max[aid::TUNE_COUNT] = {31, 15, 1, 8, 2, 1, 1, 1000, 1, 1, 8, 1, 1, 64, 1};
#else
#if defined(AID_RS_VARIANTS) || defined(AID_PPX_ORDERS)
const int g_tune_max[aid::TUNE_COUNT] = {31, 7, 1, 8, 2, 1, 1, 64, 1, 1, 8, 1, 1, 64, 1};
#else
const int g_tune_max[aid::TUNE_COUNT] = {31, 3, 1, 8, 2, 1, 1, 64, 1, 1, 8, 1, 1, 64, 1};
#endif
#endif
struct TuneTable {
    std::atomic<int> v[aid::TUNE_COUNT];        // independent integers: a knob flipped by one thread is seen by the launches of all
    TuneTable() {                               // runs once, when the library is loaded
        for (int i = 0; i < aid::TUNE_COUNT; ++i) {
            char name[64];
            snprintf(name, sizeof(name), "AID_%s", g_tune_names[i]);
            const char* e = getenv(name);
            const int x = (e && *e) ? atoi(e) : -1;
            v[i].store((x < 0 || x > g_tune_max[i]) ? -1 : x, std::memory_order_relaxed);
        }
    }
};
TuneTable g_tune;

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int round_up(int x, int a) { return (x + a - 1) / a * a; }
inline bool dtype16(int d) { return d == AID_DTYPE_F16 || d == AID_DTYPE_BF16; }
inline bool dtype_ok(int d) { return dtype16(d) || d == AID_DTYPE_F32; }
inline const char* dtype_name(int d) { return d == AID_DTYPE_F16 ? "f16" : d == AID_DTYPE_BF16 ? "bf16" : "f32"; }

int check_problem(const AidGemmProblem& q) {
    if (!q.a || !q.b || !q.c) return AID_ERR_ARG;
    if (q.m < 0 || q.n < 0 || q.k <= 0 || q.batch < 1) return AID_ERR_ARG;
    if (q.k % 8 || q.lda % 8 || q.ldb % 8 || q.ldc % 4) return AID_ERR_SHAPE;
    if (q.lda < q.k || q.ldb < q.k || (!q.trans_rows && q.ldc < round_up(q.n, 4))) return AID_ERR_SHAPE;
    if (!aligned16(q.a) || !aligned16(q.b) || !aligned16(q.c)) return AID_ERR_SHAPE;
    if ((q.stride_a % 8) || (q.stride_b % 8) || (q.stride_c % 4)) return AID_ERR_SHAPE;
    if (q.trans_rows < 0) return AID_ERR_ARG;
    if (q.trans_rows) {                                  // C transposed per frame (aid_hip.h)
        if (q.batch != 1 || q.bias || q.residual || q.m % q.trans_rows || (q.ln_stats && q.ln_side != 1)) return AID_ERR_ARG;
        if (q.trans_rows % 8 || q.ldc % 8 || q.ldc < q.trans_rows || q.stride_c % 8) return AID_ERR_SHAPE;
    }
    return AID_OK;
}

struct Carve {
    size_t q, k, vt, o, k2, vt2, xn, kip, vtip, total;
    int lp, tp;
};

Carve carve(const AidProcessorArgs& a) {
    Carve c;
    const size_t es = a.dtype == AID_DTYPE_F32 ? 4 : 2;
    const int nctx = a.ctx ? a.n_ctx : a.n_frames;
    const int l = a.ctx ? a.l : a.s;
    c.lp = round_up(l, 8);
    size_t off = 0;
    c.q = off;  off += align_up((size_t)a.n_frames * a.s * a.c * es, 256);
    c.k = off;  off += align_up((size_t)nctx * l * a.c * es, 256);
    c.vt = off; off += align_up((size_t)nctx * a.c * c.lp * es, 256);
    c.o = off;  off += align_up((size_t)a.n_frames * a.s * a.c * es, 256);
    c.k2 = c.vt2 = off;
    if (a.mode == AID_MODE_INNER) {                       // interpolated K / V^T, one row per frame
        c.k2 = off;  off += align_up((size_t)a.n_frames * l * a.c * es, 256);
        c.vt2 = off; off += align_up((size_t)a.n_frames * a.c * c.lp * es, 256);
    }
    c.xn = off;
    if (a.ln_eps > 0.f)                                   // LayerNorm(x), or only its row statistics when folded
        off += align_up(a.ln_wq ? (size_t)a.n_frames * a.s * 2 * sizeof(float) : (size_t)a.n_frames * a.s * a.c * es, 256);
    c.kip = c.vtip = off;
    c.tp = round_up(a.ip ? a.t_ip : 0, 8);
    if (a.ip) {                                           // image keys / values^T of the IP-Adapter branch
        c.kip = off;  off += align_up((size_t)a.n_ip * a.t_ip * a.c * es, 256);
        c.vtip = off; off += align_up((size_t)a.n_ip * a.c * c.tp * es, 256);
    }
    c.total = off;
    return c;
}

int check_processor(const AidProcessorArgs& a) {
    if (!a.x || !a.wq || !a.wk || !a.wv || !a.wo || !a.y) return AID_ERR_ARG;
    if (!dtype_ok(a.dtype)) return AID_ERR_DTYPE;
    if (a.dtype == AID_DTYPE_F32 && a.ln_eps > 0.f) return AID_ERR_DTYPE;      // the LayerNorm fusion is 16-bit only
    if (a.n_frames < 1 || a.s < 1 || a.c < 1 || a.heads < 1 || a.c % a.heads) return AID_ERR_ARG;
    if (a.mode < AID_MODE_PLAIN || a.mode > AID_MODE_OUTER) return AID_ERR_ARG;
    if (a.mode != AID_MODE_PLAIN && !a.coef) return AID_ERR_ARG;
    if (!aid::attn_head_dim_supported(a.c / a.heads)) return AID_ERR_SHAPE;
    if (a.c % 8) return AID_ERR_SHAPE;
    const int nctx = a.ctx ? a.n_ctx : a.n_frames;
    if (a.ctx) {
        if (a.l < 1 || a.cc < 8 || a.cc % 8 || a.n_ctx < 1) return AID_ERR_ARG;
        if (!a.ctx_map && a.n_ctx != a.n_frames) return AID_ERR_ARG;
    } else if (a.ctx_map) {
        return AID_ERR_ARG;
    }
    if (a.mode != AID_MODE_PLAIN && (a.begin < 0 || a.begin >= nctx || a.end < 0 || a.end >= nctx)) return AID_ERR_ARG;
    if (a.ip) {
        if (!a.ctx || !a.wk_ip || !a.wv_ip || a.n_ip < 1 || a.t_ip < 1) return AID_ERR_ARG;
        if (a.ip_mode != AID_IP_SAME && a.ip_mode != AID_IP_PLAIN) return AID_ERR_ARG;
        if (a.ip_mode == AID_IP_SAME && (a.mode != AID_MODE_OUTER || a.ip_begin < 0 || a.ip_begin >= a.n_ip ||
                                         a.ip_end < 0 || a.ip_end >= a.n_ip)) return AID_ERR_ARG;
        if (!a.ip_map && a.n_ip < a.n_frames) return AID_ERR_ARG;
        if (a.ip_stride % 8 || a.ip_stride < (int64_t)a.t_ip * a.cc) return AID_ERR_SHAPE;
        if (!aligned16(a.ip) || !aligned16(a.wk_ip) || !aligned16(a.wv_ip)) return AID_ERR_SHAPE;
    } else if (a.ip_mode != AID_IP_NONE) {
        return AID_ERR_ARG;
    }
    if ((a.k_cached != nullptr) != (a.vt_cached != nullptr)) return AID_ERR_ARG;
    if (a.k_cached && (!a.ctx || !aligned16(a.k_cached) || !aligned16(a.vt_cached))) return AID_ERR_ARG;
    if (a.attn_bias && ((a.fused && a.mode != AID_MODE_PLAIN) || a.ip || a.attn_bias_fs < 0 || a.attn_bias_hs < 0 || a.attn_bias_rs < 0))
        return AID_ERR_ARG;                            // the mask covers one key segment (aid_hip.h); the image branch has no mask to take
    if (!(a.ln_eps >= 0.f) || a.cu_share < 0 || a.cu_share > 8) return AID_ERR_ARG;
    if (a.ln_eps > 0.f) {
        if (!aid::layernorm_width_supported(a.c)) return AID_ERR_SHAPE;
        if ((a.ln_gamma && !aligned16(a.ln_gamma)) || (a.ln_beta && !aligned16(a.ln_beta))) return AID_ERR_SHAPE;
        if (a.ln_wq) {                              // folded: the projections run on x itself with pre-multiplied weights
            if (!a.ln_const || a.c % 64 || !aligned16(a.ln_wq)) return AID_ERR_SHAPE;
            if (!a.ctx && (!a.ln_wk || !a.ln_wv || !aligned16(a.ln_wk) || !aligned16(a.ln_wv))) return AID_ERR_ARG;
        }
    }
    return AID_OK;
}

}  // namespace

namespace aid {
int tune(int id) { return (id >= 0 && id < TUNE_COUNT) ? g_tune.v[id].load(std::memory_order_relaxed) : -1; }
}  // namespace aid

extern "C" {

int aid_abi_version(void) { return AID_ABI_VERSION; }

int aid_set_tuning(const char* name, int value) {
    if (!name) return AID_ERR_ARG;
    if (!strncmp(name, "AID_", 4)) name += 4;
    for (int i = 0; i < aid::TUNE_COUNT; ++i)
        if (!strcmp(name, g_tune_names[i])) {
            if (value > g_tune_max[i]) return AID_ERR_ARG;      // no value may change results (see g_tune_max)
            g_tune.v[i].store(value < 0 ? -1 : value, std::memory_order_relaxed);
            return AID_OK;
        }
    return AID_ERR_ARG;
}

int aid_get_tuning(const char* name, int* value) {
    if (!name || !value) return AID_ERR_ARG;
    if (!strncmp(name, "AID_", 4)) name += 4;
    for (int i = 0; i < aid::TUNE_COUNT; ++i)
        if (!strcmp(name, g_tune_names[i])) {
            *value = g_tune.v[i].load(std::memory_order_relaxed);
            return AID_OK;
        }
    return AID_ERR_ARG;
}

const char* aid_strerror(int code) {
    switch (code) {
        case AID_OK: return "ok";
        case AID_ERR_ARG: return "invalid argument (NULL pointer, negative size or inconsistent field)";
        case AID_ERR_DTYPE: return "unsupported dtype (AID_DTYPE_F16 / _BF16 / _F32; the LayerNorm entry points and the ln_* options of aid_processor_fwd are 16-bit only)";
        case AID_ERR_SHAPE: return "unsupported shape or alignment (head dim must be 40/64/80/160; see aid_hip.h)";
        case AID_ERR_WORKSPACE: return "workspace too small or misaligned";
        case AID_ERR_LAUNCH: return g_err[0] ? g_err : "kernel launch failed";
        case AID_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown error code";
    }
}

const char* aid_last_attn_variant(void) { return g_variant; }
const char* aid_last_gemm_variant(void) { return g_gemm_variant; }

int aid_device_info(int* n_cu, int* clock_khz, char* arch) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return AID_ERR_NO_DEVICE;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return AID_ERR_NO_DEVICE;
    if (n_cu) *n_cu = pr.multiProcessorCount;
    if (clock_khz) *clock_khz = pr.clockRate;
    if (arch) { strncpy(arch, pr.gcnArchName, 31); arch[31] = 0; }
    return AID_OK;
}

int aid_stream_capture_id(void* stream, unsigned long long* id) {
    if (!id) return AID_ERR_ARG;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    const hipError_t e = hipStreamGetCaptureInfo(static_cast<hipStream_t>(stream), &st, &cid);
    if (e != hipSuccess) return fail_hip(e, "aid_stream_capture_id");
    *id = st == hipStreamCaptureStatusActive ? cid : 0ull;
    return AID_OK;
}

int aid_profile_begin(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.t0); (void)hipEventDestroy(r.t1); }
    g_prof.clear();
    ++g_prof_gen;
    g_prof_on.store(true);
    return AID_OK;
}

int aid_profile_end(AidProfileEntry* entries, int max_entries) {
    g_prof_on.store(false);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    int rc = AID_OK;
    for (auto& r : g_prof) {
        hipError_t e = hipEventSynchronize(r.t1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.t0, r.t1);
        if (e != hipSuccess) rc = fail_hip(e, "aid_profile_end");
        if (entries && n < max_entries && e == hipSuccess) {
            memcpy(entries[n].kernel, r.name, sizeof(r.name));
            entries[n].ms = ms;
            entries[n].flops = r.flops;
            entries[n].bytes = r.bytes;
            entries[n].flops_executed = r.flops_exec;
            ++n;
        }
        (void)hipEventDestroy(r.t0);
        (void)hipEventDestroy(r.t1);
    }
    g_prof.clear();
    return rc == AID_OK ? n : rc;
}

int aid_gemm_nt(const AidGemmProblem* problems, int n_problems, int dtype, void* stream) {
    if (!problems || n_problems < 1 || n_problems > AID_GEMM_MAX_PROBLEMS) return AID_ERR_ARG;
    if (!dtype_ok(dtype)) return AID_ERR_DTYPE;
    if (problems[0].cu_share < 0 || problems[0].cu_share > 8) return AID_ERR_ARG;
    aid::GemmGroup g;
    memset(&g, 0, sizeof(g));
    g.n_problems = n_problems;
    for (int i = 0; i < n_problems; ++i) {
        const AidGemmProblem& q = problems[i];
        int rc = check_problem(q);
        if (rc != AID_OK) return rc;
        aid::GemmDesc& d = g.p[i];
        d.a = q.a; d.b = q.b; d.c = q.c; d.bias = q.bias; d.residual = q.residual;
        d.m = q.m; d.n = q.n; d.k = q.k;
        d.lda = q.lda; d.ldb = q.ldb; d.ldc = q.ldc;
        d.batch = q.batch;
        d.scale = q.scale == 0.f ? 1.f : q.scale;
        d.stride_a = q.stride_a; d.stride_b = q.stride_b; d.stride_c = q.stride_c;
        d.trans_rows = q.trans_rows;
        if (q.ln_stats) {
            if (!q.ln_colsum || !q.ln_shift || (q.ln_side != 1 && q.ln_side != 2) || q.stride_stats < 0) return AID_ERR_ARG;
            d.ln_stats = q.ln_stats; d.ln_colsum = q.ln_colsum; d.ln_shift = q.ln_shift;
            d.ln_side = q.ln_side; d.stride_stats = q.stride_stats;
        }
    }
    double flops = 0, bytes = 0;
    if (g_prof_on.load(std::memory_order_relaxed)) {
        for (int i = 0; i < n_problems; ++i) {
            const AidGemmProblem& q = problems[i];
            flops += 2.0 * q.m * q.n * q.k * q.batch;
            bytes += 2.0 * ((double)q.m * q.k * (q.stride_a || q.batch == 1 ? q.batch : 1) +
                            (double)q.n * q.k * (q.stride_b || q.batch == 1 ? q.batch : 1) +
                            (double)q.m * q.n * q.batch * (q.residual ? 2 : 1));
        }
    }
    hipError_t e;
    {
        ProfScope ps(static_cast<hipStream_t>(stream), "aid_gemm_nt", flops, dtype == AID_DTYPE_F32 ? 2.0 * bytes : bytes);
        if (dtype == AID_DTYPE_F32) {
            e = aid::gemm_f32_launch(g, static_cast<hipStream_t>(stream));
            g_gemm_variant = "f32";
            ps.rename("aid_gemm_f32_kernel");
        } else {
            e = aid::gemm_group_launch(g, dtype, static_cast<hipStream_t>(stream), &g_gemm_variant, problems[0].cu_share);
            // profile entries carry the kernel SYMBOL that ran, so they line up with rocprofv3's per-kernel rows
            char nm[64];
            const char* sym = !strncmp(g_gemm_variant, "rowstat", 7)      ? "aid_gemm_rs_kernel"
                              : !strncmp(g_gemm_variant, "pingpong288", 11) ? "aid_gemm_nt_ppx_kernel"
                              : !strncmp(g_gemm_variant, "pingpong", 8)   ? "aid_gemm_nt_pp_kernel"
                              : !strcmp(g_gemm_variant, "edge")           ? "aid_gemm_nt_kernel"
                                                                          : "aid_gemm_nt_pipe_kernel";
            snprintf(nm, sizeof(nm), "%s<%s>", sym, dtype_name(dtype));
            ps.rename(nm);
        }
    }
    return e == hipSuccess ? AID_OK : fail_hip(e, "aid_gemm_nt");
}

int aid_layernorm(const void* x, const void* gamma, const void* beta, void* y, int64_t rows, int32_t c, float eps,
                  int32_t dtype, void* stream) {
    if (!x || !y || rows < 0 || !(eps >= 0.f)) return AID_ERR_ARG;
    if (dtype != AID_DTYPE_F16 && dtype != AID_DTYPE_BF16) return AID_ERR_DTYPE;
    if (!aid::layernorm_width_supported(c)) return AID_ERR_SHAPE;
    if (!aligned16(x) || !aligned16(y) || (gamma && !aligned16(gamma)) || (beta && !aligned16(beta))) return AID_ERR_SHAPE;
    hipError_t e;
    {
        ProfScope ps(static_cast<hipStream_t>(stream), dtype == AID_DTYPE_F16 ? "aid_layernorm<f16>" : "aid_layernorm<bf16>",
                     8.0 * rows * c, 4.0 * rows * c);
        e = aid::layernorm_launch(x, gamma, beta, y, rows, c, eps, dtype, static_cast<hipStream_t>(stream));
    }
    return e == hipSuccess ? AID_OK : fail_hip(e, "aid_layernorm");
}

int aid_ln_stats(const void* x, float* stats, int64_t rows, int32_t c, float eps, int32_t dtype, void* stream) {
    if (!x || !stats || rows < 0 || !(eps >= 0.f)) return AID_ERR_ARG;
    if (dtype != AID_DTYPE_F16 && dtype != AID_DTYPE_BF16) return AID_ERR_DTYPE;
    if (!aid::layernorm_width_supported(c) || !aligned16(x)) return AID_ERR_SHAPE;
    hipError_t e;
    {
        ProfScope ps(static_cast<hipStream_t>(stream), dtype == AID_DTYPE_F16 ? "aid_ln_stats<f16>" : "aid_ln_stats<bf16>",
                     4.0 * rows * c, 2.0 * rows * c + 8.0 * rows);
        e = aid::ln_stats_launch(x, stats, rows, c, eps, dtype, static_cast<hipStream_t>(stream));
    }
    return e == hipSuccess ? AID_OK : fail_hip(e, "aid_ln_stats");
}

int aid_ln_fold(const void* w, const void* gamma, const void* beta, void* w_folded, float* colsum, float* shift,
                int32_t rows, int32_t c, int32_t dtype, void* stream) {
    if (!w || !w_folded || !colsum || !shift || rows < 1) return AID_ERR_ARG;
    if (dtype != AID_DTYPE_F16 && dtype != AID_DTYPE_BF16) return AID_ERR_DTYPE;
    if (!aid::layernorm_width_supported(c)) return AID_ERR_SHAPE;
    if (!aligned16(w) || !aligned16(w_folded) || (gamma && !aligned16(gamma)) || (beta && !aligned16(beta))) return AID_ERR_SHAPE;
    const hipError_t e = aid::ln_fold_launch(w, gamma, beta, w_folded, colsum, shift, rows, c, dtype,
                                             static_cast<hipStream_t>(stream));
    return e == hipSuccess ? AID_OK : fail_hip(e, "aid_ln_fold");
}

int aid_attn_fwd(const AidAttnArgs* args, void* stream) {
    if (!args) return AID_ERR_ARG;
    const AidAttnArgs& a = *args;
    if (!a.q || !a.k || !a.vt || !a.out) return AID_ERR_ARG;
    if (!dtype_ok(a.dtype)) return AID_ERR_DTYPE;
    if (a.mode < AID_MODE_PLAIN || a.mode > AID_MODE_OUTER) return AID_ERR_ARG;
    if (a.n_frames < 1 || a.n_kv < 1 || a.s < 1 || a.l < 1 || a.heads < 1) return AID_ERR_ARG;
    if (a.mode != AID_MODE_PLAIN && (!a.coef || a.begin < 0 || a.begin >= a.n_kv || a.end < 0 || a.end >= a.n_kv))
        return AID_ERR_ARG;
    if (!a.kv_map && a.n_kv < a.n_frames) return AID_ERR_ARG;
    if (a.mode == AID_MODE_INNER && (!a.k2 || !a.vt2 || !aligned16(a.k2) || !aligned16(a.vt2))) return AID_ERR_ARG;
    if (!aid::attn_head_dim_supported(a.d)) return AID_ERR_SHAPE;
    if (a.ldq % 8 || a.ldk % 8 || a.ldvt % 8 || a.ldo % 4 || a.ldvt < a.l) return AID_ERR_SHAPE;
    if (a.q_fs % 8 || a.k_fs % 8 || a.vt_fs % 8 || a.o_fs % 4) return AID_ERR_SHAPE;
    if (!aligned16(a.q) || !aligned16(a.k) || !aligned16(a.vt) || !aligned16(a.out)) return AID_ERR_SHAPE;
    if (a.bias) {
        // one mask row covers ONE segment of l keys: the reference's fused calls fail at the broadcast of the l-wide mask against the
        // 2 l scores of [own ; end-point] keys (interpolation.py:644-651), so there is nothing to be equal to
        if (a.fused && a.mode != AID_MODE_PLAIN) return AID_ERR_ARG;
        if (a.bias_fs < 0 || a.bias_hs < 0 || a.bias_rs < 0) return AID_ERR_ARG;
        if (reinterpret_cast<uintptr_t>(a.bias) % (a.dtype == AID_DTYPE_F32 ? 4 : 2)) return AID_ERR_SHAPE;
    }
    // algorithmic work: key segments per frame — plain 1, pure inner 1, fused inner 2,
    // pure outer 2, fused outer 3
    const int segs = a.mode == AID_MODE_PLAIN ? 1 : (a.mode == AID_MODE_INNER ? 1 : 2) + (a.fused ? 1 : 0);
    const double c = (double)a.heads * a.d;
    const double per_seg = 4.0 * a.s * (double)a.l * c;
    const double flops = per_seg * ((double)segs * (a.n_frames - a.n_plain) + a.n_plain);
    const double flops_exec = a.seg_executed > 0 ? per_seg * a.seg_executed : flops;
    const double bytes = 2.0 * (2.0 * a.n_frames * a.s * c + 2.0 * a.n_kv * a.l * c);
    hipError_t e = hipSuccess;
    if (a.dtype == AID_DTYPE_F32) {             // float32 storage: one correctness-first kernel for every mode (aid_f32.hip)
        char nm[64];
        snprintf(nm, sizeof(nm), "aid_attn_f32<d%d,%s>", a.d, a.mode == AID_MODE_PLAIN ? "plain" : a.mode == AID_MODE_INNER ? "inner" : "outer");
        {
            ProfScope ps(static_cast<hipStream_t>(stream), nm, flops, 2.0 * bytes, flops_exec);
            e = aid::attn_f32_launch(a, static_cast<hipStream_t>(stream));
        }
        g_variant = "aid_attn_f32";
        return e == hipSuccess ? AID_OK : fail_hip(e, "aid_attn_fwd");
    }
    // d = 64, whole key tiles: the ping-pong kernel (aid_attn_pp.hip).  It runs every kind of frame — one key segment (PLAIN, riders,
    // fused end points), two (fused INNER, one-sided OUTER), three (fused OUTER) — deciding per frame ON THE DEVICE from the
    // coefficients like aid_attn_kernel; the host-side counts below only attribute the work.
    //   default: calls it can run ALONE (several segments per frame: multiples of 512 keys), fused OUTER and INNER from 1024 keys, PLAIN
    //   and pure OUTER from 2048 — same-process A/B, us: S = 4096 plain 648 -> 588, outer 1160 -> 1078, inner 915 -> 860; S = 1024 outer
    //   163.5 -> 161.1, inner 130.6 -> 135.3, plain 93 -> 110 (a 16-tile stream on one workgroup per CU; profiles/r03_attn_notes.txt).
    //   ATTN_V2 = 0 never; 1 wherever supported (tests) — a call it cannot run alone is then split: single-segment frames here, the
    //   others on aid_attn_kernel in a second launch.
    const int n_single = a.mode == AID_MODE_PLAIN ? a.n_frames : a.n_plain + ((a.fused && a.n_frames - a.n_plain >= 2) ? 2 : 0);
    // (a call with several segments per frame: segments of whole 8-tile trips; INNER: k2 / vt2 present)
    const bool alone = a.mode == AID_MODE_PLAIN || (a.l % 512 == 0 && (a.mode == AID_MODE_OUTER || (a.k2 && a.vt2)));
    const bool dflt = a.l >= ((a.mode == AID_MODE_PLAIN || (a.mode == AID_MODE_OUTER && !a.fused)) ? 2048 : 1024);
    const int v2 = aid::tune(aid::TUNE_ATTN_V2);
    // text keys (<= 96 per segment: the cross-attention of the SDXL stack): every segment resident in LDS, independent waves, online
    // softmax over the segment's <= 3 score tiles, OUTER sides combined from the segments' maxima and row sums (aid_attn_tx.hip).  The
    // default wherever it applies (profiles/r05_attn_tx_notes.txt); ATTN_TX = 0 keeps these calls on aid_attn_kernel.  (Round 4's
    // short-stream ping-pong kernel, aid_attn_xs.hip, measured 5 - 20 % slower than aid_attn_kernel and was removed in round 5.)
    if (!a.bias && aid::tune(aid::TUNE_ATTN_TX) != 0 && v2 != 1 && aid::attn_tx_supported(a)) {
        char nm[64];
        static const char* const mn[] = {"plain", "inner", "outer"};
        snprintf(nm, sizeof(nm), "aid_attn_tx<%s,d64,%s>", a.dtype == AID_DTYPE_F16 ? "f16" : "bf16", mn[a.mode]);
        {
            ProfScope ps(static_cast<hipStream_t>(stream), nm, flops, bytes, flops_exec);
            e = aid::attn_tx_launch(a, static_cast<hipStream_t>(stream));
        }
        g_variant = a.mode == AID_MODE_OUTER ? "aid_attn_tx<d64,outer>" : a.mode == AID_MODE_INNER ? "aid_attn_tx<d64,inner>" : "aid_attn_tx<d64,plain>";
        return e == hipSuccess ? AID_OK : fail_hip(e, "aid_attn_fwd");
    }
    const bool use_pp = !a.bias && aid::attn_pp_supported(a) && (alone || n_single > 0) &&   // (a score bias: aid_attn_kernel's BIAS instantiation)
                        (v2 == 1 || (v2 < 0 && alone && dflt));
    if (use_pp) {
        char nm[64];
        snprintf(nm, sizeof(nm), "aid_attn_pp<%s,d64%s>", a.dtype == AID_DTYPE_F16 ? "f16" : "bf16",
                 a.mode == AID_MODE_PLAIN ? "" : !alone ? ",riders" : a.mode == AID_MODE_OUTER ? ",outer" : ",inner");
        const double f1 = alone ? flops : per_seg * n_single;
        ProfScope ps(static_cast<hipStream_t>(stream), nm, f1, alone ? bytes : bytes * n_single / a.n_frames,
                     alone ? flops_exec : f1);
        e = aid::attn_pp_launch(a, static_cast<hipStream_t>(stream), alone);
        g_variant = !alone || a.mode == AID_MODE_PLAIN ? "aid_attn_pp<d64>" : a.mode == AID_MODE_OUTER ? "aid_attn_pp<d64,outer>" : "aid_attn_pp<d64,inner>";
    }
    if (e == hipSuccess && !(use_pp && alone)) {
        const char* nm = aid::attn_variant_name(a);
        const double fa = use_pp ? flops - per_seg * n_single : flops;
        const double fx = use_pp ? flops_exec - per_seg * n_single : flops_exec;
        ProfScope ps(static_cast<hipStream_t>(stream), nm, fa, use_pp ? bytes * (a.n_frames - n_single) / a.n_frames : bytes, fx);
        e = aid::attn_launch(a, static_cast<hipStream_t>(stream), &g_variant, use_pp);
    }
    return e == hipSuccess ? AID_OK : fail_hip(e, "aid_attn_fwd");
}

static int lerp_kv_impl(const void* k, const void* vt, void* k2, void* vt2, const float* coef, int32_t n_frames,
                        int32_t begin, int32_t end, int64_t k_fs, int64_t vt_fs, int32_t dtype, void* stream,
                        int n_interior) {
    if (!k || !vt || !k2 || !vt2 || !coef || n_frames < 1 || begin < 0 || end < 0) return AID_ERR_ARG;
    if (!dtype_ok(dtype)) return AID_ERR_DTYPE;
    if (k_fs % 8 || vt_fs % 8 || !aligned16(k) || !aligned16(vt) || !aligned16(k2) || !aligned16(vt2)) return AID_ERR_SHAPE;
    hipError_t e;
    {
        // algorithmic traffic: the two end-point frames read once, one interpolated row written per interior frame
        const double elems = (double)(k_fs + vt_fs);
        ProfScope ps(static_cast<hipStream_t>(stream), dtype == AID_DTYPE_F16 ? "aid_lerp_kv<f16>" : dtype == AID_DTYPE_BF16 ? "aid_lerp_kv<bf16>" : "aid_lerp_kv<f32>",
                     3.0 * n_interior * elems, (dtype == AID_DTYPE_F32 ? 4.0 : 2.0) * (2.0 + n_interior) * elems);
        if (dtype == AID_DTYPE_F32)
            e = aid::lerp_kv_f32_launch(k, vt, k2, vt2, coef, n_frames, begin, end, k_fs, vt_fs, static_cast<hipStream_t>(stream));
        else
            e = aid::lerp_kv_launch(k, vt, k2, vt2, coef, n_frames, begin, end, k_fs, vt_fs, dtype,
                                    static_cast<hipStream_t>(stream));
    }
    return e == hipSuccess ? AID_OK : fail_hip(e, "aid_lerp_kv");
}

int aid_lerp_kv(const void* k, const void* vt, void* k2, void* vt2, const float* coef, int32_t n_frames, int32_t begin,
                int32_t end, int64_t k_fs, int64_t vt_fs, int32_t dtype, void* stream) {
    return lerp_kv_impl(k, vt, k2, vt2, coef, n_frames, begin, end, k_fs, vt_fs, dtype, stream,
                        n_frames > 2 ? n_frames - 2 : 0);
}

size_t aid_processor_workspace_bytes(const AidProcessorArgs* args) {
    if (!args || check_processor(*args) != AID_OK) return 0;
    return carve(*args).total;
}

int aid_processor_fwd(const AidProcessorArgs* args, void* stream) {
    if (!args) return AID_ERR_ARG;
    const AidProcessorArgs& a = *args;
    int rc = check_processor(a);
    if (rc != AID_OK) return rc;
    const Carve cv = carve(a);
    if (!a.workspace || a.workspace_bytes < cv.total || !aligned16(a.workspace)) return AID_ERR_WORKSPACE;
    char* ws = static_cast<char*>(a.workspace);
    void* q = ws + cv.q;
    const bool cached = a.k_cached != nullptr;           // step-invariant text keys / values: projected once by the caller
    const void* k = cached ? a.k_cached : ws + cv.k;
    const void* vt = cached ? a.vt_cached : ws + cv.vt;
    void* o = ws + cv.o;
    const bool cross = a.ctx != nullptr;
    const void* xin = a.x;
    const bool folded = a.ln_eps > 0.f && a.ln_wq != nullptr;
    float* stats = reinterpret_cast<float*>(ws + cv.xn);
    if (folded) {                              // 0. the block's norm1 / norm2, folded: only the row statistics are computed
        rc = aid_ln_stats(a.x, stats, (int64_t)a.n_frames * a.s, a.c, a.ln_eps, a.dtype, stream);
        if (rc != AID_OK) return rc;
    } else if (a.ln_eps > 0.f) {               // 0. ... or as its own pass
        void* xn = ws + cv.xn;
        rc = aid_layernorm(a.x, a.ln_gamma, a.ln_beta, xn, (int64_t)a.n_frames * a.s, a.c, a.ln_eps, a.dtype, stream);
        if (rc != AID_OK) return rc;
        xin = xn;
    }
    const void* e = cross ? a.ctx : xin;
    const int nctx = cross ? a.n_ctx : a.n_frames;
    const int l = cross ? a.l : a.s;
    const int cc = cross ? a.cc : a.c;
    const int d = a.c / a.heads;

    // 1. q, k and V^T projections in one grouped launch
    AidGemmProblem pr[5];
    memset(pr, 0, sizeof(pr));
    pr[0].a = xin;  pr[0].b = a.wq; pr[0].c = q;
    pr[0].m = a.n_frames * a.s; pr[0].n = a.c; pr[0].k = a.c;
    pr[0].lda = a.c; pr[0].ldb = a.c; pr[0].ldc = a.c; pr[0].batch = 1;
    pr[0].scale = 1.4426950408889634f / sqrtf((float)d);      // softmax_scale * log2(e) folded into q before its rounding
    pr[0].cu_share = a.cu_share;
    pr[1].a = e;    pr[1].b = a.wk; pr[1].c = ws + cv.k;
    pr[1].m = nctx * l; pr[1].n = a.c; pr[1].k = cc;
    pr[1].lda = cc; pr[1].ldb = cc; pr[1].ldc = a.c; pr[1].batch = 1;
    pr[2].a = a.wv; pr[2].b = e;    pr[2].c = ws + cv.vt;    // V^T[b] = Wv * E_b^T  -> [c, l]
    pr[2].m = a.c; pr[2].n = l; pr[2].k = cc;
    pr[2].lda = cc; pr[2].ldb = cc; pr[2].ldc = cv.lp; pr[2].batch = nctx;
    pr[2].stride_a = 0; pr[2].stride_b = (int64_t)l * cc; pr[2].stride_c = (int64_t)a.c * cv.lp;
    if (!cross && l % 8 == 0) {
        // self-attention: the FLAT value projection x Wv^T with the transposed epilogue (AidGemmProblem.trans_rows) — the same
        // V^T, but its tiles count like the q / k projections' (the 288-row engine then runs all three in whole CU rounds)
        pr[2].a = e;    pr[2].b = a.wv;
        pr[2].m = nctx * l; pr[2].n = a.c; pr[2].k = cc;
        pr[2].lda = cc; pr[2].ldb = cc; pr[2].ldc = cv.lp; pr[2].batch = 1;
        pr[2].stride_a = 0; pr[2].stride_b = 0; pr[2].stride_c = (int64_t)a.c * cv.lp;
        pr[2].trans_rows = l;
    }
    if (folded) {                                             // x W'^T with the LayerNorm applied in the epilogue
        pr[0].b = a.ln_wq;
        pr[0].ln_stats = stats; pr[0].ln_colsum = a.ln_const; pr[0].ln_shift = a.ln_const + a.c; pr[0].ln_side = 1;
        if (!cross) {
            pr[1].b = a.ln_wk;
            pr[1].ln_stats = stats; pr[1].ln_colsum = a.ln_const + 2 * a.c; pr[1].ln_shift = a.ln_const + 3 * a.c;
            pr[1].ln_side = 1;
            pr[2].ln_stats = stats; pr[2].ln_colsum = a.ln_const + 4 * a.c; pr[2].ln_shift = a.ln_const + 5 * a.c;
            if (pr[2].trans_rows) { pr[2].b = a.ln_wv; pr[2].ln_side = 1; }
            else                  { pr[2].a = a.ln_wv; pr[2].ln_side = 2; pr[2].stride_stats = l; }
        }
    }
    int npr = 3;
    void* kip = ws + cv.kip;
    void* vtip = ws + cv.vtip;
    if (a.ip) {                                               // K_ip = to_k_ip(ip rows), V_ip^T = Wv_ip * ip_row^T
        pr[3].a = a.ip; pr[3].b = a.wk_ip; pr[3].c = kip;
        pr[3].m = a.t_ip; pr[3].n = a.c; pr[3].k = a.cc;
        pr[3].lda = a.cc; pr[3].ldb = a.cc; pr[3].ldc = a.c; pr[3].batch = a.n_ip;
        pr[3].stride_a = a.ip_stride; pr[3].stride_b = 0; pr[3].stride_c = (int64_t)a.t_ip * a.c;
        pr[4].a = a.wv_ip; pr[4].b = a.ip; pr[4].c = vtip;
        pr[4].m = a.c; pr[4].n = a.t_ip; pr[4].k = a.cc;
        pr[4].lda = a.cc; pr[4].ldb = a.cc; pr[4].ldc = cv.tp; pr[4].batch = a.n_ip;
        pr[4].stride_a = 0; pr[4].stride_b = a.ip_stride; pr[4].stride_c = (int64_t)a.c * cv.tp;
        npr = 5;
    }
    if (cached) {                                             // the query projection (and the image keys / values) only
        for (int i = 3; i < npr; ++i) pr[i - 2] = pr[i];
        npr -= 2;
        pr[0].cu_share = a.cu_share;
    }
    rc = aid_gemm_nt(pr, npr, a.dtype, stream);
    if (rc != AID_OK) return rc;

    // 2. interpolated attention core (INNER: interpolated K / V^T of the interior frames first)
    AidAttnArgs at;
    memset(&at, 0, sizeof(at));
    at.q = q; at.k = k; at.vt = vt; at.out = o;
    if (a.mode == AID_MODE_INNER) {
        // begin / end are rows of k / vt (context rows when ctx_map is given); the interpolated rows are per FRAME
        at.k2 = ws + cv.k2; at.vt2 = ws + cv.vt2;
        const int interior = a.n_frames - a.n_plain - 2;
        rc = lerp_kv_impl(k, vt, ws + cv.k2, ws + cv.vt2, a.coef, a.n_frames, a.begin, a.end, (int64_t)l * a.c,
                          (int64_t)a.c * cv.lp, a.dtype, stream, interior > 0 ? interior : 0);
        if (rc != AID_OK) return rc;
    }
    at.coef = a.coef; at.frame_scale = nullptr; at.kv_map = a.ctx_map;
    at.n_frames = a.n_frames; at.n_kv = nctx;
    at.s = a.s; at.l = l; at.heads = a.heads; at.d = d;
    at.ldq = a.c; at.ldk = a.c; at.ldvt = cv.lp; at.ldo = a.c;
    at.q_fs = (int64_t)a.s * a.c; at.k_fs = (int64_t)l * a.c; at.vt_fs = (int64_t)a.c * cv.lp; at.o_fs = (int64_t)a.s * a.c;
    at.bias = a.attn_bias; at.bias_fs = a.attn_bias_fs; at.bias_hs = a.attn_bias_hs; at.bias_rs = a.attn_bias_rs;
    at.mode = a.mode; at.fused = a.fused; at.begin = a.begin; at.end = a.end;
    at.accumulate = 0; at.dtype = a.dtype;
    at.softmax_scale = 1.0f / sqrtf((float)d);
    at.out_scale = 1.0f;
    at.n_plain = a.n_plain;
    at.q_prescaled = 1;
    at.seg_executed = a.seg_executed;
    rc = aid_attn_fwd(&at, stream);
    if (rc != AID_OK) return rc;

    // 2b. IP-Adapter image branch: a second (short) attention launch that accumulates into o
    if (a.ip) {
        AidAttnArgs ai = at;
        ai.k = kip; ai.vt = vtip; ai.k2 = nullptr; ai.vt2 = nullptr;
        ai.kv_map = a.ip_map; ai.n_kv = a.n_ip; ai.l = a.t_ip;
        ai.bias = nullptr;
        ai.ldk = a.c; ai.ldvt = cv.tp;
        ai.k_fs = (int64_t)a.t_ip * a.c; ai.vt_fs = (int64_t)a.c * cv.tp;
        ai.accumulate = 1; ai.out_scale = a.ip_scale; ai.frame_scale = a.ip_frame_scale;
        ai.seg_executed = 0;
        if (a.ip_mode == AID_IP_SAME) {                   // same scheme as the text keys (OUTER), image end-point rows
            ai.begin = a.ip_begin; ai.end = a.ip_end;
        } else {                                          // every frame with its own image keys
            ai.mode = AID_MODE_PLAIN; ai.fused = 0; ai.coef = nullptr; ai.begin = 0; ai.end = 0; ai.n_plain = 0;
        }
        rc = aid_attn_fwd(&ai, stream);
        if (rc != AID_OK) return rc;
    }

    // 3. out projection + bias
    AidGemmProblem po;
    memset(&po, 0, sizeof(po));
    po.a = o; po.b = a.wo; po.c = a.y; po.bias = a.bo; po.residual = a.residual;
    po.m = a.n_frames * a.s; po.n = a.c; po.k = a.c;
    po.lda = a.c; po.ldb = a.c; po.ldc = a.c; po.batch = 1;
    po.cu_share = a.cu_share;
    return aid_gemm_nt(&po, 1, a.dtype, stream);
}

}  // extern "C"
