#pragma once
// marex_common.hip.h -- shared by every translation unit of libmarex_hip.so: context, error / timing helpers,
// device helpers.  Hand-written gfx950 (CDNA4) kernels + C ABI for the preprocess_data hot path:
//   marex_context.hip     context, stream, timing
//   marex_synth.hip       synthetic field (bench / tests)
//   marex_shifting.hip    K_A: shifting-baseline anomaly (general kernel, fast kernel, classification)
//   marex_thresholds.hip  K_T: day-of-year thresholds (band algorithm, sliding histograms)
//   marex_mask.hip        anomaly >= threshold, transpose
//   marex_anomalies.hip   fixed baseline, digitize, detrend, std_normalise
//   marex_quantiles.hip   exact Hobday percentile, global thresholds
//   marex_tails.hip       tail extraction, thresholds and mask from tails (the default approximate-Hobday path)
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC, one object per file, linked -shared (csrc/build.py).
// -ffp-contract=off is part of the arithmetic contract (oracle/marex_oracle.py C1-C6): every float
// add / multiply / divide below is an individually rounded IEEE operation, never fused.
//
// All kernels are HBM / LDS bound streaming or counting kernels (no MFMA): wave64, 256-thread
// workgroups, one lane per grid cell so that a wave reads 256 contiguous bytes of a (time, cell) row.
#include <hip/hip_runtime.h>

#include <cmath>
#include <type_traits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "marex_hip.h"

#define NDOY MAREX_NDOY

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct TimedLaunch {
    int kid;
    hipEvent_t a, b;
};

struct marex_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool timing = false;
    std::vector<TimedLaunch> pending;
    double total_ms[MAREX_K_COUNT] = {0};
    int64_t launches[MAREX_K_COUNT] = {0};
    int* shift_info = nullptr;  // device, SHIFT_INFO_WORDS ints: which dayofyear chunks the fast anomaly kernel takes
    int* shift_plan = nullptr;  // device, [calendar years][92 chunks][8] ints: per-year records of the fast anomaly kernel
    size_t shift_plan_years = 0;
    int* shift_lplan = nullptr;  // device, [calendar years + 2][92 chunks][8] ints: lean records of k_shift_lean
    size_t shift_lplan_years = 0;
    unsigned char* thr_scratch = nullptr;  // device, per-(tile, day, lane) state bytes of the 1024-thread threshold tiles
    size_t thr_scratch_bytes = 0;
    unsigned char* detrend_scratch = nullptr;  // device, partial sums / coefficients / means of the detrend reductions
    size_t detrend_scratch_bytes = 0;
    unsigned char* morph_scratch = nullptr;  // device, two bit-packed padded images of the morphology passes
    size_t morph_scratch_bytes = 0;
    int n_cu = 0;  // compute units of the device (queried on first use)
    std::map<std::string, int> opts;          // tuning / diagnostic options (marex_set_option; seeded from MAREX_* at creation)
    long long* row_off = nullptr;             // device, byte offset of every kept row in dayofyear order (tail extraction)
    size_t row_off_bytes = 0;
    long long* row_off_mask = nullptr;        // the same for the one-byte-per-cell mask array (mask from tails)
    size_t row_off_mask_bytes = 0;
    unsigned long long* dbg_counters = nullptr;  // device, MAREX_DBG_COUNTERS event counters of the tail kernels
};

#define MAREX_DBG_COUNTERS 8

// Option lookup: options are set through marex_set_option (or, once, from the environment variables MAREX_<NAME> that
// exist when the context is created); nothing reads the environment at launch time.
static inline int ctx_opt(marex_ctx* ctx, const char* name, int dflt) {
    auto it = ctx->opts.find(name);
    return it == ctx->opts.end() ? dflt : it->second;
}

#ifdef MAREX_ABLATION
#define MAREX_ABLATE_OPT(ctx, name) ctx_opt(ctx, name, 0)
#else
#define MAREX_ABLATE_OPT(ctx, name) 0
#endif

static inline unsigned long long* ctx_debug_counters(marex_ctx* ctx) {
    if (!ctx->dbg_counters) {
        if (hipMalloc((void**)&ctx->dbg_counters, MAREX_DBG_COUNTERS * sizeof(unsigned long long)) != hipSuccess) return nullptr;
        (void)hipMemsetAsync(ctx->dbg_counters, 0, MAREX_DBG_COUNTERS * sizeof(unsigned long long), ctx->stream);
    }
    return ctx->dbg_counters;
}

static inline int device_cus(marex_ctx* ctx) {
    if (ctx->n_cu <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || n <= 0) n = 256;
        ctx->n_cu = n;
    }
    return ctx->n_cu;
}

static inline int fail(marex_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) return fail(ctx, -2, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

struct LaunchTimer {
    marex_ctx* ctx;
    int kid;
    hipEvent_t a = nullptr, b = nullptr;
    LaunchTimer(marex_ctx* c, int k) : ctx(c), kid(k) {
        if (ctx->timing) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a, ctx->stream);
        }
    }
    ~LaunchTimer() {
        if (ctx->timing) {
            (void)hipEventRecord(b, ctx->stream);
            ctx->pending.push_back({kid, a, b});
        }
    }
};

static inline void drain_timers(marex_ctx* ctx) {
    for (auto& p : ctx->pending) {
        float ms = 0.f;
        (void)hipEventSynchronize(p.b);
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->total_ms[p.kid] += ms;
            ctx->launches[p.kid] += 1;
        }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    ctx->pending.clear();
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float nan_f() { return __builtin_nanf(""); }
__device__ __forceinline__ bool finite_f(float v) { return fabsf(v) <= 3.402823466e+38f; }

// Workgroups are dealt round-robin over the 8 XCDs (b % 8).  Map the linear block id so that the
// `inner` consecutive work items of one `outer` group (which re-read each other's rows) land on the
// same XCD / L2.  Returns false for padding blocks.
__device__ __forceinline__ bool xcd_swizzle(unsigned b, int n_outer, int n_inner, int& outer, int& inner) {
    unsigned xcd = b & 7u;
    unsigned k = b >> 3;
    inner = (int)(k % (unsigned)n_inner);
    outer = (int)((k / (unsigned)n_inner) * 8u + xcd);
    return outer < n_outer;
}
static inline unsigned xcd_grid(int n_outer, int n_inner) { return (unsigned)(((n_outer + 7) / 8) * 8) * (unsigned)n_inner; }

// Bin matrix layout: blocks of 16 consecutive cells, inside a block the dayofyear-sorted rows are
// contiguous: element (row r, cell c) at ((c >> 4) * T_out + r) * 16 + (c & 15).  The threshold kernel's
// 16-cell-wide tile rows then read whole contiguous runs (a tile row's entire day window) instead of
// 32 bytes out of every 128-byte line of a row-major [T_out, C] matrix (measured: 17x over-fetch).
__device__ __forceinline__ size_t bins_index(long r, long c, long T_out) {
    return ((size_t)(c >> 4) * (size_t)T_out + (size_t)r) * 16 + (size_t)(c & 15);
}

// np.digitize(v, edges) - 1 for an increasing table edges[0..nb] with edges[0] = -inf  (contract C4).
// The guess assumes equal-width bins above edges[1]; the two correction loops make it exact for any
// increasing table.
__device__ __forceinline__ int digitize_bin(float v, const float* e, int nb, float inv_width) {
    // straight-line: NaN / out-of-range inputs run through with a clamped guess and are fixed by selects at the end
    // the guess is clamped as a FLOAT: converting +-inf / huge values to int and adding 1 is signed-overflow UB, which
    // the compiler turned into an unclamped index (NaN clamps to 0)
    int k = 1 + (int)fminf(fmaxf((v - e[1]) * inv_width, 0.0f), (float)(nb - 2));
    k = k < 1 ? 1 : (k > nb - 1 ? nb - 1 : k);
    k += (v >= e[k + 1]) - (v < e[k]);  // the guess is off by at most one for equal-width tables
    k = k < 1 ? 1 : (k > nb - 1 ? nb - 1 : k);
    const bool inside = v >= e[1] && v < e[nb];
    if (inside && !(v >= e[k] && v < e[k + 1])) {  // arbitrary increasing tables: walk (never taken for uniform bins)
        while (k > 1 && v < e[k]) --k;
        while (k < nb - 1 && v >= e[k + 1]) ++k;
    }
    k = v < e[1] ? 0 : k;
    return (v >= e[nb] || !(v == v)) ? nb : k;
}

// Same result without touching memory, for tables that equal NumPy's float32 arange bit for bit:
// edges[j] = fl32(first + fl32(fl32(j-1) * delta)) for j >= 1 (separately rounded multiply and add, which is what
// -ffp-contract=off compiles to).  Whether a table has that form is checked once per workgroup (edges_are_arange).
__device__ __forceinline__ float arange_edge(int j, float first, float delta) { return first + (float)(j - 1) * delta; }

__device__ __forceinline__ int digitize_arange(float v, float first, float delta, float e_last, int nb, float inv_width) {
    int k = 1 + (int)fminf(fmaxf((v - first) * inv_width, 0.0f), (float)(nb - 2));  // float clamp: see digitize_bin
    k = k < 1 ? 1 : (k > nb - 1 ? nb - 1 : k);
    k += (v >= arange_edge(k + 1, first, delta)) - (v < arange_edge(k, first, delta));
    k = k < 1 ? 1 : (k > nb - 1 ? nb - 1 : k);
    const bool inside = v >= first && v < e_last;
    if (inside && !(v >= arange_edge(k, first, delta) && v < arange_edge(k + 1, first, delta))) {
        while (k > 1 && v < arange_edge(k, first, delta)) --k;
        while (k < nb - 1 && v >= arange_edge(k + 1, first, delta)) ++k;
    }
    k = v < first ? 0 : k;
    return (v >= e_last || !(v == v)) ? nb : k;
}

// block-wide: does the LDS copy e[0..nb] of the edge table have the arange form?  (all threads must call)
__device__ __forceinline__ bool edges_are_arange(const float* e, int nb) {
    const float first = e[1], delta = e[2] - e[1];
    bool ok = delta > 0.f;
    for (int j = 1 + (int)threadIdx.x; j <= nb; j += (int)blockDim.x)
        ok = ok && (__float_as_uint(e[j]) == __float_as_uint(arange_edge(j, first, delta)));
    return __syncthreads_and(ok ? 1 : 0) != 0;
}

__device__ __forceinline__ unsigned ordered_key(float v) {
    unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ unsigned hist_get(const unsigned* h, int b, int lane) {
    return (h[(b >> 1) * 64 + lane] >> ((b & 1) * 16)) & 0xFFFFu;  // [dword][64 lanes], two uint16 counters per dword
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float key_to_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// min / max of an int over the 64 lanes of the wave, in every lane (DPP inside the rows of 16, two cross-row shuffles): one LDS
// atomic per WAVE instead of one per lane on a shared address (64 lanes on one address are 64 serial LDS operations)
__device__ __forceinline__ int wave_min_i32(int m) {
    int o;
    o = __builtin_amdgcn_update_dpp(m, m, 0x128, 0xF, 0xF, false); m = o < m ? o : m;  // row_ror 8 / 4 / 2 / 1
    o = __builtin_amdgcn_update_dpp(m, m, 0x124, 0xF, 0xF, false); m = o < m ? o : m;
    o = __builtin_amdgcn_update_dpp(m, m, 0x122, 0xF, 0xF, false); m = o < m ? o : m;
    o = __builtin_amdgcn_update_dpp(m, m, 0x121, 0xF, 0xF, false); m = o < m ? o : m;
    o = __shfl_xor(m, 16, 64); m = o < m ? o : m;
    o = __shfl_xor(m, 32, 64); m = o < m ? o : m;
    return m;
}
__device__ __forceinline__ int wave_max_i32(int m) { return -wave_min_i32(-m); }

#define MASK_DOY_CHUNKS 6  // default number of pieces the dayofyear axis is cut into by the doy-grouped streaming kernels
